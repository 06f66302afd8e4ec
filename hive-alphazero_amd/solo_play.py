"""HivePlayer: the reference-sequential search behind the drop-in API of woker/solo_play.py.

Same constructor, attributes and `action(env) -> (move, [policy list, visit total])` contract as the
reference's HivePlayer (woker/solo_play.py:69-384), and the same search process: a transposition
table keyed by env.state_key, one evaluator call per simulation, virtual loss, fresh Dirichlet
noise on the root priors at every simulation, visit-count policy with the "all W negative ->
priors" fallback, argmax temperature.  With SEARCH_THREADS = 1 and a seeded numpy it reproduces the
reference's visit counts, W sums, priors and chosen move bit for bit (tests/test_mcts_golden.py).

The implementation is this repo's own: each table entry keeps its edges in flat numpy arrays
(struct-of-arrays, like the GPU search in csrc/hive_search.hip), a simulation is an explicit
descent/backup over a path stack instead of a recursion, and PUCT is one vectorised expression.
Arithmetic is arranged so that every floating-point operation happens in the reference's order and
dtype (float32 priors, float64 scores), and the numpy RNG is consumed by the same two calls.

It is env-agnostic (anything with the GamePlay API) and talks to the evaluator through the pipe
protocol (send(planes) / recv() -> (p[1584], v)).  The throughput path is hive_alphazero_amd.mcts.
"""
from collections import defaultdict
from concurrent.futures import ThreadPoolExecutor
from copy import deepcopy
from threading import Lock

import numpy as np

from . import config
from .config import ACTION_SPACE, MAX_GAME_LENGTH, PIECE_BLACK, PIECE_WHITE

# woker/solo_play.py:23-30 (module-level knobs, same names so callers can override them the same way)
simulation_num_per_move = 100
tau_decay_rate = 0.01
c_puct = 0.7
dirichlet_alpha = 0.3
noise_eps = 0.25
virtual_loss = 1
SEARCH_THREADS = config.SEARCH_THREADS

_DRAW = 5          # sentinel a drawn / length-capped line returns (solo_play.py:180-183)


class _Entry:
    """One position of the transposition table: raw network priors until its first selection, then
    per-edge arrays in legal-move order."""
    __slots__ = ("raw_p", "moves", "n", "w", "p", "sum_n")

    def __init__(self, raw_p):
        self.raw_p = raw_p
        self.moves = None
        self.n = self.w = self.p = None
        self.sum_n = 0

    def open_edges(self, legal):
        """First selection at this position: priors of the legal moves, renormalised with the reference's
        running float32 total that starts at 1e-8 (solo_play.py:304-313)."""
        prior = np.asarray(self.raw_p)[legal]
        running = np.cumsum(np.concatenate((np.asarray([1e-8], dtype=prior.dtype), prior)), dtype=prior.dtype)
        self.moves = list(legal)
        self.p = prior / running[-1]
        self.n = np.zeros(len(legal), dtype=np.int64)
        self.w = np.zeros(len(legal), dtype=np.float64)
        self.raw_p = None

    def q(self):
        with np.errstate(divide="ignore", invalid="ignore"):
            return np.where(self.n > 0, self.w / self.n, 0.0)


class _EdgeView:
    """Read-only per-edge record (n, w, q, p) for callers that inspect `player.tree[key].a`."""
    __slots__ = ("n", "w", "q", "p")

    def __init__(self, n, w, p):
        self.n, self.w, self.p = int(n), float(w), p
        self.q = self.w / self.n if self.n else 0


class _EntryView:
    def __init__(self, entry):
        self.sum_n = entry.sum_n if entry is not None else 0
        self.a = {}
        if entry is not None and entry.moves is not None:
            for i, m in enumerate(entry.moves):
                self.a[m] = _EdgeView(entry.n[i], entry.w[i], entry.p[i])


class _TreeView:
    """`player.tree[state_key]` -> object with `.a` (move -> edge record) and `.sum_n`, `len(player.tree)`."""

    def __init__(self, table):
        self._t = table

    def __getitem__(self, key):
        return _EntryView(self._t.get(key))

    def __contains__(self, key):
        return key in self._t

    def __len__(self):
        return len(self._t)


def _terminal_value(env):
    """Value of a finished position from the side to move, or the draw sentinel (solo_play.py:169-183)."""
    if env.game_is_over():
        winner = env.state.winner
        mover_is_white = env.state.player() == 0
        if winner == PIECE_WHITE:
            return 1 if mover_is_white else -1
        if winner == PIECE_BLACK:
            return -1 if mover_is_white else 1
        return _DRAW
    if env.state.turn >= MAX_GAME_LENGTH:
        return _DRAW
    return None


class HivePlayer:
    def __init__(self, pipes=None, reward=False):
        self.moves = []
        self.pipe_pool = pipes
        self.none_queue = True
        self.net = None
        self.simulation_num_per_move = simulation_num_per_move
        self.reward = reward
        self.main_key_state = None
        self.max_depth = None
        self._table = {}
        self._locks = defaultdict(Lock)

    # ------------------------------------------------------------------ reference surface
    @property
    def tree(self):
        return _TreeView(self._table)

    def reset(self):
        self._table = {}

    def action(self, env, non_queue=True):
        self.reset()
        self.max_depth = env.state.turn
        self.main_key_state = env.state_key
        self.search_moves(env)
        policy, visits = self.calc_policy(env)
        probs = self.apply_temperature(policy, int(env.state.turn + 1) / 2)
        move = int(np.random.choice(range(ACTION_SPACE), p=probs))
        return move, [list(policy), visits]

    def search_moves(self, env):
        if self.none_queue and SEARCH_THREADS > 1:
            with ThreadPoolExecutor(max_workers=SEARCH_THREADS) as pool:
                jobs = [pool.submit(self.search_my_move, deepcopy(env), True)
                        for _ in range(self.simulation_num_per_move)]
            values = [j.result() for j in jobs]
        else:
            values = [self.search_my_move(deepcopy(env), True) for _ in range(self.simulation_num_per_move)]
        return np.max(values), values[0]

    def search_my_move(self, env, is_root_node=False):
        """One simulation from `env` (consumed): returns the value seen from its side to move."""
        path = []                      # (entry, edge index) from the root down
        outcome = None
        depth = 0
        while True:
            outcome = _terminal_value(env)
            if outcome is not None:
                break
            key = env.state_key
            with self._locks[key]:
                entry = self._table.get(key)
                if entry is None:
                    leaf_p, leaf_v = self._evaluate(env)
                    self._table[key] = _Entry(leaf_p)
                    outcome = leaf_v
                    break
                edge, move = self._select(entry, env, is_root_node and depth == 0)
                entry.sum_n += virtual_loss                 # virtual loss (solo_play.py:205-208)
                entry.n[edge] += virtual_loss
                entry.w[edge] -= virtual_loss
            path.append((key, entry, edge))
            env.move(move)
            depth += 1
            if env.state.turn > self.max_depth:
                self.max_depth = env.state.turn
        # backup, deepest edge first; a drawn line scores -1 at every ply and stays a draw (solo_play.py:217-247)
        value = outcome
        for key, entry, edge in reversed(path):
            drawn = value == _DRAW
            seen = -1 if drawn else -value
            with self._locks[key]:
                entry.sum_n += -virtual_loss + 1
                entry.n[edge] += -virtual_loss + 1
                entry.w[edge] += virtual_loss + seen
            value = _DRAW if drawn else seen
        return value

    def expand_and_evaluate(self, env):
        return self.predict(env.encode_board())

    def expand_and_evaluate_with_net(self, env):
        import torch
        planes = np.ascontiguousarray(env.encode_board().transpose(2, 0, 1))
        dev = next(self.net.parameters()).device
        with torch.no_grad():
            p, v = self.net(torch.from_numpy(planes).float().to(dev).unsqueeze(0))
        return p.cpu().numpy().reshape(-1), v.cpu().numpy().reshape(-1)

    def _evaluate(self, env):
        return self.expand_and_evaluate(env) if self.none_queue else self.expand_and_evaluate_with_net(env)

    def predict(self, board_state):
        pipe = self.pipe_pool.pop()
        pipe.send(board_state)
        answer = pipe.recv()
        self.pipe_pool.append(pipe)
        return answer

    def select_action_q_and_u(self, env, is_root_node):
        """PUCT choice at env's position (solo_play.py:294-335); -1 when there is no legal move."""
        entry = self._table[env.state_key]
        return self._select(entry, env, is_root_node)[1]

    def _select(self, entry, env, at_root):
        legal = env.actions()
        if len(legal) == 0:
            if entry.moves is None:                          # the pass edge (solo_play.py:299-300)
                entry.moves, entry.raw_p = [-1], None
                entry.n, entry.w = np.zeros(1, dtype=np.int64), np.zeros(1, dtype=np.float64)
                entry.p = np.zeros(1, dtype=np.float64)
            return 0, -1
        if entry.moves is None:
            entry.open_edges(legal)
        prior = entry.p
        if at_root:
            noise = np.random.dirichlet([dirichlet_alpha] * len(entry.moves))
            prior = (1 - noise_eps) * prior + noise_eps * noise
        score = entry.q() + c_puct * prior * np.sqrt(entry.sum_n + 1) / (1 + entry.n)
        edge = int(np.argmax(score))
        return edge, entry.moves[edge]

    def apply_temperature(self, policy, turn):
        tau = np.power(tau_decay_rate, turn)
        if tau < 0.1:                                        # always true for turn >= 1: greedy
            greedy = np.zeros(ACTION_SPACE)
            greedy[np.argmax(policy)] = 1.0
            return greedy
        sharpened = np.power(policy, 1 / tau)
        return sharpened / np.sum(sharpened)

    def calc_policy(self, env):
        """Visit distribution at the root; the priors when every W is negative (solo_play.py:351-374)."""
        entry = self._table[env.state_key]
        visits = np.zeros(ACTION_SPACE)
        priors = np.zeros(ACTION_SPACE)
        if entry.moves is not None:
            for i, m in enumerate(entry.moves):
                visits[m] = entry.n[i]
                priors[m] = entry.p[i]
        total = np.sum(visits)
        policy = visits / np.sum(visits)
        if entry.moves is not None and np.max(entry.w) < 0:
            policy = priors
        return policy, total

    def finish_game(self, z):
        for move in self.moves:
            move += [z]
