"""HivePlayer: mirror of the reference's woker/solo_play.py::HivePlayer (:69-384).

This is the reference-sequential search: a dict tree keyed by env.state_key, virtual loss,
per-simulation root Dirichlet noise, one evaluator call per simulation, exactly the
arithmetic (and numpy RNG consumption order) of the reference, so that with
SEARCH_THREADS = 1 and a seeded numpy it reproduces the reference's visit counts bit for bit
(tests/test_mcts_golden.py).  It is env-agnostic (anything with the GamePlay API) and talks to
the evaluator through the same pipe protocol (send(planes) / recv() -> (p[1584], v)).

The throughput path -- thousands of trees searched concurrently on the GPU -- is
hive_alphazero_amd.mcts (flat SoA tree + HIP kernels); this class is the drop-in surface
for woker/self_play.py.
"""
from collections import defaultdict
from concurrent.futures import ThreadPoolExecutor
from copy import deepcopy
from threading import Lock

import numpy as np

from . import config
from .config import ACTION_SPACE, MAX_GAME_LENGTH, PIECE_BLACK, PIECE_WHITE

# solo_play.py:23-30
simulation_num_per_move = 100
tau_decay_rate = 0.01
c_puct = 0.7
dirichlet_alpha = 0.3
noise_eps = 0.25
virtual_loss = 1
SEARCH_THREADS = config.SEARCH_THREADS


class VisitStats:                      # solo_play.py:33-46
    def __init__(self):
        self.a = defaultdict(ActionStats)
        self.sum_n = 0
        self.actions = []


class ActionStats:                     # solo_play.py:48-66
    def __init__(self):
        self.n = 0
        self.w = 0
        self.q = 0
        self.p = 0


class HivePlayer:
    def __init__(self, pipes=None, reward=False):
        self.moves = []
        self.tree = defaultdict(VisitStats)
        self.pipe_pool = pipes
        self.node_lock = defaultdict(Lock)
        self.none_queue = True
        self.net = None
        self.simulation_num_per_move = simulation_num_per_move
        self.reward = reward
        self.main_key_state = None
        self.max_depth = None
        self.verbose = False

    def reset(self):
        self.tree = defaultdict(VisitStats)

    def action(self, env, non_queue=True):                      # solo_play.py:110-151
        self.reset()
        self.max_depth = env.state.turn
        self.main_key_state = env.state_key
        self.search_moves(env)
        policy, sum_all = self.calc_policy(env)
        p = self.apply_temperature(policy, int(env.state.turn + 1) / 2)
        my_action = int(np.random.choice(range(ACTION_SPACE), p=p))
        if self.verbose:
            print("MAX DEPTH ", self.max_depth)
            print(env.decode_action(my_action), env.state.turn)
        return my_action, [list(policy), sum_all]

    def search_moves(self, env):                                # solo_play.py:153-165
        if self.none_queue:
            futures = []
            with ThreadPoolExecutor(max_workers=SEARCH_THREADS) as executor:
                for _ in range(self.simulation_num_per_move):
                    futures.append(executor.submit(self.search_my_move, deepcopy(env), is_root_node=True))
            vals = [f.result() for f in futures]
        else:
            vals = [self.search_my_move(deepcopy(env), is_root_node=True)
                    for _ in range(self.simulation_num_per_move)]
        return np.max(vals), vals[0]

    def search_my_move(self, env, is_root_node=False):          # solo_play.py:167-247
        if env.game_is_over():
            if env.state.player() == 0:
                if env.state.winner == PIECE_WHITE:
                    return 1
                elif env.state.winner == PIECE_BLACK:
                    return -1
            else:
                if env.state.winner == PIECE_WHITE:
                    return -1
                elif env.state.winner == PIECE_BLACK:
                    return 1
            return 5
        elif env.state.turn >= MAX_GAME_LENGTH:
            return 5

        state = env.state_key
        with self.node_lock[state]:
            if state not in self.tree:
                if self.none_queue:
                    leaf_p, leaf_v = self.expand_and_evaluate(env)
                else:
                    leaf_p, leaf_v = self.expand_and_evaluate_with_net(env)
                self.tree[state].p = leaf_p
                return leaf_v

            action_t = self.select_action_q_and_u(env, is_root_node)
            my_visit_stats = self.tree[state]
            my_stats = my_visit_stats.a[action_t]
            my_visit_stats.sum_n += virtual_loss
            my_stats.n += virtual_loss
            my_stats.w += -virtual_loss
            my_stats.q = my_stats.w / my_stats.n

        env.move(action_t)
        if env.state.turn > self.max_depth:
            self.max_depth = env.state.turn
        leaf_v = self.search_my_move(env)

        reach_max = False
        if leaf_v == 5:
            leaf_v = 1
            reach_max = True
        leaf_v = -leaf_v

        with self.node_lock[state]:
            my_visit_stats.sum_n += -virtual_loss + 1
            my_stats.n += -virtual_loss + 1
            my_stats.w += virtual_loss + leaf_v
            my_stats.q = my_stats.w / my_stats.n

        if reach_max:
            leaf_v = 5
        return leaf_v

    def expand_and_evaluate_with_net(self, env):                # solo_play.py:249-258
        import torch
        board_state = env.encode_board().transpose(2, 0, 1)
        dev = next(self.net.parameters()).device
        board_state = torch.from_numpy(np.ascontiguousarray(board_state)).float().to(dev).unsqueeze(0)
        leaf_p, leaf_v = self.net(board_state)
        return leaf_p.detach().cpu().numpy().reshape(-1), leaf_v.detach().cpu().numpy().reshape(-1)

    def expand_and_evaluate(self, env):                         # solo_play.py:260-278
        board_state = env.encode_board()
        return self.predict(board_state)

    def predict(self, board_state):                             # solo_play.py:280-291
        pipe = self.pipe_pool.pop()
        pipe.send(board_state)
        ret = pipe.recv()
        self.pipe_pool.append(pipe)
        return ret

    def select_action_q_and_u(self, env, is_root_node):         # solo_play.py:294-335
        state = env.state_key
        actions = env.actions()
        if len(actions) == 0:
            return -1
        my_visitstats = self.tree[state]
        if my_visitstats.p is not None:
            tot_p = 1e-8
            for mov in actions:
                mov_p = my_visitstats.p[mov]
                my_visitstats.a[mov].p = mov_p
                tot_p += mov_p
            for a_s in my_visitstats.a.values():
                a_s.p /= tot_p
            my_visitstats.p = None

        xx_ = np.sqrt(my_visitstats.sum_n + 1)
        e = noise_eps
        dir_alpha = dirichlet_alpha
        best_s = -999
        best_a = None
        if is_root_node:
            noise = np.random.dirichlet([dir_alpha] * len(my_visitstats.a))
        i = 0
        for action, a_s in my_visitstats.a.items():
            p_ = a_s.p
            if is_root_node:
                p_ = (1 - e) * p_ + e * noise[i]
                i += 1
            b = a_s.q + c_puct * p_ * xx_ / (1 + a_s.n)
            if b > best_s:
                best_s = b
                best_a = action
        return best_a

    def apply_temperature(self, policy, turn):                  # solo_play.py:337-349
        tau = np.power(tau_decay_rate, turn)
        if tau < 0.1:
            tau = 0
        if tau == 0:
            action = np.argmax(policy)
            ret = np.zeros(ACTION_SPACE)
            ret[action] = 1.0
            return ret
        ret = np.power(policy, 1 / tau)
        ret /= np.sum(ret)
        return ret

    def calc_policy(self, env):                                 # solo_play.py:351-374
        state = env.state_key
        my_visitstats = self.tree[state]
        policy = np.zeros(ACTION_SPACE)
        policy_t = np.zeros(ACTION_SPACE)
        w = []
        for action, a_s in my_visitstats.a.items():
            policy[action] = a_s.n
            policy_t[action] = a_s.p
            w.append(a_s.w)
        sum_all = np.sum(policy)
        policy /= np.sum(policy)
        if np.max(w) < 0:
            policy = policy_t
        return policy, sum_all

    def finish_game(self, z):                                   # solo_play.py:376-384
        for move in self.moves:
            move += [z]
