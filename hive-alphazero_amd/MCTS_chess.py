"""UCTNode / UCT_search / get_policy: mirror of the reference's alpha_zero/MCTS_chess.py (:24-161).

The array tree (three 1584-wide float32 vectors per node) and its arithmetic are kept as in the
reference; positions are hive_alphazero_amd.env_hive.GamePlay objects, i.e. every move / legal-move
/ planes / game-over question is answered by the HIP kernels.  The reference hard-codes `.cuda()`
for the evaluation (:138); here the planes go to whatever device the net lives on, and `net` may be
a ChessNet (NCHW input like the reference) or an alpha_net.InferenceNet.
The stale MCTS_self_play (:177-220, calls GamePlay() without its required arguments) is not mirrored.
"""
import collections
import copy
import math

import numpy as np
import torch

from .alpha_net import InferenceNet
from .config import MAX_MAP_FULL

ACTIONS = MAX_MAP_FULL * MAX_MAP_FULL * 11


class UCTNode:
    def __init__(self, game, move, parent=None):
        self.game = game
        self.move = move
        self.is_expanded = False
        self.parent = parent
        self.children = {}
        self.child_priors = np.zeros([ACTIONS], dtype=np.float32)
        self.child_total_value = np.zeros([ACTIONS], dtype=np.float32)
        self.child_number_visits = np.zeros([ACTIONS], dtype=np.float32)
        self.action_idxes = []
        self.debug = None

    @property
    def number_visits(self):
        return self.parent.child_number_visits[self.move]

    @number_visits.setter
    def number_visits(self, value):
        self.parent.child_number_visits[self.move] = value

    @property
    def total_value(self):
        return self.parent.child_total_value[self.move]

    @total_value.setter
    def total_value(self, value):
        self.parent.child_total_value[self.move] = value

    def child_Q(self):                                   # MCTS_chess.py:52-53
        return self.child_total_value / (1 + self.child_number_visits)

    def child_U(self):                                   # MCTS_chess.py:55-57
        return math.sqrt(self.number_visits) * (abs(self.child_priors) / (1 + self.child_number_visits))

    def best_child(self):                                # MCTS_chess.py:58-64
        if len(self.action_idxes) != 0:
            bestmove = self.child_Q() + self.child_U()
            bestmove = self.action_idxes[np.argmax(bestmove[self.action_idxes])]
        else:
            bestmove = np.argmax(self.child_Q() + self.child_U())
        return bestmove

    def select_leaf(self):                               # MCTS_chess.py:66-73
        current = self
        while current.is_expanded:
            current = current.maybe_add_child(current.best_child())
        return current

    def add_dirichlet_noise(self, action_idxs, child_priors):     # MCTS_chess.py:75-79 (unused upstream too)
        valid = child_priors[action_idxs]
        valid = 0.75 * valid + 0.25 * np.random.dirichlet(np.zeros([len(valid)], dtype=np.float32) + 0.3)
        child_priors[action_idxs] = valid
        return child_priors

    def expand(self, child_priors):                      # MCTS_chess.py:81-95
        self.is_expanded = True
        action_idxs = self.game.actions()
        if len(action_idxs) == 0:
            self.debug = self.game
            self.is_expanded = False
        self.action_idxes = action_idxs
        c_p = child_priors
        legal = np.zeros(len(c_p), dtype=bool)
        legal[action_idxs] = True
        c_p[~legal] = 0.0                                # mask all illegal actions, no renormalisation
        self.child_priors = c_p

    def decode_n_move_pieces(self, board, move):
        board.move(move)
        return board

    def maybe_add_child(self, move):                     # MCTS_chess.py:102-109
        if move not in self.children:
            copy_board = copy.deepcopy(self.game)
            copy_board = self.decode_n_move_pieces(copy_board, move)
            self.children[move] = UCTNode(copy_board, move, parent=self)
        return self.children[move]

    def backup(self, value_estimate):                    # MCTS_chess.py:111-119
        current = self
        while current.parent is not None:
            current.number_visits += 1
            if current.game.player() == 1:
                current.total_value += (1 * value_estimate)
            elif current.game.player() == 0:
                current.total_value += (-1 * value_estimate)
            current = current.parent


class DummyNode(object):                                 # MCTS_chess.py:122-126
    def __init__(self):
        self.parent = None
        self.child_total_value = collections.defaultdict(float)
        self.child_number_visits = collections.defaultdict(float)


def _evaluate(net, planes_hwc):
    """planes [12,12,56] float64 (GamePlay.encode_board) -> (p float32[1584], v float)."""
    if isinstance(net, InferenceNet):
        x = torch.from_numpy(np.ascontiguousarray(planes_hwc, dtype=np.float32)).to(net.device).unsqueeze(0)
        p, v = net(x)
    else:
        dev = next(net.parameters()).device
        x = torch.from_numpy(np.ascontiguousarray(planes_hwc.transpose(2, 0, 1), dtype=np.float32)).to(dev).unsqueeze(0)
        with torch.no_grad():
            p, v = net(x)
    return p.detach().float().cpu().numpy().reshape(-1), float(v.reshape(-1)[0].item())


def UCT_search(game_state, num_reads, net):              # MCTS_chess.py:130-151
    root = UCTNode(game_state, move=None, parent=DummyNode())
    for _ in range(num_reads):
        leaf = root.select_leaf()
        child_priors, value_estimate = _evaluate(net, leaf.game.encode_board())
        if leaf.game.game_is_over():
            leaf.backup(value_estimate)
            continue
        leaf.expand(child_priors)
        leaf.backup(value_estimate)
    return np.argmax(root.child_number_visits), root, None


def do_decode_n_move_pieces(board, move):                # MCTS_chess.py:153-155
    board.move(move)
    return board


def get_policy(root):                                    # MCTS_chess.py:157-161
    policy = np.zeros([ACTIONS], dtype=np.float32)
    for idx in np.where(root.child_number_visits != 0)[0]:
        policy[idx] = root.child_number_visits[idx] / root.child_number_visits.sum()
    return policy
