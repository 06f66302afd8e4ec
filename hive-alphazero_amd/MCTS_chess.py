"""UCT_search / get_policy: the reference's alpha_zero/MCTS_chess.py search (:24-161) run by the GPU tree kernels.

The reference grows a tree of Python `UCTNode` objects (three 1584-wide float32 vectors per node, one
`deepcopy(game)` + `move` per new child, one batch-1 network call per read).  Here the tree is the flat node
pool of `csrc/hive_search.hip` in its `HIVE_SEARCH_UCT` mode -- Q = W/(1+N), U = sqrt(N_node)|P|/(1+N), illegal
priors zeroed without renormalisation, no noise, plain tree, +v / -v by the colour that played the edge -- and
every read is three kernel launches + one network call for ALL searched positions at once.  `UCT_search` keeps the
reference's signature for one game; `uct_search_batch` is the same search over many positions in lock step.

What comes back in place of the root `UCTNode` is a `UCTRoot`: the arrays callers read from it
(`child_number_visits`, `child_total_value`, `child_priors`, `action_idxes`, `game`) with the reference's dtypes.
The stale `MCTS_self_play` (:177-220, calls `GamePlay()` without its required arguments) is not provided.
"""
import numpy as np
import torch

from . import mcts
from .alpha_net import InferenceNet
from .config import ACTION_SPACE


class UCTRoot:
    """Root statistics of one finished search, shaped like the reference's root node (MCTS_chess.py:24-38)."""
    move = None
    parent = None

    def __init__(self, game, visits, total_value, priors, reads):
        self.game = game
        self.child_number_visits = visits              # float32[1584], like the reference
        self.child_total_value = total_value
        self.child_priors = priors
        self.action_idxes = [int(a) for a in np.flatnonzero(priors)] if game is None else list(game.actions())
        self.is_expanded = len(self.action_idxes) != 0
        self.number_visits = float(reads)              # every read backs up through the root (MCTS_chess.py:111-119)


def _as_evaluator(net):
    """(planes [B,12,12,56] on the GPU) -> (p [B,1584], v [B]) for a ChessNet (evaluated NCHW fp32 exactly like
    MCTS_chess.py:133-141), an InferenceNet (channels-last, its own dtype) or any callable with that signature."""
    if isinstance(net, InferenceNet):
        return net, net.dtype if hasattr(net, "dtype") else torch.float32
    if isinstance(net, torch.nn.Module):
        def run(planes):
            with torch.no_grad():
                p, v = net(planes.permute(0, 3, 1, 2).float().contiguous())
            return p, v.reshape(-1)
        return run, torch.float32
    return net, torch.float32


def uct_search_batch(boards, hist, num_reads, net, device=None, slots=1, plane_dtype=None):
    """`num_reads` reads of UCT_search from every position of a batch.

    boards / hist: uint8 [G,64] / [G,384] HiveBoard / HiveHistory records (BoardBatch.export_state).
    -> (best int64[G] = argmax visits with the lowest action id winning ties (np.argmax, MCTS_chess.py:151),
        visits, total_value, priors: float32 [G,1584] on the GPU)."""
    evaluator, dt = _as_evaluator(net)
    boards = torch.as_tensor(boards)
    G = boards.shape[0]
    ts = mcts.TreeSearch(G, num_reads, evaluator, device=device, slots=slots, plane_dtype=plane_dtype or dt, mode=mcts.UCT)
    try:
        dev = ts.device
        ts.search(boards.to(dev).contiguous(), torch.as_tensor(hist).to(dev).contiguous())
        visits, total_value, priors = ts.root_stats()
        best = torch.argmax(visits, dim=1)             # first maximum, like np.argmax
        torch.cuda.synchronize(dev)
    finally:
        ts.close()
    return best, visits, total_value, priors


def UCT_search(game_state, num_reads, net):
    """MCTS_chess.py:130-151 for one GamePlay: -> (best action id, root statistics, None)."""
    rec = torch.from_numpy(np.ascontiguousarray(game_state._rec).reshape(1, 64))
    hist = torch.from_numpy(np.ascontiguousarray(game_state._hist).reshape(1, 384))
    best, visits, total_value, priors = uct_search_batch(rec, hist, num_reads, net, device=getattr(game_state, "_device", None))
    root = UCTRoot(game_state, visits[0].cpu().numpy(), total_value[0].cpu().numpy(), priors[0].cpu().numpy(), num_reads)
    return int(best[0].item()), root, None


def do_decode_n_move_pieces(board, move):
    """MCTS_chess.py:153-155."""
    board.move(move)
    return board


def get_policy(root):
    """MCTS_chess.py:157-161: visit counts over their sum (float32), zero where an action was never tried."""
    n = np.asarray(root.child_number_visits, dtype=np.float32)
    total = n.sum()
    out = np.zeros(ACTION_SPACE, dtype=np.float32)
    np.divide(n, total, out=out, where=n != 0)
    return out
