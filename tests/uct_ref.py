"""Sequential restatement of the reference's array-tree search -- TEST INFRASTRUCTURE (checker only).

alpha_zero/MCTS_chess.py::UCTNode + UCT_search (:24-151) restated as a table of nodes (no object graph, no
properties): per node three float32[1584] vectors N / W / P, a children dict, the env copy.  Pinned to the true
reference's outputs by tests/test_host_cpu.py (tests/golden/uct.json), then used by the GPU tests as the oracle
for the HIVE_SEARCH_UCT kernels on positions the golden file does not hold.  Nothing in the product imports it.
"""
import copy
import math

import numpy as np

ACTIONS = 1584


def uct_reads(game, reads, predict):
    """-> (N float32[1584], W float32[1584], P float32[1584], best) of the root after `reads` reads.
    predict(planes float[12,12,56]) -> (p float32[1584], v float)."""
    nodes = [{"env": game, "kids": {}, "legal": None, "N": np.zeros(ACTIONS, np.float32), "W": np.zeros(ACTIONS, np.float32),
              "P": np.zeros(ACTIONS, np.float32)}]
    root_visits = np.float32(0)          # the DummyNode's counter (:122-126): visits of the root itself
    for _ in range(reads):
        # ---- select (:58-73): walk while the node has been expanded with at least one legal move
        path = []                        # (node index, action) pairs leading to the leaf
        at = 0
        visits = root_visits
        while nodes[at]["legal"]:
            nd = nodes[at]
            score = nd["W"] / (1 + nd["N"]) + math.sqrt(visits) * (np.abs(nd["P"]) / (1 + nd["N"]))      # :52-57
            legal = nd["legal"]
            a = legal[int(np.argmax(score[legal]))]
            if a not in nd["kids"]:      # :102-109
                env = copy.deepcopy(nd["env"])
                env.move(a)
                nodes.append({"env": env, "kids": {}, "legal": None, "N": np.zeros(ACTIONS, np.float32),
                              "W": np.zeros(ACTIONS, np.float32), "P": np.zeros(ACTIONS, np.float32)})
                nd["kids"][a] = len(nodes) - 1
            path.append((at, a))
            visits = nd["N"][a]
            at = nd["kids"][a]
        leaf = nodes[at]
        p, v = predict(leaf["env"].encode_board())
        if not leaf["env"].game_is_over():       # :144-147
            legal = list(leaf["env"].actions())  # :81-95: priors of illegal actions zeroed, nothing renormalised
            keep = np.zeros(ACTIONS, bool)
            keep[legal] = True
            leaf["P"] = np.where(keep, np.asarray(p, np.float32), np.float32(0))
            leaf["legal"] = legal
        # ---- backup (:111-119): +v where black is to move at the child, -v where white is
        for parent, a in reversed(path):
            nd = nodes[parent]
            nd["N"][a] += 1
            side = nodes[nd["kids"][a]]["env"].player()
            nd["W"][a] += v if side == 1 else -v
        root_visits += 1
    r = nodes[0]
    return r["N"], r["W"], r["P"], int(np.argmax(r["N"]))
