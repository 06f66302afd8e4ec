"""How many of a ply's evaluated leaf rows were already evaluated (same board record + history = same planes) during the
PREVIOUS ply's searches, or at any earlier time?  (The reference resets its tree on every move, solo_play.py:103-112, so the
subtree under the move just played is evaluated again.)  1024 lock-step games from the opening, 50 simulations."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hive_alphazero_amd import mcts
from hive_alphazero_amd.alpha_net import ChessNet, InferenceNet

torch.manual_seed(0)
inf = InferenceNet(ChessNet().cuda().eval(), dtype=torch.bfloat16, tune_gemms=False)
G = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
PLIES = int(sys.argv[2]) if len(sys.argv) > 2 else 12
gen = torch.Generator(device="cuda").manual_seed(1)
r1 = torch.randint(-2 ** 62, 2 ** 62, (12 * 12 * 56,), device="cuda", generator=gen, dtype=torch.int64)


class Probe:
    accepts_need = True
    dtype = torch.bfloat16

    def __init__(self):
        self.keys = []

    def __call__(self, planes, need=None):
        key = (planes.reshape(planes.shape[0], -1).view(torch.int16).to(torch.int64) * r1).sum(1)
        self.keys.append(key[need.bool()] if need is not None else key)
        return inf(planes, need=need)


probe = Probe()
sp = mcts.SelfPlay(G, 50, probe, seed=1234, keep_records=False, game_ids=range(G), search_options={"share_equal_leaves": False})
prev = torch.zeros(0, dtype=torch.int64, device="cuda")
ever = torch.zeros(0, dtype=torch.int64, device="cuda")
for ply in range(PLIES):
    probe.keys = []
    sp.play_ply()
    k = torch.cat(probe.keys)
    uniq = torch.unique(k)
    hit_prev = int(torch.isin(uniq, prev).sum())
    hit_ever = int(torch.isin(uniq, ever).sum())
    print(f"ply {ply:2d}: rows {k.numel():6d}  distinct {uniq.numel():6d}  of those seen in the previous ply {hit_prev:6d} "
          f"({hit_prev / uniq.numel():.3f})  seen ever {hit_ever:6d} ({hit_ever / uniq.numel():.3f})", flush=True)
    prev = uniq
    ever = torch.unique(torch.cat([ever, uniq]))
