"""How many rows of a self-play leaf batch are duplicates of another row (same planes = same position, history and turn)?
1024 lock-step games from the opening, 50 simulations per move: per ply, evaluated rows and distinct rows among them."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hive_alphazero_amd import mcts
from hive_alphazero_amd.alpha_net import ChessNet, InferenceNet

torch.manual_seed(0)
inf = InferenceNet(ChessNet().cuda().eval(), dtype=torch.bfloat16, tune_gemms=False)
G = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
PLIES = int(sys.argv[2]) if len(sys.argv) > 2 else 14
gen = torch.Generator(device="cuda").manual_seed(1)
r1 = torch.randint(-2 ** 62, 2 ** 62, (12 * 12 * 56,), device="cuda", generator=gen, dtype=torch.int64)


class Probe:
    accepts_need = True
    dtype = torch.bfloat16

    def __init__(self):
        self.rows = self.distinct = 0

    def __call__(self, planes, need=None):
        key = (planes.reshape(planes.shape[0], -1).view(torch.int16).to(torch.int64) * r1).sum(1)
        sel = need.bool() if need is not None else torch.ones(planes.shape[0], dtype=torch.bool, device=planes.device)
        self.rows += int(sel.sum())
        self.distinct += int(torch.unique(key[sel]).numel())
        return inf(planes, need=need)


probe = Probe()
sp = mcts.SelfPlay(G, 50, probe, seed=1234, keep_records=False, game_ids=range(G))
for ply in range(PLIES):
    probe.rows = probe.distinct = 0
    sp.play_ply()
    print(f"ply {ply:2d}: evaluated rows {probe.rows:6d}, distinct {probe.distinct:6d} ({probe.distinct / max(probe.rows, 1):.3f})", flush=True)
