import sys, torch
sys.path.insert(0, '/root/repo')
from hive_alphazero_amd import mcts
from hive_alphazero_amd.alpha_net import ChessNet, InferenceNet
torch.manual_seed(0)
net = InferenceNet(ChessNet().cuda().eval(), dtype=torch.bfloat16)
sp = mcts.SelfPlay(1024, 50, net, seed=1, keep_records=False)
sp.stagger(seed=3)
fr = []
for _ in range(4):
    sp.play_ply()
    pol = sp.search.policy
    live = sp.search.sum_n > 0
    fr.append(float(pol[live].max(1).values.mean().item()))
print("mean over games of the largest root visit fraction per ply:", fr)
