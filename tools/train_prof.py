"""A few bf16 training steps for a kernel-trace profile (rocprofv3 --kernel-trace -- python tools/train_prof.py)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hive_alphazero_amd.alpha_net import ChessNet, Trainer
B = 512
g = torch.Generator(device="cuda").manual_seed(0)
x = (torch.rand((B, 56, 12, 12), device="cuda", generator=g) < 0.1).float()
pi = torch.softmax(torch.randn((B, 1584), device="cuda", generator=g), 1)
z = torch.sign(torch.randn((B,), device="cuda", generator=g))
tr = Trainer(ChessNet().cuda())
for _ in range(6):
    tr.step(x, pi, z)
torch.cuda.synchronize()
