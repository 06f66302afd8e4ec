"""Which operators of a training step issue device-to-device copies / fills (torch.profiler, one step)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hive_alphazero_amd.alpha_net import ChessNet, Trainer
from torch.profiler import profile, ProfilerActivity
B = 512
g = torch.Generator(device="cuda").manual_seed(0)
x = (torch.rand((B, 56, 12, 12), device="cuda", generator=g) < 0.1).float()
pi = torch.softmax(torch.randn((B, 1584), device="cuda", generator=g), 1)
z = torch.sign(torch.randn((B,), device="cuda", generator=g))
tr = Trainer(ChessNet().cuda())
for _ in range(4):
    tr.step(x, pi, z)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    tr.step(x, pi, z)
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    if e.count >= 10:
        rows.append((e.count, e.key, str(e.input_shapes)[:100], round(e.device_time_total / 1e3, 3), round(e.self_device_time_total / 1e3, 3)))
rows.sort(key=lambda r: -r[4])
for r in rows[:45]:
    print(r)
