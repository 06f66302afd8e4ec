"""Runs the 1024-leaf forward of InferenceNet (default engine) 30 times: the target of `rocprofv3 --kernel-trace --stats`
for the per-kernel split of a forward (stem, tower, heads).  argv[1] = batch (1024), argv[2] = bf16 | fp16 | auto."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hive_alphazero_amd.alpha_net import ChessNet, InferenceNet
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dt = {"bf16": torch.bfloat16, "fp16": torch.float16, "auto": None}[sys.argv[2] if len(sys.argv) > 2 else "auto"]
torch.manual_seed(0)
inf = InferenceNet(ChessNet().cuda().eval(), dtype=dt)
x = (torch.rand((B, 12, 12, 56), device="cuda") < 0.1).to(inf.dtype)
for _ in range(3):
    inf(x)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(30):
    inf(x)
e1.record(); torch.cuda.synchronize()
print(f"{B}-leaf forward, {inf.dtype}: {e0.elapsed_time(e1) / 30:.3f} ms")
