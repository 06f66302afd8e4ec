"""Training-step throughput of hive_alphazero_amd.alpha_net.Trainer on one MI355X: batch 512 like the
reference (alpha_net.py:117-162), bf16 autocast + channels-last vs plain fp32."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hive_alphazero_amd.alpha_net import ChessNet, Trainer

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
g = torch.Generator(device="cuda").manual_seed(0)
x = (torch.rand((B, 56, 12, 12), device="cuda", generator=g) < 0.1).float()
pi = torch.softmax(torch.randn((B, 1584), device="cuda", generator=g), 1)
z = torch.sign(torch.randn((B,), device="cuda", generator=g))
GFLOP_FWD = 6.560114816
for name, dt in (("bf16 autocast, channels-last", torch.bfloat16), ("fp32", None)):
    torch.manual_seed(0)
    tr = Trainer(ChessNet().cuda(), autocast_dtype=dt)
    for _ in range(3):
        loss = tr.step(x, pi, z)
    torch.cuda.synchronize()
    n = 10
    t0 = time.perf_counter()
    for _ in range(n):
        loss = tr.step(x, pi, z)
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / n
    print(f"{name:30s} batch {B}: {el * 1e3:7.1f} ms/step = {B / el:8.0f} positions/s = "
          f"{3 * GFLOP_FWD * B / el / 1e3:6.0f} TFLOP/s (fwd+bwd ~ 3x forward FLOPs), loss {loss:.4f}")
