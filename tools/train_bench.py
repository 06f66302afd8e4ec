"""Training-step throughput of hive_alphazero_amd.alpha_net.Trainer on one MI355X: batch 512 like the
reference (alpha_net.py:117-162).  Variants in ONE process on one device (devices differ by several per cent): the HIP
step, the same with MIOpen's weight gradient (what round 1 shipped), and plain fp32.  (Capturing the whole step in a HIP
graph was tried in round 2: 16.69 ms against 16.09 ms eager on the same device -- the step is GPU-bound, not launch-bound.)"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hive_alphazero_amd import alpha_net
from hive_alphazero_amd.alpha_net import ChessNet, Trainer

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
g = torch.Generator(device="cuda").manual_seed(0)
x = (torch.rand((B, 56, 12, 12), device="cuda", generator=g) < 0.1).float()
pi = torch.softmax(torch.randn((B, 1584), device="cuda", generator=g), 1)
z = torch.sign(torch.randn((B,), device="cuda", generator=g))
GFLOP_FWD = 6.560114816
VARIANTS = (("bf16 HIP kernels", torch.bfloat16, False, True),
            ("bf16 HIP kernels, MIOpen wgrad", torch.bfloat16, False, False), ("fp32 (libraries)", None, False, True))
for rnd in range(2):                       # two rounds: the second one is read (clocks settled, libraries tuned)
    for name, dt, graph, hip_wgrad in VARIANTS:
        if dt is None and rnd == 0:
            continue
        alpha_net._Conv3x3.hip_wgrad = hip_wgrad
        torch.manual_seed(0)
        tr = Trainer(ChessNet().cuda(), autocast_dtype=dt)
        for _ in range(5):
            loss = tr.step(x, pi, z)
        torch.cuda.synchronize()
        n = 10
        t0 = time.perf_counter()
        for _ in range(n):
            loss = tr.step(x, pi, z)
        torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / n
        if rnd == 1:
            print(f"{name:32s} batch {B}: {el * 1e3:7.2f} ms/step = {B / el:8.0f} positions/s = "
                  f"{3 * GFLOP_FWD * B / el / 1e3:6.0f} TFLOP/s (fwd+bwd ~ 3x forward FLOPs), loss {loss:.4f}", flush=True)
        del tr
alpha_net._Conv3x3.hip_wgrad = True
