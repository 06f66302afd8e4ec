"""Forward latency of InferenceNet by leaf-batch size, hand-written HIP tower vs the torch/MIOpen backend, plus the
bf16 GEMM rate hipBLASLt sustains on this card (the practical MFMA ceiling the tower's TFLOP/s should be read against)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hive_alphazero_amd.alpha_net import ChessNet, InferenceNet

def timeit(fn, n):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

torch.manual_seed(0)
net = ChessNet().cuda().eval()
sizes = [int(a) for a in sys.argv[1:]] or [1, 8, 32, 128, 256, 512, 1024]
backs = {"hip": InferenceNet(net, conv="hip"), "torch": InferenceNet(net, conv="torch")}
print("| leaf batch | " + " | ".join(f"{k} ms" for k in backs) + " | hip TFLOP/s |")
print("|---|" + "---|" * (len(backs) + 1))
for B in sizes:
    x = (torch.rand((B, 12, 12, 56), device="cuda") < 0.1).to(torch.bfloat16)
    ms = {k: timeit(lambda: inf(x), 20 if B >= 128 else 50) for k, inf in backs.items()}
    print(f"| {B} | " + " | ".join(f"{ms[k]:.3f}" for k in backs) + f" | {B * 6.560114816 / ms['hip']:.0f} |")
a = torch.randn((8192, 8192), device="cuda", dtype=torch.bfloat16)
b = torch.randn((8192, 8192), device="cuda", dtype=torch.bfloat16)
ms = timeit(lambda: a @ b, 20)
print(f"\nhipBLASLt bf16 GEMM 8192^3: {ms:.3f} ms = {2 * 8192**3 / ms / 1e9:.0f} TFLOP/s (nominal dense peak 2500)")
