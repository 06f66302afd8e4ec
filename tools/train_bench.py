"""Training-step throughput of hive_alphazero_amd.alpha_net.Trainer on one MI355X: batch 512 like the
reference (alpha_net.py:117-162).  Variants in ONE process on one device (devices differ by several per cent), alternating
window by window on one Trainer: the HIP step, the same with single switches turned off, and plain fp32.  (Capturing the whole step in a HIP
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
FLAGS = {"bf16 HIP kernels": {},
         "  weights packed per convolution": {"pack_once": False},
         "  skip gradient added by autograd": {"fuse_skip_grad": False},
         "  head BatchNorms in the library": {"hip_head_bn": False},
         "  BatchNorm statistics in their own pass": {"fuse_bn_stats": False},
         "  MIOpen weight gradient": {"hip_wgrad": False}}


def set_flags(over):
    alpha_net.FusedTrainNet.pack_once = over.get("pack_once", True)
    alpha_net.FusedTrainNet.fuse_skip_grad = over.get("fuse_skip_grad", True)
    alpha_net.FusedTrainNet.hip_head_bn = over.get("hip_head_bn", True)
    alpha_net.FusedTrainNet.fuse_bn_stats = over.get("fuse_bn_stats", True)
    alpha_net._Conv3x3.hip_wgrad = over.get("hip_wgrad", True)


def window(tr, n=10):
    t0 = time.perf_counter()
    for _ in range(n):
        loss = tr.step(x, pi, z)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n, loss


# ONE trainer; the variants are class switches read at every step, so they alternate window by window on the same device,
# clocks and allocator state (separate trainers measured one after the other differed by +-0.3 ms for the same code)
torch.manual_seed(0)
tr = Trainer(ChessNet().cuda(), autocast_dtype=torch.bfloat16)
for over in FLAGS.values():
    set_flags(over)
    window(tr, 4)
res = {k: [] for k in FLAGS}
for rnd in range(11):
    for name, over in FLAGS.items():
        set_flags(over)
        window(tr, 2)
        el, loss = window(tr)
        res[name].append(el)
set_flags({})
for name, v in res.items():
    el = sorted(v)[len(v) // 2]
    print(f"{name:42s} batch {B}: {el * 1e3:7.2f} ms/step (min {min(v) * 1e3:.2f}, max {max(v) * 1e3:.2f}) = {B / el:8.0f} positions/s = "
          f"{3 * GFLOP_FWD * B / el / 1e3:6.0f} TFLOP/s (fwd+bwd ~ 3x forward FLOPs)", flush=True)
del tr
tr = Trainer(ChessNet().cuda(), autocast_dtype=None)
window(tr, 3)
el, loss = window(tr, 5)
print(f"{'fp32 (libraries)':42s} batch {B}: {el * 1e3:7.2f} ms/step = {B / el:8.0f} positions/s = "
      f"{3 * GFLOP_FWD * B / el / 1e3:6.0f} TFLOP/s", flush=True)
