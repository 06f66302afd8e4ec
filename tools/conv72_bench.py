"""hive_nn_conv72 (one convolution on the 72-tile assembly kernel) against hive_nn_conv3x3_dt at the training step's batch
sizes; interleaved, us per launch."""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hive_alphazero_amd import _lib
L = _lib.load()
P = lambda t: ctypes.c_void_p(t.data_ptr())
torch.manual_seed(0)
for B in [int(a) for a in sys.argv[1:]] or [256, 512, 1024]:
    x = torch.randn((B, 144, 256), device="cuda").to(torch.bfloat16)
    w = (torch.randn((9 * 8 * 16 * 64 * 8,), device="cuda") * 0.015).to(torch.bfloat16)
    b = torch.zeros((256,), device="cuda")
    y = torch.empty_like(x)
    r = torch.randn((B, 144, 256), device="cuda").to(torch.bfloat16) * 0.01
    forms = {"conv3x3_kernel": lambda: L.hive_nn_conv3x3_dt(P(x), 256, P(w), P(b), None, P(y), B, 0, _lib.BF16, None),
             "conv72 (asm)": lambda: L.hive_nn_conv72(P(x), P(w), P(b), P(y), B, 0, _lib.BF16, None),
             "conv72 + residual": lambda: L.hive_nn_conv72_add(P(x), P(w), P(b), P(r), P(y), B, 0, _lib.BF16, None),
             "conv72 + residual in place": lambda: L.hive_nn_conv72_add(P(x), P(w), P(b), P(r), P(r), B, 0, _lib.BF16, None),
             "conv72, then torch add": lambda: (L.hive_nn_conv72(P(x), P(w), P(b), P(y), B, 0, _lib.BF16, None), r.add_(y))}
    res = {k: [] for k in forms}
    for k, f in forms.items():
        f()
    torch.cuda.synchronize()
    for _ in range(6):
        for k, f in forms.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                f()
            e1.record()
            torch.cuda.synchronize()
            res[k].append(e0.elapsed_time(e1) / 20 * 1e3)
    print(f"batch {B}: " + ", ".join(f"{k} {sorted(v)[len(v) // 2]:.1f} us" for k, v in res.items()))
