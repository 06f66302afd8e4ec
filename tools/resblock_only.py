"""Runs only the residual tower of a given build of csrc/hive_nn.hip (argv[1] = .so or "-" for the shipped library; argv[2] =
0: launch-per-block chain (default), 1 / 2 / 3: hive_nn_tower's workgroup forms, 72: hive_nn_tower72, the 72-tile assembly tower) 20 times at 1024 boards: the target of
rocprofv3 --pmc passes that compare LDS strides / loop variants / launch forms."""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hive_alphazero_amd import _lib
so = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] != "-" else _lib.SO_PATH
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 0
L = ctypes.CDLL(so)
vp, i32 = ctypes.c_void_p, ctypes.c_int
L.hive_nn_resblock_dt.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, vp]
L.hive_nn_tower.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, vp]
L.hive_nn_tower72.argtypes = [vp, vp, vp, vp, i32, i32, i32, vp, vp, vp]
B, NBLK = 1024, 19
torch.manual_seed(0)
x = torch.randn((B, 144, 256), device="cuda").to(torch.bfloat16)
w = (torch.randn((2 * NBLK, 9 * 8 * 16 * 64 * 8), device="cuda") * 0.015).to(torch.bfloat16)
bias = torch.randn((2 * NBLK, 256), device="cuda") * 0.1
bufs = [x, torch.empty_like(x), torch.empty_like(x)]
P = lambda t: ctypes.c_void_p(t.data_ptr())
def chain():
    if mode == 72:
        assert L.hive_nn_tower72(P(x), P(w), P(bias), P(bufs[1]), B, NBLK, _lib.BF16, None, None, None) == 0
        return
    if mode:
        assert L.hive_nn_tower(P(x), P(w), P(bias), P(bufs[1]), B, NBLK, _lib.BF16, mode, None) == 0
        return
    cur = 0
    for i in range(NBLK):
        nxt = 1 if cur != 1 else 2
        assert L.hive_nn_resblock_dt(P(bufs[cur]), P(w[2 * i]), P(bias[2 * i]), P(w[2 * i + 1]), P(bias[2 * i + 1]), P(bufs[nxt]), B, _lib.BF16, None) == 0
        cur = nxt
for _ in range(3): chain()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): chain()
e1.record(); torch.cuda.synchronize()
print(f"{os.path.basename(so)} mode {mode}: {e0.elapsed_time(e1) / 20 / NBLK * 1e3:.1f} us per block")
