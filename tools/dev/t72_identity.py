"""DEBUG build 1 of gen_tower_asm.py (staging only): the kernel copies its LDS image back out, so y must equal x."""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hive_alphazero_amd import _lib
L = _lib.load()
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
for B in (2, 5, 64):
    x = torch.randn((B, 144, 256), device="cuda").to(torch.bfloat16)
    w = torch.zeros((2, 9 * 8 * 16 * 64 * 8), device="cuda", dtype=torch.bfloat16)
    bias = torch.zeros((2, 256), device="cuda")
    y = torch.full_like(x, 7.0)
    _lib.check(L.hive_nn_tower72(P(x), P(w), P(bias), P(y), B, 1, _lib.BF16, None, None, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    print("identity B=%d:" % B, torch.equal(x, y), int((x != y).sum()))
