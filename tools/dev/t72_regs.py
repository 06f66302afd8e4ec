"""DEBUG build 4 of gen_tower_asm.py: print the lane constants every thread computed."""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hive_alphazero_amd import _lib
L = _lib.load()
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
B = 2
x = torch.zeros((B, 144, 256), device="cuda", dtype=torch.bfloat16)
w = torch.zeros((2, 9 * 8 * 16 * 64 * 8), device="cuda", dtype=torch.bfloat16)
bias = torch.zeros((2, 256), device="cuda")
y = torch.zeros_like(x)
_lib.check(L.hive_nn_tower72(P(x), P(w), P(bias), P(y), B, 1, _lib.BF16, None, None, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
torch.cuda.synchronize()
d = y.view(torch.int32).flatten()[:256 * 16].view(256, 16).cpu()
names = ["tid", "wlane", "tab", "ldsw0", "ldsw1", "ldsw2", "goff", "biasoff", "s_wave", "s_wg", "s_n", "s_nblk", "s_r0", "s_r1", "y0lo", "y1lo"]
print(names)
for t in (0, 1, 15, 16, 17, 63, 64, 65, 128, 200, 255):
    print(t, d[t].tolist())
exp_goff = torch.tensor([(t & 15) * 512 + ((t >> 4) & 3) * 8 + (t >> 6) * 128 for t in range(256)], dtype=torch.int32)
print("goff right for", int((d[:, 6] == exp_goff).sum()), "of 256 threads; wave right for", int((d[:, 8] == torch.arange(256) // 64).sum()))
