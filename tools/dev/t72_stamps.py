"""DEBUG build 5 of gen_tower_asm.py: in-kernel timeline of the 72-tile tower (s_memtime stamps of wave 0 of every workgroup):
per block   0 block start | 1 conv1 loop done | 2 after the barrier + bias | 3 epilogue 1 written | 4 barrier passed |
            5 conv2 loop done | 6 barrier + bias | 7 epilogue 2 stored."""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hive_alphazero_amd import _lib
L = _lib.load()
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
B, NBLK = 1024, 19
torch.manual_seed(0)
x = torch.relu(torch.randn((B, 144, 256), device="cuda")).to(torch.bfloat16)
w = (torch.randn((2 * NBLK, 9 * 8 * 16 * 64 * 8), device="cuda") * 0.015).to(torch.bfloat16)
bias = torch.randn((2 * NBLK, 256), device="cuda") * 0.1
y = torch.zeros_like(x)
stamps = torch.zeros((B // 2, NBLK, 16), dtype=torch.int64, device="cuda")
n = torch.tensor([B], dtype=torch.int32, device="cuda")
for _ in range(3):
    _lib.check(L.hive_nn_tower72(P(x), P(w), P(bias), P(y), B, NBLK, _lib.BF16, P(stamps), P(n), None))
torch.cuda.synchronize()
s = stamps.cpu().numpy().astype("int64")
t0 = s[:, 0, 0].min()
import numpy as np
# stamp order in time: 0 1 2 3 4 5 6 8 9 7 (8 = x staged into the image, 9 = tiles written back, 7 = output stored)
order = [0, 1, 2, 3, 4, 5, 6, 8, 9, 7]
seg = ["conv1 loop", "barrier+bias", "epilogue 1", "barrier", "conv2 loop", "barrier+bias", "epi 2: x -> image", "epi 2: tiles",
       "epi 2: image -> Y", "-> next block"]
d = np.zeros((B // 2, NBLK, 10))
for k in range(9):
    d[:, :, k] = s[:, :, order[k + 1]] - s[:, :, order[k]]
d[:, :-1, 9] = s[:, 1:, 0] - s[:, :-1, 7]
start = s[:, 0, 0] - t0
first = start < np.median(start)            # workgroups of the first round
print("workgroups starting in the first round:", int(first.sum()), " start spread (cycles): first round", int(start[first].max()),
      " second round", int(start[~first].min()), "..", int(start[~first].max()))
print("total cycles per workgroup (block 0 start -> last stamp): median", int(np.median(s[:, -1, 7] - s[:, 0, 0])),
      " kernel span", int(s[:, -1, 7].max() - t0))
tot = d[:, 1:-1].sum(axis=2).mean()
for k, name in enumerate(seg):
    v = d[:, 1:-1, k]
    print(f"{name:24s} median {np.median(v):9.0f}  mean {v.mean():9.0f}  p95 {np.percentile(v, 95):9.0f}   {100 * v.mean() / tot:5.1f} % of a block")
print(f"block total (mean) {tot:.0f} cycles; MFMA issue floor 2 x 72 x 72 x 16 = {2 * 72 * 72 * 16}")
