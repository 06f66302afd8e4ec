"""Debug builds of gen_tower_asm.py against references: mode 1 identity, 2 epilogue 2 only (y = relu(b2 + x)), 3 conv1 + epilogue 1."""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hive_alphazero_amd import _lib
L = _lib.load()
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
mode = int(sys.argv[1])
torch.manual_seed(0)
for B in (2, 5):
    x = torch.relu(torch.randn((B, 144, 256), device="cuda")).to(torch.bfloat16)
    w = (torch.randn((2, 9 * 8 * 16 * 64 * 8), device="cuda") * 0.015).to(torch.bfloat16)
    bias = torch.randn((2, 256), device="cuda") * 0.1
    y = torch.full_like(x, 7.0)
    _lib.check(L.hive_nn_tower72(P(x), P(w), P(bias), P(y), B, 1, _lib.BF16, None, None, st()))
    torch.cuda.synchronize()
    if mode == 1:
        want = x
    elif mode == 2:
        want = torch.relu(bias[1].view(1, 1, 256) + x.float()).to(torch.bfloat16)
    else:
        want = torch.zeros_like(x)
        _lib.check(L.hive_nn_conv3x3_dt(P(x), 256, P(w[0]), P(bias[0]), None, P(want), B, 1, _lib.BF16, st()))
        torch.cuda.synchronize()
    d = (y != want)
    print(f"mode {mode} B={B}: identical {not bool(d.any())}; differing {int(d.sum())} of {d.numel()}; still 7.0: {int((y == 7.0).sum())}; "
          f"non-finite {int((~torch.isfinite(y.float())).sum())}")
    if d.any():
        b = int(torch.nonzero(d.any(2).any(1)).flatten()[0])
        px = torch.nonzero(d[b].any(1)).flatten().tolist()
        ch = torch.nonzero(d[b].any(0)).flatten().tolist()
        print(f"   board {b}: pixels {px[:20]} ({len(px)}), channels {ch[:20]} ({len(ch)})")
        print("   got ", y[b, px[0], ch[:8]].float().tolist(), "\n   want", want[b, px[0], ch[:8]].float().tolist())
    if mode == 2 and B == 2:
        wr = (y != 7.0)
        for b in range(B):
            px = torch.nonzero(wr[b].any(1)).flatten().tolist()
            ch = torch.nonzero(wr[b].any(0)).flatten().tolist()
            print(f"   board {b}: written pixels {px} ; channels {ch}")
            if px:
                ok = (y[b][wr[b]] == want[b][wr[b]])
                print(f"   of the written entries {int(ok.sum())} of {int(wr[b].sum())} are right")
    if mode == 3 and B == 2:
        for b in range(B):
            good_ch = torch.nonzero(~d[b].any(0)).flatten().tolist()
            good_px = torch.nonzero(~d[b].any(1)).flatten().tolist()
            print(f"   board {b}: channels right on every pixel: {good_ch}\n   pixels right on every channel: {good_px}")
            # is the rest equal to the input (never overwritten)?
            stale = (y[b] == x[b]) & d[b]
            print(f"   wrong entries equal to the staged input: {int(stale.sum())} of {int(d[b].sum())}")
