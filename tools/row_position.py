"""Is InferenceNet's output for a board independent of the ROW it sits in (tuned hipBLASLt head GEMMs included)?
A 1024-row batch made of copies of a few distinct boards at scattered rows: all copies must come back bit-identical."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hive_alphazero_amd.alpha_net import ChessNet, InferenceNet
torch.manual_seed(0)
net = ChessNet().cuda().eval()
for dtype in (torch.bfloat16, torch.float16):
    for B in (1024, 4096, 640):
        inf = InferenceNet(net, dtype=dtype, tune_gemms=(len(sys.argv) < 2))          # GEMM tuning on, as self-play runs it (any argument: off)
        g = torch.Generator(device="cuda").manual_seed(B)
        base = (torch.rand((13, 12, 12, 56), device="cuda", generator=g) < 0.08).to(dtype)
        which = torch.randint(0, 13, (B,), device="cuda", generator=g)
        x = base[which]
        for rep in range(2):
            p, v = inf(x)
        ok = True
        for k in range(13):
            rows = torch.nonzero(which == k).view(-1)
            ok &= bool((p[rows] == p[rows[0]]).all()) and bool((v[rows] == v[rows[0]]).all())
        print(dtype, B, "copies identical wherever they sit:", ok, flush=True)
