"""Correctness of hive_nn_tower72 (the 72-tile assembly tower) against the launch-per-block chain, bit for bit, in steps:
small batches, odd batches, one and several blocks, with and without a row list.  Usage: tower72_check.py [max_blocks]"""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hive_alphazero_amd import _lib

L = _lib.load()
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
DT = {torch.bfloat16: _lib.BF16, torch.float16: _lib.F16}
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
MAXB = int(sys.argv[1]) if len(sys.argv) > 1 else 19
torch.manual_seed(0)


def chain(x, w, bias, nblk, dtype):
    B = x.shape[0]
    bufs = [x, torch.zeros_like(x), torch.zeros_like(x)]
    cur = 0
    for i in range(nblk):
        nxt = 1 if cur != 1 else 2
        _lib.check(L.hive_nn_resblock_dt(P(bufs[cur]), P(w[2 * i]), P(bias[2 * i]), P(w[2 * i + 1]), P(bias[2 * i + 1]),
                                         P(bufs[nxt]), B, DT[dtype], st()))
        cur = nxt
    return bufs[cur]


ok = True
for dtype in (torch.bfloat16, torch.float16):
    for B, nblk, use_rows, balanced in ((2, 1, False, False), (5, 1, False, False), (64, 1, False, False), (7, 2, False, False),
                                        (64, 3, True, False), (300, MAXB, False, False), (1024, MAXB, True, False),
                                        (7, 2, False, True), (300, MAXB, True, True), (700, MAXB, False, True), (1024, MAXB, True, True),
                                        (1031, 3, False, True), (2500, 5, True, True)):
        nblk = min(nblk, MAXB)
        x = torch.relu(torch.randn((B, 144, 256), device="cuda")).to(dtype)
        w = (torch.randn((2 * nblk, 9 * 8 * 16 * 64 * 8), device="cuda") * 0.015).to(dtype)
        bias = torch.randn((2 * nblk, 256), device="cuda") * 0.1
        want = chain(x, w, bias, nblk, dtype)
        y = torch.full_like(x, 7.0)
        rows = nrows = None
        if use_rows:
            need = (torch.rand((B,), device="cuda") < 0.9).to(torch.int8)
            rows = torch.full((B,), -1, dtype=torch.int32, device="cuda")
            nrows = torch.zeros((1,), dtype=torch.int32, device="cuda")
            _lib.check(L.hive_nn_compact_rows(P(need), B, P(rows), P(nrows), st()))
            k = int(nrows.item())
            assert k == int(need.sum().item()) and torch.equal(rows[:k].long(), torch.nonzero(need).flatten())
        if balanced:
            ws = torch.empty((int(L.hive_nn_tower72_plan_bytes(B)),), dtype=torch.uint8, device="cuda")
            _lib.check(L.hive_nn_tower72_balanced(P(x), P(w), P(bias), P(y), B, nblk, DT[dtype], P(rows), P(nrows), P(ws), st()))
        else:
            _lib.check(L.hive_nn_tower72(P(x), P(w), P(bias), P(y), B, nblk, DT[dtype], P(rows), P(nrows), st()))
        torch.cuda.synchronize()
        if use_rows:
            sel = need.bool()
            same = torch.equal(y[sel], want[sel]) and bool((y[~sel] == 7.0).all())
        else:
            same = torch.equal(y, want)
        bad = (y != want).any(dim=2).any(dim=1) if not use_rows else ((y != want).any(dim=2).any(dim=1) & need.bool())
        md = float((y.float() - want.float()).abs().max())
        print(f"{str(dtype):16s} B={B:5d} blocks={nblk:2d} rows={use_rows!s:5s} balanced={balanced!s:5s}: {'identical' if same else 'DIFFERENT'}"
              f"  (boards differing: {int(bad.sum())}, max |d| {md:.4g}, finite {bool(torch.isfinite(y.float()).all())})", flush=True)
        if not same:
            ok = False
            b = int(torch.nonzero(bad).flatten()[0]) if bad.any() else 0
            d = (y[b] != want[b])
            px = torch.nonzero(d.any(dim=1)).flatten().tolist()
            ch = torch.nonzero(d.any(dim=0)).flatten().tolist()
            print(f"   first bad board {b}: {len(px)} pixels differ (first {px[:12]}), {len(ch)} channels differ (first {ch[:12]})")
            print("   got ", y[b, px[0], ch[:6]].float().tolist() if px else None, "\n   want", want[b, px[0], ch[:6]].float().tolist() if px else None)
            break
    if not ok:
        break
print("ALL IDENTICAL" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
