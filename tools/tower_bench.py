"""Interleaved timing of the residual tower's launch forms on random data (1024 boards unless argv says otherwise):
   resblock x N   N launches of hive_nn_resblock_dt (round-2 form)
   tower nb=1     hive_nn_tower, one board per workgroup (board stays in LDS across blocks)
   tower nb=2     hive_nn_tower, two boards per workgroup sharing every weight fragment
for bf16 and fp16; every form's output is compared bit for bit with the first one's.  The device's clock moves by 10-15 %
within seconds under this load, so the forms are interleaved: ROUNDS rounds, in each round every form runs REPS times;
median and minimum are reported.  Usage: tower_bench.py [batch] [nblocks] [extra .so ...] (extra libraries = -D variants
of csrc/hive_nn.hip to A/B against the shipped one)."""
import ctypes, os, statistics, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hive_alphazero_amd import _lib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
NBLK = int(sys.argv[2]) if len(sys.argv) > 2 else 19
extra = sys.argv[3:]
ROUNDS, REPS = 8, 3
L = _lib.load()
P = lambda t: ctypes.c_void_p(t.data_ptr())
DT = {torch.bfloat16: _lib.BF16, torch.float16: _lib.F16}
torch.manual_seed(0)
x32 = torch.randn((B, 144, 256), device="cuda")
w32 = torch.randn((2 * NBLK, 9 * 8 * 16 * 64 * 8), device="cuda") * 0.015        # keeps activations O(1) through the tower
bias = torch.randn((2 * NBLK, 256), device="cuda") * 0.1
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def declare(lib):
    vp, i32 = ctypes.c_void_p, ctypes.c_int
    lib.hive_nn_tower.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, vp]
    lib.hive_nn_resblock_dt.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, vp]
    return lib


libs = {"": L}
for so in extra:
    libs[os.path.basename(so).replace(".so", "")] = declare(ctypes.CDLL(so))

for dtype in (torch.bfloat16, torch.float16):
    x, w = x32.to(dtype), w32.to(dtype)
    forms = {}

    def chain(lib):
        def run(y):
            bufs = [x, y[0], y[1]]
            cur = 0
            for i in range(NBLK):
                nxt = 1 if cur != 1 else 2
                assert lib.hive_nn_resblock_dt(P(bufs[cur]), P(w[2 * i]), P(bias[2 * i]), P(w[2 * i + 1]), P(bias[2 * i + 1]),
                                               P(bufs[nxt]), B, DT[dtype], st()) == 0
                cur = nxt
            return bufs[cur]
        return run

    def tower(lib, nb):
        def run(y):
            assert lib.hive_nn_tower(P(x), P(w), P(bias), P(y[0]), B, NBLK, DT[dtype], nb, st()) == 0, L.hive_last_error()
            return y[0]
        return run

    def tower72(lib):
        def run(y):
            assert lib.hive_nn_tower72(P(x), P(w), P(bias), P(y[0]), B, NBLK, DT[dtype], None, None, st()) == 0, L.hive_last_error()
            return y[0]
        return run

    def tower72_chain(lib):
        def run(y):                                  # launch per block (the form that takes a per-block row list)
            bufs = [x, y[0], y[1]]
            cur = 0
            for i in range(NBLK):
                nxt = 1 if cur != 1 else 2
                assert lib.hive_nn_tower72(P(bufs[cur]), P(w[2 * i]), P(bias[2 * i]), P(bufs[nxt]), B, 1, DT[dtype], None, None, st()) == 0
                cur = nxt
            return bufs[cur]
        return run

    for tag, lib in libs.items():
        sfx = f" [{tag}]" if tag else ""
        forms[f"resblock x {NBLK}" + sfx] = chain(lib)
        forms["tower 1 board/wg" + sfx] = tower(lib, 1)
        forms["tower 2 boards/wg 8 waves" + sfx] = tower(lib, 2)
        forms["tower 1 board/wg 1 wave/SIMD" + sfx] = tower(lib, 3)
    def tower72_balanced(lib, frac):
        need = (torch.arange(B, device="cuda") < int(B * frac)).to(torch.int8)
        rows = torch.empty((B,), dtype=torch.int32, device="cuda")
        nrows = torch.empty((1,), dtype=torch.int32, device="cuda")
        ws = torch.empty((int(lib.hive_nn_tower72_plan_bytes(B)),), dtype=torch.uint8, device="cuda")
        assert lib.hive_nn_compact_rows(P(need), B, P(rows), P(nrows), st()) == 0
        def run(y):
            assert lib.hive_nn_tower72_balanced(P(x), P(w), P(bias), P(y[0]), B, NBLK, DT[dtype], P(rows), P(nrows), P(ws), st()) == 0
            return y[0]
        return run

    if hasattr(L, "hive_nn_tower72"):
        forms["tower72 asm (2 boards/wg, 72 tiles/wave)"] = tower72(L)
        forms["tower72 asm, launch per block"] = tower72_chain(L)
        forms["tower72 balanced, every board"] = tower72_balanced(L, 1.0)
        forms["tower72 balanced, 90 % of the boards"] = tower72_balanced(L, 0.9)
        forms["tower72 balanced, 75 % of the boards"] = tower72_balanced(L, 0.75)
        forms["tower72 balanced, 60 % of the boards"] = tower72_balanced(L, 0.6)
    outs, times = {}, {k: [] for k in forms}
    ys = {k: [torch.zeros_like(x), torch.zeros_like(x)] for k in forms}
    for k, f in forms.items():
        outs[k] = f(ys[k]).clone()
    torch.cuda.synchronize()
    for _ in range(ROUNDS):
        for k, f in forms.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(REPS):
                f(ys[k])
            e1.record()
            torch.cuda.synchronize()
            times[k].append(e0.elapsed_time(e1) / REPS)
    first = next(iter(forms))
    flop = 2.0 * B * 144 * 256 * 2304 * 2 * NBLK
    print(f"## {dtype}, {B} boards, {NBLK} blocks  (finite: {bool(torch.isfinite(outs[first].float()).all())}, "
          f"mean |y| {outs[first].float().abs().mean().item():.3f})")
    for k in forms:
        med, mn = statistics.median(times[k]), min(times[k])
        nsel = int(B * float(k.split(",")[1].split("%")[0]) / 100) if "% of the boards" in k else B
        same = "" if k == first else f"  identical to '{first}': {bool(torch.equal(outs[k][:nsel], outs[first][:nsel]))}"
        print(f"{k:34s} median {med:7.3f} ms ({flop / med / 1e9:5.0f} TFLOP/s, {med / NBLK * 1e3:6.1f} us per block)  min {mn:7.3f} ms{same}",
              flush=True)
