"""Does splitting the 1024-board residual chain into two independent 512-board chains on two streams (boards are independent,
only blocks of ONE board are ordered) beat the single chain?  The launch boundary of a single stream is a chip-wide barrier
per block; two streams let one half's tail overlap the other half's work."""
import ctypes, os, statistics, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hive_alphazero_amd import _lib
L = _lib.load()
B, NBLK = (int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[1] in ("parts", "seq") else 1024), 19
P = lambda t: ctypes.c_void_p(t.data_ptr())
torch.manual_seed(0)
x = torch.randn((B, 144, 256), device="cuda").to(torch.bfloat16)
w = (torch.randn((2 * NBLK, 9 * 8 * 16 * 64 * 8), device="cuda") * 0.015).to(torch.bfloat16)
bias = torch.randn((2 * NBLK, 256), device="cuda") * 0.1
y1, y2 = torch.empty_like(x), torch.empty_like(x)
def chain(lo, n, stream):
    bufs = [x[lo:lo + n], y1[lo:lo + n], y2[lo:lo + n]]
    cur = 0
    for i in range(NBLK):
        nxt = 1 if cur != 1 else 2
        assert L.hive_nn_resblock_dt(P(bufs[cur]), P(w[2 * i]), P(bias[2 * i]), P(w[2 * i + 1]), P(bias[2 * i + 1]), P(bufs[nxt]), n, _lib.BF16,
                                     ctypes.c_void_p(stream.cuda_stream)) == 0
        cur = nxt
    return bufs[cur]
main = torch.cuda.current_stream()
pool = [torch.cuda.Stream() for _ in range(8)]
def one():
    chain(0, B, main)
def two(parts=2):
    ev = torch.cuda.Event(); ev.record(main)
    streams = pool[:parts]
    for k, s in enumerate(streams):
        s.wait_event(ev)
        chain(k * (B // parts), B // parts, s)
        e = torch.cuda.Event(); e.record(s); main.wait_event(e)
def seq(chunk, nstreams):
    """chains of `chunk` boards, dealt round-robin onto `nstreams` streams (chains of one stream run one after another)"""
    ev = torch.cuda.Event(); ev.record(main)
    streams = pool[:nstreams]
    for s_ in streams:
        s_.wait_event(ev)
    for k, lo in enumerate(range(0, B, chunk)):          # (the last chain takes the remainder)
        chain(lo, min(chunk, B - lo), streams[k % nstreams])
    for s_ in streams:
        e = torch.cuda.Event(); e.record(s_); main.wait_event(e)
def two_upto(count):
    """what a compacted leaf batch would run: boards [0, 512) and [512, count) -- the second chain is shorter"""
    ev = torch.cuda.Event(); ev.record(main)
    for k, s in enumerate(pool[:2]):
        lo = k * (B // 2)
        n = min(B // 2, count - lo)
        s.wait_event(ev)
        if n > 0:
            chain(lo, n, s)
        e = torch.cuda.Event(); e.record(s); main.wait_event(e)
def two_even(count):
    """the same boards split evenly"""
    ev = torch.cuda.Event(); ev.record(main)
    h = (count + 1) // 2
    for k, s in enumerate(pool[:2]):
        s.wait_event(ev)
        chain(k * h, min(h, count - k * h), s)
        e = torch.cuda.Event(); e.record(s); main.wait_event(e)
ref = None
forms = (("one stream x 1024", one), ("two streams x 512", two), ("four streams x 256", lambda: two(4)), ("eight streams x 128", lambda: two(8)))
if len(sys.argv) > 1 and sys.argv[1] == "seq":            # python tools/two_stream_chain.py seq 4096 512:2 1024:2 512:4
    forms = (("one chain", one),) + tuple((f"chains of {c} on {k} streams", (lambda c=c, k=k: seq(c, k)))
                                          for c, k in (map(int, a.split(":")) for a in sys.argv[3:]))
elif len(sys.argv) > 1 and sys.argv[1] == "parts":          # python tools/two_stream_chain.py parts 4096 1 2 4 8
    forms = tuple((f"{k} chain(s) x {B // k}", (one if k == 1 else (lambda k=k: two(k)))) for k in map(int, sys.argv[3:]))
elif len(sys.argv) > 1 and sys.argv[1] == "counts":
    forms = (("two streams x 512", two),) + tuple((f"512 + {c - 512}", (lambda c=c: two_upto(c))) for c in (992, 960, 928, 896, 768)) \
        + tuple((f"2 x {c // 2}", (lambda c=c: two_even(c))) for c in (992, 960, 896, 768))
times = {k: [] for k, _ in forms}
for name, fn in forms:
    fn(); torch.cuda.synchronize()
    out = (y1 if NBLK % 2 else y2).clone()
    ref = out if ref is None else ref
    if "+" not in name and (" x " in name or "chain" in name) and not name.startswith("2 x"):
        print(name, "same bits as the single chain:", bool(torch.equal(out, ref)))
for _ in range(8):
    for name, fn in forms:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(main)
        for _ in range(3): fn()
        e1.record(main); torch.cuda.synchronize()
        times[name].append(e0.elapsed_time(e1) / 3)
for k, v in times.items():
    print(f"{k:22s} median {statistics.median(v):7.3f} ms  ({statistics.median(v) / NBLK * 1e3:6.1f} us per block)  min {min(v):7.3f} ms")
