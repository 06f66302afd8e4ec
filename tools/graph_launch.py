"""Does a HIP graph shorten the per-launch time of back-to-back 4096-board movegen launches?  K launches captured in one
graph (each depends on the previous: same stream) against K plain launches on the stream; interleaved rounds."""
import ctypes, os, statistics, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hive_alphazero_amd as h
from hive_alphazero_amd import playout
from hive_alphazero_amd._lib import HIVE_MASK_WORDS
L = h.load()
n, K = 4096, 200
boards = playout.random_positions(n, seed=1000)
mask = torch.empty((n, HIVE_MASK_WORDS), dtype=torch.int32, device="cuda")
count = torch.empty((n,), dtype=torch.int32, device="cuda")
bp, mp, cp = (ctypes.c_void_p(t.data_ptr()) for t in (boards, mask, count))
side = torch.cuda.Stream()
def launches(stream):
    sp = ctypes.c_void_p(stream.cuda_stream)
    for _ in range(K):
        assert L.hive_movegen_launch(bp, n, mp, cp, None, sp) == 0
with torch.cuda.stream(side):
    launches(side)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=side):
    launches(side)
torch.cuda.synchronize()
def timed(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(side):
        e0.record(side); fn(); e1.record(side)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / K * 1e3
res = {"stream": [], "graph": []}
for _ in range(10):
    res["stream"].append(timed(lambda: launches(side)))
    res["graph"].append(timed(lambda: g.replay()))
for k, v in res.items():
    print(f"{k:7s} median {statistics.median(v):6.2f} us per launch   min {min(v):6.2f}")
