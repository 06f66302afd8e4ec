"""The planes writer (hive_expand_kernel) at 65,536 boards against what the card does on the same bytes with a plain fill and
a plain copy (torch kernels): is the writer short of the store bandwidth, or is that what 1 GB of stores costs here?
Usage: encode_bench.py [boards]   (run under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE for the traffic passes)"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hive_alphazero_amd as h
from hive_alphazero_amd import _lib as _l
_l.SO_PATH = os.environ.get('HIVE_SO', _l.SO_PATH)      # a variant build of the whole library (A/B of one kernel)
from hive_alphazero_amd import playout
from hive_alphazero_amd._lib import BF16, F32, HWC, CHW
L = h.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
boards = playout.random_positions(min(n, 4096), seed=4242).repeat(max(n // 4096, 1), 1).contiguous()
hist = torch.zeros((n, 384), dtype=torch.uint8, device="cuda")
ws = torch.empty((n * 144,), dtype=torch.int64, device="cuda")
planes = torch.empty((n, 12, 12, 56), dtype=torch.bfloat16, device="cuda")
other = torch.empty_like(planes)
P = lambda t: ctypes.c_void_p(t.data_ptr())
st = torch.cuda.current_stream()
sp = ctypes.c_void_p(st.cuda_stream)
assert L.hive_encode_launch(P(boards), P(hist), n, P(planes), BF16, HWC, P(ws), sp) == 0
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps): fn()
    e1.record(st); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
gb = planes.numel() * 2 / 1e9
for name, fn, bytes_ in (
        ("hive_expand_kernel bf16 HWC", lambda: L.hive_expand_launch(P(boards), P(hist), P(ws), n, P(planes), BF16, HWC, sp), gb),
        ("hive_expand_kernel bf16 CHW", lambda: L.hive_expand_launch(P(boards), P(hist), P(ws), n, P(planes), BF16, CHW, sp), gb),
        ("torch fill_ (same bytes)", lambda: planes.fill_(1.0), gb),
        ("torch copy_ (same bytes read + written)", lambda: other.copy_(planes), 2 * gb)):
    ms = t(fn)
    print(f"{name:42s} {ms:7.4f} ms  {bytes_ / ms:7.1f} GB/s moved ({gb:.3f} GB of planes)", flush=True)
