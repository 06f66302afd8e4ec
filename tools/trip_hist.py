"""Loop-trip statistics of the piece kernel (debug build with -DHIVE_DBG_ITERS): per piece type, the mean and maximum
number of fused flood-loop trips a wave (16 boards) takes, and the histogram -- the launch time of a small batch is set
by the slowest wave, not by the mean."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hive_alphazero_amd import playout
L = ctypes.CDLL(sys.argv[1])
L.hive_movegen_launch.argtypes = [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 4
n = 4096
boards = playout.random_positions(n, seed=1000)
mask = torch.empty((n, 66), dtype=torch.int32, device="cuda")
cnt = torch.empty((n,), dtype=torch.int32, device="cuda")
L.hive_movegen_launch(boards.data_ptr(), n, mask.data_ptr(), cnt.data_ptr(), None, None)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 224)()
assert L.hive_debug_iters(buf) == 0
v = np.array(buf[:], dtype=np.int64)
names = ["queen", "beetle", "spider", "grasshopper", "ant", "pin waves"]
for t, nm in enumerate(names):
    hist = v[32 + 32 * t: 64 + 32 * t]
    print(f"{nm:12s} waves {v[8 + t]:5d}  mean trips {v[t] / max(v[8 + t], 1):5.2f}  max {v[16 + t]:3d}  hist {list(hist[:int(v[16 + t]) + 1])}")

def stamps():
    st = (ctypes.c_ulonglong * 88)()
    assert L.hive_debug_stamps(st) == 0
    return np.array(st[:], dtype=np.float64).reshape(11, 8)
for _ in range(20):
    L.hive_movegen_launch(boards.data_ptr(), n, mask.data_ptr(), cnt.data_ptr(), None, None)
torch.cuda.synchronize()
s0 = stamps()
for _ in range(10):
    L.hive_movegen_launch(boards.data_ptr(), n, mask.data_ptr(), cnt.data_ptr(), None, None)
torch.cuda.synchronize()
s = (stamps() - s0) / 160.0      # 16 sampled workgroups x 10 warm launches
print("shader cycles since kernel start (mean over sampled workgroups; build with -DHIVE_DBG_STAMPS_ONLY for clean numbers):")
print("            staged   piece work done   scattered   tail written (only the last wave of a workgroup)")
for w in range(11):
    print(f"  slot {w:2d}: " + "  ".join(f"{x:8.0f}" for x in s[w, :4]) + (f"   pin phase done {s[w, 4]:8.0f}" if s[w, 4] else ""))
print(f"  mean workgroup end: {s[:, 3].sum():.0f} cycles")
# the fused id-list variant (hive_piece_kernel<false, true>): arrival at the barrier, release, end of the wave
lst = torch.empty((n, 256), dtype=torch.int16, device="cuda")
for _ in range(20):
    L.hive_movegen_launch(boards.data_ptr(), n, mask.data_ptr(), cnt.data_ptr(), lst.data_ptr(), None)
torch.cuda.synchronize()
s0 = stamps()
for _ in range(10):
    L.hive_movegen_launch(boards.data_ptr(), n, mask.data_ptr(), cnt.data_ptr(), lst.data_ptr(), None)
torch.cuda.synchronize()
s = (stamps() - s0) / 160.0
print("fused id lists:  staged   piece work done   dest + scatter done   at barrier   released   wave end")
for w in range(11):
    print(f"  slot {w:2d}: " + "  ".join(f"{x:8.0f}" for x in (s[w, 0], s[w, 1], s[w, 2], s[w, 5], s[w, 6], s[w, 7])))
