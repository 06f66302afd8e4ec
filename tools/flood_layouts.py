"""The Ant reach flood in the shipped quad layout (16 boards per wave) and in a pair layout (32 boards per wave, three words
per lane): same inputs, outputs compared word for word, then timed at the headline size and at saturation.
tools/dev/flood_layouts.hip; VERDICT round 2 item 6(b)."""
import ctypes, os, subprocess, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hive_alphazero_amd import playout
so = os.path.join(ROOT, "build", "libflood_layouts.so")
os.makedirs(os.path.dirname(so), exist_ok=True)
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(ROOT, "tools", "dev", "flood_layouts.hip")):
    subprocess.check_call(["hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function", "-shared", "-o", so,
                           os.path.join(ROOT, "tools", "dev", "flood_layouts.hip")])
L = ctypes.CDLL(so)
L.flood_launch.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 2 + [ctypes.c_int] + [ctypes.c_void_p] * 3
boards = playout.random_positions(4096, seed=1000).cpu().numpy()
pos = boards[:, :22]
rng = np.random.default_rng(0)
occ = np.zeros((4096, 6), dtype=np.uint32)
start = np.zeros(4096, dtype=np.uint8)
for b in range(4096):
    cells = sorted(set(int(c) for c in pos[b] if c < 144))
    for c in cells:
        r, col = divmod(c, 12)
        occ[b, r >> 1] |= np.uint32(1 << (((r & 1) << 4) | col))
    start[b] = rng.choice(cells) if cells else 0
P = lambda t: ctypes.c_void_p(t.data_ptr())
res = {}
for n in (4096, 65536, 1 << 20):
    reps = n // 4096
    o = torch.from_numpy(np.tile(occ, (reps, 1))).cuda()
    s = torch.from_numpy(np.tile(start, reps)).cuda()
    outs, line = [], []
    for layout, name in ((0, "quad"), (1, "pair")):
        out = torch.zeros((n, 6), dtype=torch.int32, device="cuda")
        trips = torch.zeros((1,), dtype=torch.int32, device="cuda")
        st = torch.cuda.current_stream()
        assert L.flood_launch(layout, P(o), P(s), n, P(out), P(trips), st.cuda_stream) == 0
        torch.cuda.synchronize()
        waves = (n * (4 if layout == 0 else 2) + 63) // 64
        k = 200 if n <= 4096 else (50 if n <= 65536 else 10)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(k):
            L.flood_launch(layout, P(o), P(s), n, P(out), None, st.cuda_stream)
        e1.record(st); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / k * 1e3
        outs.append(out)
        line.append(f"{name}: {us:9.2f} us = {n / us:8.1f} Mboards/s, {trips.item() / waves:5.2f} double-steps per wave")
    assert torch.equal(outs[0], outs[1]), n
    reach = int((outs[0].cpu().numpy().view(np.uint32)[:4096] != 0).any(1).sum())
    print(f"n = {n:8d}  " + "   ".join(line) + f"   (same bits; {reach} of 4096 floods non-empty)", flush=True)
