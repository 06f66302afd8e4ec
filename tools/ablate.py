"""Timing-only ablation of the piece kernel's loops (results of the ablated builds are wrong)."""
import ctypes, glob, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hive_alphazero_amd import playout
boards = playout.random_positions(4096, seed=1000)
big = boards.repeat(256, 1).contiguous()
SIZES = [int(x) for x in os.environ.get("ABLATE_SIZES", "4096,1048576").split(",")]
for so in sorted(glob.glob(sys.argv[1] + "/abl_*.so")):
    L = ctypes.CDLL(so)
    L.hive_movegen_launch.argtypes = [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 4
    res = []
    for n in SIZES:
        b = big[:n].contiguous()
        mask = torch.empty((n, 66), dtype=torch.int32, device="cuda")
        cnt = torch.empty((n,), dtype=torch.int32, device="cuda")
        s = torch.cuda.current_stream()
        args = (b.data_ptr(), n, mask.data_ptr(), cnt.data_ptr(), None, s.cuda_stream)
        reps = 500 if n <= 8192 else (100 if n <= 65536 else 10)
        for _ in range(5): L.hive_movegen_launch(*args)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(reps): L.hive_movegen_launch(*args)
        e1.record(s); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / reps * 1e3)
    print(f"{os.path.basename(so):16s} " + "  ".join(f"n={n}: {r:.2f} us" for n, r in zip(SIZES, res)))
