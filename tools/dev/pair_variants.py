"""Times build/variants/env_*.so (hive_env.hip alone, built with -D switches) on the movegen launch: pair layout at the given
sizes, the quad layout beside it.  usage: python tools/dev/pair_variants.py [boards ...]"""
import ctypes
import glob
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import hive_alphazero_amd as h  # noqa: E402
from hive_alphazero_amd import playout  # noqa: E402
from hive_alphazero_amd.batch import HIVE_MASK_WORDS  # noqa: E402


def main():
    all_quad = "--all-quad" in sys.argv[1:]             # time the quad layout of every variant, not only of "base"
    sizes = [int(a) for a in sys.argv[1:] if a != "--all-quad"] or [65536, 1 << 20]
    h.load()
    base = playout.random_positions(4096, seed=1000)
    st = torch.cuda.current_stream()
    sp = ctypes.c_void_p(st.cuda_stream)
    libs = {}
    for path in sorted(glob.glob(os.path.join(ROOT, "build", "variants", "env_*.so"))):
        L = ctypes.CDLL(path)
        L.hive_movegen_launch.argtypes = [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 4
        L.hive_movegen_pair_threshold.argtypes = [ctypes.c_int]
        libs[os.path.basename(path)[4:-3]] = L
    for nb in sizes:
        big = base.repeat((nb + 4095) // 4096, 1)[:nb].contiguous()
        m = torch.empty((nb, HIVE_MASK_WORDS), dtype=torch.int32, device="cuda")
        c = torch.empty((nb,), dtype=torch.int32, device="cuda")
        ref = None
        reps = max(10, min(400, (1 << 22) // nb))
        times = {}
        for rnd in range(9):
            for name, L in libs.items():
                for mode, thr in (("pair", 1), ("quad", 1 << 30)):
                    if mode == "quad" and name != "base" and not all_quad:
                        continue
                    L.hive_movegen_pair_threshold(thr)
                    a = (ctypes.c_void_p(big.data_ptr()), nb, ctypes.c_void_p(m.data_ptr()), ctypes.c_void_p(c.data_ptr()), None, sp)
                    for _ in range(3):
                        assert L.hive_movegen_launch(*a) == 0
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(st)
                    for _ in range(reps):
                        L.hive_movegen_launch(*a)
                    e1.record(st)
                    torch.cuda.synchronize()
                    times.setdefault((name, mode), []).append(e0.elapsed_time(e1) / reps * 1e3)
                    if ref is None:
                        ref = (m.clone(), c.clone())
                    else:
                        assert torch.equal(m, ref[0]) and torch.equal(c, ref[1]), (name, mode)
        for (name, mode), ts in times.items():
            t = sorted(ts)[len(ts) // 2]
            print(f"{nb:8d} boards  {name:10s} {mode}: {t:9.2f} us  {nb / t:7.1f} Mboards/s", flush=True)


if __name__ == "__main__":
    main()
