"""Movegen kernel, quad layout against pair layout (hive_bb.hpp), interleaved in one process.
usage: python tools/dev/pair_bench.py [--list] [boards ...]     (default 4096 16384 65536 1048576)"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import hive_alphazero_amd as h  # noqa: E402
from hive_alphazero_amd import playout  # noqa: E402
from hive_alphazero_amd.batch import HIVE_MASK_WORDS  # noqa: E402


def main():
    with_list = "--list" in sys.argv[1:]
    sizes = [int(a) for a in sys.argv[1:] if a != "--list"] or [4096, 16384, 65536, 1 << 20]
    L = h.load()
    base = playout.random_positions(4096, seed=1000)
    st = torch.cuda.current_stream()
    sp = ctypes.c_void_p(st.cuda_stream)
    for nb in sizes:
        big = base.repeat((nb + 4095) // 4096, 1)[:nb].contiguous()
        out = {}
        bufs = {}
        for name, thr in (("quad", 1 << 30), ("pair", 1)):
            bufs[name] = (torch.empty((nb, HIVE_MASK_WORDS), dtype=torch.int32, device="cuda"),
                          torch.empty((nb,), dtype=torch.int32, device="cuda"))
        lst = torch.empty((nb, 256), dtype=torch.int16, device="cuda") if with_list else None
        LST = ctypes.c_void_p(lst.data_ptr()) if with_list else None
        reps = max(10, min(200, (1 << 22) // nb))
        times = {"quad": [], "pair": []}
        for rnd in range(7):
            for name, thr in (("quad", 1 << 30), ("pair", 1)):
                L.hive_movegen_pair_threshold(thr)
                m, c = bufs[name]
                a = (ctypes.c_void_p(big.data_ptr()), nb, ctypes.c_void_p(m.data_ptr()), ctypes.c_void_p(c.data_ptr()), LST, sp)
                for _ in range(3):
                    L.hive_movegen_launch(*a)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(st)
                for _ in range(reps):
                    L.hive_movegen_launch(*a)
                e1.record(st)
                torch.cuda.synchronize()
                times[name].append(e0.elapsed_time(e1) / reps * 1e3)
        L.hive_movegen_pair_threshold(0)
        same = torch.equal(bufs["quad"][0], bufs["pair"][0]) and torch.equal(bufs["quad"][1], bufs["pair"][1])
        q = sorted(times["quad"])[3]
        p = sorted(times["pair"])[3]
        print(f"{nb:8d} boards: quad {q:9.2f} us ({nb / q:7.1f} Mboards/s)   pair {p:9.2f} us ({nb / p:7.1f} Mboards/s)   "
              f"pair/quad {p / q:.3f}   same bits: {same}", flush=True)


if __name__ == "__main__":
    main()
