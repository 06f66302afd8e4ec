#!/usr/bin/env python3
"""bench.py -- headline benchmark of the Hive hot path on MI355X.

Default workload (BASELINE.json configs[1]): 4096 parallel boards, legal-move generation only.
A "step" is one launch of the movegen kernel over one resident batch of 4096 synthetic positions
(random-playout corpus built on the GPU by the product path itself).  value = boards/s summed over
all ranks (weak scaling: every rank owns its own 4096 boards; the path has no exchange step, so
the only torch.distributed traffic is the barrier and the max-over-ranks of the elapsed time).

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES_PER_BOARD = 262          # 64 B HiveBoard read + 198 B (1584-bit) legal mask written, SURVEY.md 8d
HBM_PEAK_GBS = 8000.0               # /opt/skills/guides/MI355X_MICROARCH.md "HBM3E peak BW 8.0 TB/s spec"


def cpu_baseline(boards_np, budget_s=12.0):
    """The C oracle (kind "port") timed on ONE host core over a bounded sample of the same corpus."""
    import numpy as np
    from oracle import oracle_py as O
    from hive_alphazero_amd import packing
    st = packing.unpack_boards(boards_np)
    n = st["turn"].shape[0]
    O.batch_legal(st["turn"][:64], st["pos"][:64], st["lvl"][:64], st["nmt_mode"][:64], want_masks=False)
    done, t0 = 0, time.perf_counter()
    while True:
        O.batch_legal(st["turn"], st["pos"], st["lvl"], st["nmt_mode"], want_masks=False)
        done += n
        el = time.perf_counter() - t0
        if el >= budget_s:
            break
    return {"value": round(done / el / 1e6, 6), "unit": "Mboards/s", "cores": 1, "kind": "port",
            "sample": f"{done} positions ({done // n} passes over the {n}-board corpus), oracle/hive_oracle.c, 1 thread"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--boards", type=int, default=4096)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sat-boards", type=int, default=1 << 20, help="batch size of the saturated side measurement")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: hive_alphazero_amd has no CPU path")
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import hive_alphazero_amd as h
    from hive_alphazero_amd import playout
    from hive_alphazero_amd._lib import HIVE_MASK_WORDS
    L = h.load()

    n = args.boards
    boards = playout.random_positions(n, seed=1000 + rank, device=local_rank)
    mask = torch.empty((n, HIVE_MASK_WORDS), dtype=torch.int32, device="cuda")
    count = torch.empty((n,), dtype=torch.int32, device="cuda")
    stream = torch.cuda.current_stream()
    sp = ctypes.c_void_p(stream.cuda_stream)
    bp, mp, cp = (ctypes.c_void_p(t.data_ptr()) for t in (boards, mask, count))

    def step():
        rc = L.hive_movegen_launch(bp, n, mp, cp, None, sp)
        if rc != 0:
            raise RuntimeError(L.hive_last_error().decode())

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record(stream)
    for _ in range(args.steps):
        step()
    e1.record(stream)
    barrier()
    wall = time.perf_counter() - t0
    dev_ms = e0.elapsed_time(e1)
    tt = torch.tensor([wall], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    wall_max = float(tt.item())
    mean_legal = float(count.float().mean().item())

    # side measurement: the same kernel on a batch large enough to fill all 256 CUs
    sat = None
    if rank == 0 and args.sat_boards > n:
        reps = args.sat_boards // n
        big = boards.repeat(reps, 1).contiguous()
        nb = big.shape[0]
        bm = torch.empty((nb, HIVE_MASK_WORDS), dtype=torch.int32, device="cuda")
        bc = torch.empty((nb,), dtype=torch.int32, device="cuda")
        args_big = (ctypes.c_void_p(big.data_ptr()), nb, ctypes.c_void_p(bm.data_ptr()), ctypes.c_void_p(bc.data_ptr()), None, sp)
        for _ in range(3):
            L.hive_movegen_launch(*args_big)
        torch.cuda.synchronize()
        s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s0.record(stream)
        for _ in range(10):
            L.hive_movegen_launch(*args_big)
        s1.record(stream)
        torch.cuda.synchronize()
        ms = s0.elapsed_time(s1) / 10
        sat = {"boards_per_launch": nb, "ms_per_launch": round(ms, 4), "Mboards_per_s": round(nb / ms / 1e3, 2),
               "achieved_GBs": round(nb * ALGO_BYTES_PER_BOARD / ms / 1e6, 2),
               "frac_of_hbm_peak": round(nb * ALGO_BYTES_PER_BOARD / ms / 1e6 / HBM_PEAK_GBS, 5)}
        del big, bm, bc

    if rank == 0:
        launch_us = dev_ms * 1e3 / args.steps
        achieved = n * ALGO_BYTES_PER_BOARD / (launch_us * 1e-6) / 1e9
        out = {
            "metric": "legal_move_gen_Mboards_per_s",
            "value": round(world * n * args.steps / wall_max / 1e6, 3),
            "unit": "Mboards/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(wall_max * 1e3 / args.steps, 6),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {"workload": f"movegen_{n}", "boards_per_step_per_gpu": n, "mean_legal_moves": round(mean_legal, 2),
                       "corpus": "GPU random playouts, every ply sampled, seed 1000+rank", "parallelism": f"shard{world}"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": None,
                         "kernel": "hive_env_kernel<false,0,0>", "launch_us": round(launch_us, 3),
                         "algorithmic_bytes_per_launch": n * ALGO_BYTES_PER_BOARD,
                         "note": "VALU/LDS-latency bound at 4096 boards (64 workgroups on 256 CUs); see saturated"},
            "saturated": sat,
        }
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(boards.cpu().numpy())
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
