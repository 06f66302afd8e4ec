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


def movegen_counters():
    """PMC counts of the movegen kernel per 4096-board dispatch (instructions, HBM bytes), read from the newest
    profiles/r*_movegen_counters.json -- the file the round's rocprofv3 passes produced (tools/dev/prof_r03.sh); they
    cannot be measured from inside this process.  Returns None if no such file is in the tree."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_movegen_counters.json")))
    if not files:
        return None
    with open(files[-1]) as f:
        c = json.load(f)
    k = c["hive_piece_kernel<false,false>"]
    n = c["boards_per_dispatch"]
    valu = k["SQ_INSTS_VALU"] / n
    out = {"file": os.path.relpath(files[-1], ROOT), "source": c["source"],
           "traffic_bytes_per_board": (c["fetch_size_correction"] * k["FETCH_SIZE_KiB"] + k["WRITE_SIZE_KiB"]) * 1024.0 / n,
           "valu_per_board": valu,
           "valu_issue_peak_mboards": 1024 * 2.4e9 / (valu * 4) / 1e6}      # 1024 SIMDs, 4 cycles per wave-instruction
    sat = c.get("saturated")              # the large-launch form of the kernel (pair layout from 16,384 boards on), counted apart
    if sat:
        ks = sat["hive_piece_kernel<false,false,PairLay>"]
        ns = sat["boards_per_dispatch"]
        out["saturated"] = {"layout": "pair", "valu_per_board": ks["SQ_INSTS_VALU"] / ns,
                            "valu_issue_peak_mboards": 1024 * 2.4e9 / (ks["SQ_INSTS_VALU"] / ns * 4) / 1e6,
                            "traffic_bytes_per_board": (c["fetch_size_correction"] * ks["FETCH_SIZE_KiB"] + ks["WRITE_SIZE_KiB"]) * 1024.0 / ns,
                            "quad_valu_per_board": sat["hive_piece_kernel<false,false,QuadLay>"]["SQ_INSTS_VALU"] / ns}
    return out


def host_cores(cap=16):
    """Host threads worth starting: the affinity mask, cut down to the cgroup CPU quota (a one-GPU box grants 16)."""
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            cores = min(cores, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return max(1, min(cores, cap))


def cpu_baseline(boards_np, budget_s=12.0, all_cores_s=4.0):
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
    out = {"value": round(done / el / 1e6, 6), "unit": "Mboards/s", "cores": 1, "kind": "port",
           "sample": f"{done} positions ({done // n} passes over the {n}-board corpus), oracle/hive_oracle.c, 1 thread"}
    # the same oracle on every host core this process may use (ctypes releases the GIL): one corpus shard per thread
    from concurrent.futures import ThreadPoolExecutor
    cores = host_cores()
    if cores > 1 and all_cores_s > 0:
        bounds = [(n * i // cores, n * (i + 1) // cores) for i in range(cores)]
        def shard(b):
            lo, hi = b
            O.batch_legal(st["turn"][lo:hi], st["pos"][lo:hi], st["lvl"][lo:hi], st["nmt_mode"][lo:hi], want_masks=False)
        done2, t1 = 0, time.perf_counter()
        with ThreadPoolExecutor(cores) as pool:
            while True:
                list(pool.map(shard, bounds))
                done2 += n
                el2 = time.perf_counter() - t1
                if el2 >= all_cores_s:
                    break
        out["all_cores"] = {"value": round(done2 / el2 / 1e6, 6), "unit": "Mboards/s", "cores": cores,
                            "sample": f"{done2} positions, {cores} threads, one corpus shard each"}
    return out


GFLOP_PER_LEAF = 6.560114816         # 3,280,057,408 MAC x 2 per board, SURVEY.md 8a a22
MFMA_PEAK_TFLOPS = 2500.0            # MI355X dense bf16/fp16 matrix peak, MI355X_MICROARCH.md


def _leaf_accounting(hist, launched, executed, el):
    """MFMA roofline of a self-play window.  `launched` = rows of the leaf batches (every simulation hands the evaluator one
    board per game slot, whatever the leaf turned out to be); `executed` = rows the tower really evaluated (the search flags
    finished / capped / idle leaves and the kernels skip their boards: hive_search_leaf_need, mcts.TreeSearch.evals_run);
    `needed` = simulations whose leaf was a position to evaluate (root / new position), from the search's leaf histogram."""
    needed = hist.get("root_evaluated", 0) + hist.get("expanded_evaluated", 0)
    tf = executed * GFLOP_PER_LEAF / el / 1e3
    return {"leaf_kinds": hist, "leaf_rows_launched": launched, "leaf_evals_executed": executed, "leaf_evals_needed": needed,
            "useful_leaf_fraction": round(needed / max(executed, 1), 4),
            "roofline": {"bound": "mfma", "achieved": round(tf, 2), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(tf / MFMA_PEAK_TFLOPS, 5), "traffic": None,
                         "achieved_needed_only": round(needed * GFLOP_PER_LEAF / el / 1e3, 2),
                         "frac_needed_only": round(needed * GFLOP_PER_LEAF / el / 1e3 / MFMA_PEAK_TFLOPS, 5),
                         "note": "achieved counts the FLOPs of the boards the tower evaluated (rows of finished / capped / idle "
                                 "leaves are skipped by the kernels; the two head GEMMs, 2 % of a forward, still run on "
                                 "every row and are counted for evaluated rows only); *_needed_only counts only leaves "
                                 "the search's histogram books as evaluated"}}


def selfplay_measure(args, rank, local_rank, world):
    """BASELINE M1 (games/min at `sims` simulations per move, random-init ChessNet in bf16, root noise on), two ways:

    whole_games   BASELINE configs[2] literally: `games` games from the opening position, every one played to its end
                  (queen surrounded or the 55-turn cap), no replacement; value = games / wall time.  Slots whose game
                  has ended idle until the last game ends, so this is the conservative figure.
    steady_state  the same engine with finished games replaced at once and the games staggered over plies 0..53, i.e.
                  what a long self-play run sustains; finished games COUNTED over the timed window (and the old
                  plies / mean-length extrapolation beside it)."""
    import torch
    from hive_alphazero_amd import dist as hd
    from hive_alphazero_amd import mcts
    from hive_alphazero_amd.alpha_net import ChessNet, InferenceNet
    torch.manual_seed(0)
    base = ChessNet().cuda().eval()
    # "auto" (default): InferenceNet measures the checkpoint's range and takes fp16 if it is safe, else bf16
    net = InferenceNet(base, dtype={"auto": None, "bf16": torch.bfloat16, "fp16": torch.float16}[args.net_dtype])
    net_dtype = net.dtype
    dname = str(net_dtype).replace("torch.", "").replace("bfloat16", "bf16").replace("float16", "fp16")
    out = {"workload": f"selfplay_{args.games}x{args.sims}sims" + (f"_slots{args.slots}" if args.slots > 1 else ""),
           "net": f"ChessNet 20-block ResNet, random init (torch.manual_seed(0)), {dname} channels-last, HIP-graph replay",
           "net_dtype": dname, "net_precision_choice": net.precision_report}
    other = None
    if rank == 0 and world == 1:                 # (N > 1: no rank-0-only legs while the other ranks wait at the all-reduce)
        # leaf-evaluator latency of the two 16-bit engines, interleaved (fp16: three more mantissa bits, same MFMA rate;
        # parity of both against the reference's fp32 outputs: tests/test_net.py::test_inference_net_wide_parity)
        other = InferenceNet(base, dtype=torch.float16 if net_dtype == torch.bfloat16 else torch.bfloat16)
        x = (torch.rand((args.games * args.slots, 12, 12, 56), device="cuda") < 0.1)
        xs = {id(net): x.to(net_dtype), id(other): x.to(other.dtype)}
        for e in (net, other):
            e(xs[id(e)])
        torch.cuda.synchronize()
        lat = {id(net): [], id(other): []}
        for _ in range(6):
            for e in (net, other):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    e(xs[id(e)])
                e1.record()
                torch.cuda.synchronize()
                lat[id(e)].append(e0.elapsed_time(e1) / 5)
        med = lambda v: sorted(v)[len(v) // 2]
        out["leaf_forward_ms"] = {"leaves": args.games * args.slots,
                                  str(net.dtype).replace("torch.", ""): round(med(lat[id(net)]), 4),
                                  str(other.dtype).replace("torch.", ""): round(med(lat[id(other)]), 4)}
        del xs, x

    if args.whole_games:
        lo = rank * args.games                              # global game ids: rank r plays games r*G .. (r+1)*G - 1
        sp = mcts.SelfPlay(args.games, args.sims, net, device=local_rank, slots=args.slots, seed=1234, keep_records=False,
                           game_ids=range(lo, lo + args.games))
        sp.play_ply()                                       # graph capture / GEMM tuning outside the timed region
        torch.cuda.synchronize()
        sp.close()
        def whole_games_leg(search_options, evaluator=net):
            sp = mcts.SelfPlay(args.games, args.sims, evaluator, device=local_rank, slots=args.slots, seed=1234, keep_records=False,
                               game_ids=range(lo, lo + args.games), search_options=search_options)
            torch.cuda.synchronize()
            ev0 = int(sp.search.evals_run.item())
            t0 = time.perf_counter()
            plies = 0
            while True:
                sp.play_ply()
                plies += 1
                if sp.running() == 0 or plies > 60:
                    break
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            launched = plies * args.games * args.sims           # every simulation hands over one board per game slot
            executed = int(sp.search.evals_run.item()) - ev0 if sp.search.skip_unread_rows else launched
            leg = {"games": sp.finished, "wall_s": round(el, 3), "plies_played": plies,
                   "games_per_min": round(sp.finished / el * 60.0, 2),
                   "mean_plies_per_game": round(sp.mean_game_length(), 2), "ms_per_ply": round(el / plies * 1e3, 2),
                   "illegal_moves": sp.env.illegal_count(),
                   "results": {"white": sp.white_wins, "black": sp.black_wins, "draw_or_cap": sp.draws},
                   "leaf_batch": {"skip_unread_rows": sp.search.skip_unread_rows, "share_equal_leaves": sp.search.share_equal_leaves}}
            leg.update(_leaf_accounting(sp.leaf_histogram(), launched, executed, el))
            if sp.search._store is not None:
                # rows answered from evaluations kept across plies: counted apart, never inside `achieved` (the roofline
                # above divides the FLOPs of the rows the tower really evaluated by the wall time)
                served, stored = sp.search.rows_served()
                leg["leaf_rows_served_from_store"] = served
                leg["leaf_rows_stored"] = stored
            sp.close()
            return leg

        whole = whole_games_leg(None)
        if args.every_row:
            # the same games with every launched row evaluated (no row skipped, no equal leaves shared): the search and the
            # games are identical bit for bit (tests/test_gpu_search.py), only the rows the tower computes differ
            every = whole_games_leg({"skip_unread_rows": False})
            out["whole_games_every_row_evaluated"] = {k: every[k] for k in ("games", "wall_s", "games_per_min", "ms_per_ply", "results",
                                                                           "leaf_batch", "leaf_rows_launched", "leaf_evals_executed",
                                                                           "roofline")}
        if args.reuse:
            # the same games with evaluations kept ACROSS plies (the reference empties its tree on every move, solo_play.py:103-112);
            # identical search bit for bit (tests/test_gpu_search.py); off by default in the engine
            reuse = whole_games_leg({"reuse_store": 1 << 17})
            out["whole_games_with_reuse"] = {k: reuse[k] for k in ("games", "wall_s", "games_per_min", "ms_per_ply", "results",
                                                                   "leaf_rows_launched", "leaf_evals_executed",
                                                                   "leaf_rows_served_from_store", "leaf_rows_stored", "roofline")}
        out["whole_games"] = whole
        out["whole_games_" + dname] = {k: whole[k] for k in ("games", "wall_s", "games_per_min", "ms_per_ply", "results", "roofline")}
        out["games_per_min"] = whole["games_per_min"]
        if other is not None and args.both_dtypes:
            # the same 1024 games (same seed, same ids) through the OTHER 16-bit engine: M1 at both precisions
            oname = "bf16" if dname == "fp16" else "fp16"
            sp = mcts.SelfPlay(args.games, args.sims, other, device=local_rank, slots=args.slots, seed=1234, keep_records=False,
                               game_ids=range(lo, lo + args.games))
            sp.play_ply()                                       # graph capture of this engine outside the timed region
            torch.cuda.synchronize()
            sp.close()
            o = whole_games_leg(None, other)
            out["whole_games_" + oname] = {k: o[k] for k in ("games", "wall_s", "games_per_min", "ms_per_ply", "results", "roofline")}
    other = None

    if args.whole_games and args.records:
        # the same whole games with the per-ply records ON (woker/self_play.py:159-160: a row per ply; :178-193 values at the
        # end): features + policy + mover copied to the host every ply, finished games cut out and handed over
        # (drain_finished_packed: the array batches SelfPlayWorker's children send and play_<ts>.npz holds) -- what a
        # producer does; the engine is otherwise identical (same seed, same game ids)
        lo = rank * args.games
        sp = mcts.SelfPlay(args.games, args.sims, net, device=local_rank, slots=args.slots, seed=1234, keep_records=True,
                           game_ids=range(lo, lo + args.games), max_finished_kept=2 * args.games, packed_records=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        plies, games_out, rows_out = 0, 0, 0
        while True:
            sp.play_ply()
            plies += 1
            batch = sp.drain_finished_packed()
            if batch is not None:
                games_out += len(batch["game_val"])
                rows_out += len(batch["meta"])
            if sp.running() == 0 or plies > 60:
                break
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        rec = {"games": sp.finished, "games_with_records": games_out, "rows": rows_out, "wall_s": round(el, 3),
               "plies_played": plies, "games_per_min": round(sp.finished / el * 60.0, 2),
               "ms_per_ply": round(el / plies * 1e3, 2), "illegal_moves": sp.env.illegal_count(),
               "dropped_games": sp.dropped_games, "unrecorded_games": sp.unrecorded_games}
        if "whole_games" in out:
            rec["vs_engine_only"] = round(rec["games_per_min"] / out["whole_games"]["games_per_min"], 4)
        sp.close()
        out["whole_games_records_on"] = rec
        out["games_per_min_engine_only"] = out.get("games_per_min")
        out["games_per_min"] = rec["games_per_min"]          # the producer number is the M1 figure

    if args.selfplay_plies > 0:
        sp = mcts.SelfPlay(args.games, args.sims, net, device=local_rank, slots=args.slots, seed=1234, keep_records=False,
                           game_ids=hd.game_id_stream(rank, world))
        sp.stagger(seed=77 + rank)
        for _ in range(args.selfplay_warmup):
            sp.play_ply()
        torch.cuda.synchronize()
        f0, h0, ev0 = sp.finished, sp.leaf_histogram(), int(sp.search.evals_run.item())
        t0 = time.perf_counter()
        for _ in range(args.selfplay_plies):
            sp.play_ply()
        sp._retire_finished()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        finished = sp.finished - f0
        h1 = sp.leaf_histogram()
        hist = {k: h1[k] - h0[k] for k in h1}
        launched = args.selfplay_plies * args.games * args.sims
        executed = int(sp.search.evals_run.item()) - ev0 if sp.search.skip_unread_rows else launched
        mean_len = (sum(sp.finished_lengths) / len(sp.finished_lengths) - 1.0) if sp.finished_lengths else 54.0
        steady = {"window_plies": args.selfplay_plies, "window_s": round(el, 3), "finished_games_in_window": finished,
                  "games_per_min_counted": round(finished / el * 60.0, 2),
                  "games_per_min_extrapolated": round(args.games * args.selfplay_plies / mean_len / el * 60.0, 2),
                  "mean_plies_per_finished_game": round(mean_len, 2), "ms_per_ply": round(el / args.selfplay_plies * 1e3, 2),
                  "illegal_moves": sp.env.illegal_count(),
                  "results": {"white": sp.white_wins, "black": sp.black_wins, "draw_or_cap": sp.draws}}
        steady.update(_leaf_accounting(hist, launched, executed, el))
        sp.close()
        out["steady_state"] = steady
        # what a long run sustains (finished games replaced at once, games spread over all plies): the figure to quote;
        # `games_per_min` beside it is BASELINE configs[2] literally (1024 games from the opening, lock step, records on)
        out["games_per_min_steady_state"] = steady["games_per_min_counted"]
        out.setdefault("games_per_min", steady["games_per_min_counted"])
        out["leaf_evals_per_s"] = round(executed / el, 1)
    del net
    if args.worker and world == 1 and args.whole_games:
        out["producer_selfplay_worker"] = selfplay_worker_measure(args, local_rank)
    if args.cpu_baseline_selfplay and world == 1:            # host baselines are an N = 1 measurement
        out["cpu_baseline"] = selfplay_cpu_baseline(args.sims)
        out["cpu_baseline"]["all_cores"] = selfplay_cpu_baseline_all_cores(args.sims)
    return out


def selfplay_worker_measure(args, local_rank, gpus=None):
    """M1 as the PRODUCER the reference's woker/self_play.py is (:37-75 pool, :100-112 files): SelfPlayWorker spawns one
    child for this GPU, the child builds the network and the engine, plays `games` whole games with records on and sends
    every finished game through the queue; the parent writes compact play_<ts>.npz files.  Timed from the spawn to the
    last flushed file (`wall_s_total`: includes the child's interpreter start, `import torch`, network build, HIP-graph
    capture and GEMM tuning -- a one-off per run) and from the child's "ready" message (`wall_s_playing`)."""
    import shutil
    import tempfile
    from hive_alphazero_amd.self_play import SelfPlayWorker
    d = tempfile.mkdtemp(prefix="hive_bench_selfplay_")
    try:
        gpus = [local_rank] if gpus is None else list(gpus)
        w = SelfPlayWorker(total_games=args.games * len(gpus), games_per_gpu=args.games, sims=args.sims, gpus=gpus, seed=1234,
                           slots=args.slots, datapath=d, games_per_file=256, report_every=0, row_format="compact",
                           log=lambda *_: None, warmup=True, net_dtype=args.net_dtype, keep_results=False)
        t0 = time.time()
        res = w.start(timeout_s=900)
        t1 = time.time()
        # playing time runs from the LAST child's "ready" (every GPU has its network and engine) to the last flushed file
        ready = max(w.ready_at.values()) if w.ready_at else t0
        files = [os.path.getsize(f) for f in w.files]
        rows = int(sum(w.game_lens))
        return {"gpus": len(gpus), "games": len(res), "rows": rows, "files": len(files), "file_bytes": int(sum(files)),
                "wall_s_total": round(t1 - t0, 3), "child_startup_s": round(ready - t0, 3),
                "child_ready_s_per_rank": {str(r): round(t - t0, 3) for r, t in sorted(w.ready_at.items())},
                "wall_s_playing": round(t1 - ready, 3),
                "games_per_min_playing": round(len(res) / max(t1 - ready, 1e-9) * 60.0, 2),
                "games_per_min_total": round(len(res) / (t1 - t0) * 60.0, 2),
                "parent": w.parent_stats, "net_dtype": {str(r): p.get("dtype") for r, p in sorted(w.precision.items())},
                "row_format": "compact (.npz of packed features + sparse policies; records.rows_from_game expands to the "
                              "reference's JSON rows)",
                "note": "child_startup_s = interpreter + import torch + network + engine + one warm-up ply (graph capture, "
                        "GEMM tuning): paid once per run, not per 1024 games"}
    finally:
        shutil.rmtree(d, ignore_errors=True)


def training_measure(steps, batch=512):
    """Next-row side measurement: the reference's training step (alpha_net.py:117-162, batch 512) through
    hive_alphazero_amd.alpha_net.Trainer -- bf16, hand-written BatchNorm / convolution kernels."""
    import torch
    from hive_alphazero_amd.alpha_net import ChessNet, Trainer
    g = torch.Generator(device="cuda").manual_seed(0)
    x = (torch.rand((batch, 56, 12, 12), device="cuda", generator=g) < 0.1).float()
    pi = torch.softmax(torch.randn((batch, 1584), device="cuda", generator=g), 1)
    z = torch.sign(torch.randn((batch,), device="cuda", generator=g))
    torch.manual_seed(0)
    tr = Trainer(ChessNet().cuda(), ddp=False, freeze_gc=True)
    for _ in range(5):                          # the libraries' first-call searches (stem / head convolutions) end here
        tr.step(x, pi, z)
    torch.cuda.synchronize()
    import gc
    per_step, notes = [], []
    gc_runs = [0]
    def _gc_cb(phase, info):
        if phase == "stop":
            gc_runs[0] += 1
    gc.callbacks.append(_gc_cb)
    for i in range(steps):
        s0, g0 = torch.cuda.memory_stats(), gc_runs[0]
        t0 = time.perf_counter()
        loss = tr.step(x, pi, z)                # (returns the loss as a float: one synchronise per step, like the reference's loop)
        per_step.append(time.perf_counter() - t0)
        s1 = torch.cuda.memory_stats()
        notes.append({"ms": round(per_step[-1] * 1e3, 2), "device_mallocs": s1["num_device_alloc"] - s0["num_device_alloc"],
                      "gc_runs": gc_runs[0] - g0})
    gc.callbacks.remove(_gc_cb)
    # headline = the MEAN (what a training run pays).  Round 2's 106 ms outlier was a full CPython garbage collection
    # landing in a step (profiles/r03_training_step.md); Trainer(freeze_gc=True) keeps those passes short
    el = sum(per_step) / len(per_step)
    return {"workload": f"train_step_batch{batch}", "ms_per_step": round(el * 1e3, 2), "positions_per_s": round(batch / el, 1),
            "ms_per_step_median": round(sorted(per_step)[len(per_step) // 2] * 1e3, 2), "ms_per_step_max": round(max(per_step) * 1e3, 2),
            "TFLOPs_fwd_bwd_as_3x_fwd": round(3 * GFLOP_PER_LEAF * batch / el / 1e3, 1), "dtype": "bf16 (fp32 master weights)",
            "fused_hip_kernels": bool(tr.fused), "loss": round(float(loss), 4), "per_step": notes}


def selfplay_cpu_baseline(sims, budget_s=10.0):
    """The reference's process model on ONE host core: sequential HivePlayer (hive_alphazero_amd.solo_play, pinned
    bit-exact to woker/solo_play.py) over the CPU oracle env, with a stub evaluator -- i.e. env + tree only, the
    network cost is EXCLUDED (the reference ran it on a separate GPU thread).  Whole games from the opening position,
    as many as fit in `budget_s` seconds."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import hive_alphazero_amd.solo_play as sp
    from mcts_stub import StubPipe
    from oracle_env import OracleGamePlay
    sp.SEARCH_THREADS = 1
    np.random.seed(0)
    player = sp.HivePlayer(pipes=[StubPipe()])
    player.simulation_num_per_move = sims
    t0 = time.perf_counter()
    plies = games = 0
    while time.perf_counter() - t0 < budget_s:
        g = OracleGamePlay()
        while not (g.game_is_over() or g.state.turn >= 55) and time.perf_counter() - t0 < budget_s:
            a, _ = player.action(g)
            g.move(a)
            plies += 1
        games += 1
    el = time.perf_counter() - t0
    return {"value": round(60.0 / (el / max(plies, 1) * 54.0), 4), "unit": "games/min", "cores": 1, "kind": "port",
            "sample": f"{plies} searched plies ({games} games from the opening, the last one cut by the {budget_s:.0f} s budget) at "
                      f"{sims} sims, sequential HivePlayer mirror + C oracle env, stub evaluator (network cost excluded), "
                      "per 54-ply game"}


def _cpu_selfplay_proc(a):
    """One host core's worth of the reference's process model (one game per process, woker/self_play.py:54-56)."""
    sims, budget_s, seed = a
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import hive_alphazero_amd.solo_play as sp
    from mcts_stub import StubPipe
    from oracle_env import OracleGamePlay
    sp.SEARCH_THREADS = 1
    np.random.seed(seed)
    player = sp.HivePlayer(pipes=[StubPipe()])
    player.simulation_num_per_move = sims
    t0 = time.perf_counter()
    plies = 0
    while time.perf_counter() - t0 < budget_s:
        g = OracleGamePlay()
        while not (g.game_is_over() or g.state.turn >= 55) and time.perf_counter() - t0 < budget_s:
            a_, _ = player.action(g)
            g.move(a_)
            plies += 1
    return plies, time.perf_counter() - t0


def selfplay_cpu_baseline_all_cores(sims, budget_s=10.0):
    """The one-core baseline above on every host core the box grants: one self-play process per core (spawned: this
    process has a live HIP context), each playing whole games for `budget_s` seconds; stub evaluator, network cost excluded."""
    import multiprocessing as mp
    cores = host_cores()
    with mp.get_context("spawn").Pool(cores) as pool:
        res = pool.map(_cpu_selfplay_proc, [(sims, budget_s, 100 + i) for i in range(cores)])
    plies = sum(r[0] for r in res)
    el = max(r[1] for r in res)
    return {"value": round(plies / 54.0 / el * 60.0, 3), "unit": "games/min", "cores": cores, "kind": "port",
            "sample": f"{plies} searched plies in {el:.1f} s over {cores} processes (one game per process, the reference's "
                      f"model), {sims} sims, stub evaluator (network cost excluded), per 54-ply game"}


ENCODE_BYTES_PER_BOARD = 16128       # 56 x 144 x 2 B planes written per board, bf16


def encode_measure(L, n=65536, steps=20):
    """The one HBM-bound kernel of the path: hive_expand_kernel, the planes writer (GamePlay.encode_board's output,
    env_hive.py:320-447): packed features (1,152 B / board) + history (384 B) + record (64 B) in, 16,128 B of bf16 planes
    out per board.  Also the whole encode (hive_piece_kernel<true> + the writer)."""
    import torch
    from hive_alphazero_amd import playout
    from hive_alphazero_amd._lib import BF16, HWC
    boards = playout.random_positions(4096, seed=4242).repeat(n // 4096, 1).contiguous()
    hist = torch.zeros((n, 384), dtype=torch.uint8, device="cuda")
    ws = torch.empty((n * 144,), dtype=torch.int64, device="cuda")
    planes = torch.empty((n, 12, 12, 56), dtype=torch.bfloat16, device="cuda")
    st = torch.cuda.current_stream()
    sp = ctypes.c_void_p(st.cuda_stream)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    def full():
        rc = L.hive_encode_launch(P(boards), P(hist), n, P(planes), BF16, HWC, P(ws), sp)
        if rc != 0:
            raise RuntimeError(L.hive_last_error().decode())
    def writer():
        rc = L.hive_expand_launch(P(boards), P(hist), P(ws), n, P(planes), BF16, HWC, sp)
        if rc != 0:
            raise RuntimeError(L.hive_last_error().decode())
    out = {"workload": f"encode_{n}", "boards_per_launch": n}
    for name, fn in (("encode (features + planes writer)", full), ("planes writer (hive_expand_kernel)", writer)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(steps):
            fn()
        e1.record(st)
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / steps
        key = "encode" if name.startswith("encode") else "planes_writer"
        out[key] = {"kernel": name, "ms_per_launch": round(ms, 4), "Mboards_per_s": round(n / ms / 1e3, 2)}
        if key == "planes_writer":
            alg = n * (ENCODE_BYTES_PER_BOARD + 1152 + 64)
            traffic, traffic_source = None, None
            import glob
            files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_encode_counters.json")))
            if files:                           # HBM bytes per launch from the round's PMC passes, scaled to this launch's boards
                with open(files[-1]) as f:
                    c = json.load(f)
                k = c["hive_expand_kernel<2,0,true>"]
                traffic = int((c["fetch_size_correction"] * k["FETCH_SIZE_KiB"] + k["WRITE_SIZE_KiB"]) * 1024.0
                              * n / c["boards_per_dispatch"])
                traffic_source = os.path.relpath(files[-1], ROOT) + ": " + c["source"]
            out[key]["roofline"] = {"bound": "hbm", "achieved": round(alg / ms / 1e6, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                    "frac": round(alg / ms / 1e6 / HBM_PEAK_GBS, 4), "traffic": traffic,
                                    "traffic_source": traffic_source,
                                    "algorithmic_bytes_per_launch": alg,
                                    "note": "16,128 B written + 1,216 B read per board (history empty in this corpus); PMC "
                                            "FETCH/WRITE_SIZE passes: profiles/r03_encode_pmc.md"}
    return out


def relaunch_under_torchrun(argv, gpus):
    """`python bench.py --gpus N` without a launcher: start the N ranks under torch.distributed.run as a CHILD process (this
    process has not touched the GPU yet and never will), relay rank 0's JSON line, exit with the child's code."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--windows", type=int, default=25, help="timed regions of --steps launches each; value = their median")
    ap.add_argument("--boards", type=int, default=4096)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sat-boards", type=int, default=1 << 20, help="batch size of the saturated side measurement")
    ap.add_argument("--games", type=int, default=1024, help="concurrent self-play games per GPU (BASELINE configs[2])")
    ap.add_argument("--sims", type=int, default=50)
    ap.add_argument("--slots", type=int, default=1, help="leaves in flight per tree (virtual loss); BASELINE configs[4] uses 250 sims")
    ap.add_argument("--selfplay-plies", type=int, default=40, help="timed plies of the steady-state self-play window (0 = skip)")
    ap.add_argument("--no-whole-games", dest="whole_games", action="store_false",
                    help="skip the whole-game self-play leg (games from the opening to their end)")
    ap.add_argument("--selfplay-warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline-selfplay", dest="cpu_baseline_selfplay", action="store_false")
    ap.add_argument("--train-steps", type=int, default=10, help="timed steps of the training-step side measurement (0 = skip)")
    ap.add_argument("--no-overlap", action="store_true", help="skip the 4-stream overlapped side measurement (profiling runs: "
                    "concurrent launches stretch the per-kernel durations rocprof reports)")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL) on a multi-GPU node; gloo only to rehearse "
                    "the multi-rank path on a single GPU")
    ap.add_argument("--net-dtype", default="auto", choices=["auto", "bf16", "fp16"],
                    help="leaf-evaluator precision of the self-play legs (auto: fp16 if InferenceNet's range probe passes, else bf16)")
    ap.add_argument("--no-reuse", dest="reuse", action="store_false",
                    help="skip the whole-games leg that keeps leaf evaluations across plies (hive_leaf_store_*)")
    ap.add_argument("--no-both-dtypes", dest="both_dtypes", action="store_false",
                    help="skip the whole-games leg at the other 16-bit precision")
    ap.add_argument("--no-every-row", dest="every_row", action="store_false",
                    help="skip the whole-games leg that evaluates every launched row (no row skipping, no shared leaves)")
    ap.add_argument("--no-records", dest="records", action="store_false", help="skip the records-on whole-game self-play leg")
    ap.add_argument("--no-worker", dest="worker", action="store_false", help="skip the SelfPlayWorker (spawned producer) leg")
    ap.add_argument("--encode-boards", type=int, default=65536, help="boards per launch of the planes-writer side measurement (0 = skip)")
    args = ap.parse_args()

    # ---- N ranks: either the driver started us under torch.distributed.run (RANK / WORLD_SIZE set), or we do it ourselves
    if "RANK" not in os.environ:
        if args.gpus > 1:
            raise SystemExit(relaunch_under_torchrun(sys.argv[1:], args.gpus))
    elif int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={os.environ.get('WORLD_SIZE')}: start one rank per GPU "
                         "(python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...)")

    t_start = time.perf_counter()
    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: hive_alphazero_amd has no CPU path")
    local_rank = local_rank % torch.cuda.device_count()          # (rehearsals put several ranks on one GPU)
    torch.cuda.set_device(local_rank)
    if world > 1:
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.dist_backend)

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

    t_setup = time.perf_counter() - t_start
    for _ in range(args.warmup):
        step()

    def timed_windows(fn, windows):
        """`windows` timed regions of exactly --steps launches each, every one bracketed by barrier + synchronize on both
        sides and timed with HIP events on the launch stream (a host clock around a 0.15 ms region mostly measures the
        closing synchronize).  Returns per-window (device seconds, host seconds), max over ranks."""
        rows = []
        for _ in range(windows):
            barrier()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            e0.record(stream)
            for _ in range(args.steps):
                fn()
            e1.record(stream)
            barrier()
            rows.append((e0.elapsed_time(e1) * 1e-3, time.perf_counter() - t0))
        tt = torch.tensor(rows, dtype=torch.float64, device="cuda")
        if world > 1:
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return tt.cpu().tolist()

    def spread(vals):
        v = sorted(vals)
        return {"median": v[len(v) // 2], "min": v[0], "max": v[-1], "windows": len(v)}

    win = timed_windows(step, args.windows)
    dev_s = spread([w[0] for w in win])
    host_s = spread([w[1] for w in win])
    # headline = the MEDIAN window (each window = --steps launches between two barrier + synchronize brackets, max over ranks)
    wall_max, host_wall_max = dev_s["median"], host_s["median"]
    dev_ms = wall_max * 1e3
    mean_legal = float(count.float().mean().item())

    t_side0 = time.perf_counter()
    selfplay = None
    if args.selfplay_plies > 0 or args.whole_games:
        try:
            selfplay = selfplay_measure(args, rank, local_rank, world)
        except Exception as exc:                 # the headline line must still come out
            selfplay = {"error": repr(exc), "games_per_min": 0.0, "leaf_evals_per_s": 0.0}
        if world > 1:
            t = torch.tensor([selfplay.get("games_per_min", 0.0), selfplay.get("leaf_evals_per_s", 0.0)], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            selfplay["games_per_min_all_gpus"] = round(float(t[0].item()), 2)
            selfplay["leaf_evals_per_s_all_gpus"] = round(float(t[1].item()), 1)

    # the PRODUCER on all N GPUs as one thing (woker/self_play.py:37-75,100-112: ONE parent gathers every game and writes the
    # files): rank 0 drives SelfPlayWorker(gpus = every rank's GPU) while the ranks -- their engines closed, their memory
    # released -- wait on the rendezvous store (a host-side wait: no collective kernel spins on the GPUs meanwhile)
    if world > 1 and args.worker and args.whole_games and selfplay is not None and "error" not in selfplay:
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        torch.cuda.synchronize()
        dist.barrier()
        store = dist.distributed_c10d._get_default_store()
        if rank == 0:
            try:
                ndev = torch.cuda.device_count()
                selfplay["producer_selfplay_worker_all_gpus"] = selfplay_worker_measure(args, local_rank,
                                                                                        gpus=[r % ndev for r in range(world)])
            except Exception as exc:
                selfplay["producer_selfplay_worker_all_gpus"] = {"error": repr(exc)}
            store.set("hive_bench_producer_done", "1")
        else:
            import datetime
            store.wait(["hive_bench_producer_done"], datetime.timedelta(seconds=1800))

    # side measurement: legal set AND its compaction into sorted action ids (GamePlay.encode_action, env_hive.py:287-304)
    # -- what GamePlay.actions() returns -- by the fused launch
    with_list = None
    if rank == 0:
        lst = torch.empty((n, 256), dtype=torch.int16, device="cuda")
        lp = ctypes.c_void_p(lst.data_ptr())
        for _ in range(10):
            L.hive_movegen_launch(bp, n, mp, cp, lp, sp)
        torch.cuda.synchronize()
        lwin = []
        for _ in range(args.windows):
            torch.cuda.synchronize()
            l0, l1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            l0.record(stream)
            for _ in range(args.steps):
                L.hive_movegen_launch(bp, n, mp, cp, lp, sp)
            l1.record(stream)
            torch.cuda.synchronize()
            lwin.append(l0.elapsed_time(l1) / args.steps)
        lsp = spread(lwin)
        lms = lsp["median"]
        with_list = {"Mboards_per_s": round(n / lms / 1e3, 2), "ms_per_step": round(lms, 6),
                     "ms_per_step_min": round(lsp["min"], 6), "ms_per_step_max": round(lsp["max"], 6), "windows": lsp["windows"],
                     "note": "ONE launch: hive_piece_kernel<false, true> keeps the destination boards in LDS and its waves build the "
                             "ascending int16 id lists (GamePlay.actions()) behind one barrier; median of the windows"}
    # side measurement: the same 4096-board steps, independent batches issued round-robin on 4 HIP streams
    # (what a self-play engine with several game groups does); NOT the headline value
    overlapped = None
    if rank == 0 and not args.no_overlap:
        streams = [torch.cuda.Stream() for _ in range(4)]
        outs = [(torch.empty_like(mask), torch.empty_like(count)) for _ in streams]
        torch.cuda.synchronize()
        def ostep(i):
            st = streams[i % 4]
            m, c = outs[i % 4]
            L.hive_movegen_launch(bp, n, ctypes.c_void_p(m.data_ptr()), ctypes.c_void_p(c.data_ptr()), None,
                                  ctypes.c_void_p(st.cuda_stream))
        for i in range(16):
            ostep(i)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(args.steps):
            ostep(i)
        torch.cuda.synchronize()
        el = time.perf_counter() - t1
        overlapped = {"streams": 4, "Mboards_per_s": round(n * args.steps / el / 1e6, 2),
                      "ms_per_step": round(el * 1e3 / args.steps, 6)}

    # side measurement: the same kernel on a batch large enough to fill all 256 CUs
    ctr = movegen_counters()
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
        pair_from = L.hive_movegen_pair_threshold(-1)
        cs = None if ctr is None else (ctr.get("saturated") if nb >= pair_from else ctr)
        sat = {"boards_per_launch": nb, "ms_per_launch": round(ms, 4), "Mboards_per_s": round(nb / ms / 1e3, 2),
               "lane_layout": "pair (one board = 2 lanes, 32 boards per workgroup)" if nb >= pair_from else "quad",
               "achieved_GBs": round(nb * ALGO_BYTES_PER_BOARD / ms / 1e6, 2),
               "frac_of_hbm_peak": round(nb * ALGO_BYTES_PER_BOARD / ms / 1e6 / HBM_PEAK_GBS, 5),
               "valu_issue_roof": None if cs is None else {
                   "valu_wave_instr_per_board": round(cs["valu_per_board"], 1),
                   "peak_Mboards_per_s": round(cs["valu_issue_peak_mboards"], 1),
                   "frac": round(nb / ms / 1e3 / cs["valu_issue_peak_mboards"], 4), "source": ctr["file"]}}
        # the same launch held in the quad layout (what round 3 measured here), for the comparison
        L.hive_movegen_pair_threshold(1 << 30)
        try:
            for _ in range(3):
                L.hive_movegen_launch(*args_big)
            s0.record(stream)
            for _ in range(10):
                L.hive_movegen_launch(*args_big)
            s1.record(stream)
            torch.cuda.synchronize()
        finally:
            L.hive_movegen_pair_threshold(0)
        msq = s0.elapsed_time(s1) / 10
        sat["quad_layout"] = {"ms_per_launch": round(msq, 4), "Mboards_per_s": round(nb / msq / 1e3, 2)}
        del big, bm, bc

    # side measurement: the boundary handing over HOST buffers -- pinned boards in, pinned mask + count out, per step
    pcie = None
    if rank == 0 and not args.no_overlap:
        hb = boards.cpu().pin_memory()
        hm, hc = torch.empty(mask.shape, dtype=mask.dtype).pin_memory(), torch.empty(count.shape, dtype=count.dtype).pin_memory()
        db = torch.empty_like(boards)
        dbp = ctypes.c_void_p(db.data_ptr())
        def hstep():
            db.copy_(hb, non_blocking=True)
            L.hive_movegen_launch(dbp, n, mp, cp, None, sp)
            hm.copy_(mask, non_blocking=True)
            hc.copy_(count, non_blocking=True)
        for _ in range(20):
            hstep()
        torch.cuda.synchronize()
        k = max(args.steps // 4, 1)
        t1 = time.perf_counter()
        for _ in range(k):
            hstep()
        torch.cuda.synchronize()
        el = time.perf_counter() - t1
        pcie = {"Mboards_per_s": round(n * k / el / 1e6, 2), "ms_per_step": round(el * 1e3 / k, 6),
                "host_bytes_per_step": int(hb.numel() * hb.element_size() + hm.numel() * 4 + hc.numel() * 4),
                "note": "pinned host buffers, one stream, copies and kernel serialised; never the headline value"}
        assert torch.equal(hm, mask.cpu())

    encode = None
    if rank == 0 and args.encode_boards > 0:
        try:
            encode = encode_measure(L, args.encode_boards)
        except Exception as exc:
            encode = {"error": repr(exc)}

    training = None
    if world == 1 and args.train_steps > 0:                   # single-GPU side measurement (no DDP group to join)
        import gc
        gc.collect()                                          # the self-play engines, their graphs and pools are gone for good
        torch.cuda.empty_cache()
        torch.cuda.synchronize()
        try:
            training = training_measure(args.train_steps)
        except Exception as exc:                 # a side measurement never takes the headline line down
            training = {"error": repr(exc)}

    t_side = time.perf_counter() - t_side0
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
            "ms_per_step_min": round(dev_s["min"] * 1e3 / args.steps, 6),
            "ms_per_step_max": round(dev_s["max"] * 1e3 / args.steps, 6),
            "windows": dev_s["windows"],
            "timing": "median of `windows` timed regions of exactly `steps` launches each, every region between two barrier + "
                      "synchronize brackets, HIP events on the launch stream, max over ranks per region; min / max = the spread",
            "host_wall_ms_per_step": round(host_wall_max * 1e3 / args.steps, 6),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {"workload": f"movegen_{n}", "boards_per_step_per_gpu": n, "mean_legal_moves": round(mean_legal, 2),
                       "corpus": "GPU random playouts, every ply sampled, seed 1000+rank", "parallelism": f"shard{world}"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 6),
                         "traffic": None if ctr is None else int(round(ctr["traffic_bytes_per_board"] * n)),
                         "traffic_source": None if ctr is None else f"{ctr['file']}: {ctr['source']} (separate rocprofv3 --pmc "
                                           "passes of this command, gfx950 FETCH_SIZE correction applied; not measurable in-run)",
                         "kernel": "hive_piece_kernel<false>", "launch_us": round(launch_us, 3),
                         "valu_issue_roof": None if ctr is None else {
                             "valu_wave_instr_per_board": round(ctr["valu_per_board"], 1),
                             "peak_Mboards_per_s": round(ctr["valu_issue_peak_mboards"], 1),
                             "frac": round(n / launch_us / ctr["valu_issue_peak_mboards"], 4),
                             "note": "the bound that applies: 1024 SIMDs x 2.4 GHz / (4 cycles x VALU wave-instructions per board); "
                                     "a 4096-board launch is one 11-wave workgroup per CU, its time the busiest SIMD's issue time "
                                     "plus staging and the dispatch ramp"},
                         "algorithmic_bytes_per_launch": n * ALGO_BYTES_PER_BOARD,
                         "note": "VALU-issue bound (~489 wave-instructions per board in this launch's quad layout), not HBM bound: see saturated.valu_issue_roof; "
                                 "4096 boards = 256 workgroups x 11 waves = one workgroup per CU; the 256 KB corpus is re-read "
                                 "every step, so the read side is served by L2 / Infinity Cache and the HBM label is nominal"},
            "movegen_with_sorted_id_list": with_list,
            "overlapped_4_streams": overlapped,
            "saturated": sat,
            "host_buffers_pcie_inclusive": pcie,
            "encode": encode,
            "selfplay": selfplay,
            "training": training,
        }
        t_cpu0 = time.perf_counter()
        out["cpu_baseline"] = None                              # timed on rank 0 at N = 1 only
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(boards.cpu().numpy())
        # where the process's wall time goes (the timed region is only the K headline steps)
        out["wall_s"] = {"imports_and_corpus": round(t_setup, 2), "timed_region": round(wall_max, 4),
                         "side_measurements_incl_selfplay_training_and_cpu_baselines": round(t_side, 2),
                         "movegen_cpu_baseline": round(time.perf_counter() - t_cpu0, 2)}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
