"""A GPU-free stand-in for hive_alphazero_amd.self_play._game_worker -- TEST INFRASTRUCTURE.

Plays no Hive at all: for every game id of its shard it fabricates a "game" whose length, result and rows are a pure
function of (seed, game id) -- which is exactly the property the real worker gets from the kernels' noise keys -- and
reports which GPU the parent assigned to it.  Lets the CPU suite exercise SelfPlayWorker's spawn / shard / gather /
merge / flush / report logic at world size 2."""
import os
import random


def fabricate(seed, game_id):
    rng = random.Random(seed * 1_000_003 + game_id)
    plies = rng.randint(5, 12)
    value_white = rng.choice([1, -1, 0])
    rows = [[[game_id, k], [rng.random()], value_white if k % 2 == 0 else -value_white, [plies, k]] for k in range(plies)]
    return value_white, rows


def worker(rank, world, cfg, out):
    from hive_alphazero_amd.dist import game_id_stream
    ids = list(game_id_stream(rank, world, cfg["total_games"]))
    gpu = os.environ.get("HIP_VISIBLE_DEVICES")
    assert "CUDA_VISIBLE_DEVICES" not in os.environ          # the alias must not be able to contradict the mask
    if cfg["seed"] == 666 and rank == world - 1:
        ids = ids[:-1]                                        # a rank that loses a game (and says so)
        lost = {"dropped": 1, "unrecorded": 0}
    elif cfg["seed"] == 667 and rank == 0:
        ids = ids[1:]                                         # a rank that loses a game silently
        lost = None
    else:
        lost = {"dropped": 0, "unrecorded": 0}
    # finish the games out of order, like lock-step slots do
    order = ids[::2] + ids[1::2]
    for g in order:
        vw, rows = fabricate(cfg["seed"], g)
        rows[0][0].append(gpu)            # smuggle the device assignment out for the test
        out.put(("game", rank, g, vw, rows))
    out.put(("done", rank, len(ids), 0, {"root_evaluated": len(ids)}, lost))


def fabricate_compact(seed, game_id):
    """A compact entry (value_white, plies, game id) that is a pure function of (seed, game id)."""
    import numpy as np
    rng = np.random.default_rng(seed * 1_000_003 + game_id)
    plies = []
    for t in range(int(rng.integers(3, 9))):
        pol = np.zeros(1584, np.float32)
        k = int(rng.integers(1, 12))
        pol[rng.choice(1584, k, replace=False)] = (rng.random(k) + 0.05).astype(np.float32)
        plies.append((rng.integers(0, 2 ** 56, 144, dtype=np.uint64), rng.integers(0, 2 ** 28, (4, 2, 6), dtype=np.uint32) & np.uint32(0x0FFF0FFF),
                      min(t, 4), t + 1, pol, t & 1))
    return (int(rng.integers(-1, 2)), plies, game_id)


def packed_worker(rank, world, cfg, out):
    """Sends finished games the way the real compact child does: packed batches of several games per message."""
    from hive_alphazero_amd import records
    from hive_alphazero_amd.dist import game_id_stream
    ids = list(game_id_stream(rank, world, cfg["total_games"]))
    order = ids[::2] + ids[1::2]
    for lo in range(0, len(order), 3):
        out.put(("games", rank, records.pack_games([fabricate_compact(cfg["seed"], g) for g in order[lo:lo + 3]])))
    out.put(("done", rank, len(ids), 0, {"root_evaluated": len(ids)}, {"dropped": 0, "unrecorded": 0}))


def fabricate_wave(seed, ids, rows_per_game=54):
    """One lock-step wave as the real compact child sends it (mcts.SelfPlay.drain_finished_packed): all games of the wave
    in ONE packed batch, `rows_per_game` plies each.  Built with array operations (a 1024-game wave is 55 k rows, ~90 MB)
    and with the sparsity of real records (a few feature bits per cell, ~50 policy entries per row), so that the parent's
    unpickling, slicing and compression see what they see behind a GPU.  A pure function of (seed, ids)."""
    import numpy as np
    ids = np.asarray(list(ids), dtype=np.int64)
    g, r = len(ids), len(ids) * rows_per_game
    rng = np.random.default_rng(seed * 1_000_003 + int(ids[0]) if g else seed)
    feat = np.zeros((r, 144), np.uint64)
    cells = rng.integers(0, 144, (r, 22))                         # ~22 occupied cells per position
    np.put_along_axis(feat, cells, rng.integers(1, 2 ** 40, (r, 22), dtype=np.uint64), axis=1)
    feat[:, 0] |= np.repeat(ids, rows_per_game).astype(np.uint64) << np.uint64(40)     # the game id travels in the data
    hist = (rng.integers(0, 2 ** 28, (r, 4, 2, 6), dtype=np.uint32) & np.uint32(0x00410041))
    turn = np.tile(np.arange(1, rows_per_game + 1), g)
    meta = np.stack([np.minimum(turn, 4), turn, turn & 1], axis=1).astype(np.uint8)
    per_row = 50
    pol_ptr = np.arange(r + 1, dtype=np.int64) * per_row
    pol_idx = (np.sort(rng.integers(0, 1584 - per_row, (r, per_row)), axis=1) + np.arange(per_row)).astype(np.int16).reshape(-1)
    pol_val = np.full(r * per_row, 1.0 / per_row, np.float32)
    return {"feat": feat, "hist": hist, "meta": meta, "pol_idx": pol_idx, "pol_val": pol_val, "pol_ptr": pol_ptr,
            "game_ptr": np.arange(g + 1, dtype=np.int64) * rows_per_game, "game_val": (ids % 3 - 1).astype(np.int8),
            "game_id": ids}


def soak_worker(rank, world, cfg, out):
    """A rank that hands over real-size waves at a fixed pace: every `wave_period_s` seconds one packed batch holding the next
    `games_per_gpu` games of its shard -- the 8-GPU parent-ingest rehearsal (tests/test_host_cpu.py)."""
    import time
    from hive_alphazero_amd.dist import game_id_stream
    from hive_alphazero_amd.self_play import send_packed, spool_dir
    spool = spool_dir() if cfg.get("spool", True) else None
    ids = list(game_id_stream(rank, world, cfg["total_games"]))
    per_wave, period = cfg["games_per_gpu"], cfg["slots"] / 1000.0       # (the pace travels in the otherwise unused `slots`)
    waves = [ids[lo:lo + per_wave] for lo in range(0, len(ids), per_wave)]
    batches = [fabricate_wave(cfg["seed"], w) for w in waves]             # fabricated up front: the clock sees only the hand-over
    out.put(("ready", rank, time.time(), 0, None))
    t0 = time.time()
    for k, b in enumerate(batches):
        delay = t0 + k * period - time.time()
        if delay > 0:
            time.sleep(delay)
        send_packed(out, rank, b, spool)                   # exactly the real child's hand-over (tmpfs blob, or the queue)
    out.put(("done", rank, len(ids), 0, {"root_evaluated": len(ids)}, {"dropped": 0, "unrecorded": 0}))
