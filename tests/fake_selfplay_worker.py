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
