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
