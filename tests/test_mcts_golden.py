"""HivePlayer (the mirror of woker/solo_play.py) against the reference's own search: seeded numpy,
SEARCH_THREADS = 1, stub evaluator -- visit counts, W, priors, policy and chosen action must be
identical.  CPU variant drives the search with the oracle env (config C1); the GPU variant with
the HIP-backed GamePlay façade."""
import gzip
import json
import os

import numpy as np
import pytest

from mcts_stub import StubPipe

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mcts.json.gz")


def _load():
    with gzip.open(GOLD, "rt") as f:
        return json.load(f)


def _run_cases(make_env, ncases=None):
    import hive_alphazero_amd.solo_play as sp
    sp.SEARCH_THREADS = 1
    data = _load()
    for case in data["cases"][:ncases]:
        g = make_env()
        for a in case["prefix"]:
            g.move(a)
        assert g.state.turn == case["turn"]
        player = sp.HivePlayer(pipes=[StubPipe()])
        player.simulation_num_per_move = case["sims"]
        np.random.seed(case["seed"])
        action, (policy, sum_all) = player.action(g)
        root = player.tree[g.state_key]
        edges = [[int(k), int(v.n), float(v.w), float(v.p)] for k, v in root.a.items()]
        assert edges == case["root_edges"], f"turn {case['turn']}"
        assert action == case["action"]
        assert float(sum_all) == case["sum_all"]
        assert [[i, float(x)] for i, x in enumerate(policy) if x != 0] == case["policy_nz"]
        assert len(player.tree) == case["tree_size"]
    return data


def test_hiveplayer_matches_reference_cpu_oracle_env():
    from oracle_env import OracleGamePlay
    data = _run_cases(OracleGamePlay)
    # the four-ply searched segment (two players, one numpy stream)
    import hive_alphazero_amd.solo_play as sp
    seg = data["segment"]
    g = OracleGamePlay()
    np.random.seed(seg["seed"])
    players = [sp.HivePlayer(pipes=[StubPipe()]), sp.HivePlayer(pipes=[StubPipe()])]
    for rec in seg["plies"]:
        pl = players[g.state.player()]
        pl.simulation_num_per_move = seg["sims"]
        action, (policy, sum_all) = pl.action(g)
        assert action == rec["action"] and float(sum_all) == rec["sum_all"]
        g.move(action)


@pytest.mark.gpu
def test_hiveplayer_matches_reference_gpu_env():
    import torch
    assert torch.cuda.is_available()
    from hive_alphazero_amd.env_hive import GamePlay
    _run_cases(lambda: GamePlay(1050, 900), ncases=4)


@pytest.mark.gpu
def test_gameplay_facade_matches_golden_game(golden_games):
    """The single-game drop-in API (GamePlay) replayed over one golden game: actions(), state_key,
    encode_board(), game_is_over()/winner, deepcopy independence."""
    import copy
    import torch
    assert torch.cuda.is_available()
    from hive_alphazero_amd.env_hive import GamePlay
    from hive_alphazero_amd.config import PIECE_BLACK, PIECE_WHITE
    for gm in (golden_games[0], next(g for g in golden_games if g["plies"][-1]["over"])):
        g = GamePlay(1050, 900)
        for i, rec in enumerate(gm["plies"]):
            assert g.state.turn == rec["t"] and g.actions() == rec["legal"], i
            assert g.state_key == rec["key"], i
            assert g.game_is_over() == rec["over"]
            if rec["over"]:
                want = {0: None, 1: PIECE_WHITE, 2: PIECE_BLACK}[rec["win"]]
                assert g.state.winner == want
            pl = g.encode_board().copy()
            assert pl.dtype == np.float64 and pl.shape == (12, 12, 56)
            pl[:, :, 31] = 0
            got = sorted(int((x * 12 + y) * 56 + p) for x, y, p in np.argwhere(pl != 0))
            assert got == rec["planes"], i
            if rec["a"] is None:
                break
            if i == 5:
                c = copy.deepcopy(g)
                c.move(rec["a"])
                assert g.state.turn == rec["t"] and c.state.turn == rec["t"] + 1
            g.move(rec["a"])
    key, core = g.decode_action(858)
    assert key == "<class 'pieces.Queen'>0" and core == ("N", "13")


@pytest.mark.gpu
def test_sl_ingest_gpu_matches_reference_rows():
    """the batched SL ingest (hive_alphazero_amd.sl.get_buffers: every recorded game advances in the same
    kernel launches) against the rows the TRUE reference's woker/sl.py::get_buffer wrote for the same recorded games
    (tests/golden/sl.json.gz): planes, one-hot / bot-weighted policy, value, [game_len, counter] -- plus a game with an
    illegal step, which must yield no rows without disturbing the others."""
    import zlib
    import torch
    assert torch.cuda.is_available()
    from hive_alphazero_amd.sl import get_buffer, get_buffers
    with gzip.open(os.path.join(os.path.dirname(GOLD), "sl.json.gz"), "rt") as f:
        gold = json.load(f)
    games = [c["steps"] for c in gold["cases"]]
    broken = [list(s) for s in games[0][:9]]
    broken[6] = ["Q", "H", "7", broken[6][3], 0]           # a queen drop far from the hive: not legal
    out = get_buffers(games + [broken])
    assert out[-1] == []
    for data, case in zip(out, gold["cases"]):
        assert len(data) == len(case["rows"]) > 0
        for (state, policy, value, lens), row in zip(data, case["rows"]):
            assert int(zlib.crc32(np.asarray(state, dtype=np.float32).tobytes())) == row["crc"]
            assert [[i, float(x)] for i, x in enumerate(policy) if x != 0] == row["pol"]
            assert value == row["v"] and lens == row["lens"]
    one, same = get_buffer(games[1])
    assert same is games[1] and len(one) == len(gold["cases"][1]["rows"])


DEEP = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mcts_deep.json.gz")


def _deep_cases():
    with gzip.open(DEEP, "rt") as f:
        return json.load(f)["cases"]


def test_hiveplayer_matches_reference_through_transpositions():
    """600 simulations of the TRUE reference without root noise from mid-game positions (tests/golden/mcts_deep.json.gz,
    oracle/gen_golden.py mcts_deep): deep enough that descents run through dict entries shared by two move orders.
    The mirror over the oracle env must reproduce the root statistics, the policy, the move and the number of entries."""
    import hive_alphazero_amd.solo_play as sp
    from oracle_env import OracleGamePlay
    sp.SEARCH_THREADS = 1
    old_eps = sp.noise_eps
    sp.noise_eps = 0.0
    try:
        for case in _deep_cases():
            g = OracleGamePlay()
            for a in case["prefix"]:
                g.move(a)
            assert g.state.turn == case["turn"]
            player = sp.HivePlayer(pipes=[StubPipe()])
            player.simulation_num_per_move = case["sims"]
            np.random.seed(0)
            action, (policy, sum_all) = player.action(g)
            root = player.tree[g.state_key]
            assert [[int(k), int(v.n), float(v.w)] for k, v in root.a.items()] == case["root_edges"]
            assert action == case["action"] and float(sum_all) == case["sum_all"]
            assert [[i, float(x)] for i, x in enumerate(policy) if x != 0] == case["policy_nz"]
            assert len(player.tree) == case["tree_size"]
    finally:
        sp.noise_eps = old_eps


@pytest.mark.gpu
def test_gpu_search_matches_reference_through_transpositions():
    """The same golden cases against the GPU tree search directly (hash-table merging on): policy within fp32
    rounding of the reference's, same move and visit total; and the searches did go through shared entries."""
    import torch
    assert torch.cuda.is_available()
    from hive_alphazero_amd import batch, mcts
    from hive_alphazero_amd.env_hive import GamePlay
    from test_gpu_search import _host_stub_evaluator
    for case in _deep_cases():
        g = GamePlay(1050, 900)
        for a in case["prefix"]:
            g.move(a)
        B = batch.BoardBatch(1)
        B.import_state(g._rec.reshape(1, 64), g._hist.reshape(1, 384))
        rb, rh = B.export_state()
        ts = mcts.TreeSearch(1, case["sims"], _host_stub_evaluator, plane_dtype=torch.float32, noise_eps=0.0)
        action, policy, sum_n = ts.search(rb, rh)
        want = np.zeros(1584)
        for i, x in case["policy_nz"]:
            want[i] = x
        assert np.abs(policy[0].cpu().numpy().astype(np.float64) - want).max() < 1e-6
        assert int(action[0].item()) == case["action"] and float(sum_n[0].item()) == case["sum_all"]
        assert int(ts.transposition_hits()[0].item()) >= 4
        ts.close(); B.close()
