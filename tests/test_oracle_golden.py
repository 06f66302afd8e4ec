"""Pins oracle/hive_oracle.c (the CPU restatement) to golden vectors produced by the
TRUE reference (oracle/gen_golden.py).  CPU-only."""
import numpy as np

from oracle import oracle_py as O


def test_tables(golden_tables):
    nbr, order, line = O.tables()
    assert order.tolist() == golden_tables["board_order"]
    assert nbr.tolist() == golden_tables["nbr"]
    for a in range(144):
        assert np.nonzero(line[a])[0].tolist() == golden_tables["line"][a]
    # dist == 1 <=> adjacency, dist > 1 <=> distinct and not adjacent (move_checker.py:192-201)
    adj = {(a, b) for a in range(144) for b in golden_tables["nbr"][a]}
    assert adj == set(map(tuple, golden_tables["dist_eq_1_pairs"]))
    assert adj == set(map(tuple, golden_tables["dist_not_gt_1_pairs"]))
    assert golden_tables["start_cell"] == [78]


def test_new_game(golden_tables):
    g = O.OracleGame()
    assert g.actions() == golden_tables["first_legal"]
    assert g.state_key() == "." * 144 + "0"


def _check_ply(g, rec, ctx):
    pos, lvl = g.pieces()
    assert g.turn == rec["t"], ctx
    assert pos.tolist() == rec["pos"], ctx
    assert [int(l) if p != 255 else 0 for p, l in zip(pos, lvl)] == rec["lvl"], ctx
    assert g.nmt() == rec["nmt"], ctx
    assert g.actions() == rec["legal"], ctx
    over, win = g.game_is_over()
    assert over == rec["over"] and win == rec["win"], ctx
    assert g.state_key() == rec["key"], ctx
    pl = g.encode_board()
    assert np.all(pl[:, :, 31] == rec["t"]), ctx
    pl[:, :, 31] = 0
    nz = np.argwhere(pl != 0)
    assert np.all(pl[pl != 0] == 1), ctx
    got = sorted(int((x * 12 + y) * 56 + p) for x, y, p in nz)
    if got != rec["planes"]:
        diff = sorted(set(got) ^ set(rec["planes"]))
        raise AssertionError(f"{ctx}: plane mismatch (cell,plane)={[(d // 56, d % 56) for d in diff]}")


def test_games_replay(golden_games):
    npos = 0
    for gm in golden_games:
        g = O.OracleGame()
        for i, rec in enumerate(gm["plies"]):
            _check_ply(g, rec, f"seed {gm['seed']} ply {i}")
            npos += 1
            if rec["a"] is not None:
                g.move(rec["a"])
    assert npos > 3000


def test_import_matches_replay(golden_games):
    """ho_import (packed position -> movegen) agrees with replay for every golden position."""
    for gm in golden_games[:20]:
        for rec in gm["plies"]:
            mode = 1 if rec["t"] == 1 else (2 if rec["t"] == 2 else 0)
            g = O.OracleGame.from_position(rec["t"], rec["pos"], rec["lvl"], mode)
            assert g.actions() == rec["legal"]
            assert g.nmt() == rec["nmt"]
