#!/usr/bin/env python3
"""Golden-vector generator: runs the TRUE reference (imported from /root/reference)
and writes small data-only fixtures under tests/golden/.

TEST INFRASTRUCTURE ONLY.  Runs in the build container (the reference never
travels to the GPU box); the fixtures it writes are plain data (inputs and
expected outputs), never reference source.

Usage (from any cwd; it chdirs to a scratch dir because importing
alpha_zero/alpha_net.py does `mkdir ./datasets/iter3/` in the cwd):

    python3 -B oracle/gen_golden.py tables
    python3 -B oracle/gen_golden.py games   --games 64 --procs 8
    python3 -B oracle/gen_golden.py mcts    --plies 6
    python3 -B oracle/gen_golden.py net

Fixture families:
  tables.json        board geometry: neighbour order per cell, is_straight_line LUT,
                     axial-distance classes, action-slot keys, core_index labels
  games_*.json.gz    random-playout games: per ply the packed position, sorted legal
                     list, next_move_tiles, sparse 56-plane encoding, state_key,
                     terminal flag/winner and the action taken
  mcts.json.gz       HivePlayer (SEARCH_THREADS=1, seeded numpy, stub evaluator)
                     visit counts / chosen action
  mcts_deep.json.gz  the same at 600 simulations without root noise: searches deep enough
                     to run through dict entries shared by two move orders
  net.json           ChessNet seeded-init output checksums
  net_wide.npz       ChessNet outputs (full p[1584], v) on 64 golden positions, for the seeded init and for the same
                     init with peaked policy / stretched value heads
"""
import argparse
import gzip
import json
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
GOLD = os.path.join(REPO, "tests", "golden")
REF = "/root/reference"

sys.dont_write_bytecode = True
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(HERE, "stubs"))
os.chdir(tempfile.mkdtemp(prefix="hive_oracle_"))

import numpy as np  # noqa: E402

SLOT_TYPES = "QBBSSGGGAAA"


def _new_game():
    from settings import WIDTH, HEIGHT
    from hive_engine.env_hive import GamePlay
    return GamePlay(HEIGHT_MAP=HEIGHT - 100, WIDTH_MAP=WIDTH - 500)


def _cell(tile):
    return int(tile.index_xy[0]) * 12 + int(tile.index_xy[1])


def snapshot(g, with_planes=True):
    """Everything observable about the reference env at this ply, as plain ints."""
    from settings import PIECE_WHITE, PIECE_BLACK
    pos, lvl = [], []
    for pset in (g.white_pieces_set, g.black_pieces_set):
        for key, (tile, level, piece) in pset.items():
            if tile.axial_coords == (99, 99):
                pos.append(255)
                lvl.append(0)
            else:
                pos.append(_cell(tile))
                lvl.append(int(level))
                assert tile.pieces[level] is piece
    over = bool(g.game_is_over())
    w = g.state.winner
    rec = {
        "t": int(g.state.turn),
        "pos": pos,
        "lvl": lvl,
        "nmt": sorted(_cell(t) for t in g.next_move_tiles),
        "legal": [int(a) for a in g.actions()],
        "over": over,
        "win": 0 if w is None else (1 if w == PIECE_WHITE else 2),
        "key": g.state_key,
    }
    assert rec["legal"] == sorted(rec["legal"])
    if with_planes:
        planes = g.encode_board()
        assert planes.shape == (12, 12, 56)
        assert np.all(planes[:, :, 31] == g.state.turn)
        pl = planes.copy()
        pl[:, :, 31] = 0
        nz = np.argwhere(pl != 0)
        assert np.all(pl[pl != 0] == 1)
        rec["planes"] = sorted(int((x * 12 + y) * 56 + p) for x, y, p in nz)
    return rec


def choose(g, rng, policy):
    """Pick the next action.  Policies only bias WHICH legal move is taken so that
    rarer situations (beetle stacks, surrounded queens, passes) show up."""
    acts = g.actions()
    if not acts:
        return -1
    if policy == "passy" and g.state.turn > 2 and rng.random() < 0.06:
        return -1            # a voluntary pass: move(-1) is what skip-like callers do (env_hive.py:100-103)
    if policy in ("uniform", "passy") or rng.random() < 0.35:
        return int(acts[rng.integers(len(acts))])
    occ = {}
    for pset in (g.white_pieces_set, g.black_pieces_set):
        for key, (tile, level, piece) in pset.items():
            if tile.axial_coords != (99, 99):
                occ[_cell(tile)] = occ.get(_cell(tile), 0) + 1
    if policy == "beetle":
        cand = [a for a in acts if (a % 11) in (1, 2) and (a // 11) in occ]
        if cand:
            return int(cand[rng.integers(len(cand))])
        cand = [a for a in acts if (a % 11) in (1, 2)]
        if cand and rng.random() < 0.5:
            return int(cand[rng.integers(len(cand))])
    if policy == "attack":
        eset = g.black_pieces_set if g.state.player() == 0 else g.white_pieces_set
        qt = eset["<class 'pieces.Queen'>0"][0]
        if qt.axial_coords != (99, 99):
            near = {_cell(t) for t in qt.adjacent_tiles}
            cand = [a for a in acts if (a // 11) in near]
            if cand:
                return int(cand[rng.integers(len(cand))])
    return int(acts[rng.integers(len(acts))])


def play_game(args):
    seed, policy, with_planes = args
    rng = np.random.default_rng(seed)
    g = _new_game()
    plies = []
    while True:
        rec = snapshot(g, with_planes)
        if rec["over"] or g.state.turn >= 55:
            rec["a"] = None
            plies.append(rec)
            break
        a = choose(g, rng, policy)
        rec["a"] = a
        plies.append(rec)
        g.move(a)
    return {"seed": seed, "policy": policy, "plies": plies}


def cmd_tables(_):
    import move_checker as mc
    g = _new_game()
    tiles = [t for t in g.state.board_tiles if t.axial_coords != (99, 99)]
    assert len(tiles) == 144
    order = [_cell(t) for t in tiles]                      # board_tiles order (state_key order)
    by_cell = {_cell(t): t for t in tiles}
    nbr = [[_cell(a) for a in by_cell[c].adjacent_tiles] for c in range(144)]
    core = [list(by_cell[c].core_index) for c in range(144)]
    start = [c for c in range(144) if type(by_cell[c]).__name__ == "Start_Tile"]
    line = []      # line[a] = list of b with is_straight_line(index_xy[a], index_xy[b])
    dist1 = []     # ordered pairs whose effective dist == 1 (adjacent or axial distance 1)
    distle1 = []   # ordered pairs a!=b whose effective dist is NOT > 1
    for a in range(144):
        ta = by_cell[a]
        line.append([b for b in range(144)
                     if mc.is_straight_line(ta.index_xy, by_cell[b].index_xy)])
        for b in range(144):
            tb = by_cell[b]
            d = mc.axial_distance(ta.axial_coords, tb.axial_coords)
            if ta in tb.adjacent_tiles:
                d = 1
            if d == 1:
                dist1.append([a, b])
            if a != b and not (d > 1):
                distle1.append([a, b])
    out = {
        "board_order": order,
        "nbr": nbr,
        "core_index": core,
        "start_cell": start,
        "line": line,
        "dist_eq_1_pairs": dist1,
        "dist_not_gt_1_pairs": distle1,
        "slot_keys": list(g.white_pieces_set.keys()),
        "first_legal": [int(a) for a in g.actions()],
        "piece_keys_white": [g.pieces_keys[v[2]] for v in g.white_pieces_set.values()],
        "piece_keys_black": [g.pieces_keys[v[2]] for v in g.black_pieces_set.values()],
    }
    with open(os.path.join(GOLD, "tables.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print("tables.json written; dist==1 pairs", len(dist1), "line pairs", sum(map(len, line)))


def cmd_games(a):
    import multiprocessing as mp
    jobs = []
    pols = ["passy"] if a.passy else ["uniform", "beetle", "attack"]
    for i in range(a.games):
        jobs.append((a.seed0 + i, pols[i % len(pols)], not a.no_planes))
    with mp.Pool(a.procs) as pool:
        games = []
        for k, gm in enumerate(pool.imap_unordered(play_game, jobs)):
            games.append(gm)
            print(f"[{k + 1}/{len(jobs)}] seed {gm['seed']} {gm['policy']} plies {len(gm['plies'])} "
                  f"over={gm['plies'][-1]['over']} win={gm['plies'][-1]['win']}", flush=True)
    games.sort(key=lambda x: x["seed"])
    name = a.out or ("games_movegen.json.gz" if a.no_planes else "games_full.json.gz")
    with gzip.open(os.path.join(GOLD, name), "wt", compresslevel=9) as f:
        json.dump({"games": games}, f, separators=(",", ":"))
    npos = sum(len(g["plies"]) for g in games)
    print(name, "games", len(games), "positions", npos)


def stub_predict(planes):
    """Deterministic stand-in for the network: (p[1584] float32, v float) as a pure function of the
    planes.  Uses its own RandomState so the global numpy stream (Dirichlet noise, final choice)
    is consumed only by the search itself.  tests/mcts_stub.py holds the identical function."""
    import zlib
    h = zlib.crc32(np.ascontiguousarray(planes, dtype=np.float32).tobytes())
    rs = np.random.RandomState(h)
    p = rs.dirichlet(np.full(1584, 0.5)).astype(np.float32)
    v = float(rs.uniform(-1.0, 1.0))
    return p, v


class StubPipe:
    def send(self, x):
        self._x = x

    def recv(self):
        return stub_predict(self._x)


def cmd_mcts(a):
    """HivePlayer (woker/solo_play.py) at SEARCH_THREADS=1, seeded numpy, stub evaluator."""
    import contextlib
    import io
    import woker.solo_play as sp
    sp.SEARCH_THREADS = 1
    cases = []
    for ci, (pre_plies, seed) in enumerate([(0, 11), (1, 12), (4, 13), (9, 14), (16, 15), (27, 16)]):
        rng = np.random.default_rng(100 + ci)
        g = _new_game()
        prefix = []
        for _ in range(pre_plies):
            act = choose(g, rng, "uniform")
            prefix.append(act)
            g.move(act)
        player = sp.HivePlayer(pipes=[StubPipe()])
        player.simulation_num_per_move = a.sims
        np.random.seed(seed)
        with contextlib.redirect_stdout(io.StringIO()):
            action, (policy, sum_all) = player.action(g)
        root = player.tree[g.state_key]
        edges = [[int(k), int(v.n), float(v.w), float(v.p)] for k, v in root.a.items()]
        cases.append({"prefix": prefix, "seed": seed, "sims": a.sims, "turn": int(g.state.turn),
                      "action": int(action), "sum_all": float(sum_all),
                      "policy_nz": [[i, float(x)] for i, x in enumerate(policy) if x != 0],
                      "root_edges": edges, "tree_size": len(player.tree)})
        print("case", ci, "turn", g.state.turn, "action", action, "tree", len(player.tree), flush=True)
    # a short self-play segment: four consecutive searched plies from one seed
    g = _new_game()
    np.random.seed(99)
    seq = []
    players = [sp.HivePlayer(pipes=[StubPipe()]), sp.HivePlayer(pipes=[StubPipe()])]
    for ply in range(4):
        pl = players[g.state.player()]
        pl.simulation_num_per_move = a.sims
        with contextlib.redirect_stdout(io.StringIO()):
            action, (policy, sum_all) = pl.action(g)
        seq.append({"action": int(action), "sum_all": float(sum_all)})
        g.move(action)
    with gzip.open(os.path.join(GOLD, "mcts.json.gz"), "wt", compresslevel=9) as f:
        json.dump({"cases": cases, "segment": {"seed": 99, "sims": a.sims, "plies": seq}}, f, separators=(",", ":"))
    print("mcts.json.gz written")


def cmd_uct(a):
    """alpha_zero/MCTS_chess.py::UCTNode driven by UCT_search's loop (:130-151) with the stub evaluator
    (UCT_search itself hard-codes .cuda() at :138, so its five-line loop is restated here)."""
    from alpha_zero.MCTS_chess import UCTNode, DummyNode
    cases = []
    plan = [(300 + ci, pp) for ci, pp in enumerate([0, 3, 11, 22])]
    out_name = "uct.json"
    if getattr(a, "deep", False):          # later positions, more reads: deeper trees, finished games inside the tree
        plan = [(340 + ci, pp) for ci, pp in enumerate([30, 38, 44, 48, 51])]
        out_name = "uct_deep.json"
    for ci, (seed, pre_plies) in enumerate(plan):
        rng = np.random.default_rng(seed)
        g = _new_game()
        prefix = []
        for _ in range(pre_plies):
            if g.game_is_over() or not g.actions():
                break
            act = choose(g, rng, "uniform")
            prefix.append(act)
            g.move(act)
        root = UCTNode(g, move=None, parent=DummyNode())
        for _ in range(a.reads):
            leaf = root.select_leaf()
            p, v = stub_predict(leaf.game.encode_board())
            if leaf.game.game_is_over():
                leaf.backup(v)
                continue
            leaf.expand(p)
            leaf.backup(v)
        nz = np.nonzero(root.child_number_visits)[0]
        cases.append({"prefix": prefix, "reads": a.reads, "best": int(np.argmax(root.child_number_visits)),
                      "visits": [[int(i), float(root.child_number_visits[i]), float(root.child_total_value[i])] for i in nz]})
        print("uct case", ci, cases[-1]["best"], len(nz), flush=True)
    with open(os.path.join(GOLD, out_name), "w") as f:
        json.dump({"cases": cases}, f, separators=(",", ":"))


def cmd_selfplay(a):
    """woker/self_play_with_train.py::self_play_buffer (the working twin of woker/self_play.py)
    at SEARCH_THREADS=1, `sims` simulations, stub evaluator, seeded numpy: one full game."""
    import contextlib
    import io
    import zlib
    import woker.solo_play as sp
    sp.SEARCH_THREADS = 1
    sp.simulation_num_per_move = a.sims
    import woker.self_play_with_train as st
    np.random.seed(a.seed)
    cur = [[StubPipe()]]
    with contextlib.redirect_stdout(io.StringIO()):
        data, value_white = st.self_play_buffer(cur)
    rows = []
    for state, policy, value, lens in data:
        arr = np.asarray(state, dtype=np.float32)
        rows.append({"crc": int(zlib.crc32(arr.tobytes())), "turn": int(arr[0, 0, 31]),
                     "pol": [[i, float(x)] for i, x in enumerate(policy) if x != 0], "v": value, "lens": lens})
    with gzip.open(os.path.join(GOLD, "selfplay.json.gz"), "wt", compresslevel=9) as f:
        json.dump({"seed": a.seed, "sims": a.sims, "value_white": value_white, "rows": rows}, f, separators=(",", ":"))
    print("selfplay.json.gz rows", len(rows), "value_white", value_white)


def recorded_game(gm, drop=None, bots=()):
    """A golden random game as woker/sl.py's recorded-game steps; `drop` removes one ply so that the next
    step is out of turn (exercises skip_turn), `bots` marks plies played by a bot."""
    from hive_engine.config import index_char, index_number
    codes = ["Q", "B1", "B2", "S1", "S2", "G1", "G2", "G3", "A1", "A2", "A3"]
    steps = []
    for i, rec in enumerate(gm["plies"]):
        a = rec["a"]
        if a is None or a < 0 or i == drop:
            continue
        cell, slot = divmod(a, 11)
        steps.append([codes[slot], index_char[cell // 12], index_number[cell % 12], "W" if rec["t"] % 2 == 1 else "B",
                      1 if i in bots else 0])
    return steps


def cmd_sl(a):
    """woker/sl.py::get_buffer on recorded games derived from golden games (one complete, one with a dropped ply)."""
    import contextlib
    import io
    import zlib
    import woker.sl as sl
    with gzip.open(os.path.join(GOLD, "games_full.json.gz"), "rt") as f:
        games = json.load(f)["games"]
    won = next(g for g in games if g["plies"][-1]["over"] and g["plies"][-1]["win"] != 0)
    cases = []
    def with_double_move(seed, at, total):
        """Random recorded game in which the player to move at ply `at` is skipped (their opponent moves twice)."""
        from hive_engine.config import index_char, index_number
        codes = ["Q", "B1", "B2", "S1", "S2", "G1", "G2", "G3", "A1", "A2", "A3"]
        rng = np.random.default_rng(seed)
        g = _new_game()
        steps = []
        for ply in range(total):
            if ply == at:
                g.skip_turn()
            acts = g.actions()
            if not acts:
                break
            act = int(acts[rng.integers(len(acts))])
            cell, slot = divmod(act, 11)
            steps.append([codes[slot], index_char[cell // 12], index_number[cell % 12], "W" if g.player() == 0 else "B", 0])
            g.move(act)
        return steps

    for steps in (recorded_game(won, None, (3, 8)), with_double_move(5, 13, 26), recorded_game(games[2], None, (0,))[:12]):
        with contextlib.redirect_stdout(io.StringIO()):
            data, _ = sl.get_buffer(steps)
        rows = []
        for state, policy, value, lens in data:
            arr = np.asarray(state, dtype=np.float32)
            rows.append({"crc": int(zlib.crc32(arr.tobytes())), "pol": [[i, float(x)] for i, x in enumerate(policy) if x != 0],
                         "v": value, "lens": lens})
        cases.append({"steps": steps, "rows": rows})
        print("sl case: steps", len(steps), "rows", len(rows), "value0", rows[0]["v"] if rows else None, flush=True)
    with gzip.open(os.path.join(GOLD, "sl.json.gz"), "wt", compresslevel=9) as f:
        json.dump({"cases": cases}, f, separators=(",", ":"))


def cmd_mcts_deep(a):
    """HivePlayer of the TRUE reference at 600 simulations without root noise from mid-game positions where several
    descents run through tree entries shared by two move orders (the dict keyed by state_key): pins the transposition
    behaviour that the 50-simulation cases of `mcts` never reach.  Minutes per case (the reference env is ~60 ms/move)."""
    import contextlib
    import io
    import woker.solo_play as sp
    sp.SEARCH_THREADS = 1
    sp.noise_eps = 0.0
    cases = []
    for prefix_seed, plies in [(44, 24), (52, 24)]:
        rng = np.random.default_rng(prefix_seed)
        g = _new_game()
        prefix = []
        for _ in range(plies):
            acts = g.actions()
            act = int(acts[rng.integers(len(acts))])
            prefix.append(act)
            g.move(act)
        player = sp.HivePlayer(pipes=[StubPipe()])
        player.simulation_num_per_move = a.sims
        np.random.seed(0)
        with contextlib.redirect_stdout(io.StringIO()):
            action, (policy, sum_all) = player.action(g)
        root = player.tree[g.state_key]
        cases.append({"prefix_seed": prefix_seed, "prefix": prefix, "sims": a.sims, "turn": int(g.state.turn),
                      "action": int(action), "sum_all": float(sum_all),
                      "policy_nz": [[i, float(x)] for i, x in enumerate(policy) if x != 0],
                      "root_edges": [[int(k), int(v.n), float(v.w)] for k, v in root.a.items()],
                      "tree_size": len(player.tree)})
        print("case", prefix_seed, "turn", g.state.turn, "action", action, "sum_all", sum_all, "tree", len(player.tree), flush=True)
    with gzip.open(os.path.join(GOLD, "mcts_deep.json.gz"), "wt", compresslevel=9) as f:
        json.dump({"cases": cases}, f, separators=(",", ":"))


TRAIN_DIGEST_KEYS = ["conv.conv1.weight", "conv.bn1.weight", "conv.bn1.running_mean", "res_0.conv1.weight", "res_9.bn2.bias",
                     "res_18.conv2.weight", "res_18.bn2.running_var", "outblock.fc.weight", "outblock.fc.bias",
                     "outblock.fc2.weight", "outblock.bn.num_batches_tracked"]


def train_dataset():
    """The 20-row synthetic dataset of the `train` fixture (also rebuilt by tests/test_net.py)."""
    rng = np.random.default_rng(7)
    ds = np.empty((20, 3), dtype=object)
    for i in range(20):
        ds[i, 0] = (rng.random((12, 12, 56)) < 0.1).astype(np.float32)
        p = rng.random(1584).astype(np.float32)
        ds[i, 1] = p / p.sum()
        ds[i, 2] = float(rng.choice([-1.0, 1.0]))
    return ds


def cmd_train(a):
    """alpha_zero/alpha_net.py::train (fp32, CPU): one epoch of ten batches of two rows from a seeded init; the fixture
    holds the reported loss and sums of a few tensors of the trained network (data, no weights)."""
    import contextlib
    import io
    import torch
    from alpha_zero.alpha_net import ChessNet, train
    os.makedirs("model_data", exist_ok=True)
    torch.manual_seed(3)
    net = ChessNet()
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        train(net, train_dataset(), 0, 1, cpu=5, batch_size=2)
    line = [l for l in buf.getvalue().splitlines() if "total loss per batch" in l][-1]
    loss = float(line.rsplit(":", 1)[1])
    sd = net.state_dict()
    digest = {k: [float(sd[k].double().sum()), float((sd[k].double() ** 2).sum())] for k in TRAIN_DIGEST_KEYS}
    with open(os.path.join(GOLD, "train.json"), "w") as f:
        json.dump({"init_seed": 3, "cpu": 5, "batch_size": 2, "epochs": 1, "printed_loss_3dp": loss, "digest": digest}, f, indent=1)
    print("loss", loss, "fc.bias", digest["outblock.fc.bias"])


def cmd_net(a):
    """alpha_zero/alpha_net.py::ChessNet: (1) same-seed init of the build's ChessNet gives identical
    tensors, (2) outputs of the reference net on planes of golden positions (CPU fp32)."""
    import torch
    sys.path.insert(0, REPO)
    from alpha_zero.alpha_net import ChessNet as RefNet
    from hive_alphazero_amd.alpha_net import ChessNet as MyNet
    torch.manual_seed(a.seed)
    ref = RefNet().eval()
    torch.manual_seed(a.seed)
    mine = MyNet().eval()
    sd_r, sd_m = ref.state_dict(), mine.state_dict()
    assert list(sd_r.keys()) == list(sd_m.keys()), "state_dict keys differ"
    for k in sd_r:
        assert torch.equal(sd_r[k], sd_m[k]), k
    mine.load_state_dict(sd_r)          # checkpoint compatibility (train.py:35-38)
    with gzip.open(os.path.join(GOLD, "games_full.json.gz"), "rt") as f:
        games = json.load(f)["games"]
    picks = [(0, 10), (0, 31), (1, 20), (2, 45)]
    xs = []
    for gi, ply in picks:
        rec = games[gi]["plies"][ply]
        pl = np.zeros((12, 12, 56), dtype=np.float32)
        pl.reshape(-1)[rec["planes"]] = 1.0
        pl[:, :, 31] = rec["t"]
        xs.append(pl.transpose(2, 0, 1))
    x = torch.from_numpy(np.ascontiguousarray(np.stack(xs))).contiguous()
    with torch.no_grad():
        p_r, v_r = ref(x)
        p_m, v_m = mine(x)
    assert torch.allclose(p_r, p_m, atol=1e-7) and torch.allclose(v_r, v_m, atol=1e-6)
    out = {"seed": a.seed, "picks": picks, "n_keys": len(sd_r), "n_params": int(sum(v.numel() for v in ref.parameters())),
           "v": [float(t) for t in v_r.view(-1)],
           "p_top": [[[int(i), float(p_r[b, i])] for i in torch.topk(p_r[b], 8).indices] for b in range(len(picks))],
           "p_first16": [[float(t) for t in p_r[b, :16]] for b in range(len(picks))],
           "p_sum": [float(p_r[b].sum()) for b in range(len(picks))],
           "torch": torch.__version__}
    with open(os.path.join(GOLD, "net.json"), "w") as f:
        json.dump(out, f)
    print("net.json written", out["v"])


NET_WIDE_PEAK = {"policy_scale": 30.0, "value_scale": 6.0, "value_shift": 2.9}


def net_wide_picks(games):
    """64 positions of the `games_full` fixture: an early and a late ply of each of the first 32 games."""
    picks = []
    for gi in range(32):
        n = len(games[gi]["plies"])
        for ply in (5 + gi % 7, n - 3 - (gi % 5)):
            picks.append((gi, max(0, min(n - 1, ply))))
    return picks


def peak_state_dict(sd, peak=NET_WIDE_PEAK):
    """The second weight set of the `net_wide` fixture: the seeded init with the policy head's logits scaled (priors
    become peaked: a random-init network is near-uniform, max p ~ 1e-3) and the value head's output stretched and
    centred (v spreads over (-1, 1) instead of [-0.9, -0.1]).  In place; tests/test_net.py applies the same edit."""
    sd["outblock.fc.weight"] *= peak["policy_scale"]
    sd["outblock.fc.bias"] *= peak["policy_scale"]
    sd["outblock.fc2.weight"] *= peak["value_scale"]
    sd["outblock.fc2.bias"].mul_(peak["value_scale"]).add_(peak["value_shift"])
    return sd


def cmd_net_wide(a):
    """alpha_zero/alpha_net.py::ChessNet (the TRUE reference, CPU fp32) on 64 golden positions under two weight sets --
    the seeded init and the same init with peaked heads -- full p[1584] and v per position -> tests/golden/net_wide.npz.
    Gives the reduced-precision engines (bf16 / fp16 leaf evaluator) a signal larger than their tolerance."""
    import torch
    from alpha_zero.alpha_net import ChessNet as RefNet
    with gzip.open(os.path.join(GOLD, "games_full.json.gz"), "rt") as f:
        games = json.load(f)["games"]
    picks = net_wide_picks(games)
    xs = []
    for gi, ply in picks:
        rec = games[gi]["plies"][ply]
        pl = np.zeros((12, 12, 56), dtype=np.float32)
        pl.reshape(-1)[rec["planes"]] = 1.0
        pl[:, :, 31] = rec["t"]
        xs.append(pl.transpose(2, 0, 1))
    x = torch.from_numpy(np.ascontiguousarray(np.stack(xs))).contiguous()
    out = {"picks": np.asarray(picks, dtype=np.int32), "seed": np.int32(a.seed),
           "peak": np.asarray([NET_WIDE_PEAK["policy_scale"], NET_WIDE_PEAK["value_scale"], NET_WIDE_PEAK["value_shift"]], dtype=np.float64)}
    for tag in ("init", "peak"):
        torch.manual_seed(a.seed)
        ref = RefNet().eval()
        if tag == "peak":
            ref.load_state_dict(peak_state_dict(ref.state_dict()))
        with torch.no_grad():
            p, v = ref(x)
        out["p_" + tag] = p.numpy().astype(np.float32)
        out["v_" + tag] = v.view(-1).numpy().astype(np.float32)
        print(tag, "max p: median %.4f min %.4f max %.4f;  v min %.3f max %.3f" % (
            float(p.max(1).values.median()), float(p.max(1).values.min()), float(p.max(1).values.max()),
            float(v.min()), float(v.max())))
    np.savez_compressed(os.path.join(GOLD, "net_wide.npz"), **out)
    print("net_wide.npz written:", os.path.getsize(os.path.join(GOLD, "net_wide.npz")), "bytes")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    sub = ap.add_subparsers(dest="cmd", required=True)
    sub.add_parser("tables")
    pg = sub.add_parser("games")
    pg.add_argument("--games", type=int, default=48)
    pg.add_argument("--procs", type=int, default=8)
    pg.add_argument("--seed0", type=int, default=0)
    pg.add_argument("--no-planes", action="store_true")
    pg.add_argument("--out", default=None)
    pg.add_argument("--passy", action="store_true", help="games with voluntary passes")
    pm = sub.add_parser("mcts")
    pm.add_argument("--sims", type=int, default=50)
    pd = sub.add_parser("mcts_deep")
    pd.add_argument("--sims", type=int, default=600)
    sub.add_parser("train")
    pn = sub.add_parser("net")
    pn.add_argument("--seed", type=int, default=0)
    pw = sub.add_parser("net_wide")
    pw.add_argument("--seed", type=int, default=0)
    pu = sub.add_parser("uct")
    pu.add_argument("--reads", type=int, default=40)
    pu.add_argument("--deep", action="store_true", help="write uct_deep.json: late positions (use --reads 120)")
    ps = sub.add_parser("selfplay")
    ps.add_argument("--sims", type=int, default=5)
    ps.add_argument("--seed", type=int, default=4)
    sub.add_parser("sl")
    a = ap.parse_args()
    {"tables": cmd_tables, "games": cmd_games, "mcts": cmd_mcts, "mcts_deep": cmd_mcts_deep, "train": cmd_train, "net": cmd_net, "net_wide": cmd_net_wide, "uct": cmd_uct,
     "selfplay": cmd_selfplay, "sl": cmd_sl}[a.cmd](a)
