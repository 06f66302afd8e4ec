"""GPU tree search (include/hive_search.h) -- invariants, and agreement with the reference-exact
sequential HivePlayer when the root noise is switched off (both searches are then deterministic)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from mcts_stub import StubPipe, stub_predict   # noqa: E402


def _host_stub_evaluator(planes):
    x = planes.float().cpu().numpy()
    ps, vs = zip(*(stub_predict(x[i]) for i in range(x.shape[0])))
    return torch.from_numpy(np.stack(ps)).cuda(), torch.tensor(vs, dtype=torch.float32).cuda()


def test_search_invariants():
    assert torch.cuda.is_available()
    from hive_alphazero_amd import batch, mcts, packing, playout
    G, sims = 96, 24
    boards = playout.random_positions(G, seed=5)
    B = batch.BoardBatch(G)
    B.import_state(boards)
    rb, rh = B.export_state()
    ts = mcts.TreeSearch(G, sims, _host_stub_evaluator, plane_dtype=torch.float32, seed=3)
    action, policy, sum_n = ts.search(rb, rh)
    torch.cuda.synchronize()
    over, _ = B.terminal()
    mask, count, _ = B.legal()
    m = mask.cpu().numpy().view(np.uint32)
    nodes = ts.node_counts().cpu().numpy()
    st = packing.unpack_boards(rb.cpu().numpy())
    for g in range(G):
        if over[g].item() or st["turn"][g] >= 55:
            assert action[g].item() == -2
            continue
        legal = packing.mask_to_actions(m[g])
        a = action[g].item()
        assert (a in legal) if legal else a == -1
        assert sum_n[g].item() == sims - 1          # the first simulation only evaluates the root
        assert 1 <= nodes[g] <= sims
        pol = policy[g].cpu().numpy()
        assert abs(pol.sum() - 1.0) < 1e-4 or not legal
        assert set(np.nonzero(pol)[0]).issubset(set(legal))
    ts.close(); B.close()


@pytest.mark.parametrize("prefix_seed,plies", [(21, 0), (22, 3), (23, 8), (24, 14)])
def test_search_matches_sequential_reference_without_noise(prefix_seed, plies):
    """noise_eps = 0: the GPU search and solo_play.HivePlayer (pinned to the reference by
    tests/test_mcts_golden.py) must visit the same root edges the same number of times."""
    assert torch.cuda.is_available()
    import hive_alphazero_amd.solo_play as sp
    from hive_alphazero_amd import batch, mcts
    from hive_alphazero_amd.env_hive import GamePlay
    rng = np.random.default_rng(prefix_seed)
    g = GamePlay(1050, 900)
    for _ in range(plies):
        acts = g.actions()
        g.move(int(acts[rng.integers(len(acts))]))
    sims = 40
    sp.SEARCH_THREADS = 1
    old_eps = sp.noise_eps
    sp.noise_eps = 0.0
    try:
        player = sp.HivePlayer(pipes=[StubPipe()])
        player.simulation_num_per_move = sims
        np.random.seed(0)
        player.action(g)
        ref = {int(a): int(s.n) for a, s in player.tree[g.state_key].a.items()}
    finally:
        sp.noise_eps = old_eps
    B = batch.BoardBatch(1)
    B.import_state(g._rec.reshape(1, 64), g._hist.reshape(1, 384))
    rb, rh = B.export_state()
    ts = mcts.TreeSearch(1, sims, _host_stub_evaluator, plane_dtype=torch.float32, noise_eps=0.0)
    action, policy, sum_n = ts.search(rb, rh)
    pol = policy[0].cpu().numpy()
    n = int(sum_n[0].item())
    got = {int(a): int(round(pol[a] * n)) for a in np.nonzero(pol)[0]}
    assert n == sims - 1
    assert got == {a: c for a, c in ref.items() if c > 0}
    ts.close(); B.close()


@pytest.mark.parametrize("prefix_seed,plies", [(44, 24), (45, 32), (52, 24), (55, 48)])
def test_search_merges_transpositions_like_the_reference_dict(prefix_seed, plies):
    """The reference's tree is a dict keyed by state_key (solo_play.py:167-197): two move orders into one position
    share ONE entry.  At 600 simulations from these positions that happens 6-10 times; the GPU search (hash-table
    merging, the simulation's env carried along the path) must return the policy, visit total and move of the
    sequential HivePlayer, and the plain-tree mode must be what deviates."""
    assert torch.cuda.is_available()
    import hive_alphazero_amd.solo_play as sp
    from hive_alphazero_amd import batch, mcts
    from hive_alphazero_amd.env_hive import GamePlay
    sims = 600
    rng = np.random.default_rng(prefix_seed)
    g = GamePlay(1050, 900)
    for _ in range(plies):
        acts = g.actions()
        g.move(int(acts[rng.integers(len(acts))]))
    sp.SEARCH_THREADS = 1
    old_eps = sp.noise_eps
    sp.noise_eps = 0.0
    try:
        player = sp.HivePlayer(pipes=[StubPipe()])
        player.simulation_num_per_move = sims
        np.random.seed(0)
        move, (ref_policy, ref_visits) = player.action(g)       # visit distribution, or the priors when every W < 0
        ref_policy = np.asarray(ref_policy, dtype=np.float64)
    finally:
        sp.noise_eps = old_eps
    B = batch.BoardBatch(1)
    B.import_state(g._rec.reshape(1, 64), g._hist.reshape(1, 384))
    rb, rh = B.export_state()
    out = {}
    for merge in (True, False):
        ts = mcts.TreeSearch(1, sims, _host_stub_evaluator, plane_dtype=torch.float32, noise_eps=0.0, transpositions=merge)
        action, policy, sum_n = ts.search(rb, rh)
        out[merge] = (float(np.abs(policy[0].cpu().numpy().astype(np.float64) - ref_policy).max()), int(action[0].item()),
                      int(sum_n[0].item()), int(ts.transposition_hits()[0].item()))
        ts.close()
    B.close()
    diff, act, n, hits = out[True]
    assert hits >= 4                                       # the case really exercises shared entries
    assert diff < 1e-6 and act == move and n == int(ref_visits) >= sims - 1      # lines that return to the root position add root visits
    assert out[False][3] == 0 and out[False][0] > 1e-3     # without merging the same search comes out differently


def test_selfplay_engine_runs_and_stays_legal():
    assert torch.cuda.is_available()
    from hive_alphazero_amd import mcts
    from hive_alphazero_amd.alpha_net import ChessNet, InferenceNet
    torch.manual_seed(0)
    net = InferenceNet(ChessNet().cuda().eval(), dtype=torch.bfloat16)
    sp = mcts.SelfPlay(32, 8, net, seed=1)
    sp.stagger(seed=2)
    for _ in range(6):
        sp.play_ply()
    torch.cuda.synchronize()
    assert sp.env.illegal_count() == 0
    assert sp.plies == 6 and len(sp.records) == 6
    feats, policy, mover, gid = sp.records[-1]
    assert feats.shape == (32, 144) and policy.shape == (32, 1584)
    sp.close()


@pytest.mark.parametrize("fixture", ["uct.json", "uct_deep.json"])
def test_uct_kernels_reproduce_reference_golden(fixture):
    """SURVEY row a20 on the GPU: UCT_search (HIVE_SEARCH_UCT mode of csrc/hive_search.hip over the env kernels) against the
    TRUE reference's UCTNode search (tests/golden/uct*.json, written by oracle/gen_golden.py from alpha_zero/MCTS_chess.py;
    40 reads from early positions, 120 reads from late ones incl. a finished game as the root): root visit counts, total
    values (fp32, bit for bit) and the chosen move."""
    import json
    import os
    assert torch.cuda.is_available()
    from hive_alphazero_amd.MCTS_chess import UCT_search, get_policy
    from hive_alphazero_amd.env_hive import GamePlay
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", fixture)) as f:
        gold = json.load(f)
    for case in gold["cases"]:
        g = GamePlay(1050, 900)
        for a in case["prefix"]:
            g.move(a)
        best, root, extra = UCT_search(g, case["reads"], _host_stub_evaluator)
        N, W = root.child_number_visits, root.child_total_value
        assert N.dtype == np.float32 and W.dtype == np.float32 and extra is None
        assert [[int(i), float(N[i]), float(W[i])] for i in np.nonzero(N)[0]] == case["visits"]
        pol = get_policy(root)
        assert pol.dtype == np.float32
        if not case["visits"]:                              # the root is a finished game: never expanded, nothing visited
            assert g.game_is_over() and N.sum() == 0 and pol.sum() == 0
            continue
        assert best == case["best"]
        assert abs(float(pol.sum()) - 1.0) < 1e-6
        assert N.sum() == case["reads"] - 1                 # the first read expands the root
        assert set(np.nonzero(root.child_priors)[0]) <= set(g.actions()) and root.action_idxes == g.actions()


def test_uct_kernels_match_sequential_restatement_in_batch():
    """The same search for 24 positions in lock step at 150 reads (deeper trees than the golden file holds, stacked
    beetles, finished games reached inside the tree) against tests/uct_ref.py -- the restatement pinned to the reference by
    tests/test_host_cpu.py -- over the CPU oracle env: N and W of every root edge bit for bit."""
    assert torch.cuda.is_available()
    from oracle_env import OracleGamePlay
    from uct_ref import uct_reads
    from hive_alphazero_amd.MCTS_chess import uct_search_batch
    from hive_alphazero_amd.env_hive import GamePlay
    rng = np.random.default_rng(77)
    envs, recs, hists = [], [], []
    for k in range(24):
        g, o = GamePlay(1050, 900), OracleGamePlay()
        for _ in range(int(rng.integers(0, 46))):
            acts = g.actions()
            if not acts or g.game_is_over():
                break
            a = int(acts[rng.integers(len(acts))])
            g.move(a); o.move(a)
        envs.append(o); recs.append(g._rec.copy()); hists.append(g._hist.copy())
    reads = 150
    best, N, W, P = uct_search_batch(np.stack(recs), np.stack(hists), reads, _host_stub_evaluator)
    N, W, P, best = N.cpu().numpy(), W.cpu().numpy(), P.cpu().numpy(), best.cpu().numpy()
    deep = 0
    for k, o in enumerate(envs):
        rn, rw, rp, rbest = uct_reads(o, reads, stub_predict)
        assert np.array_equal(N[k], rn), k
        assert np.array_equal(W[k], rw), k
        assert np.array_equal(P[k], rp), k
        if rn.sum() > 0:
            assert int(best[k]) == rbest
        deep += int(rn.max() > 20)
    assert deep >= 8                                        # trees several levels deep, not just root fans


def test_uct_search_and_model_api_gpu(golden_games):
    """UCT_search with the real network (InferenceNet fp32 and the plain ChessNet, evaluated NCHW like MCTS_chess.py:133-141)
    and the pipe-served evaluator (api_hive.HiveModelAPI) on the GPU env / GPU net."""
    assert torch.cuda.is_available()
    from hive_alphazero_amd.MCTS_chess import UCT_search, get_policy
    from hive_alphazero_amd.alpha_net import ChessNet, InferenceNet
    from hive_alphazero_amd.api_hive import HiveModelAPI
    from hive_alphazero_amd.env_hive import GamePlay
    torch.manual_seed(0)
    net = ChessNet().cuda().eval()
    inf = InferenceNet(net, dtype=torch.float32)
    g = GamePlay(1050, 900)
    for rec in golden_games[0]["plies"][:6]:
        g.move(rec["a"])
    legal = g.actions()
    best, root, _ = UCT_search(g, 16, inf)
    assert int(best) in legal
    pol = get_policy(root)
    assert abs(pol.sum() - 1.0) < 1e-5 and set(np.nonzero(pol)[0]).issubset(set(legal))
    assert root.child_number_visits.sum() == 15          # the first read expands the root
    best2, root2, _ = UCT_search(g, 16, net)             # plain ChessNet, NCHW fp32 like the reference
    assert np.array_equal(root.child_number_visits, root2.child_number_visits)

    planes = []
    for rec in golden_games[1]["plies"][3:9]:
        pl = np.zeros((12, 12, 56), dtype=np.float64)
        pl.reshape(-1)[rec["planes"]] = 1.0
        pl[:, :, 31] = rec["t"]
        planes.append(pl)
    x = torch.from_numpy(np.stack(planes).astype(np.float32)).cuda()
    p_ref, v_ref = inf(x)
    torch.cuda.synchronize()
    api = HiveModelAPI(inf)
    pipes = [api.create_pipe() for _ in range(6)]
    api.start()
    for i, pl in enumerate(planes):
        pipes[i].send(pl)
    for i, pipe in enumerate(pipes):
        assert pipe.poll(30)
        p, v = pipe.recv()
        assert isinstance(v, float) and p.shape == (1584,)
        assert np.allclose(p, p_ref[i].cpu().numpy(), atol=1e-5) and abs(v - float(v_ref[i])) < 1e-4
    api.stop()


def test_search_with_inflight_slots_virtual_loss():
    """BASELINE config 5 shape: several leaves in flight per tree (virtual loss keeps them apart);
    every simulation is accounted for and the chosen moves stay legal."""
    assert torch.cuda.is_available()
    from hive_alphazero_amd import batch, mcts, packing, playout
    G, sims, slots = 48, 41, 4
    boards = playout.random_positions(G, seed=9)
    B = batch.BoardBatch(G)
    B.import_state(boards)
    rb, rh = B.export_state()
    ts = mcts.TreeSearch(G, sims, _host_stub_evaluator, plane_dtype=torch.float32, seed=5, slots=slots)
    action, policy, sum_n = ts.search(rb, rh)
    torch.cuda.synchronize()
    over, _ = B.terminal()
    mask, _, _ = B.legal()
    m = mask.cpu().numpy().view(np.uint32)
    st = packing.unpack_boards(rb.cpu().numpy())
    nodes = ts.node_counts().cpu().numpy()
    for g in range(G):
        if over[g].item() or st["turn"][g] >= 55:
            continue
        legal = packing.mask_to_actions(m[g])
        assert (action[g].item() in legal) if legal else action[g].item() == -1
        # collisions give their virtual loss back, so visits <= sims - 1 and every visit is a real backup
        assert 0 < sum_n[g].item() <= sims - 1
        assert nodes[g] <= sims and nodes[g] <= ts.max_nodes
        pol = policy[g].cpu().numpy()
        assert set(np.nonzero(pol)[0]).issubset(set(legal))
    ts.close(); B.close()


def test_selfplay_records_match_env_planes(tmp_path):
    """First 'next' row: the rows the GPU self-play engine emits are the reference's wire
    format and their state planes equal GamePlay.encode_board of the recorded positions (replayed
    through the env from the moves implied by consecutive records is not possible, so the planes are
    cross-checked against a second, independent encode of the same packed record + history)."""
    assert torch.cuda.is_available()
    from hive_alphazero_amd import batch, mcts, records

    def cheap_eval(planes):
        B = planes.shape[0]
        p = torch.full((B, 1584), 1.0 / 1584, device="cuda")
        v = torch.zeros((B,), device="cuda")
        return p, v

    sp = mcts.SelfPlay(48, 6, cheap_eval, seed=3, plane_dtype=torch.float32)
    checked = 0
    for _ in range(56):                      # from the opening: rows only exist for games logged from their first ply
        if len(sp.finished_games) >= 3:
            break
        boards, hist = sp.env.export_state()
        want = sp.env.encode(torch.float32, "hwc").cpu().numpy()
        over, _ = sp.env.terminal()
        turn = boards[:, 33].cpu().numpy()
        alive = (over.cpu().numpy() == 0) & (turn < 55)
        can_move = sp.env.legal()[1].cpu().numpy() > 0
        sp.play_ply()
        for g in range(48):
            rec = sp.last_ply_record(g) if alive[g] else None
            if rec is not None:
                words, hw, hlen, t, pol, mover = rec
                got = records.unpack_features(words, t, records.history_planes(hw, hlen))
                assert np.array_equal(got, want[g].astype(np.float64)), g
                # a side without a legal move passes: its recorded policy is empty
                assert (abs(pol.sum() - 1.0) < 1e-4) if can_move[g] else (pol.sum() == 0)
                checked += 1
    sp._retire_finished()
    assert checked > 300 and sp.env.illegal_count() == 0
    assert len(sp.finished_games) > 0
    rows = sp.finished_game_rows(0)
    vw, plies, gid = sp.finished_games[0]
    assert 0 <= gid < 48 + sp.finished and len(rows) == len(plies) >= 7
    assert [r[3] for r in rows if r[3][1] == 1] == [[(len(rows) + 1) // 2, 1], [len(rows) // 2, 1]]     # [game_len, counter]
    assert all(len(r) == 4 and np.asarray(r[0]).shape == (12, 12, 56) and len(r[1]) == 1584 for r in rows)
    assert set(r[2] for r in rows) <= ({-1} if vw == 0 else {1, -1})
    path = records.flush_buffer(rows, str(tmp_path))
    back = records.load_data(path)
    assert len(back) == len(rows)
    taken = sp.drain_finished()
    assert len(taken) >= 1 and sp.finished_games == [] and sp.dropped_games == 0
    sp.close()


def test_selfplay_to_training_loop_closes():
    """GPU self-play rows -> the reference's dataset layout (train.py:10-33: object array of [s, p, v]) ->
    alpha_net.train / Trainer: the loop the reference runs between self_play.py and train.py."""
    assert torch.cuda.is_available()
    from hive_alphazero_amd import mcts
    from hive_alphazero_amd.alpha_net import ChessNet, InferenceNet, Trainer, train
    torch.manual_seed(0)
    net = ChessNet().cuda()
    sp = mcts.SelfPlay(64, 4, InferenceNet(net.eval()), seed=21)
    for _ in range(56):
        sp.play_ply()
        if len(sp.finished_games) >= 3:
            break
    sp._retire_finished()
    assert sp.finished_games
    rows = [r for k in range(min(3, len(sp.finished_games))) for r in sp.finished_game_rows(k)]
    dataset = np.empty((len(rows), 3), dtype=object)
    for i, (state, policy, value, _) in enumerate(rows):
        dataset[i, 0], dataset[i, 1], dataset[i, 2] = np.array(state, dtype=np.float32), np.array(policy, dtype=np.float32), float(value)
    losses = train(net, dataset[:16], epoch_start=0, epoch_stop=2, cpu=0, batch_size=8, log=lambda *_: None)
    assert len(losses) == 2 and all(np.isfinite(l) for l in losses)
    tr = Trainer(net)
    x = torch.from_numpy(np.stack([d.transpose(2, 0, 1) for d in dataset[:8, 0]]))
    pol = torch.from_numpy(np.stack(list(dataset[:8, 1])))
    val = torch.tensor([float(v) for v in dataset[:8, 2]])
    assert np.isfinite(tr.step(x, pol, val))
    sp.close()


def test_selfplay_rows_equal_reference_written_rows():
    """SURVEY 8f-1 on the GPU path: the wire rows mcts.SelfPlay emits for a game -- planes, value, [game_len, counter] --
    against the rows the TRUE reference's self_play_buffer wrote for that same game (tests/golden/selfplay.json.gz).
    The golden file holds rows, not moves, so the moves are first recovered on the CPU oracle env (the action after which
    the next recorded planes appear); slot 0 of a 4-game engine is then forced along them while the other slots play
    freely.  (The visit policies differ by construction: the reference searched with numpy's noise stream.)"""
    import copy
    import gzip
    import json
    import os
    import zlib
    assert torch.cuda.is_available()
    from oracle_env import OracleGamePlay
    from hive_alphazero_amd import mcts
    with gzip.open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "selfplay.json.gz"), "rt") as f:
        gold = json.load(f)
    rows = gold["rows"]
    crc = lambda env: int(zlib.crc32(np.asarray(env.encode_board(), dtype=np.float32).tobytes()))
    env, moves = OracleGamePlay(), []
    assert crc(env) == rows[0]["crc"]
    for k in range(len(rows)):
        found = None
        for a in env.actions():
            nxt = copy.deepcopy(env)
            nxt.move(a)
            if k + 1 < len(rows):
                if crc(nxt) == rows[k + 1]["crc"]:
                    found = a
                    break
            elif nxt.game_is_over() and (nxt.state.winner == (250, 250, 250)) == (gold["value_white"][0] == 1):
                found = a
                break
        assert found is not None, k
        moves.append(found)
        env.move(found)

    def flat_eval(planes):
        B = planes.shape[0]
        return torch.full((B, 1584), 1.0 / 1584, device="cuda"), torch.zeros((B,), device="cuda")

    sp = mcts.SelfPlay(4, 5, flat_eval, seed=9, plane_dtype=torch.float32, game_ids=range(4))
    entry = None
    for k in range(len(moves) + 1):
        forced = torch.tensor([moves[k] if k < len(moves) else -2, -2, -2, -2], dtype=torch.int32)
        sp.play_ply(forced)
        done = [e for e in sp.drain_finished() if e[2] == 0]
        if done:
            entry = done[0]
            break
    assert entry is not None and sp.env.illegal_count() == 0
    got = mcts.SelfPlay.game_rows(entry)
    assert entry[0] == gold["value_white"][0] and len(got) == len(rows)
    for (state, policy, value, lens), row in zip(got, rows):
        assert int(zlib.crc32(np.asarray(state, dtype=np.float32).tobytes())) == row["crc"]
        assert value == row["v"] and lens == row["lens"]
        assert abs(sum(policy) - 1.0) < 1e-4
    sp.close()


@pytest.mark.gpu
def test_skipping_unread_leaf_rows_does_not_change_the_search():
    """TreeSearch(skip_unread_rows=True) (default with InferenceNet): hive_search_leaf_need flags the leaves whose
    prediction hive_search_backup never reads (finished games, length cap, collisions, idle trees) and the evaluator's
    kernels skip those boards.  Same trees, same noise, late-game positions (many finished leaves), one and four leaves
    in flight: actions, visit policies and visit totals must be IDENTICAL to the search that evaluates every row, and the
    rows-evaluated counter must sit between the histogram's "expanded and evaluated" and its "network may have been asked" kinds."""
    assert torch.cuda.is_available()
    from hive_alphazero_amd import batch, mcts, playout
    from hive_alphazero_amd.alpha_net import ChessNet, InferenceNet
    torch.manual_seed(0)
    net = ChessNet().cuda().eval()
    inf = InferenceNet(net, dtype=torch.bfloat16, tune_gemms=False)
    G, sims = 192, 40
    boards = playout.random_positions(G, seed=23)
    B = batch.BoardBatch(G)
    B.import_state(boards)
    rb, rh = B.export_state()
    active = torch.ones((G,), dtype=torch.int8, device="cuda")
    active[::7] = 0                                              # idle trees
    for slots in (1, 4):
        out = {}
        for skip in (False, True):
            ts = mcts.TreeSearch(G, sims, inf, slots=slots, seed=9, skip_unread_rows=skip, share_equal_leaves=False)
            assert ts.skip_unread_rows == skip
            action, policy, sum_n = ts.search(rb, rh, active=active, selfplay=True)
            hist = ts.leaf_histogram().sum(0).cpu().numpy()
            out[skip] = (action.clone(), policy.clone(), sum_n.clone(), hist, int(ts.evals_run.item()), ts.evals_launched)
            ts.close()
        a0, p0, n0, h0, _, launched0 = out[False]
        a1, p1, n1, h1, run1, launched1 = out[True]
        assert torch.equal(a0, a1) and torch.equal(p0, p1) and torch.equal(n0, n1), slots
        assert (h0 == h1).all() and launched0 == launched1
        # root (also a root AT the length cap), expanded + evaluated, created meanwhile by another slot (also a finished one)
        asked = int(h1[1] + h1[2] + h1[7])
        print(f"slots {slots}: {launched1} rows launched, {run1} evaluated, histogram says {asked} were read; kinds {h1.tolist()}")
        assert int(h1[2]) <= run1 <= asked and run1 < launched1
    B.close()


@pytest.mark.gpu
def test_packed_records_equal_the_row_wise_records():
    """SelfPlay(packed_records=True) builds the finished games as packed array batches straight from the per-ply host
    copies (what SelfPlayWorker's children send); SelfPlay's row-wise path cuts the same games out one ply_record at a
    time.  Same seed, same game ids, slots refilled from a longer id stream: every game must come out identical -- features,
    history words, valid-history count, turn, mover, value, and the dense policy rebuilt from the sparse one."""
    assert torch.cuda.is_available()
    from hive_alphazero_amd import mcts, records
    from hive_alphazero_amd.alpha_net import ChessNet, InferenceNet
    torch.manual_seed(0)
    inf = InferenceNet(ChessNet().cuda().eval(), dtype=torch.bfloat16, tune_gemms=False)
    G, total = 48, 80
    out = {}
    for packed in (False, True):
        sp = mcts.SelfPlay(G, 6, inf, seed=77, game_ids=iter(range(total)), packed_records=packed, max_finished_kept=4096)
        got = []
        for _ in range(130):
            sp.play_ply()
            if packed:
                b = sp.drain_finished_packed()
                if b is not None:
                    got += [records.unpack_game(b, g) for g in range(records.packed_games(b))]
            else:
                got += sp.drain_finished()
            if sp.running() == 0:
                break
        assert sp.running() == 0 and sp.dropped_games == 0 and sp.unrecorded_games == 0
        assert sp.env.illegal_count() == 0
        sp.close()
        out[packed] = {g[2]: g for g in got}
    assert sorted(out[False]) == sorted(out[True]) == list(range(total))
    rows = 0
    for gid, a in out[False].items():
        b = out[True][gid]
        assert a[0] == b[0] and len(a[1]) == len(b[1]) and len(a[1]) > 0
        for x, y in zip(a[1], b[1]):
            assert all(np.array_equal(np.asarray(u), np.asarray(v)) for u, v in zip(x, y)), gid
        rows += len(a[1])
    print(f"{total} games, {rows} rows: packed batches == row-wise records")
    # staggered start (the first game of most slots begins mid-game and is not recorded): same games kept, same rows
    kept = {}
    for packed in (False, True):
        sp = mcts.SelfPlay(24, 4, inf, seed=5, game_ids=iter(range(60)), packed_records=packed, max_finished_kept=4096)
        sp.stagger(seed=3)
        got = []
        for _ in range(200):
            sp.play_ply()
            if packed:
                b = sp.drain_finished_packed()
                if b is not None:
                    got += [records.unpack_game(b, g) for g in range(records.packed_games(b))]
            else:
                got += sp.drain_finished()
            if sp.running() == 0:
                break
        assert sp.running() == 0 and sp.dropped_games == 0
        kept[packed] = ({g[2]: g for g in got}, sp.unrecorded_games)
        sp.close()
    assert kept[False][1] == kept[True][1] > 0 and sorted(kept[False][0]) == sorted(kept[True][0])
    for gid, a in kept[False][0].items():
        b = kept[True][0][gid]
        assert a[0] == b[0] and len(a[1]) == len(b[1])
        assert all(np.array_equal(np.asarray(u), np.asarray(v)) for x, y in zip(a[1], b[1]) for u, v in zip(x, y)), gid
    print(f"staggered: {len(kept[True][0])} games recorded, {kept[True][1]} first games of a slot not recorded, in both forms")


@pytest.mark.gpu
def test_sharing_equal_leaves_does_not_change_the_search():
    """TreeSearch(share_equal_leaves=True) (default with InferenceNet): equal leaves of a batch -- 256 trees searching the
    SAME opening position and positions a few plies in, one and four leaves in flight -- are evaluated once
    (hive_leaf_dedup_launch) and the duplicates take the representative's tower output before the heads.  Actions, visit
    policies and totals must be identical to the search that evaluates every needed row, with far fewer rows evaluated."""
    assert torch.cuda.is_available()
    from hive_alphazero_amd import batch, mcts
    from hive_alphazero_amd.alpha_net import ChessNet, InferenceNet
    torch.manual_seed(0)
    inf = InferenceNet(ChessNet().cuda().eval(), dtype=torch.bfloat16)
    G, sims = 256, 30
    B = batch.BoardBatch(G)
    gen = torch.Generator(device="cuda").manual_seed(3)
    for plies in (0, 2):
        for _ in range(plies):                                   # a couple of random moves: partly equal positions
            _, count, lst = B.legal(want_list=True)
            from hive_alphazero_amd.playout import pick_uniform
            a = pick_uniform(count, lst, gen)
            a = torch.where(torch.arange(G, device="cuda") % 4 == 0, a, a[0].expand(G).clone())   # 3 of 4 games follow game 0
            B.step(a, sync=False)
        rb, rh = B.export_state()
        for slots in (1, 4):
            out = {}
            for share in (False, True):
                ts = mcts.TreeSearch(G, sims, inf, slots=slots, seed=5, share_equal_leaves=share)
                assert ts.share_equal_leaves == share
                action, policy, sum_n = ts.search(rb, rh, selfplay=True)
                out[share] = (action.clone(), policy.clone(), sum_n.clone(), int(ts.evals_run.item()))
                ts.close()
            assert all(torch.equal(x, y) for x, y in zip(out[False][:3], out[True][:3])), (plies, slots)
            print(f"plies {plies} slots {slots}: rows evaluated {out[False][3]} -> {out[True][3]}")
            assert out[True][3] < 0.7 * out[False][3]
    assert B.illegal_count() == 0
    B.close()


@pytest.mark.gpu
def test_reusing_stored_evaluations_does_not_change_the_search():
    """TreeSearch(reuse_store=N) keeps (p, v) of evaluated leaves ACROSS searches (hive_leaf_store_*): the reference empties its
    tree on every move (solo_play.py:103-112) and evaluates the positions under the played move again.  Two self-play engines
    on the same 192 games, one with the store: every ply's actions, visit policies and visit totals must be identical bit for
    bit -- a position's (p, v) do not depend on the batch or row it is evaluated in (hive_nn_heads, the bit-identical tower
    forms) -- while from the second ply on rows are served from the store and fewer go through the tower.  New weights
    (InferenceNet.refresh) void the store."""
    assert torch.cuda.is_available()
    from hive_alphazero_amd import mcts
    from hive_alphazero_amd.alpha_net import ChessNet, InferenceNet
    torch.manual_seed(0)
    net = ChessNet().cuda().eval()
    inf = InferenceNet(net)
    assert inf.batch_independent_bits
    G, sims = 192, 24
    plain = mcts.SelfPlay(G, sims, inf, seed=9, keep_records=False, game_ids=range(G))
    kept = mcts.SelfPlay(G, sims, inf, seed=9, keep_records=False, game_ids=range(G), search_options={"reuse_store": 1 << 15})
    assert kept.search._store is not None and plain.search._store is None
    served_before = 0
    for ply in range(5):
        e0p, e0k = int(plain.search.evals_run.item()), int(kept.search.evals_run.item())
        plain.play_ply()
        kept.play_ply()
        for a, b in ((plain.search.action, kept.search.action), (plain.search.policy, kept.search.policy),
                     (plain.search.sum_n, kept.search.sum_n)):
            assert torch.equal(a, b), ply
        served, inserted = kept.search.rows_served()
        ran_p, ran_k = int(plain.search.evals_run.item()) - e0p, int(kept.search.evals_run.item()) - e0k
        print(f"ply {ply}: rows through the tower {ran_p} -> {ran_k}, served from the store {served - served_before}, stored so far {inserted}")
        assert ran_k + (served - served_before) == ran_p          # every served row is a row the plain engine evaluated
        if ply >= 1:
            assert served > served_before and ran_k < ran_p
        served_before = served
    assert plain.env.illegal_count() == 0 and kept.env.illegal_count() == 0
    # new weights: the stored answers are dropped, the searches still agree
    torch.manual_seed(1)
    inf.refresh(ChessNet().cuda().eval())
    plain.play_ply()
    kept.play_ply()
    assert torch.equal(plain.search.action, kept.search.action) and torch.equal(plain.search.policy, kept.search.policy)
    assert kept.search.rows_served()[0] < served_before + G * sims  # (the counters restarted with the cleared store)
    plain.close()
    kept.close()
