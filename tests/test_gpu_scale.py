"""BASELINE-size searches, the production (noise-on) search mode, game-id keyed reproducibility and the façade under the
reference's default threading -- all through the C ABI on the GPU."""
import json
import os
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from mcts_stub import StubPipe, stub_predict   # noqa: E402


def _host_stub_evaluator(planes):
    x = planes.float().cpu().numpy()
    ps, vs = zip(*(stub_predict(x[i]) for i in range(x.shape[0])))
    return torch.from_numpy(np.stack(ps)).cuda(), torch.tensor(vs, dtype=torch.float32).cuda()


def _support_inside_legal(policy, mask):
    """policy fp32[G,1584] > 0 only where the legal set (uint32[G,66] destination boards) has the bit -- on the device."""
    a = torch.arange(1584, device=policy.device)
    cell, slot = a // 11, a % 11
    word = slot * 6 + (cell // 12) // 2
    bit = (((cell // 12) & 1) << 4) | (cell % 12)
    legal = (mask[:, word] >> bit.to(torch.int32)) & 1
    return bool(((policy > 0) & (legal == 0)).sum().item() == 0), legal


@pytest.fixture(scope="module")
def bf16_net():
    from hive_alphazero_amd.alpha_net import ChessNet, InferenceNet
    torch.manual_seed(0)
    return InferenceNet(ChessNet().cuda().eval(), dtype=torch.bfloat16)


def test_config2_1024_games_50_sims_two_plies_and_oracle_lockstep(bf16_net):
    """BASELINE configs[2] at full size: 1024 games x 50 simulations, random-init network, root noise on.  Two plies from the
    opening: every simulation accounted for, policy support inside the legal set, no node pool exhausted, nothing refused
    by the env; the moves of 64 of the games replayed through the CPU oracle in lock step (legal sets after every move)."""
    from oracle import oracle_py as O
    from hive_alphazero_amd import mcts, packing
    G, sims = 1024, 50
    sp = mcts.SelfPlay(G, sims, bf16_net, seed=11, keep_records=False)
    oracle_games = [O.OracleGame() for _ in range(64)]
    for ply in range(2):
        mask, count, _ = sp.env.legal()
        sp.play_ply()
        action, policy, sum_n = sp.search.action, sp.search.policy, sp.search.sum_n
        assert bool((sum_n == sims - 1).all().item())                   # the first simulation opens the root, no collisions
        ok, legal = _support_inside_legal(policy, mask)
        assert ok
        assert bool((legal.gather(1, action.long().view(-1, 1)) == 1).all().item())
        assert float((policy.sum(1) - 1).abs().max().item()) < 1e-4
        nodes = sp.search.node_counts()
        assert int(nodes.max().item()) <= sims and int(nodes.min().item()) >= 1 and sims < sp.search.max_nodes
        acts = action[:64].cpu().tolist()
        after, _, _ = sp.env.legal()
        m = after[:64].cpu().numpy().view(np.uint32)
        for i, g in enumerate(oracle_games):
            assert acts[i] in g.actions()
            g.move(acts[i])
            assert packing.mask_to_actions(m[i]) == g.actions(), (ply, i)
    hist = sp.leaf_histogram()
    assert hist["root_evaluated"] == 2 * G and sum(hist.values()) == 2 * G * sims and hist["collision"] == 0
    assert sp.env.illegal_count() == 0
    sp.close()


def test_config4_1024_games_250_sims_four_slots_one_ply(bf16_net):
    """BASELINE configs[4] at full size: 1024 games x 250 simulations with four leaves in flight per tree (virtual loss),
    one ply from mid-game positions: visits accounted for (collisions give theirs back), pool never exhausted, legal moves."""
    from hive_alphazero_amd import batch, mcts, playout
    G, sims, slots = 1024, 250, 4
    boards = playout.random_positions(G, seed=31)
    B = batch.BoardBatch(G)
    B.import_state(boards)
    rb, rh = B.export_state()
    over, _ = B.terminal()
    mask, count, _ = B.legal()
    ts = mcts.TreeSearch(G, sims, bf16_net, seed=7, slots=slots)
    action, policy, sum_n = ts.search(rb, rh)
    torch.cuda.synchronize()
    live = (over == 0) & (rb[:, 33] < 55)
    assert int(live.sum().item()) > 900
    ok, legal = _support_inside_legal(policy, mask)
    assert ok
    has_move = live & (count > 0)
    assert bool((legal[has_move].gather(1, action[has_move].long().view(-1, 1)) == 1).all().item())
    assert bool((action[live & (count == 0)] == -1).all().item()) and bool((action[~live] == -2).all().item())
    hist = ts.leaf_histogram()
    # every in-flight slot of a live game is one histogram entry; a collision gives its visit back, nothing else is lost
    # (a root that is a finished game is "searched" too: one new-finished-position leaf, then known-finished leaves)
    assert bool((hist.sum(1) == sims).all().item()) and bool((hist[~live][:, [1, 2, 4]] == 0).all().item())
    # (a line that returns to the root position through a transposition passes the root's edges twice: solo_play.py:188)
    hits = ts.transposition_hits()
    plain = live & (hits == 0)
    assert bool((sum_n[plain] == sims - 1 - hist[plain][:, 4]).all().item()) and int(plain.sum().item()) > 100
    assert bool((sum_n[live] >= sims - 1 - hist[live][:, 4]).all().item())
    assert bool((sum_n[live] > sims // 2).all().item())
    nodes = ts.node_counts()
    assert int(nodes.max().item()) <= sims and sims + slots < ts.max_nodes + 1
    ts.close(); B.close()


@pytest.mark.parametrize("alpha,k", [(0.3, 30), (0.5, 50), (0.3, 131)])
def test_kernel_dirichlet_moments_and_marginal(alpha, k):
    """The search's own noise generator (Marsaglia-Tsang gammas with an iteration cap, Box-Muller on fast intrinsics;
    csrc/hive_search.hip) against np.random.dirichlet's law (solo_play.py:322-323, self_play.py:151): 100,000 draws --
    each sums to 1, every coordinate's mean is 1/k and variance (1/k)(1 - 1/k)/(k alpha + 1) within 5 sigma, and one
    coordinate's marginal is Beta(alpha, (k - 1) alpha) by a Kolmogorov-Smirnov test."""
    import ctypes
    from scipy import stats
    import hive_alphazero_amd as h
    L = h.load()
    draws = 100_000
    out = torch.empty((draws, k), dtype=torch.float32, device="cuda")
    rc = L.hive_search_sample_noise(12345, 0, 3, ctypes.c_float(alpha), k, draws, None, ctypes.c_float(0.0),
                                    ctypes.c_void_p(out.data_ptr()), None)
    assert rc == 0, L.hive_last_error()
    torch.cuda.synchronize()
    x = out.double().cpu().numpy()
    assert np.abs(x.sum(1) - 1).max() < 1e-5 and x.min() >= 0
    mean, var = 1.0 / k, (1.0 / k) * (1 - 1.0 / k) / (k * alpha + 1)
    z_mean = (x.mean(0) - mean) / np.sqrt(var / draws)
    assert np.abs(z_mean).max() < 5, z_mean
    # variance of the sample variance from the Beta's fourth central moment
    a, b = alpha, (k - 1) * alpha
    m4 = stats.beta(a, b).moment(4) - 4 * mean * stats.beta(a, b).moment(3) + 6 * mean ** 2 * stats.beta(a, b).moment(2) - 3 * mean ** 4
    z_var = (x.var(0) - var) / np.sqrt((m4 - var ** 2) / draws)
    assert np.abs(z_var).max() < 5, z_var
    # off-diagonal covariance -mean^2/(k alpha + 1): one pair
    cov = np.mean((x[:, 0] - mean) * (x[:, 1] - mean))
    assert abs(cov + mean * mean / (k * alpha + 1)) < 6 * var / np.sqrt(draws)
    ks = stats.kstest(x[:, k // 2], stats.beta(a, b).cdf)
    assert ks.pvalue > 1e-4, ks
    # different games / turns draw different noise; the same key draws the same
    again = torch.empty_like(out)
    L.hive_search_sample_noise(12345, 0, 3, ctypes.c_float(alpha), k, draws, None, ctypes.c_float(0.0),
                               ctypes.c_void_p(again.data_ptr()), None)
    other = torch.empty_like(out)
    L.hive_search_sample_noise(12345, 0, 4, ctypes.c_float(alpha), k, draws, None, ctypes.c_float(0.0),
                               ctypes.c_void_p(other.data_ptr()), None)
    torch.cuda.synchronize()
    assert torch.equal(again, out) and not torch.equal(other, out)
    assert torch.equal(out[1:], _shifted(L, alpha, k, draws)[:-1])        # draw d of first_game 1 = draw d + 1 of first_game 0


def _shifted(L, alpha, k, draws):
    import ctypes
    t = torch.empty((draws, k), dtype=torch.float32, device="cuda")
    L.hive_search_sample_noise(12345, 1, 3, ctypes.c_float(alpha), k, draws, None, ctypes.c_float(0.0),
                               ctypes.c_void_p(t.data_ptr()), None)
    torch.cuda.synchronize()
    return t


def test_root_noise_mixing_on_a_fixed_prior():
    """(1 - eps) p + eps eta with eps = 0.25 on a fixed prior (solo_play.py:323): exactly that combination of the prior
    and the noise the same key draws alone; still a distribution."""
    import ctypes
    import hive_alphazero_amd as h
    L = h.load()
    k, draws, eps = 40, 4096, 0.25
    prior = torch.softmax(torch.randn(k, generator=torch.Generator().manual_seed(0)), 0).cuda()
    eta = torch.empty((draws, k), dtype=torch.float32, device="cuda")
    mix = torch.empty_like(eta)
    args = (99, 1000, 9, ctypes.c_float(0.3), k, draws)
    assert L.hive_search_sample_noise(*args, None, ctypes.c_float(0.0), ctypes.c_void_p(eta.data_ptr()), None) == 0
    assert L.hive_search_sample_noise(*args, ctypes.c_void_p(prior.data_ptr()), ctypes.c_float(eps), ctypes.c_void_p(mix.data_ptr()), None) == 0
    torch.cuda.synchronize()
    want = (1 - eps) * prior.double()[None, :] + eps * eta.double()
    assert float((mix.double() - want).abs().max().item()) < 1e-7
    assert float((mix.sum(1) - 1).abs().max().item()) < 1e-5


def test_selfplay_move_resampling_frequencies():
    """self_play.py:139-157 in the kernel (search_policy_kernel): on turn 3 the move is drawn from
    (1 - e) pi + e Dirichlet(0.5), e = 0.7 - 0.15 * 2 = 0.4.  8192 games searched from the SAME position with the root
    noise off (so pi is the same for all) differ only in their game id: the frequencies of the chosen moves must follow
    E[(1 - e) pi + e eta] = (1 - e) pi + e / k (chi-square, k - 1 degrees of freedom)."""
    from scipy import stats
    from hive_alphazero_amd import batch, mcts
    from hive_alphazero_amd.env_hive import GamePlay
    g = GamePlay(1050, 900)
    for _ in range(2):                                     # one piece each: turn 3, the first turn with a real choice
        g.move(g.actions()[0])
    assert g.state.turn == 3
    G, sims = 8192, 12
    rb = torch.from_numpy(np.repeat(g._rec.reshape(1, 64), G, 0)).cuda()
    rh = torch.from_numpy(np.repeat(g._hist.reshape(1, 384), G, 0)).cuda()

    def flat_eval(planes):
        B = planes.shape[0]
        return torch.full((B, 1584), 1.0 / 1584, device="cuda"), torch.zeros((B,), device="cuda")

    ts = mcts.TreeSearch(G, sims, flat_eval, plane_dtype=torch.float32, seed=5, noise_eps=0.0)
    action, policy, _ = ts.search(rb, rh, selfplay=True)
    torch.cuda.synchronize()
    pi = policy[0].double().cpu().numpy()
    assert bool((policy == policy[0]).all().item())
    legal = g.actions()
    k = len(legal)
    e = 0.7 - int(3 + 1) / 2 * 0.15
    expect = (1 - e) * pi[legal] + e / k
    got = np.bincount(action.cpu().numpy(), minlength=1584)[legal]
    assert got.sum() == G                                   # never an action outside the legal set
    chi = stats.chisquare(got, expect / expect.sum() * G)
    assert chi.pvalue > 1e-4, (chi, k)
    # and without the self-play flag every game plays the argmax
    a2, _, _ = ts.search(rb, rh, selfplay=False)
    assert bool((a2 == a2[0]).all().item())
    ts.close()


def test_game_records_do_not_depend_on_batch_slot_or_size():
    """game i's noise is keyed on (seed, i, turn, simulation) only.  Games 132..163 played as slots 32..63 of a
    64-game engine and as slots 0..31 of a 32-game engine (root noise ON, self-play resampling ON) choose the same moves
    with the same visit policies at every ply -- so a game's record is the same on 1 GPU or 8."""
    from hive_alphazero_amd import mcts
    runs = {}
    for games, first in ((64, 100), (32, 132)):
        sp = mcts.SelfPlay(games, 10, _host_stub_evaluator, seed=2024, plane_dtype=torch.float32, keep_records=False,
                           game_ids=range(first, first + games))
        trace = []
        for _ in range(9):
            sp.play_ply()
            trace.append((sp.search.action.clone(), sp.search.policy.clone()))
        assert sp.env.illegal_count() == 0
        runs[games] = trace
        sp.close()
    differs = False
    for (a64, p64), (a32, p32) in zip(runs[64], runs[32]):
        assert torch.equal(a64[32:], a32) and torch.equal(p64[32:], p32)
        differs |= not torch.equal(a64[:32], a32)
    assert differs                                          # other game ids do play other games


def test_finished_slots_take_the_next_game_id_and_go_idle():
    """A fixed set of game ids: finishing slots take the next id, then fall idle; every id is played exactly once and
    drained with its id; staggered (unlogged) games yield no rows."""
    from hive_alphazero_amd import mcts

    def flat_eval(planes):
        B = planes.shape[0]
        return torch.full((B, 1584), 1.0 / 1584, device="cuda"), torch.zeros((B,), device="cuda")

    sp = mcts.SelfPlay(16, 4, flat_eval, seed=3, plane_dtype=torch.float32, game_ids=range(1000, 1040))
    seen = []
    for _ in range(400):
        sp.play_ply()
        seen += [(e[2], len(e[1])) for e in sp.drain_finished()]
        if sp.running() == 0:
            break
    assert sp.running() == 0 and sp.finished == 40
    assert sorted(i for i, _ in seen) == list(range(1000, 1040))
    assert all(n >= 7 for _, n in seen)                    # whole games from the opening (the queen must be down by turn 8)
    assert sp.env.illegal_count() == 0 and sp.dropped_games == 0
    sp.play_ply()
    assert bool((sp.search.action == -2).all().item())     # idle slots are skipped by every kernel
    sp.close()
    sp2 = mcts.SelfPlay(16, 4, flat_eval, seed=3, plane_dtype=torch.float32, game_ids=range(16))
    sp2.stagger(seed=1)
    for _ in range(60):
        sp2.play_ply()
    assert sp2.finished == 16 and sp2.unrecorded_games + len(sp2.finished_games) == 16
    assert sp2.unrecorded_games >= 12
    sp2.close()


def test_facade_under_the_reference_default_threading():
    """woker/solo_play.py:153-165 searches with a ThreadPoolExecutor; a GamePlay call borrows its own single-board slot
    (env_hive._EnginePool), so eight racing search threads over deep copies of one GamePlay stay consistent: every
    simulation lands, the policy lives on the legal moves, nothing is refused by the env."""
    import hive_alphazero_amd.solo_play as sp
    from hive_alphazero_amd._lib import HiveError
    from hive_alphazero_amd.env_hive import GamePlay, _EnginePool
    rng = np.random.default_rng(8)
    g = GamePlay(1050, 900)
    for _ in range(13):
        acts = g.actions()
        g.move(int(acts[rng.integers(len(acts))]))
    legal = g.actions()
    old_threads, old_eps = sp.SEARCH_THREADS, sp.noise_eps
    try:
        sp.SEARCH_THREADS = 8
        for trial in range(3):
            player = sp.HivePlayer(pipes=[StubPipe() for _ in range(8)])
            player.simulation_num_per_move = 96
            np.random.seed(trial)
            try:
                move, (policy, visits) = player.action(g)
            except HiveError as exc:                       # a torn load -> step -> store would surface here
                raise AssertionError(f"env refused a move under threading: {exc}")
            policy = np.asarray(policy)
            assert move in legal and set(np.nonzero(policy)[0]) <= set(legal)
            assert abs(policy.sum() - 1.0) < 1e-6
            root = player.tree[g.state_key]
            assert sum(e.n for e in root.a.values()) == visits >= 96 - 8      # racing first simulations may all see an empty tree
            for entry in player._table.values():
                if entry.moves is not None:
                    assert (entry.n >= 0).all() and entry.sum_n == int(entry.n.sum())      # every virtual loss taken back
        assert 1 <= _EnginePool.get(None).created <= 9
    finally:
        sp.SEARCH_THREADS, sp.noise_eps = old_threads, old_eps


def test_facade_single_move_latency():
    """Latency of the single-game façade (BASELINE configs[0]-shaped use): one GamePlay.move = load, step, legal set and
    store in one borrowed slot.  Recorded under gpurun_out/ for DESIGN.md; the bound only catches a regression to
    several round trips per call."""
    from hive_alphazero_amd.env_hive import GamePlay
    rng = np.random.default_rng(3)
    g = GamePlay(1050, 900)
    t_move, t_over, t_enc = [], [], []
    for ply in range(40):
        acts = g.actions()
        if not acts or g.game_is_over():
            break
        a = int(acts[rng.integers(len(acts))])
        t0 = time.perf_counter(); g.move(a); t1 = time.perf_counter()
        g.game_is_over(); t2 = time.perf_counter()
        g.encode_board(); t3 = time.perf_counter()
        t_move.append(t1 - t0); t_over.append(t2 - t1); t_enc.append(t3 - t2)
    med = {k: round(float(np.median(v[5:])) * 1e6, 1) for k, v in (("move_us", t_move), ("game_is_over_us", t_over), ("encode_board_us", t_enc))}
    os.makedirs("gpurun_out", exist_ok=True)
    with open(os.path.join("gpurun_out", "facade_latency.json"), "w") as f:
        json.dump(med, f)
    print("facade latency (median us):", med)
    assert med["move_us"] < 5000


def test_selfplay_worker_real_children_on_the_gpu(tmp_path):
    """SURVEY 8e end to end with the REAL producer: SelfPlayWorker spawns two children (both on this box's one GPU, each
    with HIP_VISIBLE_DEVICES set before it starts), every child builds the network, plays its shard of the global game ids
    in lock step on the GPU and streams finished games back; the parent ends up with every game exactly once, ordered by
    id, as reference-format rows, plus the files it flushed."""
    from hive_alphazero_amd.self_play import SelfPlayWorker
    lines = []
    w = SelfPlayWorker(total_games=6, games_per_gpu=4, sims=3, gpus=[0, 0], seed=5, datapath=str(tmp_path), games_per_file=4,
                       report_every=2, log=lines.append)
    res = w.start(timeout_s=600)
    assert list(res) == [0, 1, 2, 3, 4, 5]
    for gid, (value_white, rows) in res.items():
        assert value_white in (-1, 0, 1) and 7 <= len(rows) <= 54
        counters = {"W": 0, "B": 0}
        for k, (state, policy, value, lens) in enumerate(rows):
            side = "W" if k % 2 == 0 else "B"                     # nobody passes this early with every piece in hand
            counters[side] += 1
            assert np.asarray(state).shape == (12, 12, 56) and len(policy) == 1584
            assert abs(sum(policy) - 1.0) < 1e-3 or sum(policy) == 0
            assert value == (-1 if value_white == 0 else (value_white if side == "W" else -value_white))
            assert lens[1] == counters[side]
        assert rows[0][0][0][0][31] == 1 and rows[1][0][0][0][31] == 2           # plane 31 = the raw turn number
    assert len(lines) == 3 and len(w.files) == 2
    assert w.leaf_kinds["root_evaluated"] >= sum(len(r) for _, r in res.values())


def test_noisy_search_follows_the_sequential_reference_process_in_distribution():
    """The production search mode end to end: with the per-simulation root Dirichlet noise ON the GPU search draws from its
    own generator, the reference-exact sequential HivePlayer from numpy's -- the runs cannot be compared one to one, but
    both are samples of the same random process.  512 GPU searches of one position (game ids 0..511) against 160 sequential
    searches (numpy seeds 0..159), 24 simulations each, same stub evaluator: the mean visit policies agree within 4.5
    standard errors on every action, and so does the frequency of the most-chosen move."""
    import hive_alphazero_amd.solo_play as sp
    from hive_alphazero_amd import mcts
    from hive_alphazero_amd.env_hive import GamePlay
    rng = np.random.default_rng(4)
    g = GamePlay(1050, 900)
    for _ in range(9):
        acts = g.actions()
        g.move(int(acts[rng.integers(len(acts))]))
    legal = g.actions()
    sims, G, R = 24, 512, 160
    rb = torch.from_numpy(np.repeat(g._rec.reshape(1, 64), G, 0)).cuda()
    rh = torch.from_numpy(np.repeat(g._hist.reshape(1, 384), G, 0)).cuda()
    cache = {}

    def cached_stub(planes):                                 # the 512 trees meet the same few hundred positions
        x = planes.float().cpu().numpy()
        ps, vs = [], []
        for i in range(x.shape[0]):
            key = x[i].tobytes()
            if key not in cache:
                cache[key] = stub_predict(x[i])
            ps.append(cache[key][0]); vs.append(cache[key][1])
        return torch.from_numpy(np.stack(ps)).cuda(), torch.tensor(vs, dtype=torch.float32).cuda()

    ts = mcts.TreeSearch(G, sims, cached_stub, plane_dtype=torch.float32, seed=77)
    action, policy, sum_n = ts.search(rb, rh)
    gp = policy.double().cpu().numpy()[:, legal]
    ga = action.cpu().numpy()
    ts.close()
    old_threads = sp.SEARCH_THREADS
    sp.SEARCH_THREADS = 1
    try:
        rp, ra = [], []
        for seed in range(R):
            player = sp.HivePlayer(pipes=[StubPipe()])
            player.simulation_num_per_move = sims
            np.random.seed(seed)
            move, (pol, _) = player.action(g)
            rp.append(np.asarray(pol, dtype=np.float64)[legal]); ra.append(move)
    finally:
        sp.SEARCH_THREADS = old_threads
    rp = np.stack(rp)
    se = np.sqrt(gp.var(0) / G + rp.var(0) / R) + 1e-4
    z = (gp.mean(0) - rp.mean(0)) / se
    assert np.abs(z).max() < 4.5, (z, gp.mean(0), rp.mean(0))
    top = int(np.bincount(np.asarray(ra), minlength=1584).argmax())
    fg, fr = float((ga == top).mean()), float((np.asarray(ra) == top).mean())
    assert abs(fg - fr) < 4.5 * np.sqrt(fg * (1 - fg) / G + fr * (1 - fr) / R) + 0.02, (fg, fr)


def test_selfplay_worker_compact_game_files(tmp_path):
    """The format that keeps up with the GPU: row_format="compact" keeps finished games as they leave the device (packed
    features, history bitboards, sparse policy; ~1.5 KB per row instead of ~32 KB of JSON text) and writes play_<ts>.npz;
    expanding them gives the same kind of rows the JSON route gives, and the trainer's arrays directly."""
    from hive_alphazero_amd import records
    from hive_alphazero_amd.self_play import SelfPlayWorker
    w = SelfPlayWorker(total_games=4, games_per_gpu=4, sims=3, gpus=[0], seed=5, datapath=str(tmp_path), games_per_file=2,
                       report_every=0, row_format="compact")
    res = w.start(timeout_s=600)
    assert list(res) == [0, 1, 2, 3] and len(w.files) == 2 and all(f.endswith(".npz") for f in w.files)
    loaded = [g for f in w.files for g in records.load_games(f)]
    assert sorted(g[2] for g in loaded) == [0, 1, 2, 3]
    by_id = {g[2]: g for g in loaded}
    n_rows = 0
    for gid, entry in res.items():
        rows = records.rows_from_game(entry)
        assert json.dumps(rows) == json.dumps(records.rows_from_game(by_id[gid]))
        assert np.asarray(rows[0][0]).shape == (12, 12, 56) and rows[0][0][0][0][31] == 1
        n_rows += len(rows)
    states, policies, values = records.dataset_from_games(loaded)
    assert states.shape == (n_rows, 12, 12, 56) and policies.shape == (n_rows, 1584) and values.shape == (n_rows,)
    assert sum(os.path.getsize(f) for f in w.files) < 4000 * n_rows


def test_facade_refuses_illegal_moves_and_keeps_its_state():
    """GamePlay.move of an action outside the legal set: HIVE_E_ILLEGAL from hive_single_advance, the position, its legal list
    and state_key untouched (the reference applies such a move blindly, env_hive.py:129-144); the handle stays usable."""
    from hive_alphazero_amd import HiveError
    from hive_alphazero_amd.env_hive import GamePlay
    g = GamePlay(1050, 900)
    g.move(g.actions()[0])
    before = (g._rec.copy(), g._hist.copy(), list(g.actions()), g.state_key, g.state.turn)
    bad = next(a for a in range(1584) if a not in g.actions())
    for _ in range(2):                                      # twice: the refused-move counter of the handle must not stick
        with pytest.raises(HiveError) as ei:
            g.move(bad)
        assert ei.value.code == -3
        assert np.array_equal(g._rec, before[0]) and np.array_equal(g._hist, before[1])
        assert g.actions() == before[2] and g.state_key == before[3] and g.state.turn == before[4]
    g.move(g.actions()[-1])
    assert g.state.turn == before[4] + 1 and not g.game_is_over()


def test_search_survives_an_exhausted_node_pool_and_uct_with_slots():
    """A node pool far too small for the simulations asked (the search treats a descent that cannot allocate as a
    collision and gives its visit back): no crash, every simulation accounted for, legal moves.  And the UCT mode with four
    leaves in flight (virtual loss 1): visits accounted for, support inside the legal set."""
    from hive_alphazero_amd import batch, mcts, playout
    G = 64
    boards = playout.random_positions(G, seed=12)
    B = batch.BoardBatch(G)
    B.import_state(boards)
    rb, rh = B.export_state()
    over, _ = B.terminal()
    mask, count, _ = B.legal()
    live = (over == 0) & (rb[:, 33] < 55)

    def flat_eval(planes):
        n = planes.shape[0]
        return torch.full((n, 1584), 1.0 / 1584, device="cuda"), torch.zeros((n,), device="cuda")

    ts = mcts.TreeSearch(G, 40, flat_eval, plane_dtype=torch.float32, seed=1, max_nodes=10)
    action, policy, sum_n = ts.search(rb, rh)
    torch.cuda.synchronize()
    hist = ts.leaf_histogram()
    assert bool((hist.sum(1)[live] == 40).all().item()) and int(ts.node_counts().max().item()) <= 10
    assert int(hist[:, 4].sum().item()) > 0                 # the pool really ran out
    ok, legal = _support_inside_legal(policy, mask)
    assert ok
    has = live & (count > 0)
    assert bool((legal[has].gather(1, action[has].long().view(-1, 1)) == 1).all().item())
    ts.close()
    tu = mcts.TreeSearch(G, 41, flat_eval, plane_dtype=torch.float32, slots=4, mode=mcts.UCT)
    action, policy, sum_n = tu.search(rb, rh)
    torch.cuda.synchronize()
    hist = tu.leaf_histogram()
    assert bool((sum_n[live] == 40 - hist[live][:, 4]).all().item())
    ok, _ = _support_inside_legal(policy, mask)
    assert ok
    visits, total_value, priors = tu.root_stats()
    assert bool((visits.sum(1)[live] == sum_n[live].float()).all().item())
    tu.close(); B.close()


def test_compact_games_widen_to_trainer_tensors_on_the_gpu():
    """records.dataset_tensors_gpu (hive_expand_launch on stored packed features) == records.dataset_from_games (host): planes in
    both layouts and three dtypes, policies, discounted values -- on games the GPU engine really played."""
    from hive_alphazero_amd import mcts, records

    def flat_eval(planes):
        n = planes.shape[0]
        return torch.full((n, 1584), 1.0 / 1584, device="cuda"), torch.zeros((n,), device="cuda")

    sp = mcts.SelfPlay(8, 3, flat_eval, seed=4, plane_dtype=torch.float32, game_ids=range(8))
    games = []
    while sp.running():
        sp.play_ply()
        games += sp.drain_finished()
    sp.close()
    assert len(games) == 8
    states, policies, values = records.dataset_from_games(games)
    for dtype in (torch.float32, torch.bfloat16, torch.float16):
        chw, pol, val = records.dataset_tensors_gpu(games, dtype=dtype, layout="chw")
        hwc, _, _ = records.dataset_tensors_gpu(games, dtype=dtype, layout="hwc")
        torch.cuda.synchronize()
        assert np.array_equal(hwc.float().cpu().numpy(), states)
        assert np.array_equal(chw.float().cpu().numpy(), states.transpose(0, 3, 1, 2))
        assert np.array_equal(pol.cpu().numpy(), policies) and np.allclose(val.cpu().numpy(), values, atol=1e-6)
    # the packed form (what the children send and the files hold) widens without a Python object per row: same tensors
    packed = records.pack_games(games)
    assert np.array_equal(records.packed_values(packed), values)
    for dtype, layout in ((torch.float32, "chw"), (torch.bfloat16, "hwc")):
        a = records.dataset_tensors_gpu(games, dtype=dtype, layout=layout)
        b = records.dataset_tensors_gpu_packed(packed, dtype=dtype, layout=layout)
        torch.cuda.synchronize()
        assert all(torch.equal(x, y) for x, y in zip(a, b)), (dtype, layout)
    empty = records.dataset_tensors_gpu_packed(records.pack_games([]))
    assert empty[0].shape == (0, 56, 12, 12) and empty[1].shape == (0, 1584) and empty[2].shape == (0,)


def _replay_game_through_the_oracle(value_white, plies, gid):
    """One recorded game (value_white, plies, game id) judged by the CPU oracle: -> (plies checked, passes).  See
    test_finished_selfplay_games_replay_through_the_oracle for what is asserted."""
    from oracle import oracle_py as O
    from hive_alphazero_amd import records
    plies_checked = passes = 0
    g = O.OracleGame()
    planes = [records.unpack_features(w, t, records.history_planes(hw, hl)) for (w, hw, hl, t, pol, mover) in plies]
    for k, (w, hw, hl, t, pol, mover) in enumerate(plies):
        assert g.turn == t and mover == (0 if t % 2 == 1 else 1), (gid, k)
        assert np.array_equal(planes[k], g.encode_board().astype(np.float64)), (gid, k)      # recorded == oracle planes
        legal = g.actions()
        support = np.flatnonzero(pol).tolist()
        assert set(support) <= set(legal), (gid, k)
        plies_checked += 1
        if not legal:
            assert pol.sum() == 0
            g.move(-1)
            passes += 1
            continue
        # (when max W < 0 the policy is the masked prior renormalised the reference's way, p / (sum p + 1e-8),
        # solo_play.py:304-313,372-373: with one legal move of prior 9e-5 that is 0.99989 -- seen in a 1024-game soak)
        assert abs(pol.sum() - 1.0) < 2e-3, (gid, k, t, float(pol.sum()), len(legal), np.flatnonzero(pol).size)
        found = None
        if k + 1 < len(plies):
            for a in legal:                    # (not only the policy's support: turns 1..6 resample with uniform noise)
                nxt = g.copy()
                nxt.move(a)
                if nxt.turn == plies[k + 1][3] and np.array_equal(nxt.encode_board().astype(np.float64), planes[k + 1]):
                    found = a
                    break
        else:                                  # the last ply: some legal move ends the game the way it was scored
            for a in legal:
                nxt = g.copy()
                nxt.move(a)
                over, w_ = nxt.game_is_over()
                vw = 1 if w_ == 1 else (-1 if w_ == 2 else 0)
                if (over and vw == value_white) or (not over and nxt.turn >= 55 and value_white == 0):
                    found = a
                    break
        assert found is not None, (gid, k, t)
        g.move(found)
    over, w_ = g.game_is_over()
    assert over or g.turn >= 55, gid
    assert value_white == (1 if w_ == 1 else (-1 if w_ == 2 else 0)) or (not over and value_white == 0), gid
    rows = records.rows_from_game((value_white, plies, gid))
    n_side = [sum(1 for p in plies if p[5] == s) for s in (0, 1)]
    seen = [0, 0]
    for (state, policy, value, lens), p in zip(rows, plies):
        side = p[5]
        seen[side] += 1
        assert lens == [n_side[side], seen[side]]
        assert value == (-1 if value_white == 0 else (value_white if side == 0 else -value_white))
    return plies_checked, passes


def test_finished_selfplay_games_replay_through_the_oracle(bf16_net):
    """The whole producer chain -- noisy search, move resampling on turns 1..6, env step, per-ply records, scoring -- for
    512 games played to their end (50 simulations per move, the default fp16 engine), judged by the CPU oracle instead of by the GPU's
    own legal mask: the moves are recovered from consecutive recorded planes on oracle_py.OracleGame (the action after
    which the next recorded planes appear), so every played move is in the ORACLE's legal set, every recorded plane
    tensor equals the oracle's encode_board of that position, a side with an empty recorded policy really had no move,
    the recorded result is the oracle's winner (draw / length cap -> 0 -> value -1 for both sides), and the rows'
    value and [game_len, counter] fields are consistent (woker/self_play.py:116-193)."""
    from oracle import oracle_py as O
    from hive_alphazero_amd import mcts, records
    # half of BASELINE configs[2] at its 50 simulations by default (512 games: the leaf batches run on the balanced 72-tile
    # tower with row lists); HIVE_SOAK_GAMES=1024 is the full size
    G, sims = int(os.environ.get("HIVE_SOAK_GAMES", "512")), int(os.environ.get("HIVE_SOAK_SIMS", "50"))
    slots = int(os.environ.get("HIVE_SOAK_SLOTS", "1"))             # leaves in flight per tree (virtual loss; BASELINE configs[4]: 4)
    if os.environ.get("HIVE_SOAK_DTYPE", "auto") != "bf16":         # the default engine (fp16 by the range probe) through the chain
        from hive_alphazero_amd.alpha_net import ChessNet, InferenceNet
        torch.manual_seed(0)
        bf16_net = InferenceNet(ChessNet().cuda().eval())
    sp = mcts.SelfPlay(G, sims, bf16_net, seed=21, keep_records=True, game_ids=range(G), max_finished_kept=2 * G, slots=slots)
    games = []
    for _ in range(60):
        sp.play_ply()
        games += sp.drain_finished()
        if sp.running() == 0:
            break
    assert sp.running() == 0 and len(games) == G and sp.dropped_games == 0 and sp.unrecorded_games == 0
    assert sp.env.illegal_count() == 0
    assert sorted(e[2] for e in games) == list(range(G))
    decisive = plies_checked = passes = 0
    for value_white, plies, gid in games:
        c, p_ = _replay_game_through_the_oracle(value_white, plies, gid)
        plies_checked, passes, decisive = plies_checked + c, passes + p_, decisive + int(value_white != 0)
    print(f"oracle replay: {G} games, {plies_checked} plies, {passes} passes, {decisive} decisive games")
    assert plies_checked > 40 * G
    sp.close()


def test_selfplay_worker_files_replay_through_the_oracle(tmp_path):
    """The PRODUCER end to end, judged by the oracle: SelfPlayWorker (spawned child, compact format) plays three rounds of
    games per slot -- finished slots are refilled from the id shard, games end on different plies -- with the default
    engine (two tower chains, unread rows skipped, equal leaves shared, packed record batches through the queue, files
    written on writer threads); every game is then READ BACK FROM THE FILES and replayed through oracle_py.OracleGame
    exactly as test_finished_selfplay_games_replay_through_the_oracle does, and the lazy `results` mapping must hold the
    same games.  HIVE_SOAK_WORKER_GAMES / HIVE_SOAK_SIMS scale it up (2048 x 50: ~1 min of play + the replay); HIVE_SOAK_REPLAY = how
    many of the games to replay (every total/REPLAY-th id) when the run is long, HIVE_SOAK_PER_GPU the engine's batch."""
    from hive_alphazero_amd import records
    from hive_alphazero_amd.self_play import SelfPlayWorker
    total, sims = int(os.environ.get("HIVE_SOAK_WORKER_GAMES", "48")), int(os.environ.get("HIVE_SOAK_SIMS", "6"))
    per_gpu = int(os.environ.get("HIVE_SOAK_PER_GPU", max(16, total // 3)))
    w = SelfPlayWorker(total_games=total, games_per_gpu=per_gpu, sims=sims, gpus=[0], seed=11, datapath=str(tmp_path),
                       games_per_file=max(7, total // 5), report_every=0, row_format="compact", log=lambda *_: None)
    res = w.start(timeout_s=1100)
    assert list(res) == list(range(total)) and w.leaf_kinds.get("root_evaluated", 0) > 0
    batches = [records.load_packed(f) for f in w.files]
    assert sorted(int(i) for b_ in batches for i in b_["game_id"]) == list(range(total))
    assert sum(len(b_["meta"]) for b_ in batches) == sum(w.game_lens)
    plies_checked = passes = decisive = replayed = 0
    lengths = set()
    stride = max(1, total // int(os.environ.get("HIVE_SOAK_REPLAY", total)))      # long soaks replay every stride-th game
    for b_ in batches:
        lengths.update(np.diff(b_["game_ptr"]).tolist())
        for g, gid in enumerate(b_["game_id"].tolist()):
            if gid % stride:
                continue
            value_white, plies, _ = records.unpack_game(b_, g)
            c, p_ = _replay_game_through_the_oracle(value_white, plies, gid)
            plies_checked, passes, decisive, replayed = plies_checked + c, passes + p_, decisive + int(value_white != 0), replayed + 1
            mine = res[gid]
            assert mine[0] == value_white and len(mine[1]) == len(plies)
            assert all(np.array_equal(np.asarray(u), np.asarray(v)) for x, y in zip(mine[1], plies) for u, v in zip(x, y))
    print(f"worker files -> oracle: {total} games in {len(w.files)} files ({sum(w.game_lens)} rows), {replayed} replayed: "
          f"{plies_checked} plies, {passes} passes, {decisive} decisive games; {len(lengths)} distinct game lengths")
    assert plies_checked > 40 * replayed and replayed >= total // stride


def test_config3_shards_of_8192_game_ids_rehearsed_on_one_gpu(bf16_net):
    """BASELINE configs[3] (8192 concurrent games sharded over 8 GPUs, 1024 each) rehearsed on ONE GPU: the eight ranks'
    shards of the global game ids 0..8191 (dist.game_id_stream) are disjoint and complete, and each shard is played one
    after another for two plies on engines of 1024 games; shard 5's games are then played again inside a differently
    shaped engine (slots 100..1123 of a 1280-game engine whose other slots hold other ids): same moves, same visit
    policies -- a game's record does not depend on the rank, the batch position or the batch size it ran in."""
    from hive_alphazero_amd import dist as hd
    from hive_alphazero_amd import mcts
    world, total, per = 8, 8192, 1024
    shards = [list(hd.game_id_stream(r, world, total)) for r in range(world)]
    assert all(len(s) == per for s in shards)
    assert sorted(i for s in shards for i in s) == list(range(total))
    assert all(s == list(range(r * per, (r + 1) * per)) for r, s in enumerate(shards))
    sims, plies = 50, 2
    played = {}
    for r in range(world):
        sp = mcts.SelfPlay(per, sims, bf16_net, seed=99, keep_records=False, game_ids=iter(shards[r]))
        assert sp.game_id.cpu().tolist() == shards[r]
        moves = []
        for _ in range(plies):
            sp.play_ply()
            moves.append((sp.search.action.cpu().clone(), sp.search.policy.cpu().clone()))
        assert sp.env.illegal_count() == 0
        played[r] = moves
        sp.close()
    # the same ids in another batch shape
    ids = list(range(90000, 90100)) + shards[5] + list(range(91000, 91156))
    sp = mcts.SelfPlay(1280, sims, bf16_net, seed=99, keep_records=False, game_ids=iter(ids))
    for k in range(plies):
        sp.play_ply()
        a, p = sp.search.action.cpu()[100:1124], sp.search.policy.cpu()[100:1124]
        assert torch.equal(a, played[5][k][0])
        assert float((p - played[5][k][1]).abs().max()) < 1e-6
    sp.close()
    # games of different shards differ (the noise is keyed on the game id, not on the slot)
    assert not torch.equal(played[0][1][0], played[1][1][0])
