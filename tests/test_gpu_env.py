"""GPU parity tests: the HIP env kernels (through the C ABI) against the golden vectors of
the true reference and against the CPU oracle on seeded random playouts.  Bit-exact."""
import gzip
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.fail("GPU test run without a visible GPU")


@pytest.fixture(scope="module")
def hv():
    _need_gpu()
    import hive_alphazero_amd as h
    from hive_alphazero_amd import batch, packing
    h.load()
    return h, batch, packing


def _mask_rows_to_lists(mask):
    """uint32[n][66] destination boards (hive_abi.h) -> ascending action-id lists."""
    from hive_alphazero_amd import packing
    m = mask.cpu().numpy().view(np.uint32)
    return [packing.mask_to_actions(r) for r in m]


def _mode_of(rec):
    return 1 if rec["t"] == 1 else (2 if rec["t"] == 2 else 0)


def test_device_tables(hv, golden_tables):
    h, batch, packing = hv
    import ctypes
    L = h.load()
    line = torch.zeros(144 * 6, dtype=torch.int32, device="cuda")
    nbr = torch.zeros(144 * 8, dtype=torch.uint8, device="cuda")
    assert L.hive_debug_tables(ctypes.c_void_p(line.data_ptr()), ctypes.c_void_p(nbr.data_ptr()), None) == 0
    torch.cuda.synchronize()
    line = line.cpu().numpy().view(np.uint32).reshape(144, 6)
    nbr = nbr.cpu().numpy().reshape(144, 8)
    for a in range(144):
        assert packing.words_to_cells(line[a]) == golden_tables["line"][a]
        assert nbr[a, :6].tolist() == golden_tables["nbr"][a]


def _golden_positions(games):
    recs = [rec for gm in games for rec in gm["plies"]]
    turn = np.array([r["t"] for r in recs])
    pos = np.array([r["pos"] for r in recs], dtype=np.uint8)
    lvl = np.array([r["lvl"] for r in recs], dtype=np.uint8)
    mode = np.array([_mode_of(r) for r in recs], dtype=np.uint8)
    return recs, turn, pos, lvl, mode


def test_movegen_golden_positions(hv, golden_games):
    """Config C2 shape: a flat batch of positions -> legal mask/count/list, bit-exact vs the reference."""
    h, batch, packing = hv
    recs, turn, pos, lvl, mode = _golden_positions(golden_games)
    boards = torch.from_numpy(packing.pack_boards(turn, pos, lvl, mode)).cuda()
    mask, count, lst = batch.movegen(boards, want_list=True)
    torch.cuda.synchronize()
    got = _mask_rows_to_lists(mask)
    cnt = count.cpu().numpy()
    lst = lst.cpu().numpy()
    bad = 0
    for i, r in enumerate(recs):
        if got[i] != r["legal"]:
            bad += 1
            if bad < 5:
                print("mismatch at", i, "turn", r["t"], "extra", sorted(set(got[i]) - set(r["legal"])),
                      "missing", sorted(set(r["legal"]) - set(got[i])))
        assert cnt[i] == len(r["legal"])
        assert lst[i, :cnt[i]].tolist() == r["legal"]
        assert np.all(lst[i, cnt[i]:] == -1)
    assert bad == 0, f"{bad} of {len(recs)} positions differ"


def test_movegen_golden_movegen_set(hv):
    path = os.path.join(GOLD, "games_movegen.json.gz")
    if not os.path.exists(path):
        pytest.skip("games_movegen.json.gz not generated")
    h, batch, packing = hv
    with gzip.open(path, "rt") as f:
        games = json.load(f)["games"]
    recs, turn, pos, lvl, mode = _golden_positions(games)
    boards = torch.from_numpy(packing.pack_boards(turn, pos, lvl, mode)).cuda()
    mask, count, _ = batch.movegen(boards)
    got = _mask_rows_to_lists(mask)
    bad = sum(1 for i, r in enumerate(recs) if got[i] != r["legal"])
    assert bad == 0, f"{bad} of {len(recs)} positions differ"


def test_replay_golden_games(hv, golden_games):
    """step + movegen + encode + terminal + history, ply by ply, against the reference's games."""
    h, batch, packing = hv
    games = golden_games
    n = len(games)
    B = batch.BoardBatch(n)
    maxlen = max(len(g["plies"]) for g in games)
    for ply in range(maxlen):
        mask, count, _ = B.legal()
        planes = B.encode(torch.float32, "hwc")
        over, winner = B.terminal()
        boards, _ = B.export_state()
        torch.cuda.synchronize()
        got = _mask_rows_to_lists(mask)
        planes = planes.cpu().numpy()
        over = over.cpu().numpy()
        winner = winner.cpu().numpy()
        st = packing.unpack_boards(boards.cpu().numpy())
        acts = np.full(n, -2, dtype=np.int32)
        for gi, gm in enumerate(games):
            if ply >= len(gm["plies"]):
                continue
            rec = gm["plies"][ply]
            ctx = f"seed {gm['seed']} ply {ply}"
            assert st["turn"][gi] == rec["t"], ctx
            assert st["pos"][gi].tolist() == rec["pos"], ctx
            assert [int(l) if p != 255 else 0 for p, l in zip(st["pos"][gi], st["lvl"][gi])] == rec["lvl"], ctx
            assert got[gi] == rec["legal"], ctx
            assert bool(over[gi]) == rec["over"], ctx
            if rec["over"]:
                assert int(winner[gi]) == rec["win"], ctx
            pl = planes[gi]
            assert np.all(pl[:, :, 31] == rec["t"]), ctx
            pl = pl.copy()
            pl[:, :, 31] = 0
            nz = np.argwhere(pl != 0)
            assert np.all(pl[pl != 0] == 1), ctx
            gotp = sorted(int((x * 12 + y) * 56 + p) for x, y, p in nz)
            if gotp != rec["planes"]:
                diff = sorted(set(gotp) ^ set(rec["planes"]))
                raise AssertionError(f"{ctx}: planes differ at (cell,plane) {[(d // 56, d % 56) for d in diff]}")
            if rec["a"] is not None:
                acts[gi] = rec["a"]
        B.step(acts, sync=True)
    B.close()


def test_encode_variants_agree(hv, golden_games):
    h, batch, packing = hv
    games = golden_games[:16]
    B = batch.BoardBatch(len(games))
    for ply in range(12):
        acts = np.array([g["plies"][ply]["a"] for g in games], dtype=np.int32)
        B.step(acts)
    ref = B.encode(torch.float32, "hwc").cpu()
    for dt in (torch.float32, torch.float16, torch.bfloat16):
        hwc = B.encode(dt, "hwc").float().cpu()
        chw = B.encode(dt, "chw").float().cpu()
        assert torch.equal(hwc, ref)
        assert torch.equal(chw, ref.permute(0, 3, 1, 2))
    B.close()


def test_illegal_action_is_refused(hv):
    h, batch, packing = hv
    from hive_alphazero_amd import HiveError
    B = batch.BoardBatch(3)
    before, _ = B.export_state()
    with pytest.raises(HiveError) as ei:
        B.step(np.array([858, 0, 859], dtype=np.int32))     # board 1: cell 0 is not the start tile
    assert ei.value.code == -3
    after, _ = B.export_state()
    st0, st1 = packing.unpack_boards(before.cpu().numpy()), packing.unpack_boards(after.cpu().numpy())
    assert st1["turn"].tolist() == [2, 1, 2]
    assert st1["pos"][1].tolist() == st0["pos"][1].tolist()
    assert B.illegal_count() == 1
    B.close()


def test_stateless_step_counts_refused_moves(hv):
    """hive_step_launch_counted: the stateless form of GamePlay.move with the batch form's contract (include/hive_abi.h):
    an action outside its board's legal mask leaves the board unchanged and is counted; hive_step_launch refuses it
    silently; the fused movegen launch (mask + count + sorted ids in one kernel) equals the mask-only launch + the
    separate list kernel of the batch handle; a misaligned id list is refused."""
    h, batch, packing = hv
    import ctypes
    from hive_alphazero_amd import playout
    from hive_alphazero_amd._lib import HIVE_MASK_WORDS
    L = h.load()
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    n = 300                                                        # not a multiple of the 16-board workgroup
    boards = playout.random_positions(n, seed=77)
    hist = torch.zeros((n, 384), dtype=torch.uint8, device="cuda")
    mask = torch.zeros((n, HIVE_MASK_WORDS), dtype=torch.int32, device="cuda")
    count = torch.zeros((n,), dtype=torch.int32, device="cuda")
    lst = torch.full((n, 256), 7, dtype=torch.int16, device="cuda")
    assert L.hive_movegen_launch(P(boards), n, P(mask), P(count), P(lst), None) == 0          # one fused launch
    mask2, count2 = torch.zeros_like(mask), torch.zeros_like(count)
    assert L.hive_movegen_launch(P(boards), n, P(mask2), P(count2), None, None) == 0          # mask-only launch
    torch.cuda.synchronize()
    assert torch.equal(mask, mask2) and torch.equal(count, count2)
    B = batch.BoardBatch(n)
    B.import_state(boards)
    m3, c3, l3 = B.legal(want_list=True)
    assert torch.equal(l3, lst) and torch.equal(c3, count)
    rows = _mask_rows_to_lists(mask)
    cnt = count.cpu().numpy()
    ls = lst.cpu().numpy()
    for i in range(n):
        assert ls[i, :cnt[i]].tolist() == rows[i] and np.all(ls[i, cnt[i]:] == -1)
    odd = lst.view(-1)[1:1 + 256 * 4]                              # 2-byte aligned only
    assert L.hive_movegen_launch(P(boards), 4, P(mask), P(count), ctypes.c_void_p(odd.data_ptr()), None) == -1
    assert b"8-byte" in L.hive_last_error()
    # every third board gets an action that is not in its legal set
    legal_first = torch.where(count > 0, lst[:, 0].to(torch.int32), torch.full_like(count, -1))
    acts = legal_first.clone()
    bad_rows = []
    for i in range(0, n, 3):
        if cnt[i] > 0:
            illegal = next(a for a in range(1584) if a not in set(rows[i]))
            acts[i] = illegal
            bad_rows.append(i)
    counter = torch.zeros((1,), dtype=torch.int64, device="cuda")
    b1, b2 = boards.clone(), boards.clone()
    assert L.hive_step_launch_counted(P(b1), P(hist.clone()), n, P(acts), P(mask), P(counter), None) == 0
    assert L.hive_step_launch(P(b2), P(hist.clone()), n, P(acts), P(mask), None) == 0
    torch.cuda.synchronize()
    assert int(counter.item()) == len(bad_rows) > 50
    assert torch.equal(b1, b2)                                     # same state either way: refused boards unchanged
    assert torch.equal(b1[bad_rows], boards[bad_rows])
    moved = [i for i in range(n) if i not in set(bad_rows) and cnt[i] > 0]
    assert bool((b1[moved, 33] == boards[moved, 33] + 1).all())    # the others advanced one turn
    assert L.hive_step_launch_counted(P(b1), None, n, P(acts), None, P(counter), None) == -1    # a counter needs a mask
    B.close()


def _oracle_corpus(n, seed):
    """n positions sampled from oracle random playouts (seeded)."""
    from oracle import oracle_py as O
    rng = np.random.default_rng(seed)
    turn, pos, lvl, mode, legal = [], [], [], [], []
    while len(turn) < n:
        g = O.OracleGame()
        while True:
            acts = g.actions()
            p, l = g.pieces()
            turn.append(g.turn); pos.append(p); lvl.append(l); mode.append(g.nmt_mode); legal.append(acts)
            over, _ = g.game_is_over()
            if over or g.turn >= 55 or len(turn) >= n:
                break
            g.move(int(acts[rng.integers(len(acts))]) if acts else -1)
    return np.array(turn), np.array(pos), np.array(lvl), np.array(mode, dtype=np.uint8), legal


def test_movegen_4096_vs_oracle(hv):
    """BASELINE config 2 at full size: 4096 boards, bit-exact against the CPU oracle."""
    h, batch, packing = hv
    turn, pos, lvl, mode, legal = _oracle_corpus(4096, seed=1234)
    boards = torch.from_numpy(packing.pack_boards(turn, pos, lvl, mode)).cuda()
    mask, count, lst = batch.movegen(boards, want_list=True)
    got = _mask_rows_to_lists(mask)
    assert got == legal
    assert count.cpu().numpy().tolist() == [len(x) for x in legal]
    # size-independent properties: the list is sorted, strictly increasing and agrees with the mask
    lst = lst.cpu().numpy()
    for i in range(0, 4096, 97):
        k = len(legal[i])
        assert lst[i, :k].tolist() == legal[i]
    # ragged batch sizes (tail workgroup handling)
    for n in (1, 63, 65, 127):
        m2, c2, _ = batch.movegen(boards[:n].contiguous())
        assert _mask_rows_to_lists(m2) == legal[:n]


def test_movegen_pair_layout_equals_quad_layout_and_oracle(hv, golden_games):
    """Launches of >= 16,384 boards run the kernel's pair layout (one board = two lanes, 32 boards per
    workgroup: hive_bb.hpp).  Forced on at every size here: the reference's golden positions, the 4096-board oracle
    corpus, ragged tails (1 .. 127 boards: a workgroup's 32 slots partly empty), and the default switch at 16,384."""
    h, batch, packing = hv
    L = h.load()
    prev = L.hive_movegen_pair_threshold(1)
    try:
        assert L.hive_movegen_pair_threshold(-1) == 1
        recs, turn, pos, lvl, mode = _golden_positions(golden_games)
        boards = torch.from_numpy(packing.pack_boards(turn, pos, lvl, mode)).cuda()
        mask, count, _ = batch.movegen(boards)
        got = _mask_rows_to_lists(mask)
        assert [g for g in got] == [r["legal"] for r in recs]
        assert count.cpu().numpy().tolist() == [len(r["legal"]) for r in recs]

        turn, pos, lvl, mode, legal = _oracle_corpus(4096, seed=1234)
        boards = torch.from_numpy(packing.pack_boards(turn, pos, lvl, mode)).cuda()
        mask, count, _ = batch.movegen(boards)
        assert _mask_rows_to_lists(mask) == legal
        assert count.cpu().numpy().tolist() == [len(x) for x in legal]
        for n in (1, 2, 31, 33, 63, 65, 127):
            for want_list in (False, True):
                m2, c2, l2 = batch.movegen(boards[:n].contiguous(), want_list=want_list)
                assert _mask_rows_to_lists(m2) == legal[:n]
                assert c2.cpu().numpy().tolist() == [len(x) for x in legal[:n]]
                if want_list:
                    l2 = l2.cpu().numpy()
                    for i in range(n):
                        assert l2[i, :len(legal[i])].tolist() == legal[i] and np.all(l2[i, len(legal[i]):] == -1)
        # the fused id list in pairs (32 boards' lists shared out over the 11 waves)
        m3, c3, l3 = batch.movegen(boards, want_list=True)
        assert torch.equal(m3, mask) and torch.equal(c3, count)
        l3 = l3.cpu().numpy()
        for i in range(4096):
            assert l3[i, :len(legal[i])].tolist() == legal[i] and np.all(l3[i, len(legal[i]):] == -1)
        L.hive_movegen_pair_threshold(1 << 30)
        mq, cq, lq = batch.movegen(boards, want_list=True)
        assert torch.equal(mq, mask) and torch.equal(cq, count) and np.array_equal(lq.cpu().numpy(), l3)
    finally:
        L.hive_movegen_pair_threshold(0)
    assert L.hive_movegen_pair_threshold(-1) == 16384 and prev == 16384
    # the default switch: 16,384 boards go through pairs, 16,383 through quads; same bits either way
    from hive_alphazero_amd import playout
    base = playout.random_positions(4096, seed=78)
    big = base.repeat(4, 1).contiguous()
    m0, c0, _ = batch.movegen(base)
    mp, cp, lp = batch.movegen(big, want_list=True)
    mq, cq, lq = batch.movegen(big[:16383].contiguous(), want_list=True)
    assert torch.equal(mp, m0.repeat(4, 1)) and torch.equal(cp, c0.repeat(4))
    assert torch.equal(mq, mp[:16383]) and torch.equal(cq, cp[:16383]) and torch.equal(lq, lp[:16383])


def test_random_playout_vs_oracle_lockstep(hv):
    """256 games stepped on the GPU and in the oracle with the same actions; every ply compared
    (legal sets, planes, terminal flags), including passes and finished games."""
    h, batch, packing = hv
    from oracle import oracle_py as O
    n = 256 * (8 if os.environ.get("HIVE_TEST_HEAVY") else 1)
    rng = np.random.default_rng(7)
    B = batch.BoardBatch(n)
    games = [O.OracleGame() for _ in range(n)]
    done = np.zeros(n, dtype=bool)
    for ply in range(56):
        mask, count, _ = B.legal()
        planes = B.encode(torch.bfloat16, "hwc").float().cpu().numpy()
        over, winner = B.terminal()
        got = _mask_rows_to_lists(mask)
        over = over.cpu().numpy(); winner = winner.cpu().numpy()
        acts = np.full(n, -2, dtype=np.int32)
        for i, g in enumerate(games):
            if done[i]:
                continue
            want = g.actions()
            assert got[i] == want, (i, ply)
            o, w = g.game_is_over()
            assert bool(over[i]) == o and (not o or int(winner[i]) == w), (i, ply)
            assert np.array_equal(planes[i], g.encode_board()), (i, ply)
            if o or g.turn >= 55:
                done[i] = True
                continue
            a = int(want[rng.integers(len(want))]) if want else -1
            if ply > 2 and rng.random() < 0.04:
                a = -1          # a forced pass / skip_turn (env_hive.py:100-103,493-496): history must not advance
            acts[i] = a
            g.move(a)
        if done.all():
            break
        B.step(acts, sync=True)
    assert done.all()
    B.close()


def test_size_independent_properties_large_batch(hv):
    """65,536 boards (16x the bench batch): count == popcount(mask), sorted strictly increasing lists that
    agree with the mask, idempotence, and permutation invariance of the batch."""
    h, batch, packing = hv
    from hive_alphazero_amd import playout
    base = playout.random_positions(4096, seed=77)
    perm = torch.randperm(65536, device="cuda", generator=torch.Generator(device="cuda").manual_seed(3))
    big = base.repeat(16, 1)[perm].contiguous()
    mask, count, lst = batch.movegen(big, want_list=True)
    mask2, count2, _ = batch.movegen(big)
    assert torch.equal(mask, mask2) and torch.equal(count, count2)                     # idempotent
    m0, c0, _ = batch.movegen(base)
    assert torch.equal(mask, m0.repeat(16, 1)[perm]) and torch.equal(count, c0.repeat(16)[perm])   # permutation invariant
    bits = mask.view(torch.uint8)
    pop = torch.zeros_like(count)
    for sh in range(8):
        pop += ((bits >> sh) & 1).sum(dim=1).to(torch.int32)
    assert torch.equal(pop, count)
    l = lst.to(torch.int32)
    valid = torch.arange(256, device="cuda").view(1, -1) < count.view(-1, 1)
    assert bool(((l >= 0) == valid).all())
    inc = (l[:, 1:] > l[:, :-1]) | ~valid[:, 1:]
    assert bool(inc.all())
    idx = torch.where(valid, l, torch.zeros_like(l)).long()
    cell, slot = idx // 11, idx % 11                                                   # HIVE_MASK_TEST (hive_abi.h)
    row, col = cell // 12, cell % 12
    word = mask.gather(1, slot * 6 + (row >> 1))
    assert bool(((((word >> (((row & 1) << 4) | col)) & 1) == 1) | ~valid).all())


def test_abi_argument_errors(hv):
    h, batch, packing = hv
    import ctypes
    L = h.load()
    hd = ctypes.c_void_p()
    assert L.hive_batch_create(0, 0, ctypes.byref(hd)) == -1                 # empty batch
    assert L.hive_batch_create(4, 99, ctypes.byref(hd)) == -1                # bad device
    assert L.hive_movegen_launch(None, 4, None, None, None, None) == -1      # NULL boards
    b = torch.zeros((4, 64), dtype=torch.uint8, device="cuda")
    lst = torch.zeros((4, 256), dtype=torch.int16, device="cuda")
    assert L.hive_movegen_launch(ctypes.c_void_p(b.data_ptr()), 4, None, None, ctypes.c_void_p(lst.data_ptr()), None) == -1
    assert b"mask" in L.hive_last_error()
    B = batch.BoardBatch(2)
    pl = torch.zeros((2, 12, 12, 56), dtype=torch.float32, device="cuda")
    assert L.hive_batch_encode(B._h, ctypes.c_void_p(pl.data_ptr()), 7, 0) == -1    # unknown dtype
    # a fresh game: five placements on the start tile, planes all-zero except the turn plane
    mask, count, _ = B.legal()
    assert count.tolist() == [5, 5]
    pl = B.encode(torch.float32, "hwc")
    assert float(pl[..., 31].min()) == 1.0 and float(pl.sum()) == 2 * 144.0
    B.close()
    # the leaf-batch entry points of round 3 refuse what they cannot do instead of faulting
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    y = torch.zeros((4, 144, 256), dtype=torch.bfloat16, device="cuda")
    rep = torch.arange(4, dtype=torch.int32, device="cuda")
    assert L.hive_nn_copy_rows(P(y), P(rep), 4, 144 * 256 * 2, None) == 0
    assert L.hive_nn_copy_rows(P(y), P(rep), 4, 10, None) == -1                 # rows must be a multiple of 16 bytes
    assert L.hive_nn_copy_rows(P(y), None, 4, 144 * 256 * 2, None) == -1
    assert L.hive_nn_copy_rows(ctypes.c_void_p(y.data_ptr() + 2), P(rep), 3, 144 * 256 * 2, None) == -1   # misaligned rows
    w = torch.zeros((9 * 8 * 16 * 64 * 8,), dtype=torch.bfloat16, device="cuda")
    bias = torch.zeros((256,), dtype=torch.float32, device="cuda")
    need = torch.zeros((4,), dtype=torch.int8, device="cuda")
    assert L.hive_nn_resblock_sel(P(y), P(w), P(bias), P(w), P(bias), P(y), 4, 2, P(need), None) == -1      # y aliases x
    y2 = torch.full_like(y, 3.0)
    assert L.hive_nn_resblock_sel(P(y), P(w), P(bias), P(w), P(bias), P(y2), 4, 2, P(need), None) == 0
    torch.cuda.synchronize()
    assert float(y2.float().min()) == 3.0                                          # no board selected: nothing written
    assert L.hive_nn_resblock_sel(P(y), P(w), P(bias), P(w), P(bias), P(y2), 4, 9, P(need), None) == -1     # unknown dtype


def test_planes_writer_streaming_launch_equals_small_launches(hv):
    """hive_expand_kernel switches to streaming (nontemporal) stores from 8192 boards per launch on: a 8192 + 37 board encode
    (f32 / bf16, HWC / CHW, history on) equals the same boards encoded in launches of 1000 (the plain-store instantiation)."""
    h, batch, packing = hv
    import ctypes
    from hive_alphazero_amd import playout
    from hive_alphazero_amd._lib import BF16, F32, HWC, CHW
    L = h.load()
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    n = 8192 + 37
    base = playout.random_positions(4096, seed=9)
    boards = torch.cat([base, base, base[:37]]).contiguous()
    g = torch.Generator(device="cuda").manual_seed(5)
    hist = torch.randint(0, 256, (n, 384), dtype=torch.uint8, device="cuda", generator=g)
    hist[:, 0::4] &= 0xFF; hist[:, 1::4] &= 0x0F; hist[:, 3::4] &= 0x0F            # 12-bit rows in 16-bit fields
    boards[:, 35] = torch.randint(0, 5, (n,), dtype=torch.uint8, device="cuda", generator=g) * 17      # history lengths 0..4 / 0..4
    ws = torch.empty((n * 144,), dtype=torch.int64, device="cuda")
    for dt, tdt in ((BF16, torch.bfloat16), (F32, torch.float32)):
        for ly, shape in ((HWC, (n, 12, 12, 56)), (CHW, (n, 56, 12, 12))):
            big = torch.full(shape, float("nan"), dtype=tdt, device="cuda")
            assert L.hive_encode_launch(P(boards), P(hist), n, P(big), dt, ly, P(ws), None) == 0
            small = torch.full(shape, float("nan"), dtype=tdt, device="cuda")
            for lo in range(0, n, 1000):
                k = min(1000, n - lo)
                assert L.hive_expand_launch(P(boards[lo:]), P(hist[lo:]), P(ws[lo * 144:]), k, P(small[lo:]), dt, ly, None) == 0
            torch.cuda.synchronize()
            assert torch.equal(big.view(torch.int16 if tdt == torch.bfloat16 else torch.int32),
                               small.view(torch.int16 if tdt == torch.bfloat16 else torch.int32)), (dt, ly)
            assert bool(torch.isfinite(big.float()).all())


@pytest.mark.gpu
def test_leaf_dedup_picks_the_first_equal_row():
    """hive_leaf_dedup_launch against a dictionary on the host: random 448-byte rows with planted copies (also copies that
    differ in ONE byte of the board or of the history, and copies of rows that are not needed), random need flags, several
    batch sizes up to the 4096-row limit.  A needed row equal to an earlier needed row must be switched off and point at the
    FIRST such row; everything else keeps need and points at itself; the counter loses exactly the rows switched off."""
    assert torch.cuda.is_available()
    import ctypes
    from hive_alphazero_amd import _lib
    L = _lib.load()
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    rng = np.random.default_rng(8)
    for n in (1, 5, 1024, 3000, 4096):
        distinct = max(1, n // 7)
        src = rng.integers(0, distinct, n)
        base_b = rng.integers(0, 256, (distinct, 64), dtype=np.uint8)
        base_h = rng.integers(0, 256, (distinct, 384), dtype=np.uint8)
        boards, hist = base_b[src].copy(), base_h[src].copy()
        for i in rng.choice(n, n // 10, replace=False):          # near copies: one byte off
            if i % 2:
                boards[i, rng.integers(0, 64)] ^= 1 << int(rng.integers(0, 8))
            else:
                hist[i, rng.integers(0, 384)] ^= 1 << int(rng.integers(0, 8))
        need = (rng.random(n) < 0.8).astype(np.int8)
        first, want_rep, want_need = {}, np.arange(n, dtype=np.int32), need.copy()
        for i in range(n):
            if not need[i]:
                continue
            key = boards[i].tobytes() + hist[i].tobytes()
            if key in first:
                want_rep[i], want_need[i] = first[key], 0
            else:
                first[key] = i
        tb, th = torch.from_numpy(boards).cuda(), torch.from_numpy(hist).cuda()
        tn = torch.from_numpy(need).cuda()
        rep = torch.full((n,), -7, dtype=torch.int32, device="cuda")
        keys = torch.zeros((n,), dtype=torch.int64, device="cuda")
        total = torch.tensor([int(need.sum())], dtype=torch.int64, device="cuda")
        _lib.check(L.hive_leaf_dedup_launch(P(tb), P(th), n, P(tn), P(rep), P(keys), P(total), None))
        torch.cuda.synchronize()
        assert np.array_equal(rep.cpu().numpy(), want_rep), n
        assert np.array_equal(tn.cpu().numpy(), want_need), n
        assert int(total.item()) == int(want_need.sum())
    assert L.hive_leaf_dedup_launch(P(tb), P(th), 4097, P(tn), P(rep), P(keys), None, None) != 0      # over the row limit
    assert L.hive_leaf_dedup_launch(P(tb), P(th), 8, None, P(rep), P(keys), None, None) != 0
