"""TreeSearch / SelfPlay: thousands of PUCT searches in lock step on one MI355X.

Python only sequences kernel launches; the tree (flat SoA node pool), selection, expansion and
backup are HIP kernels behind include/hive_search.h, leaf positions are expanded and encoded by
the env kernels (hive_leaf_launch), and the only MFMA work is the batched network forward
(alpha_net.InferenceNet) on [games x slots] leaves per simulation.
"""
import ctypes

import torch

from . import _lib
from ._lib import BF16, F16, F32, HIVE_MASK_WORDS, HWC, check, load
from .config import MAX_GAME_LENGTH

_DT = {torch.float32: F32, torch.float16: F16, torch.bfloat16: BF16}


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


PUCT, UCT = 0, 1           # HiveSearchMode (include/hive_search.h)
LEAF_KINDS = ("unused", "root_evaluated", "expanded_evaluated", "known_finished_game", "collision", "length_cap",
              "new_finished_or_capped", "already_created_in_flight")


class _Params(ctypes.Structure):
    _fields_ = [("c_puct", ctypes.c_float), ("noise_eps", ctypes.c_float), ("dirichlet_alpha", ctypes.c_float),
                ("max_game_length", ctypes.c_int32), ("mode", ctypes.c_int32), ("virtual_loss", ctypes.c_float)]


class TreeSearch:
    """`games` independent searches of `sims` simulations each (HivePlayer.action, solo_play.py:110-165).

    evaluator(planes[B,12,12,56]) -> (p fp32 [B,1584] softmax, v fp32 [B])."""

    def __init__(self, games, sims, evaluator, device=None, slots=1, seed=0, plane_dtype=None,
                 c_puct=0.7, noise_eps=0.25, dirichlet_alpha=0.3, max_nodes=None, transpositions=True, mode=PUCT,
                 virtual_loss=None, max_game_length=None, skip_unread_rows=True, share_equal_leaves=True, reuse_store=0):
        """mode = PUCT: woker/solo_play.py::HivePlayer; mode = UCT: alpha_zero/MCTS_chess.py::UCT_search (plain tree, no
        noise, no length cap; with one slot the virtual loss is 0 so that W sums exactly like the sequential reference)."""
        L = load()
        if L.hive_device_count() <= 0 or not torch.cuda.is_available():
            raise _lib.HiveError(-2, "no HIP device visible: hive_alphazero_amd has no CPU path")
        if plane_dtype is None:      # the evaluator's own input type (the plane values 0 / 1 / turn are exact in all three)
            plane_dtype = getattr(evaluator, "dtype", torch.bfloat16)
            if plane_dtype not in _DT:
                plane_dtype = torch.bfloat16
        self.L = L
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        self.games, self.sims, self.slots, self.evaluator = games, sims, slots, evaluator
        self.max_nodes = max_nodes or (sims + slots + 2)
        self._h = ctypes.c_void_p()
        check(L.hive_search_create(games, self.max_nodes, slots, self.device.index, seed, ctypes.byref(self._h)))
        self.mode = mode
        if mode == UCT:
            noise_eps, transpositions = 0.0, False
            if virtual_loss is None:
                virtual_loss = 0.0 if slots == 1 else 1.0
            if max_game_length is None:
                max_game_length = 250
        prm = _Params(c_puct, noise_eps, dirichlet_alpha, MAX_GAME_LENGTH if max_game_length is None else max_game_length,
                      mode, 1.0 if virtual_loss is None else virtual_loss)
        check(L.hive_search_set_params(self._h, ctypes.byref(prm)))
        # the reference's tree is a dict keyed by state_key: positions reached by two move orders share one entry
        check(L.hive_search_set_transpositions(self._h, 1 if transpositions else 0))
        n = games * slots
        dev = self.device
        self.plane_dtype = plane_dtype
        self.leaf_boards = torch.zeros((n, 64), dtype=torch.uint8, device=dev)
        self.leaf_hist = torch.zeros((n, 384), dtype=torch.uint8, device=dev)
        self.leaf_mask = torch.zeros((n, HIVE_MASK_WORDS), dtype=torch.int32, device=dev)
        self.leaf_count = torch.zeros((n,), dtype=torch.int32, device=dev)
        self.leaf_over = torch.zeros((n,), dtype=torch.int8, device=dev)
        self.leaf_winner = torch.zeros((n,), dtype=torch.int8, device=dev)
        self.planes = torch.zeros((n, 12, 12, 56), dtype=plane_dtype, device=dev)
        self.workspace = torch.zeros((n * 144,), dtype=torch.int64, device=dev)
        self.policy = torch.zeros((games, 1584), dtype=torch.float32, device=dev)
        self.action = torch.zeros((games,), dtype=torch.int32, device=dev)
        self.sum_n = torch.zeros((games,), dtype=torch.int32, device=dev)
        self.root_planes = None
        # rows of the leaf batch whose prediction the backup never reads (finished games, length cap, collisions, idle
        # trees: the reference does not call its model for them either, solo_play.py:169-197) are skipped by an evaluator
        # that can (InferenceNet: its tower kernels take the flags); evals_run counts the rows that were evaluated
        self.skip_unread_rows = bool(skip_unread_rows) and bool(getattr(evaluator, "accepts_need", False))
        self.leaf_need = torch.ones((n,), dtype=torch.int8, device=dev)
        self.evals_run = torch.zeros((1,), dtype=torch.int64, device=dev)
        self.evals_launched = 0
        # equal leaves of one batch (same board record and history = same planes) are evaluated once: lock-step games from
        # the opening ask about the same few positions (hive_leaf_dedup_launch; batches of up to 4096 rows)
        self.share_equal_leaves = (self.skip_unread_rows and bool(share_equal_leaves) and n <= 4096
                                   and bool(getattr(evaluator, "accepts_rep", False)))
        if self.skip_unread_rows and share_equal_leaves and n > 4096 and getattr(evaluator, "accepts_rep", False):
            import warnings
            warnings.warn(f"TreeSearch: leaf batches of {n} rows exceed hive_leaf_dedup_launch's 4096-row table; equal leaves "
                          "are evaluated separately (share_equal_leaves off)")
        self.leaf_rep = torch.arange(n, dtype=torch.int32, device=dev)
        self.leaf_keys = torch.zeros((n,), dtype=torch.int64, device=dev)
        # evaluations kept ACROSS searches (hive_leaf_store_*, include/hive_abi.h): the reference empties its tree on every
        # move (solo_play.py:103-112) and evaluates the subtree under the played move again; reuse_store = entries (0 = off,
        # the default).  Needs the equal-leaf keys; the served rows are counted apart from the evaluated ones (rows_served).
        self._store = None
        self._store_version = getattr(evaluator, "weights_version", 0)
        if reuse_store and self.share_equal_leaves and getattr(evaluator, "batch_independent_bits", False):
            self._store = ctypes.c_void_p()
            check(L.hive_leaf_store_create(self.device.index, int(reuse_store), ctypes.byref(self._store)))
            self.leaf_hit = torch.full((n,), -1, dtype=torch.int32, device=dev)

    def close(self):
        if getattr(self, "_h", None):
            self.L.hive_search_destroy(self._h)
            self._h = None
        if getattr(self, "_store", None):
            self.L.hive_leaf_store_destroy(self._store)
            self._store = None

    def rows_served(self):
        """(rows answered from the store, rows inserted into it) since it was created or cleared."""
        if self._store is None:
            return 0, 0
        a, b = ctypes.c_uint64(), ctypes.c_uint64()
        check(self.L.hive_leaf_store_stats(self._store, ctypes.byref(a), ctypes.byref(b)))
        return int(a.value), int(b.value)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _stream(self):
        s = ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        check(self.L.hive_search_set_stream(self._h, s))
        return s

    def search(self, root_boards, root_hist, active=None, selfplay=False, keep_root_planes=False):
        """-> (action int32[games] (-1 pass, -2 inactive/terminal root), policy fp32[games,1584], sum_n int32[games])"""
        L, G = self.L, self.games
        s = self._stream()
        check(L.hive_search_set_roots(self._h, _p(root_boards), _p(root_hist), _p(active)))
        done = 0
        first = True
        while done < self.sims:
            k = 1 if first else min(self.slots, self.sims - done)      # the first simulation only opens the root
            for sl in range(k):
                check(L.hive_search_select(self._h, sl, _p(self.leaf_boards[sl * G:]), _p(self.leaf_hist[sl * G:])))
            n = k * G
            check(L.hive_leaf_launch(_p(self.leaf_boards), _p(self.leaf_hist), n, _p(self.planes), _DT[self.plane_dtype],
                                     HWC, _p(self.workspace), _p(self.leaf_mask), _p(self.leaf_count),
                                     _p(self.leaf_over), _p(self.leaf_winner), s))
            if first and keep_root_planes:
                self.root_planes = self.workspace[:G * 144].clone()
            if self.skip_unread_rows:
                check(L.hive_search_leaf_need(self._h, k, _p(self.leaf_boards), _p(self.leaf_over), _p(self.leaf_need),
                                              _p(self.evals_run)))
                if self.share_equal_leaves:
                    check(L.hive_leaf_dedup_launch(_p(self.leaf_boards), _p(self.leaf_hist), n, _p(self.leaf_need),
                                                   _p(self.leaf_rep), _p(self.leaf_keys), _p(self.evals_run), s))
                    if self._store is not None:
                        if getattr(self.evaluator, "weights_version", 0) != self._store_version:      # new weights: old answers are void
                            check(L.hive_leaf_store_clear(self._store, s))
                            self._store_version = getattr(self.evaluator, "weights_version", 0)
                        check(L.hive_leaf_store_lookup(self._store, _p(self.leaf_boards), _p(self.leaf_hist), n, _p(self.leaf_keys),
                                                       _p(self.leaf_need), _p(self.leaf_hit), _p(self.evals_run), s))
                    p, v = self.evaluator(self.planes[:n], need=self.leaf_need[:n], rep=self.leaf_rep[:n])
                    if self._store is not None:
                        p = p.float().contiguous()
                        v = v.float().contiguous().view(-1)
                        check(L.hive_leaf_store_update(self._store, _p(self.leaf_boards), _p(self.leaf_hist), n, _p(self.leaf_keys),
                                                       _p(self.leaf_need), _p(self.leaf_rep), _p(self.leaf_hit), _p(p), _p(v), s))
                else:
                    p, v = self.evaluator(self.planes[:n], need=self.leaf_need[:n])
            else:
                p, v = self.evaluator(self.planes[:n])
            self.evals_launched += n
            p = p.float().contiguous()
            v = v.float().contiguous().view(-1)
            for sl in range(k):
                o = sl * G
                check(L.hive_search_backup(self._h, sl, _p(self.leaf_boards[o:]), _p(self.leaf_hist[o:]),
                                           _p(self.leaf_mask[o:]), _p(self.leaf_over[o:]), _p(self.leaf_winner[o:]),
                                           _p(p[o:]), _p(v[o:])))
            done += k
            first = False
        check(L.hive_search_policy(self._h, _p(self.policy), _p(self.action), _p(self.sum_n), 1 if selfplay else 0))
        return self.action, self.policy, self.sum_n

    def set_game_ids(self, ids):
        """Global game index per tree (int64[games]): the only per-game key of the noise streams (hive_search.h)."""
        ids = torch.as_tensor(ids, dtype=torch.int64, device=self.device).contiguous()
        assert ids.numel() == self.games
        self._stream()
        check(self.L.hive_search_set_game_ids(self._h, _p(ids)))

    def root_stats(self):
        """(visits, total_value, priors) fp32[games,1584]: the root's UCTNode arrays (MCTS_chess.py:33-35)."""
        out = [torch.empty((self.games, 1584), dtype=torch.float32, device=self.device) for _ in range(3)]
        self._stream()
        check(self.L.hive_search_root_stats(self._h, _p(out[0]), _p(out[1]), _p(out[2])))
        return out

    def leaf_histogram(self):
        """int32[games,8] leaves by kind since create (LEAF_KINDS)."""
        out = torch.zeros((self.games, 8), dtype=torch.int32, device=self.device)
        self._stream()
        check(self.L.hive_search_leaf_histogram(self._h, _p(out)))
        return out

    def node_counts(self):
        out = torch.zeros((self.games,), dtype=torch.int32, device=self.device)
        self._stream()
        check(self.L.hive_search_node_counts(self._h, _p(out)))
        return out

    def transposition_hits(self):
        """int32[games]: descents that continued through a node first reached by another move order."""
        out = torch.zeros((self.games,), dtype=torch.int32, device=self.device)
        self._stream()
        check(self.L.hive_search_transposition_hits(self._h, _p(out)))
        return out


class SelfPlay:
    """`games` self-play games advanced ply by ply (woker/self_play.py:116-193 for every game at once):
    search -> move selection (self-play noise on turns 1..6) -> env step; finished games (queen
    surrounded or turn >= 55) are scored and their slot takes the next game id.  Records per ply: the packed
    56-bit-per-cell features of the position, the visit policy and the mover; the value is filled
    in when the game ends (draw or length cap => -1 for both sides, self_play.py:188-189).

    Game ids: `game_ids` is an iterator of GLOBAL game indices this engine may play (default 0, 1, 2, ...; a
    multi-GPU run hands every rank its own shard, dist.game_id_stream).  All randomness of game i -- root noise,
    move resampling -- is keyed on (seed, i, turn, simulation), so its record does not depend on the slot, batch
    size, process or GPU it ran on.  When the iterator is exhausted a finishing slot goes idle.

    Finished games pile up in `finished_games` as (value_white, plies, game_id) until the caller takes them with
    `drain_finished()`; more than `max_finished_kept` undrained games are dropped and counted (`dropped_games`,
    one warning).  packed_records=True collects them as packed batches instead (records.pack_games' arrays, built with
    array operations straight from the per-ply host copies: no per-row Python objects, sparse policies) --
    `drain_finished_packed()`; what SelfPlayWorker's children send to the parent."""

    def __init__(self, games, sims, evaluator, device=None, slots=1, seed=0, plane_dtype=None,
                 keep_records=True, game_ids=None, max_finished_kept=1024, report_every=0, log=print, packed_records=False,
                 search_options=None):
        import itertools
        from .batch import BoardBatch
        self.games, self.sims = games, sims
        self.env = BoardBatch(games, device)
        self.device = self.env.device
        self.search = TreeSearch(games, sims, evaluator, self.device.index, slots, seed, plane_dtype,
                                 **(search_options or {}))      # e.g. skip_unread_rows / share_equal_leaves = False
        self.keep_records = keep_records
        self.finished = 0
        self.white_wins = self.black_wins = self.draws = 0
        self.plies = 0
        self.finished_lengths = []   # mean final turn number of the games retired at one ply (bench statistics)
        self.finished_turns = []     # final turn number of every finished game
        self.records = []            # last 8 plies on the device: (features int64[games,144], policy, mover, game_id)
        self._ids = iter(game_ids) if game_ids is not None else itertools.count(0)
        first = [next(self._ids, -1) for _ in range(games)]
        self.game_id = torch.tensor(first, dtype=torch.int64, device=self.device)
        self.active = (self.game_id >= 0).to(torch.int8)
        self.search.set_game_ids(torch.clamp(self.game_id, min=0))
        # host copies of every ply still needed by a running game (features, history, policy, records, moved?)
        self._log = []
        self._pinned_pool = []       # page-locked staging sets waiting to be reused (play_ply / _settle)
        self._start_ply = [0] * games   # ply counter at which the game in each slot started
        self._unlogged = [False] * games   # opening plies of this slot's game were played outside play_ply (stagger)
        self.finished_games = []
        self.packed_records = bool(packed_records)
        self.finished_packed = []    # packed batches of finished games (packed_records=True)
        self._packed_waiting = 0
        self.max_finished_kept = max_finished_kept
        self.dropped_games = self.unrecorded_games = 0
        self.report_every, self._log_fn, self._reported = report_every, log, 0
        # per-ply record copies leave on a side stream into pinned host buffers while the next search runs
        self._copy_stream = torch.cuda.Stream(self.device) if keep_records else None
        self._copy_done = None       # event of the newest record copy (entries of self._log are valid once it has passed)

    def stagger(self, seed=0):
        """Spread the games over plies 0..53 with uniformly random legal moves so that a timed window
        sees the steady state.  Those opening plies are not searched and not logged, so the
        games they belong to produce no training rows (their [game_len, counter] would be wrong)."""
        from .playout import pick_uniform
        gen = torch.Generator(device=self.device)
        gen.manual_seed(seed)
        target = torch.randint(0, MAX_GAME_LENGTH - 1, (self.games,), device=self.device, generator=gen)
        for ply in range(MAX_GAME_LENGTH - 1):
            over, _ = self.env.terminal()
            _, count, lst = self.env.legal(want_list=True)
            a = pick_uniform(count, lst, gen)
            go = (target > ply) & (over == 0) & (self.active != 0)
            self.env.step(torch.where(go, a, torch.full_like(a, -2)), sync=False)
        self._unlogged = [bool(t > 0) for t in target.cpu().tolist()]

    def running(self):
        """Number of slots that still hold a game."""
        return int(self.active.sum().item())

    def _retire_finished(self):
        boards, _ = self.env.export_state()
        over, winner = self.env.terminal()
        turn = boards[:, 33].to(torch.int32)
        done = ((over != 0) | (turn >= MAX_GAME_LENGTH)) & (self.active != 0)
        nd = int(done.sum().item())
        if nd:
            w = winner[done]
            self.white_wins += int((w == 1).sum().item())
            self.black_wins += int((w == 2).sum().item())
            self.draws += int((w == 0).sum().item())
            self.finished += nd
            self.finished_lengths.append(turn[done].float().mean().item())
            self.finished_turns += turn[done].cpu().tolist()
            idx = torch.nonzero(done).view(-1).to(torch.int32)
            slots = idx.cpu().tolist()
            if self.keep_records:
                self._close_games(slots, winner.cpu().tolist(), self.game_id.cpu().tolist())
            else:
                for s in slots:
                    self._start_ply[s], self._unlogged[s] = self.plies, False
            fresh = [next(self._ids, -1) for _ in slots]
            for sl, f in zip(slots, fresh):
                if f < 0:
                    self._start_ply[sl] = 1 << 60          # idle from now on: nothing of the log belongs to it
            self.game_id[idx.long()] = torch.tensor(fresh, dtype=torch.int64, device=self.device)
            self.active = (self.game_id >= 0).to(torch.int8)
            self.search.set_game_ids(torch.clamp(self.game_id, min=0))
            self.env.reset(idx)
            self._report()
        return nd

    retire_finished = _retire_finished

    def _report(self):
        """woker/self_play.py:69-75: every `report_every` games -- games so far, mean game length, white's win rate."""
        if self.report_every and self.finished - self._reported >= self.report_every:
            self._reported = self.finished - self.finished % self.report_every
            self._log_fn(f" Total_game {self.finished} ---  Mean_game_len {self.mean_game_length():.2f} ---  "
                         f"White_Win % {self.white_wins / max(self.finished, 1):.2f} ---  "
                         f"(white {self.white_wins} black {self.black_wins} draw_or_cap {self.draws})")

    def mean_game_length(self):
        """Mean number of plies of the finished games (final turn number - 1)."""
        return (sum(self.finished_turns) / len(self.finished_turns) - 1.0) if self.finished_turns else 0.0

    def _log_ready(self):
        """Wait for the newest per-ply record copy (issued a whole search ago: it has long finished)."""
        if self._copy_done is not None:
            self._copy_done.synchronize()

    def _close_games(self, slots, winner, ids):
        """self_play.py:165-191: value_white = +1 / -1 / 0; a draw or the length cap scores -1 for both."""
        import warnings
        self._log_ready()
        keep = []                        # (slot, first logged ply, value_white, game id)
        for s in slots:
            start, self._start_ply[s] = self._start_ply[s], self.plies
            unlogged, self._unlogged[s] = self._unlogged[s], False
            if unlogged:
                self.unrecorded_games += 1
                continue
            if len(self.finished_games) + self._packed_waiting + len(keep) >= self.max_finished_kept:
                if self.dropped_games == 0:
                    warnings.warn(f"SelfPlay: more than {self.max_finished_kept} finished games are waiting; call "
                                  "drain_finished() -- further games are dropped (dropped_games counts them)")
                self.dropped_games += 1
                continue
            keep.append((s, start, 1 if winner[s] == 1 else (-1 if winner[s] == 2 else 0), ids[s]))
        if self.packed_records:
            packed = self._pack_from_log(keep) if keep else None
            if packed is not None:
                self.finished_packed.append(packed)
                self._packed_waiting += len(packed["game_val"])
            return
        for s, start, vw, gid in keep:
            plies = [self.ply_record(e, s) for e in self._log if e["ply"] >= start and e["moved"][s]]
            if plies:
                self.finished_games.append((vw, plies, gid))

    @staticmethod
    def _entry_csr(e):
        """The visit policies of one logged ply (all game slots) as CSR, computed once per ply."""
        if "csr" not in e:
            from .records import _csr
            e["csr"] = _csr(e["policy"])
        return e["csr"]

    def _settle(self, e):
        """A logged ply whose device-to-host copy has completed leaves its page-locked staging buffers: the packed route
        keeps the sparse policies (the dense 1584-wide rows, 6.5 MB per ply at 1024 games, are dropped) and pageable copies of
        the small arrays; the row-wise route keeps pageable copies of everything.  The staging set goes back to the pool."""
        import numpy as np
        pinned = e.pop("_pinned", None)
        if pinned is None:
            return
        if self.packed_records:
            self._entry_csr(e)
            e.pop("policy", None)
        for k in ("feat", "policy", "boards", "hist", "moved"):
            if k in e:
                e[k] = np.array(e[k])
        self._pinned_pool.append(pinned)

    def _pack_from_log(self, keep):
        """The games in `keep` as one packed batch (records.pack_games' arrays; rows grouped by game, plies ascending) --
        the same rows ply_record would cut out one by one, gathered per logged ply with array operations."""
        import numpy as np
        S = np.asarray([k[0] for k in keep], dtype=np.int64)
        start = np.asarray([k[1] for k in keep], dtype=np.int64)
        feat, hist, meta, cnt, pidx, pval, gk, ply = [], [], [], [], [], [], [], []
        for e in self._log:
            k = np.flatnonzero(np.asarray(e["moved"])[S] & (e["ply"] >= start))
            if k.size == 0:
                continue
            sl = S[k]
            b = e["boards"]
            turn = b[sl, 33].astype(np.int64)
            persp = np.where(turn % 2 == 1, 0, 1)
            hl = b[sl, 35].astype(np.int64)
            hlen = np.where(persp == 0, hl & 15, hl >> 4)
            feat.append(e["feat"][sl])
            hist.append(e["hist"][sl, persp])
            meta.append(np.stack([hlen, turn, persp], axis=1).astype(np.uint8))
            ptr, idx, val = self._entry_csr(e)
            c = ptr[sl + 1] - ptr[sl]
            src = np.repeat(ptr[sl] - (np.cumsum(c) - c), c) + np.arange(int(c.sum()))
            cnt.append(c)
            pidx.append(idx[src])
            pval.append(val[src])
            gk.append(k)
            ply.append(np.full(k.size, e["ply"], dtype=np.int64))
        if not feat:
            return None
        gk, ply, cnt = np.concatenate(gk), np.concatenate(ply), np.concatenate(cnt)
        order = np.lexsort((ply, gk))                           # rows by game, plies ascending
        old_ptr = np.cumsum(cnt) - cnt
        c = cnt[order]
        new_ptr = np.zeros(len(c) + 1, dtype=np.int64)
        np.cumsum(c, out=new_ptr[1:])
        src = np.repeat(old_ptr[order] - new_ptr[:-1], c) + np.arange(int(new_ptr[-1]))
        rows_per_game = np.bincount(gk, minlength=len(keep))
        has = rows_per_game > 0                                 # a game without a logged row leaves no entry (as the row-wise path)
        game_ptr = np.zeros(int(has.sum()) + 1, dtype=np.int64)
        np.cumsum(rows_per_game[has], out=game_ptr[1:])
        return {"feat": np.concatenate(feat)[order], "hist": np.concatenate(hist)[order], "meta": np.concatenate(meta)[order],
                "pol_idx": np.concatenate(pidx)[src], "pol_val": np.concatenate(pval)[src], "pol_ptr": new_ptr,
                "game_ptr": game_ptr, "game_val": np.asarray([k[2] for k in keep], dtype=np.int8)[has],
                "game_id": np.asarray([k[3] for k in keep], dtype=np.int64)[has]}

    def drain_finished(self):
        """Take (and forget) every finished game collected so far: list of (value_white, plies, game_id);
        `game_rows(entry)` turns one into the reference's rows."""
        out, self.finished_games = self.finished_games, []
        return out

    def drain_finished_packed(self):
        """packed_records=True: take (and forget) the finished games collected so far as ONE packed batch
        (records.pack_games' arrays; records.unpack_game / PackedGames expand them), or None if there are none."""
        from . import records
        batches, self.finished_packed, self._packed_waiting = self.finished_packed, [], 0
        return records.concat_packed(batches) if batches else None

    @staticmethod
    def ply_record(entry, g):
        """(packed features uint64[144], history words uint32[4,2,6], valid entries, turn, policy, mover) of game
        slot g at one logged ply."""
        turn = int(entry["boards"][g, 33])
        persp = 0 if turn % 2 == 1 else 1
        hl = int(entry["boards"][g, 35])
        hlen = (hl & 15) if persp == 0 else (hl >> 4)
        if "policy" in entry:
            policy = entry["policy"][g].copy()
        else:                                    # a settled ply of the packed route keeps the sparse form only
            import numpy as np
            ptr, idx, val = entry["csr"]
            policy = np.zeros(1584, dtype=np.float32)
            policy[idx[ptr[g]:ptr[g + 1]]] = val[ptr[g]:ptr[g + 1]]
        return (entry["feat"][g].copy(), entry["hist"][g, persp].copy(), hlen, turn, policy, persp)

    def last_ply_record(self, g):
        self._log_ready()
        return self.ply_record(self._log[-1], g) if self._log and self._log[-1]["moved"][g] else None

    @staticmethod
    def game_rows(entry):
        """One finished game as the reference's JSON rows [state, policy, value, [game_len, counter]]."""
        from . import records
        return records.rows_from_game(entry)

    def finished_game_rows(self, k):
        """The k-th waiting finished game as the reference's rows."""
        return self.game_rows(self.finished_games[k])

    def play_ply(self, forced=None):
        """One move for every game.  Returns the number of games that finished before this move.
        forced: optional int32[games]; entries >= -1 replace the searched move of that slot (replaying a recorded game
        through the record path; -2 = keep the search's choice)."""
        nd = self._retire_finished()
        boards, hist = self.env.export_state()
        if self._copy_done is not None:
            # the search's policy buffer is rewritten at the end of this search: it waits for the previous ply's record copy
            torch.cuda.current_stream(self.device).wait_event(self._copy_done)
        action, policy, sum_n = self.search.search(boards, hist, active=self.active, selfplay=True,
                                                   keep_root_planes=self.keep_records)
        if self.keep_records:
            mover = (1 - (boards[:, 33] & 1)).to(torch.int8)
            feat = self.search.root_planes.view(self.games, 144)
            self.records.append((feat, policy.clone(), mover, self.game_id.clone()))
            if len(self.records) > 8:
                self.records.pop(0)
            # host copies of this ply for the per-game records (8 MB per ply at 1024 games; whole arrays, the per-game
            # rows are cut out only when a game ends).  They leave asynchronously: a side stream copies into pinned
            # buffers while the main stream goes on with the env step and the next search (self_play.py:159-160 appends a
            # row per ply in the game loop itself; here that costs the engine nothing but the PCIe transfer)
            moved = action != -2
            src = {"feat": feat, "policy": policy, "boards": boards, "hist": hist, "moved": moved}
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream(self.device))
            prev_copy = self._copy_done
            # page-locked staging buffers are recycled: a ply's arrays move to pageable memory one ply later (_settle), so
            # two or three sets exist per engine instead of one per ply of the longest running game (8-32 MB each)
            host = self._pinned_pool.pop() if self._pinned_pool else \
                {k: torch.empty(v.shape, dtype=v.dtype, pin_memory=True) for k, v in src.items()}
            with torch.cuda.stream(self._copy_stream):
                self._copy_stream.wait_event(ready)
                for k, v in src.items():
                    host[k].copy_(v, non_blocking=True)
                    v.record_stream(self._copy_stream)
                self._copy_done = torch.cuda.Event()
                self._copy_done.record(self._copy_stream)
            self._log.append({
                "ply": self.plies,
                "feat": host["feat"].numpy().view("uint64"),
                "policy": host["policy"].numpy(),
                "boards": host["boards"].numpy(),
                "hist": host["hist"].numpy().view("uint32").reshape(self.games, 2, 4, 2, 6),
                "moved": host["moved"].numpy(),
                "_pinned": host,                 # keeps the pinned tensors (the arrays above are views) alive
            })
            oldest = min(self._start_ply)
            while self._log and self._log[0]["ply"] < oldest:
                self._log.pop(0)
            if len(self._log) >= 2 and prev_copy is not None:
                # the previous ply's copy was issued a whole search ago: its sparse policies are built now, while the GPU
                # runs this ply's search (the host has nothing else to do), not when a thousand games end at once
                prev_copy.synchronize()
                self._settle(self._log[-2])
        if forced is not None:
            forced = torch.as_tensor(forced, dtype=torch.int32, device=self.device)
            action = torch.where((forced >= -1) & (action != -2), forced, action)
        # the env re-derives the legal masks and refuses anything not in them: the search's edges come
        # from the same kernels, so illegal_count() must stay 0 (asserted by the tests)
        self.env.step(action, sync=False)
        self.plies += 1
        return nd

    def leaf_histogram(self):
        """{kind: count} over every simulation since create (TreeSearch.leaf_histogram summed over the games)."""
        h = self.search.leaf_histogram().sum(0).cpu().tolist()
        return {k: int(v) for k, v in zip(LEAF_KINDS, h) if k != "unused"}

    def close(self):
        self.search.close()
        self.env.close()
