"""SelfPlayWorker: the self-play producer of woker/self_play.py (:35-112) on a node of MI355Xs.

The reference keeps 120 games in flight in a ProcessPoolExecutor of 60 single-game workers that all talk to ONE
network thread through pipes (self_play.py:37-75).  Here one process owns one GPU (`HIP_VISIBLE_DEVICES`, set before
the child touches HIP) and plays `games_per_gpu` games in lock step entirely on that GPU -- env, tree search and
network (mcts.SelfPlay) -- so there is no cross-process inference traffic and no inter-GPU traffic at all.  The
parent only gathers finished games from a queue, prints the reference's progress line every 10 games
(self_play.py:69-75), and flushes `play_<ts>.json` files of the reference's rows every `games_per_file` games
(self_play.py:64-67,100-112).

Games are numbered globally: with `total_games` = T the ranks own contiguous shards of 0 .. T-1 (dist.game_id_stream)
and every random draw of game i is keyed on (seed, i, turn, simulation).  A game's record therefore does not depend
on how many GPUs took part, and `results` (game id -> (value_white, rows)) is identical for 1, 2 or 8 GPUs.
"""
import multiprocessing as mp
import os
import queue as queue_mod
import time


def child_device_env(gpu, environ=None):
    """Environment changes that pin a child process to logical GPU `gpu` of THIS process: {name: value or None (= unset)}.

    The parent counts devices inside whatever mask it inherited (a scheduler's HIP_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES
    = "4,5,6,7" makes them logical 0..3), so the child's mask is the gpu-th entry of that mask, not the bare index; the
    CUDA_ alias is cleared so that it cannot contradict it.  (ROCR_VISIBLE_DEVICES acts one layer below and is inherited
    unchanged: HIP's indices are already relative to it.)"""
    environ = os.environ if environ is None else environ
    mask = environ.get("HIP_VISIBLE_DEVICES") or environ.get("CUDA_VISIBLE_DEVICES")
    if mask:
        entries = [e.strip() for e in mask.split(",") if e.strip()]
        if gpu >= len(entries):
            raise RuntimeError(f"SelfPlayWorker: GPU {gpu} is outside the inherited device mask {mask!r}")
        value = entries[gpu]
    else:
        value = str(gpu)
    return {"HIP_VISIBLE_DEVICES": value, "CUDA_VISIBLE_DEVICES": None}


def _child_main(env, worker, rank, world, cfg, out):
    """Entry point of a spawned child: pin the device BEFORE anything imports torch / touches HIP, then run the worker."""
    for name, value in env.items():
        if value is None:
            os.environ.pop(name, None)
        else:
            os.environ[name] = value
    worker(rank, world, cfg, out)


def spool_dir(need_bytes=1 << 30):
    """Where the children park finished waves for the parent: tmpfs (/dev/shm) when it has room, else None (the arrays then
    travel through the queue itself).  A container's default /dev/shm is 64 MB: checked, not assumed."""
    d = os.environ.get("HIVE_SPOOL_DIR", "/dev/shm")
    try:
        st = os.statvfs(d)
        if os.access(d, os.W_OK) and st.f_bavail * st.f_frsize >= need_bytes:
            return d
    except OSError:
        pass
    return None


_SEND_SEQ = [0]


def send_packed(out, rank, packed, spool):
    """A child's hand-over of one packed batch: as a mapped tmpfs file (name + layout through the queue) when a spool
    directory is given and has room for it, else the arrays themselves through the queue."""
    if spool is not None:
        from . import records
        nbytes = sum(int(packed[k].nbytes) for k in records.PACKED_KEYS)
        try:
            st = os.statvfs(spool)
            if st.f_bavail * st.f_frsize > 2 * nbytes + (64 << 20):
                _SEND_SEQ[0] += 1
                path = os.path.join(spool, "hive_wave_%d_%d_%d.bin" % (os.getpid(), rank, _SEND_SEQ[0]))
                try:
                    layout = records.packed_to_blob(packed, path)
                except OSError:
                    if os.path.exists(path):
                        os.unlink(path)
                    raise
                out.put(("games_blob", rank, path, layout))
                return
        except OSError:
            pass
    out.put(("games", rank, packed))


def _game_worker(rank, world, cfg, out):
    """One GPU's producer (child process; _child_main has set HIP_VISIBLE_DEVICES = this rank's GPU)."""
    import torch
    from . import mcts
    from .alpha_net import ChessNet, InferenceNet
    from .dist import game_id_stream
    try:
        torch.manual_seed(cfg["net_seed"])
        net = ChessNet()
        if cfg.get("checkpoint"):
            # the reference's checkpoints are {'state_dict': ...} (self_play.py:92-96); tensors only, nothing executed
            net.load_state_dict(torch.load(cfg["checkpoint"], map_location="cpu", weights_only=True)["state_dict"])
        # "auto": fp16 when the checkpoint passes InferenceNet.range_probe, else bf16 (alpha_net.InferenceNet.__init__)
        dtype = {"auto": None, "bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[cfg.get("net_dtype", "auto")]
        evaluator = InferenceNet(net.cuda().eval(), dtype=dtype)
        if cfg.get("warmup"):
            # one ply of a throw-away engine: HIP-graph capture and GEMM tuning of this batch size happen here, once per
            # process, before the first game is on the clock
            warm = mcts.SelfPlay(cfg["games_per_gpu"], cfg["sims"], evaluator, device=0, slots=cfg["slots"], seed=cfg["seed"],
                                 keep_records=False)
            warm.play_ply()
            torch.cuda.synchronize()
            warm.close()
        ids = game_id_stream(rank, world, cfg["total_games"])
        # games that start in lock step mostly end on the same ply (the 55-turn cap): every slot's game can be waiting in
        # finished_games at once, so the keep limit must cover the whole batch (this loop drains after every ply)
        compact = cfg["row_format"] == "compact"
        sp = mcts.SelfPlay(cfg["games_per_gpu"], cfg["sims"], evaluator, device=0, slots=cfg["slots"], seed=cfg["seed"],
                           game_ids=ids, max_finished_kept=max(1024, 2 * cfg["games_per_gpu"]), packed_records=compact)
        out.put(("ready", rank, time.time(), 0, dict(evaluator.precision_report)))
        spool = spool_dir() if cfg.get("spool", True) else None
        while True:
            sp.play_ply()
            if compact:
                # the games as they left the GPU (packed features, sparse policies), every game that ended on this ply in
                # ONE message of arrays: a thousand lock-step games ending together are 55 k rows -- 430 MB to pickle as
                # per-row tuples with dense policies, 80 MB like this
                packed = sp.drain_finished_packed()
                if packed is not None:
                    send_packed(out, rank, packed, spool)
            else:
                for entry in sp.drain_finished():
                    # the reference's rows (lists of 8,064 numbers per row: fine for a few hundred games)
                    out.put(("game", rank, entry[2], entry[0], mcts.SelfPlay.game_rows(entry)))
            if sp.running() == 0:
                break
        illegal, leaves = sp.env.illegal_count(), sp.leaf_histogram()
        sp.close()
        out.put(("done", rank, sp.finished, illegal, leaves, {"dropped": sp.dropped_games, "unrecorded": sp.unrecorded_games}))
    except BaseException as exc:                 # the parent must not wait for a rank that died
        out.put(("error", rank, repr(exc), 0, None))
        raise


class SelfPlayWorker:
    def __init__(self, total_games, games_per_gpu=1024, sims=50, gpus=None, seed=0, net_seed=0, checkpoint=None, slots=1,
                 datapath="../dataSelf", games_per_file=100, report_every=10, worker=_game_worker, log=print, row_format="json",
                 net_dtype="auto", warmup=False, keep_results=True, spool=True):
        """row_format "json": results / files hold the reference's rows (play_<ts>.json, self_play.py:100-112);
        "compact": results hold the games as SelfPlay collects them and the files are play_<ts>.npz (records.save_games;
        records.dataset_from_games / rows_from_game expand them) -- the format that keeps up with a node of GPUs."""
        """keep_results False (compact only): `results` keeps the game ids and row counts but not the rows -- a long run's
        parent then holds one file's worth of games, not all of them (the reference's pool returns nothing either,
        self_play.py:54-75).  spool: children hand finished waves over as tmpfs files (spool_dir) instead of pickles."""
        if row_format not in ("json", "compact"):
            raise ValueError("row_format must be 'json' or 'compact'")
        self.keep_results = bool(keep_results) or row_format != "compact"
        self.row_format = row_format
        self.cfg = {"total_games": int(total_games), "games_per_gpu": int(games_per_gpu), "sims": int(sims), "seed": int(seed),
                    "net_seed": int(net_seed), "checkpoint": checkpoint, "slots": int(slots), "row_format": row_format,
                    "net_dtype": net_dtype, "warmup": bool(warmup), "spool": bool(spool)}
        if gpus is None:
            import torch
            gpus = list(range(torch.cuda.device_count()))      # counting devices does not initialise HIP
        self.gpus = [int(g) for g in gpus]
        if not self.gpus:
            raise RuntimeError("SelfPlayWorker: no GPU given / visible (there is no CPU path)")
        self.datapath, self.games_per_file, self.report_every = datapath, games_per_file, report_every
        self._worker, self._log = worker, log
        self.results = {}            # game id -> (value_white, rows); compact: records.PackedGames (a game expands when asked for)
        self.win_lose, self.game_lens, self.files = [], [], []
        self.buffer = []
        self._buffer_games = 0       # games waiting in `buffer` for the next file
        self._file_no = 0
        self._writers, self._pending_files = None, []
        self.precision = {}          # rank -> InferenceNet.precision_report of its evaluator (dtype chosen, probe numbers)
        self.parent_stats = {}       # filled by start(): CPU seconds of this process, deepest queue backlog seen, batches taken
        self.leaf_kinds = {}         # leaves of every simulation by kind, summed over the ranks (mcts.LEAF_KINDS)
        self.ready_at = {}           # rank -> wall-clock time its engine was built (network on the GPU, trees allocated)

    # ---------------------------------------------------------------- parent side
    def _spawn(self, out):
        ctx = mp.get_context("spawn")
        procs = []
        world = len(self.gpus)
        for rank, gpu in enumerate(self.gpus):
            # the device mask travels as an argument and is applied by the child itself before it imports torch: the
            # parent's own environment is never touched
            p = ctx.Process(target=_child_main, args=(child_device_env(gpu), self._worker, rank, world, self.cfg, out),
                            daemon=True)
            p.start()
            procs.append(p)
        return procs

    def _take(self, game_id, value_white, rows):
        if self.row_format == "compact":          # a child that sends one whole (value, plies, id) entry per message
            from . import records
            return self._take_packed(records.pack_games([rows]))
        self.results[game_id] = (value_white, rows)
        self.win_lose.append(value_white)
        self.game_lens.append(len(rows))
        self.buffer += rows
        self._buffer_games += 1
        self._after_game()

    def _after_game(self):
        n = len(self.win_lose)
        if self.games_per_file and n % self.games_per_file == 0:
            self.flush_buffer()
        if self.report_every and n % self.report_every == 0:
            wins = sum(1 for v in self.win_lose if v == 1)
            self._log(f" Total_game {n} ---  Mean_game_len {sum(self.game_lens) / n:.2f} ---  "
                      f"White_Win % {wins / n:.2f} --- ")

    def _take_packed(self, packed):
        """A packed batch of finished games (records.pack_games' arrays) from a child.  A lock-step wave hands over a
        thousand games in one batch: the files are cut out of it with array slices (one slice per file boundary, not one
        per game), and the bookkeeping is array arithmetic."""
        from . import records
        if not isinstance(self.results, records.PackedGames):
            self.results = records.PackedGames()
        if self.keep_results:
            self.results.add(packed)
        else:
            self.results.add_ids(packed)
        lens = (packed["game_ptr"][1:] - packed["game_ptr"][:-1])
        g_total = len(lens)
        before = len(self.win_lose)
        self.win_lose.extend(packed["game_val"].tolist())
        self.game_lens.extend(lens.tolist())
        lo = 0
        while lo < g_total:
            room = (self.games_per_file - self._buffer_games) if self.games_per_file else g_total - lo
            hi = min(g_total, lo + max(room, 1))
            self.buffer.append(packed if (lo == 0 and hi == g_total) else records.slice_packed(packed, lo, hi))
            self._buffer_games += hi - lo
            lo = hi
            if self.games_per_file and self._buffer_games >= self.games_per_file:
                self.flush_buffer()
        if self.report_every:
            n = len(self.win_lose)
            for m in range((before // self.report_every + 1) * self.report_every, n + 1, self.report_every):
                wins = sum(1 for v in self.win_lose[:m] if v == 1)
                self._log(f" Total_game {m} ---  Mean_game_len {sum(self.game_lens[:m]) / m:.2f} ---  "
                          f"White_Win % {wins / m:.2f} --- ")

    def flush_buffer(self):
        from . import records
        if self.buffer and self.datapath:
            if self.row_format == "compact":
                import datetime
                from concurrent.futures import ThreadPoolExecutor
                os.makedirs(self.datapath, exist_ok=True)
                # (the running file number keeps two files cut from one batch in the same microsecond apart)
                self._file_no += 1
                name = "play_%s_%04d.npz" % (datetime.datetime.now().strftime("%Y%m%d-%H%M%S.%f"), self._file_no)
                path = os.path.join(self.datapath, name)
                # compression (zlib, outside the GIL) runs on writer threads while the parent goes on receiving games;
                # start() / wait_files() joins them; a path enters `files` only once its file is on disk
                if self._writers is None:
                    self._writers = ThreadPoolExecutor(max_workers=max(4, min(8, (os.cpu_count() or 8) // 2)))
                batch = self.buffer
                self._pending_files.append((path, self._writers.submit(
                    lambda: records.save_packed(path, records.concat_packed(batch)))))
            else:
                self.files.append(records.flush_buffer(self.buffer, self.datapath))
        self.buffer = []
        self._buffer_games = 0

    def wait_files(self, raise_errors=True):
        """Every file handed to the writer threads is on disk when this returns (errors of a writer surface here)."""
        pending, self._pending_files = self._pending_files, []
        first_error = None
        for path, f in pending:
            try:
                f.result()
                self.files.append(path)
            except BaseException as exc:                  # keep joining the other writers; report the first failure
                first_error = first_error or exc
        if self._writers is not None:
            self._writers.shutdown(wait=True)
            self._writers = None
        if first_error is not None and raise_errors:
            raise first_error

    def start(self, timeout_s=None):
        """Play games 0 .. total_games-1; returns {game id: (value_white, rows)} ordered by game id.  Raises if a rank
        fails, if the env refused a move, if a rank dropped or could not record a game, or if any game id is missing."""
        from . import records  # noqa: F401  (flush_buffer uses it; imported here so that spawning stays torch-free)
        ctx = mp.get_context("spawn")
        out = ctx.Queue()
        procs = self._spawn(out)
        pending, t0 = set(range(len(procs))), time.time()
        cpu0 = time.process_time()
        lost = {}
        ok = False
        depth_max = batches = 0
        try:
            while pending:
                if timeout_s is not None and time.time() - t0 > timeout_s:       # checked on every turn of the loop
                    raise TimeoutError(f"SelfPlayWorker.start: ranks {sorted(pending)} still running after {timeout_s} s")
                try:
                    msg = out.get(timeout=1.0)
                except queue_mod.Empty:
                    dead = [r for r in pending if not procs[r].is_alive()]
                    if dead and out.empty():
                        raise RuntimeError(f"self-play rank(s) {dead} exited without reporting")
                    continue
                try:
                    depth_max = max(depth_max, out.qsize() + 1)     # messages waiting behind this one (where the OS can tell)
                except NotImplementedError:
                    pass
                if msg[0] == "game":
                    self._take(msg[2], msg[3], msg[4])
                    batches += 1
                elif msg[0] == "games":
                    self._take_packed(msg[2])
                    batches += 1
                elif msg[0] == "games_blob":
                    # (the name is gone as soon as it is mapped: a parent that dies later leaves nothing in /dev/shm)
                    self._take_packed(records.packed_from_blob(msg[2], msg[3], unlink=True))
                    batches += 1
                elif msg[0] == "ready":
                    self.ready_at[msg[1]] = msg[2]
                    if len(msg) > 4 and msg[4]:
                        self.precision[msg[1]] = msg[4]
                elif msg[0] == "done":
                    pending.discard(msg[1])
                    for kind, count in (msg[4] or {}).items():
                        self.leaf_kinds[kind] = self.leaf_kinds.get(kind, 0) + count
                    if msg[3]:
                        raise RuntimeError(f"rank {msg[1]}: the env refused {msg[3]} moves of the search")
                    extra = msg[5] if len(msg) > 5 and msg[5] else {}
                    if extra.get("dropped") or extra.get("unrecorded"):
                        lost[msg[1]] = extra
                else:
                    raise RuntimeError(f"self-play rank {msg[1]} failed: {msg[2]}")
            self.flush_buffer()               # the last file is compressed while the children shut their GPUs down
            ok = True
        finally:
            if not ok:
                for p in procs:                   # a failed run: the children may be blocked on a queue nobody drains
                    if p.is_alive():
                        p.terminate()
                try:                              # waves already parked in the spool directory: take their files away
                    while True:
                        m = out.get(timeout=0.2)
                        if m[0] == "games_blob" and os.path.exists(m[2]):
                            os.unlink(m[2])
                except (queue_mod.Empty, OSError, EOFError):
                    pass
                self.wait_files(raise_errors=False)   # join the writer threads; `files` lists what really is on disk
            for p in procs:
                p.join(timeout=10)
                if p.is_alive():
                    p.terminate()
        self.flush_buffer()
        self.wait_files()
        self.parent_stats = {"cpu_s": round(time.process_time() - cpu0, 3), "wall_s": round(time.time() - t0, 3),
                             "queue_depth_max": depth_max, "batches": batches}
        if isinstance(self.results, dict):
            self.results = dict(sorted(self.results.items()))
        else:
            self.results.sort()
        missing = sorted(set(range(self.cfg["total_games"])) - set(self.results))
        if lost or missing:
            raise RuntimeError(f"SelfPlayWorker: games lost -- per rank {lost}, missing ids {missing[:16]}"
                               f"{' ...' if len(missing) > 16 else ''} ({len(missing)} of {self.cfg['total_games']})")
        return self.results
