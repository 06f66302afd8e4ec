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

from . import records
from .dist import game_id_stream


def _game_worker(rank, world, cfg, out):
    """One GPU's producer (child process; the parent exported HIP_VISIBLE_DEVICES = this rank's GPU for it)."""
    import torch
    from . import mcts
    from .alpha_net import ChessNet, InferenceNet
    try:
        torch.manual_seed(cfg["net_seed"])
        net = ChessNet()
        if cfg.get("checkpoint"):
            # the reference's checkpoints are {'state_dict': ...} (self_play.py:92-96); tensors only, nothing executed
            net.load_state_dict(torch.load(cfg["checkpoint"], map_location="cpu", weights_only=True)["state_dict"])
        evaluator = InferenceNet(net.cuda().eval(), dtype=torch.bfloat16)
        ids = game_id_stream(rank, world, cfg["total_games"])
        sp = mcts.SelfPlay(cfg["games_per_gpu"], cfg["sims"], evaluator, device=0, slots=cfg["slots"], seed=cfg["seed"],
                           game_ids=ids)
        while True:
            sp.play_ply()
            for entry in sp.drain_finished():
                # "json": the reference's rows (lists of 8,064 numbers per row: fine for a few hundred games); "compact": the
                # game as it left the GPU (packed features, sparse policy), expanded only when somebody asks
                out.put(("game", rank, entry[2], entry[0], mcts.SelfPlay.game_rows(entry) if cfg["row_format"] == "json" else entry))
            if sp.running() == 0:
                break
        illegal, leaves = sp.env.illegal_count(), sp.leaf_histogram()
        sp.close()
        out.put(("done", rank, sp.finished, illegal, leaves))
    except BaseException as exc:                 # the parent must not wait for a rank that died
        out.put(("error", rank, repr(exc), 0, None))
        raise


class SelfPlayWorker:
    def __init__(self, total_games, games_per_gpu=1024, sims=50, gpus=None, seed=0, net_seed=0, checkpoint=None, slots=1,
                 datapath="../dataSelf", games_per_file=100, report_every=10, worker=_game_worker, log=print, row_format="json"):
        """row_format "json": results / files hold the reference's rows (play_<ts>.json, self_play.py:100-112);
        "compact": results hold the games as SelfPlay collects them and the files are play_<ts>.npz (records.save_games;
        records.dataset_from_games / rows_from_game expand them) -- the format that keeps up with a node of GPUs."""
        if row_format not in ("json", "compact"):
            raise ValueError("row_format must be 'json' or 'compact'")
        self.row_format = row_format
        self.cfg = {"total_games": int(total_games), "games_per_gpu": int(games_per_gpu), "sims": int(sims), "seed": int(seed),
                    "net_seed": int(net_seed), "checkpoint": checkpoint, "slots": int(slots), "row_format": row_format}
        if gpus is None:
            import torch
            gpus = list(range(torch.cuda.device_count()))      # counting devices does not initialise HIP
        self.gpus = [int(g) for g in gpus]
        if not self.gpus:
            raise RuntimeError("SelfPlayWorker: no GPU given / visible (there is no CPU path)")
        self.datapath, self.games_per_file, self.report_every = datapath, games_per_file, report_every
        self._worker, self._log = worker, log
        self.results = {}            # game id -> (value_white, rows)
        self.win_lose, self.game_lens, self.files = [], [], []
        self.buffer = []
        self.leaf_kinds = {}         # leaves of every simulation by kind, summed over the ranks (mcts.LEAF_KINDS)

    # ---------------------------------------------------------------- parent side
    def _spawn(self, out):
        ctx = mp.get_context("spawn")
        procs = []
        world = len(self.gpus)
        saved = os.environ.get("HIP_VISIBLE_DEVICES")
        try:
            for rank, gpu in enumerate(self.gpus):
                os.environ["HIP_VISIBLE_DEVICES"] = str(gpu)    # inherited at spawn: set before the child's first HIP call
                p = ctx.Process(target=self._worker, args=(rank, world, self.cfg, out), daemon=True)
                p.start()
                procs.append(p)
        finally:
            if saved is None:
                os.environ.pop("HIP_VISIBLE_DEVICES", None)
            else:
                os.environ["HIP_VISIBLE_DEVICES"] = saved
        return procs

    def _take(self, game_id, value_white, rows):
        compact = self.row_format == "compact"
        self.results[game_id] = rows if compact else (value_white, rows)        # compact: the whole (value, plies, id) entry
        self.win_lose.append(value_white)
        self.game_lens.append(len(rows[1]) if compact else len(rows))
        if compact:
            self.buffer.append(rows)
        else:
            self.buffer += rows
        n = len(self.win_lose)
        if self.games_per_file and n % self.games_per_file == 0:
            self.flush_buffer()
        if self.report_every and n % self.report_every == 0:
            wins = sum(1 for v in self.win_lose if v == 1)
            self._log(f" Total_game {n} ---  Mean_game_len {sum(self.game_lens) / n:.2f} ---  "
                      f"White_Win % {wins / n:.2f} --- ")

    def flush_buffer(self):
        if self.buffer and self.datapath:
            if self.row_format == "compact":
                import datetime
                os.makedirs(self.datapath, exist_ok=True)
                name = "play_%s.npz" % datetime.datetime.now().strftime("%Y%m%d-%H%M%S.%f")
                self.files.append(records.save_games(os.path.join(self.datapath, name), self.buffer))
            else:
                self.files.append(records.flush_buffer(self.buffer, self.datapath))
        self.buffer = []

    def start(self, timeout_s=None):
        """Play games 0 .. total_games-1; returns {game id: (value_white, rows)} ordered by game id."""
        ctx = mp.get_context("spawn")
        out = ctx.Queue()
        procs = self._spawn(out)
        pending, t0 = set(range(len(procs))), time.time()
        try:
            while pending:
                try:
                    msg = out.get(timeout=1.0)
                except queue_mod.Empty:
                    dead = [r for r in pending if not procs[r].is_alive()]
                    if dead and out.empty():
                        raise RuntimeError(f"self-play rank(s) {dead} exited without reporting")
                    if timeout_s is not None and time.time() - t0 > timeout_s:
                        raise TimeoutError("SelfPlayWorker.start: timeout")
                    continue
                if msg[0] == "game":
                    self._take(msg[2], msg[3], msg[4])
                elif msg[0] == "done":
                    pending.discard(msg[1])
                    for kind, count in (msg[4] or {}).items():
                        self.leaf_kinds[kind] = self.leaf_kinds.get(kind, 0) + count
                    if msg[3]:
                        raise RuntimeError(f"rank {msg[1]}: the env refused {msg[3]} moves of the search")
                else:
                    raise RuntimeError(f"self-play rank {msg[1]} failed: {msg[2]}")
        finally:
            for p in procs:
                p.join(timeout=10)
                if p.is_alive():
                    p.terminate()
        self.flush_buffer()
        self.results = dict(sorted(self.results.items()))
        return self.results
