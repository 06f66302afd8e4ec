"""Self-play records in the reference's wire format (first "next" row) and in a compact one.

woker/self_play.py:100-112,178-193 writes `play_<ts>.json` = list of
`[state[12][12][56], policy[1584], value, [game_len, counter]]`; woker/optimize.py:42-65 loads it
and applies `value * 0.99 ** (game_len - step)`.  The GPU engine keeps, per ply, the packed 56-bit
features of every game (1,152 B instead of the 64 KB float64 planes), the visit policy and the mover;
this module expands finished games into those entries.

One MI355X plays ~4,000 games/min = ~3,400 rows/s; as JSON that is ~13 GB/min of text (8,064 numbers per row), more than
Python can format.  `save_games / load_games` therefore keep finished games as they leave the GPU -- packed features,
history bitboards, sparse visit policy: ~1.5 KB per row in an .npz -- and `dataset_from_games` / `rows_from_game` expand
them into exactly what the JSON route yields (same planes, same discounted values) when the trainer asks.
"""
import json
import os
from datetime import datetime

import numpy as np

from .config import DISCOUNTED_REWARD


def unpack_features(words, turn, history_planes=None):
    """uint64[144] packed features (+ the raw turn number) -> float64 [12,12,56] planes as
    GamePlay.encode_board returns them.  History planes 36..43 are not part of the packed word; pass
    them (bool [12,12,8]) when the trainer needs them."""
    w = np.asarray(words).astype(np.uint64).reshape(144)
    bits = ((w[:, None] >> np.arange(56, dtype=np.uint64)[None, :]) & np.uint64(1)).astype(np.float64)
    planes = bits.reshape(12, 12, 56)
    planes[:, :, 31] = float(turn)
    if history_planes is not None:
        planes[:, :, 36:44] = history_planes
    return planes


def history_planes(hist_words, hist_len):
    """uint32[4,2,6] history bitboards of the mover's perspective (+ how many entries are valid) ->
    bool [12,12,8] planes 36..43 (env_hive.py:431-434)."""
    out = np.zeros((12, 12, 8))
    hw = np.asarray(hist_words).astype(np.uint32).reshape(4, 2, 6)
    for age in range(min(int(hist_len), 4)):
        for k in range(2):
            for r in range(6):
                v = int(hw[age, k, r])
                for bit in range(32):
                    if (v >> bit) & 1:
                        out[2 * r + (bit >> 4), bit & 15, 2 * age + k] = 1.0
    return out


def game_entries(plies, value_white):
    """plies: list of (planes [12,12,56], policy [1584], player 'W'/'B') in game order ->
    the reference's data rows (self_play.py:178-191)."""
    counts = {"W": sum(1 for p in plies if p[2] == "W"), "B": sum(1 for p in plies if p[2] == "B")}
    seen = {"W": 0, "B": 0}
    out = []
    for planes, policy, player in plies:
        seen[player] += 1
        value = value_white if player == "W" else -value_white
        if value_white == 0:
            value = -1                      # draw or length cap counts as a loss for both sides
        out.append([np.asarray(planes).tolist(), [float(x) for x in policy], value, [counts[player], seen[player]]])
    return out


def write_game_data_to_file(path, data):    # woker/sl.py:49-60
    with open(path, "wt") as f:
        json.dump(data, f)


def flush_buffer(buffer, datapath="../dataSelf"):       # self_play.py:100-112
    os.makedirs(datapath, exist_ok=True)
    game_id = datetime.now().strftime("%Y%m%d-%H%M%S.%f")
    path = os.path.join(datapath, "play_%s.json" % game_id)
    write_game_data_to_file(path, buffer)
    return path


def load_data(filename):                     # woker/optimize.py:42-65
    with open(filename, "rt") as f:
        data = json.load(f)
    rows = []
    for state, policy, value, game_lens in data:
        game_len, step = game_lens[0], game_lens[1]
        if step != game_len:
            value = value * DISCOUNTED_REWARD ** (game_len - step)
        rows.append([np.array(state), np.array(policy, dtype=np.float32), value])
    return rows


# ------------------------------------------------------------------ compact game files
def rows_from_game(entry):
    """One finished game (value_white, plies[, game id]) as SelfPlay collects it -> the reference's rows."""
    expanded = []
    for words, hist, hlen, turn, policy, mover in entry[1]:
        expanded.append((unpack_features(words, turn, history_planes(hist, hlen)), policy, "W" if mover == 0 else "B"))
    return game_entries(expanded, entry[0])


PACKED_KEYS = ("feat", "hist", "meta", "pol_idx", "pol_val", "pol_ptr", "game_ptr", "game_val", "game_id")


def pack_games(games):
    """Finished games (value_white, plies[, game id]) -> the arrays a play_<ts>.npz holds (and the form SelfPlayWorker's
    children send): feat u64[R,144], hist u32[R,4,2,6], meta u8[R,3] = (valid history entries, turn, mover), the visit
    policies as one CSR matrix (pol_ptr i64[R+1], pol_idx i16, pol_val f32), game_ptr i64[G+1] (rows of game g),
    game_val i8[G], game_id i64[G] (-1 = unknown).  Rows are grouped by game, plies ascending."""
    feat, hist, meta, pol, game_ptr, game_val, game_id = [], [], [], [], [0], [], []
    for entry in games:
        for words, hw, hlen, turn, policy, mover in entry[1]:
            feat.append(np.asarray(words, dtype=np.uint64).reshape(144))
            hist.append(np.asarray(hw, dtype=np.uint32).reshape(4, 2, 6))
            meta.append((int(hlen), int(turn), int(mover)))
            pol.append(np.asarray(policy, dtype=np.float32).reshape(1584))
        game_ptr.append(len(feat))
        game_val.append(int(entry[0]))
        game_id.append(int(entry[2]) if len(entry) > 2 else -1)
    pol_ptr, pol_idx, pol_val = _csr(np.stack(pol) if pol else np.zeros((0, 1584), np.float32))
    return {"feat": np.stack(feat) if feat else np.zeros((0, 144), np.uint64),
            "hist": np.stack(hist) if hist else np.zeros((0, 4, 2, 6), np.uint32),
            "meta": np.asarray(meta, dtype=np.uint8).reshape(-1, 3), "pol_idx": pol_idx, "pol_val": pol_val, "pol_ptr": pol_ptr,
            "game_ptr": np.asarray(game_ptr, dtype=np.int64), "game_val": np.asarray(game_val, dtype=np.int8),
            "game_id": np.asarray(game_id, dtype=np.int64)}


def _csr(dense):
    """float32 [R,1584] -> (ptr i64[R+1], column i16[N], value f32[N]) of the non-zero entries, columns ascending per row."""
    r, c = np.nonzero(dense)
    ptr = np.zeros(dense.shape[0] + 1, dtype=np.int64)
    np.cumsum(np.bincount(r, minlength=dense.shape[0]), out=ptr[1:])
    return ptr, c.astype(np.int16), dense[r, c].astype(np.float32)


def packed_games(packed):
    return len(packed["game_val"])


def slice_packed(packed, lo, hi):
    """Games lo .. hi-1 of a packed batch as a packed batch of their own."""
    r0, r1 = int(packed["game_ptr"][lo]), int(packed["game_ptr"][hi])
    p0, p1 = int(packed["pol_ptr"][r0]), int(packed["pol_ptr"][r1])
    return {"feat": packed["feat"][r0:r1], "hist": packed["hist"][r0:r1], "meta": packed["meta"][r0:r1],
            "pol_idx": packed["pol_idx"][p0:p1], "pol_val": packed["pol_val"][p0:p1],
            "pol_ptr": packed["pol_ptr"][r0:r1 + 1] - p0, "game_ptr": packed["game_ptr"][lo:hi + 1] - r0,
            "game_val": packed["game_val"][lo:hi], "game_id": packed["game_id"][lo:hi]}


def concat_packed(batches):
    """Several packed batches -> one (game order = batch order)."""
    batches = list(batches)
    if len(batches) == 1:
        return batches[0]
    if not batches:
        return pack_games([])
    out = {k: np.concatenate([b[k] for b in batches]) for k in ("feat", "hist", "meta", "pol_idx", "pol_val", "game_val", "game_id")}
    for key, unit in (("pol_ptr", "pol_idx"), ("game_ptr", "meta")):
        parts, base = [np.zeros(1, dtype=np.int64)], 0
        for b in batches:
            parts.append(b[key][1:] + base)
            base += len(b[unit])
        out[key] = np.concatenate(parts)
    return out


def unpack_game(packed, g):
    """Game g of a packed batch as SelfPlay collects it: (value_white, plies, game id), plies =
    (features uint64[144], history words uint32[4,2,6], valid entries, turn, dense policy float32[1584], mover)."""
    pol_idx, pol_val, pol_ptr, meta = packed["pol_idx"], packed["pol_val"], packed["pol_ptr"], packed["meta"]
    plies = []
    for r in range(int(packed["game_ptr"][g]), int(packed["game_ptr"][g + 1])):
        policy = np.zeros(1584, dtype=np.float32)
        lo, hi = int(pol_ptr[r]), int(pol_ptr[r + 1])
        policy[pol_idx[lo:hi]] = pol_val[lo:hi]
        plies.append((packed["feat"][r], packed["hist"][r], int(meta[r, 0]), int(meta[r, 1]), policy, int(meta[r, 2])))
    return (int(packed["game_val"][g]), plies, int(packed["game_id"][g]))


class PackedGames:
    """Read-only mapping game id -> (value_white, plies, game id) over packed batches; a game is expanded when it is asked
    for (SelfPlayWorker.results in the compact format: 1024 finished games are 55 k dense 1584-wide policies otherwise)."""

    def __init__(self):
        self._where = {}

    def add(self, packed):
        for g, gid in enumerate(packed["game_id"].tolist()):
            self._where[int(gid)] = (packed, g)

    def add_ids(self, packed):
        """Remember which games were seen and how long they were, not their rows (SelfPlayWorker(keep_results=False))."""
        lens = (packed["game_ptr"][1:] - packed["game_ptr"][:-1]).tolist()
        for g, gid in enumerate(packed["game_id"].tolist()):
            self._where[int(gid)] = (None, lens[g])

    def sort(self):
        self._where = dict(sorted(self._where.items()))

    def __getitem__(self, gid):
        packed, g = self._where[gid]
        if packed is None:
            raise KeyError(f"game {gid} was written to a file and not kept (keep_results=False)")
        return unpack_game(packed, g)

    def __iter__(self):
        return iter(self._where)

    def __len__(self):
        return len(self._where)

    def __contains__(self, gid):
        return gid in self._where

    def keys(self):
        return self._where.keys()

    def values(self):
        return (self[k] for k in self._where)

    def items(self):
        return ((k, self[k]) for k in self._where)

    def rows_of(self, gid):
        """Number of rows (plies) of a game without expanding it."""
        packed, g = self._where[gid]
        if packed is None:
            return int(g)
        return int(packed["game_ptr"][g + 1] - packed["game_ptr"][g])


def save_packed(path, packed, level=1):
    """One packed batch -> one compressed .npz (what np.savez_compressed writes, read back by np.load), ~1.6 KB per row.
    Deflate level 1: 2.5x faster than numpy's fixed level 6 for 9 % larger files -- the writer threads of an 8-GPU
    SelfPlayWorker parent compress ~55 MB/s of records (tests/test_host_cpu.py, parent-ingest soak)."""
    import zipfile
    with zipfile.ZipFile(path, "w", zipfile.ZIP_DEFLATED, compresslevel=level) as z:
        for k in PACKED_KEYS:
            with z.open(k + ".npy", "w", force_zip64=True) as f:
                np.lib.format.write_array(f, np.ascontiguousarray(packed[k]), allow_pickle=False)
    return path


def save_games(path, games):
    """Finished games (SelfPlay.finished_games / drain_finished entries) -> one compressed .npz."""
    return save_packed(path, pack_games(games))


def packed_to_blob(packed, path):
    """A packed batch as ONE raw file (arrays back to back, 64-byte aligned) + the layout that finds them again.  The hand-over
    format between a self-play child and the gathering parent (self_play.SelfPlayWorker): 90 MB through a multiprocessing
    queue cost the parent's main thread ~0.5-1.2 s of pipe reads and unpickling per lock-step wave; mapped from a tmpfs
    file it costs a millisecond, and the writer threads compress straight out of the mapping."""
    layout, off = [], 0
    for k in PACKED_KEYS:
        a = np.ascontiguousarray(packed[k])
        layout.append((k, a.dtype.str, tuple(a.shape), off))
        off += (a.nbytes + 63) // 64 * 64
    with open(path, "wb") as f:
        f.truncate(max(off, 1))
        for (k, _, _, o) in layout:
            f.seek(o)
            f.write(np.ascontiguousarray(packed[k]).data)
    return layout


def packed_from_blob(path, layout, unlink=True):
    """Inverse of packed_to_blob: read-only array views into a private mapping of the file.  With `unlink` the name is
    removed at once -- the mapping keeps the bytes alive exactly as long as somebody holds a view."""
    import mmap
    with open(path, "rb") as f:
        size = os.fstat(f.fileno()).st_size
        mm = mmap.mmap(f.fileno(), size, prot=mmap.PROT_READ)
    if unlink:
        os.unlink(path)
    out = {}
    for k, dt, shape, off in layout:
        n = int(np.prod(shape)) if len(shape) else 1
        out[k] = np.frombuffer(mm, dtype=np.dtype(dt), count=n, offset=off).reshape(shape)
    return out


def load_packed(path):
    with np.load(path) as z:                      # plain arrays only: nothing in the file is executed
        return {k: z[k] for k in PACKED_KEYS}


def load_games(path):
    """Inverse of save_games: list of (value_white, plies, game id) with plies as SelfPlay.ply_record builds them."""
    packed = load_packed(path)
    return [unpack_game(packed, g) for g in range(packed_games(packed))]


def dataset_from_games(games):
    """What woker/optimize.py:42-65 builds from the JSON rows, straight from compact games: (states float32 [N,12,12,56],
    policies float32 [N,1584], values float32 [N]) with value * 0.99 ** (game_len - step) applied."""
    states, policies, values = [], [], []
    for entry in games:
        vw, plies = entry[0], entry[1]
        total = [sum(1 for p in plies if p[5] == s) for s in (0, 1)]
        seen = [0, 0]
        for words, hw, hlen, turn, policy, mover in plies:
            seen[mover] += 1
            value = -1.0 if vw == 0 else float(vw if mover == 0 else -vw)
            if seen[mover] != total[mover]:
                value = value * DISCOUNTED_REWARD ** (total[mover] - seen[mover])
            states.append(unpack_features(words, turn, history_planes(hw, hlen)).astype(np.float32))
            policies.append(np.asarray(policy, dtype=np.float32))
            values.append(value)
    return np.stack(states), np.stack(policies), np.asarray(values, dtype=np.float32)


def dataset_tensors_gpu(games, device=None, dtype=None, layout="chw"):
    """dataset_from_games on the GPU: the packed features, history bitboards and turn bytes of every row are uploaded
    (1.5 KB per row) and widened to planes by the env's own plane writer (hive_expand_launch) -- what a trainer that
    keeps up with the self-play GPUs uses.  -> (states [N,56,12,12] ("chw", what ChessNet takes) or [N,12,12,56],
    policies float32 [N,1584], values float32 [N]) on `device`, values discounted like woker/optimize.py:42-65."""
    import ctypes
    import torch
    from . import _lib
    L = _lib.load()
    dev = torch.device("cuda", torch.cuda.current_device() if device is None else device)
    dtype = dtype or torch.float32
    rows = [(g[0], p, sum(1 for q in g[1] if q[5] == p[5]), sum(1 for q in g[1][:k + 1] if q[5] == p[5]))
            for g in games for k, p in enumerate(g[1])]
    n = len(rows)
    if n == 0:                      # nothing to widen (hive_expand_launch refuses n <= 0)
        shape = (0, 56, 12, 12) if layout == "chw" else (0, 12, 12, 56)
        return (torch.zeros(shape, dtype=dtype, device=dev), torch.zeros((0, 1584), dtype=torch.float32, device=dev),
                torch.zeros((0,), dtype=torch.float32, device=dev))
    boards = np.zeros((n, 64), dtype=np.uint8)
    hist = np.zeros((n, 2, 4, 2, 6), dtype=np.uint32)
    feat = np.zeros((n, 144), dtype=np.uint64)
    policies = np.zeros((n, 1584), dtype=np.float32)
    values = np.zeros(n, dtype=np.float32)
    for i, (vw, (words, hw, hlen, turn, policy, mover), total, seen) in enumerate(rows):
        feat[i] = np.asarray(words, dtype=np.uint64).reshape(144)
        hist[i, mover] = np.asarray(hw, dtype=np.uint32).reshape(4, 2, 6)
        boards[i, 33] = turn
        boards[i, 35] = (int(hlen) & 15) if mover == 0 else (int(hlen) << 4)
        policies[i] = policy
        value = -1.0 if vw == 0 else float(vw if mover == 0 else -vw)
        values[i] = value if seen == total else value * DISCOUNTED_REWARD ** (total - seen)
    shape = (n, 56, 12, 12) if layout == "chw" else (n, 12, 12, 56)
    planes = torch.empty(shape, dtype=dtype, device=dev)
    tb, th = torch.from_numpy(boards).to(dev), torch.from_numpy(hist.view(np.uint8).reshape(n, 384)).to(dev)
    tf = torch.from_numpy(feat.view(np.int64)).to(dev)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    dt = {torch.float32: _lib.F32, torch.float16: _lib.F16, torch.bfloat16: _lib.BF16}[dtype]
    _lib.check(L.hive_expand_launch(p(tb), p(th), p(tf), n, p(planes), dt, _lib.CHW if layout == "chw" else _lib.HWC,
                                    ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
    return planes, torch.from_numpy(policies).to(dev), torch.from_numpy(values).to(dev)


def packed_values(packed):
    """float32[R]: the training value of every row of a packed batch -- value_white from the mover's side (draw / length cap
    = -1 for both, self_play.py:178-191), discounted by 0.99 ** (moves of that side still to come), optimize.py:42-65."""
    gp = packed["game_ptr"]
    n = int(gp[-1]) if len(gp) else 0
    gi = np.repeat(np.arange(len(gp) - 1), np.diff(gp))
    mover = packed["meta"][:, 2].astype(np.int64)
    vw = packed["game_val"].astype(np.int64)[gi]
    value = np.where(vw == 0, -1.0, np.where(mover == 0, vw, -vw).astype(np.float64))
    left = np.zeros(n, dtype=np.int64)                        # moves of the row's side after this one
    for side in (0, 1):
        c = np.concatenate([[0], np.cumsum(mover == side)])
        seen = c[1:] - c[gp[:-1]][gi]
        total = (c[gp[1:]] - c[gp[:-1]])[gi]
        left = np.where(mover == side, total - seen, left)
    table = np.asarray([DISCOUNTED_REWARD ** k for k in range(int(left.max()) + 1 if n else 1)])
    return np.where(left > 0, value * table[left], value).astype(np.float32)


def dataset_tensors_gpu_packed(packed, device=None, dtype=None, layout="chw"):
    """dataset_tensors_gpu for a packed batch (records.load_packed / SelfPlay.drain_finished_packed) without a Python
    object per row: the packed features, history words and turn bytes go up as they are (1.5 KB per row), the env's plane
    writer widens them (hive_expand_launch), the sparse policies are scattered into the dense matrix on the GPU."""
    import ctypes
    import torch
    from . import _lib
    L = _lib.load()
    dev = torch.device("cuda", torch.cuda.current_device() if device is None else device)
    dtype = dtype or torch.float32
    n = len(packed["meta"])
    shape = (n, 56, 12, 12) if layout == "chw" else (n, 12, 12, 56)
    if n == 0:
        return (torch.zeros(shape, dtype=dtype, device=dev), torch.zeros((0, 1584), dtype=torch.float32, device=dev),
                torch.zeros((0,), dtype=torch.float32, device=dev))
    meta = packed["meta"].astype(np.int64)
    hlen, turn, mover = meta[:, 0], meta[:, 1], meta[:, 2]
    boards = np.zeros((n, 64), dtype=np.uint8)
    boards[:, 33] = turn
    boards[:, 35] = np.where(mover == 0, hlen & 15, (hlen << 4) & 255)
    hist = np.zeros((n, 2, 4, 2, 6), dtype=np.uint32)
    hist[np.arange(n), mover] = packed["hist"]
    planes = torch.empty(shape, dtype=dtype, device=dev)
    tb, th = torch.from_numpy(boards).to(dev), torch.from_numpy(hist.view(np.uint8).reshape(n, 384)).to(dev)
    tf = torch.from_numpy(np.ascontiguousarray(packed["feat"]).view(np.int64)).to(dev)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    dt = {torch.float32: _lib.F32, torch.float16: _lib.F16, torch.bfloat16: _lib.BF16}[dtype]
    _lib.check(L.hive_expand_launch(p(tb), p(th), p(tf), n, p(planes), dt, _lib.CHW if layout == "chw" else _lib.HWC,
                                    ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
    rows = np.repeat(np.arange(n, dtype=np.int64), np.diff(packed["pol_ptr"]))
    flat = torch.from_numpy(rows * 1584 + packed["pol_idx"].astype(np.int64)).to(dev)
    policies = torch.zeros((n * 1584,), dtype=torch.float32, device=dev)
    policies[flat] = torch.from_numpy(np.ascontiguousarray(packed["pol_val"])).to(dev)
    return planes, policies.view(n, 1584), torch.from_numpy(packed_values(packed)).to(dev)
