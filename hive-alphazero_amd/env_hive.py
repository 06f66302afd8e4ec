"""GamePlay: drop-in mirror of the reference's hive_engine/env_hive.py::GamePlay (:24-507).

Same constructor, methods, attributes and return types, so woker/self_play.py,
woker/solo_play.py and alpha_zero/MCTS_chess.py can import this class instead.  The rules never
run in Python: the position lives in a 64-byte HiveBoard record (+ 384-byte history) and every
move / legal-move / planes / game-over question is answered by the HIP kernels of
libhive_hip.so through the C ABI -- one hive_single_advance call per move (position in, step ->
legal set -> game over as one launch chain, answers out, one synchronise).  A GamePlay object keeps
only host copies of its own record, so copy.deepcopy (solo_play.py:158, MCTS_chess.py:104) is a
copy of ~700 bytes.

For thousands of concurrent games use hive_alphazero_amd.batch.BoardBatch (same kernels, no
per-call host round trip); this class is the single-game API surface.
"""
import contextlib
import copy
import threading

import ctypes

import numpy as np
import torch

from . import _lib, packing
from .config import (MAX_MAP_FULL, PIECE_BLACK, PIECE_KEYS, PIECE_WHITE, SLOT_KEYS, STATE_FEATURES,
                     index_char, index_number)

# state.board_tiles order (tile.py:180-198): rows 11 -> 0, columns 0 -> 11
_BOARD_ORDER = [j * 12 + c for j in range(11, -1, -1) for c in range(12)]


class _Tile:
    """What callers read from board_matrix[x, y] (tile.py:10-24): index_xy and core_index."""
    __slots__ = ("index_xy", "core_index", "cell")

    def __init__(self, x, y):
        self.index_xy = [x, y]
        self.core_index = (index_char[x], index_number[y])
        self.cell = x * 12 + y


_BOARD_MATRIX = np.empty((MAX_MAP_FULL, MAX_MAP_FULL), dtype=object)
for _x in range(12):
    for _y in range(12):
        _BOARD_MATRIX[_x, _y] = _Tile(_x, _y)


class _State:
    """The slice of game_state.py::Game_State the hot-path callers touch."""

    def __init__(self):
        self.turn = 1
        self.winner = None

    def player(self):                       # game_state.py:58-62
        return 0 if self.turn % 2 == 1 else 1


class _Single:
    """One HiveSingle handle of include/hive_abi.h: a stream, a device block and its pinned host mirror."""

    def __init__(self, dev):
        self.L = _lib.load()
        if self.L.hive_device_count() <= 0:
            raise _lib.HiveError(-2, "no HIP device visible: hive_alphazero_amd has no CPU path")
        self._h = ctypes.c_void_p()
        _lib.check(self.L.hive_single_create(int(dev), ctypes.byref(self._h)))

    def advance(self, rec, hist, action, legal_before):
        """-> (record uint8[64], history uint8[384], legal set uint32[66], over, winner) after `action`
        (-1 pass, -2 none, -3 new game); raises HiveError(HIVE_E_ILLEGAL) for an action outside legal_before."""
        rec_out, hist_out = np.empty(64, dtype=np.uint8), np.empty(384, dtype=np.uint8)
        mask = np.empty(_lib.HIVE_MASK_WORDS, dtype=np.uint32)
        over, winner = ctypes.c_int8(0), ctypes.c_int8(0)
        p = lambda a: a.ctypes.data_as(ctypes.c_void_p) if a is not None else None
        _lib.check(self.L.hive_single_advance(self._h, p(rec), p(hist), int(action), p(legal_before), p(rec_out), p(hist_out), p(mask),
                                              None, ctypes.byref(over), ctypes.byref(winner)))
        return rec_out, hist_out, mask, bool(over.value), int(winner.value)

    def encode(self, rec, hist):
        planes = np.empty((12, 12, STATE_FEATURES), dtype=np.float32)
        p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        _lib.check(self.L.hive_single_encode(self._h, p(rec), p(hist), p(planes)))
        return planes

    def __del__(self):
        try:
            if self._h:
                self.L.hive_single_destroy(self._h)
        except Exception:
            pass


class _EnginePool:
    """Single-position handles of one device.  A GamePlay call borrows a handle for its one ABI call and hands it
    back, so concurrent callers (the reference searches with a ThreadPoolExecutor of SEARCH_THREADS = 32 workers,
    solo_play.py:153-165) never share one: a C-ABI handle is not thread-safe (include/hive_abi.h), distinct handles are
    independent.  At most as many handles exist as callers ever overlapped."""
    _inst = {}
    _inst_lock = threading.Lock()

    @classmethod
    def get(cls, device=None):
        dev = torch.cuda.current_device() if device is None else device
        with cls._inst_lock:
            if dev not in cls._inst:
                cls._inst[dev] = cls(dev)
            return cls._inst[dev]

    def __init__(self, dev):
        self.dev = dev
        self._free = []
        self._lock = threading.Lock()
        self.created = 0

    @contextlib.contextmanager
    def slot(self):
        with self._lock:
            b = self._free.pop() if self._free else None
            if b is None:
                self.created += 1
        if b is None:
            b = _Single(self.dev)
        try:
            yield b
        finally:
            with self._lock:
                self._free.append(b)


class GamePlay:
    def __init__(self, HEIGHT_MAP=None, WIDTH_MAP=None, second_force=False, device=None):
        self.HEIGHT_MAP, self.WIDTH_MAP = HEIGHT_MAP, WIDTH_MAP      # pixel geometry only in the reference
        self._device = device
        self.board_matrix = _BOARD_MATRIX
        self.second_force = True
        self.new_game()

    # ------------------------------------------------------------------ state plumbing
    def _slot(self):
        return _EnginePool.get(self._device).slot()

    def _advance(self, action):
        """One ABI call: [new game | move | pass] -> record, legal set, game over -- what the reference recomputes at the
        end of move() (env_hive.py:169-171: encoded_action; the planes are built lazily)."""
        with self._slot() as h:
            if action is None:
                out = h.advance(None, None, -3, None)
            else:
                out = h.advance(self._rec, self._hist, int(action), self._mask if int(action) >= 0 else None)
        self._rec, self._hist, self._mask, self._over, self._winner = out
        self.encoded_action = packing.mask_to_actions(self._mask)
        self._planes = None
        st = packing.unpack_boards(self._rec)
        self._pos, self._lvl = st["pos"][0], st["lvl"][0]
        self.state.turn = int(st["turn"][0])

    def new_game(self):                      # env_hive.py:61-97
        self.state = _State()
        self._advance(None)
        self.state_key = "." * 144 + "0"

    @property
    def white_pieces_set(self):
        return self._pieces_set(0)

    @property
    def black_pieces_set(self):
        return self._pieces_set(1)

    def _pieces_set(self, color):
        """{slot key: [tile or None (in hand), level, piece key]} in the reference's key order."""
        out = {}
        for s, key in enumerate(SLOT_KEYS):
            c = int(self._pos[color * 11 + s])
            tile = None if c == 255 else _BOARD_MATRIX[c // 12, c % 12]
            pk = PIECE_KEYS[s] if color == 0 else PIECE_KEYS[s].lower()
            out[key] = [tile, int(self._lvl[color * 11 + s]) if c != 255 else 0, pk]
        return out

    def _build_state_key(self):              # env_hive.py:151-168
        stacks = {}
        for p in range(22):
            c = int(self._pos[p])
            if c != 255:
                stacks.setdefault(c, []).append((int(self._lvl[p]), p))
        parts = []
        for c in _BOARD_ORDER:
            if c in stacks:
                for _, p in sorted(stacks[c]):
                    parts.append(PIECE_KEYS[p % 11] if p < 11 else PIECE_KEYS[p % 11].lower())
            else:
                parts.append(".")
        return "".join(parts) + str(self.state.player())

    # ------------------------------------------------------------------ reference API
    def game_is_over(self):                  # env_hive.py:58-59, move_checker.py:140-165
        # (evaluated by the kernels together with the move that produced this position)
        if self._winner == 1:
            self.state.winner = PIECE_WHITE
        elif self._winner == 2:
            self.state.winner = PIECE_BLACK
        return self._over

    def move(self, move, with_skip=False):   # env_hive.py:99-171
        self._advance(move)
        if int(move) == -1:
            self.state_key = self.state_key[:-1] + str(self.state.player())
        else:
            self.state_key = self._build_state_key()

    def actions(self):                       # env_hive.py:182-183
        return self.encoded_action

    def skip_turn(self):                     # env_hive.py:493-496 (state_key is NOT updated there)
        self._advance(-1)

    def encode_board(self, player="N"):      # env_hive.py:306-318
        mover = "W" if self.state.player() == 0 else "B"
        if player == "N":
            player = mover
        if player != mover:
            raise KeyError(player)           # state_final only holds the mover's planes
        if self._planes is None:
            with self._slot() as h:
                self._planes = h.encode(self._rec, self._hist).astype(np.float64)
        return self._planes

    def turn(self):
        return self.state.turn

    def player(self):
        return self.state.player()

    def decode_action(self, action):         # env_hive.py:498-507
        cell, slot = divmod(int(action), 11)
        return SLOT_KEYS[slot], _BOARD_MATRIX[cell // 12, cell % 12].core_index

    def encode_action(self, action_list):    # env_hive.py:287-304
        ids = set()
        for piece, tiles in action_list.items():
            slot = SLOT_KEYS.index(piece)
            for tile in tiles:
                ids.add((tile.index_xy[0] * 12 + tile.index_xy[1]) * 11 + slot)
        return sorted(ids)

    def __deepcopy__(self, memo):
        g = GamePlay.__new__(GamePlay)
        g.__dict__.update(self.__dict__)
        g.state = copy.copy(self.state)
        g._rec, g._hist, g._mask = self._rec.copy(), self._hist.copy(), self._mask.copy()
        g.encoded_action = list(self.encoded_action)
        return g
