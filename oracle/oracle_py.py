"""ctypes binding of oracle/libhive_oracle.so (the CPU restatement).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never by the product package.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libhive_oracle.so")


def build(force=False):
    src = os.path.join(_HERE, "hive_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libhive_oracle.so"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        vp, ip, u8p = ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p
        L.ho_state_size.restype = ip
        L.ho_new_game.argtypes = [vp]
        L.ho_import.argtypes = [vp, ip, u8p, u8p, ip]
        L.ho_move.argtypes = [vp, ip]
        L.ho_skip_turn.argtypes = [vp]
        L.ho_legal.argtypes = [vp, u8p]
        L.ho_legal.restype = ip
        L.ho_encode.argtypes = [vp, vp]
        L.ho_game_is_over.argtypes = [vp, vp]
        L.ho_game_is_over.restype = ip
        L.ho_state_key.argtypes = [vp, ctypes.c_char_p]
        L.ho_state_key.restype = ip
        L.ho_get_nmt.argtypes = [vp, u8p]
        L.ho_turn.argtypes = [vp]
        L.ho_turn.restype = ip
        L.ho_nmt_mode.argtypes = [vp]
        L.ho_nmt_mode.restype = ip
        L.ho_last_pushed.argtypes = [vp]
        L.ho_last_pushed.restype = ip
        L.ho_get_pieces.argtypes = [vp, u8p, u8p]
        L.ho_get_history.argtypes = [vp, u8p]
        L.ho_get_history.restype = ip
        L.ho_tables.argtypes = [vp, vp, vp]
        L.ho_batch_legal.argtypes = [ip, u8p, u8p, u8p, u8p, u8p]
        L.ho_batch_legal.restype = ctypes.c_long
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


class OracleGame:
    """One reference-semantics game (mirrors hive_engine/env_hive.py::GamePlay)."""

    def __init__(self):
        L = lib()
        self._buf = ctypes.create_string_buffer(L.ho_state_size())
        L.ho_new_game(self._buf)

    @classmethod
    def from_position(cls, turn, pos, lvl, nmt_mode=0):
        g = cls.__new__(cls)
        L = lib()
        g._buf = ctypes.create_string_buffer(L.ho_state_size())
        pos = np.asarray(pos, dtype=np.uint8)
        lvl = np.asarray(lvl, dtype=np.uint8)
        L.ho_import(g._buf, int(turn), _p(pos), _p(lvl), int(nmt_mode))
        return g

    def copy(self):
        g = OracleGame.__new__(OracleGame)
        g._buf = ctypes.create_string_buffer(self._buf.raw, len(self._buf))
        return g

    def move(self, a):
        lib().ho_move(self._buf, int(a))

    def skip_turn(self):
        lib().ho_skip_turn(self._buf)

    def legal_mask(self):
        m = np.zeros(1584, dtype=np.uint8)
        lib().ho_legal(self._buf, _p(m))
        return m

    def actions(self):
        return np.nonzero(self.legal_mask())[0].tolist()

    def encode_board(self):
        pl = np.zeros((12, 12, 56), dtype=np.float32)
        lib().ho_encode(self._buf, _p(pl))
        return pl

    def game_is_over(self):
        w = ctypes.c_int(0)
        over = lib().ho_game_is_over(self._buf, ctypes.byref(w))
        return bool(over), int(w.value)

    def state_key(self):
        b = ctypes.create_string_buffer(512)
        lib().ho_state_key(self._buf, b)
        return b.value.decode()

    def nmt(self):
        m = np.zeros(144, dtype=np.uint8)
        lib().ho_get_nmt(self._buf, _p(m))
        return np.nonzero(m)[0].tolist()

    @property
    def turn(self):
        return lib().ho_turn(self._buf)

    @property
    def nmt_mode(self):
        return lib().ho_nmt_mode(self._buf)

    @property
    def last_pushed(self):
        return lib().ho_last_pushed(self._buf)

    def pieces(self):
        pos = np.zeros(22, dtype=np.uint8)
        lvl = np.zeros(22, dtype=np.uint8)
        lib().ho_get_pieces(self._buf, _p(pos), _p(lvl))
        return pos, lvl

    def history(self):
        """(count, uint8[4,2,144]) -- the history entries visible to the current planes."""
        h = np.zeros((4, 2, 144), dtype=np.uint8)
        n = lib().ho_get_history(self._buf, _p(h))
        return n, h


def tables():
    nbr = np.zeros((144, 6), dtype=np.int32)
    order = np.zeros(144, dtype=np.int32)
    line = np.zeros((144, 144), dtype=np.uint8)
    lib().ho_tables(_p(nbr), _p(order), _p(line))
    return nbr, order, line


def batch_legal(turn, pos, lvl, nmt_mode, want_masks=True):
    """Movegen over n packed positions; returns (total legal count, masks uint8[n,1584] or None)."""
    turn = np.ascontiguousarray(turn, dtype=np.uint8)
    pos = np.ascontiguousarray(pos, dtype=np.uint8)
    lvl = np.ascontiguousarray(lvl, dtype=np.uint8)
    nmt_mode = np.ascontiguousarray(nmt_mode, dtype=np.uint8)
    n = turn.shape[0]
    masks = np.zeros((n, 1584), dtype=np.uint8) if want_masks else None
    total = lib().ho_batch_legal(n, _p(turn), _p(pos), _p(lvl), _p(nmt_mode),
                                 _p(masks) if want_masks else None)
    return int(total), masks
