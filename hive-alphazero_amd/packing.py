"""Host-side (numpy) packing of positions into the HBM record formats of include/hive_abi.h."""
import numpy as np

from ._lib import BOARD_BYTES, HIVE_IN_HAND

SLOT_TYPES = "QBBSSGGGAAA"      # inventory_frame.py:47-99 piece order (reference)


def mask_words(cells):
    """iterable of cell ids -> uint32[6] bitboard (word r = rows 2r | 2r+1 << 16)."""
    w = np.zeros(6, dtype=np.uint32)
    for c in cells:
        row, col = divmod(int(c), 12)
        w[row >> 1] |= np.uint32(1 << (((row & 1) << 4) | col))
    return w


def words_to_cells(w):
    out = []
    for r in range(6):
        v = int(w[r])
        for bit in range(32):
            if (v >> bit) & 1:
                out.append((2 * r + (bit >> 4)) * 12 + (bit & 15))
    return sorted(out)


def pack_boards(turn, pos, lvl, nmt_mode, pushed=None, hist_len=None):
    """Arrays (n,), (n,22), (n,22), (n,) [, (n,), (n,2)] -> uint8[n,64] HiveBoard records."""
    turn = np.asarray(turn)
    n = turn.shape[0]
    pos = np.asarray(pos, dtype=np.uint8).reshape(n, 22)
    lvl = np.asarray(lvl, dtype=np.uint8).reshape(n, 22)
    lvl = np.where(pos == HIVE_IN_HAND, 0, lvl).astype(np.uint8)
    rec = np.zeros((n, BOARD_BYTES), dtype=np.uint8)
    rec[:, 0:22] = pos
    rec[:, 22:33] = (lvl[:, 0::2] & 15) | ((lvl[:, 1::2] & 15) << 4)
    rec[:, 33] = turn.astype(np.uint8)
    flags = np.asarray(nmt_mode, dtype=np.uint8) & 3
    if pushed is not None:
        flags = flags | (np.asarray(pushed, dtype=np.uint8) << 2)
    rec[:, 34] = flags
    if hist_len is not None:
        hl = np.asarray(hist_len, dtype=np.uint8).reshape(n, 2)
        rec[:, 35] = (hl[:, 0] & 15) | ((hl[:, 1] & 15) << 4)
    return rec


def unpack_boards(rec):
    """uint8[n,64] -> dict of arrays (turn, pos, lvl, nmt_mode, pushed, hist_len)."""
    rec = np.asarray(rec, dtype=np.uint8).reshape(-1, BOARD_BYTES)
    n = rec.shape[0]
    pos = rec[:, 0:22].copy()
    lvl = np.zeros((n, 22), dtype=np.uint8)
    lvl[:, 0::2] = rec[:, 22:33] & 15
    lvl[:, 1::2] = rec[:, 22:33] >> 4
    return {
        "turn": rec[:, 33].astype(np.int32),
        "pos": pos,
        "lvl": lvl,
        "nmt_mode": rec[:, 34] & 3,
        "pushed": (rec[:, 34] >> 2) & 1,
        "hist_len": np.stack([rec[:, 35] & 15, rec[:, 35] >> 4], axis=1),
    }


def mask_to_actions(mask_row):
    """uint32[66] legal set (11 destination boards, include/hive_abi.h) -> ascending list of action ids
    (GamePlay.encode_action, env_hive.py:287-304)."""
    w = np.asarray(mask_row, dtype=np.uint32).reshape(11, 6)
    bits = np.unpackbits(w.view(np.uint8).reshape(11, 6, 4), axis=2, bitorder="little").reshape(11, 6, 2, 16)[..., :12]
    on = bits.reshape(11, 144)                      # [slot][cell]: word r = rows 2r, 2r+1
    slot, cell = np.nonzero(on)
    return np.sort(cell * 11 + slot).tolist()


def actions_to_mask(actions):
    """ascending action ids -> uint32[66] destination boards (inverse of mask_to_actions)."""
    w = np.zeros((11, 6), dtype=np.uint32)
    for a in actions:
        cell, slot = divmod(int(a), 11)
        row, col = divmod(cell, 12)
        w[slot, row >> 1] |= np.uint32(1 << (((row & 1) << 4) | col))
    return w.reshape(66)
