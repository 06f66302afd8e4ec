"""BoardBatch: n Hive positions resident in HBM, driven through the C ABI (include/hive_abi.h).

PyTorch is used only for device memory and streams (plumbing); every env operation is a
hand-written HIP kernel inside libhive_hip.so.
"""
import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import (BF16, BOARD_BYTES, CHW, F16, F32, HISTORY_BYTES, HIVE_LIST_CAP, HIVE_MASK_WORDS, HWC,
                   check, load)

_DT = {torch.float32: F32, torch.float16: F16, torch.bfloat16: BF16}


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


class BoardBatch:
    """Batched twin of the reference's GamePlay (hive_engine/env_hive.py:24-507)."""

    def __init__(self, n, device=None):
        L = load()
        if L.hive_device_count() <= 0 or not torch.cuda.is_available():
            raise _lib.HiveError(-2, "no HIP device visible: hive_alphazero_amd has no CPU path")
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        self.n = int(n)
        self._h = ctypes.c_void_p()
        check(L.hive_batch_create(self.n, self.device.index, ctypes.byref(self._h)))
        self._L = L
        self._sync_stream()

    def _sync_stream(self):
        s = torch.cuda.current_stream(self.device).cuda_stream
        check(self._L.hive_batch_set_stream(self._h, ctypes.c_void_p(s)))

    def close(self):
        if getattr(self, "_h", None):
            self._L.hive_batch_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- GamePlay.new_game
    def reset(self, idx=None):
        self._sync_stream()
        if idx is None:
            check(self._L.hive_batch_reset(self._h, None, self.n))
        else:
            idx = torch.as_tensor(idx, dtype=torch.int32, device=self.device).contiguous()
            check(self._L.hive_batch_reset(self._h, _ptr(idx), idx.numel()))

    # ---- GamePlay.move / skip_turn
    def step(self, actions, sync=True):
        self._sync_stream()
        actions = torch.as_tensor(actions, dtype=torch.int32, device=self.device).contiguous()
        assert actions.numel() == self.n
        check(self._L.hive_batch_step(self._h, _ptr(actions), 1 if sync else 0))

    def illegal_count(self):
        c = ctypes.c_int64(0)
        check(self._L.hive_batch_illegal_count(self._h, ctypes.byref(c)))
        return int(c.value)

    # ---- GamePlay.actions
    def legal(self, want_list=False):
        """-> (mask uint32 [n,66] viewed as int32 (11 destination boards per position), count int32 [n], list int16 [n,256] or None)."""
        self._sync_stream()
        mask = torch.empty((self.n, HIVE_MASK_WORDS), dtype=torch.int32, device=self.device)
        count = torch.empty((self.n,), dtype=torch.int32, device=self.device)
        lst = torch.empty((self.n, HIVE_LIST_CAP), dtype=torch.int16, device=self.device) if want_list else None
        check(self._L.hive_batch_legal(self._h, _ptr(mask), _ptr(count), _ptr(lst)))
        return mask, count, lst

    # ---- GamePlay.encode_board
    def encode(self, dtype=torch.float32, layout="hwc", out=None):
        """-> planes [n,12,12,56] (hwc, the reference's layout) or [n,56,12,12] (chw)."""
        self._sync_stream()
        shape = (self.n, 12, 12, 56) if layout == "hwc" else (self.n, 56, 12, 12)
        if out is None:
            out = torch.empty(shape, dtype=dtype, device=self.device)
        assert out.is_contiguous() and out.numel() == self.n * 8064 and out.dtype == dtype
        check(self._L.hive_batch_encode(self._h, _ptr(out), _DT[dtype], HWC if layout == "hwc" else CHW))
        return out

    # ---- GamePlay.game_is_over
    def terminal(self):
        self._sync_stream()
        over = torch.empty((self.n,), dtype=torch.int8, device=self.device)
        winner = torch.empty((self.n,), dtype=torch.int8, device=self.device)
        check(self._L.hive_batch_terminal(self._h, _ptr(over), _ptr(winner)))
        return over, winner

    # ---- raw records
    def export_state(self):
        self._sync_stream()
        boards = torch.empty((self.n, BOARD_BYTES), dtype=torch.uint8, device=self.device)
        hist = torch.empty((self.n, HISTORY_BYTES), dtype=torch.uint8, device=self.device)
        check(self._L.hive_batch_export(self._h, _ptr(boards), _ptr(hist)))
        return boards, hist

    def import_state(self, boards, hist=None):
        self._sync_stream()
        boards = torch.as_tensor(np.asarray(boards) if not torch.is_tensor(boards) else boards).to(
            self.device, torch.uint8).contiguous()
        assert boards.numel() == self.n * BOARD_BYTES
        if hist is not None:
            hist = torch.as_tensor(np.asarray(hist) if not torch.is_tensor(hist) else hist)
            if hist.dtype != torch.uint8:
                hist = torch.from_numpy(np.ascontiguousarray(hist.cpu().numpy()).view(np.uint8))
            hist = hist.to(self.device).contiguous()
            assert hist.numel() == self.n * HISTORY_BYTES
        check(self._L.hive_batch_import(self._h, _ptr(boards), _ptr(hist)))


def movegen(boards, want_list=False, stream=None):
    """Stateless legal-move generation over caller-owned HiveBoard records (uint8 [n,64] on the GPU)."""
    L = load()
    n = boards.shape[0]
    dev = boards.device
    mask = torch.empty((n, HIVE_MASK_WORDS), dtype=torch.int32, device=dev)
    count = torch.empty((n,), dtype=torch.int32, device=dev)
    lst = torch.empty((n, HIVE_LIST_CAP), dtype=torch.int16, device=dev) if want_list else None
    s = (stream or torch.cuda.current_stream(dev)).cuda_stream
    check(L.hive_movegen_launch(_ptr(boards), n, _ptr(mask), _ptr(count), _ptr(lst), ctypes.c_void_p(s)))
    return mask, count, lst
