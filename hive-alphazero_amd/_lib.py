"""Loader of libhive_hip.so (the C ABI of include/hive_abi.h) through ctypes.

There is no CPU implementation behind this package: if the shared library is missing or no
HIP device is visible, calls fail loudly (HiveError) instead of falling back.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
SO_PATH = os.path.join(_HERE, "libhive_hip.so")

HIVE_CELLS = 144
HIVE_PIECES = 22
HIVE_ACTIONS = 1584
HIVE_PLANES = 56
HIVE_MASK_WORDS = 66
HIVE_LIST_CAP = 256
HIVE_IN_HAND = 255
BOARD_BYTES = 64
HISTORY_BYTES = 384

F32, F16, BF16 = 0, 1, 2
HWC, CHW = 0, 1

# every symbol include/hive_abi.h declares (tests/test_abi_symbols.py checks the header against this)
ABI_SYMBOLS = [
    "hive_last_error", "hive_version", "hive_device_count", "hive_batch_create", "hive_batch_destroy",
    "hive_batch_size", "hive_batch_set_stream", "hive_batch_reset", "hive_batch_step",
    "hive_batch_illegal_count", "hive_batch_legal", "hive_batch_encode", "hive_batch_terminal",
    "hive_batch_export", "hive_batch_import", "hive_movegen_launch", "hive_movegen_pair_threshold", "hive_encode_launch",
    "hive_terminal_launch", "hive_step_launch", "hive_step_launch_counted", "hive_leaf_launch", "hive_expand_launch", "hive_leaf_dedup_launch",
    "hive_single_create", "hive_single_destroy", "hive_single_advance", "hive_single_encode",
    "hive_leaf_store_create", "hive_leaf_store_destroy", "hive_leaf_store_clear", "hive_leaf_store_lookup", "hive_leaf_store_update",
    "hive_leaf_store_stats",
    # include/hive_search.h
    "hive_search_create", "hive_search_destroy", "hive_search_set_stream", "hive_search_set_params",
    "hive_search_set_roots", "hive_search_select", "hive_search_backup", "hive_search_policy",
    "hive_search_node_counts", "hive_search_set_transpositions", "hive_search_transposition_hits",
    "hive_search_set_game_ids", "hive_search_root_stats", "hive_search_leaf_histogram", "hive_search_sample_noise",
    "hive_search_leaf_need",
    # include/hive_nn.h
    "hive_nn_conv3x3", "hive_nn_resblock", "hive_nn_conv3x3_dt", "hive_nn_resblock_dt", "hive_nn_tower",
    "hive_nn_conv3x3_sel", "hive_nn_resblock_sel", "hive_nn_copy_rows", "hive_nn_tower72", "hive_nn_compact_rows", "hive_nn_tower72_balanced", "hive_nn_tower72_plan_bytes", "hive_nn_conv72", "hive_nn_conv72_add", "hive_nn_conv72_stats",
    "hive_nn_heads", "hive_nn_heads_workspace_bytes", "hive_nn_heads_splits",
    "hive_nn_bn_workspace_floats", "hive_nn_bn_act_fwd", "hive_nn_bn_act_fwd_partial", "hive_nn_bn_act_bwd",
    "hive_nn_pack_conv3x3_weights", "hive_nn_pack_conv3x3_weights_multi", "hive_nn_conv3x3_wgrad", "hive_nn_conv3x3_wgrad_layout", "hive_nn_wgrad_workspace_floats",
]


class HiveError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"hive ABI error {code}: {msg}")
        self.code = code


def build(force=False):
    """Compile csrc/*.hip for gfx950 into libhive_hip.so (hipcc cross-compiles without a GPU)."""
    srcs = [os.path.join(_CSRC, f) for f in os.listdir(_CSRC) if f.endswith((".hip", ".hpp", "gen_tower_asm.py"))]
    srcs.append(os.path.join(os.path.dirname(_HERE), "include", "hive_abi.h"))
    stale = (not os.path.exists(SO_PATH)) or any(os.path.getmtime(s) > os.path.getmtime(SO_PATH) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", _CSRC, "-s"] + (["-B"] if force else []))
    return SO_PATH


_lib = None


def load():
    """dlopen the library and declare the signatures.  Raises if the .so is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise HiveError(-2, f"{SO_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                            "(the HIP extension is mandatory, there is no CPU path)")
    L = ctypes.CDLL(SO_PATH)
    vp, i32 = ctypes.c_void_p, ctypes.c_int
    L.hive_last_error.restype = ctypes.c_char_p
    L.hive_version.restype = ctypes.c_char_p
    L.hive_device_count.restype = i32
    L.hive_batch_create.argtypes = [i32, i32, ctypes.POINTER(vp)]
    L.hive_batch_destroy.argtypes = [vp]
    L.hive_batch_size.argtypes = [vp]
    L.hive_batch_set_stream.argtypes = [vp, vp]
    L.hive_batch_reset.argtypes = [vp, vp, i32]
    L.hive_batch_step.argtypes = [vp, vp, i32]
    L.hive_batch_illegal_count.argtypes = [vp, ctypes.POINTER(ctypes.c_int64)]
    L.hive_batch_legal.argtypes = [vp, vp, vp, vp]
    L.hive_batch_encode.argtypes = [vp, vp, i32, i32]
    L.hive_batch_terminal.argtypes = [vp, vp, vp]
    L.hive_batch_export.argtypes = [vp, vp, vp]
    L.hive_batch_import.argtypes = [vp, vp, vp]
    L.hive_movegen_launch.argtypes = [vp, i32, vp, vp, vp, vp]
    L.hive_movegen_pair_threshold.argtypes = [i32]
    L.hive_encode_launch.argtypes = [vp, vp, i32, vp, i32, i32, vp, vp]
    L.hive_debug_tables.argtypes = [vp, vp, vp]
    L.hive_expand_launch.argtypes = [vp, vp, vp, i32, vp, i32, i32, vp]
    L.hive_terminal_launch.argtypes = [vp, i32, vp, vp, vp]
    L.hive_step_launch.argtypes = [vp, vp, i32, vp, vp, vp]
    L.hive_step_launch_counted.argtypes = [vp, vp, i32, vp, vp, vp, vp]
    L.hive_leaf_launch.argtypes = [vp, vp, i32, vp, i32, i32, vp, vp, vp, vp, vp, vp]
    L.hive_single_create.argtypes = [i32, ctypes.POINTER(vp)]
    L.hive_single_destroy.argtypes = [vp]
    L.hive_single_advance.argtypes = [vp, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp]
    L.hive_single_encode.argtypes = [vp, vp, vp, vp]
    L.hive_search_create.argtypes = [i32, i32, i32, i32, ctypes.c_uint64, ctypes.POINTER(vp)]
    L.hive_search_destroy.argtypes = [vp]
    L.hive_search_set_stream.argtypes = [vp, vp]
    L.hive_search_set_params.argtypes = [vp, vp]
    L.hive_search_set_roots.argtypes = [vp, vp, vp, vp]
    L.hive_search_select.argtypes = [vp, i32, vp, vp]
    L.hive_search_backup.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, vp]
    L.hive_search_policy.argtypes = [vp, vp, vp, vp, i32]
    L.hive_search_node_counts.argtypes = [vp, vp]
    L.hive_search_set_transpositions.argtypes = [vp, i32]
    L.hive_search_transposition_hits.argtypes = [vp, vp]
    L.hive_search_set_game_ids.argtypes = [vp, vp]
    L.hive_search_root_stats.argtypes = [vp, vp, vp, vp]
    L.hive_search_leaf_histogram.argtypes = [vp, vp]
    L.hive_search_sample_noise.argtypes = [ctypes.c_uint64, ctypes.c_int64, i32, ctypes.c_float, i32, i32, vp, ctypes.c_float,
                                           vp, vp]
    L.hive_nn_conv3x3.argtypes = [vp, i32, vp, vp, vp, vp, i32, i32, vp]
    L.hive_nn_resblock.argtypes = [vp, vp, vp, vp, vp, vp, i32, vp]
    L.hive_nn_conv3x3_dt.argtypes = [vp, i32, vp, vp, vp, vp, i32, i32, i32, vp]
    L.hive_nn_resblock_dt.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, vp]
    L.hive_nn_tower.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, vp]
    L.hive_nn_conv3x3_sel.argtypes = [vp, i32, vp, vp, vp, vp, i32, i32, i32, vp, vp]
    L.hive_nn_resblock_sel.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, vp, vp]
    L.hive_search_leaf_need.argtypes = [vp, i32, vp, vp, vp, vp]
    L.hive_nn_copy_rows.argtypes = [vp, vp, i32, ctypes.c_longlong, vp]
    L.hive_nn_tower72.argtypes = [vp, vp, vp, vp, i32, i32, i32, vp, vp, vp]
    L.hive_nn_compact_rows.argtypes = [vp, i32, vp, vp, vp]
    L.hive_nn_tower72_balanced.argtypes = [vp, vp, vp, vp, i32, i32, i32, vp, vp, vp, vp]
    L.hive_nn_conv72.argtypes = [vp, vp, vp, vp, i32, i32, i32, vp]
    L.hive_nn_conv72_add.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, vp]
    L.hive_nn_conv72_stats.argtypes = [vp, vp, vp, vp, i32, i32, i32, vp, vp]
    L.hive_nn_tower72_plan_bytes.argtypes = [i32]
    L.hive_nn_tower72_plan_bytes.restype = ctypes.c_longlong
    L.hive_nn_heads.argtypes = [vp, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.hive_nn_heads_workspace_bytes.argtypes = [i32]
    L.hive_nn_heads_workspace_bytes.restype = ctypes.c_longlong
    L.hive_nn_heads_splits.argtypes = [i32]
    L.hive_leaf_dedup_launch.argtypes = [vp, vp, i32, vp, vp, vp, vp, vp]
    L.hive_leaf_store_create.argtypes = [i32, i32, ctypes.POINTER(vp)]
    L.hive_leaf_store_destroy.argtypes = [vp]
    L.hive_leaf_store_clear.argtypes = [vp, vp]
    L.hive_leaf_store_lookup.argtypes = [vp, vp, vp, i32, vp, vp, vp, vp, vp]
    L.hive_leaf_store_update.argtypes = [vp, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp]
    L.hive_leaf_store_stats.argtypes = [vp, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]
    f32, i64 = ctypes.c_float, ctypes.c_longlong
    L.hive_nn_bn_workspace_floats.restype = i32
    L.hive_nn_bn_act_fwd.argtypes = [vp, vp, vp, vp, vp, vp, f32, f32, vp, vp, vp, vp, i64, i32, i32, vp]
    L.hive_nn_bn_act_fwd_partial.argtypes = [vp, vp, vp, vp, vp, vp, f32, f32, vp, vp, vp, vp, vp, i32, i64, i32, vp]
    L.hive_nn_bn_act_bwd.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, i32, vp]
    L.hive_nn_pack_conv3x3_weights.argtypes = [vp, i32, i32, i32, vp, vp]
    L.hive_nn_pack_conv3x3_weights_multi.argtypes = [vp, i32, i32, vp, vp, vp]
    L.hive_nn_conv3x3_wgrad.argtypes = [vp, vp, vp, i32, vp, vp]
    L.hive_nn_conv3x3_wgrad_layout.argtypes = [vp, vp, vp, i32, vp, i32, vp]
    L.hive_nn_wgrad_workspace_floats.restype = i32
    for name in ABI_SYMBOLS:
        getattr(L, name)
    _lib = L
    return L


def check(rc):
    if rc != 0:
        raise HiveError(rc, load().hive_last_error().decode())
