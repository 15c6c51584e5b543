"""ctypes binding of the C ABI in include/rgx.h (librgx_hip.so, built in-tree by regengo_amd/build.py).

There is deliberately no fallback: if the library is missing, or a compute call is made without a usable
GPU, this raises.  The matching itself only ever runs in the HIP kernels.
"""
from __future__ import annotations

import ctypes as C
import os

from . import build as _build


class RgxError(RuntimeError):
    def __init__(self, status: int, msg: str):
        super().__init__("rgx status %d (%s): %s" % (status, _STATUS.get(status, "?"), msg))
        self.status = status


_STATUS = {0: "ok", -1: "invalid", -2: "syntax", -3: "unsupported", -4: "too large", -5: "no device", -6: "hip",
           -7: "nomem", -8: "capacity", -9: "bad blob", -10: "buffer too small", -11: "diverges from the reference"}

RGX_OK = 0
ABI_VERSION = 5            # include/rgx.h: RGX_ABI_VERSION (checked against rgx_abi_version() when the library is loaded)
RGX_E_INVALID = -1
RGX_E_SYNTAX = -2
RGX_E_UNSUPPORTED = -3
RGX_E_NO_DEVICE = -5
RGX_E_CAPACITY = -8
RGX_E_BUFFER_TOO_SMALL = -10
RGX_E_DIVERGES = -11
TRANSFORM_REPLACE, TRANSFORM_SELECT, TRANSFORM_REJECT = 0, 1, 2
FLAG_UNMATCHED_MINUS1 = 1
FLAG_STDLIB_SEMANTICS = 2
FLAG_FORCE_TDFA = 1 << 2            # regengo.Options.ForceTDFA
FLAG_NO_PREFILTER_SCAN = 1 << 3     # FindAll never takes rgx_scan_fc.hip (measurements, tests of the other scan kernels)


class Info(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "abi_version", "ncap", "min_match_len", "max_match_len", "default_max_leftover", "min_buffer_size", "n_inst",
        "n_states", "n_classes", "anchored", "fixed_captures", "can_match_empty", "ref_match_engine", "ref_find_engine",
        "lookahead_mode", "table_bytes", "needs_valid_utf8", "sync_states", "scan_kernel", "ref_match_offered", "ref_find_offered", "unicode_version", "utf8_screened",
        "ref_findall_offered", "ref_stream_offered", "ref_tdfa_states")] + [("flags", C.c_uint32), ("ref_replace_offered", C.c_int32)]


class Tuning(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("frozen", "scan_kernel_choice", "fc_us_per_gib", "other_us_per_gib", "fc_gave_up", "captures_long_rows",
                                         "sync_automaton", "exact_sync_points", "rewinding_walk", "ascii_twin", "batch_tiny_level", "batch_tdfa_wide")] + [("reserved", C.c_int32 * 4)]


class ShardRange(C.Structure):
    _fields_ = [("lo", C.c_int64), ("hi", C.c_int64), ("win_lo", C.c_int64), ("win_hi", C.c_int64)]


class ShardedInfo(C.Structure):
    _fields_ = [("n_local", C.c_int32), ("world", C.c_int32), ("first_rank", C.c_int32), ("uses_rccl", C.c_int32)]


class ShardWindow(C.Structure):
    _fields_ = [("buf", C.c_void_p), ("len", C.c_size_t), ("own_lo", C.c_int64), ("own_hi", C.c_int64), ("base", C.c_int64),
                ("is_host", C.c_int32), ("starts_at_sync", C.c_int32), ("last", C.c_int32), ("starts_only", C.c_int32),
                ("d_spans", C.c_void_p), ("cap_records", C.c_size_t),
                ("reader_buffer_size", C.c_int64), ("reader_max_leftover", C.c_int64)]


class ChunksResult(C.Structure):
    _fields_ = [("rows", C.c_int64), ("chunks", C.c_int64), ("next_from", C.c_int64), ("ncap", C.c_int32), ("mode", C.c_int32),
                ("kernel_ms", C.c_float), ("reserved", C.c_int32)]


class ShardRound(C.Structure):
    _fields_ = [("count", C.c_int64), ("have", C.c_int32), ("unsynced", C.c_int32), ("truncated", C.c_int32), ("stop", C.c_int32),
                ("status", C.c_int32), ("kernel_ms", C.c_float)]


class Result(C.Structure):
    _fields_ = [("total", C.c_int64), ("written", C.c_int64), ("ncap", C.c_int32), ("unsynced", C.c_int32),
                ("kernel_ms", C.c_float)]


class StreamConfig(C.Structure):
    _fields_ = [("buffer_size", C.c_int64), ("max_leftover", C.c_int64)]


# every symbol include/rgx.h declares; tests check the library exports each of them
SYMBOLS = {
    "rgx_compile": (C.c_int, [C.c_char_p, C.c_uint32, C.POINTER(C.c_void_p)]),
    "rgx_program_blob_size": (C.c_int64, [C.c_void_p]),
    "rgx_program_blob_write": (C.c_int64, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "rgx_program_from_blob": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "rgx_program_destroy": (None, [C.c_void_p]),
    "rgx_abi_version": (C.c_int, []),
    "rgx_program_info": (C.c_int, [C.c_void_p, C.POINTER(Info)]),
    "rgx_program_tuning": (C.c_int, [C.c_void_p, C.POINTER(Tuning)]),
    "rgx_program_freeze": (C.c_int, [C.c_void_p]),
    "rgx_program_capture_names": (C.c_int64, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "rgx_program_reset_bytes": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rgx_unicode_table": (C.c_int64, [C.c_char_p, C.c_void_p, C.c_size_t]),
    "rgx_device_count": (C.c_int, []),
    "rgx_program_to_device": (C.c_int, [C.c_void_p, C.c_int]),
    "rgx_stream_ctx_create": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "rgx_stream_ctx_create_on_stream": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]),
    "rgx_stream_ctx_rebind": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rgx_stream_ctx_destroy": (None, [C.c_void_p]),
    "rgx_stream_ctx_hip_stream": (C.c_void_p, [C.c_void_p]),
    "rgx_stream_ctx_set_timing": (C.c_int, [C.c_void_p, C.c_int]),
    "rgx_match_bytes_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_int)]),
    "rgx_find_all_bytes_device": (C.c_int64, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int64, C.c_void_p,
                                              C.c_size_t, C.POINTER(Result)]),
    "rgx_find_all_bytes_device_owned": (C.c_int64, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int64, C.c_void_p,
                                                    C.c_size_t, C.c_int64, C.c_int64, C.POINTER(Result)]),
    "rgx_find_all_submit": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int64, C.c_void_p, C.c_size_t, C.c_int64,
                                      C.c_int64]),
    "rgx_find_all_wait": (C.c_int64, [C.c_void_p, C.c_void_p, C.POINTER(Result)]),
    "rgx_find_all_bytes": (C.c_int64, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int64, C.c_void_p, C.c_size_t,
                                       C.POINTER(Result)]),
    "rgx_find_all_starts_device": (C.c_int64, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int64, C.c_void_p,
                                              C.c_size_t, C.POINTER(Result)]),
    "rgx_find_all_starts": (C.c_int64, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int64, C.c_void_p, C.c_size_t, C.POINTER(Result)]),
    "rgx_replace_all_bytes_device": (C.c_int64, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_int,
                                                 C.c_void_p, C.c_size_t, C.POINTER(C.c_int64), C.POINTER(Result)]),
    "rgx_replace_all_bytes": (C.c_int64, [C.c_void_p, C.c_void_p, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_int,
                                          C.c_void_p, C.c_size_t, C.POINTER(C.c_int64), C.POINTER(Result)]),
    "rgx_replace_template_check": (C.c_int, [C.c_char_p, C.c_size_t]),
    "rgx_transform_chunk_device": (C.c_int64, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_char_p, C.c_size_t,
                                               C.c_void_p, C.c_size_t, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(Result)]),
    "rgx_transform_chunk": (C.c_int64, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_char_p, C.c_size_t,
                                        C.c_void_p, C.c_size_t, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(Result)]),
    "rgx_transform_template_check": (C.c_int, [C.c_void_p, C.c_char_p, C.c_size_t]),
    "rgx_program_capture_template": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rgx_count_all_device": (C.c_int64, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(Result)]),
    "rgx_find_batch_device": (C.c_int64, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p,
                                          C.c_void_p]),
    "rgx_match_bytes": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_int)]),
    "rgx_find_bytes": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_int)]),
    "rgx_find_batch": (C.c_int64, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "rgx_match_batch_device": (C.c_int64, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "rgx_multi_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]),
    "rgx_multi_destroy": (None, [C.c_void_p]),
    "rgx_find_batch_multi_device": (C.c_int64, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rgx_stream_config_resolve": (C.c_int, [C.c_void_p, C.POINTER(StreamConfig), C.POINTER(StreamConfig)]),
    "rgx_find_chunk": (C.c_int64, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int64, C.c_void_p,
                                   C.c_size_t, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(Result)]),
    "rgx_find_chunks_device": (C.c_int64, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int64, C.c_int64, C.c_int, C.c_void_p, C.c_size_t,
                                           C.POINTER(ChunksResult)]),
    "rgx_find_chunks": (C.c_int64, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int64, C.c_int64, C.c_int, C.c_void_p, C.c_size_t,
                                    C.POINTER(ChunksResult)]),
    "rgx_count_chunk": (C.c_int64, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int64, C.POINTER(C.c_int64),
                                    C.POINTER(C.c_int64), C.POINTER(Result)]),
    "rgx_count_all_device_owned": (C.c_int64, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int64, C.c_int64,
                                               C.POINTER(Result)]),
    "rgx_shard_plan": (C.c_int, [C.c_int64, C.c_int, C.c_int32, C.c_int64, C.c_int64, C.POINTER(ShardRange)]),
    "rgx_sharded_create": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_void_p)]),
    "rgx_sharded_unique_id": (C.c_int, [C.c_void_p, C.c_size_t]),
    "rgx_sharded_create_rank": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "rgx_sharded_destroy": (None, [C.c_void_p]),
    "rgx_sharded_shape": (C.c_int, [C.c_void_p, C.POINTER(ShardedInfo)]),
    "rgx_sharded_program": (C.c_void_p, [C.c_void_p, C.c_int]),
    "rgx_sharded_hip_stream": (C.c_void_p, [C.c_void_p, C.c_int, C.c_int]),
    "rgx_sharded_set_timing": (C.c_int, [C.c_void_p, C.c_int]),
    "rgx_sharded_round_submit": (C.c_int, [C.c_void_p, C.POINTER(ShardWindow), C.c_int]),
    "rgx_sharded_round_wait": (C.c_int64, [C.c_void_p, C.c_int, C.POINTER(ShardRound)]),
    "rgx_sharded_round": (C.c_int64, [C.c_void_p, C.POINTER(ShardWindow), C.c_int, C.c_int, C.POINTER(ShardRound)]),
    "rgx_sharded_rows": (C.c_int64, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]),
    "rgx_sharded_gather": (C.c_int64, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "rgx_sharded_gather_offsets": (C.c_int64, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "rgx_sharded_find_all_bytes": (C.c_int64, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int64, C.c_void_p, C.c_size_t, C.POINTER(Result)]),
    "rgx_last_error": (C.c_char_p, []),
    "rgx_status_str": (C.c_char_p, [C.c_int]),
}

_lib = None


def lib():
    """Load librgx_hip.so (building it if the sources changed).  Raises if it cannot be had."""
    global _lib
    if _lib is not None:
        return _lib
    path = _build.product_lib_path()
    if os.environ.get("RGX_NO_BUILD") != "1":
        try:
            path = _build.build_product()
        except Exception as ex:
            # a stale library must not stand in for sources that no longer compile -- except where no compiler exists at all
            # (a box that only received the prebuilt library)
            import shutil
            if not os.path.exists(path) or shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc"):
                raise
            import warnings
            warnings.warn("librgx_hip.so could not be rebuilt (%s); using the prebuilt library" % ex)
    if not os.path.exists(path):
        raise RuntimeError("librgx_hip.so is missing: the HIP extension is required (no CPU fallback exists)")
    # PyTorch's wheel carries its own libamdhip64; whichever copy is loaded first serves the whole process.  Load torch's
    # first (as bench.py and the tests always did): device pointers are shared with torch tensors, and a process that
    # loaded /opt/rocm's runtime through this library first makes torch's later device initialisation fail.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(path)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    if L.rgx_abi_version() != ABI_VERSION:
        raise RuntimeError("librgx_hip.so has ABI version %d, this binding is written for %d" % (L.rgx_abi_version(), ABI_VERSION))
    _lib = L
    return L


def check(rc: int) -> int:
    if rc < 0:
        raise RgxError(int(rc), (lib().rgx_last_error() or b"").decode("utf-8", "replace"))
    return rc
