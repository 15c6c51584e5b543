"""Python mirror of the sharded entry points (include/rgx.h: rgx_sharded_*), for the tests and bench.py: the multi-GPU FindReader /
FindAllBytes lives in the C library (csrc/rgx_sharded.hip) -- program copies, contexts, host threads and the RCCL communicator are
the library's; torch only lends device memory here."""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence

from . import _capi


class Sharded:
    """One process, several devices: Sharded(compiled_or_blob, devices=[0, 1, ...]).
    One process per device (torch.distributed.run): Sharded(compiled_or_blob, device=d, rank=r, world=w, uid=bytes128)."""

    def __init__(self, program, devices: Optional[Sequence[int]] = None, device: Optional[int] = None, rank: Optional[int] = None,
                 world: Optional[int] = None, uid: Optional[bytes] = None):
        self._lib = _capi.lib()
        blob = program if isinstance(program, (bytes, bytearray)) else program.blob()
        self._blob = bytes(blob)
        h = C.c_void_p()
        if rank is None:
            devs = (C.c_int * len(devices))(*devices)
            _capi.check(self._lib.rgx_sharded_create(self._blob, len(self._blob), devs, len(devices), C.byref(h)))
        else:
            u = bytes(uid) if uid is not None else None
            _capi.check(self._lib.rgx_sharded_create_rank(self._blob, len(self._blob), device, rank, world, u, len(u) if u else 0, C.byref(h)))
        self._h = h
        info = _capi.ShardedInfo()
        _capi.check(self._lib.rgx_sharded_shape(self._h, C.byref(info)))
        self.n_local, self.world, self.first_rank, self.uses_rccl = info.n_local, info.world, info.first_rank, bool(info.uses_rccl)
        import os
        # what the world's exchange runs over (rgx_sharded_shape.uses_rccl + the library named by RGX_SHARDED_CCL_LIB, if any)
        self.communicator = ("none (one rank, or logical shards of one process)" if not self.uses_rccl else
                             "test double %s" % os.path.basename(os.environ["RGX_SHARDED_CCL_LIB"]) if os.environ.get("RGX_SHARDED_CCL_LIB") else "librccl")
        self.info = _capi.Info()
        _capi.check(self._lib.rgx_program_info(self._lib.rgx_sharded_program(self._h, 0), C.byref(self.info)))
        self.ncap = self.info.ncap
        self._keep = [None, None]       # the buffers of the rounds in flight (host bytes must outlive their round)
        self._n_sub = 0

    @staticmethod
    def unique_id() -> bytes:
        b = C.create_string_buffer(128)
        n = _capi.check(_capi.lib().rgx_sharded_unique_id(b, 128))
        return b.raw[:n]

    def close(self):
        if getattr(self, "_h", None):
            self._lib.rgx_sharded_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_timing(self, on: bool = True):
        self._lib.rgx_sharded_set_timing(self._h, 1 if on else 0)

    def plan(self, total_len: int, parts: Optional[int] = None, halo_left: int = 4096):
        parts = parts or self.world
        out = (_capi.ShardRange * parts)()
        _capi.check(self._lib.rgx_shard_plan(total_len, parts, self.info.max_match_len, halo_left, 0, out))
        return [(o.lo, o.hi, o.win_lo, o.win_hi) for o in out]

    def _windows(self, windows):
        """windows: one entry per LOCAL shard: None, or a dict(buf=uint8 tensor on that device | bytes, own=(lo, hi), base=int,
        starts_at_sync=bool, last=bool, starts_only=bool, reader=(BufferSize, MaxLeftover) | None, out=int32 tensor [cap, ncap] ([cap] with
        starts_only) on that device | None)."""
        import torch
        arr = (_capi.ShardWindow * self.n_local)()
        keep = []
        for i, w in enumerate(windows):
            if w is None:
                continue
            buf = w["buf"]
            if isinstance(buf, torch.Tensor):
                arr[i].buf, arr[i].len, arr[i].is_host = buf.data_ptr(), buf.numel(), 0 if buf.is_cuda else 1
            else:
                buf = bytes(buf)
                arr[i].buf, arr[i].len, arr[i].is_host = C.cast(C.c_char_p(buf), C.c_void_p).value, len(buf), 1
            keep.append(buf)
            lo, hi = w.get("own", (0, arr[i].len))
            arr[i].own_lo, arr[i].own_hi, arr[i].base = int(lo), int(hi), int(w.get("base", 0))
            arr[i].starts_at_sync = 1 if w.get("starts_at_sync", False) else 0
            arr[i].last = 1 if w.get("last", False) else 0
            arr[i].starts_only = 1 if w.get("starts_only", False) else 0      # rows = int32 match starts (fixed-template programs)
            if w.get("reader"):                                               # (BufferSize, MaxLeftover): the window is a run of FindReader chunks
                arr[i].reader_buffer_size, arr[i].reader_max_leftover = int(w["reader"][0]), int(w["reader"][1])
            out = w.get("out")
            if out is not None:
                arr[i].d_spans, arr[i].cap_records = out.data_ptr(), out.shape[0]
                keep.append(out)
        return arr, keep

    def submit(self, windows, count_only: bool = False) -> int:
        arr, keep = self._windows(windows)
        slot = _capi.check(self._lib.rgx_sharded_round_submit(self._h, arr, 1 if count_only else 0))
        self._keep[slot] = (arr, keep)
        return slot

    def wait(self, stop: bool = False):
        out = (_capi.ShardRound * self.world)()
        total = _capi.check(self._lib.rgx_sharded_round_wait(self._h, 1 if stop else 0, out))
        return int(total), [dict(count=o.count, have=bool(o.have), unsynced=bool(o.unsynced), truncated=bool(o.truncated),
                                 stop=bool(o.stop), status=o.status, kernel_ms=o.kernel_ms) for o in out]

    def round(self, windows, count_only: bool = False, stop: bool = False):
        self.submit(windows, count_only)
        return self.wait(stop)

    def rows_ptr(self, local_index: int):
        p, base = C.c_void_p(), C.c_int64()
        n = _capi.check(self._lib.rgx_sharded_rows(self._h, local_index, C.byref(p), C.byref(base)))
        return int(n), p.value, base.value

    def gather(self, dst_rank: int = 0, out=None, host: bool = False):
        """out: int64 tensor [cap, ncap] on dst's device (rows land there) -> returns the row count; host=True: returns a numpy
        int64 array [n, ncap] (the host copy a Go caller would take)."""
        import numpy as np
        p = C.c_void_p()
        if host:
            cap = sum(self._last_counts) if getattr(self, "_last_counts", None) else 0
            h = np.empty((max(cap, 1), self.ncap), dtype=np.int64)
            n = _capi.check(self._lib.rgx_sharded_gather(self._h, dst_rank, None, h.ctypes.data, h.shape[0], C.byref(p)))
            return h[:n]
        if out is not None:
            return int(_capi.check(self._lib.rgx_sharded_gather(self._h, dst_rank, out.data_ptr(), None, out.shape[0], C.byref(p))))
        return int(_capi.check(self._lib.rgx_sharded_gather(self._h, dst_rank, None, None, 0, C.byref(p))))

    def gather_offsets(self, dst_rank: int = 0, out=None, host: bool = False):
        """The compact form (rgx_sharded_gather_offsets): one uint64 per match, bits [0, 40) the stream-absolute start, bits [40, 64)
        the length.  out: int64/uint64 tensor [cap] on dst's device -> the match count; host=True: a numpy uint64 array [n]."""
        import numpy as np
        p = C.c_void_p()
        if host:
            cap = sum(self._last_counts) if getattr(self, "_last_counts", None) else 0
            h = np.empty(max(cap, 1), dtype=np.uint64)
            n = _capi.check(self._lib.rgx_sharded_gather_offsets(self._h, dst_rank, None, h.ctypes.data, h.shape[0], C.byref(p)))
            return h[:n]
        if out is not None:
            return int(_capi.check(self._lib.rgx_sharded_gather_offsets(self._h, dst_rank, out.data_ptr(), None, out.shape[0], C.byref(p))))
        return int(_capi.check(self._lib.rgx_sharded_gather_offsets(self._h, dst_rank, None, None, 0, C.byref(p))))

    @staticmethod
    def split_offsets(words):
        """(start, end) int64 arrays / tensors of gather_offsets' words."""
        mask = (1 << 40) - 1
        start = words & mask
        return start, start + (words >> 40)

    def round_counts(self, windows, **kw):
        total, rs = self.round(windows, **kw)
        self._last_counts = [r["count"] for r in rs]
        return total, rs

    def find_all_bytes(self, data: bytes, n: int = -1, capacity: Optional[int] = None):
        """rgx_sharded_find_all_bytes: host bytes in, numpy int32 rows [count, ncap] (buffer-absolute) out."""
        import numpy as np
        data = bytes(data)
        res = _capi.Result()
        cap = capacity if capacity is not None else len(data) // max(self.info.min_match_len, 1) + 1
        out = np.empty((max(cap, 1), self.ncap), dtype=np.int32)
        w = _capi.check(self._lib.rgx_sharded_find_all_bytes(self._h, data, len(data), n, out.ctypes.data, cap, C.byref(res)))
        return out[:w], res
