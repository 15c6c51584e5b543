"""Host-side mirror of the reference's generated `Compiled<Name>` API over the C ABI.

The reference emits, per pattern, a zero-size struct with ~40 methods (README.md:99-146).  This module keeps
the names, argument meaning and error behaviour of the hot-path ones -- MatchBytes, FindBytes, FindAllBytes,
FindReader, FindReaderCount, FindReaderFirst, MatchLengthInfo, DefaultMaxLeftover, ReplaceAllBytes, ReplaceReader,
SelectReader, RejectReader, NewTransformReader -- so the parity tests read
like the reference's own generated tests (internal/compiler/test_gen.go:72-239).  PyTorch is used for device
memory only; the matching runs in librgx_hip.so.
"""
from __future__ import annotations

import ctypes as C
from typing import Callable, List, Optional, Sequence

from . import _capi
from . import transform as _transform
from .stream import Config, ErrBufferTooSmall, Match


def upper_first(s: str) -> str:  # internal/codegen/constants.go:34-40
    return s[:1].upper() + s[1:] if s else s


def field_names(cap_names: Sequence[str]) -> List[str]:
    """Result-struct field names, captures.go:63-76: UpperFirst(name) | Group<i>, collision -> suffix <i>."""
    used = {"Match"}
    out = ["Match"]
    for i in range(1, len(cap_names)):
        f = upper_first(cap_names[i]) if cap_names[i] else "Group%d" % i
        if f in used:
            f = "%s%d" % (f, i)
        used.add(f)
        out.append(f)
    return out


class BytesResult:
    """`<Name>BytesResult` (captures.go:45-118): one `[]byte` per group, aliasing the input."""
    __slots__ = ("_fields", "_vals", "spans")

    def __init__(self, fields, vals, spans):
        self._fields = fields
        self._vals = vals
        self.spans = spans

    def __getattr__(self, k):
        try:
            return self._vals[self._fields.index(k)]
        except ValueError:
            raise AttributeError(k)

    def CaptureByIndex(self, idx: int):
        return self._vals[idx] if 0 <= idx < len(self._vals) else None

    def __repr__(self):
        return "BytesResult(%s)" % ", ".join("%s=%r" % kv for kv in zip(self._fields, self._vals))


class Compiled:
    def __init__(self, pattern: str, name: str = "Pattern", flags: int = 0, device: Optional[int] = None, stdlib: bool = False,
                 force_tdfa: bool = False, no_prefilter_scan: bool = False):
        """stdlib=False (the default): MatchBytes / FindBytes / FindBatch behave like the reference's emitted functions,
        restart rule included (a failed attempt resumes behind its failure offset, stepping over some matches);
        stdlib=True: the plain leftmost-first search (RGX_FLAG_STDLIB_SEMANTICS).  FindAllBytes is the same in both."""
        self._lib = _capi.lib()
        self.pattern = pattern
        self.name = name
        if stdlib:
            flags |= _capi.FLAG_STDLIB_SEMANTICS
        if no_prefilter_scan:   # FindAll on the program's other scan kernel from the first call (RGX_FLAG_NO_PREFILTER_SCAN)
            flags |= _capi.FLAG_NO_PREFILTER_SCAN
        if force_tdfa:          # regengo.Options.ForceTDFA: the reference's Tagged DFA for the capture functions whenever it can be built
            flags |= _capi.FLAG_FORCE_TDFA
        self.stdlib = bool(flags & _capi.FLAG_STDLIB_SEMANTICS)
        h = C.c_void_p()
        _capi.check(self._lib.rgx_compile(pattern.encode("utf-8"), flags, C.byref(h)))
        self._h = h
        self._ctx = None
        self._ctx_owner = None      # another Compiled whose context this one shares (to(ctx_of=...))
        self._ctx_shared = False
        self._device = None
        info = _capi.Info()
        _capi.check(self._lib.rgx_program_info(self._h, C.byref(info)))
        self.info = info
        n = self._lib.rgx_program_capture_names(self._h, None, 0)
        buf = C.create_string_buffer(int(n) + 1)
        self._lib.rgx_program_capture_names(self._h, buf, n)
        self.capture_names = buf.raw[:n].decode("utf-8").split("\0")[:-1]
        self.fields = field_names(self.capture_names)
        self.ncap = info.ncap
        self.MinMatchLen = info.min_match_len
        self.MaxMatchLen = info.max_match_len
        if device is not None:
            self.to(device)

    # ---- lifecycle
    def __del__(self):
        try:
            if getattr(self, "_ctx", None) and getattr(self, "_ctx_owner", None) is None:
                self._lib.rgx_stream_ctx_destroy(self._ctx)
            if getattr(self, "_h", None):
                self._lib.rgx_program_destroy(self._h)
        except Exception:
            pass

    def blob(self) -> bytes:
        n = _capi.check(self._lib.rgx_program_blob_size(self._h))
        b = C.create_string_buffer(n)
        _capi.check(self._lib.rgx_program_blob_write(self._h, b, n))
        return b.raw

    def reset_bytes(self) -> bytes:
        b = C.create_string_buffer(256)
        _capi.check(self._lib.rgx_program_reset_bytes(self._h, b))
        return b.raw

    def to(self, device: int = 0, ctx_of: Optional["Compiled"] = None) -> "Compiled":
        """ctx_of: share that object's context (stream + device scratch) instead of creating one -- many patterns over the same
        large buffers then need the scratch once (rgx_stream_ctx_rebind before every call; one call at a time)."""
        _capi.check(self._lib.rgx_program_to_device(self._h, device))
        if ctx_of is not None and self._ctx is None:
            ctx_of._need_dev()
            self._ctx, self._ctx_owner, self._ctx_shared = ctx_of._ctx, ctx_of, True
            ctx_of._ctx_shared = True
        if self._ctx is None:
            # the context runs on torch's CURRENT stream of that device: scans are ordered with the torch kernels and copies that
            # produce their inputs and reuse their outputs (a private stream would race with them)
            import torch
            st = torch.cuda.current_stream(device).cuda_stream
            c = C.c_void_p()
            _capi.check(self._lib.rgx_stream_ctx_create_on_stream(self._h, C.c_void_p(st), 1, C.byref(c)))
            self._ctx = c
        self._device = device
        _capi.check(self._lib.rgx_program_info(self._h, C.byref(self.info)))     # table_bytes / scan_kernel are known now
        return self

    def tuning(self) -> dict:
        """what the program has learned about its texts so far (rgx_program_tuning)"""
        t = _capi.Tuning()
        _capi.check(self._lib.rgx_program_tuning(self._h, C.byref(t)))
        return {n: int(getattr(t, n)) for n, _ in t._fields_ if n != "reserved"}

    def freeze(self) -> "Compiled":
        """end the learning: the program is immutable from here on (rgx_program_freeze)"""
        _capi.check(self._lib.rgx_program_freeze(self._h))
        return self

    def set_timing(self, on: bool = True):
        self._need_dev()
        self._lib.rgx_stream_ctx_set_timing(self._ctx, 1 if on else 0)

    def _need_dev(self):
        if self._ctx is None:
            self.to(0)
        if self._ctx_shared:
            _capi.check(self._lib.rgx_stream_ctx_rebind(self._ctx, self._h))

    # ---- generated-API surface
    def MatchLengthInfo(self):
        return self.MinMatchLen, self.MaxMatchLen

    def DefaultMaxLeftover(self) -> int:
        return self.info.default_max_leftover

    def _as_device(self, data):
        """bytes-like or torch tensor -> (tensor on the program's device, length)."""
        import torch
        if isinstance(data, torch.Tensor):
            t = data
            if t.dtype != torch.uint8:
                raise TypeError("input tensor must be uint8")
            if not t.is_cuda:
                t = t.to("cuda:%d" % self._device)
            return t.contiguous(), t.numel()
        b = bytes(data)
        t = torch.frombuffer(bytearray(b), dtype=torch.uint8) if len(b) else torch.empty(0, dtype=torch.uint8)
        return t.to("cuda:%d" % self._device), len(b)

    def MatchBytes(self, data) -> bool:
        """compiler.go:740-871.  Host bytes go through rgx_match_bytes (what the cgo stub calls), device tensors through
        rgx_match_bytes_device."""
        self._need_dev()
        m = C.c_int(0)
        if isinstance(data, (bytes, bytearray, memoryview)):
            b = bytes(data)
            _capi.check(self._lib.rgx_match_bytes(self._h, self._ctx, b, len(b), C.byref(m)))
            return bool(m.value)
        t, n = self._as_device(data)
        _capi.check(self._lib.rgx_match_bytes_device(self._h, self._ctx, t.data_ptr() if n else None, n, C.byref(m)))
        return bool(m.value)

    MatchString = MatchBytes

    def FindAllSpans(self, data, n: int = -1, capacity: Optional[int] = None, out=None, own=None):
        """Flat accessor next to the drop-in slice-of-structs: int32 tensor [count, ncap] on the device.
        Returns (spans, result).  own=(lo, hi): shard mode -- the FindAll chain runs over the whole buffer, only matches
        whose start lies in [lo, hi) are reported (rgx_find_all_bytes_device_owned)."""
        import torch
        self._need_dev()
        t, ln = self._as_device(data)
        res = _capi.Result()
        if n == 0 or ln == 0:
            return torch.empty((0, self.ncap), dtype=torch.int32, device=t.device), res
        if capacity is None:
            capacity = ln // max(self.MinMatchLen, 1) + 1
            if n > 0:
                capacity = min(capacity, n)
        if out is None or out.numel() < capacity * self.ncap:
            out = torch.empty((capacity, self.ncap), dtype=torch.int32, device=t.device)
        if own is None:
            w = self._lib.rgx_find_all_bytes_device(self._h, self._ctx, t.data_ptr(), ln, n, out.data_ptr(), capacity,
                                                    C.byref(res))
        else:
            w = self._lib.rgx_find_all_bytes_device_owned(self._h, self._ctx, t.data_ptr(), ln, n, out.data_ptr(), capacity,
                                                          int(own[0]), int(own[1]), C.byref(res))
        _capi.check(w)
        return out.view(-1, self.ncap)[:w], res

    # ---- the same scan, asynchronously (rgx_find_all_submit / rgx_find_all_wait): at most two in flight
    def FindAllSubmit(self, data, n: int = -1, capacity: Optional[int] = None, out=None, own=None) -> None:
        """Queue the scan of `data` (a device uint8 tensor; it and `out` must stay untouched until FindAllWait returns
        it).  Patterns / buffers the asynchronous launch is not offered for are scanned right here, synchronously; the
        result is handed out by FindAllWait in order all the same."""
        import torch
        self._need_dev()
        if not hasattr(self, "_pending"):
            self._pending = []
        t, ln = self._as_device(data)
        if capacity is None:
            capacity = ln // max(self.MinMatchLen, 1) + 1
            if n > 0:
                capacity = min(capacity, n)
        if out is None or out.numel() < capacity * self.ncap:
            out = torch.empty((max(capacity, 1), self.ncap), dtype=torch.int32, device=t.device)
        lo, hi = (int(own[0]), int(own[1])) if own is not None else (0, -1)
        rc = self._lib.rgx_find_all_submit(self._h, self._ctx, t.data_ptr() if ln else None, ln, n, out.data_ptr(), capacity, lo, hi)
        if rc == _capi.RGX_E_UNSUPPORTED:
            self._pending.append(("done", self.FindAllSpans(t, n, capacity, out, own)))
            return
        _capi.check(rc)
        self._pending.append(("queued", (t, out)))

    def FindAllWait(self):
        """(spans, result) of the oldest FindAllSubmit."""
        kind, payload = self._pending.pop(0)
        if kind == "done":
            return payload
        t, out = payload
        res = _capi.Result()
        w = _capi.check(self._lib.rgx_find_all_wait(self._h, self._ctx, C.byref(res)))
        return out.view(-1, self.ncap)[:w], res

    def capture_template(self):
        """(offsets[ncap], match_len) for fixed-template patterns: span slot c = start + offsets[c]."""
        o = (C.c_int32 * self.ncap)()
        k = _capi.check(self._lib.rgx_program_capture_template(self._h, o))
        return list(o), k

    def FindAllStarts(self, data, n: int = -1, capacity: Optional[int] = None, out=None):
        """Compact result for fixed-template patterns: int32 tensor [count] of match starts (device)."""
        import torch
        self._need_dev()
        t, ln = self._as_device(data)
        res = _capi.Result()
        if n == 0 or ln == 0:
            return torch.empty(0, dtype=torch.int32, device=t.device), res
        if capacity is None:
            capacity = ln // max(self.MinMatchLen, 1) + 1
        if out is None or out.numel() < capacity:
            out = torch.empty(capacity, dtype=torch.int32, device=t.device)
        w = _capi.check(self._lib.rgx_find_all_starts_device(self._h, self._ctx, t.data_ptr(), ln, n, out.data_ptr(), capacity,
                                                             C.byref(res)))
        return out[:w], res

    def CountAll(self, data):
        self._need_dev()
        t, ln = self._as_device(data)
        res = _capi.Result()
        w = _capi.check(self._lib.rgx_count_all_device(self._h, self._ctx, t.data_ptr() if ln else None, ln, C.byref(res)))
        return int(w), res

    def CountAllOwned(self, data, own) -> int:
        """Shard mode of CountAll: matches whose start lies in own=(lo, hi) of the window (rgx_count_all_device_owned)."""
        self._need_dev()
        t, ln = self._as_device(data)
        return int(_capi.check(self._lib.rgx_count_all_device_owned(self._h, self._ctx, t.data_ptr() if ln else None, ln,
                                                                    int(own[0]), int(own[1]), None)))

    def _make_result(self, src: bytes, rec) -> BytesResult:
        vals = []
        for g in range(self.ncap // 2):
            a, b = int(rec[2 * g]), int(rec[2 * g + 1])
            # find.go:394-406: `if captures[a] <= captures[b] && captures[b] <= len(input) { input[a:b] } else { nil }`
            vals.append(src[a:b] if 0 <= a <= b <= len(src) else None)
        return BytesResult(self.fields, vals, [int(x) for x in rec])

    def FindAllBytes(self, data, n: int = -1) -> List[BytesResult]:
        src = bytes(data) if not hasattr(data, "is_cuda") else bytes(data.cpu().numpy().tobytes())
        spans, _ = self.FindAllSpans(data, n)
        return [self._make_result(src, r) for r in spans.cpu().tolist()]

    def FindBytes(self, data):
        """(result, ok): FindBytes / FindBytesReuse, find.go:469-591, through the host-buffer entry point rgx_find_bytes."""
        self._need_dev()
        b = bytes(data)
        spans = (C.c_int32 * self.ncap)()
        f = C.c_int(0)
        _capi.check(self._lib.rgx_find_bytes(self._h, self._ctx, b, len(b), spans, C.byref(f)))
        return (self._make_result(b, list(spans)), True) if f.value else (None, False)

    # ---- Replace path (replace.go:205-363; template syntax replace/template.go)
    def ReplaceAllDevice(self, data, template: str, first_only: bool = False):
        """data: bytes or a device uint8 tensor.  Returns (device uint8 tensor with the result, matches replaced).
        A malformed template raises RgxError(RGX_E_INVALID) -- the reference panics (replace.go:215-217)."""
        import torch
        self._need_dev()
        t, ln = self._as_device(data)
        tb = template.encode("utf-8")
        res = _capi.Result()
        need = C.c_int64(0)
        cap = ln + max(64, ln // 8)
        out = torch.empty(max(cap, 1), dtype=torch.uint8, device=t.device)
        for _ in range(2):
            w = self._lib.rgx_replace_all_bytes_device(self._h, self._ctx, t.data_ptr() if ln else None, ln, tb, len(tb),
                                                       1 if first_only else 0, out.data_ptr(), out.numel(), C.byref(need), C.byref(res))
            if w == _capi.RGX_E_CAPACITY:
                out = torch.empty(max(int(need.value), 1), dtype=torch.uint8, device=t.device)
                continue
            break
        _capi.check(w)
        return out[:int(need.value)], int(res.total)

    def _replace_host(self, data, template: str, first_only: bool) -> bytes:
        """Host buffers in and out: rgx_replace_all_bytes, what the generated ReplaceAllBytesAppend / ReplaceFirstBytes call."""
        self._need_dev()
        b = bytes(data)
        tb = template.encode("utf-8")
        need = C.c_int64(0)
        out = C.create_string_buffer(len(b) + max(64, len(b) // 8))
        for _ in range(2):
            w = self._lib.rgx_replace_all_bytes(self._h, self._ctx, b, len(b), tb, len(tb), 1 if first_only else 0, out, len(out),
                                                C.byref(need), None)
            if w == _capi.RGX_E_CAPACITY:
                out = C.create_string_buffer(max(int(need.value), 1))
                continue
            break
        _capi.check(w)
        return out.raw[:int(w)]

    def ReplaceAllBytes(self, data, template: str) -> bytes:
        if isinstance(data, (bytes, bytearray, memoryview)):
            return self._replace_host(data, template, False)
        out, _ = self.ReplaceAllDevice(data, template)
        return bytes(out.cpu().numpy().tobytes())

    def ReplaceFirstBytes(self, data, template: str) -> bytes:
        if isinstance(data, (bytes, bytearray, memoryview)):
            return self._replace_host(data, template, True)
        out, _ = self.ReplaceAllDevice(data, template, first_only=True)
        return bytes(out.cpu().numpy().tobytes())

    def FindBatch(self, strings: Sequence[bytes]):
        import torch
        self._need_dev()
        offs = [0]
        for s in strings:
            offs.append(offs[-1] + len(s))
        dev = "cuda:%d" % self._device
        concat = torch.frombuffer(bytearray(b"".join(strings) or b"\0"), dtype=torch.uint8).to(dev)
        offsets = torch.tensor(offs, dtype=torch.int64, device=dev)
        found, spans = self.FindBatchDevice(concat, offsets)
        f = found.cpu().tolist()
        sp = spans.cpu().tolist()
        return [self._make_result(strings[i], sp[i]) if f[i] else None for i in range(len(strings))]

    def FindBatchDevice(self, concat, offsets, out=None):
        """out=(found uint8 [nstr], spans int32 [nstr, ncap]): caller-owned result buffers (a steady stream of batches reuses them)."""
        import torch
        self._need_dev()
        nstr = offsets.numel() - 1
        if out is not None:
            found, spans = out
        else:
            found = torch.empty(nstr, dtype=torch.uint8, device=concat.device)
            spans = torch.empty((nstr, self.ncap), dtype=torch.int32, device=concat.device)
        _capi.check(self._lib.rgx_find_batch_device(self._h, self._ctx, concat.data_ptr(), offsets.data_ptr(), nstr,
                                                    found.data_ptr(), spans.data_ptr()))
        return found, spans

    def MatchBatchDevice(self, concat, offsets):
        import torch
        self._need_dev()
        nstr = offsets.numel() - 1
        m = torch.empty(nstr, dtype=torch.uint8, device=concat.device)
        _capi.check(self._lib.rgx_match_batch_device(self._h, self._ctx, concat.data_ptr(), offsets.data_ptr(), nstr,
                                                     m.data_ptr()))
        return m

    # ---- streaming (streaming.go:85-317)
    def _resolve(self, cfg: Config) -> Config:
        cin = _capi.StreamConfig(cfg.BufferSize, cfg.MaxLeftover)
        cout = _capi.StreamConfig()
        rc = self._lib.rgx_stream_config_resolve(self._h, C.byref(cin), C.byref(cout))
        if rc == _capi.RGX_E_BUFFER_TOO_SMALL:
            raise ErrBufferTooSmall(cfg.BufferSize, self.info.min_buffer_size)
        _capi.check(rc)
        return Config(cout.buffer_size, cout.max_leftover)

    def FindReader(self, r, cfg: Config, on_match: Callable[[Match], bool]) -> None:
        """`r.read(k)` returns up to k bytes, b"" at EOF.  Raises ErrBufferTooSmall like Config.Validate; reader
        errors propagate.  Results are only valid during the callback (stream.go:51-65)."""
        self._need_dev()
        cfg = self._resolve(cfg)
        buf = bytearray(cfg.BufferSize)
        leftover = 0
        stream_offset = 0
        chunk_index = 0
        cap = cfg.BufferSize // max(self.MinMatchLen, 1) + 2
        spans = (C.c_int32 * (cap * self.ncap))()
        committed = C.c_int64()
        keep = C.c_int64()
        res = _capi.Result()
        tdfa_reuse = self.info.ref_find_engine == 1 and not self.stdlib
        held = [None] * (self.ncap // 2)
        while True:
            data = r.read(cfg.BufferSize - leftover)
            n = len(data)
            if n == 0:
                is_full, data_len = False, leftover
                if leftover == 0:
                    return
            else:
                buf[leftover:leftover + n] = data
                data_len = leftover + n
                is_full = n == cfg.BufferSize - leftover
            cbuf = (C.c_uint8 * data_len).from_buffer(buf)
            w = _capi.check(self._lib.rgx_find_chunk(self._h, self._ctx, cbuf, data_len, 1 if is_full else 0,
                                                     cfg.MaxLeftover, spans, cap, C.byref(committed), C.byref(keep),
                                                     C.byref(res)))
            del cbuf
            chunk = bytes(buf[:data_len])
            for i in range(w):
                rec = spans[i * self.ncap:(i + 1) * self.ncap]
                if tdfa_reuse:
                    # ONE result struct for the whole stream (streaming.go:117), its fields slices of `buf`; the Tagged DFA's
                    # result construction assigns a group only when its start tag is set (tdfa.go:1031-1046; the record says
                    # (-1, -1) otherwise) -- the field then still aliases the bytes of an earlier match's group, whatever
                    # lies there by now
                    for g in range(self.ncap // 2):
                        if rec[2 * g] >= 0:
                            held[g] = (rec[2 * g], rec[2 * g + 1])
                    vals = [None if h is None else bytes(buf[h[0]:h[1]]) for h in held]
                    m = Match(BytesResult(self.fields, vals, [int(x) for x in rec]), stream_offset + rec[0], chunk_index)
                else:
                    m = Match(self._make_result(chunk, rec), stream_offset + rec[0], chunk_index)
                if not on_match(m):
                    return
            if n == 0:
                return
            if is_full:
                k = keep.value
                leftover = data_len - k
                stream_offset += k
                buf[:leftover] = buf[k:data_len]
            else:
                leftover = 0
            chunk_index += 1

    # ---- FindReader over a run of chunks (rgx_find_chunks_device): the fixed-stride grid a buffer-filling reader produces
    def FindChunksDevice(self, data, cfg: Config, final: bool = True, capacity: Optional[int] = None, out=None, count_only: bool = False):
        """data: uint8 tensor on the program's device holding the stream from a chunk start on; cfg: a RESOLVED Config.  Returns
        (rows int32 tensor [n, ncap] block-relative | None, ChunksResult)."""
        import torch
        self._need_dev()
        res = _capi.ChunksResult()
        if count_only:
            _capi.check(self._lib.rgx_find_chunks_device(self._h, self._ctx, data.data_ptr(), data.numel(), cfg.BufferSize, cfg.MaxLeftover,
                                                         1 if final else 0, None, 0, C.byref(res)))
            return None, res
        cap = capacity if capacity is not None else data.numel() // max(self.MinMatchLen, 1) + 2
        for _ in range(2):
            if out is None or out.shape[0] < cap:
                out = torch.empty((cap, self.ncap), dtype=torch.int32, device=data.device)
            w = self._lib.rgx_find_chunks_device(self._h, self._ctx, data.data_ptr(), data.numel(), cfg.BufferSize, cfg.MaxLeftover,
                                                 1 if final else 0, out.data_ptr(), out.shape[0], C.byref(res))
            if w == _capi.RGX_E_CAPACITY and res.rows > out.shape[0]:
                cap = int(res.rows) + 16
                out = None
                continue
            break
        _capi.check(w)
        return out[:w], res

    def FindReaderBlocks(self, r, cfg: Config, on_match: Callable[[Match], bool], block_bytes: int = 64 << 20) -> None:
        """FindReader with the chunks of a RUN answered by one call (rgx_find_chunks).  The reads are the reference's own -- BufferSize
        bytes, then BufferSize - MaxLeftover at a time (streaming.go:123) -- appended to one block while they come back full; a
        short read, EOF or a full block ends the run.  Same callbacks as FindReader (StreamOffset, ChunkIndex, the reused struct of a
        Tagged-DFA program included); a run the library does not vouch for (RGX_E_DIVERGES) goes chunk by chunk through rgx_find_chunk."""
        self._need_dev()
        cfg = self._resolve(cfg)
        B, ML = cfg.BufferSize, cfg.MaxLeftover
        S = B - ML
        nmax = max((block_bytes - B) // S + 1, 1)              # full chunks per run
        block = bytearray((nmax - 1) * S + B)
        cap = len(block) // max(self.MinMatchLen, 1) + 2
        spans = (C.c_int32 * (cap * self.ncap))()
        res = _capi.ChunksResult()
        tdfa_reuse = self.info.ref_find_engine == 1 and not self.stdlib
        held = [None] * (self.ncap // 2)
        prev_chunk = bytes(B)      # the chunk in front of the run's first (what the reference's buffer still holds behind a short chunk)
        stream_offset = 0          # of block[0]
        chunk_index = 0            # of the run's first chunk
        leftover = 0               # bytes at the head of the block carried over from the run before
        while True:
            fill, nfull, final, eof = leftover, 0, False, False
            while nfull < nmax:
                want = B - (fill - nfull * S)
                data = r.read(want)
                n = len(data)
                if n == 0:                                      # EOF: what is left over is one more chunk, which reports everything
                    eof = True
                    final = fill - nfull * S > 0
                    break
                block[fill:fill + n] = data
                fill += n
                if n < want:                                    # a short read: this chunk is not full (streaming.go:177)
                    final = True
                    break
                nfull += 1
            if nfull == 0 and not final:
                return
            rows = self._run_rows(block, fill, B, ML, final, spans, cap, res)
            nchunks = nfull + (1 if final else 0)
            for rec in rows:
                k = min(rec[0] // S, nchunks - 1)
                cs = k * S
                clen = min(B, fill - cs)
                rel = [rec[0] - cs, rec[1] - cs]
                for g in range(1, self.ncap // 2):
                    a, b = rec[2 * g], rec[2 * g + 1]
                    rel += [a, b] if ((a == 0 and b == 0) or a < 0) else [a - cs, b - cs]
                if tdfa_reuse:
                    # ONE result struct for the stream (streaming.go:117), its fields slices of the reference's ONE buffer, which holds
                    # chunk k now: a field an earlier match set reads whatever lies at its offsets in THIS chunk -- behind a short chunk's
                    # end, what the chunk before left there
                    for g in range(self.ncap // 2):
                        if rel[2 * g] >= 0:
                            held[g] = (rel[2 * g], rel[2 * g + 1])
                    before = bytes(block[cs - S:cs - S + B]) if cs >= S else prev_chunk
                    bufnow = bytes(block[cs:cs + clen]) + before[clen:]
                    vals = [None if h is None else bufnow[h[0]:h[1]] for h in held]
                    m = Match(BytesResult(self.fields, vals, rel), stream_offset + rec[0], chunk_index + k)
                else:
                    m = Match(self._make_result(bytes(block[cs:cs + clen]), rel), stream_offset + rec[0], chunk_index + k)
                if not on_match(m):
                    return
            if eof:
                return
            if nfull > 0:
                prev_chunk = bytes(block[(nfull - 1) * S:(nfull - 1) * S + B])
            stream_offset += nfull * S
            if final:
                # behind a short read the reference zeroes leftover and does NOT advance streamOffset past the short chunk (streaming.go:241-244)
                prev_chunk = bytes(block[nfull * S:fill]) + prev_chunk[fill - nfull * S:]
                leftover = 0
                chunk_index += nfull + 1
            else:
                leftover = fill - nfull * S                     # == MaxLeftover
                block[:leftover] = block[nfull * S:fill]
                chunk_index += nfull

    def _run_rows(self, block, fill, B, ML, final, spans, cap, res):
        """rows (lists of ncap ints, block-relative) the reference's loop reports from the chunks of block[:fill]"""
        S = B - ML
        cbuf = (C.c_uint8 * fill).from_buffer(block)
        w = self._lib.rgx_find_chunks(self._h, self._ctx, cbuf, fill, B, ML, 1 if final else 0, spans, cap, C.byref(res))
        del cbuf
        if w != _capi.RGX_E_DIVERGES:
            _capi.check(w)
            return [[int(x) for x in spans[i * self.ncap:(i + 1) * self.ncap]] for i in range(w)]
        # chunk by chunk: rgx_find_chunk vouches for (or refuses) one chunk at a time
        rows = []
        kfull = (fill - B) // S + 1 if fill >= B else 0
        nchunks = kfull + (1 if final and kfull * S < fill else 0)
        committed, keep, r1 = C.c_int64(), C.c_int64(), _capi.Result()
        for k in range(nchunks):
            cs = k * S
            clen = min(B, fill - cs)
            piece = (C.c_uint8 * clen).from_buffer_copy(bytes(block[cs:cs + clen]))
            w = _capi.check(self._lib.rgx_find_chunk(self._h, self._ctx, piece, clen, 1 if k < kfull else 0, ML, spans, cap, C.byref(committed),
                                                     C.byref(keep), C.byref(r1)))
            for i in range(w):
                rec = [int(x) for x in spans[i * self.ncap:(i + 1) * self.ncap]]
                out = [rec[0] + cs, rec[1] + cs]
                for g in range(1, self.ncap // 2):
                    a, b = rec[2 * g], rec[2 * g + 1]
                    out += [a, b] if ((a == 0 and b == 0) or a < 0) else [a + cs, b + cs]
                rows.append(out)
        return rows

    def _loop_rows(self, data: bytes):
        """The rows of the emitted loop "FindBytesReuse on data[matchEnd:]" over one buffer (rgx_find_chunk on a chunk that is not full:
        nothing deferred): what the host side of NewTransformReader walks for a Tagged-DFA program, whose FindAllBytes is another loop."""
        self._need_dev()
        cap = len(data) // max(self.MinMatchLen, 1) + 2
        spans = (C.c_int32 * (cap * self.ncap))()
        committed, keep, res = C.c_int64(), C.c_int64(), _capi.Result()
        cbuf = (C.c_uint8 * max(len(data), 1)).from_buffer_copy(data or b"\0")
        w = _capi.check(self._lib.rgx_find_chunk(self._h, self._ctx, cbuf, len(data), 0, self.DefaultMaxLeftover(), spans, cap,
                                                 C.byref(committed), C.byref(keep), C.byref(res)))
        return [[int(x) for x in spans[i * self.ncap:(i + 1) * self.ncap]] for i in range(w)]

    # ---- streaming Transform (transform.go:28-571; regengo_amd/transform.py)
    NewTransformReader = _transform.NewTransformReader
    ReplaceReader = _transform.ReplaceReader
    SelectReader = _transform.SelectReader
    RejectReader = _transform.RejectReader

    def FindReaderCount(self, r, cfg: Config) -> int:
        """streaming.go:258-277: the number of callbacks FindReader would make.  Same read loop and commit/defer rule, but per
        chunk rgx_count_chunk: no span table crosses PCIe and no result struct is built."""
        self._need_dev()
        cfg = self._resolve(cfg)
        buf = bytearray(cfg.BufferSize)
        leftover = 0
        cnt = 0
        committed = C.c_int64()
        keep = C.c_int64()
        while True:
            data = r.read(cfg.BufferSize - leftover)
            n = len(data)
            if n == 0:
                is_full, data_len = False, leftover
                if leftover == 0:
                    return cnt
            else:
                buf[leftover:leftover + n] = data
                data_len = leftover + n
                is_full = n == cfg.BufferSize - leftover
            cbuf = (C.c_uint8 * data_len).from_buffer(buf)
            w = _capi.check(self._lib.rgx_count_chunk(self._h, self._ctx, cbuf, data_len, 1 if is_full else 0, cfg.MaxLeftover,
                                                      C.byref(committed), C.byref(keep), None))
            del cbuf
            cnt += int(w)
            if n == 0:
                return cnt
            if is_full:
                k = keep.value
                leftover = data_len - k
                buf[:leftover] = buf[k:data_len]
            else:
                leftover = 0

    def FindReaderFirst(self, r, cfg: Config):
        out = [None, 0]

        def cb(m):
            out[0], out[1] = m.Result, m.StreamOffset
            return False

        self.FindReader(r, cfg, cb)
        return out[0], out[1]


class Package:
    """Several compiled patterns matched against one batch of strings in ONE pass per group of programs (rgx_multi_*, include/rgx.h):
    FindBytes per string and program -- found bits, counts, optionally (start, end) of the found ones.  Programs that cannot take part
    (`accepted[i] == False`: large tables, UTF-8 screen, reference mode not offered) keep their own Compiled.FindBatchDevice."""

    def __init__(self, compiled: Sequence["Compiled"]):
        self._lib = _capi.lib()
        self.progs = list(compiled)
        for c in self.progs:
            c._need_dev()
        n = len(self.progs)
        arr = (C.c_void_p * n)(*[c._h for c in self.progs])
        acc = (C.c_uint8 * n)()
        h = C.c_void_p()
        self.launches = _capi.check(self._lib.rgx_multi_create(arr, n, acc, C.byref(h)))
        self._h = h
        self.accepted = [bool(a) for a in acc]
        self._ctx_owner = self.progs[0]

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._lib.rgx_multi_destroy(self._h)
        except Exception:
            pass

    def FindBatchBits(self, concat, offsets, want_se: bool = False):
        """-> (bits int64 [n, ceil(nstr/64)], counts int64 [n], se int32 [n, nstr, 2] or None); rows of programs that are not
        accepted are zero."""
        import torch
        nstr = offsets.numel() - 1
        n = len(self.progs)
        words = (nstr + 63) // 64
        bits = torch.zeros((n, max(words, 1)), dtype=torch.int64, device=concat.device)
        counts = torch.zeros(n, dtype=torch.int64, device=concat.device)
        se = torch.zeros((n, nstr, 2), dtype=torch.int32, device=concat.device) if want_se else None
        own = self._ctx_owner
        own._need_dev()
        _capi.check(self._lib.rgx_find_batch_multi_device(self._h, own._ctx, concat.data_ptr(), offsets.data_ptr(), nstr, bits.data_ptr(),
                                                          counts.data_ptr(), se.data_ptr() if want_se else None))
        return bits, counts, se
