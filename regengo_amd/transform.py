"""Host-side mirror of the reference's streaming Transform API over the C ABI.

  stream.Transformer       /root/reference/stream/transformer.go:66-332   -> Transformer (the read loop, buffer
                                                                              compaction, MaxLeftover rule: host)
  NewTransformReader       internal/compiler/transform.go:28-93           -> Compiled.NewTransformReader
  ReplaceReader            transform.go:172-256                           -> Compiled.ReplaceReader
  SelectReader/RejectReader transform.go:322-378, 433-483                 -> Compiled.SelectReader / RejectReader

Where the bytes go: every buffer the Transformer hands to the processor is matched on the GPU.  ReplaceReader,
SelectReader(pred=None) and RejectReader(pred=None) splice on the device too (rgx_transform_chunk: FindAllBytes +
prefix-summed output offsets + gap/replacement kernels) and only the output bytes come back; arbitrary callbacks and
predicates run on the host over the span table of the buffer (FindAllBytes on the device), following the emitted
processTransform / processSelect / processReject loops.  There is no CPU matcher here: without the HIP library this
module raises.

Readers are file-like: `read(k)` returns up to k bytes and b"" at EOF, so a Transformer can be the source of the next
one (the reference's pipelines).  Context cancellation and the pooled constructor are Go-runtime concerns and not
mirrored.
"""
from __future__ import annotations

import ctypes as C
from typing import Callable, Optional

from . import _capi
from .stream import Config


def DefaultTransformConfig() -> Config:  # stream/transformer.go:39-45
    return Config(64 * 1024, 0)


class _BytesSource:
    def __init__(self, data: bytes):
        self._d, self._p = bytes(data), 0

    def read(self, k: int = -1) -> bytes:
        if k < 0:
            k = len(self._d) - self._p
        out = self._d[self._p:self._p + k]
        self._p += len(out)
        return out


def as_reader(src):
    return _BytesSource(src) if isinstance(src, (bytes, bytearray, memoryview)) else src


class ErrReader:
    """<name>TransformErrReader (transform.go:266-282): every read yields the error."""

    def __init__(self, err: Exception):
        self.err = err

    def read(self, k: int = -1) -> bytes:
        raise self.err

    def read_all(self) -> bytes:
        raise self.err


class Transformer:
    """stream.Transformer.  kind: "replace" (template), "select" / "reject" (pred or None = always true),
    "transform" (on_match(result, emit))."""

    def __init__(self, compiled, source, buffer_size: int, max_leftover: int, kind: str, template: Optional[str] = None,
                 callback: Optional[Callable] = None):
        import torch
        if buffer_size == 0:                      # newTransformer, transformer.go:131-137
            buffer_size = 64 * 1024
        if max_leftover == 0:
            max_leftover = buffer_size // 2
        self.c = compiled
        self.c._need_dev()
        if self.c.info.can_match_empty:
            raise _capi.RgxError(_capi.RGX_E_UNSUPPORTED, "Transform of a pattern that matches empty is not offered (DESIGN.md Q13)")
        self.source = as_reader(source)
        self.max_leftover = max_leftover
        self.kind = kind
        self.template = None if template is None else template.encode("utf-8")
        self.callback = callback
        self.device_splice = kind == "replace" or (kind in ("select", "reject") and callback is None)
        self.mode = {"replace": _capi.TRANSFORM_REPLACE, "select": _capi.TRANSFORM_SELECT, "reject": _capi.TRANSFORM_REJECT}.get(kind, -1)
        self.dl10 = self.c.info.default_max_leftover // 10
        self.input = bytearray(buffer_size)
        self.istart = self.iend = 0
        self.out = bytearray()
        self.ostart = 0
        self.source_eof = False
        self._torch = torch
        if self.device_splice:
            self._h_out = bytearray(buffer_size + buffer_size // 4 + 256)
        else:
            dev = "cuda:%d" % self.c._device
            self._d_in = torch.empty(buffer_size + 64, dtype=torch.uint8, device=dev)
            self._h_in = torch.empty(buffer_size, dtype=torch.uint8).pin_memory()
        self.chunks = 0
        self.matches = 0

    # ---- io.Reader side (transformer.go:187-254)
    def _take(self, n: int) -> bytes:
        if n < 0:
            n = len(self.out) - self.ostart
        b = bytes(self.out[self.ostart:self.ostart + n])
        self.ostart += len(b)
        if self.ostart == len(self.out):
            self.ostart = 0
            self.out = bytearray()
        return b

    def read(self, n: int = -1) -> bytes:
        if n < 0:
            return self.read_all()
        if n == 0:
            return b""
        stall = 0
        while self.ostart == len(self.out):
            before = (self.istart, self.iend, self.source_eof)
            if not self._process_more():
                return b""
            stall = stall + 1 if before == (self.istart, self.iend, self.source_eof) and self.ostart == len(self.out) else 0
            if stall > 2:
                raise RuntimeError("Transformer cannot advance: the buffer is full, nothing was processed and MaxLeftover "
                                   "does not release anything (the reference spins here); use a larger BufferSize")
        return self._take(n)

    def read_all(self) -> bytes:
        parts = []
        while True:
            b = self.read(1 << 20)
            if not b:
                return b"".join(parts)
            parts.append(b)

    # ---- processMore (transformer.go:258-322)
    def _process_more(self) -> bool:
        if self.source_eof and self.istart >= self.iend:
            return False
        if self.istart > 0:
            rem = self.iend - self.istart
            self.input[:rem] = self.input[self.istart:self.iend]
            self.istart, self.iend = 0, rem
        if not self.source_eof:
            room = len(self.input) - self.iend
            if room > 0:                         # Go: Read into a zero-length slice is (0, nil)
                data = self.source.read(room)
                if len(data) == 0:
                    self.source_eof = True
                self.input[self.iend:self.iend + len(data)] = data
                self.iend += len(data)
        if self.iend == 0:
            return False
        n = self.iend
        self.chunks += 1
        self.istart += self._process(n, self.source_eof)
        if not self.source_eof:
            leftover = self.iend - self.istart
            if leftover > self.max_leftover and self.max_leftover >= 0:
                excess = leftover - self.max_leftover
                self.out += self.input[self.istart:self.istart + excess]
                self.istart += excess
        return True

    # ---- the processor: one buffer
    def _upload(self, n: int):
        torch = self._torch
        self._h_in[:n] = torch.frombuffer(self.input, dtype=torch.uint8)[:n]
        self._d_in[:n].copy_(self._h_in[:n], non_blocking=True)
        torch.cuda.current_stream(self._d_in.device).synchronize()

    def _process(self, n: int, is_eof: bool) -> int:
        if self.device_splice:
            return self._process_device(n, is_eof)
        self._upload(n)
        return self._process_host(n, is_eof)

    def _process_device(self, n: int, is_eof: bool) -> int:
        """One rgx_transform_chunk call: the buffer goes down, the processor's output bytes come back."""
        lib = self.c._lib
        need, done, res = C.c_int64(0), C.c_int64(0), _capi.Result()
        tb = self.template or b""
        cin = (C.c_uint8 * n).from_buffer(self.input)
        try:
            for _ in range(2):
                cout = (C.c_uint8 * len(self._h_out)).from_buffer(self._h_out)
                w = lib.rgx_transform_chunk(self.c._h, self.c._ctx, cin, n, 1 if is_eof else 0, self.mode, tb, len(tb), cout,
                                            len(self._h_out), C.byref(need), C.byref(done), C.byref(res))
                del cout
                if w == _capi.RGX_E_CAPACITY:
                    self._h_out = bytearray(int(need.value) + 256)
                    continue
                break
        finally:
            del cin
        _capi.check(w)
        self.matches += int(res.total)
        if w > 0:
            self.out += memoryview(self._h_out)[:w]
        return int(done.value)

    def _process_host(self, n: int, is_eof: bool) -> int:
        """processTransform / processSelect / processReject over the buffer's span table (transform.go:96-170, 380-431,
        485-571); matches in their true context (what FindAllBytes reports)."""
        data = bytes(self.input[:n])
        if self.c.info.ref_find_engine == 1 and not self.c.stdlib:
            # a Tagged-DFA program in reference mode: the processor's own loop (its matches are the engine's, longest-on-path) with ONE
            # result struct per call (transform.go:123) -- a group the engine leaves alone ((-1, -1)) keeps what the struct held
            recs = self.c._loop_rows(data)
            held = [0] * self.c.ncap
            for rec in recs:
                for g in range(1, self.c.ncap // 2):
                    if rec[2 * g] >= 0:
                        held[2 * g], held[2 * g + 1] = rec[2 * g], rec[2 * g + 1]
                    else:
                        rec[2 * g], rec[2 * g + 1] = held[2 * g], held[2 * g + 1]
        else:
            spans, _ = self.c.FindAllSpans(self._d_in[:n])
            recs = spans.cpu().tolist()
        emit = self.out.extend
        processed = 0
        self.matches += len(recs)
        for rec in recs:
            ms, me = rec[0], rec[1]
            if self.kind != "select" and ms > processed:
                emit(data[processed:ms])
            result = self.c._make_result(data, rec)
            if self.kind == "transform":
                self.callback(result, emit)
            elif self.kind == "select":
                if self.callback(result):
                    emit(data[ms:me])
            elif not self.callback(result):
                emit(data[ms:me])
            processed = me
        if is_eof:
            if self.kind != "select" and processed < n:
                emit(data[processed:])
            return n
        if self.kind == "select":
            return processed
        safe = max(n - self.dl10, processed)
        if safe > processed:
            emit(data[processed:safe])
        return safe


# ---- the generated methods (bound onto api.Compiled)
def NewTransformReader(self, r, cfg: Config, on_match) -> Transformer:  # transform.go:28-93
    bs = cfg.BufferSize or 64 * 1024
    ml = cfg.MaxLeftover or self.info.default_max_leftover
    return Transformer(self, r, bs, ml, "transform", callback=on_match)


def ReplaceReader(self, r, template: str, cfg: Optional[Config] = None):  # transform.go:172-256
    """cfg is an extension (the reference always uses DefaultTransformConfig here): a larger BufferSize is what makes
    the GPU worth calling."""
    tb = template.encode("utf-8")
    rc = self._lib.rgx_transform_template_check(self._h, tb, len(tb))
    if rc < 0:
        return ErrReader(_capi.RgxError(rc, (self._lib.rgx_last_error() or b"").decode("utf-8", "replace")))
    cfg = cfg or DefaultTransformConfig()
    bs = cfg.BufferSize or 64 * 1024
    ml = cfg.MaxLeftover or self.info.default_max_leftover
    return Transformer(self, r, bs, ml, "replace", template=template)


def SelectReader(self, r, pred=None, cfg: Optional[Config] = None) -> Transformer:  # transform.go:322-378
    """pred(result) -> bool; None = keep every match (spliced on the device)."""
    cfg = cfg or DefaultTransformConfig()
    return Transformer(self, r, cfg.BufferSize or 64 * 1024, self.info.default_max_leftover, "select", callback=pred)


def RejectReader(self, r, pred=None, cfg: Optional[Config] = None) -> Transformer:  # transform.go:433-483
    cfg = cfg or DefaultTransformConfig()
    return Transformer(self, r, cfg.BufferSize or 64 * 1024, self.info.default_max_leftover, "reject", callback=pred)
