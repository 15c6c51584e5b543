"""ORACLE (test infrastructure, not product): restatement of the reference's streaming Transform path --
`stream.Transformer` (/root/reference/stream/transformer.go:66-332: Read, processMore, emitOutput) and the emitted
NewTransformReader/processTransform, ReplaceReader, SelectReader/processSelect, RejectReader/processReject
(/root/reference/internal/compiler/transform.go:28-571).  Pinned by tests/golden/transform_kats.json: the literal
vectors of tests/integration/streaming/transform_test.go and stream/transformer_test.go.

What the emitted processors do with one buffer `data` (everything read and not yet consumed):
  * loop: FindBytesReuse(data[processed:]) -> locate the match by bytes.Index (Q4') -> emit the gap, call the
    callback, advance to the match end (an EMPTY match advances one byte WITHOUT emitting it: the byte is lost,
    transform.go:158-162);
  * no further match, not EOF: processTransform/processReject pass everything up to
    safePoint = max(processed, len(data) - defaultLeftover/10) through and keep the rest; processSelect keeps
    everything after the last match;
  * the Transformer then passes any leftover beyond MaxLeftover through UNCHANGED (transformer.go:311-319) -- also
    for SelectReader, where those bytes are non-matches.
ReplaceReader resolves the template with ValidateAndResolve (a bad reference gives a reader that only returns the
error) and expands group texts through getCaptureByIndex, which knows NAMED groups only (transform.go:288-320): `$1`
of an unnamed group expands to nothing.

`quirks=True` reproduces Q1/Q4'/Q12 (see oracle/replace.py) by running the slice-relative FindBytesReuse exactly as
emitted.  `quirks=False` is the same chunk protocol over the true leftmost-first matches of `data` in their real
context -- what the GPU path computes; it is only defined for patterns that cannot match empty.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Tuple

from . import engines as E
from . import replace as R
from . import syntax as S

EOF = "EOF"
Reader = Callable[[int], Tuple[bytes, Optional[str]]]      # io.Reader: read(k) -> (up to k bytes, None | EOF | error text)


def bytes_reader(data: bytes, max_read: Optional[int] = None) -> Reader:
    """strings.Reader / bytes.Reader: (n, nil) while bytes remain (also for a zero-length p), then (0, io.EOF)."""
    pos = [0]

    def read(k: int):
        if pos[0] >= len(data):
            return b"", EOF
        if max_read is not None:
            k = min(k, max_read)
        out = data[pos[0]:pos[0] + k]
        pos[0] += len(out)
        return out, None
    return read


def transform_default_leftover(max_len: int) -> int:  # transform.go:32-42 (same rule as streaming.go:87-96)
    return E.default_max_leftover(max_len)


class Transformer:
    """stream.Transformer (transformer.go:66-332).  `processor(data, is_eof, emit) -> processed`."""

    def __init__(self, source: Reader, buffer_size: int, max_leftover: int, processor):
        if buffer_size == 0:                       # newTransformer, transformer.go:131-137
            buffer_size = 64 * 1024
        if max_leftover == 0:
            max_leftover = buffer_size // 2
        self.source = source
        self.max_leftover = max_leftover
        self.processor = processor
        self.input = bytearray(buffer_size)
        self.istart = 0
        self.iend = 0
        self.out = bytearray()
        self.ostart = 0
        self.source_eof = False
        self.err: Optional[str] = None
        self.chunks = 0

    def _emit(self, b: bytes):  # transformer.go:325-332
        self.out += b

    def _take(self, n: int) -> bytes:
        b = bytes(self.out[self.ostart:self.ostart + n])
        self.ostart += len(b)
        if self.ostart == len(self.out):
            self.ostart = 0
            self.out = bytearray()
        return b

    def Read(self, n: int) -> Tuple[bytes, Optional[str]]:  # transformer.go:187-254 (no Context here)
        if self.ostart < len(self.out):
            return self._take(n), None
        if self.err is not None:
            return b"", self.err
        stall = 0
        while self.ostart == len(self.out):
            before = (self.istart, self.iend, self.source_eof)
            err = self._process_more()
            if err is not None:
                if self.ostart < len(self.out):
                    self.err = err
                    return self._take(n), None
                return b"", err
            stall = stall + 1 if before == (self.istart, self.iend, self.source_eof) and self.ostart == len(self.out) else 0
            if stall > 2:
                raise RuntimeError("reference Transformer would spin forever (full buffer, nothing processed)")
        return self._take(n), None

    def _process_more(self) -> Optional[str]:  # transformer.go:258-322
        if self.source_eof and self.istart >= self.iend:
            return EOF
        if self.istart > 0:
            rem = self.iend - self.istart
            self.input[:rem] = self.input[self.istart:self.iend]
            self.istart, self.iend = 0, rem
        if not self.source_eof:
            data, err = self.source(len(self.input) - self.iend)
            self.input[self.iend:self.iend + len(data)] = data
            self.iend += len(data)
            if err is not None:
                if err == EOF:
                    self.source_eof = True
                else:
                    return err
        if self.iend == 0:
            return EOF
        data = bytes(self.input[self.istart:self.iend])
        self.chunks += 1
        self.istart += self.processor(data, self.source_eof, self._emit)
        if not self.source_eof:
            leftover = self.iend - self.istart
            if leftover > self.max_leftover and self.max_leftover >= 0:
                excess = leftover - self.max_leftover
                self._emit(bytes(self.input[self.istart:self.istart + excess]))
                self.istart += excess
        return None

    def read_all(self, piece: int = 512) -> Tuple[bytes, Optional[str]]:
        """io.ReadAll: the bytes and the error that stopped it (None for a clean EOF)."""
        out = bytearray()
        while True:
            b, err = self.Read(piece)
            out += b
            if err == EOF:
                return bytes(out), None
            if err is not None:
                return bytes(out), err


class ReferencePanic(Exception):
    """The emitted Go code panics here (an empty match at the very end of `data` leaves processed = len + 1 and the
    next data[processed:] is out of range, transform.go:111-113,158-162)."""


class _ErrReader:  # <name>TransformErrReader, transform.go:266-282
    def __init__(self, err: str):
        self.err = err

    def Read(self, n: int):
        return b"", self.err

    def read_all(self, piece: int = 512):
        return b"", self.err


# ------------------------------------------------------------------ the emitted processors
def _matcher(c: "E.Compiled", quirks: bool):
    """next(data, processed) -> (start, end, caps relative to `base`, base) | None."""
    if quirks:
        tdfa = getattr(c, "tdfa", None)
        held = [0] * c.prog.numcap          # Tagged-DFA programs: what the processor's reused struct holds (transform.go:123), offsets into data

        def nxt(data: bytes, processed: int):
            if processed > len(data):
                raise ReferencePanic("slice bounds out of range [%d:%d]" % (processed, len(data)))
            rem = data[processed:]
            if tdfa is not None:
                # FindBytesReuse is the Tagged DFA's; a group whose start tag is unset keeps what the struct held (tdfa.go:1031-1046);
                # a fresh struct per call of the processor (processed == 0 only at its beginning: every match moves it on)
                if processed == 0:
                    for k in range(len(held)):
                        held[k] = 0
                t = tdfa.find(rem)
                if t is None:
                    return None
                for g in range(1, c.prog.numcap // 2):
                    if t[2 * g] >= 0:
                        held[2 * g], held[2 * g + 1] = t[2 * g] + processed, t[2 * g + 1] + processed
                m = rem[t[0]:t[1]]
                idx = rem.find(m)
                if idx < 0:
                    return "break"
                return processed + idx, processed + idx + len(m), [processed + t[0], processed + t[1]] + held[2:], 0
            caps = c.find_machine.find(rem)                     # FindBytesReuse(data[processed:])
            if caps is None:
                return None
            m = rem[caps[0]:caps[1]]
            idx = rem.find(m)                                   # bytes.Index (Q4')
            if idx < 0:
                return "break"
            return processed + idx, processed + idx + len(m), caps, processed
        return nxt
    cache = {}

    def nxt(data: bytes, processed: int):
        if cache.get("data") is not data:
            ms = c.find_machine.find_all(data)
            if any(m[0] == m[1] for m in ms):
                raise ValueError("quirk-free Transform is undefined for patterns that match empty")
            cache["data"], cache["ms"] = data, ms
        for m in cache["ms"]:
            if m[0] >= processed:
                return m[0], m[1], m, 0
        return None
    return nxt


def _process(c, quirks: bool, kind: str, dl10: int, on_match):
    """processTransform (transform.go:96-170) / processSelect (:380-431) / processReject (:485-571).
    on_match(text, caps, emit) for "transform"; pred(text, caps) -> bool for "select"/"reject"."""
    nxt = _matcher(c, quirks)

    def processor(data: bytes, is_eof: bool, emit) -> int:
        processed = 0
        while True:
            r = nxt(data, processed)
            if r is None:
                if is_eof:
                    if kind != "select" and processed < len(data):
                        emit(data[processed:])
                    return len(data)
                if kind == "select":
                    return processed
                safe = len(data) - dl10
                if safe < processed:
                    safe = processed
                if safe > processed:
                    emit(data[processed:safe])
                return safe
            if r == "break":
                return processed
            ms, me, caps, base = r
            text = data[base:]
            if kind != "select" and ms > processed:
                emit(data[processed:ms])
            if kind == "transform":
                on_match(text, caps, emit)
            elif kind == "select":
                if on_match(text, caps):
                    emit(data[ms:me])
            else:
                if not on_match(text, caps):
                    emit(data[ms:me])
            processed = me if me > ms else processed + 1
    return processor


def new_transform_reader(c: "E.Compiled", source: Reader, buffer_size: int, max_leftover: int, on_match,
                         quirks: bool = True) -> Transformer:  # transform.go:28-93
    dl = transform_default_leftover(c.sel.max_len)
    if buffer_size == 0:
        buffer_size = 64 * 1024
    if max_leftover == 0:
        max_leftover = dl
    return Transformer(source, buffer_size, max_leftover, _process(c, quirks, "transform", dl // 10, on_match))


def replace_reader(c: "E.Compiled", source: Reader, template: str, quirks: bool = True, buffer_size: int = 64 * 1024,
                   max_leftover: int = 0):  # transform.go:172-256
    """buffer_size / max_leftover other than the defaults are what NewTransformReader with the same callback gives."""
    names = {n: i for i, n in enumerate(S.capture_names(c.ast)) if i >= 1 and n}
    ngroups = c.prog.numcap // 2 - 1
    try:
        segs = R.validate_and_resolve(R.parse(template), names, ngroups)
    except R.TemplateError as e:
        return _ErrReader(str(e))
    named = set(names.values())

    def on_match(text: bytes, caps: List[int], emit):
        out = bytearray()
        for s in segs:
            if s["type"] == R.LIT:
                out += s["literal"].encode("utf-8", "surrogateescape")
            elif s["type"] == R.FULL:
                out += text[caps[0]:caps[1]]
            else:
                g = s["index"]
                if g == 0:
                    out += text[caps[0]:caps[1]]
                elif g in named:                      # getCaptureByIndex: named groups only
                    out += text[caps[2 * g]:caps[2 * g + 1]]
        emit(bytes(out))
    return new_transform_reader(c, source, buffer_size, max_leftover, on_match, quirks)


def select_reader(c: "E.Compiled", source: Reader, pred, quirks: bool = True, buffer_size: int = 64 * 1024) -> Transformer:
    dl = transform_default_leftover(c.sel.max_len)                        # transform.go:322-378
    return Transformer(source, buffer_size, dl, _process(c, quirks, "select", dl // 10, pred))


def reject_reader(c: "E.Compiled", source: Reader, pred, quirks: bool = True, buffer_size: int = 64 * 1024) -> Transformer:
    dl = transform_default_leftover(c.sel.max_len)                        # transform.go:433-483
    return Transformer(source, buffer_size, dl, _process(c, quirks, "reject", dl // 10, pred))
