"""ctypes view of librgx_hosttest.so: the TEST-ONLY CPU walker over the product's tables (never part of the product)."""
import ctypes as C

from regengo_amd import build

_lib = C.CDLL(build.build_hosttest())
_lib.rgxt_compile.restype = C.c_void_p
_lib.rgxt_compile.argtypes = [C.c_char_p, C.c_uint32]
_lib.rgxt_free.argtypes = [C.c_void_p]
_lib.rgxt_compile_search.restype = C.c_void_p
_lib.rgxt_compile_search.argtypes = [C.c_char_p, C.c_uint32]
_lib.rgxt_w_sync.argtypes = [C.c_void_p, C.c_char_p, C.c_int64, C.c_int64, C.c_void_p]
_lib.rgxt_search_first.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p, C.c_int64, C.c_void_p]
_lib.rgxt_last_error.restype = C.c_char_p
_lib.rgxt_find_all.restype = C.c_int64
_lib.rgxt_find_all.argtypes = [C.c_void_p, C.c_char_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64]
_lib.rgxt_find_all_sa.restype = C.c_int64
_lib.rgxt_find_all_sa.argtypes = [C.c_void_p, C.c_char_p, C.c_int64, C.c_void_p, C.c_int64]
_lib.rgxt_prog_dump.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
_lib.rgxt_info.argtypes = [C.c_void_p, C.c_void_p]
_lib.rgxt_match.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
_lib.rgxt_roundtrip.restype = C.c_void_p
_lib.rgxt_roundtrip.argtypes = [C.c_void_p]
_lib.rgxt_sa_info.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
_lib.rgxt_reset_bytes.argtypes = [C.c_void_p, C.c_void_p]
_lib.rgxt_reset_pairs.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]

_lib.rgxt_sanitize_utf8.restype = C.c_int64
_lib.rgxt_sanitize_utf8.argtypes = [C.c_char_p, C.c_int64, C.c_void_p]

_lib.rgxt_compile_us.restype = C.c_void_p
_lib.rgxt_compile_us.argtypes = [C.c_char_p, C.c_uint32, C.c_int, C.c_int]
_lib.rgxt_free_us.argtypes = [C.c_void_p]
_lib.rgxt_us_info.argtypes = [C.c_void_p, C.c_void_p]
_lib.rgxt_us_find_all.restype = C.c_int64
_lib.rgxt_us_find_all.argtypes = [C.c_void_p, C.c_char_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_int64]
_lib.rgxt_us_simple.argtypes = [C.c_void_p]
_lib.rgxt_us_find_all_simple.restype = C.c_int64
_lib.rgxt_us_find_all_simple.argtypes = [C.c_void_p, C.c_char_p, C.c_int64, C.c_void_p, C.c_int64]
_lib.rgxt_ref_find.argtypes = [C.c_void_p, C.c_char_p, C.c_int64, C.c_void_p]
_lib.rgxt_ref_match.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]

_lib.rgxt_tdfa_header.argtypes = [C.c_void_p, C.c_void_p]
_lib.rgxt_tdfa_tables.argtypes = [C.c_void_p] + [C.c_void_p] * 5
_lib.rgxt_tdfa_find.argtypes = [C.c_void_p, C.c_char_p, C.c_int64, C.c_void_p]
_lib.rgxt_tdfa_merged_find.argtypes = [C.c_void_p, C.c_char_p, C.c_int64, C.c_void_p]
_lib.rgxt_tdfa_acc_last.argtypes = [C.c_void_p]
_lib.rgxt_memo_find.argtypes = [C.c_void_p, C.c_char_p, C.c_int64, C.c_void_p]
_lib.rgxt_memo_match.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
_lib.rgxt_tiny_find.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p, C.c_int64, C.c_int, C.c_void_p]

INFO = ["ncap", "min", "max", "ninst", "nstates", "ncls", "anchored", "fixed", "empty", "refm", "reff", "look", "maxthr"]


def sanitize_utf8(b: bytes):
    """(bytes as the run time matches them, number of broken lead bytes replaced by 0xFF)."""
    out = C.create_string_buffer(max(len(b), 1))
    n = _lib.rgxt_sanitize_utf8(b, len(b), out)
    return out.raw[:len(b)], int(n)


class HostProgram:
    def __init__(self, pattern: str, flags: int = 0, handle=None):
        if handle is None:
            handle = _lib.rgxt_compile(pattern.encode("utf-8"), flags)
            if not handle:
                raise ValueError(_lib.rgxt_last_error().decode())
        self.h = C.c_void_p(handle)
        a = (C.c_int32 * 16)()
        n = _lib.rgxt_info(self.h, a)
        self.info = dict(zip(INFO, list(a)[:n]))
        k, ex = C.c_int32(), C.c_int32()
        _lib.rgxt_sa_info(self.h, C.byref(k), C.byref(ex))
        self.sa_k, self.sa_exact = k.value, bool(ex.value)

    def __del__(self):
        try:
            _lib.rgxt_free(self.h)
        except Exception:
            pass

    def roundtrip(self) -> "HostProgram":
        h = _lib.rgxt_roundtrip(self.h)
        if not h:
            raise ValueError("blob round trip failed")
        return HostProgram("", handle=h)

    def find_all(self, b: bytes, n: int = -1):
        ncap = self.info["ncap"]
        cap = len(b) + 2
        out = (C.c_int32 * (cap * ncap))()
        c = _lib.rgxt_find_all(self.h, b, len(b), n, out, cap)
        return [list(out[i * ncap:(i + 1) * ncap]) for i in range(c)]

    @property
    def onepass(self) -> bool:
        return bool(_lib.rgxt_onepass(self.h))

    def captures_onepass(self, b: bytes, s: int, e: int):
        """The forward-only capture resolution of one-pass automata (None: the program is not one-pass)."""
        out = (C.c_int32 * self.info["ncap"])()
        return list(out) if _lib.rgxt_captures_onepass(self.h, b, len(b), s, e, out) else None

    def find_all_sa(self, b: bytes):
        ncap = self.info["ncap"]
        cap = len(b) + 2
        out = (C.c_int32 * (cap * ncap))()
        c = _lib.rgxt_find_all_sa(self.h, b, len(b), out, cap)
        if c < 0:
            return None
        return [list(out[i * ncap:(i + 1) * ncap]) for i in range(c)]

    def search_program(self, pattern: str, flags: int = 0):
        """The search automaton of `pattern` (None when it exceeds its state budget)."""
        h = _lib.rgxt_compile_search(pattern.encode("utf-8"), flags)
        return HostProgram("", handle=h) if h else None

    def search_first(self, sp: "HostProgram", b: bytes):
        out = (C.c_int32 * self.info["ncap"])()
        r = _lib.rgxt_search_first(sp.h, self.h, b, len(b), out)
        assert r >= 0, "winning thread without Capture 0"
        return list(out) if r == 1 else None

    def memo_match(self, b: bytes):
        """csrc/rgx_memo.h: MemoMatch on the host (the emitted MatchBytes interpreted): True / False, or None when it gives up / the
        program is not interpreted."""
        r = _lib.rgxt_memo_match(self.h, b, len(b))
        return None if r < 0 else bool(r)

    def tiny_find(self, sp: "HostProgram", b: bytes, ref: bool):
        """csrc/rgx_tiny.h on the host (what batch_tiny_kernel runs per lane): (code, record) -- code 0 no match, 1 found, 2 found but
        the reference's attempts step over its start (ref_fix_kernel's case), -2 string too long, -3 the automaton is not tiny."""
        out = (C.c_int32 * self.info["ncap"])()
        r = _lib.rgxt_tiny_find(sp.h, self.h, b, len(b), 1 if ref else 0, out)
        return r, list(out)

    def w_sync(self, b: bytes, y: int = 0):
        """(number of W states, flags[len+1]): flags[i] = the sync automaton started blind at y is empty at offset i."""
        f = (C.c_uint8 * (len(b) + 1))()
        n = _lib.rgxt_w_sync(self.h, b, len(b), y, f)
        return n, bytes(f)

    def match(self, b: bytes) -> bool:
        return bool(_lib.rgxt_match(self.h, b, len(b)))

    def ref_find(self, b: bytes):
        """FindBytesReuse with the reference's restart rule (Q1): spans, None, or NotImplemented."""
        out = (C.c_int32 * self.info["ncap"])()
        r = _lib.rgxt_ref_find(self.h, b, len(b), out)
        if r == -3:
            return NotImplemented
        return list(out) if r == 1 else None

    def ref_match(self, b: bytes):
        r = _lib.rgxt_ref_match(self.h, b, len(b))
        return NotImplemented if r == -3 else bool(r)

    def tdfa_tables(self):
        """The reference's Tagged DFA as the product built it, in the shape of oracle.tdfa.TDFA.tables() (None: no TDFA)."""
        hd = (C.c_int32 * 8)()
        ns = _lib.rgxt_tdfa_header(self.h, hd)
        if ns <= 0:
            return None
        ntags, sb, sa, npool, ib, ia = list(hd)[1:7]
        trans = (C.c_int16 * (ns * 128))(); act = (C.c_uint16 * (ns * 128))(); acc = (C.c_uint8 * ns)()
        acc_act = (C.c_uint16 * ns)(); pool = (C.c_int16 * npool)()
        _lib.rgxt_tdfa_tables(self.h, trans, act, acc, acc_act, pool)

        def lst(at):
            return [[pool[at + 1 + 2 * a], pool[at + 2 + 2 * a]] for a in range(pool[at])]
        return {"n_states": ns, "ntags": ntags, "start_begin": sb, "start_any": sa,
                "transitions": [[trans[s * 128 + c] for c in range(128)] for s in range(ns)],
                "tag_actions": [[lst(act[s * 128 + c]) for c in range(128)] for s in range(ns)],
                "accept": [bool(acc[s] & 1) for s in range(ns)], "accept_eot": [bool(acc[s] & 2) for s in range(ns)],
                "accept_actions": [lst(acc_act[s]) for s in range(ns)],
                "initial_begin": lst(ib), "initial_any": lst(ia)}

    def tdfa_find(self, b: bytes):
        """tdfa.go:831-1052 over the product's tables: raw tags (see oracle.tdfa.TDFA.find), None, or NotImplemented."""
        out = (C.c_int32 * 64)()
        r = _lib.rgxt_tdfa_find(self.h, b, len(b), out)
        if r == -3:
            return NotImplemented
        return list(out[:self.info["ncap"]]) if r == 1 else None

    def tdfa_merged_find(self, b: bytes):
        """The winning attempt's (start, end) through the merged-attempts automaton (rgx_dfa.h: BuildTdfaMerged), walked as the device
        walks it; None: no match; NotImplemented: not built for this program / text."""
        out = (C.c_int32 * 2)()
        r = _lib.rgxt_tdfa_merged_find(self.h, b, len(b), out)
        if r == -3:
            return NotImplemented
        return (out[0], out[1]) if r == 1 else None

    def tdfa_acc_last(self):
        """TdfaDev::tag_acc_last as the product computes it: 1 / 0, None when the program has no packed tag table."""
        r = _lib.rgxt_tdfa_acc_last(self.h)
        return None if r < 0 else r

    def memo_find(self, b: bytes):
        """FindBytesReuse of a program the reference emits with its memoising backtracker, as the device computes it (rgx_memo.h):
        spans, None, NotImplemented (not such a program), or raises when the interpreter gave up / disagreed with the automaton."""
        out = (C.c_int32 * self.info["ncap"])()
        r = _lib.rgxt_memo_find(self.h, b, len(b), out)
        if r == -3:
            return NotImplemented
        assert r >= 0, "memo interpreter: %d" % r
        return list(out) if r == 1 else None

    def reset_bytes(self):
        a = (C.c_uint8 * 256)()
        _lib.rgxt_reset_bytes(self.h, a)
        return bytes(a)

    def reset_pairs(self):
        """(class of every byte value, ncls, ncls x ncls matrix of class pairs after which every live state is dead)."""
        cls = (C.c_uint8 * 256)()
        out = (C.c_uint8 * (256 * 256))()
        n = _lib.rgxt_reset_pairs(self.h, cls, out, 256 * 256)
        if n < 0:
            raise ValueError("too many classes")
        return bytes(cls), n, bytes(out[: n * n])


def prog_dump(pattern: str) -> str:
    buf = C.create_string_buffer(1 << 16)
    r = _lib.rgxt_prog_dump(pattern.encode("utf-8"), buf, 1 << 16)
    if r < 0:
        raise ValueError(_lib.rgxt_last_error().decode())
    return buf.value.decode("utf-8")


class StartSearch:
    """The start-tracking search automaton (rgx_dfa.h: StartSearch) walked on the CPU the way the scan_us kernel walks it."""

    def __init__(self, pattern: str, flags: int = 0, max_states: int = 0, max_regs: int = 0):
        self.h = _lib.rgxt_compile_us(pattern.encode("utf-8"), flags, max_states, max_regs)
        if not self.h:
            raise ValueError(_lib.rgxt_last_error().decode())
        a = (C.c_int32 * 6)()
        _lib.rgxt_us_info(self.h, a)
        self.nstates, self.ncls, self.lookahead, self.ctx_sensitive, self.nregs, self.nstates_raw = list(a)

    def __del__(self):
        if getattr(self, "h", None):
            _lib.rgxt_free_us(self.h)
            self.h = None

    def find_all(self, b: bytes, from_pos: int = 0, slice: int = 0):
        cap = len(b) + 1
        out = (C.c_int32 * (2 * cap))()
        n = _lib.rgxt_us_find_all(self.h, b, len(b), from_pos, out, cap, slice)
        assert n >= 0
        return [(out[2 * i], out[2 * i + 1]) for i in range(n)]

    @property
    def simple(self) -> bool:
        return bool(_lib.rgxt_us_simple(self.h))

    def find_all_simple(self, b: bytes):
        """The register-free walk of simple automata: offsets of loads and final edges, starts derived afterwards."""
        cap = len(b) + 1
        out = (C.c_int32 * (2 * cap))()
        n = _lib.rgxt_us_find_all_simple(self.h, b, len(b), out, cap)
        assert n >= 0, n
        return [(out[2 * i], out[2 * i + 1]) for i in range(n)]
