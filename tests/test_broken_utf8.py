"""Broken UTF-8 (instructions.go:205-295 decode with utf8.DecodeRune): a lead byte without its continuation bytes is
(RuneError, 1) to the reference.  The product screens the input of programs whose classes hold U+FFFD and matches an input with
such bytes through a copy in which they read 0xFF.  Here, on the CPU: the host restatement of that copy (tests/_hosttest.py)
followed by the table walker must give exactly what the oracle's machine (oracle/engines.py: decode_rune on the ORIGINAL bytes)
gives, on fuzzed byte strings full of truncated and malformed sequences."""
import random

import pytest

from oracle import engines as E
from tests._hosttest import HostProgram, sanitize_utf8

PATTERNS = [r"[^a]+", r"\W+", r"\D{2,}", r"[^\x00-\x7f]+", r"\P{L}+", r"(?P<k>[^=]+)=(?P<v>[^;]*);", r"x[^y]y", r"\S+\s", r"[^a-z]é[^a-z]",
            r"\p{Greek}+", r"[α-ω]+[^α-ω]", r"a.b", r"(?s)a.b", r"\b\W\b", r"[^\n]*\n", r"[\x{80}-\x{10FFFF}]+",
            r"[^\x{FFFD}]+", r"[\x{FFFD}]+"]

PIECES = [b"a", b"b", b"y", b"x", b"=", b";", b" ", b"\n", b"k", b"K", "é".encode(), "α".encode(), "ω".encode(), "€".encode(),
          "\U0001F600".encode(), "�".encode(), "K".encode(),
          b"\xc3", b"\xe2", b"\xe2\x82", b"\xf0", b"\xf0\x9f", b"\xf0\x9f\x98", b"\x80", b"\xbf", b"\xc0", b"\xc1", b"\xf5", b"\xff",
          b"\xe0\x80", b"\xe0\x9f\xbf", b"\xed\xa0\x80", b"\xf4\x90\x80\x80", b"\xf0\x8f\xbf\xbf", b"\xc2", b"\xdf", b"\xef\xbf", b"\xf4\x8f\xbf"]


def _fuzz(rng, n):
    return b"".join(rng.choice(PIECES) for _ in range(n))


def test_sanitised_walk_equals_the_oracle_on_broken_utf8(built):
    rng = random.Random(20260928)
    total = replaced = 0
    for pat in PATTERNS:
        o = E.Compiled(pat)
        hp = HostProgram(pat)
        for trial in range(60):
            b = _fuzz(rng, rng.randrange(1, 40))
            sb, n = sanitize_utf8(b)
            replaced += n
            exp = [list(r) for r in o.FindAllBytes(b)]
            got = [list(r) for r in hp.find_all(sb)]
            assert got == exp, (pat, b, sb)
            total += 1
    assert total == len(PATTERNS) * 60 and replaced > 3000


def test_which_programs_are_screened(built):
    import ctypes as C
    from regengo_amd import _capi
    lib = _capi.lib()
    for pat, want in [(r"[^a]+", 1), (r"\W", 1), (r"\P{L}", 1), (r"[a-z]+", 0), (r"\p{Greek}+", 0), (r"é+", 0), (r"a.b", 0), (r"\w+@\w+", 0),
                      (r"[\x{FFFD}]", 0), (r"[\x{FFFD}a]", 1)]:
        h = C.c_void_p()
        assert lib.rgx_compile(pat.encode(), 0, C.byref(h)) == 0
        info = _capi.Info()
        lib.rgx_program_info(h, C.byref(info))
        assert info.utf8_screened == want and info.needs_valid_utf8 == 0, pat
        lib.rgx_program_destroy(h)


def test_sanitiser_is_decode_rune(built):
    """Byte for byte: a position is replaced iff it is a lead byte (C2-F4) that the oracle's decode_rune reports as (RuneError, 1)."""
    rng = random.Random(7)
    for _ in range(400):
        b = _fuzz(rng, rng.randrange(1, 30))
        sb, _n = sanitize_utf8(b)
        for i in range(len(b)):
            r, w = E.decode_rune(b, i)
            broken = 0xC2 <= b[i] <= 0xF4 and r == 0xFFFD and w == 1
            assert (sb[i] == 0xFF and b[i] != 0xFF) == broken or (b[i] == 0xFF and sb[i] == 0xFF), (b, i)
