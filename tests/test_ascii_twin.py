"""Tables built for texts without a byte >= 0x80 (rgx_dfa.h: kFlagAsciiText; rgx_capi.cc: AsciiTwin).  On such a text they must give
the full tables' matches and groups -- the automaton takes the same transitions when no high byte occurs -- while being far smaller
for patterns with large Unicode classes, small enough for the one-step-per-byte kernels."""
import random

from oracle import engines as E

ASCII_TEXT = 1 << 31


def _ascii_inputs(inputs, rng):
    out = []
    for s in inputs:
        b = s.encode()
        b = bytes(c if c < 0x80 else 0x20 + (c & 0x3F) for c in b)
        out.append(b)
    joined = b" ".join(out * 3) + b"\n" + b"".join(out)
    out.append(joined)
    for _ in range(4):
        k = rng.randrange(1, 200)
        out.append(bytes(rng.choice(b"abcXYZ019 _-.@/:\n\t\"<>=[]") for _ in range(k)))
    return out


def test_ascii_tables_equal_the_full_tables_on_ascii_text(corpus, kats, hostlib):
    rng = random.Random(7)
    done = 0
    items = [(e["pattern"], e["inputs"]) for e in corpus] + [(c["pattern"], c["inputs"]) for c in kats["curated_cases"]]
    for p, inputs in items:
        try:
            full = hostlib.HostProgram(p)
            twin = hostlib.HostProgram(p, ASCII_TEXT)
        except ValueError:
            continue
        for b in _ascii_inputs(inputs, rng):
            assert twin.find_all(b) == full.find_all(b), (p, b)
            done += 1
    assert done > 1500


def test_unicode_classes_shrink_to_their_ascii_members(hostlib):
    o = E.Compiled(r"[\p{L}\p{N}]+")
    text = b"abc 123 x_y  Zed9;q"
    for p in (r"\p{L}+", r"[\p{L}\p{N}]+", r"\p{Greek}+", r"(?i)k+", r"[^a]+", r".+", r"caf\x{e9}|tea"):
        full = hostlib.HostProgram(p)
        twin = hostlib.HostProgram(p, ASCII_TEXT)
        assert twin.find_all(text) == full.find_all(text), p
    assert [tuple(m[:2]) for m in o.FindAllBytes(text)] == [tuple(m[:2]) for m in hostlib.HostProgram(r"[\p{L}\p{N}]+", ASCII_TEXT).find_all(text)]
    # the start-tracking automaton of the twin is tiny: it reaches the register-free kernels (<= 250 states, <= 32 classes)
    u = hostlib.StartSearch(r"\p{L}+", ASCII_TEXT)
    assert u.simple and u.nstates <= 8 and u.ncls <= 4
    assert hostlib.StartSearch(r"\p{L}+").nstates > 250
