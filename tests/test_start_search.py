"""The start-tracking search automaton (rgx_dfa.h: StartSearch; DESIGN.md "one step per byte"): FindAll by ONE table step per
input byte with the match start kept in a register, walked on the CPU exactly as the scan_us kernel walks it per lane,
against the oracle's FindAllBytes (find.go:130-466 restated in oracle/engines.py) on the corpus, its mutations and seeded
random patterns."""
import random

import pytest

from oracle import engines as E
from tests import _fuzzgen as G
from tests.test_host_tables import _mutations


def _items(corpus, kats):
    return [(e["pattern"], e["inputs"]) for e in corpus] + [(c["pattern"], c["inputs"]) for c in kats["curated_cases"]]


def test_start_search_equals_oracle_findall_on_corpus(corpus, kats, hostlib):
    rng = random.Random(99)
    eligible = total = simple_total = 0
    why = {}
    for p, inputs in _items(corpus, kats):
        try:
            u = hostlib.StartSearch(p)
        except ValueError as ex:
            why[str(ex)] = why.get(str(ex), 0) + 1
            continue
        eligible += 1
        o = E.Compiled(p)
        bs = [s.encode() for s in inputs]
        big = b" ".join(bs * 3) + b"\n" + b"".join(bs)
        for b in [big] + _mutations(inputs, rng):
            exp = [tuple(m[:2]) for m in o.find_machine.find_all(b)]
            assert u.find_all(b) == exp, (p, b)
            # cut into slices of start positions the way the kernel's lanes own them (stop rule: StartSearch.oldest)
            assert u.find_all(b, 0, 5) == exp, ("slices of 5", p, b)
            assert u.find_all(b, 0, 64) == exp, ("slices of 64", p, b)
            if u.simple:
                assert u.find_all_simple(b) == exp, ("register-free walk", p, b)
                simple_total += 1
            total += 1
    # anchored patterns (156 of the corpus) and patterns that can match empty take other kernels
    assert set(why) <= {"ineligible: anchored", "ineligible: can match empty", "ineligible: age", "ineligible: state budget"}, why
    assert why.get("ineligible: age", 0) + why.get("ineligible: state budget", 0) <= 2, why
    assert eligible >= 95 and total > 2000 and simple_total > 1000, (eligible, total, simple_total, why)


def test_start_search_from_a_later_position(hostlib):
    """The walk may begin at any FindAll sync point, with the previous byte as look-behind context only."""
    for p, text in [(r"\bfoo\b", b"xfoo foo foox foo"), (r"(?m)^ab+", b"ab\nabb ab\nab"), (r"\w+@\w+", b"a@b c@d"), (r"\d+", b"12 34")]:
        u = hostlib.StartSearch(p)
        o = E.Compiled(p)
        exp = [tuple(m[:2]) for m in o.find_machine.find_all(text)]
        for pos in range(len(text) + 1):
            inside = any(s < pos < e for s, e in exp)
            if inside:
                continue        # not a position FindAll ever stands at
            assert u.find_all(text, pos) == [m for m in exp if m[0] >= pos], (p, pos)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_start_search_random_patterns(hostlib, seed):
    """Seeded random patterns (tests/_fuzzgen.py), ASCII and UTF-8 flavours: every eligible one against the oracle."""
    rng = random.Random(1000 + seed)
    done = eligible = 0
    pats = G.gen_patterns(3000 + seed, 300) + G.gen_patterns_u(4000 + seed, 120)
    for n, p in enumerate(pats):
        utf8 = n >= 300
        try:
            o = E.Compiled(p)
        except Exception:
            continue
        if G.has_empty_loop(o.prog) and not o.find_machine.memo:
            continue                       # the reference's own Find* would not terminate on this pattern
        try:
            u = hostlib.StartSearch(p)
        except ValueError:
            continue
        eligible += 1
        for _ in range(6):
            b = G.gen_input_u(rng, rng.choice([0, 1, 4, 30, 120])) if utf8 else G.gen_input(rng, rng.choice([0, 1, 5, 40, 200]))
            exp = [tuple(m[:2]) for m in o.find_machine.find_all(b, q8=False)]
            assert u.find_all(b) == exp, (p, b)
            assert u.find_all(b, 0, 3) == exp, ("slices of 3", p, b)
            if u.simple:
                assert u.find_all_simple(b) == exp, ("register-free walk", p, b)
            done += 1
    assert eligible > 60 and done > 360, (eligible, done)


def test_start_search_with_few_registers(corpus, kats, hostlib):
    """One or two registers only: younger groups keep exact ages in the state identity (more states, fewer eligible
    patterns, same answers)."""
    rng = random.Random(7)
    n = 0
    for regs in (1, 2):
        for p, inputs in _items(corpus, kats)[::3]:
            try:
                u = hostlib.StartSearch(p, max_regs=regs)
            except ValueError:
                continue
            assert u.nregs <= regs
            o = E.Compiled(p)
            for b in _mutations(inputs, rng)[:10]:
                exp = [tuple(m[:2]) for m in o.find_machine.find_all(b)]
                assert u.find_all(b, 0, 16) == exp, (regs, p, b)
                n += 1
    assert n > 300


def test_five_concurrent_starts_need_five_registers(hostlib):
    """`(\\w+\\s+){5}\\w+`: five word starts are alive at once.  Four registers are not enough (the product tries 1, 2, 4, 8 and the
    register kernel has an instance for eight since round 3); with eight the walk equals the oracle."""
    p = r"(?P<words>(?P<word>\w+\s+){5})(?P<end>\w+)"
    with pytest.raises(ValueError):
        hostlib.StartSearch(p, 0, 2000, 4)
    u = hostlib.StartSearch(p, 0, 2000, 8)
    assert 4 < u.nregs <= 8
    o = E.Compiled(p)
    text = b"a bb  ccc d e f g hh, i j k l m n o p q\nr s t u v w\tx y z 1 2 3 4 5 6 7 8 9 0 - a b c d e f"
    exp = [tuple(m[:2]) for m in o.find_machine.find_all(text)]
    assert len(exp) >= 3
    assert u.find_all(text) == exp
    assert u.find_all(text, 0, 5) == exp
