"""The reference's Tagged-DFA capture engine (tdfa.go), restated in oracle/tdfa.py, pinned by the literal tables of the
three checked-in TDFA matchers, and related to the leftmost-first semantics the GPU path implements."""
import json
import os

import pytest

from oracle import engines as E
from oracle import syntax as S
from oracle import tdfa

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def tables():
    return json.load(open(os.path.join(GOLDEN, "tdfa_tables.json")))


def test_construction_reproduces_emitted_tables(tables):
    """transitions / tagAction* / acceptStates / acceptStatesEOT / acceptAction* literals of URLCapture.go,
    TDFASemVer.go and ipv4_pattern.go, state numbering included."""
    for name, d in tables.items():
        t = tdfa.build_for_pattern(d["pattern"])
        assert t is not None, name
        tb = t.tables()
        assert tb["transitions"] == d["transitions"], name
        assert tb["accept"] == d["accept"] and tb["accept_eot"] == d["accept_eot"], name
        assert tb["tag_action_count"] == d["tag_action_count"], name
        for s in range(tb["n_states"]):
            for c in range(128):
                n = d["tag_action_count"][s][c]
                ref = [[d["tag_action_tags"][s][c][k], d["tag_action_offsets"][s][c][k]] for k in range(n)]
                assert ref == tb["tag_actions"][s][c], (name, s, c)
            n = d["accept_action_count"][s]
            ref = [[d["accept_action_tags"][s][k], d["accept_action_offsets"][s][k]] for k in range(n)]
            assert ref == tb["accept_actions"][s], (name, s)
        assert (t.start_begin, t.start_any) == (d["start_begin"], d["start_any"])


def test_tdfa_feasibility_matches_generated_files(progs):
    """Files that contain Ins<i> blocks in their Find functions were NOT emitted with TDFA although the analysis asks
    for it (state budget exceeded): the restated builder must agree (e.g. (?P<outer>(?P<inner>a+)+)b)."""
    for e in progs:
        ast, p = S.compile_pattern(e["pattern"])
        sel = E.select(ast, p)
        if sel.find_engine != "tdfa?":
            continue
        t = tdfa.build_for_pattern(e["pattern"])
        emitted_backtracking = "inst" in e
        assert (t is None) == emitted_backtracking, e["file"]


def _norm(caps):
    """backtracking convention (0,0) for unmatched groups -> (-1,-1), to compare with TDFA tags."""
    out = list(caps)
    for g in range(1, len(out) // 2):
        if out[2 * g] == 0 and out[2 * g + 1] == 0:
            out[2 * g] = out[2 * g + 1] = -1
    return out


def test_tdfa_find_equals_leftmost_first_on_reference_inputs(tables, kats, corpus):
    """On the reference's own test inputs (curated cases + e2e corpus) the TDFA result is the leftmost-first result --
    that is what its generated tests assert against stdlib (test_gen.go:72-239)."""
    items = []
    for c in kats["curated_cases"]:
        items.append((c["pattern"], c["inputs"]))
    for e in corpus:
        if "TDFA" in e["engine_labels"]:
            items.append((e["pattern"], e["inputs"]))
    checked = 0
    for pat, inputs in items:
        t = tdfa.build_for_pattern(pat)
        if t is None:
            continue
        ast, p = S.compile_pattern(pat)
        if not E.select(ast, p).catastrophic:
            continue
        m = E.Machine(p)
        for s in inputs:
            b = s.encode()
            if any(x >= 128 for x in b):
                continue
            got = t.find(b)
            fa = m.find_all_stdlib_like(b)
            exp = fa[0] if fa else None
            assert (got is None) == (exp is None), (pat, b)
            if got is not None:
                assert got[:2] == exp[:2], (pat, b)
                assert got == [x if x >= 0 else -1 for x in exp], (pat, b)
            checked += 1
    assert checked >= 40


def test_q6_and_q11_are_real():
    """Where the TDFA differs from leftmost-first (documented, not emulated by the GPU path)."""
    t = tdfa.build_for_pattern(r"(?P<a>x+?)(?P<b>y+)?z?")
    # Q11: FindAllBytes advances by the match LENGTH: a match not at the start of the slice is reported again
    u = tdfa.build_for_pattern(r"(?P<protocol>https?)://(?P<host>[\w\.-]+)(?::(?P<port>\d+))?(?P<path>/[\w\./]*)?")
    b = b"a long preamble before it: http://a.b and more"
    dup = u.find_all(b)
    ok = u.find_all_fixed(b)
    assert len(ok) == 1 and len(dup) > 1 and dup[0] == dup[1]
    assert t is not None


def test_product_tables_are_the_emitted_tables(tables):
    """The C++ construction (csrc/rgx_ref_engine.cc: BuildRefTdfa) -- the tables the HIP walker stages -- equals the literal tables of
    the three checked-in TDFA matchers, and the oracle's on every TDFA-class pattern of the corpus (blob round trip included)."""
    from tests._hosttest import HostProgram
    for name, d in tables.items():
        hp = HostProgram(d["pattern"])
        for tb in (hp.tdfa_tables(), hp.roundtrip().tdfa_tables()):
            assert tb is not None, name
            assert tb["transitions"] == d["transitions"], name
            assert tb["accept"] == d["accept"] and tb["accept_eot"] == d["accept_eot"], name
            for s in range(tb["n_states"]):
                for c in range(128):
                    n = d["tag_action_count"][s][c]
                    ref = [[d["tag_action_tags"][s][c][k], d["tag_action_offsets"][s][c][k]] for k in range(n)]
                    assert ref == tb["tag_actions"][s][c], (name, s, c)
                n = d["accept_action_count"][s]
                ref = [[d["accept_action_tags"][s][k], d["accept_action_offsets"][s][k]] for k in range(n)]
                assert ref == tb["accept_actions"][s], (name, s)
            assert (tb["start_begin"], tb["start_any"]) == (d["start_begin"], d["start_any"])


def test_product_tables_equal_the_oracle_on_the_corpus(kats, corpus):
    import random
    import zlib
    from tests._hosttest import HostProgram
    pats = [c["pattern"] for c in kats["curated_cases"]] + [e["pattern"] for e in corpus]
    seen = built_merged = 0
    for pat in dict.fromkeys(pats):
        o = E.Compiled(pat)
        hp = HostProgram(pat)
        tb = hp.tdfa_tables()
        assert (tb is None) == (o.tdfa is None), pat
        if tb is None:
            continue
        seen += 1
        ob = o.tdfa.tables()
        for k in ("n_states", "transitions", "tag_actions", "accept", "accept_eot", "accept_actions"):
            assert tb[k] == ob[k], (pat, k)
        assert tb["initial_begin"] == [list(a) for a in o.tdfa.initial_begin] and tb["initial_any"] == [list(a) for a in o.tdfa.initial_any]
        assert tb["ntags"] == 2 * max(o.tdfa.ncap_names, 1)
        # the find loop over the product's tables (what a lane of rgx_tdfa.hip runs) against the restated emitted loop
        rnd = random.Random(zlib.crc32(pat.encode()))
        alpha = sorted({c for row in ob["transitions"] for c in range(128) if row[c] >= 0})
        texts = [b"", b"\xc3\xa9"]
        for _ in range(60):
            n = rnd.randint(1, 60)
            texts.append(bytes(rnd.choice(alpha) if rnd.random() < 0.93 else rnd.choice([32, 10, 200]) for _ in range(n)))
        merged = 0
        for b in texts:
            want = o.tdfa.find(b)
            assert hp.tdfa_find(b) == want, (pat, b)
            # ... and the same loop as ONE forward walk over the merged-attempts automaton (rgx_dfa.h: BuildTdfaMerged; what
            # rgx_tdfa.hip's per-string kernel walks): the winning attempt's start and end
            m = hp.tdfa_merged_find(b)
            if m is not NotImplemented:
                merged += 1
                assert m == (None if want is None else (want[0], want[1])), (pat, b, m, want)
        built_merged += 1 if merged else 0
    assert seen >= 12 and built_merged >= 8


def test_product_tables_equal_the_oracle_on_random_patterns():
    """The same comparison over RANDOM patterns of the class (tests/_fuzzgen.py: captures + nested quantifiers): the product's
    construction (csrc/rgx_ref_engine.cc: BuildRefTdfa) against the oracle's restatement of tdfa.go, tables and find loop.  (Round 5:
    `.[^a]` -- a set whose `.` thread came after a class thread that consumes '\n' lost the newline from its possible bytes in the
    product; no pattern of the corpus has such a set.)"""
    import random
    import zlib
    from tests import _fuzzgen as F
    from tests._hosttest import HostProgram
    seen = finds = 0
    for seed in range(50, 70):
        for pat in F.gen_patterns(seed, 60):
            try:
                o = E.Compiled(pat)
            except Exception:
                continue
            hp = HostProgram(pat)
            tb = hp.tdfa_tables()
            assert (tb is None) == (o.tdfa is None), pat
            if tb is None:
                continue
            seen += 1
            ob = o.tdfa.tables()
            for k in ("n_states", "transitions", "tag_actions", "accept", "accept_eot", "accept_actions"):
                assert tb[k] == ob[k], (pat, k)
            assert tb["initial_begin"] == [list(a) for a in o.tdfa.initial_begin] and tb["initial_any"] == [list(a) for a in o.tdfa.initial_any], pat
            if len(o.tdfa.states) > 120:
                continue
            ob["start_any"] = o.tdfa.start_any
            rnd = random.Random(zlib.crc32(pat.encode()))
            for _ in range(12):
                b = F.tdfa_guided_text(ob, rnd, rnd.randint(1, 60))
                want = o.tdfa.find(b)
                assert hp.tdfa_find(b) == want, (pat, b)
                m = hp.tdfa_merged_find(b)
                if m is not NotImplemented:
                    assert m == (None if want is None else (want[0], want[1])), (pat, b, m, want)
                finds += 1
    assert seen >= 120 and finds >= 1000, (seen, finds)


def test_accept_actions_once_flag(kats, corpus):
    """TdfaDev::tag_acc_last (the tag walk applies the accept actions once, behind its last byte) is set exactly when no earlier
    accept's write can be the last write of its tag (every way on to a state the walk can stop in writes the tag again) --
    recomputed here from the oracle's tables -- and, independently of the argument in rgx_ref_engine.cc, a walk that applies the
    accept actions only at its last accept gives the emitted loop's tags on those programs (simulated here over the oracle's tables
    on guided texts)."""
    import random
    import zlib
    from tests._hosttest import HostProgram
    from tests import _fuzzgen as F
    pats = [c["pattern"] for c in kats["curated_cases"]] + [e["pattern"] for e in corpus]
    pats += [r"(?P<x>(?:a+)+?)(?P<y>b+?)", r"(?P<x>(?:[a-c]+,)+?)(?P<y>\d+)?", r"(?P<a>(?:x+)+)(?P<b>y)?(?P<c>z+)?"]
    # ... and random patterns of the class (tests/_fuzzgen.py): automata the corpus does not have, programs the flag is OFF for among them
    for seed in range(50, 58):
        pats += F.gen_patterns(seed, 60)
    on = off = sims = 0
    for pat in dict.fromkeys(pats):
        try:
            o = E.Compiled(pat)
        except Exception:
            continue
        if o.tdfa is None:
            continue
        got = HostProgram(pat).tdfa_acc_last()
        if got is None:
            continue
        t = o.tdfa
        # pend[q][tag]: arriving in q with an earlier accept's write of `tag` still standing, the walk can stop with it standing
        ns_, ntags_ = len(t.states), max(t.ncap_names, 1) * 2
        acc_tags = {q: {tag for tag, _ in t.accept_actions.get(q, [])} for q in range(ns_)}
        stops = [q for q in range(ns_) if t.accept.get(q) or t.accept_eot.get(q)]
        pend = {(q, tag) for q in stops for tag in range(ntags_) if tag not in acc_tags[q]}
        grew = True
        while grew:
            grew = False
            for q in range(ns_):
                for c, nq in t.trans[q].items():
                    wr = {tag for tag, _ in t.tag_actions[q].get(c, [])}
                    for tag in range(ntags_):
                        if (q, tag) not in pend and (nq, tag) in pend and tag not in wr:
                            pend.add((q, tag))
                            grew = True
        fine = all(not ((nq, tag) in pend and tag not in {x for x, _ in t.tag_actions[q].get(c, [])})
                   for q in range(ns_) if t.accept.get(q) for tag in acc_tags[q] for c, nq in t.trans[q].items())
        assert got == (1 if fine else 0), (pat, got, fine)
        on += got
        off += 1 - got
        if not got:
            continue
        tb = t.tables()
        tb["start_any"] = t.start_any
        rnd = random.Random(zlib.crc32(pat.encode()) ^ 7)
        for _ in range(40):
            b = F.tdfa_guided_text(tb, rnd, rnd.randint(1, 80))
            want = t.find(b)
            if want is None:
                continue
            # the winning attempt again, accept actions at its end only
            start, stop = want[0], want[1]
            ntags = max(t.ncap_names, 1) * 2
            tags = [-1] * ntags
            tags[0] = start
            state = t.start_begin if start == 0 else t.start_any
            for tg, _ in (t.initial_begin if start == 0 else t.initial_any):
                tags[tg] = start
            for i in range(start, stop):
                c = b[i]
                for tg, of in t.tag_actions[state].get(c, []):
                    tags[tg] = i + 1 - of
                state = t.trans[state][c]
            for tg, of in t.accept_actions.get(state, []):
                tags[tg] = stop - of
            tags[1] = stop
            for g in range(1, t.ncap_names):
                if tags[2 * g] >= 0:
                    if tags[2 * g + 1] < 0:
                        tags[2 * g + 1] = stop
                else:
                    tags[2 * g + 1] = -1
            assert tags == want, (pat, b, tags, want)
            sims += 1
    assert on >= 30 and off >= 3 and sims >= 600, (on, off, sims)


def test_c_port_of_the_emitted_tdfa_equals_the_restatement(kats, corpus):
    """oracle/tdfa_c.py (the emitted tables as C arrays + the emitted loop: bulk checker and bench.py's cpu_baseline for
    --config c3 --force-tdfa) against oracle/tdfa.py, single finds and the FindReader chunk loop."""
    import random
    import zlib
    import numpy as np
    from oracle.tdfa_c import CTdfa
    from tests import _fuzzgen as F
    pats = [(c["pattern"], False) for c in kats["curated_cases"]] + [(e["pattern"], False) for e in corpus] + [(r"(?P<user>\w+)@(?P<domain>\w+)", True)]
    seen = 0
    for pat, force in dict.fromkeys(pats):
        try:
            ct = CTdfa(pat, force=force)
        except ValueError:
            continue
        seen += 1
        t = ct.t
        tb = t.tables()
        tb["start_any"] = t.start_any
        rnd = random.Random(zlib.crc32(pat.encode()) ^ 99)
        for _ in range(60):
            b = F.tdfa_guided_text(tb, rnd, rnd.randint(0, 100))
            assert ct.find(b) == t.find(b), (pat, b)
        text = b" ".join(F.tdfa_guided_text(tb, rnd, rnd.randint(5, 60)) for _ in range(40))
        rows, sp = [], 0
        while sp < len(text):
            r = t.find(text[sp:])
            if r is None:
                break
            rows.append([x + sp if x >= 0 else -1 for x in r])
            sp = sp + r[1] if r[1] > r[0] else sp + 1
        got = ct.chain_np(np.frombuffer(text, dtype=np.uint8))
        assert got.tolist() == rows, pat
    assert seen >= 12
