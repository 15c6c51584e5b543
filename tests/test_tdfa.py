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
