"""Replace path, oracle side (CPU tier): the template parser against the reference's literal vectors
(tests/golden/replace_kats.json <- replace/template_test.go) and the two readings of the emitted loop."""
import json
import os

import pytest

from oracle import engines as E
from oracle import replace as R

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def rkats():
    return json.load(open(os.path.join(GOLDEN, "replace_kats.json")))


def test_template_parse_kats(rkats):
    assert len(rkats["parse"]) == 18
    for k in rkats["parse"]:
        if k["want_err"]:
            with pytest.raises(R.TemplateError):
                R.parse(k["template"])
            continue
        got = R.parse(k["template"])
        want = k.get("segments", [])
        assert len(got) == len(want), k["name"]
        for g, w in zip(got, want):
            assert g["type"] == w["type"], k["name"]
            assert g.get("literal", "") == w.get("literal", ""), k["name"]
            assert g.get("index", 0) == w.get("index", 0), k["name"]
            assert g.get("name", "") == w.get("name", ""), k["name"]


def test_validate_and_resolve_kats(rkats):
    assert len(rkats["validate"]) == 7
    for k in rkats["validate"]:
        segs = R.parse(k["template"])
        if k["want_err"]:
            with pytest.raises(R.TemplateError):
                R.validate_and_resolve(segs, k["capture_names"], k["num_captures"])
        else:
            out = R.validate_and_resolve(segs, k["capture_names"], k["num_captures"])
            assert all(s["type"] != R.NAME for s in out), k["name"]


def test_replace_loop_readings():
    c = E.Compiled(r"(?P<user>\w+)@(?P<domain>\w+)")
    inp = b"mail bob@example and alice_1@host9 now"
    assert R.replace_all(c, inp, "<$user at ${domain}>") == b"mail <bob at example> and <alice_1 at host9> now"
    assert R.replace_all(c, inp, "$2:$1:$0:$9:$nope:$$") == b"mail example:bob:bob@example:::$ and host9:alice_1:alice_1@host9:::$ now"
    assert R.replace_all(c, inp, "X", first_only=True) == b"mail X and alice_1@host9 now"
    # on this input the emitted loop (quirks and all) agrees with the quirk-free reading
    assert R.replace_all(c, inp, "[$0]", quirks=True) == R.replace_all(c, inp, "[$0]")
    # Q4': bytes.Index finds an earlier occurrence of the match text -> the reference splices at the wrong place
    d = E.Compiled(r"(?P<y>\d{4})-(?P<m>\d{2})")
    q = b"x2024-01 12024-01"          # second match text "2024-01" occurs... the Q1 restart also skips it
    assert R.replace_all(d, q, "<$y>") == b"x<2024> 1<2024>"
    assert R.replace_all(d, q, "<$y>", quirks=True) != R.replace_all(d, q, "<$y>")
    # empty matches: one expansion at every position, also at the very end (FindBytesReuse tries offset len)
    e = E.Compiled(r"(x*)")
    assert R.replace_all(e, b"ab", "-") == b"-a-b-"
    assert R.replace_all(e, b"ab", "-", quirks=True) == b"-a-b-"
    # an empty match is allowed right after a non-empty one (Q3): "[]" follows "[xx]"
    assert R.replace_all(e, b"axxb", "[$1]") == b"[]a[xx][]b[]"
    assert R.replace_all(e, b"axxb", "[$1]", quirks=True) == b"[]a[xx][]b[]"


def test_tagged_dfa_loop_reuses_one_result_struct():
    """replace.go:216 + tdfa.go:1031-1046: `r` is one struct for the whole loop and the Tagged-DFA engine assigns a group's field only when
    the group's start tag is set -- a group the match leaves out keeps the text an earlier match gave it (the zero struct's empty field
    before that)."""
    from oracle import engines as E
    from oracle import replace as R
    o = E.Compiled(r"(?P<protocol>https?)://(?P<host>[\w\.-]+)(?::(?P<port>\d+))?(?P<path>/[\w\./]*)?")
    assert o.tdfa is not None
    assert R.replace_all(o, b"http://a:80/x http://b", "[$port]", quirks=True) == b"[80] [80]"
    assert R.replace_all(o, b"http://b http://a:80/x http://c", "[$port$path]", quirks=True) == b"[] [80/x] [80/x]"
    assert R.replace_all(o, b"http://a:80/x http://b", "[$port]", quirks=True, first_only=True) == b"[80] http://b"
