"""Transform path, oracle side (CPU tier): stream.Transformer and the emitted ReplaceReader / SelectReader /
RejectReader / NewTransformReader restated in oracle/transform.py, against the literal vectors the reference's own
tests hold (tests/golden/transform_kats.json <- tests/integration/streaming/transform_test.go, stream/transformer_test.go)."""
import json
import os
import random

import pytest

from oracle import engines as E
from oracle import replace as R
from oracle import syntax as S
from oracle import transform as T

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def tkats():
    return json.load(open(os.path.join(GOLDEN, "transform_kats.json")))


_compiled = {}


def compiled(pattern):
    if pattern not in _compiled:
        _compiled[pattern] = E.Compiled(pattern)
    return _compiled[pattern]


def make_pred(c, pr):
    if pr["kind"] == "all":
        return lambda text, caps: True
    g = [n.lower() for n in S.capture_names(c.ast)].index(pr["group"])
    val = pr["value"].encode()
    return lambda text, caps: text[caps[2 * g]:caps[2 * g + 1]] == val


def chain(case, inp: bytes, quirks: bool):
    """Build the reader chain of one vector (each stage reads the previous one, as in the reference tests)."""
    src = T.bytes_reader(inp)
    first = True
    for st in case["steps"]:
        c = compiled(st["pattern"])
        bs = case.get("buffer_size", 0) if first else 0
        ml = case.get("max_leftover", 0) if first else 0
        if st["op"] == "replace":
            tr = T.replace_reader(c, src, st["template"], quirks)
        elif st["op"] == "transform":
            lits = [e.encode() for e in st["emits"]]
            tr = T.new_transform_reader(c, src, bs, ml, lambda text, caps, emit, lits=lits: [emit(x) for x in lits], quirks)
        elif st["op"] == "select":
            tr = T.select_reader(c, src, make_pred(c, st["pred"]), quirks)
        else:
            tr = T.reject_reader(c, src, make_pred(c, st["pred"]), quirks)
        src = tr.Read
        first = False
    return tr


def check_case(case, out: bytes):
    if "expected" in case:
        assert out == case["expected"].encode(), case["name"]
        return
    for s in case.get("contains", []):
        assert s.encode() in out, (case["name"], s)
    for s in case.get("not_contains", []):
        assert s.encode() not in out, (case["name"], s)
    pos = [out.find(s.encode()) for s in case.get("in_order", [])]
    assert all(p >= 0 for p in pos) and pos == sorted(pos), case["name"]


@pytest.mark.parametrize("quirks", [True, False])
def test_integration_vectors(tkats, quirks):
    assert len(tkats["integration"]) == 17
    for case in tkats["integration"]:
        if case.get("expect") == "same_as_replace_all":
            for s in case["inputs"]:
                c = compiled(case["steps"][0]["pattern"])
                out, err = chain(case, s.encode(), quirks).read_all()
                assert err is None and out == R.replace_all(c, s.encode(), case["steps"][0]["template"], quirks=quirks), s
            continue
        out, err = chain(case, case["input"].encode(), quirks).read_all(case.get("read_piece", 512))
        assert err is None, case["name"]
        check_case(case, out)


def literal_processor(find: bytes, emits):
    """The reference tests' own model processor (stream/transformer_test.go:13-52): literal search, safe point
    len(data) - len(find) + 1."""
    def processor(data: bytes, is_eof: bool, emit) -> int:
        processed = 0
        while True:
            idx = data.find(find, processed)
            if idx < 0:
                if is_eof:
                    if processed < len(data):
                        emit(data[processed:])
                    return len(data)
                safe = max(len(data) - len(find) + 1, processed)
                if safe > processed:
                    emit(data[processed:safe])
                return safe
            if idx > processed:
                emit(data[processed:idx])
            for e in emits:
                emit(find if e.get("match") else e["lit"].encode())
            processed = idx + len(find)
    return processor


def test_transformer_vectors(tkats):
    assert len(tkats["transformer"]) == 13
    for case in tkats["transformer"]:
        tr = T.Transformer(T.bytes_reader(case["input"].encode()), case.get("buffer_size", 64 * 1024), case.get("max_leftover", 0),
                           literal_processor(case["find"].encode(), case["emits"]))
        out, err = tr.read_all(case.get("read_piece", 512))
        assert err is None and out == case["expected"].encode(), case["name"]


def test_empty_input_is_eof():
    tr = T.replace_reader(compiled(r"(\d+)"), T.bytes_reader(b""), "x")
    assert tr.Read(10) == (b"", T.EOF)           # stream/transformer_test.go:81-97


def test_bad_template_gives_error_reader():
    c = compiled(r"(?P<user>\w+)@(?P<domain>\w+)")
    for tmpl in ("$nosuch", "$3", "${"):
        # replace.Parse or ValidateAndResolve fails: the emitted code returns a reader that only yields that error
        out, err = T.replace_reader(c, T.bytes_reader(b"a@b"), tmpl).read_all()
        assert out == b"" and err


def test_unnamed_groups_expand_to_nothing_in_replace_reader():
    # getCaptureByIndex knows named groups only (transform.go:288-320); ReplaceAllBytes expands $1 normally
    c = compiled(r"(\d{4})-(?P<m>\d{2})")
    out, _ = T.replace_reader(c, T.bytes_reader(b"on 2024-05 ok"), "<$1|$m|$0>").read_all()
    assert out == b"on <|05|2024-05> ok"
    assert R.replace_all(c, b"on 2024-05 ok", "<$1|$m|$0>") == b"on <2024|05|2024-05> ok"


def test_chunking_matches_whole_buffer_when_matches_fit():
    """For a bounded pattern and full reads the chunk protocol must not change the result as long as no match
    straddles a safe point: compare against in-memory replace over many buffer sizes."""
    c = compiled(r"(?P<y>\d{4})-(?P<m>\d{2})-(?P<d>\d{2})")
    rnd = random.Random(7)
    body = bytearray()
    while len(body) < 20000:
        body += bytes(rnd.choice(b"abcdefghijk \n\t") for _ in range(rnd.randrange(0, 60)))
        body += b"2024-%02d-%02d" % (rnd.randrange(1, 13), rnd.randrange(1, 29))
    inp = bytes(body)
    want = R.replace_all(c, inp, "$d/$m/$y")
    for bs in (1200, 2048, 4097, 65536):      # (a buffer below MaxLeftover = 1024 makes the reference spin)
        on = lambda text, caps, emit: emit(text[caps[6]:caps[7]] + b"/" + text[caps[4]:caps[5]] + b"/" + text[caps[2]:caps[3]])
        for quirks in (True, False):
            tr = T.new_transform_reader(c, T.bytes_reader(inp), bs, 0, on, quirks)
            out, err = tr.read_all(1000)
            assert err is None
            # defaultLeftover/10 = 102 bytes are held back at each safe point and max 10-byte matches fit: exact
            assert out == want, (bs, quirks)


def test_reference_splits_matches_at_safe_points_for_unbounded_patterns():
    """An unbounded pattern keeps defaultLeftover/10 = 104857 bytes; with a small buffer safePoint = processed, the
    Transformer's MaxLeftover rule then pushes raw bytes out -- a match cut by that rule is lost.  The oracle
    reproduces this (it is the reference's behaviour, not a quirk of one engine)."""
    c = compiled(r"(\d+)")
    inp = b"a" * 10 + b"1234567890" * 3 + b"b" * 10
    tr = T.new_transform_reader(c, T.bytes_reader(inp), 16, 4, lambda text, caps, emit: emit(b"#"), True)
    out, _ = tr.read_all()
    assert out.count(b"#") > 1          # the 30-digit run is replaced piecewise
    tr2 = T.new_transform_reader(c, T.bytes_reader(inp), 64, 0, lambda text, caps, emit: emit(b"#"), True)
    assert tr2.read_all()[0] == b"a" * 10 + b"#" + b"b" * 10


def test_empty_matches_drop_bytes_in_the_reference_loop():
    # transform.go:158-162: an empty match advances `processed` by one without emitting that byte
    c = compiled(r"x*")
    with pytest.raises(T.ReferencePanic):     # ... and an empty match at the end of the data walks off the slice
        T.replace_reader(c, T.bytes_reader(b"abc"), "-", quirks=True).read_all()
    c2 = compiled(r"x*y")                      # cannot match empty: fine
    assert T.replace_reader(c2, T.bytes_reader(b"axxyb"), "-", quirks=True).read_all()[0] == b"a-b"
    c3 = compiled(r"\d*")                      # the bytes before the empty matches are lost, then the panic at the end
    tr = T.new_transform_reader(c3, T.bytes_reader(b"ab12"), 0, 0, lambda text, caps, emit: emit(b"<" + text[caps[0]:caps[1]] + b">"), True)
    with pytest.raises(T.ReferencePanic):
        tr.read_all()
    assert bytes(tr.out) == b"<><><12><>"
    with pytest.raises(ValueError):
        T.replace_reader(c, T.bytes_reader(b"abc"), "-", quirks=False).read_all()
