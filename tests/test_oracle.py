"""The oracle against the reference's own golden data (CPU tier).

Pins, in order: (1) the front-end restatement against Go's real syntax.Prog recovered from the reference's
generated files; (2) the analyses against the literal tables of the reference's unit tests; (3) the matcher
machine against the reference's literal capture KATs and streaming-offset scenario; (4) a secondary cross-check
against CPython `re` (leftmost-first backtracking) on the ASCII / non-empty-match subset.
"""
import io
import re

import pytest

from oracle import engines as E
from oracle import syntax as S


def _inst_bytes(ins):
    out = []
    r = ins.rune
    if ins.op == S.InstRune1:
        r = [r[0], r[0]]
    for i in range(0, len(r), 2):
        lo, hi = r[i], min(r[i + 1], 127)
        if lo <= hi:
            if out and out[-1][1] == lo - 1:
                out[-1][1] = hi
            else:
                out.append([lo, hi])
    return out


def test_frontend_reproduces_go_progs(progs):
    """Every Prog recovered from the reference's checked-in generated matchers is reproduced instruction for
    instruction (op, out, arg, byte set) -- pins oracle/syntax.py against Go's regexp/syntax."""
    n = 0
    for e in progs:
        if "inst" not in e:
            continue
        n += 1
        ast, p = S.compile_pattern(e["pattern"])
        assert len(p.inst) == len(e["inst"]), e["file"]
        assert p.start == e["start"] and p.numcap == e["numcap"], e["file"]
        for i, (a, b) in enumerate(zip(p.inst, e["inst"])):
            name = S.INST_NAMES[a.op]
            assert name == b["op"], (e["file"], i)
            if name in ("alt", "cap", "empty"):
                assert (a.out, a.arg) == (b["out"], b["arg"]), (e["file"], i)
            if name in ("rune", "rune1", "any", "anynotnl", "nop"):
                assert a.out == b["out"], (e["file"], i)
            if name in ("rune", "rune1") and "bytes" in b:
                assert _inst_bytes(a) == b["bytes"], (e["file"], i)
    assert n >= 15


def test_emitted_constants(progs):
    """<Name>MinMatchLen / MaxMatchLen / DefaultMaxLeftover() literals in the generated files."""
    for e in progs:
        ast, p = S.compile_pattern(e["pattern"])
        if "MinMatchLen" in e:
            assert E.min_match_len(ast) == e["MinMatchLen"], e["file"]
            assert E.max_match_len(ast) == e["MaxMatchLen"], e["file"]
        if e.get("DefaultMaxLeftover") is not None:
            assert E.default_max_leftover(E.max_match_len(ast)) == e["DefaultMaxLeftover"], e["file"]


def test_engine_selection_matches_generated_files(progs):
    """Which engine the reference emitted (visible in the files) == what the restated selection predicts."""
    for e in progs:
        ast, p = S.compile_pattern(e["pattern"])
        sel = E.select(ast, p)
        assert ("thompson" if sel.thompson_for_match else "backtracking") == e["match_engine"], e["file"]
        if "inst" in e:
            # a backtracking Find was emitted: TDFA was not used for this pattern
            assert sel.find_engine in ("backtracking", "tdfa?"), e["file"]
            assert sel.per_capture_checkpoint == e["per_capture_checkpoint"], e["file"]


def test_match_length_literals(kats):
    for pat, want in kats["min_match_len"]:
        assert E.min_match_len(S.simplify(S.parse(pat))) == want, pat
    for pat, want in kats["max_match_len"]:
        assert E.max_match_len(S.simplify(S.parse(pat))) == want, pat
    for pat, mn, mx in kats["analyze_match_length"]:
        ast = S.simplify(S.parse(pat))
        assert (E.min_match_len(ast), E.max_match_len(ast)) == (mn, mx), pat
    for mx, want in kats["default_max_leftover"]:
        assert E.default_max_leftover(mx) == want
    for mx, want in kats["min_buffer_size"]:
        # MinBufferSize (analysis_match_len.go:275-290) == the emitted minBuffer for these cases
        assert E.min_buffer(mx) == want


def test_complexity_literals(kats):
    for pat, want in kats["nested_quantifiers"]:
        assert E.detect_nested_quantifiers(S.parse(pat)) == want, pat
    for pat, cat, thompson in kats["analyze_complexity"]:
        ast, p = S.compile_pattern(pat)
        sel = E.select(ast, p)
        assert sel.catastrophic == cat, pat
        assert ((sel.catastrophic or sel.nested_loops) and not sel.end_anchor) == thompson, pat


def test_stream_config_literals(kats):
    for c in kats["config_validate"]:
        cfg = E.StreamConfig(c["cfg"]["BufferSize"], c["cfg"]["MaxLeftover"])
        assert (cfg.validate(c["minBuffer"]) is not None) == c["wantErr"]
    for c in kats["config_apply_defaults"]:
        cfg = E.StreamConfig(c["cfg"]["BufferSize"], c["cfg"]["MaxLeftover"]).apply_defaults(c["minBuffer"], c["defaultLeftover"])
        assert (cfg.BufferSize, cfg.MaxLeftover) == (c["wantBufferSize"], c["wantMaxLeftover"])


def test_repeating_capture_kats(kats):
    """repeating_test.go:45-98: literal stdlib answers, e.g. ((\\w)+) on abc -> ["abc","c"]."""
    for k in kats["repeating_captures"]:
        c = E.Compiled(k["pattern"])
        b = k["input"].encode()
        caps = c.FindBytes(b)
        assert caps is not None
        assert b[caps[0]:caps[1]].decode() == k["full"]
        got = [b[caps[2 * g]:caps[2 * g + 1]].decode() for g in range(1, len(caps) // 2)]
        assert got == k["captures"], k["pattern"]
        alls = c.FindAllBytes(b)
        assert alls and alls[0] == caps


def _reader(data: bytes):
    bio = io.BytesIO(data)
    return bio.read


def test_streaming_boundary_offsets(kats):
    """streaming_test.go:190-280: 100 KiB, dates at 100/32768/65530/65550/70000/99000, 64 KiB buffer."""
    sb = kats["streaming_boundary"]
    data = bytearray(sb["fill"].encode() * sb["total_size"])
    for pos, d in zip(sb["positions"], sb["dates"]):
        data[pos:pos + len(d)] = d.encode()
    c = E.Compiled(sb["pattern"])
    got = []
    err = c.FindReader(_reader(bytes(data)), E.StreamConfig(BufferSize=sb["buffer_size"]),
                       lambda m: got.append((m.StreamOffset, m.match_bytes.decode())) or True)
    assert err is None
    assert got == list(zip(sb["positions"], sb["dates"]))
    # and FindAllBytes agrees
    assert [x[0] for x in c.FindAllBytes(bytes(data))] == sb["positions"]


def test_reference_quirks_are_modelled():
    """Q1 (restart from the failure offset) and Q3 (no attempt at len; empty after non-empty)."""
    c = E.Compiled(r"(?P<year>\d{4})-(?P<month>\d{2})-(?P<day>\d{2})")
    assert c.MatchBytes(b"12024-01-15") is False        # stdlib: true
    assert c.FindBytes(b"12024-01-15") is None
    assert [x[:2] for x in c.FindAllBytes(b"12024-01-15")] == [[1, 11]]
    e = E.Compiled(r"(a*)")
    assert [x[:2] for x in e.FindAllBytes(b"baaac")] == [[0, 0], [1, 4], [4, 4]]   # Q3: empty at 4 right after [1,4); none at len


ASCII_SAFE = re.compile(r"^[\x20-\x7e]*$")


def _py_ok(pattern: str) -> bool:
    if not ASCII_SAFE.match(pattern):
        return False
    for bad in ("\\z", "\\p", "\\P", "\\Q", "[[:", "\\C", "(?U", "\\pL", "\\x{"):
        if bad in pattern:
            return False
    return True


def test_cross_check_with_python_re(corpus):
    """Secondary: CPython's backtracking `re` on bytes has the same leftmost-first semantics for ASCII patterns
    without empty matches.  Divergences must be explained by a Go/Python rule difference; none are expected here."""
    checked = 0
    for e in corpus:
        p = e["pattern"]
        if not _py_ok(p):
            continue
        try:
            pyre = re.compile(p.replace("(?P<", "(?P<").encode())
        except re.error:
            continue
        c = E.Compiled(p)
        if c.sel.min_len == 0:   # empty matches: Q3 differs from stdlib/Python by rule
            continue
        for s in e["inputs"]:
            b = s.encode()
            if any(x >= 0x80 for x in b) or (b.endswith(b"\n") and "$" in p):   # Python's $ also matches before a final \n
                continue
            exp = [[m.start(), m.end()] for m in pyre.finditer(b)]
            got = [x[:2] for x in c.find_machine.find_all(b)]
            assert got == exp, (p, b)
            checked += 1
    assert checked > 700


def test_generated_c_matches_python_machine(corpus, built):
    """oracle/gen_c.py (the bulk checker / cpu_baseline) == oracle/engines.py on the corpus."""
    from oracle.gen_c import CMatcher
    n = 0
    for e in corpus[::4]:
        try:
            cm = CMatcher(e["pattern"])
        except NotImplementedError:
            continue
        o = E.Compiled(e["pattern"])
        for s in e["inputs"] + [" ".join(e["inputs"])]:
            b = s.encode()
            assert cm.find_all(b) == o.find_machine.find_all(b), (e["pattern"], b)
            assert cm.find(b) == o.find_machine.find(b), (e["pattern"], b)
            n += 1
    assert n > 300


@pytest.mark.parametrize("pattern", [
    r"(?P<year>\d{4})-(?P<month>\d{2})-(?P<day>\d{2})",                                                        # plain backtracking
    r"(?P<full>(?P<proto>https?|ftp)://(?P<host>[\w.-]+)(?P<port>:\d+)?(?P<path>/[\w./-]*)?)",                # memoising
    r"(?P<protocol>https?)://(?P<host>[\w\.-]+)(?::(?P<port>\d+))?(?P<path>/[\w\./]*)?",                      # Tagged DFA (URLCapture)
    r"\w+",                                                                                                    # suffix matches behind a dropped straddler
])
def test_c_find_reader_equals_the_restated_loop(pattern):
    """oracle/gen_c.py: m_find_reader and oracle/tdfa_c.py: t_find_reader -- the bulk checkers of the GPU tier's chunk runs
    (tests/test_gpu_chunks.py) -- against oracle.engines.find_reader, the block-for-block restatement of streaming.go:85-255 that the
    reference's literal streaming scenario pins (tests/golden/kats.json): the same callbacks (StreamOffset, ChunkIndex, spans) over a
    stream of several chunks, at the default Config and at ones whose keep points fall elsewhere."""
    import numpy as np
    from oracle import engines as E
    from regengo_amd import synth
    comp = E.Compiled(pattern)
    if comp.tdfa is not None:
        from oracle.tdfa_c import CTdfa
        cm = CTdfa(pattern)
    else:
        from oracle.gen_c import CMatcher
        cm = CMatcher(pattern)
    data = synth.web_log_tile(1 << 20)[:150000]
    arr = np.frombuffer(data, dtype=np.uint8)
    for B, ML in [(65536, 0), (65536, 100), (70000, 33000)]:
        rows = []
        pos = [0]

        def read(k):
            d = data[pos[0]:pos[0] + k]
            pos[0] += len(d)
            return d
        assert comp.FindReader(read, E.StreamConfig(B, ML), lambda m: rows.append(m) or True) is None
        rc = E.StreamConfig(B, ML).apply_defaults(E.min_buffer(comp.sel.max_len), E.default_max_leftover(comp.sel.max_len))
        got = cm.find_reader_np(arr, rc.BufferSize, rc.MaxLeftover)
        assert len(rows) == got.shape[0] and len(rows) > 100, (pattern, B, ML, len(rows), got.shape)
        for m, g in zip(rows, got):
            assert (m.StreamOffset, m.ChunkIndex) == (g[0], g[1])
            # m.caps are relative to chunk[searchPos:]; the C rows to the chunk: the match's own start ties the two together
            sp = int(g[3]) - m.caps[0]
            want = [c + sp if c >= 0 else -1 for c in m.caps]
            if comp.tdfa is None:
                want = [c + sp for c in m.caps]
            assert [int(x) for x in g[3:]] == want, (pattern, m.StreamOffset)
