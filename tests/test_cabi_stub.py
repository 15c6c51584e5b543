"""tests/cabi_stub.c is the C twin of the emitted cgo file (tests/golden/codegen/*_gpu.go): same call sequence, compiled with
`gcc -std=c99 -Wall -Wextra -Werror -pedantic` against include/rgx.h and linked to librgx_hip.so -- the stand-in for `go build` in
an image without a Go toolchain.  CPU tier: it compiles, links, loads a blob and takes the no-device path (INIT -5: the Go path
stays).  GPU tier: the protocol against the oracle and the reference's literal streaming vectors."""
import json
import os
import subprocess

import pytest

from regengo_amd import codegen

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "tests", "_build")
DATE = r"(?P<year>\d{4})-(?P<month>\d{2})-(?P<day>\d{2})"
URL_CAPTURE = r"(?P<protocol>https?)://(?P<host>[\w\.-]+)(?::(?P<port>\d+))?(?P<path>/[\w\./]*)?"


@pytest.fixture(scope="module")
def stub(built):
    os.makedirs(BUILD, exist_ok=True)
    exe = os.path.join(BUILD, "cabi_stub")
    libdir = os.path.join(ROOT, "regengo_amd", "lib")
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "cabi_stub.c"), "-L" + libdir, "-lrgx_hip", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib",
           "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def _tables(tmp_path, pattern, name="P", flags=0):
    text, blob = codegen.emit_go(pattern, name, "p", flags=flags)
    path = os.path.join(str(tmp_path), name + "_tables.bin")
    open(path, "wb").write(blob)
    return path, text


def _run(exe, blob, data: bytes, tmp_path, *args, env=None):
    inp = os.path.join(str(tmp_path), "input.bin")
    open(inp, "wb").write(data)
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run([exe, blob, inp] + [str(a) for a in args], capture_output=True, timeout=600, env=e)
    return r.returncode, r.stdout


def test_twin_compiles_as_c99_and_uses_what_the_go_file_uses(stub, tmp_path):
    """Every rgx_* function the emitted Go calls is called by the twin too (so a signature change breaks a COMPILE here)."""
    import re
    _, go = _tables(tmp_path, DATE, "Date")
    go_syms = set(re.findall(r"C\.(rgx_[a-z_0-9]+)\(", go))
    c_syms = set(re.findall(r"\b(rgx_[a-z_0-9]+)\(", open(os.path.join(ROOT, "tests", "cabi_stub.c")).read()))
    # rgx_transform_* belong to stream.Transformer's processor, which has no C twin (the Transformer is Go's)
    missing = {s for s in go_syms - c_syms if not s.startswith("rgx_transform")}
    assert not missing, missing


def test_twin_without_a_device_keeps_the_go_path(stub, tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a device is present: the GPU tier runs the protocol")
    blob, _ = _tables(tmp_path, DATE, "Date")
    rc, out = _run(stub, blob, b"x 2024-01-15 y", tmp_path, "info")
    assert rc == 0
    lines = out.decode().splitlines()
    assert lines[0].startswith("INIT -5")                                # RGX_E_NO_DEVICE: <name>Prog stays nil, every method falls back
    assert "abi 5 ncap 8 min 10 max 10 findall 1 stream 1 find 1 match 1 engine 0 flags 0 replace 1" in lines[1]
    rc, out = _run(stub, blob, b"x 2024-01-15 y", tmp_path, "findall")
    assert rc == 1 and out.startswith(b"INIT -5")


@pytest.mark.gpu
def test_twin_find_all_with_capacity_retry(stub, tmp_path):
    from oracle.engines import Compiled as O
    blob, _ = _tables(tmp_path, DATE, "Date")
    # dense input: more than one match per 64 bytes + 1024 -> the first call answers RGX_E_CAPACITY, the retry has the exact count
    data = (b"2024-01-15 " * 40000) + b"tail 1999-12-31"
    rc, out = _run(stub, blob, data, tmp_path, "findall", -1)
    assert rc == 0
    lines = out.decode().splitlines()
    exp = O(DATE).FindAllBytes(data)
    # (the Date pattern is a fixed template: the stub fetches 4 bytes per match -- rgx_find_all_starts -- and rebuilds the records)
    assert lines[0] == "RETRY %d" % len(exp) and lines[1] == "STARTS %d" % len(exp) and lines[2] == "COUNT %d" % len(exp)
    assert [list(map(int, l.split()[1:])) for l in lines[3:]] == exp
    rc, out = _run(stub, blob, data, tmp_path, "findall", 5)
    assert [list(map(int, l.split()[1:])) for l in out.decode().splitlines() if l.startswith("ROW")] == exp[:5]
    # a pattern without a fixed template takes the full records
    blob2, _ = _tables(tmp_path, r"(?P<k>[a-z]+)=(?P<v>\d+)", "KV")
    d2 = b"alpha=1 beta=22 x= gamma=333 " * 2000
    rc, out = _run(stub, blob2, d2, tmp_path, "findall", -1)
    assert b"STARTS" not in out
    assert [list(map(int, l.split()[1:])) for l in out.decode().splitlines() if l.startswith("ROW")] == O(r"(?P<k>[a-z]+)=(?P<v>\d+)").FindAllBytes(d2)
    # several devices in the list (the same one twice on a one-GPU box): rgx_sharded_find_all_bytes, same rows
    rc, out = _run(stub, blob, data, tmp_path, "sharded", 2, -1)
    lines = out.decode().splitlines()
    assert "SHARDED 1" in lines
    rows = [list(map(int, l.split()[1:])) for l in lines if l.startswith("ROW")]
    assert rows == exp


@pytest.mark.gpu
def test_twin_find_reader_reproduces_the_reference_streaming_vector(stub, tmp_path, kats):
    """streaming_test.go:190-280: six dates at literal offsets around the 64 KiB chunk boundary; offsets and chunk indices equal
    the oracle's FindReader, for full reads and for a reader that returns 4000 bytes at a time."""
    from oracle import engines as E
    sb = kats["streaming_boundary"]
    data = bytearray(sb["fill"].encode() * sb["total_size"])
    for pos, d in zip(sb["positions"], sb["dates"]):
        data[pos:pos + len(d)] = d.encode()
    data = bytes(data)
    blob, _ = _tables(tmp_path, sb["pattern"], "D")
    o = E.Compiled(sb["pattern"])
    for read_size in (0, 4000):
        exp = []
        pos = [0]

        def read(n):
            k = min(n, read_size) if read_size else n
            b = data[pos[0]:pos[0] + k]
            pos[0] += len(b)
            return b

        assert o.FindReader(read, E.StreamConfig(BufferSize=sb["buffer_size"]), lambda m: exp.append((m.StreamOffset, m.ChunkIndex, m.match_bytes)) or True) is None
        rc, out = _run(stub, blob, data, tmp_path, "reader", sb["buffer_size"], 0, read_size)
        assert rc == 0
        got = [(int(l.split()[1]), int(l.split()[2]), l.split()[3].encode()) for l in out.decode().splitlines() if l.startswith("MATCH")]
        assert got == exp, read_size
        if not read_size:
            # (after a SHORT read the reference does not advance streamOffset, streaming.go:241-244: its offsets are then relative to
            # nothing useful -- 768 for the date at 32768 -- and the stub reproduces exactly that)
            assert [g[0] for g in got] == sb["positions"]
        rc, out = _run(stub, blob, data, tmp_path, "count", sb["buffer_size"], 0, read_size)
        assert out.decode().splitlines()[-1] == "COUNT %d" % len(exp)
    rc, out = _run(stub, blob, data, tmp_path, "reader", 100, 0, 0)           # cfg.Validate: stream.ErrBufferTooSmall
    assert out.decode().startswith("CONFIG_ERROR -10")


@pytest.mark.gpu
def test_twin_refusals_and_fallbacks(stub, tmp_path):
    from regengo_amd import _capi
    # Q4: an earlier copy of the match text in the gap -> bytes.Index finds it first -> RGX_E_DIVERGES: the chunk goes to the Go loop
    blob, _ = _tables(tmp_path, r"(?P<a>ab+)c?", "A")
    rc, out = _run(stub, blob, b"xx ab yy abbc ab", tmp_path, "reader", 65536, 0, 0)
    assert rc == 0
    # the reference's Tagged DFA (URLCapture): FindAll is the emitted WRAPPER's loop, which advances by the match LENGTH and so reports a
    # match behind the offset again (compiler.go:646-651) -- reproduced since round 5: here "https://example.com/a" (offset 4, 21 bytes)
    # is found from offsets 0 and 21 is past it; the rows are oracle.tdfa.find_all's, unset groups (-1, -1)
    from oracle import engines as E
    blob, go = _tables(tmp_path, URL_CAPTURE, "U")
    data = b"see https://example.com/a and http://h.org:80/x then https://plain.net"
    rc, out = _run(stub, blob, data, tmp_path, "findall", -1)
    exp_rows = E.Compiled(URL_CAPTURE).FindAllBytes(data)
    lines = out.decode().splitlines()
    assert rc == 0 and lines[0] == "COUNT %d" % len(exp_rows), lines[:2]
    assert [[int(x) for x in l.split()[1:]] for l in lines[1:1 + len(exp_rows)]] == exp_rows
    assert len(exp_rows) >= 3
    # ... Replace runs the engine's own loop over ONE result struct (round 5): a group the third match does not assign expands to what
    # the second left there -- as oracle.replace says
    from oracle import replace as R
    rc, out = _run(stub, blob, data, tmp_path, "replace", "<$host$port>", 0)
    exp = R.replace_all(E.Compiled(URL_CAPTURE), data, "<$host$port>", quirks=True)
    assert rc == 0 and out == b"OUT %d\n%s\nEND\n" % (len(exp), exp)
    assert b"<plain.net80>" in out
    # ... while FindBytes and FindReader run the engine itself: the third match has neither port nor path, and the reused result
    # struct still shows the second match's (tdfa.go:1031-1046 leaves the fields untouched) -- as oracle.find_reader(reuse=True) says
    o = E.Compiled(URL_CAPTURE)
    exp = []
    import io
    assert o.FindReader(io.BytesIO(data).read, E.StreamConfig(65536, 0), lambda m: exp.append((m.StreamOffset, m.match_bytes, m.fields)) or True) is None
    rc, out = _run(stub, blob, data, tmp_path, "reader", 65536, 0, 0)
    lines = out.decode().splitlines()
    got, cur = [], None
    for l in lines:
        if l.startswith("MATCH"):
            cur = (int(l.split()[1]), l.split()[3].encode(), [])
            got.append(cur)
        elif l.startswith("FIELD"):
            txt = l.split(" ", 2)[2] if len(l.split(" ", 2)) > 2 else ""
            cur[2].append(None if txt == "<nil>" else txt.encode())
    assert [(a, b, c) for a, b, c in got] == exp and len(exp) == 3
    assert exp[2][2][3] == b"80" and exp[2][2][4] == b"/x" and exp[0][2][3] is None          # port, path: stale; nil before any assignment
    rc, out = _run(stub, blob, data, tmp_path, "find")
    assert out.decode().strip() == "ROW " + " ".join(map(str, o.FindBytes(data)))
    # ... and everything is answered as Go's regexp would with tables compiled under RGX_FLAG_STDLIB_SEMANTICS
    from oracle.engines import Compiled as O
    blob, _ = _tables(tmp_path, URL_CAPTURE, "U2", flags=_capi.FLAG_STDLIB_SEMANTICS)
    rc, out = _run(stub, blob, data, tmp_path, "findall", -1)
    rows = [list(map(int, l.split()[1:])) for l in out.decode().splitlines() if l.startswith("ROW")]
    assert rows == O(URL_CAPTURE).FindAllLeftmostFirst(data) and len(rows) == 3


@pytest.mark.gpu
def test_twin_replace_match_find(stub, tmp_path):
    from oracle.engines import Compiled as O
    blob, _ = _tables(tmp_path, DATE, "Date")
    data = b"from 2024-01-15 to 2024-02-29."
    rc, out = _run(stub, blob, data, tmp_path, "replace", "$day/$month/$year", 0)
    assert out == b"OUT 30\nfrom 15/01/2024 to 29/02/2024.\nEND\n"
    rc, out = _run(stub, blob, data, tmp_path, "replace", "${", 0)
    assert out.startswith(b"PANIC invalid replace template")
    big = data + b"x" * 100 + (b" 2024-03-0%d" % 1) * 2000        # output longer than len + len/8 + 64: the capacity loop
    rc, out = _run(stub, blob, big, tmp_path, "replace", "<<<<<<<<<<<<<<<<<<<<$0>>>>>>>>>>>>>>>>>>>>", 0)
    assert out.startswith(b"RETRY ")
    o = O(DATE)
    for s in (b"x 12024-01-15", b"a 2024-01-15 b", b"nothing"):
        rc, out = _run(stub, blob, s, tmp_path, "match")
        assert out.decode().strip() == "MATCHED %d" % int(o.MatchBytes(s)), s      # reference semantics (Q1: 12024-01-15 does not match)
        rc, out = _run(stub, blob, s, tmp_path, "find")
        e = o.FindBytes(s)
        assert out.decode().strip() == ("NOTFOUND" if e is None else "ROW " + " ".join(map(str, e))), s


def _matches(out: bytes):
    got, cur = [], None
    for l in out.decode("latin-1").splitlines():
        if l.startswith("MATCH"):
            p = l.split(" ", 3)
            cur = (int(p[1]), int(p[2]), p[3], [])
            got.append(cur)
        elif l.startswith("FIELD"):
            cur[3].append(l.split(" ", 2)[2] if len(l.split(" ", 2)) > 2 else "")
    return got


@pytest.mark.gpu
@pytest.mark.parametrize("pattern,name", [(DATE, "Date"), (URL_CAPTURE, "U")])
def test_twin_find_reader_in_runs_of_chunks(stub, tmp_path, kats, pattern, name):
    """The emitted FindReader of round 6 (<name>ReadRuns + rgx_find_chunks): the same MATCH / FIELD lines as the chunk-by-chunk loop it
    replaces (`reader`: rgx_find_chunk per chunk), for full reads, for a reader that returns 4000 bytes at a time, for blocks of one,
    three and many chunks, over one device and cut across two (rgx_sharded_round with reader windows + rgx_sharded_gather) -- and the
    reference's literal streaming vector at its literal offsets."""
    from regengo_amd import synth
    sb = kats["streaming_boundary"]
    data = bytearray(sb["fill"].encode() * sb["total_size"])
    for pos, d in zip(sb["positions"], sb["dates"]):
        data[pos:pos + len(d)] = d.encode()
    blob, _ = _tables(tmp_path, pattern, name)
    for what, stream in (("vector", bytes(data)), ("weblog", synth.web_log_tile(1 << 20)[:700001])):
        if what == "vector" and pattern != DATE:
            continue
        for bufsize, ml in ((65536, 0), (70000, 20000)):
            for read_size in (0, 4000):
                rc, ref = _run(stub, blob, stream, tmp_path, "reader", bufsize, ml, read_size)
                assert rc == 0 and b"GOFALLBACK" not in ref
                want = _matches(ref)
                assert len(want) > 3
                for block, ndev in ((1, 1), (200000, 1), (64 << 20, 1), (64 << 20, 2), (300000, 2)):
                    rc, out = _run(stub, blob, stream, tmp_path, "runs", bufsize, ml, read_size, block, ndev)
                    assert rc == 0 and b"GOFALLBACK" not in out, out[-200:]
                    assert _matches(out) == want, (what, bufsize, ml, read_size, block, ndev)
                    assert out.splitlines()[-1] == ref.splitlines()[-1]                       # COUNT
                    if ndev == 2 and read_size == 0 and block > (1 << 20) and len(stream) > 4 * bufsize:
                        assert b"SHARDEDRUN 2" in out
                rc, out = _run(stub, blob, stream, tmp_path, "countruns", bufsize, ml, read_size, 64 << 20, 1)
                assert out.splitlines()[-1] == ref.splitlines()[-1]
        if what == "vector" and pattern == DATE:
            rc, out = _run(stub, blob, stream, tmp_path, "runs", sb["buffer_size"], 0, 0, 64 << 20, 1)
            assert [m[0] for m in _matches(out)] == sb["positions"]
