"""GPU tier: the filter + candidate kernel (csrc/rgx_scan_fc.hip) against the oracle's C restatement of the reference's FindAllBytes
(find.go:130-466; oracle/gen_c.py), on inputs built to hit what is special about it: candidates at every offset around a tile border
(tiles own 16064 bytes), matches that run out of the tile's window (the walk goes on in memory), candidates inside an earlier
candidate's match (the chain's serial path), more candidates in a tile than it has lanes (the second round; beyond that the launch
gives up and the program's other kernel answers), a halo without a reset byte (gives up too), groups assigned by a continuation that
did not match (`host:` without a digit: resolved again out of the tables in memory), owned ranges, counts, texts without a match."""
import random
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

URL = r"(?P<full>(?P<proto>https?|ftp)://(?P<host>[\w.-]+)(?P<port>:\d+)?(?P<path>/[\w./-]*)?)"
TILE = 16064
CASES = [
    # (pattern, alphabet of the random texts, words planted in them)
    (URL, "htps:/f.w-1 \n", [b"http://a.b/c", b"https://w-1.f:11/p/", b"ftp://h", b"http://w:x", b"http:/w", b"https://f.f:1"]),
    (r"https?://[^\s]+", "htps:/x \n", [b"http://x/", b"https://ss", b"http:/x"]),
    (r"(GET|POST) (/\S*)", "GETPOS /x\n", [b"GET /x", b"POST /", b"GET  /", b"POST /xx/x"]),
    (r"\[(INFO|WARN)\]", "[]INFOWAR x\n", [b"[INFO]", b"[WARN]", b"[INFO", b"[WARN]]"]),
    (r"(?P<k>id|took)=(?P<v>\w*)", "idtok=_9 \n", [b"id=9", b"took=", b"took=_9k", b"id ="]),
    (r"admin@(\w+)\.(\w+)", "admin@.w \n", [b"admin@w.w", b"admin@a.", b"admin@mm.nn.w"]),
]


# patterns WITHOUT a reset byte (some thread survives any byte): the tiles' chains are joined by the look-back (the end of a tile's last
# match travels with its count), a tile that finds an earlier match reaching past its first candidate gives the call up
CARRY_CASES = [
    (r'\[(?P<d>[^\]]+)\] "(?P<m>[^"]+)"', '[]" ab\n', [b'[a b] "x"', b'[ab] "', b'[] "x"', b'[a]  "x"', b'[a\nb] "q\nr"']),
    (r'GET (?P<p>[^?]+)\?(?P<q>[^#]+)#', 'GET ?#ab\n', [b'GET a?b#', b'GET ?b#', b'GET a b?\n#', b'GET a?#', b'GET a??b##']),
    (r'\[(?P<ts>[^\]]+)\]\s+(?P<lvl>[A-Z]+):\s+(?P<msg>[^;]+);', '[]A: ;x\n', [b'[x] A: x;', b'[x]A: x;', b'[x x]\nAA:  x x;', b'[] A: x;', b'[x] A:;', b'[[x] A: ];']),
]


@pytest.fixture(scope="module")
def torch_dev(built):
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU; there is no CPU fallback")
    return torch


def _texts(rng, alphabet, words, sizes):
    out = []
    for n in sizes:
        k = rng.random()
        parts, size = [], 0
        while size < n:
            r = rng.random()
            if r < (0.15 if k < 0.5 else 0.6):
                w = rng.choice(words)
            elif r < 0.8:
                w = "".join(rng.choice(alphabet) for _ in range(rng.randrange(1, 12))).encode()
            else:
                w = (rng.choice(alphabet) * rng.choice([1, 3, 70, 300])).encode()
            parts.append(w + (b" " if rng.random() < 0.7 else b""))
            size += len(parts[-1])
        out.append(b"".join(parts)[:n])
    return out


def _check(c, cm, b, torch, what=""):
    arr = np.frombuffer(b, dtype=np.uint8).copy() if b else np.zeros(0, dtype=np.uint8)
    exp, cnt = cm.find_all_np(arr)
    spans, res = c.FindAllSpans(b)
    got = spans.cpu().numpy()
    assert res.total == cnt and got.shape == exp.shape and np.array_equal(got, exp), (c.pattern, len(b), cnt, int(res.total), what)
    if len(b):
        n, _ = c.CountAll(torch.from_numpy(arr).cuda())
        assert n == cnt, (c.pattern, len(b), what, "count")
    return exp


@pytest.mark.parametrize("pattern,alphabet,words", CASES)
def test_fc_kernel_equals_oracle(torch_dev, pattern, alphabet, words):
    from oracle.gen_c import CMatcher
    from regengo_amd import Compiled
    c = Compiled(pattern).to(0)
    assert c.info.scan_kernel == 7, (pattern, c.info.scan_kernel)
    cm = CMatcher(pattern, q8=False)
    rng = random.Random(zlib.crc32(pattern.encode()) & 0xFFFF)
    sizes = [64, 65, 127, 1000, TILE - 1, TILE, TILE + 1, TILE + 255, TILE + 257, 2 * TILE, 2 * TILE + 63, 40000, 70001, 200000, 1 << 20]
    for b in _texts(rng, alphabet, words, sizes) + [b"", b"a", words[0], words[0] * 3000, (alphabet[0] * 5000).encode()]:
        exp = _check(c, cm, b, torch_dev)
        if len(b) >= 1000:                       # shard ownership: only matches that START inside [lo, hi)
            lo, hi = len(b) // 3, 2 * len(b) // 3 + 1
            own, _ = c.FindAllSpans(b, own=(lo, hi))
            keep = exp[(exp[:, 0] >= lo) & (exp[:, 0] < hi)]
            assert np.array_equal(own.cpu().numpy(), keep), (pattern, len(b), "owned range")


def test_fc_candidates_around_tile_borders(torch_dev):
    """A match planted at every offset around the first two tile borders and around the end of the text, with the byte in front of it
    a reset byte or not; the URL's groups come out of the candidate's walk (mode 2)."""
    from oracle.gen_c import CMatcher
    from regengo_amd import Compiled
    c = Compiled(URL).to(0)
    cm = CMatcher(URL, q8=False)
    unit = b"https://w-1.f:80/p/q.r"
    for border in (TILE, 2 * TILE):
        for shift in range(-len(unit) - 2, 4):
            for tail in (0, 1, 7, 300):
                buf = bytearray(b"x " * ((border + 400) // 2 + 1))[: border + len(unit) + tail]
                at = border + shift
                buf[at:at + len(unit)] = unit
                _check(c, cm, bytes(buf[: max(at + len(unit), border + tail)]), torch_dev, (border, shift, tail))


def test_fc_walk_leaves_the_window_and_chain_conflicts(torch_dev):
    from oracle.gen_c import CMatcher
    from regengo_amd import Compiled
    c = Compiled(URL).to(0)
    cm = CMatcher(URL, q8=False)
    # a match that starts near the end of a tile and runs for kilobytes (under the walk's step bound): out of the window, on in memory
    long_host = b"http://" + b"w" * 1500 + b"/p"
    buf = b"x " * ((TILE - 40) // 2) + long_host + b" tail http://a/b "
    _check(c, cm, buf, torch_dev, "long walk")
    # candidates inside an earlier candidate's match: `http://a/http://b` -- the path takes `/http`, `://b` is no match
    nested = (b"http://a/http://b http://c/d/ftp://e " * 700)
    _check(c, cm, nested, torch_dev, "nested candidates")
    # ... across a wave's and a tile's border, densely
    _check(c, cm, (b"http://a/http://b/http://c/http://d " * 1200), torch_dev, "nested, dense")
    # groups assigned by a continuation that does not match: `host:` without a digit, `host:1x`
    _check(c, cm, (b"http://w.w:x http://w.w: http://w.w:1x/ http://w.w:12 " * 900), torch_dev, "dirty slots")


def test_fc_gives_up_and_the_other_kernel_answers(torch_dev):
    """More candidates in a tile than two rounds of lanes (512), a candidate that walks further than the bound, a tile whose halo holds
    no reset byte: the launch is void, the program's other kernel answers the call (and the program stays with it from the second time)."""
    from oracle.gen_c import CMatcher
    from regengo_amd import Compiled
    c = Compiled(URL).to(0)
    cm = CMatcher(URL, q8=False)
    dense = b"http://a " * 6000                         # a candidate every 9 bytes: ~1800 per tile
    _check(c, cm, dense, torch_dev, "dense")
    c2 = Compiled(URL).to(0)
    very_long = b"x " * 9000 + b"http://" + b"w" * 6000 + b" end http://a/b"
    _check(c2, cm, very_long, torch_dev, "walk bound")
    c3 = Compiled(URL).to(0)
    no_reset = b"http://a.b " + b"w" * 40000 + b" http://c.d/e " * 50       # 40 KB without a byte that resets the automaton
    _check(c3, cm, no_reset, torch_dev, "no sync point")
    # between one and two rounds: 257..512 candidates in a tile
    c4 = Compiled(URL).to(0)
    for gap in (36, 44, 60):
        unit = b"http://a.b/c" + b" " * (gap - 12)
        _check(c4, cm, unit * (3 * TILE // gap), torch_dev, ("two rounds", gap))


def test_fc_on_a_context_shared_with_other_kernels(torch_dev):
    """One context serves several programs in turn (a suite: rgx_stream_ctx_rebind): the exact kernel cleans the scratch set of the NEXT
    scan for as many descriptors as IT uses -- a scan of this kernel behind it needs sixteen times as many zeroed.  [Round 5: the
    descriptor count of the filter + candidate kernel was added to every program's, the exact kernel's scans then vouched for words they
    had not cleared, and the C5 suite's pattern behind the Date pattern placed its rows by stale counts.]"""
    from oracle.gen_c import CMatcher
    from regengo_amd import Compiled, synth
    date = Compiled(r"(?P<year>\d{4})-(?P<month>\d{2})-(?P<day>\d{2})").to(0)
    url = Compiled(URL).to(0, ctx_of=date)
    mail = Compiled(r"(?P<user>\w+)@(?P<domain>\w+)").to(0, ctx_of=date)
    tile = synth.web_log_tile()
    data = torch_dev.frombuffer(bytearray(tile * 24), dtype=torch_dev.uint8).cuda()
    exp = {}
    for c in (date, url, mail):
        exp[c.pattern], _ = CMatcher(c.pattern, q8=False).find_all_np(np.frombuffer(tile * 24, dtype=np.uint8))
    for order in ((date, url, mail, url, date, mail), (url, date, url, mail, date, url)):
        for c in order:
            for _ in range(2):
                got = c.FindAllSpans(data)[0].cpu().numpy()
                assert np.array_equal(got, exp[c.pattern]), c.pattern


def test_fc_text_without_a_match(torch_dev):
    from regengo_amd import Compiled
    c = Compiled(URL).to(0)
    data = torch_dev.full((96 << 20,), ord("x"), dtype=torch_dev.uint8, device="cuda")
    data[::61] = ord(" ")
    spans, res = c.FindAllSpans(data)
    assert res.total == 0 and spans.shape[0] == 0
    n, _ = c.CountAll(data)
    assert n == 0


@pytest.mark.parametrize("pattern,alphabet,words", CARRY_CASES)
def test_fc_patterns_without_a_reset_byte(torch_dev, pattern, alphabet, words):
    from oracle.gen_c import CMatcher
    from regengo_amd import Compiled
    c = Compiled(pattern).to(0)
    assert c.info.scan_kernel == 7, (pattern, c.info.scan_kernel)
    assert not any(c.reset_bytes()), pattern
    cm = CMatcher(pattern, q8=False)
    rng = random.Random(zlib.crc32(pattern.encode()) & 0xFFFF)
    sizes = [64, 65, 1000, TILE - 1, TILE, TILE + 1, 2 * TILE + 63, 40000, 70001, 200000, 1 << 20, 3 << 20]
    for b in _texts(rng, alphabet, words, sizes) + [b"", words[0], words[0] * 3000]:
        _check(c, cm, b, torch_dev)
    # a match that runs across a tile border (and one across several tiles, past the first candidates of the tiles behind it: the call
    # is given up, the program's other kernel answers) -- every time with a fresh program: one that gave up twice stays with the other kernel
    for gap in (10, 300, 2000):
        for at in (TILE - 5, TILE - gap // 2, 2 * TILE - 1):
            c2 = Compiled(pattern).to(0)
            w = words[0]
            cut = max(1, len(w) // 2)
            body = (alphabet[-2] * gap).encode()                  # a byte the open part of the match runs over
            buf = bytearray(b" " * (3 * TILE))
            span = w[:cut] + body + w[cut:]
            buf[at - cut:at - cut + len(span)] = span
            for k in range(0, 3 * TILE - 64, 997):                # short matches everywhere else
                if not (at - cut - len(w) - 2 < k < at - cut + len(span) + 2):
                    buf[k:k + len(w)] = w
            _check(c2, cm, bytes(buf), torch_dev, ("straddle", gap, at))
