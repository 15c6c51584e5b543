"""rgx_find_batch_device over LINES (tens to hundreds of bytes) for programs whose search automaton is not tiny: the general kernel
(batch_search_kernel) keeps a string's state trace in an LDS RING since round 6 -- the back-trace needs the rows between the match's
start and the walk's last byte only -- instead of writing the trace of every string beyond 62 bytes to memory.  Rows == the oracle's C
port of the emitted matcher (reference mode) and == Python's `re` (plain leftmost-first), for matches that fit the ring, matches longer
than the ring (the lane takes the trace in memory), walks that run on far behind a short match, and strings of every length mixed."""
import random
import re

import numpy as np
import pytest

DATE = r"(?P<year>\d{4})-(?P<month>\d{2})-(?P<day>\d{2})"
URL = r"(?P<full>(?P<proto>https?|ftp)://(?P<host>[\w.-]+)(?P<port>:\d+)?(?P<path>/[\w./-]*)?)"
EMAIL_FULL = r"(?P<user>[\w\.+-]+)@(?P<domain>[\w\.-]+)\.(?P<tld>[\w\.-]+)"
KV_REST = r"(?P<k>[a-z]+)=(?P<v>.*)"                     # the match runs to the end of the line
A_THEN_B = r"(?P<a>\d+) .*? (?P<b>end)"                   # lazy: the walk goes on far behind the first digit


@pytest.fixture(scope="module")
def torch_dev():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU; there is no CPU fallback")
    return torch


def _lines(nstr, lo, hi, seed):
    rng = random.Random(seed)
    words = [b"GET", b"POST", b"2024-01-15", b"1999-12-31", b"http://a.b-c.org/x/y.html", b"https://example.com:8080/" + b"p/" * 40,
             b"ftp://h.io", b"bob@example.com", b"first.last+tag@sub.domain.co.uk", b"id=42 rest of it", b"k=", b"7 apples and the end",
             b"12 " + b"x" * 90 + b" end", b"error", b"warn", b"--", b"::", b"http:/x", b"@", b"a@", b"user", b"took", b"ms", b"\xc3\xa9t\xc3\xa9"]
    out, offs = bytearray(), [0]
    for _ in range(nstr):
        n = rng.randrange(lo, hi + 1)
        line = bytearray()
        while len(line) < n:
            line += rng.choice(words) + b" "
        out += line[:n]
        offs.append(len(out))
    return np.frombuffer(bytes(out), dtype=np.uint8).copy(), np.array(offs, dtype=np.int64)


@pytest.mark.gpu
@pytest.mark.parametrize("pattern", [DATE, URL, EMAIL_FULL, KV_REST, A_THEN_B])
def test_general_kernel_on_lines(torch_dev, pattern):
    torch = torch_dev
    from oracle.gen_c import CMatcher
    from regengo_amd import Compiled, _capi
    cm = CMatcher(pattern)
    rx = re.compile(pattern.encode(), re.ASCII)
    for nstr, lo, hi, seed in ((40_000, 8, 200, 1), (20_000, 100, 400, 2), (30_000, 0, 70, 3), (3_000, 300, 1500, 4)):
        data, offs = _lines(nstr, lo, hi, seed)
        uoffs = offs.astype(np.uint64)
        exp_found = np.zeros(nstr, dtype=np.uint8)
        exp_spans = np.zeros((nstr, cm.ncap), dtype=np.int32)
        cm.lib.m_find_batch(data.ctypes.data, uoffs.ctypes.data, nstr, exp_found.ctypes.data, exp_spans.ctypes.data)
        concat, doffs = torch.from_numpy(data).cuda(), torch.from_numpy(offs).cuda()
        c = Compiled(pattern).to(0)
        try:
            found, spans = c.FindBatchDevice(concat, doffs)
        except _capi.RgxError as ex:
            # (reference mode bounds a string's length where the replay of its attempts is quadratic: a refusal, never a wrong row)
            assert ex.status == _capi.RGX_E_UNSUPPORTED and hi > 256, (pattern, lo, hi, str(ex))
        else:
            f, sp = found.cpu().numpy(), spans.cpu().numpy()
            assert np.array_equal(f, exp_found), (pattern, lo, hi, int((f != exp_found).sum()))
            m = exp_found.astype(bool)
            assert np.array_equal(sp[m], exp_spans[m]), (pattern, lo, hi)
        cs = Compiled(pattern, stdlib=True).to(0)
        found, spans = cs.FindBatchDevice(concat, doffs)
        f, sp = found.cpu().numpy(), spans.cpu().numpy()
        for i in range(0, nstr, 7):
            s = bytes(data[offs[i]:offs[i + 1]])
            if any(b >= 0x80 for b in s):
                continue                      # (Go's `.` and classes over UTF-8 against Python's bytes: not the same function)
            mm = rx.search(s)
            assert bool(f[i]) == (mm is not None), (pattern, i, s)
            if mm:
                assert (sp[i, 0], sp[i, 1]) == mm.span(), (pattern, i, s, sp[i].tolist())
                for nm, g in rx.groupindex.items():
                    a, b = mm.span(nm)
                    if a >= 0:
                        assert (sp[i, 2 * g], sp[i, 2 * g + 1]) == (a, b), (pattern, nm, i, s, sp[i].tolist())
