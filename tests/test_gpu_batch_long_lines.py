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


@pytest.mark.gpu
def test_lines_of_random_patterns(torch_dev):
    """RANDOM patterns (tests/_fuzzgen.py: every engine class) over batches of LINES of 0-300 bytes, three calls each (a program learns
    its instance -- the register kernel's wide levels, the Tagged-DFA kernel's wide window -- from the first): found flags and records ==
    the oracle's C port of the matcher the reference emits (oracle/gen_c.py: m_find_batch; Tagged-DFA programs: oracle/tdfa_c.py), or a
    refusal -- never another answer.  The short random strings of tests/test_gpu_reference_mode.py never leave the narrow instances."""
    torch = torch_dev
    from oracle import engines as E
    from oracle.gen_c import CMatcher
    from oracle.tdfa_c import CTdfa
    from regengo_amd import Compiled, _capi
    from tests import _fuzzgen as F
    rng = random.Random(4242)
    pats = compared = refused = wide = tdfa_wide = 0
    for seed in F.fuzz_seeds(700, 702):
        for pat in F.gen_patterns(seed, 40):
            try:
                o = E.Compiled(pat)
            except Exception:
                continue
            if F.has_empty_loop(o.prog) and not o.find_machine.memo:
                continue
            if o.tdfa is not None and len(o.tdfa.states) > 200:
                continue
            try:
                c = Compiled(pat).to(0)
            except _capi.RgxError:
                continue
            if not c.info.ref_find_offered:
                continue
            try:
                port = CTdfa(pat) if c.info.ref_find_engine == 1 else CMatcher(pat)
            except Exception:
                continue
            nstr = 256 * 5 + 9
            # three kinds of batch in turn: lines of up to 250 bytes (the register kernel's wide levels), lines of up to 300 (a line beyond
            # the tag bytes in every group: the narrow level stays, the general kernel takes everything), short strings with a long line
            # in one of sixty (the groups that hold one are left to the general kernel, the rest stays in registers)
            kind = pats % 3
            if kind == 2:
                strings = [F.gen_input(rng, rng.randrange(60, 301) if rng.random() < 1 / 60 else rng.randrange(0, 41)) for _ in range(nstr)]
            else:
                top = 301 if kind else 251
                strings = [F.gen_input(rng, rng.randrange(0, top)) for _ in range(nstr)]
            data = np.frombuffer(b"".join(strings) + b"\0", dtype=np.uint8).copy()
            offs = np.zeros(nstr + 1, dtype=np.int64)
            np.cumsum([len(s) for s in strings], out=offs[1:])
            if c.info.ref_find_engine == 1:
                ef, er = port.find_batch_np(data, offs)
            else:
                uoffs = offs.astype(np.uint64)
                ef = np.zeros(nstr, dtype=np.uint8)
                er = np.zeros((nstr, port.ncap), dtype=np.int32)
                port.lib.m_find_batch(data.ctypes.data, uoffs.ctypes.data, nstr, ef.ctypes.data, er.ctypes.data)
            concat, doffs = torch.from_numpy(data).cuda(), torch.from_numpy(offs).cuda()
            pats += 1
            for call in range(3):
                try:
                    found, spans = c.FindBatchDevice(concat, doffs)
                except _capi.RgxError as ex:
                    assert ex.status == _capi.RGX_E_UNSUPPORTED, (pat, ex)
                    refused += 1
                    break
                f, sp = found.cpu().numpy(), spans.cpu().numpy()
                assert np.array_equal(f != 0, ef != 0), (pat, call, int(((f != 0) != (ef != 0)).sum()), c.tuning())
                m = ef.astype(bool)
                assert np.array_equal(sp[m], er[m]), (pat, call, c.tuning())
                compared += nstr
            t = c.tuning()
            wide += t["batch_tiny_level"] > 0
            tdfa_wide += t["batch_tdfa_wide"] > 0
    print("patterns", pats, "strings compared", compared, "refused", refused, "programs at a wide register level", wide, "Tagged-DFA programs at the wide window", tdfa_wide)
    if F.fuzz_default():
        assert pats >= 40 and compared >= 100_000 and wide >= 3 and tdfa_wide >= 1, (pats, compared, refused, wide, tdfa_wide)
