"""GPU tier: broken UTF-8 through the C ABI (the input screen + sanitised copy of programs whose classes hold U+FFFD) against the
oracle's machine on the original bytes: FindAllBytes over fuzzed buffers large enough for the tiled kernels, MatchBytes, the
batch entry points (a sequence must not borrow continuation bytes from the next string), Replace (output bytes come from the
ORIGINAL input) and FindReader chunks."""
import io
import random

import pytest

pytestmark = pytest.mark.gpu

from tests.test_broken_utf8 import PATTERNS, PIECES, _fuzz      # noqa: E402


@pytest.fixture(scope="module")
def gpu(built):
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU; there is no CPU fallback")
    return torch


def test_find_all_on_broken_utf8(gpu):
    from oracle import engines as E
    from regengo_amd import Compiled
    rng = random.Random(99)
    checked = 0
    for pat in PATTERNS:
        o = E.Compiled(pat)
        c = Compiled(pat, stdlib=True).to(0)
        for n in (1, 7, 40, 300, 3000):
            b = _fuzz(rng, n)
            exp = [list(r) for r in o.FindAllBytes(b)]
            got = c.FindAllSpans(b)[0].cpu().tolist()
            assert got == exp, (pat, n, b[:80])
            assert c.CountAll(b)[0] == len(exp)
            checked += 1
    assert checked == len(PATTERNS) * 5


def test_batch_and_match_on_broken_utf8(gpu):
    from oracle import engines as E
    from regengo_amd import Compiled
    rng = random.Random(5)
    for pat in [r"[^a]+", r"\W+", r"(?P<k>[^=]+)=(?P<v>[^;]*);", r"x[^y]y", r"[^a-z]é[^a-z]", r"\P{L}+"]:
        o = E.Compiled(pat)
        c = Compiled(pat, stdlib=True).to(0)
        strs = [_fuzz(rng, rng.randrange(0, 12)) for _ in range(700)]
        # strings that END in a truncated sequence right in front of a string that BEGINS with continuation bytes
        strs += [b"x\xe2\x82", b"\xac y", b"a\xf0\x9f", b"\x98\x80", b"\xc3", b"\xa9"] * 20
        offs = [0]
        for x in strs:
            offs.append(offs[-1] + len(x))
        concat = gpu.frombuffer(bytearray(b"".join(strs)), dtype=gpu.uint8).to("cuda:0")
        offsets = gpu.tensor(offs, dtype=gpu.int64, device="cuda:0")
        found, spans = c.FindBatchDevice(concat, offsets)
        found, spans = found.cpu().tolist(), spans.cpu().tolist()
        matched = c.MatchBatchDevice(concat, offsets).cpu().tolist()
        for i, s in enumerate(strs):
            e = [list(x) for x in o.FindAllBytes(s, 1)]
            # FindBytes also tries at offset len (find.go:545-569): an empty match there is not a FindAll match
            if e:
                assert found[i] and spans[i] == e[0], (pat, s, spans[i], e)
            elif found[i]:
                assert spans[i][0] == spans[i][1] == len(s), (pat, s, spans[i])
            assert bool(matched[i]) == bool(found[i]), (pat, s)
        for s in strs[:60]:
            assert c.MatchBytes(s) == (len(o.FindAllBytes(s, 1)) > 0 or c.info.can_match_empty == 1), (pat, s)


def test_replace_copies_original_bytes(gpu):
    from oracle import engines as E
    from oracle import replace as R
    from regengo_amd import Compiled
    rng = random.Random(11)
    pat = r"(?P<k>[^=;]+)=(?P<v>[^;]*);"
    o = E.Compiled(pat)
    c = Compiled(pat).to(0)
    for n in (5, 60, 900):
        b = _fuzz(rng, n) + b"k\xe2\x82=v\xf0\x9f;"
        assert c.ReplaceAllBytes(b, "<$v|$k>") == R.replace_all(o, b, "<$v|$k>"), b[:60]
