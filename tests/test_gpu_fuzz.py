"""GPU tier: differential test over seeded random regular expressions (tests/_fuzzgen.py) -- FindAllBytes through the C
ABI against the oracle's C restatement, on inputs from 0 bytes to several hundred KiB (many tiles, slices without sync
points, long matches across tile borders).  Q8 (stale memo entries between FindAll iterations) is switched off in the
oracle here: the kernels compute the fresh-search reading."""
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_dev(built):
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU; there is no CPU fallback")
    return torch


@pytest.mark.parametrize("seed", [11, 12, 13])
def test_random_patterns_bit_exact(torch_dev, seed):
    from oracle import engines as E
    from oracle.gen_c import CMatcher
    from regengo_amd import Compiled, _capi
    from tests import _fuzzgen as F
    torch = torch_dev
    rng = random.Random(seed)
    compared = refused = hangs = unsynced = batch = 0
    for p in F.gen_patterns(seed, 45):
        try:
            o = E.Compiled(p)
        except Exception:
            continue
        if F.has_empty_loop(o.prog) and not o.find_machine.memo:
            hangs += 1                     # the reference's own Find* would not terminate on this pattern
            continue
        try:
            c = Compiled(p, stdlib=True).to(0)      # the batch rows below are compared with the plain search
        except _capi.RgxError:
            refused += 1
            continue
        cm = CMatcher(p, q8=False)
        for n in (0, 1, 63, 64, 65, 300, 4095, 20000, rng.choice([150000, 400000])):
            b = F.gen_input(rng, n)
            arr = np.frombuffer(b, dtype=np.uint8).copy() if n else np.zeros(0, dtype=np.uint8)
            if n >= 20000 and cm.memo:
                continue                   # the oracle's memo table is NINST x len bits per call: keep it small
            exp, cnt = cm.find_all_np(arr)
            spans, res = c.FindAllSpans(b)
            got = spans.cpu().numpy()
            assert res.total == cnt and got.shape == exp.shape and np.array_equal(got, exp), (p, n, cnt, int(res.total))
            unsynced += int(res.unsynced)
            compared += 1
        # the batch kernels (one string per lane, search automaton / restart loop): first match of each string
        strs = [F.gen_input(rng, rng.randrange(0, 70)) for _ in range(300)]
        offs = np.zeros(len(strs) + 1, dtype=np.int64)
        offs[1:] = np.cumsum([len(x) for x in strs])
        concat = np.frombuffer(b"".join(strs) or b"\0", dtype=np.uint8).copy()
        found, bspans = c.FindBatchDevice(torch.from_numpy(concat).cuda(), torch.from_numpy(offs).cuda())
        mt = c.MatchBatchDevice(torch.from_numpy(concat).cuda(), torch.from_numpy(offs).cuda()).cpu().numpy()
        found, bspans = found.cpu().numpy(), bspans.cpu().numpy()
        assert np.array_equal(mt, found), p
        for i, x in enumerate(strs):
            arr = np.frombuffer(x, dtype=np.uint8).copy() if x else np.zeros(0, dtype=np.uint8)
            exp, cnt = cm.find_all_np(arr, n=1)
            if len(x) == 0:
                continue                   # FindAllBytes makes no attempt on an empty input (Q3); the batch entry point does
            assert bool(found[i]) == (cnt > 0), (p, x)
            if cnt:
                assert bspans[i].tolist() == exp[0].tolist(), (p, x)
            batch += 1
    print("seed", seed, "compared", compared, "refused", refused, "non-terminating in the reference", hangs, "unsynced slices", unsynced, "batch strings", batch)
    assert compared > 200 and refused <= 5


def test_random_patterns_replace_and_transform(torch_dev):
    """The Replace and Transform rows on random patterns: ReplaceAllBytes / ReplaceFirstBytes against oracle/replace.py and
    ReplaceReader / SelectReader / RejectReader (device splice, small buffers) against oracle/transform.py, both read
    quirk-free.  Patterns that can match empty or look at their context are left out (Q12/Q13)."""
    from oracle import engines as E
    from oracle import replace as R
    from oracle import transform as T
    from regengo_amd import Compiled, _capi
    from regengo_amd.stream import Config
    from tests import _fuzzgen as F
    rng = random.Random(321)
    templates = ["", "<$0>", "[$1|$2]", "$$x${1}y", "$g7$0$0"]
    done = answered = refused = 0
    for p in F.gen_patterns(4242, 120):
        if any(a in p for a in ("^", "$", "\\b", "\\B")):
            continue
        try:
            o = E.Compiled(p)
            c = Compiled(p, stdlib=True).to(0)      # the quirk-free reading: no identical-or-refused check
            cref = Compiled(p).to(0)                # reference mode: the emitted loop's result, or RGX_E_DIVERGES
        except Exception:
            continue
        if c.info.can_match_empty or (F.has_empty_loop(o.prog) and not o.find_machine.memo):
            continue
        names = {n: i for i, n in enumerate(c.names) if i and n} if hasattr(c, "names") else {}
        for _ in range(3):
            b = F.gen_input(rng, rng.choice([0, 7, 300, 2500]))
            for tmpl in templates:
                assert c.ReplaceAllBytes(b, tmpl) == R.replace_all(o, b, tmpl), (p, tmpl, b)
            if cref.info.ref_replace_offered:
                try:
                    got = cref.ReplaceAllBytes(b, "<$0>")
                    try:
                        assert got == R.replace_all(o, b, "<$0>", quirks=True), (p, b)
                        if o.tdfa is None:
                            assert got == R.replace_all(o, b, "<$0>"), (p, b)      # answered: the quirks did not bite (Tagged-DFA programs: their own matches)
                    except NotImplementedError:
                        pass
                    answered += 1
                except _capi.RgxError as ex:
                    assert ex.status == _capi.RGX_E_DIVERGES, (p, b, ex)
                    refused += 1
            assert c.ReplaceFirstBytes(b, "<$0>") == R.replace_all(o, b, "<$0>", first_only=True), (p, b)
            # streaming: buffers just above the pattern's MaxLeftover so that many chunks are cut
            dl = c.info.default_max_leftover
            bs = dl + 200 if dl < (1 << 20) else 700
            ml = 0 if dl < (1 << 20) else 100
            for tmpl in ("", "<$0>"):
                want = T.replace_reader(o, T.bytes_reader(b), tmpl, quirks=False, buffer_size=bs, max_leftover=ml)
                wout, werr = want.read_all(333)
                got = c.ReplaceReader(b, tmpl, Config(bs, ml))
                if werr is not None:
                    continue
                assert got.read_all() == wout, ("reader", p, tmpl, bs, ml, b)
        done += 1
    print("patterns", done, "reference mode: answered", answered, "refused", refused)
    assert done >= 40 and answered > 40


@pytest.mark.parametrize("pat", [r"(?:a|\B)", r"(?:b|(?:01aa-)?|\B)", r"(?:ab)?\b", r"(\w+|\B)", r"(?:0+|\b)"])
def test_lookahead_empty_matches_regression(torch_dev, pat):
    """Empty matches that come from an empty-width assertion (no start_accept in the tables): a match and the empty match
    right behind it share their end -- the generic kernel's start/end pairing has to give up there (found by the sweep,
    scripts/gpu_fuzz_sweep.py; a wrong end used to reach the capture kernel)."""
    from oracle.gen_c import CMatcher
    from regengo_amd import Compiled
    from tests import _fuzzgen as F
    c = Compiled(pat).to(0)
    cm = CMatcher(pat, q8=False)
    rng = random.Random(9)
    for n in (64, 1000, 70000):
        b = F.gen_input(rng, n)
        arr = np.frombuffer(b, dtype=np.uint8).copy()
        if n >= 20000 and cm.memo:
            continue
        exp, cnt = cm.find_all_np(arr)
        spans, res = c.FindAllSpans(b)
        assert res.total == cnt and np.array_equal(spans.cpu().numpy(), exp), (pat, n)


def test_long_match_across_a_tile_edge_with_the_sync_automaton(torch_dev):
    """Regression (round-3 fuzz sweep, seed 1023; profiles/HISTORY.md section 6b).  `[^a]a{1,2}[^a]+` has no reset byte: the generic kernel takes its sync points from
    the sync automaton, and when the four staged halo slices hold none -- here a match of 474 bytes crosses the first tile edge --
    from the far look-behind, IN FRONT of the staged window.  The Shift-And prefilter then read LDS in front of the window and the
    lane behind the match reported [16429, 16461] for [16460, 16532].  (The sync automaton itself was sound: tests/_hosttest w_sync
    shows no sync point inside the match for any blind start.)"""
    import os
    from oracle.gen_c import CMatcher
    from regengo_amd import Compiled
    p = r"[^a]a{1,2}[^a]+"
    t = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "regress", "seed1023_head.bin"), "rb").read()[:16600]
    exp, cnt = CMatcher(p, q8=False).find_all_np(np.frombuffer(t, dtype=np.uint8).copy())
    spans, res = Compiled(p, stdlib=True).to(0).FindAllSpans(t)
    assert res.total == cnt and np.array_equal(spans.cpu().numpy(), exp)
