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
            c = Compiled(p).to(0)
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
