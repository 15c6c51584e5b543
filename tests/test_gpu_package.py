"""GPU tier: a package of patterns over one batch in one pass (rgx_multi_*, rgx_kernels.hip: batch_multi_kernel) against every
program's own rgx_find_batch_device -- found flags, counts, (start, end) -- in reference mode and in stdlib mode; the ^/$-anchored
patterns of the C5 suite over lines of the corpus plus lines planted to match them (the corpus itself matches none)."""
import json
import os
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bits_to_flags(bits, nstr):
    b = bits.cpu().numpy().view(np.uint64)
    out = np.zeros((b.shape[0], b.shape[1] * 64), dtype=np.uint8)
    for k in range(64):
        out[:, k::64] = ((b >> np.uint64(k)) & np.uint64(1)).astype(np.uint8)
    return out[:, :nstr]


def test_package_equals_every_programs_own_find_batch(built, corpus):
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU; there is no CPU fallback")
    from regengo_amd import Compiled, Package, _capi, synth
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "c5_counts.json")))
    ents = [e for e in fx["patterns"] if e["mode"] == "line"]
    by_pat = {e["pattern"]: e["inputs"] for e in corpus}
    rng = random.Random(9)
    tile = synth.web_log_tile(1 << 18)
    lines = [l for l in tile.split(b"\n") if l][:3000]
    progs = []
    for e in ents:
        stdlib = e.get("semantics") != "reference"
        try:
            c = Compiled(e["pattern"], stdlib=stdlib).to(0)
        except _capi.RgxError:
            continue
        progs.append(c)
        for s in by_pat.get(e["pattern"], [])[:3]:            # the reference's own test inputs: lines these validators DO match
            b = s.encode()
            if b"\n" not in b and len(b) < 400:
                lines.insert(rng.randrange(len(lines)), b)
    lines += [b"", b"x", b"2024-01-15", b"a" * 300]
    offs = np.zeros(len(lines) + 1, dtype=np.int64)
    offs[1:] = np.cumsum([len(l) for l in lines])
    concat = torch.frombuffer(bytearray(b"".join(lines) + b"\0" * 16), dtype=torch.uint8).to("cuda:0")
    offsets = torch.from_numpy(offs).to("cuda:0")
    pk = Package(progs)
    assert sum(pk.accepted) >= 0.9 * len(progs) and pk.launches >= 1
    bits, counts, se = pk.FindBatchBits(concat, offsets, want_se=True)
    flags = _bits_to_flags(bits, len(lines))
    se = se.cpu().numpy()
    counts = counts.cpu().tolist()
    nfound = 0
    for k, c in enumerate(progs):
        if not pk.accepted[k]:
            assert not flags[k].any() and counts[k] == 0
            continue
        try:
            found, spans = c.FindBatchDevice(concat, offsets)
        except _capi.RgxError as ex:
            assert ex.status == _capi.RGX_E_UNSUPPORTED
            continue
        f = found.cpu().numpy()
        assert np.array_equal(flags[k], f), (c.pattern, int(np.nonzero(flags[k] != f)[0][0]))
        assert counts[k] == int(f.sum())
        sp = spans.cpu().numpy()
        hit = np.nonzero(f)[0]
        assert np.array_equal(se[k][hit], sp[hit, :2]), c.pattern
        nfound += len(hit)
    assert nfound > 100          # the planted lines are found: the comparison is not vacuous


def test_package_on_random_patterns(built):
    """The package (many programs, one pass over the lines: rgx_multi_*) over RANDOM patterns, reference mode and stdlib mode: every
    accepted program's found flags, counts and (start, end) == its own FindBatch (which tests/test_gpu_reference_mode.py holds against
    the oracle on the same kind of patterns), and == the oracle directly on a sample."""
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU; there is no CPU fallback")
    from oracle import engines as E
    from regengo_amd import Compiled, Package, _capi
    from tests import _fuzzgen as F
    rng = random.Random(4)
    lines = [b"", b"a", b"\xc3\xa9 x"] + [F.gen_input(rng, rng.choice([1, 4, 12, 30, 70])) for _ in range(1500)]
    lines = [l.replace(b"\n", b" ") for l in lines]
    offs = np.zeros(len(lines) + 1, dtype=np.int64)
    offs[1:] = np.cumsum([len(l) for l in lines])
    concat = torch.frombuffer(bytearray(b"".join(lines) + b"\0" * 16), dtype=torch.uint8).to("cuda:0")
    offsets = torch.from_numpy(offs).to("cuda:0")
    compared = direct = 0
    for stdlib in (False, True):
        progs, oracles = [], []
        for seed in (100, 101):
            for pat in F.gen_patterns(seed, 60):
                try:
                    o = E.Compiled(pat)
                except Exception:
                    continue
                if F.has_empty_loop(o.prog) and not o.find_machine.memo:
                    continue
                try:
                    c = Compiled(pat, stdlib=stdlib).to(0)
                except _capi.RgxError:
                    continue
                progs.append(c)
                oracles.append(o)
        pk = Package(progs)
        assert sum(pk.accepted) >= len(progs) // 3 and pk.launches >= 1, (sum(pk.accepted), len(progs))
        bits, counts, se = pk.FindBatchBits(concat, offsets, want_se=True)
        flags = _bits_to_flags(bits, len(lines))
        se = se.cpu().numpy()
        counts = counts.cpu().tolist()
        for k, c in enumerate(progs):
            if not pk.accepted[k]:
                assert not flags[k].any() and counts[k] == 0
                continue
            try:
                found, spans = c.FindBatchDevice(concat, offsets)
            except _capi.RgxError as ex:
                assert ex.status == _capi.RGX_E_UNSUPPORTED
                continue
            f = found.cpu().numpy()
            assert np.array_equal(flags[k], f), (c.pattern, stdlib, int(np.nonzero(flags[k] != f)[0][0]))
            assert counts[k] == int(f.sum()), c.pattern
            sp = spans.cpu().numpy()
            hit = np.nonzero(f)[0]
            assert np.array_equal(se[k][hit], sp[hit, :2]), (c.pattern, stdlib)
            compared += 1
            o = oracles[k]
            if not stdlib and o.tdfa is None:
                for i in range(0, 60):
                    exp = o.FindBytes(lines[i])
                    assert bool(flags[k][i]) == (exp is not None), (c.pattern, lines[i])
                    if exp is not None:
                        assert list(se[k][i]) == exp[:2], (c.pattern, lines[i])
                    direct += 1
    assert compared >= 80 and direct >= 2000, (compared, direct)
