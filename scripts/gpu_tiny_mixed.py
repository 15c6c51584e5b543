"""Timing of rgx_find_batch_device over a C3-shaped batch of short lines with long lines sprinkled in (the tiny kernel's group deferral,
DESIGN 5.6): ms per call for no long lines, one in 100003, one in 997, one in 50; RGX_NO_TINY=1 in the environment times the general
kernel on the same batches.  Usage: python scripts/gpu_tiny_mixed.py [nstr [every,every,... [lo,hi]]]  (lo,hi: the long lines' lengths)"""
import os, sys, time, random
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from regengo_amd import Compiled, synth

EMAIL = r"(?P<user>\w+)@(?P<domain>\w+)"
nstr = int(sys.argv[1]) if len(sys.argv) > 1 else 8_000_000
# RGX_FORCE_TDFA=1: the reference-mode program through the reference's Tagged DFA (Options.ForceTDFA)
# RGX_BATCH_LENS=lo,hi: every string's length drawn from U[lo,hi] (default: C3's U[8,40])
BLO, BHI = (int(x) for x in os.environ.get("RGX_BATCH_LENS", "8,40").split(","))
data, offs = synth.email_batch_np(nstr, seed=0x5EED0003, lo=BLO, hi=BHI)
lens = np.diff(offs).astype(np.int64)
rng = random.Random(5)
LO, HI = (int(x) for x in sys.argv[3].split(",")) if len(sys.argv) > 3 else (60, 200)
EVERY = tuple(int(x) for x in sys.argv[2].split(",")) if len(sys.argv) > 2 else (0, 100_003, 997, 50)
for every in EVERY:
    new_lens = lens.copy()
    if every:
        idx = np.arange(0, nstr, every)
        new_lens[idx] = np.array([rng.randrange(LO, HI + 1) for _ in idx], dtype=np.int64)
    noffs = np.zeros(nstr + 1, dtype=np.int64)
    np.cumsum(new_lens, out=noffs[1:])
    out = np.full(int(noffs[-1]), ord("x"), dtype=np.uint8)
    keep = new_lens == lens
    sel = np.repeat(keep, new_lens)
    out[sel] = data[np.repeat(keep, lens)]
    concat = torch.from_numpy(out).cuda()
    doffs = torch.from_numpy(noffs).cuda()
    for stdlib in (False, True):
        c = Compiled(EMAIL, stdlib=stdlib, force_tdfa=bool(os.environ.get("RGX_FORCE_TDFA")) and not stdlib).to(0)
        for _ in range(4):
            c.FindBatchDevice(concat, doffs)
        lvl = c.tuning()["batch_tiny_level"]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            found, spans = c.FindBatchDevice(concat, doffs)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 100
        alg = len(out) + 8 * (nstr + 1) + nstr + nstr * c.ncap * 4
        print("every=%-7d stdlib=%d  %.3f ms  %.1f GB/s of input  %.3f of HBM peak by algorithmic bytes  level=%d found=%d"
              % (every, stdlib, ms, len(out) / ms / 1e6, alg / ms / 1e6 / 8000.0, lvl, int(found.sum())), flush=True)
