"""rgx_match_batch_device (MatchBytes per string) over short strings and over lines: ms per call and GB/s of input for a few programs.
Usage: python scripts/gpu_match_batch.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from regengo_amd import Compiled, synth

PATS = {"email": r"(?P<user>\w+)@(?P<domain>\w+)", "date": r"(?P<year>\d{4})-(?P<month>\d{2})-(?P<day>\d{2})",
        "email_full": r"(?P<user>[\w\.+-]+)@(?P<domain>[\w\.-]+)\.(?P<tld>[\w\.-]+)", "anchored": r"^[a-z]+\d*$"}
for lo, hi, nstr in ((8, 40, 8_000_000), (8, 200, 2_000_000)):
    data, offs = synth.email_batch_np(nstr, seed=0x5EED0003, lo=lo, hi=hi)
    concat, doffs = torch.from_numpy(data).cuda(), torch.from_numpy(offs).cuda()
    for name, pat in PATS.items():
        for stdlib in (False, True):
            c = Compiled(pat, stdlib=stdlib).to(0)
            for _ in range(3):
                m = c.MatchBatchDevice(concat, doffs)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                m = c.MatchBatchDevice(concat, doffs)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) * 100
            print("U[%d,%d] %-10s stdlib=%d  %.3f ms  %.1f GB/s  matched=%d" % (lo, hi, name, stdlib, ms, len(data) / ms / 1e6, int(m.sum())), flush=True)
