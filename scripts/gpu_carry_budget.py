"""Texts that keep a pattern's attempts running (no reset byte, no sync point): which path each takes, how long the call holds the
device, and whether the serial carry pass's step budget turns the quadratic cases into a refusal.
python scripts/gpu_carry_budget.py [size]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from regengo_amd import Compiled, _capi
from oracle.gen_c import CMatcher
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
CASES = [
    (r"[^q]{1,200}z", b"c"),
    (r"(?s).{1,250}z", b"c"),
    (r"[^q]+z", b"c"),
    (r"\pL{1,100}9", b"c"),
    (r"\pL+9", b"c"),
    (r"(\w+\s+){5}z", b"ab "),
    (r"\b[^q]+z\b", b"c"),
    (r"(?:[a-c]+\s?)+z", b"abc "),          # (the oracle's backtracker is exponential here: no CPU count)
]
NO_CPU = {r"(?:[a-c]+\s?)+z"}
for pat, unit in CASES:
    data = (unit * (n // len(unit) + 1))[:n]
    try:
        c = Compiled(pat, stdlib=True).to(0)
    except Exception as ex:
        print("%-22s compile: %s" % (pat, str(ex)[:80])); continue
    t0 = time.perf_counter()
    try:
        spans, res = c.FindAllSpans(data)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        exp, cnt = CMatcher(pat, q8=False).find_all_np(np.frombuffer(data, dtype=np.uint8).copy()) if n <= 1 << 17 and pat not in NO_CPU else (None, None)
        print("%-22s kernel=%d total=%d (cpu %s) unsynced=%d  %.3f s" % (pat, c.info.scan_kernel, res.total, cnt, res.unsynced, dt), flush=True)
    except _capi.RgxError as ex:
        print("%-22s kernel=%d REFUSED status=%d after %.3f s: %s" % (pat, c.info.scan_kernel, ex.status, time.perf_counter() - t0, str(ex)[:90]), flush=True)
