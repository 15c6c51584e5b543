"""Parity of chosen patterns on the web-log tile against the oracle's C port + their time per GiB:
python scripts/gpu_check_patterns.py <index in c5_counts.json or pattern> ..."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from regengo_amd import Compiled, synth
from oracle.gen_c import CMatcher
fx = json.load(open(os.path.join(ROOT, "tests", "golden", "c5_counts.json")))["patterns"]
tile = synth.web_log_tile(); tile = tile[:tile.rfind(b"\n") + 1]
for a in sys.argv[1:]:
    pat = fx[int(a)]["pattern"] if a.isdigit() else a
    c = Compiled(pat, stdlib=True).to(0); c.set_timing(True)
    ok = True
    for cut in (len(tile), 300_001, 70_000):
        data = (tile * 2)[:len(tile) + cut]
        exp, cnt = CMatcher(pat, q8=False).find_all_np(np.frombuffer(data, dtype=np.uint8).copy())
        spans, res = c.FindAllSpans(data)
        got = spans.cpu().numpy()
        if res.total != cnt or not np.array_equal(got, exp):
            ok = False
            bad = int(np.nonzero((got[:min(len(got), len(exp))] != exp[:min(len(got), len(exp))]).any(axis=1))[0][0]) if len(got) and len(exp) and (got[:min(len(got), len(exp))] != exp[:min(len(got), len(exp))]).any() else -1
            print("  MISMATCH", cut, res.total, cnt, bad, got[bad].tolist() if bad >= 0 else None, exp[bad].tolist() if bad >= 0 else None)
    nt = (1 << 30) // len(tile)
    big = torch.frombuffer(bytearray(tile), dtype=torch.uint8).to("cuda:0").repeat(nt).contiguous()
    cap = 300_000_000 // c.ncap
    out = torch.empty((cap, c.ncap), dtype=torch.int32, device="cuda:0")
    try:
        for _ in range(2):
            spans, res = c.FindAllSpans(big, out=out, capacity=cap)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3):
            spans, res = c.FindAllSpans(big, out=out, capacity=cap)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3 * 1e3
        print("%s parity=%s kernel=%d full_ms=%.3f scan_kernel_ms=%.3f matches=%d unsynced=%d  %s" % (a, ok, c.info.scan_kernel, dt, res.kernel_ms, res.total, res.unsynced, pat[:70]))
    except Exception as ex:
        print(a, "parity=%s" % ok, "ERROR", str(ex)[:200], pat[:60])
    del big, out
