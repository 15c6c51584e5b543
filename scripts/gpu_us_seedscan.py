"""tests/test_gpu_us.py seeds its texts with hash(pattern), which changes from process to process: this scans fixed seeds for a pattern
and keeps the first text on which the device and the oracle disagree (gpurun_out/us_fail_<seed>.bin).
python scripts/gpu_us_seedscan.py '<pattern>' '<alphabet>' first n"""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from oracle.gen_c import CMatcher
from regengo_amd import Compiled, _capi
from tests.test_gpu_us import _texts
pattern, alphabet = sys.argv[1], sys.argv[2].encode().decode("unicode_escape")
first, n = int(sys.argv[3]), int(sys.argv[4])
c = Compiled(pattern).to(0)
cm = CMatcher(pattern, q8=False)
sizes = [64, 65, 127, 128, 129, 1000, 16383, 16384, 16385, 16384 + 255, 16384 + 257, 32768, 40000, 70001, 200000]
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
bad = 0
for seed in range(first, first + n):
    rng = random.Random(seed)
    for b in _texts(rng, alphabet, sizes):
        arr = np.frombuffer(b, dtype=np.uint8).copy()
        exp, cnt = cm.find_all_np(arr)
        try:
            spans, res = c.FindAllSpans(b)
        except _capi.RgxError as ex:
            print("seed", seed, "len", len(b), "REFUSED", str(ex)[:60], flush=True)
            continue
        got = spans.cpu().numpy()
        if res.total != cnt or not np.array_equal(got, exp):
            bad += 1
            m = min(len(got), len(exp))
            d = np.nonzero((got[:m] != exp[:m]).any(axis=1))[0]
            k = int(d[0]) if len(d) else m
            print("seed", seed, "len", len(b), "MISMATCH gpu", int(res.total), "oracle", cnt, "first bad row", k,
                  "gpu", got[k].tolist() if k < len(got) else None, "oracle", exp[k].tolist() if k < len(exp) else None, "unsynced", res.unsynced, flush=True)
            open(os.path.join(ROOT, "gpurun_out", "us_fail_%d_%d.bin" % (seed, len(b))), "wb").write(b)
            # again: the same call twice more (a race would not repeat)
            for _ in range(2):
                spans2, res2 = c.FindAllSpans(b)
                print("   again:", int(res2.total), flush=True)
    if bad >= 3:
        break
print("done; bad", bad)
