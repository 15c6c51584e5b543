"""CPU-only analysis for profiles/HISTORY.md 7: the scan-mode patterns of the C5 suite that have NO reset byte (every byte value keeps some
state alive, so the one-step-per-byte kernels cannot take them and they run on the generic kernel with the sync automaton).
For each: how many class PAIRS kill every live state within two steps, how dense such pairs are in the web-log corpus and the
longest stretch without one -- i.e. whether "reset pairs" would give those kernels their sync points.
usage: python scripts/analyze_reset_pairs.py"""
import json
import sys

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import _hosttest as H            # noqa: E402
from regengo_amd import synth    # noqa: E402

tile = synth.web_log_tile()
tile = np.frombuffer(tile[: tile.rfind(b"\n") + 1], dtype=np.uint8)
fx = json.load(open("tests/golden/c5_counts.json"))
rows = []
for i, e in enumerate(fx["patterns"]):
    if e["mode"] != "scan":
        continue
    try:
        t = H.HostProgram(e["pattern"], 1)          # stdlib semantics, as the suite compiles its scan-mode patterns
    except ValueError:
        continue
    if sum(t.reset_bytes()):
        continue
    cls, n, m = t.reset_pairs()
    m = np.frombuffer(m, dtype=np.uint8).reshape(n, n)
    k = np.frombuffer(cls, dtype=np.uint8)[tile]
    hit = m[k[:-1], k[1:]] != 0                       # hit[j]: bytes j, j+1 form a reset pair
    pos = np.nonzero(hit)[0]
    gap = int(np.diff(pos).max()) if pos.size > 1 else len(tile)
    rows.append((i, int(m.sum()), n * n, float(hit.mean()), gap, e["pattern"][:70]))
print("%4s %6s %6s %9s %8s  %s" % ("#", "pairs", "of", "density", "max gap", "pattern"))
for r in rows:
    print("%4d %6d %6d %9.4f %8d  %s" % r)
print("patterns without a reset byte:", len(rows), " with reset pairs every <= 1 KiB of this corpus:", sum(1 for r in rows if r[4] <= 1024))
