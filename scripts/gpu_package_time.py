"""Time of the package pass (rgx_multi_*) over the C5 suite's per-line patterns on the 1 GiB corpus: python scripts/gpu_package_time.py"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from regengo_amd import Compiled, Package, synth
fx = json.load(open(os.path.join(ROOT, "tests", "golden", "c5_counts.json")))
tile = synth.web_log_tile(); tile = tile[:tile.rfind(b"\n") + 1]
nt = (1 << 30) // len(tile)
big = torch.frombuffer(bytearray(tile), dtype=torch.uint8).to("cuda:0").repeat(nt).contiguous()
nl = big == 10
ends = torch.nonzero(nl).flatten()
starts = torch.cat([torch.zeros(1, dtype=torch.int64, device="cuda:0"), ends[:-1] + 1])
lines = big[~nl].contiguous()
offs = torch.cat([torch.zeros(1, dtype=torch.int64, device="cuda:0"), torch.cumsum(ends - starts, 0)]).contiguous()
progs = [Compiled(e["pattern"], stdlib=e.get("semantics") != "reference").to(0) for e in fx["patterns"] if e["mode"] == "line"]
pk = Package(progs)
for _ in range(2):
    pk.FindBatchBits(lines, offs)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    bits, cnt, _ = pk.FindBatchBits(lines, offs)
torch.cuda.synchronize()
print("budget=%s window=%s programs=%d launches=%d ms_per_pass=%.3f found=%d" % (os.environ.get("RGX_MULTI_BUDGET"), os.environ.get("RGX_MULTI_WINDOW"),
      sum(pk.accepted), pk.launches, (time.perf_counter() - t0) / 5 * 1e3, int(cnt.sum())))
