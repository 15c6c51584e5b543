"""Per-pattern timing of the C5 suite's scan-mode patterns (+ the C4 URL pattern) over the 1 GiB corpus: which kernel each takes,
count-only kernel time, full FindAllSpans wall time (events around the call: scan + carry + capture passes), slices without a sync
point.  usage: python scripts/gpu_suite_times.py [max_patterns] > gpurun_out/suite_times.txt"""
import json
import sys
sys.path.insert(0, ".")
import torch
from regengo_amd import Compiled, synth, _capi
tile = synth.web_log_tile(); tile = tile[:tile.rfind(b"\n") + 1]
N = (1 << 30) // len(tile) * len(tile)
big = torch.frombuffer(bytearray(tile), dtype=torch.uint8).cuda().repeat(N // len(tile)).contiguous()
fx = json.load(open("tests/golden/c5_counts.json"))
pats = [("C4-URL", r"(?P<full>(?P<proto>https?|ftp)://(?P<host>[\w.-]+)(?P<port>:\d+)?(?P<path>/[\w./-]*)?)", None)]
pats += [(str(i), e["pattern"], e) for i, e in enumerate(fx["patterns"]) if e["mode"] == "scan"]
if len(sys.argv) > 1:
    pats = pats[:int(sys.argv[1])]
ntiles = N // len(tile)
rows = []
first = None
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
for tag, p, e in pats:
    c = Compiled(p, stdlib=True).to(0, ctx_of=first)
    if first is None:
        first = c
    c.set_timing(True)
    n, r = c.CountAll(big)
    n, r = c.CountAll(big)
    cap = int(n) + 16
    if cap * c.ncap * 4 > 48 << 30:
        rows.append((0.0, r.kernel_ms, 0.0, c.info.scan_kernel, int(n), r.unsynced, tag, p[:70]))
        continue
    out = torch.empty((cap, c.ncap), dtype=torch.int32, device="cuda:0")
    for _ in range(3):          # (a program with two kernels times both once before it settles: rgx_capi.cc fc_pref)
        c.FindAllSpans(big, out=out, capacity=cap)
    n, r = c.CountAll(big)
    ev[0].record()
    sp, r2 = c.FindAllSpans(big, out=out, capacity=cap)
    ev[1].record(); ev[1].synchronize()
    rows.append((ev[0].elapsed_time(ev[1]), r.kernel_ms, r2.kernel_ms, c.info.scan_kernel, int(n), r2.unsynced, tag, p[:70]))
    del out
rows.sort(reverse=True)
print("%9s %9s %9s %4s %11s %8s  %s" % ("full_ms", "count_k", "full_k", "kind", "matches", "unsynced", "pattern"))
for r in rows:
    print("%9.3f %9.3f %9.3f %4d %11d %8d  %s %s" % r)
print("sum full_ms %.1f  sum count kernel %.1f  patterns %d" % (sum(r[0] for r in rows), sum(r[1] for r in rows), len(rows)))
