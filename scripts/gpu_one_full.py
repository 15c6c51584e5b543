"""One pattern over the 1 GiB web-log corpus: count-only and full FindAllSpans, wall time (events) against the reported scan-kernel
time -- a gap means the scan was launched more than once (look-back fallback, carry pass) or that the capture pass dominates.
usage: gpu_one_full.py <pattern> [iters]"""
import sys
sys.path.insert(0, ".")
import torch
from regengo_amd import Compiled, synth
pat = sys.argv[1]
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
tile = synth.web_log_tile(); tile = tile[:tile.rfind(b"\n") + 1]
N = (1 << 30) // len(tile) * len(tile)
big = torch.frombuffer(bytearray(tile), dtype=torch.uint8).cuda().repeat(N // len(tile)).contiguous()
c = Compiled(pat, stdlib=True).to(0); c.set_timing(True)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
for i in range(iters):
    ev[0].record(); n, r = c.CountAll(big); ev[1].record(); ev[1].synchronize()
    print("count: wall %.3f ms kernel %.3f ms  n=%d unsynced=%d kind=%d" % (ev[0].elapsed_time(ev[1]), r.kernel_ms, n, r.unsynced, c.info.scan_kernel))
cap = int(n) + 16
out = torch.empty((cap, c.ncap), dtype=torch.int32, device="cuda:0")
for i in range(iters):
    ev[0].record(); sp, r = c.FindAllSpans(big, out=out, capacity=cap); ev[1].record(); ev[1].synchronize()
    print("full : wall %.3f ms kernel %.3f ms  n=%d unsynced=%d" % (ev[0].elapsed_time(ev[1]), r.kernel_ms, sp.shape[0], r.unsynced))
