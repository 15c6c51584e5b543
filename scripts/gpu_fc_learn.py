"""How a program's scans go while it learns which of its two kernels is faster (rgx_capi.cc: fc_pref): wall and kernel time of every
call.  usage: RGX_FC_VERBOSE=1 gpu_fc_learn.py <c5 pattern index> ..."""
import json
import sys
sys.path.insert(0, ".")
import torch
from regengo_amd import Compiled, synth
tile = synth.web_log_tile(); tile = tile[:tile.rfind(b"\n") + 1]
N = (1 << 30) // len(tile) * len(tile)
big = torch.frombuffer(bytearray(tile), dtype=torch.uint8).cuda().repeat(N // len(tile)).contiguous()
fx = json.load(open("tests/golden/c5_counts.json"))
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
first = None
for idx in map(int, sys.argv[1:]):
    p = fx["patterns"][idx]["pattern"]
    c = Compiled(p, stdlib=True).to(0, ctx_of=first)
    first = first or c
    c.set_timing(True)
    print("PATTERN", idx, p[:80], "kind", c.info.scan_kernel, flush=True)
    for i in range(3):
        ev[0].record(); n, r = c.CountAll(big); ev[1].record(); ev[1].synchronize()
        print("  count %d: wall %.3f ms kernel %.3f ms n=%d" % (i, ev[0].elapsed_time(ev[1]), r.kernel_ms, n), flush=True)
    cap = int(n) + 16
    out = torch.empty((cap, c.ncap), dtype=torch.int32, device="cuda:0")
    for i in range(4):
        ev[0].record(); sp, r = c.FindAllSpans(big, out=out, capacity=cap); ev[1].record(); ev[1].synchronize()
        print("  full  %d: wall %.3f ms kernel %.3f ms rows=%d" % (i, ev[0].elapsed_time(ev[1]), r.kernel_ms, sp.shape[0]), flush=True)
