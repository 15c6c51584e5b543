"""FindAllBytes of the Tagged-DFA programs of the C5 suite as the emitted wrapper answers it (compiler.go:602-655, quirk Q11) over the
web-log corpus: rows against the C port of the emitted code (oracle/tdfa_c.py: t_find_all) on the first CHECK MiB, time of the whole
pipeline (rgx_find_all_bytes_device, events) on that piece and on 1 GiB.  usage: gpu_tdfa_findall.py [check_mib] [big_mib]"""
import json
import sys
import time
sys.path.insert(0, ".")
import numpy as np
import torch
from regengo_amd import Compiled, synth
from oracle.tdfa_c import CTdfa
check_mib = int(sys.argv[1]) if len(sys.argv) > 1 else 64
big_mib = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
tile = synth.web_log_tile(); tile = tile[:tile.rfind(b"\n") + 1]
fx = json.load(open("tests/golden/c5_counts.json"))
pats = [e["pattern"] for e in fx["patterns"] if e["mode"] == "scan"]
big = torch.frombuffer(bytearray(tile), dtype=torch.uint8).cuda().repeat((big_mib << 20) // len(tile) + 1)[:big_mib << 20].contiguous()
small_np = np.frombuffer((tile * ((check_mib << 20) // len(tile) + 1))[:check_mib << 20], dtype=np.uint8)
small = torch.from_numpy(small_np.copy()).cuda()
print("%-8s %-10s %-10s %-9s %-9s %-6s %s" % ("rows/MiB", "ms(check)", "ms(big)", "GB/s big", "cpu s", "ok", "pattern"))
for p in pats:
    c = Compiled(p).to(0)
    if c.info.ref_findall_offered != 2:
        continue
    c.set_timing(True)
    o = CTdfa(p)
    t0 = time.time(); exp = o.find_all_np(small_np); cpu_s = time.time() - t0
    rows, res = c.FindAllSpans(small, capacity=len(exp) + 16)
    ok = res.total == len(exp) and bool(np.array_equal(rows.cpu().numpy(), exp))
    rows, res = c.FindAllSpans(small, capacity=len(exp) + 16)
    ms_small = res.kernel_ms
    del rows
    nbig, rb = c.CountAll(big)
    out = torch.empty((nbig + 16, c.ncap), dtype=torch.int32, device="cuda")
    _, rb = c.FindAllSpans(big, out=out, capacity=nbig + 16)
    _, rb = c.FindAllSpans(big, out=out, capacity=nbig + 16)
    print("%-8.0f %-10.3f %-10.3f %-9.1f %-9.2f %-6s %s" % (len(exp) / check_mib, ms_small, rb.kernel_ms, big.numel() / rb.kernel_ms / 1e6, cpu_s, ok, p[:70]), flush=True)
    del out
