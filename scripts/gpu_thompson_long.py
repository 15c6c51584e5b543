"""MatchBytes of ONE long text by the interpreted Thompson matcher (thompson_scan_kernel): ms per call over the web-log corpus repeated to
`mib` MiB, without a match and with one planted at the very end.  Usage: python scripts/gpu_thompson_long.py [mib]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from regengo_amd import Compiled, synth

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 256
tile = synth.web_log_tile()
t = torch.frombuffer(bytearray(tile), dtype=torch.uint8).cuda()
text = t.repeat((mib << 20) // len(tile) + 1)[: mib << 20].contiguous()
for pat, hit in ((r"(\w+\s+)+(?:end\b|fin)!", b"go to fin!"), (r"(a+)+(?:\bx| y)", b"aa y")):
    c = Compiled(pat).to(0)
    for planted in (False, True):
        x = text.clone()
        if planted:
            x[-len(hit):] = torch.frombuffer(bytearray(hit), dtype=torch.uint8).cuda()
        torch.cuda.synchronize()
        r = c.MatchBytes(x)
        t0 = time.perf_counter()
        for _ in range(5):
            r = c.MatchBytes(x)
        ms = (time.perf_counter() - t0) * 200
        print("%-28s %4d MiB planted=%d -> %s  %.2f ms  %.1f GB/s" % (pat, mib, planted, r, ms, (mib << 20) / ms / 1e6), flush=True)
