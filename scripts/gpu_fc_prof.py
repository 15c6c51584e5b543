"""Phase cycle counts of rgx_scan_fc.hip (build with RGX_EXTRA_FLAGS=-DRGX_FC_PROFILE).  usage: gpu_fc_prof.py [pattern ...]"""
import sys
sys.path.insert(0, ".")
import torch
from regengo_amd import Compiled, synth
URL = r"(?P<full>(?P<proto>https?|ftp)://(?P<host>[\w.-]+)(?P<port>:\d+)?(?P<path>/[\w./-]*)?)"
pats = sys.argv[1:] or [URL, r"(GET|POST) (/\S*)"]
tile = synth.web_log_tile(); tile = tile[:tile.rfind(b"\n") + 1]
reps = int(1.6 * (1 << 30)) // len(tile)
big = torch.frombuffer(bytearray(tile), dtype=torch.uint8).cuda().repeat(reps).contiguous()
for pat in pats:
    c = Compiled(pat).to(0); c.set_timing(True)
    cap = 9100 * reps
    out = torch.empty((cap, c.ncap), dtype=torch.int32, device="cuda")
    for _ in range(3):
        sp, r = c.FindAllSpans(big, out=out, capacity=cap)
        print(pat[:30], r.total, "kernel_ms %.3f" % r.kernel_ms, flush=True)
