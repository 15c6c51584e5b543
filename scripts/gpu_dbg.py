"""Phase timing of the exact kernel under RGX_DEBUG switches (count-only / full), 1 GiB date log."""
import sys, os
sys.path.insert(0, ".")
import torch
from regengo_amd import Compiled, synth
N = 1 << 30
c = Compiled(r"(?P<year>\d{4})-(?P<month>\d{2})-(?P<day>\d{2})").to(0)
big = synth.date_log_torch(N, "cuda:0")
c.set_timing(True)
cap = N // 10 + 1
out = torch.empty((cap, 8), dtype=torch.int32, device="cuda:0")
ks = [c.FindAllSpans(big, out=out, capacity=cap)[1].kernel_ms for _ in range(8)]
ks2 = [c.CountAll(big)[1].kernel_ms for _ in range(8)]
ks3 = [c.FindAllStarts(big)[1].kernel_ms for _ in range(8)]
print("RGX_DEBUG=%s full %.3f count %.3f starts %.3f" % (os.environ.get("RGX_DEBUG", "0"), min(ks), min(ks2), min(ks3)))
