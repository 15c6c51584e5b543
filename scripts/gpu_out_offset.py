"""Does the placement of the span table relative to the input matter (HBM channel phase)?  Kernel time per output offset."""
import sys
sys.path.insert(0, ".")
import torch
from regengo_amd import Compiled, synth
N = 1 << 30
c = Compiled(r"(?P<year>\d{4})-(?P<month>\d{2})-(?P<day>\d{2})").to(0)
big = synth.date_log_torch(N, "cuda:0")
c.set_timing(True)
cap = N // 10 + 1
raw = torch.empty(cap * 8 + (1 << 22), dtype=torch.int32, device="cuda:0")
print("in ptr %x  raw out ptr %x" % (big.data_ptr(), raw.data_ptr()))
for off_bytes in (0, 256, 1024, 4096, 4096 + 256, 65536, 65536 + 1024, 1 << 20, (1 << 20) + 4096 + 512, 1 << 21):
    out = raw[off_bytes // 4: off_bytes // 4 + cap * 8].view(cap, 8)
    ks = []
    for it in range(12):
        sp, res = c.FindAllSpans(big, out=out, capacity=cap)
        ks.append(res.kernel_ms)
    ks.sort()
    print("offset %8d: min %.4f med %.4f mean %.4f" % (off_bytes, ks[0], ks[len(ks) // 2], sum(ks) / len(ks)), flush=True)
