"""C3 batch on the default engine (batch_tiny_kernel): FindBatchDevice, time by events.  usage: gpu_tiny_batch.py [nstr] [reps] [lo hi]
(lo hi: string lengths drawn evenly from [lo, hi] instead of the C3 corpus)"""
import sys
sys.path.insert(0, ".")
import numpy as np
import torch
from regengo_amd import Compiled, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
data, offsets = synth.email_batch_np(n, seed=0x5EED0003)
if len(sys.argv) > 4:
    lo, hi = int(sys.argv[3]), int(sys.argv[4])
    rng = np.random.default_rng(5)
    lens = rng.integers(lo, hi + 1, size=n)
    offsets = np.zeros(n + 1, dtype=np.int64); np.cumsum(lens, out=offsets[1:])
    data = np.resize(data, int(offsets[-1]))
concat = torch.from_numpy(data).cuda(); offs = torch.from_numpy(offsets).cuda()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
c = Compiled(r"(?P<user>\w+)@(?P<domain>\w+)").to(0)
out = (torch.empty(n, dtype=torch.uint8, device="cuda"), torch.empty((n, c.ncap), dtype=torch.int32, device="cuda"))
ts = []
for _ in range(reps):
    ev[0].record(); f, sp = c.FindBatchDevice(concat, offs, out=out); ev[1].record(); ev[1].synchronize()
    ts.append(ev[0].elapsed_time(ev[1]))
print("tiny batch %d strings, %d bytes: %.3f ms  found=%d  checksum=%d" % (n, len(data), min(ts), int((f != 0).sum().item()), int(sp[f != 0].to(torch.int64).sum().item())))
