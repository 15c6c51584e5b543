"""C3 batch through the reference's Tagged DFA (ForceTDFA): FindBatchDevice, time by events.  usage: gpu_tdfa_batch.py [nstr] [reps]"""
import sys
sys.path.insert(0, ".")
import torch
from regengo_amd import Compiled, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
data, offsets = synth.email_batch_np(n, seed=0x5EED0003)
concat = torch.from_numpy(data).cuda(); offs = torch.from_numpy(offsets).cuda()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
c = Compiled(r"(?P<user>\w+)@(?P<domain>\w+)", force_tdfa=True).to(0)
out = (torch.empty(n, dtype=torch.uint8, device="cuda"), torch.empty((n, c.ncap), dtype=torch.int32, device="cuda"))
ts = []
for _ in range(reps):
    ev[0].record(); f, sp = c.FindBatchDevice(concat, offs, out=out); ev[1].record(); ev[1].synchronize()
    ts.append(ev[0].elapsed_time(ev[1]))
print("tdfa batch %d strings: %.3f ms  found=%d  checksum=%d" % (n, min(ts), int(f.sum().item()), int(sp[f != 0].to(torch.int64).sum().item())))
