"""C3 batch (10M strings): FindBatchDevice in reference mode against the plain search (stdlib=True), kernel time by events.
usage: python scripts/gpu_batch_modes.py [nstr]"""
import sys
sys.path.insert(0, ".")
import torch
from regengo_amd import Compiled, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
data, offsets = synth.email_batch_np(n)
concat = torch.from_numpy(data).cuda(); offs = torch.from_numpy(offsets).cuda()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
for name, kw in (("reference", {}), ("stdlib", {"stdlib": True})):
    for pat in (r"(?P<user>\w+)@(?P<domain>\w+)", r"(?P<year>\d{4})-(?P<month>\d{2})-(?P<day>\d{2})"):
        c = Compiled(pat, **kw).to(0)
        for what in ("find", "match"):
            ts = []
            for _ in range(4):
                ev[0].record()
                r = c.FindBatchDevice(concat, offs) if what == "find" else c.MatchBatchDevice(concat, offs)
                ev[1].record(); ev[1].synchronize()
                ts.append(ev[0].elapsed_time(ev[1]))
            f = r[0] if what == "find" else r
            print("%-9s %-5s %-40s %.3f ms  found=%d" % (name, what, pat[:40], min(ts), int(f.sum().item())))
