"""Quick timing of the 1 GiB date-log scan (parity-checked).  usage: python scripts/gpu_time.py [iters]"""
import sys, time
sys.path.insert(0, ".")
import torch
from regengo_amd import Compiled, synth
N = 1 << 30
c = Compiled(r"(?P<year>\d{4})-(?P<month>\d{2})-(?P<day>\d{2})").to(0)
big = synth.date_log_torch(N, "cuda:0")
c.set_timing(True)
cap = N // 10 + 1
out = torch.empty((cap, 8), dtype=torch.int32, device="cuda:0")
ks = []
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    sp, res = c.FindAllSpans(big, out=out, capacity=cap)
    ks.append(res.kernel_ms)
exp = torch.from_numpy(synth.date_log_expected(N)).cuda()
ok = sp.shape == exp.shape and bool(torch.equal(sp, exp))
cnt, r2 = c.CountAll(big)
ks2 = [c.CountAll(big)[1].kernel_ms for _ in range(4)]
print("full kernel_ms min %.3f med %.3f | count-only min %.3f | parity %s count %d" % (min(ks), sorted(ks)[len(ks)//2], min(ks2), ok, cnt))
adv = synth.date_log_torch(1 << 26, "cuda:0", adversarial=True)
from oracle.gen_c import CMatcher
e2, n2 = CMatcher(r"(?P<year>\d{4})-(?P<month>\d{2})-(?P<day>\d{2})").find_all_np(adv.cpu().numpy())
s2, r3 = c.FindAllSpans(adv)
print("adversarial 64MiB parity", bool((s2.cpu().numpy() == e2).all()) and r3.total == n2, r3.total, "unsynced", r3.unsynced)
st, r4 = c.FindAllStarts(big)
ks3 = [c.FindAllStarts(big)[1].kernel_ms for _ in range(5)]
print("starts-only kernel_ms min %.3f | parity %s" % (min(ks3), bool(torch.equal(st, exp[:, 0].contiguous()))), c.capture_template())
