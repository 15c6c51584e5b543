"""C3: Email pattern over a 10M-string batch (CSR), FindBytes per string; parity on a sample vs the oracle."""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
from regengo_amd import Compiled, synth
from oracle.gen_c import CMatcher
EMAIL = r"(?P<user>\w+)@(?P<domain>\w+)"
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
t0 = time.time(); data, offsets = synth.email_batch_np(n); print("gen %.1fs bytes %d" % (time.time() - t0, data.size))
c = Compiled(EMAIL).to(0)
d = torch.from_numpy(data).cuda(); o = torch.from_numpy(offsets).cuda()
for it in range(3):
    torch.cuda.synchronize(); t0 = time.time()
    found, spans = c.FindBatchDevice(d, o)
    torch.cuda.synchronize(); dt = time.time() - t0
    print("find batch %.3f ms  %.1f M strings/s  %.1f GB/s" % (dt * 1e3, n / dt / 1e6, data.size / dt / 1e9))
torch.cuda.synchronize(); t0 = time.time(); m = c.MatchBatchDevice(d, o); torch.cuda.synchronize(); dt = time.time() - t0
print("match batch %.3f ms" % (dt * 1e3))
f = found.cpu().numpy(); sp = spans.cpu().numpy(); cm = CMatcher(EMAIL)
bad = 0
for i in range(0, n, max(1, n // 20000)):
    s = np.ascontiguousarray(data[offsets[i]:offsets[i + 1]])
    exp, cnt = cm.find_all_np(s, n=1)
    if bool(f[i]) != (cnt > 0) or (cnt and sp[i].tolist() != exp[0].tolist()): bad += 1
print("sample parity bad =", bad, "found frac %.3f" % f.mean())
