"""First GPU contact: small parity checks + a quick 1 GiB timing.  Ad-hoc driver for gpurun."""
import sys, time, json
sys.path.insert(0, ".")
import numpy as np, torch
from regengo_amd import Compiled, synth
from oracle.engines import Compiled as OCompiled

dev = "cuda:0"
print("device", torch.cuda.get_device_name(0))
DATE = r"(?P<year>\d{4})-(?P<month>\d{2})-(?P<day>\d{2})"
c = Compiled(DATE).to(0)
print("info states", c.info.n_states, "classes", c.info.n_classes, "fixed", c.info.fixed_captures, "table_bytes", c.info.table_bytes)
o = OCompiled(DATE)
for inp in [b"x 2024-01-15 y 12024-01-15 1234-56-7890-12-34", b"", b"2024-01-15", b"a" * 100 + b"2024-01-15" * 30]:
    sp, res = c.FindAllSpans(inp)
    exp = o.FindAllBytes(inp)
    got = sp.cpu().tolist()
    print("small", len(inp), got == exp, res.total, res.unsynced)
    if got != exp: print(got, exp)
for n in [1000, 16384, 16385, 100000, 1 << 20]:
    for adv in (False, True):
        buf = synth.date_log_np(n, adversarial=adv)
        t = torch.from_numpy(buf).to(dev)
        sp, res = c.FindAllSpans(t)
        got = sp.cpu().numpy()
        if not adv:
            exp = synth.date_log_expected(n)
        else:
            exp = np.array(o.FindAllBytes(buf.tobytes()), dtype=np.int32).reshape(-1, 8) if n <= 100000 else None
        ok = exp is None or (got.shape == exp.shape and (got == exp).all())
        print("n", n, "adv", adv, "count", res.total, "unsynced", res.unsynced, "ok", ok)
# torch generator == numpy generator
a = synth.date_log_np(100000, adversarial=True); b = synth.date_log_torch(100000, dev, adversarial=True).cpu().numpy()
print("synth torch==np", (a == b).all())
# timing
N = 1 << 30
big = synth.date_log_torch(N, dev)
torch.cuda.synchronize()
c.set_timing(True)
cap = N // 10 + 1
out = torch.empty((cap, 8), dtype=torch.int32, device=dev)
for it in range(5):
    t0 = time.time()
    sp, res = c.FindAllSpans(big, out=out, capacity=cap)
    torch.cuda.synchronize()
    dt = time.time() - t0
    print("1GiB iter", it, "count", res.total, "kernel_ms %.3f" % res.kernel_ms, "wall_ms %.3f" % (dt * 1e3), "GB/s kernel %.1f" % (N / res.kernel_ms / 1e6))
exp = synth.date_log_expected(N)
got = sp.cpu().numpy()
print("1GiB parity", got.shape == exp.shape and (got == exp).all())
cnt, res = c.CountAll(big)
print("count-only", cnt, "kernel_ms %.3f" % res.kernel_ms)
