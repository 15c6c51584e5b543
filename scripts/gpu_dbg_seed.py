import random, sys
sys.path.insert(0, ".")
import numpy as np
from oracle.gen_c import CMatcher
from regengo_amd import Compiled
from tests import _fuzzgen as F
seed, p = 1023, r"[^a]a{1,2}[^a]+"
# reproduce the sweep's input: same rng consumption order as gpu_fuzz_sweep.py for this pattern
rng = random.Random(seed)
pats = [(q, False) for q in F.gen_patterns(seed, 30)] + [(q, True) for q in F.gen_patterns_u(seed, 12)]
from oracle import engines as E
target = None
for q, uni in pats:
    try:
        o = E.Compiled(q)
    except Exception:
        continue
    if F.has_empty_loop(o.prog) and not o.find_machine.memo:
        continue
    cmq = CMatcher(q, q8=False)
    slow = False
    for n in (0, 3, 64, 1000, 70000):
        if n >= 20000 and (cmq.memo or slow):
            continue
        b = F.gen_input_u(rng, max(n // 2, 1) if n else 0) if uni else F.gen_input(rng, n)
        if q == p and n == 70000:
            target = b
        import time
        t1 = time.time(); cmq.find_all_np(np.frombuffer(b, dtype=np.uint8).copy() if len(b) else np.zeros(0, dtype=np.uint8)); t2 = time.time()
        slow = slow or (t2 - t1 > 0.005 * max(1, len(b) // 1000))
assert target is not None
open("gpurun_out/seed1023.bin", "wb").write(target)
cm = CMatcher(p, q8=False)
c = Compiled(p, stdlib=True).to(0)
for lo, hi in ((0, len(target)), (0, 20000), (0, 16600), (16000, 17000), (16300, 16700), (8000, 24000)):
    b = target[lo:hi]
    exp, cnt = cm.find_all_np(np.frombuffer(b, dtype=np.uint8).copy())
    sp, res = c.FindAllSpans(b)
    got = sp.cpu().numpy()
    ok = res.total == cnt and np.array_equal(got, exp)
    print(lo, hi, "ok" if ok else "BAD", res.total, cnt, "unsynced", res.unsynced)
    if not ok:
        m = min(len(got), len(exp)); d = np.nonzero((got[:m] != exp[:m]).any(axis=1))[0]
        if len(d):
            k = int(d[0]); print("   row", k, got[k].tolist(), exp[k].tolist(), "prev", exp[k - 1].tolist() if k else None)
