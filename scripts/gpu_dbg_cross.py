import sys
sys.path.insert(0, ".")
import numpy as np
from oracle.gen_c import CMatcher
from regengo_amd import Compiled
p = sys.argv[1] if len(sys.argv) > 1 else r"[^a]a{1,2}[^a]+"
cm = CMatcher(p, q8=False)
c = Compiled(p, stdlib=True).to(0)
print("reset bytes:", sum(c.reset_bytes()), "kernel", c.info.scan_kernel, "sync_states", c.info.sync_states)
for name, unit in (("ascii", b"b"), ("utf8", "ü".encode()), ("mixed", "bü".encode())):
    for runlen in (100, 300, 474, 1000, 3000):
        for lead in (16384 - 40, 16384 - 200, 16384 - 450, 16384 - 2900):
            if lead < 10:
                continue
            pre = (b"za" + unit * 8 + b"a ") * 4000
            pre = pre[:lead - 2]
            body = b"za" + (unit * runlen)[:runlen] + b"a-aab" + unit * 10 + b"a" + (b" xa" + unit * 5 + b"a") * 50
            t = pre + body
            exp, cnt = cm.find_all_np(np.frombuffer(t, dtype=np.uint8).copy())
            sp, res = c.FindAllSpans(t)
            got = sp.cpu().numpy()
            ok = res.total == cnt and np.array_equal(got, exp)
            if not ok:
                m = min(len(got), len(exp)); d = np.nonzero((got[:m] != exp[:m]).any(axis=1))[0]
                k = int(d[0]) if len(d) else -1
                print(name, "run", runlen, "lead", lead, "BAD", res.total, cnt, "row", k, got[k].tolist() if k >= 0 else None, exp[k].tolist() if k >= 0 else None, "unsynced", res.unsynced)
print("done")
