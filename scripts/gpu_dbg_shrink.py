import sys
sys.path.insert(0, ".")
import numpy as np
from oracle.gen_c import CMatcher
from regengo_amd import Compiled
p = r"[^a]a{1,2}[^a]+"
cm = CMatcher(p, q8=False)
c = Compiled(p, stdlib=True).to(0)
t0 = open("tests/golden/regress/seed1023_head.bin", "rb").read()[:16600]
def run(name, t):
    exp, cnt = cm.find_all_np(np.frombuffer(t, dtype=np.uint8).copy())
    sp, res = c.FindAllSpans(t)
    got = sp.cpu().numpy()
    ok = res.total == cnt and np.array_equal(got, exp)
    print(name, "ok" if ok else "BAD", res.total, cnt, "unsynced", res.unsynced, "last rows", got[-3:, :2].tolist(), exp[-3:, :2].tolist())
    return ok
run("orig", t0)
run("prefix blank to 15900", b" " * 15900 + t0[15900:])
run("prefix blank to 15950", b" " * 15950 + t0[15950:])
body = t0[15956:16430]
asc = bytes(b if b < 128 else ord("b") for b in body)
run("match body ascii", t0[:15956] + asc + t0[16430:])
run("blank prefix + ascii body", b" " * 15956 + asc + t0[16430:])
run("blank prefix + ascii body + simple tail", b" " * 15956 + asc + b"a" + b"b" * 29 + b"-aab" + b"b" * 60 + b"a  ")
for L in (100, 200, 250, 260, 300, 400, 474):
    t = b" " * (16430 - L) + b"z" + b"a" + b"b" * (L - 2) + b"a" + b"b" * 29 + b"-aab" + b"b" * 60 + b"a  " + b" " * 100
    run("synthetic L=%d" % L, t)
