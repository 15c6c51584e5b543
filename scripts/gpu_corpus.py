"""GPU vs oracle on the reference's e2e corpus (inputs + simple mutations). Ad-hoc driver for gpurun."""
import sys, json, time
sys.path.insert(0, ".")
import torch
from regengo_amd import Compiled, _capi
from oracle.engines import Compiled as OCompiled
d = json.load(open("tests/golden/e2e_corpus.json"))
ok = bad = uns = 0
t0 = time.time()
for e in d:
    p = e["pattern"]
    try:
        c = Compiled(p).to(0)
    except _capi.RgxError as ex:
        uns += 1
        continue
    o = OCompiled(p)
    ins = [s.encode() for s in e["inputs"]]
    ins += [b" ".join(ins), b"x" + ins[0] if ins else b"x"]
    for b in ins:
        exp = o.find_machine.find_all(b)
        sp, res = c.FindAllSpans(b)
        got = sp.cpu().tolist()
        if got != exp:
            bad += 1
            if bad < 10: print("BAD", repr(p), b, exp, got, "sa", c.info.n_states)
        else:
            ok += 1
print("corpus ok", ok, "bad", bad, "unsupported", uns, "time %.1fs" % (time.time() - t0))
