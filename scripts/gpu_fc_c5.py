"""The C5 suite's scan-mode patterns that take rgx_scan_fc.hip, over the 1 GiB corpus: count and row checksums against the fixture and
against the program's other kernel; kernel times of both.  usage: gpu_fc_c5.py [GiB]"""
import json, os, sys, time
sys.path.insert(0, ".")
import torch
from regengo_amd import Compiled, synth, rowsum, _capi
fx = json.load(open("tests/golden/c5_counts.json"))
tile = synth.web_log_tile(); tile = tile[:tile.rfind(b"\n") + 1]
T = len(tile)
nbytes = int(float(sys.argv[1]) * (1 << 30)) if len(sys.argv) > 1 else 1 << 30
ntiles = nbytes // T
big = torch.frombuffer(bytearray(tile), dtype=torch.uint8).cuda().repeat(ntiles).contiguous()
tot_fc = tot_other = 0.0
for i, e in enumerate(fx["patterns"]):
    if e["mode"] != "scan":
        continue
    stdlib = Compiled(e["pattern"]).info.ref_findall_offered != 1
    c = Compiled(e["pattern"], stdlib=stdlib).to(0)
    if c.info.scan_kernel != 7:
        continue
    o = Compiled(e["pattern"], stdlib=stdlib, no_prefilter_scan=True).to(0)
    c.set_timing(True); o.set_timing(True)
    exp = e["a"] + (ntiles - 2) * e["u"] + e["z"]
    cap = exp + 64
    out = torch.empty((cap, c.ncap), dtype=torch.int32, device="cuda")
    res = {}
    for name, p in (("fc", c), ("other", o)):
        try:
            for _ in range(3):                      # (the third call: the program has chosen)
                t0 = time.perf_counter(); sp, r = p.FindAllSpans(big, out=out, capacity=cap); torch.cuda.synchronize(); dt = time.perf_counter() - t0
            res[name] = (int(r.total), rowsum.device(sp), r.kernel_ms, dt * 1e3)
        except _capi.RgxError as ex:
            res[name] = ("ERR %s" % ex, None, 0, 0)
    ok = res["fc"][0] == exp and res["fc"][:2] == res["other"][:2]
    if "rs" in e:
        n_exp, h1, h2 = rowsum.periodic(e["rs"]["a"], e["rs"]["u"], e["rs"]["z"], ntiles, T)
        ok = ok and res["fc"][1] == (h1, h2)
    tot_fc += res["fc"][3]; tot_other += res["other"][3]
    print("%3d %-5s n=%d exp=%d fc %.3f/%.3f ms other %.3f/%.3f ms  %s" % (i, "ok" if ok else "BAD", res["fc"][0] if isinstance(res["fc"][0], int) else -1, exp,
          res["fc"][2], res["fc"][3], res["other"][2], res["other"][3], e["pattern"][:70]), flush=True)
print("sum of call ms: fc-first %.1f, other %.1f" % (tot_fc, tot_other))
