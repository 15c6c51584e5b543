"""Timing breakdown for individual patterns over a web-log corpus (diagnostic)."""
import sys, time, json
sys.path.insert(0, ".")
import numpy as np, torch
from regengo_amd import Compiled, synth
tile = synth.web_log_tile(); tile = tile[:tile.rfind(b"\n") + 1]
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 16
big = torch.frombuffer(bytearray(tile), dtype=torch.uint8).cuda().repeat((mib << 20) // len(tile)).contiguous()
corpus = json.load(open("tests/golden/e2e_corpus.json")); kats = json.load(open("tests/golden/kats.json"))
pats = [e["pattern"] for e in corpus] + [c["pattern"] for c in kats["curated_cases"]]
for i in [int(x) for x in sys.argv[2].split(",")]:
    p = pats[i]
    t0 = time.time(); c = Compiled(p).to(0); c.set_timing(True); t1 = time.time()
    torch.cuda.synchronize(); sp, res = c.FindAllSpans(big); torch.cuda.synchronize(); t2 = time.time()
    print(i, "compile %.2fs scan %.3fs kernel_ms %.3f total %d unsynced %d states %d fixed %d min %d max %d" % (
        t1 - t0, t2 - t1, res.kernel_ms, res.total, res.unsynced, c.info.n_states, c.info.fixed_captures, c.MinMatchLen, c.MaxMatchLen), repr(p)[:60], flush=True)
