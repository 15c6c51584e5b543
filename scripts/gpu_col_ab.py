"""A/B of the column-form scan kernel against the pair-table kernel (experiment build: RGX_EXTRA_FLAGS=-DRGX_EXPERIMENT), 1 GiB web log.
usage: gpu_col_ab.py            (spawns itself per variant)"""
import os
import subprocess
import sys
sys.path.insert(0, ".")
PATS = [r"(?P<full>(?P<proto>https?|ftp)://(?P<host>[\w.-]+)(?P<port>:\d+)?(?P<path>/[\w./-]*)?)", r"(?P<user>\w+)@(?P<domain>\w+)", r"(\d+)", r"\b[a-z]+\b"]
if len(sys.argv) > 1:
    import torch
    from regengo_amd import Compiled, synth
    tile = synth.web_log_tile(); tile = tile[:tile.rfind(b"\n") + 1]
    N = 1 << 30
    big = torch.frombuffer(bytearray(tile), dtype=torch.uint8).cuda().repeat(N // len(tile)).contiguous()
    for pat in PATS:
        c = Compiled(pat).to(0); c.set_timing(True)
        best_c = best_s = 1e9
        for _ in range(4):
            n, r = c.CountAll(big); best_c = min(best_c, r.kernel_ms)
        out = None
        for _ in range(4):
            sp, r = c.FindAllSpans(big, out=out); best_s = min(best_s, r.kernel_ms); out = sp if out is None else out
        print("  kernel %d count %.3f ms spans %.3f ms  n=%d  %s" % (c.info.scan_kernel, best_c, best_s, n, pat[:40]), flush=True)
    sys.exit(0)
for env in ({}, {"RGX_NO_US_COL": "1"}, {"RGX_US_PER_CU": "4"}, {"RGX_US_PER_CU": "5"}, {"RGX_NO_US_COL": "1", "RGX_US_PER_CU": "4"}):
    print(env, flush=True)
    e = dict(os.environ); e.update(env)
    subprocess.run([sys.executable, sys.argv[0], "child"], env=e)
