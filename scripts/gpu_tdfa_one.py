"""One Tagged-DFA program, FindAllBytes (the emitted wrapper) and FindReader's chain over 1 GiB of the web log, a few times: run under
rocprofv3 --kernel-trace --stats for the per-kernel split.  usage: gpu_tdfa_one.py [pattern index 0|1] [mib]"""
import sys
sys.path.insert(0, ".")
import torch
from regengo_amd import Compiled, synth
from regengo_amd.stream import Config
PATS = [r"(?P<protocol>https?)://(?P<host>[\w\.-]+)(?::(?P<port>\d+))?(?P<path>/[\w\./]*)?",
        r"(?P<major>\d+)\.(?P<minor>\d+)\.(?P<patch>\d+)(?:-(?P<prerelease>[\w.-]+))?(?:\+(?P<build>[\w.-]+))?"]
p = PATS[int(sys.argv[1]) if len(sys.argv) > 1 else 0]
mib = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
tile = synth.web_log_tile(); tile = tile[:tile.rfind(b"\n") + 1]
big = torch.frombuffer(bytearray(tile), dtype=torch.uint8).cuda().repeat((mib << 20) // len(tile) + 1)[:mib << 20].contiguous()
c = Compiled(p).to(0)
c.set_timing(True)
n, _ = c.CountAll(big)
out = torch.empty((n + 16, c.ncap), dtype=torch.int32, device="cuda")
for _ in range(3):
    _, r = c.FindAllSpans(big, out=out, capacity=n + 16)
print("FindAll wrapper: %d rows, %.3f ms" % (r.total, r.kernel_ms))
cfg = c._resolve(Config(4 << 20, 0))
import time
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    rows, res = c.FindChunksDevice(big, cfg, final=True, out=out)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("FindReader chunks (4 MiB): %d rows, mode %d, %.3f ms wall" % (res.rows, res.mode, dt * 1e3))
