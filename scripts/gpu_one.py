"""One pattern over the 1 GiB web-log corpus, a few launches (for rocprofv3 --pmc).  usage: gpu_one.py <pattern> [count]"""
import sys
sys.path.insert(0, ".")
import torch
from regengo_amd import Compiled, synth
pat = sys.argv[1]
tile = synth.web_log_tile(); tile = tile[:tile.rfind(b"\n") + 1]
N = 1 << 30
big = torch.frombuffer(bytearray(tile), dtype=torch.uint8).cuda().repeat(N // len(tile)).contiguous()
c = Compiled(pat).to(0); c.set_timing(True)
for _ in range(3):
    n, r = c.CountAll(big)
print(pat, n, r.kernel_ms)
