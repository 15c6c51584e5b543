"""Wall time of FindAllSpans (scan + capture back-trace) for the URL pattern over the 1 GiB web log."""
import sys, time
sys.path.insert(0, ".")
import torch
from regengo_amd import Compiled, synth
URL = r"(?P<full>(?P<proto>https?|ftp)://(?P<host>[\w.-]+)(?P<port>:\d+)?(?P<path>/[\w./-]*)?)"
tile = synth.web_log_tile(); tile = tile[:tile.rfind(b"\n") + 1]
N = 1 << 30
big = torch.frombuffer(bytearray(tile), dtype=torch.uint8).cuda().repeat(N // len(tile)).contiguous()
c = Compiled(URL).to(0); c.set_timing(True)
cap = N // c.MinMatchLen + 1
out = torch.empty((cap, c.ncap), dtype=torch.int32, device="cuda:0")
for it in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    own = (4096, N - (1 << 20)) if len(sys.argv) > 1 else None
    sp, r = c.FindAllSpans(big, out=out, capacity=cap, own=own)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("wall %.3f ms  scan kernel %.3f ms  matches %d" % (dt * 1e3, r.kernel_ms, r.total), flush=True)
