import time, torch, sys
sys.path.insert(0, "/root/repo")
from regengo_amd import Compiled, synth
URL = r"(?P<full>(?P<proto>https?|ftp)://(?P<host>[\w.-]+)(?P<port>:\d+)?(?P<path>/[\w./-]*)?)"
c = Compiled(URL).to(0)
tile = synth.web_log_tile()
tile = tile[:tile.rfind(b"\n") + 1]
n = (1 << 30) // len(tile)
buf = torch.frombuffer(bytearray(tile), dtype=torch.uint8).to("cuda:0").repeat(n)
def t(f, k=8):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e3
cnt = c.CountAll(buf)
cnt = int(cnt[0] if isinstance(cnt, tuple) else cnt)
out = torch.empty((cnt + 1024, c.ncap), dtype=torch.int32, device="cuda:0")
print("bytes", buf.numel(), "matches", cnt)
print("count-only ms", round(t(lambda: c.CountAll(buf)), 3))
print("full ms      ", round(t(lambda: c.FindAllSpans(buf, out=out)), 3))
