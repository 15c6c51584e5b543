"""Kernel time of the generic scan kernels on a 1 GiB web-log corpus: full spans vs count-only, per pattern."""
import sys
sys.path.insert(0, ".")
import torch
from regengo_amd import Compiled, synth
PATS = {
 "url": r"(?P<full>(?P<proto>https?|ftp)://(?P<host>[\w.-]+)(?P<port>:\d+)?(?P<path>/[\w./-]*)?)",
 "email": r"(?P<user>\w+)@(?P<domain>\w+)",
 "digits": r"(\d+)",
 "time": r"(\d{2}):(\d{2}):(\d{2})",
 "level": r"\[(INFO|WARN)\]",
 "word_b": r"\b[a-z]+\b",
}
tile = synth.web_log_tile(); tile = tile[:tile.rfind(b"\n") + 1]
N = 1 << 30
big = torch.frombuffer(bytearray(tile), dtype=torch.uint8).cuda().repeat(N // len(tile)).contiguous()
for name, pat in PATS.items():
    c = Compiled(pat).to(0); c.set_timing(True)
    sp, r = c.FindAllSpans(big)
    ks = [c.FindAllSpans(big)[1].kernel_ms for _ in range(3)]
    cs = [c.CountAll(big)[1].kernel_ms for _ in range(3)]
    print("%-7s matches %9d  scan kernel full %.3f ms  count-only %.3f ms  (%.0f GB/s)  states %d fixed %d" % (
        name, r.total, min(ks), min(cs), big.numel() / min(ks) / 1e6, c.info.n_states, c.info.fixed_captures), flush=True)
