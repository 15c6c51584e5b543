"""Generic-kernel parity + timing on a 256 MiB web-log corpus for several patterns."""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
from regengo_amd import Compiled, synth
from oracle.gen_c import CMatcher
PATS = {
 "url": r"(?P<protocol>https?)://(?P<host>[\w\.-]+)(?::(?P<port>\d+))?(?P<path>/[\w\./]*)?",
 "email": r"(?P<user>\w+)@(?P<domain>\w+)",
 "emailcap": r"(?P<user>[\w\.+-]+)@(?P<domain>[\w\.-]+)\.(?P<tld>[\w\.-]+)",
 "date": r"(?P<year>\d{4})-(?P<month>\d{2})-(?P<day>\d{2})",
 "time": r"(\d{2}):(\d{2}):(\d{2})",
 "digits": r"(\d+)",
 "word_b": r"\b[a-z]+\b",
 "level": r"\[(INFO|WARN)\]",
}
tile = synth.web_log_tile()
N = 1 << 28
big = synth.tile_repeat_torch(tile, N, "cuda:0")
small = np.frombuffer(tile * 4, dtype=np.uint8)
sm = torch.from_numpy(small.copy()).cuda()
for name, pat in PATS.items():
    c = Compiled(pat).to(0)
    c.set_timing(True)
    sp, res = c.FindAllSpans(sm)
    exp, cnt = CMatcher(pat).find_all_np(small)
    ok = res.total == cnt and np.array_equal(sp.cpu().numpy(), exp)
    cap = N // max(c.MinMatchLen, 1) + 1
    try:
        out = torch.empty((min(cap, 1 << 26), c.ncap), dtype=torch.int32, device="cuda:0")
        ks = []
        for _ in range(3):
            sp2, r2 = c.FindAllSpans(big, out=out, capacity=out.shape[0])
            ks.append(r2.kernel_ms)
        t0 = time.time(); sp2, r2 = c.FindAllSpans(big, out=out, capacity=out.shape[0]); torch.cuda.synchronize(); wall = time.time() - t0
        print("%-9s parity %s  states %3d K? fixed %d | 256MiB: matches %9d scan %.3f ms (%.0f GB/s) total wall %.3f ms unsynced %d" % (
            name, ok, c.info.n_states, c.info.fixed_captures, r2.total, min(ks), N / min(ks) / 1e6, wall * 1e3, r2.unsynced))
    except Exception as e:
        print(name, "parity", ok, "ERR", e)
