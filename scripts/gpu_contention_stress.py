"""Several PROCESSES scanning on one GPU at once (what the multi-process tests over the CCL double do): every scan of the same window must
give the same rows, whatever the other processes are doing.  usage: gpu_contention_stress.py [nproc] [iters]     (parent starts the children)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(idx, iters):
    import torch
    from regengo_amd import Compiled
    from tests._sharded_rank_worker import stream_bytes
    data = stream_bytes()
    buf = torch.frombuffer(bytearray(data), dtype=torch.uint8).to("cuda:0")
    bad = 0
    for name, pattern in (("date", r"(?P<year>\d{4})-(?P<month>\d{2})-(?P<day>\d{2})"),
                          ("url", r"(?P<full>(?P<proto>https?|ftp)://(?P<host>[\w.-]+)(?P<port>:\d+)?(?P<path>/[\w./-]*)?)")):
        c = Compiled(pattern).to(0)
        n = len(data)
        wins = []
        for k in range(20):
            lo, hi = k * n // 20, (k + 1) * n // 20
            wl = max(0, lo - 4096)
            wl -= wl % 16
            wh = min(n, hi + (1 << 20))
            wins.append((buf[wl:wh].clone(), (lo - wl, hi - wl)))
        ref = None
        for it in range(iters):
            got = []
            for w, own in wins:
                if it % 2:
                    c.FindAllSubmit(w, own=own)
                    spans, res = c.FindAllWait()
                else:
                    spans, res = c.FindAllSpans(w, own=own)
                got.append((int(spans.shape[0]), int(spans.to(torch.int64).sum().item())))
            if ref is None:
                ref = got
            elif got != ref:
                bad += 1
                d = [(k, a, b) for k, (a, b) in enumerate(zip(got, ref)) if a != b]
                print("MISMATCH proc %d %s iter %d: %s" % (idx, name, it, d[:3]), flush=True)
    print("proc %d done, %d bad iterations" % (idx, bad), flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        sys.exit(child(int(sys.argv[2]), int(sys.argv[3])))
    nproc = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--child", str(i), str(iters)]) for i in range(nproc)]
    rc = 0
    for p in ps:
        rc |= p.wait()
    print("STRESS", "FAILED" if rc else "OK")
    sys.exit(rc)
