"""The filter + candidate kernel (rgx_scan_fc.hip) against the oracle's C port and against the program's other kernel, then timed over
1.6 GiB of the web-log corpus.  usage: gpu_fc_check.py [pattern ...]"""
import os
import subprocess
import sys
import time

sys.path.insert(0, ".")
import numpy as np
import torch

from regengo_amd import Compiled, synth

URL = r"(?P<full>(?P<proto>https?|ftp)://(?P<host>[\w.-]+)(?P<port>:\d+)?(?P<path>/[\w./-]*)?)"
PATS = sys.argv[1:] or [URL, r"https?://[^\s]+", r"(GET|POST) (/\S*)", r"(?P<k>id|took)=(?P<v>\w*)", r"\[(INFO|WARN|ERROR)\]", r"admin@(\w+)\.(\w+)"]


def oracle_rows(pat, data):
    from oracle.gen_c import CMatcher
    rows, cnt = CMatcher(pat).find_all_np(np.frombuffer(data, dtype=np.uint8))
    return rows


def main():
    tile = synth.web_log_tile()
    tile = tile[:tile.rfind(b"\n") + 1]
    small = [tile[:70000], tile[777:200000], tile * 2 + tile[:12345], b"x" * 100 + b" http://a.b/c " + b"y" * 300 + b"http://q", tile[:100]]
    reps = int(1.6 * (1 << 30)) // len(tile)
    big = torch.frombuffer(bytearray(tile), dtype=torch.uint8).cuda().repeat(reps).contiguous()
    for pat in PATS:
        c = Compiled(pat).to(0)
        c.set_timing(True)
        print("PATTERN", pat, "kernel", c.info.scan_kernel, "ncap", c.ncap, flush=True)
        ok = True
        for d in small:
            got = c.FindAllSpans(d)[0].cpu().numpy()
            exp = oracle_rows(pat, d)
            if got.shape != exp.shape or not np.array_equal(got, exp):
                ok = False
                k = 0
                while k < min(len(got), len(exp)) and np.array_equal(got[k], exp[k]):
                    k += 1
                print("  MISMATCH len", len(d), "rows", got.shape, exp.shape, "first diff row", k, got[k:k + 2].tolist() if k < len(got) else None,
                      exp[k:k + 2].tolist() if k < len(exp) else None, flush=True)
        print("  small inputs vs oracle:", "ok" if ok else "FAILED", flush=True)
        cap = 9100 * reps
        out = torch.empty((cap, c.ncap), dtype=torch.int32, device="cuda")
        for _ in range(2):
            sp, r = c.FindAllSpans(big, out=out, capacity=cap)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        kms = []
        for _ in range(5):
            sp, r = c.FindAllSpans(big, out=out, capacity=cap)
            kms.append(r.kernel_ms)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        h = (int(sp.to(torch.int64).sum().item()), int(r.total))
        print("  big: %d matches, scan kernel %.3f ms, call %.3f ms, %.1f GB/s, checksum %s" % (r.total, sum(kms) / len(kms), dt * 1e3, big.numel() / dt / 1e9, h), flush=True)
        if os.environ.get("RGX_NO_FC_KERNEL") is None and c.info.scan_kernel == 7:
            # the same through the program's other kernel, in a process of its own
            p = subprocess.run([sys.executable, __file__, pat], env=dict(os.environ, RGX_NO_FC_KERNEL="1"), capture_output=True, text=True)
            for ln in p.stdout.splitlines():
                if ln.startswith("  big") or ln.startswith("PATTERN") or "FAILED" in ln:
                    print("    [other kernel]", ln.strip(), flush=True)
            if p.returncode != 0:
                print("    [other kernel] rc", p.returncode, p.stderr[-500:])


if __name__ == "__main__":
    main()
