"""Throughput of the streaming Transform row (ReplaceReader over a synthetic date log), parity-checked by closed form.
usage: python scripts/gpu_transform.py [MiB]   -> JSON lines in gpurun_out/transform.json

Two views:
  * end to end through the io.Reader protocol (host bytes in, host bytes out; PCIe + Python loop inclusive), per
    BufferSize -- the number comparable with the reference's published ~90 MB/s (docs/transform-api.md:157);
  * the device part of one buffer alone (rgx_transform_chunk_device on resident data): FindAllBytes + splice kernels.
"""
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, ".")
import numpy as np
import torch

from regengo_amd import Compiled, _capi, synth
from regengo_amd.stream import Config

MIB = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N = MIB << 20
c = Compiled(r"(?P<year>\d{4})-(?P<month>\d{2})-(?P<day>\d{2})").to(0)
dlog = synth.date_log_torch(N, "cuda:0")
data = dlog.cpu().numpy().tobytes()
rows = []


class Src:
    def __init__(self, d):
        self.d, self.p = memoryview(d), 0

    def read(self, k):
        out = self.d[self.p:self.p + k]
        self.p += len(out)
        return out


want = None
for bs in (64 << 10, 1 << 20, 16 << 20, 64 << 20):
    if bs > N:
        continue
    best = None
    for rep in range(2):
        r = c.ReplaceReader(Src(data), "$day/$month/$year", Config(bs, 0))
        t0 = time.perf_counter()
        out = r.read_all()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    if want is None:
        a = np.frombuffer(out, dtype=np.uint8)
        pos = np.arange(0, N - 9, 50)
        ok = len(out) == N and all((a[pos + k] == b"15/01/2024"[k]).all() for k in range(10)) and r.matches == (N - 10) // 50 + 1
        want = out
    else:
        ok = out == want
    rows.append({"view": "reader_end_to_end", "buffer_size": bs, "input_MiB": MIB, "seconds": round(best, 4),
                 "MB_per_s": round(N / best / 1e6, 1), "chunks": r.chunks, "matches": r.matches, "parity": bool(ok)})
    print(rows[-1], flush=True)

# device part of one buffer
lib = _capi.lib()
for bs in (1 << 20, 16 << 20, 256 << 20):
    if bs > N:
        continue
    d_out = torch.empty(bs + bs // 4, dtype=torch.uint8, device="cuda:0")
    need, done, res = C.c_int64(), C.c_int64(), _capi.Result()
    tb = b"$day/$month/$year"
    ts = []
    for it in range(6):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        w = lib.rgx_transform_chunk_device(c._h, c._ctx, dlog.data_ptr(), bs, 0, 0, tb, len(tb), d_out.data_ptr(), d_out.numel(),
                                           C.byref(need), C.byref(done), C.byref(res))
        ts.append(time.perf_counter() - t0)
    assert w >= 0, w
    t = min(ts)
    rows.append({"view": "device_chunk", "buffer_size": bs, "ms": round(t * 1e3, 3), "GB_per_s_in": round(bs / t / 1e9, 1),
                 "processed": int(done.value), "out_len": int(w), "matches": int(res.total)})
    print(rows[-1], flush=True)

os.makedirs("gpurun_out", exist_ok=True)
json.dump(rows, open("gpurun_out/transform.json", "w"), indent=1)
