"""The reference's Tagged DFA on the device, timed: FindReader / FindReaderCount of URLCapture (reference mode: rgx_find_chunk runs the
engine's own loop, csrc/rgx_tdfa.hip) over the web-log corpus from host memory, per BufferSize; and FindBytes per string of a batch.
usage: gpu_tdfa_reader.py [MiB]      (run under rocprofv3 --kernel-trace --stats for the per-kernel split)"""
import io
import sys
import time
sys.path.insert(0, ".")
import torch
from regengo_amd import Compiled, synth
from regengo_amd.stream import Config

URLC = r"(?P<protocol>https?)://(?P<host>[\w\.-]+)(?::(?P<port>\d+))?(?P<path>/[\w\./]*)?"
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 256
tile = synth.web_log_tile(); tile = tile[:tile.rfind(b"\n") + 1]
data = tile * (mib * (1 << 20) // len(tile))
c = Compiled(URLC, name="URLCapture").to(0)
assert c.info.ref_find_engine == 1 and c.info.ref_stream_offered
for bufsize in (1 << 20, 16 << 20, 64 << 20):
    c.FindReaderCount(io.BytesIO(data[:bufsize * 2]), Config(bufsize, 0))          # warm: scratch allocation
    t0 = time.perf_counter()
    n = c.FindReaderCount(io.BytesIO(data), Config(bufsize, 0))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("FindReaderCount BufferSize %3d MiB: %d matches, %.1f ms, %.2f GB/s (host bytes in: PCIe + kernels)" % (bufsize >> 20, n, dt * 1e3, len(data) / dt / 1e9), flush=True)
got = [0]
t0 = time.perf_counter()
c.FindReader(io.BytesIO(data[:64 << 20]), Config(16 << 20, 0), lambda m: got.__setitem__(0, got[0] + 1) or True)
dt = time.perf_counter() - t0
print("FindReader (callbacks in Python) 64 MiB: %d matches, %.1f ms" % (got[0], dt * 1e3))
