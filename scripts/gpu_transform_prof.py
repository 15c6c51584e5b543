"""Device part of one Transform buffer, repeated (for rocprofv3 --kernel-trace --stats).  usage: [MiB] [mode]"""
import ctypes as C, sys
sys.path.insert(0, ".")
import torch
from regengo_amd import Compiled, _capi, synth
MIB = int(sys.argv[1]) if len(sys.argv) > 1 else 256
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 0
N = MIB << 20
PAT = sys.argv[3] if len(sys.argv) > 3 else r"(?P<year>\d{4})-(?P<month>\d{2})-(?P<day>\d{2})"
c = Compiled(PAT).to(0)
dlog = synth.date_log_torch(N, "cuda:0")
d_out = torch.empty(N + N // 4, dtype=torch.uint8, device="cuda:0")
need, done, res = C.c_int64(), C.c_int64(), _capi.Result()
tb = b"$day/$month/$year"
lib = _capi.lib()
for it in range(6):
    w = lib.rgx_transform_chunk_device(c._h, c._ctx, dlog.data_ptr(), N, 0, mode, tb, len(tb), d_out.data_ptr(), d_out.numel(),
                                       C.byref(need), C.byref(done), C.byref(res))
    assert w >= 0, w
print("ok", w, done.value, res.total)
