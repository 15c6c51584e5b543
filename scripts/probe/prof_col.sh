RGX_EXTRA_FLAGS="-DRGX_EXPERIMENT -DRGX_US_PROFILE" python -c "
from regengo_amd import build; build.build_product()" 2>&1 | tail -2
cat > /tmp/one.py <<'PY'
import sys
sys.path.insert(0, ".")
import torch
from regengo_amd import Compiled, synth
pat = r"(?P<full>(?P<proto>https?|ftp)://(?P<host>[\w.-]+)(?P<port>:\d+)?(?P<path>/[\w./-]*)?)"
tile = synth.web_log_tile(); tile = tile[:tile.rfind(b"\n") + 1]
N = 1 << 30
big = torch.frombuffer(bytearray(tile), dtype=torch.uint8).cuda().repeat(N // len(tile)).contiguous()
c = Compiled(pat).to(0); c.set_timing(True)
n, r = c.CountAll(big)
torch.cuda.synchronize()
print(pat[:30], n, r.kernel_ms, c.info.scan_kernel)
PY
export RGX_EXTRA_FLAGS="-DRGX_EXPERIMENT -DRGX_US_PROFILE"
RGX_US_PER_CU=4 python /tmp/one.py 2>&1 | grep -v amdgpu | tail -12
RGX_US_PER_CU=4 RGX_NO_US_COL=1 python /tmp/one.py 2>&1 | grep -v amdgpu | tail -12
