"""GPU tier: multi-GPU behind the C ABI (include/rgx.h: rgx_sharded_*, csrc/rgx_sharded.hip) on a one-GPU box: a device list of
one; two LOGICAL shards on one device (the communicator-less path: peer copies); a world of one with a real RCCL communicator
(RGX_SHARDED_FORCE_RCCL: ncclCommInitRank, the all-gather of the round and the gather run through librccl); two rounds in
flight; halos without a sync point; refusals.  Everything is compared with ONE scan of the whole buffer (itself checked against
the oracle by tests/test_gpu_parity.py) and with the oracle directly."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DATE = r"(?P<year>\d{4})-(?P<month>\d{2})-(?P<day>\d{2})"
URL = r"(?P<full>(?P<proto>https?|ftp)://(?P<host>[\w.-]+)(?P<port>:\d+)?(?P<path>/[\w./-]*)?)"
EMAIL = r"(?P<user>\w+)@(?P<domain>\w+)"


@pytest.fixture(scope="module")
def gpu(built):
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU; there is no CPU fallback")
    return torch


def _tile():
    from regengo_amd import synth
    t = synth.web_log_tile()
    return t[:t.rfind(b"\n") + 1]


def _windows(torch, buf, plan, out_rows=None):
    L = buf.numel()
    ws = []
    for i, (lo, hi, wl, wh) in enumerate(plan):
        if hi <= lo:
            ws.append(None)
            continue
        ws.append(dict(buf=buf[wl:wh].clone(), own=(lo - wl, hi - wl), base=wl, starts_at_sync=wl == 0, last=wh >= L,
                       out=None if out_rows is None else out_rows[i]))
    return ws


@pytest.mark.parametrize("pattern", [DATE, URL, EMAIL])
@pytest.mark.parametrize("ndev", [1, 2, 3])
def test_logical_shards_equal_one_scan(gpu, pattern, ndev):
    torch = gpu
    from regengo_amd import Compiled
    from regengo_amd.sharded import Sharded
    tile = _tile()
    data = (tile * 3)[: 2 * len(tile) + 54321]
    buf = torch.frombuffer(bytearray(data), dtype=torch.uint8).to("cuda:0")
    c = Compiled(pattern).to(0)
    whole = c.FindAllSpans(buf)[0].to(torch.int64)
    s = Sharded(c, devices=[0] * ndev)
    assert (s.n_local, s.world, s.uses_rccl) == (ndev, ndev, False)            # repeated devices: the communicator-less path
    plan = s.plan(len(data))
    outs = [torch.empty((len(data) // max(c.MinMatchLen, 1) // ndev + 4096, c.ncap), dtype=torch.int32, device="cuda:0") for _ in range(ndev)]
    total, rs = s.round_counts(_windows(torch, buf, plan, outs))
    assert total == whole.shape[0] and not any(r["unsynced"] or r["truncated"] for r in rs)
    # rows per shard: window-relative, in the caller's buffers; owned = the starts inside [lo, hi)
    row = 0
    for i, (lo, hi, wl, wh) in enumerate(plan):
        n, ptr, base = s.rows_ptr(i)
        assert n == rs[i]["count"] and base == wl and (n == 0 or ptr == outs[i].data_ptr())
        mine = whole[(whole[:, 0] >= lo) & (whole[:, 0] < hi)]
        assert n == mine.shape[0]
        got = outs[i][:n].to(torch.int64)
        unset = (got[:, 2:].view(n, -1, 2) == 0).all(dim=2).repeat_interleave(2, dim=1)
        glob = got + wl
        glob[:, 2:] = torch.where(unset, got[:, 2:], glob[:, 2:])
        assert torch.equal(glob, mine)
        row += n
    # the gather: stream-absolute int64 rows in rank order on device 0 -- into a caller buffer, a library buffer, and the host
    dst = torch.empty((total + 8, c.ncap), dtype=torch.int64, device="cuda:0")
    assert s.gather(0, out=dst) == total and torch.equal(dst[:total], whole)
    h = s.gather(0, host=True)
    assert np.array_equal(h, whole.cpu().numpy())
    # count-only rounds (FindReaderCount across devices)
    total2, _ = s.round(_windows(torch, buf, plan), count_only=True)
    assert total2 == total
    s.close()


def test_find_all_bytes_host_buffer_against_the_oracle(gpu):
    from oracle.gen_c import CMatcher
    from regengo_amd import Compiled, synth
    from regengo_amd.sharded import Sharded
    buf = synth.date_log_np(3_000_017, adversarial=True)
    exp, cnt = CMatcher(DATE).find_all_np(buf)
    c = Compiled(DATE).to(0)
    for ndev in (1, 2, 4):
        s = Sharded(c, devices=[0] * ndev)
        rows, res = s.find_all_bytes(buf.tobytes())
        assert res.total == cnt and np.array_equal(rows, exp), ndev
        rows5, _ = s.find_all_bytes(buf.tobytes(), n=5)
        assert np.array_equal(rows5, exp[:5])
        with pytest.raises(Exception) as ei:                    # the capacity contract of rgx_find_all_bytes
            s.find_all_bytes(buf.tobytes(), capacity=10)
        assert getattr(ei.value, "status", None) == -8
        s.close()


def test_halo_without_a_sync_point_is_reported_and_widened(gpu):
    torch = gpu
    from oracle.engines import Compiled as O
    from regengo_amd import Compiled
    from regengo_amd.sharded import Sharded
    # \\w+@\\w+ resets on anything but a word byte or '@': a 40 KB run of word bytes across the second shard's range start leave its 4 KiB
    # left halo without a sync point -- the round says so (count 0, unsynced) and rgx_sharded_find_all_bytes widens until it has one
    data = b"mail a@b then " + b"x" * 40000 + b"y@z and c@d end"
    half = -(-len(data) // 2)
    assert b" " not in data[half - 4100:half + 16]
    c = Compiled(EMAIL).to(0)
    s = Sharded(c, devices=[0, 0])
    buf = torch.frombuffer(bytearray(data), dtype=torch.uint8).to("cuda:0")
    total, rs = s.round(_windows(torch, buf, s.plan(len(data))))
    assert rs[1]["unsynced"] and rs[1]["count"] == 0 and not rs[0]["unsynced"]
    rows, res = s.find_all_bytes(data)
    assert rows.tolist() == O(EMAIL).FindAllBytes(data)
    s.close()


def test_two_rounds_in_flight(gpu):
    torch = gpu
    from regengo_amd import Compiled
    from regengo_amd.sharded import Sharded
    tile = _tile()
    c = Compiled(URL).to(0)
    s = Sharded(c, devices=[0, 0])
    bufs = [torch.frombuffer(bytearray(tile[k * 1000:] + tile[:k * 1000]), dtype=torch.uint8).to("cuda:0") for k in range(4)]
    exp = [c.FindAllSpans(b)[0].shape[0] for b in bufs]
    w = [_windows(torch, b, s.plan(b.numel())) for b in bufs]
    got = []
    s.submit(w[0])
    s.submit(w[1])
    with pytest.raises(Exception) as ei:
        s.submit(w[2])                       # a third round in flight is refused
    assert getattr(ei.value, "status", None) == -1
    got.append(s.wait()[0])                  # round 0 (its rows stay valid while round 1 runs: a slot of its own)
    s.submit(w[2])
    got.append(s.wait()[0])
    s.submit(w[3])
    got.append(s.wait()[0])
    got.append(s.wait()[0])
    assert got == exp
    s.close()


def test_refused_programs_are_refused_here_too(gpu):
    torch = gpu
    from regengo_amd import Compiled, _capi
    from regengo_amd.sharded import Sharded
    tdfa_class = r"(?P<protocol>https?)://(?P<host>[\w\.-]+)(?::(?P<port>\d+))?(?P<path>/[\w\./]*)?"
    c = Compiled(tdfa_class).to(0)
    s = Sharded(c, devices=[0, 0])
    with pytest.raises(_capi.RgxError) as ei:
        s.find_all_bytes(b"see https://example.com/a and http://h.org:80/x")
    assert ei.value.status == _capi.RGX_E_UNSUPPORTED
    s.close()


_RANK_SCRIPT = r"""
import sys, os
sys.path.insert(0, %r)
import numpy as np, torch
from regengo_amd import Compiled, synth
from regengo_amd.sharded import Sharded
URL = %r
tile = synth.web_log_tile(1 << 19)
c = Compiled(URL).to(0)
buf = torch.frombuffer(bytearray(tile), dtype=torch.uint8).to("cuda:0")
whole = c.FindAllSpans(buf)[0].to(torch.int64)
uid = Sharded.unique_id()
assert len(uid) == 128
s = Sharded(c, device=0, rank=0, world=1, uid=uid)
assert s.uses_rccl, "RGX_SHARDED_FORCE_RCCL=1 must give a world of one a real communicator"
total, rs = s.round_counts([dict(buf=buf, own=(0, buf.numel()), base=0, starts_at_sync=True, last=True)])
assert total == whole.shape[0] and rs[0]["count"] == total, (total, whole.shape)
dst = torch.empty((total + 1, c.ncap), dtype=torch.int64, device="cuda:0")
assert s.gather(0, out=dst) == total and torch.equal(dst[:total], whole)
s.close()
print("RCCL_WORLD1_OK", total)
"""


def test_world_of_one_through_rccl(gpu):
    """The RCCL plumbing on one GPU: dlopen, ncclGetUniqueId, ncclCommInitRank(world 1), the round's ncclAllGather, the gather's
    group -- in a process of its own (a communicator that fails to come up must not take the test session with it)."""
    env = dict(os.environ, RGX_SHARDED_FORCE_RCCL="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", _RANK_SCRIPT % (ROOT, URL)], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "RCCL_WORLD1_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])
