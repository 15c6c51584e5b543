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
    torch.cuda.synchronize()          # the copies run on torch's stream, the scans on the library's own: they have to be there first
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
    # the compact form: one 64-bit word per match, start | length << 40 (rgx_sharded_gather_offsets) -- device buffer and host copy
    words = torch.empty(total + 8, dtype=torch.int64, device="cuda:0")
    assert s.gather_offsets(0, out=words) == total
    st, en = Sharded.split_offsets(words[:total])
    assert torch.equal(st, whole[:, 0]) and torch.equal(en, whole[:, 1])
    hw = s.gather_offsets(0, host=True).astype(np.int64)
    hst, hen = Sharded.split_offsets(hw)
    assert np.array_equal(hst, whole[:, 0].cpu().numpy()) and np.array_equal(hen, whole[:, 1].cpu().numpy())
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


def _build_shim():
    build = os.path.join(ROOT, "tests", "_build")
    os.makedirs(build, exist_ok=True)
    shim = os.path.join(build, "libccl_shim.so")
    r = subprocess.run(["gcc", "-shared", "-fPIC", "-O1", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                        os.path.join(ROOT, "tests", "ccl_shim.c"), "-L/opt/rocm/lib", "-lamdhip64", "-lrt", "-Wl,-rpath,/opt/rocm/lib", "-o", shim],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return shim


@pytest.mark.parametrize("nranks", [2, 8])
def test_bench_starts_its_own_ranks(gpu, nranks):
    """`python bench.py --gpus N` with NO launcher in front starts N ranks itself (torch.distributed.run
    inside), the library forms a world of N (over the CCL test double on this one-GPU box: RGX_BENCH_ONE_DEVICE=1), and the line says so:
    n_gpus N, ranks_formed N, the gathered table checked on rank 0.  Without the exception variable, asking for more GPUs than the
    box has is an error -- never a silent one-GPU run that prints n_gpus: 1."""
    import json
    torch = gpu
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(RGX_SHARDED_CCL_LIB=_build_shim(), RGX_BENCH_ONE_DEVICE="1", RGX_BENCH_PREWARM="0.2")
    env.pop("RGX_SHARDED_NO_RCCL", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(nranks), "--bytes", str(1 << 26), "--steps", "3", "--warmup", "1",
                        "--no-cpu-baseline", "--no-other-configs"], capture_output=True, text=True, timeout=1500, cwd=ROOT, env=env)
    rows = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert p.returncode == 0 and len(rows) == 1, (p.stdout[-2000:], p.stderr[-3000:])
    j = json.loads(rows[0])
    cfg = j["config"]
    assert j["n_gpus"] == nranks and cfg["ranks_formed"] == nranks and cfg["gather_rows_checked"] is True and cfg["parity_closed_form"] is True, j
    assert cfg["communicator"].startswith("test double") and "FALLBACK" not in cfg["path"], cfg
    assert cfg["matches_total"] == ((nranks << 26) - 10) // 50 + 1 and cfg["strong_scaling"]["parity_count"] is True
    if torch.cuda.device_count() < 2 and nranks == 2:
        env.pop("RGX_BENCH_ONE_DEVICE")
        q = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--bytes", str(1 << 26), "--no-cpu-baseline"],
                           capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
        assert q.returncode != 0 and not any(ln.startswith("{") for ln in q.stdout.splitlines()), q.stdout[-1000:]
        assert "GPU(s) visible" in q.stderr


@pytest.mark.parametrize("world", [2, 4, 8])
def test_multi_process_world(gpu, tmp_path, world):
    """The multi-RANK path of the C library -- rgx_sharded_create_rank(world = 2 / 4 / 8), the per-round exchange ([count, flags, base,
    status] through the all-gather entry point), the grouped send / recv gather to the first and to the LAST rank (world - 1 sources),
    an uneven last round (windows % world != 0: ranks without a window still take part), a failing MIDDLE rank (every rank gets the
    error, none hangs), a stop request raised by a middle rank, count-only rounds, and rounds of FindReader chunk ranges
    (rgx_shard_window::reader_buffer_size) against the oracle's C port of the read loop -- as `world` PROCESSES on device 0.  RCCL cannot
    form such a world on one GPU, so the ten nccl* entry points the library dlopens come from tests/ccl_shim.c (RGX_SHARDED_CCL_LIB;
    staged through POSIX shared memory, every wait bounded): what is under test is the library's protocol, rank arithmetic and error
    paths, not RCCL.  (when a real 8-GPU lease comes, the only new thing is the transport.)"""
    torch = gpu
    import json
    from regengo_amd import Compiled
    from tests import _sharded_rank_worker as W
    shim = _build_shim()
    env = dict(os.environ)
    env["RGX_SHARDED_CCL_LIB"] = shim
    env.pop("RGX_SHARDED_NO_RCCL", None)
    worker = os.path.join(ROOT, "tests", "_sharded_rank_worker.py")
    procs = [subprocess.Popen([sys.executable, worker, str(rk), str(tmp_path), str(world)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for rk in range(world)]
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=900)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("a rank hung")
        outs.append(o.decode(errors="replace"))
    assert all(p.returncode == 0 for p in procs), "\n----\n".join(o[-3000:] for o in outs)
    ranks = [json.load(open(os.path.join(str(tmp_path), "rank%d.json" % rk))) for rk in range(world)]
    data = W.stream_bytes()
    buf = torch.frombuffer(bytearray(data), dtype=torch.uint8).to("cuda:0")
    nwin = W.nwindows(world)
    nrounds = -(-nwin // world)
    mid = world // 2
    for name, pattern in (("date", DATE), ("url", URL)):
        c = Compiled(pattern).to(0)
        whole = c.FindAllSpans(buf)[0].cpu().numpy().astype(np.int64)
        from regengo_amd.sharded import Sharded
        plan = Sharded(c, devices=[0]).plan(len(data), parts=nwin)

        def owned(k):
            return (whole[:, 0] >= plan[k][0]) & (whole[:, 0] < plan[k][1])
        for rd in range(nrounds):
            ks = [k for k in range(rd * world, (rd + 1) * world) if k < nwin]
            exp = whole[(whole[:, 0] >= plan[ks[0]][0]) & (whole[:, 0] < plan[ks[-1]][1])]
            exp_counts = [int(owned(rd * world + r).sum()) if rd * world + r < nwin else 0 for r in range(world)]
            for rk in range(world):
                R = ranks[rk][name]["rounds"][rd]
                assert R["counts"] == exp_counts and R["total"] == len(exp) and R["status"] == [0] * world and R["unsynced"] == [False] * world, (name, rd, rk, R)
                assert R["have"] == [rd * world + r < nwin for r in range(world)], (name, rd, R["have"])
            # the gathered table, on rank 0 and on the last rank: rank order = stream order, stream-absolute offsets
            for rk in (0, world - 1):
                got = [t for (r_, dst, t) in ranks[rk][name]["tables"] if r_ == rd and dst == rk]
                assert len(got) == 1 and np.array_equal(np.array(got[0], dtype=np.int64).reshape(-1, c.ncap), exp), (name, rd, rk)
                gw = [t for (r_, dst, t) in ranks[rk][name]["offset_tables"] if r_ == rd and dst == rk]
                w = np.array(gw[0], dtype=np.int64)
                assert len(gw) == 1 and np.array_equal(w & ((1 << 40) - 1), exp[:, 0]) and np.array_equal((w & ((1 << 40) - 1)) + (w >> 40), exp[:, 1]), (name, rd, rk)
        per = [int(owned(r).sum()) for r in range(world)]
        for rk in range(world):
            res = ranks[rk][name]
            assert res["fail"] == -1, (name, rk, res["fail"])                  # RGX_E_INVALID of the middle rank's window, seen by EVERY rank
            assert res["stop_seen"] == [r == mid for r in range(world)]
            assert res["after_fail_counts"] == per, (name, rk, res["after_fail_counts"], per, res["after_fail_flags"])
            assert res["after_fail_total"] == sum(per) and res["count_only"] == per
            # starts beyond 2^40 on the middle rank: RGX_E_TOO_LARGE (-4) there and on rank 0, which receives the words; the records go through
            assert res["offsets_overflow"] == (-4 if rk in (0, mid) and per[mid] > 0 else 0), (name, rk, res["offsets_overflow"])
            assert res["far_gather_rows"] == (sum(per) if rk == 0 else 0), (name, rk, res["far_gather_rows"])
        # ---- the reader rounds: the reference's FindReader callbacks over the whole stream, in stream order on the last rank
        from oracle.gen_c import CMatcher
        B, ML = ranks[0][name]["reader_cfg"]
        exp = CMatcher(pattern).find_reader_np(np.frombuffer(data, dtype=np.uint8), B, ML)
        last = ranks[world - 1][name]["reader_rounds"]
        starts = np.array([x for e in last for x in e["starts"]], dtype=np.int64)
        ends = np.array([x for e in last for x in e["ends"]], dtype=np.int64)
        assert np.array_equal(starts, exp[:, 0]) and np.array_equal(ends, exp[:, 2] + exp[:, 4]), (name, len(starts), exp.shape)
        for rk in range(world):
            assert [e["total"] for e in ranks[rk][name]["reader_rounds"]] == [e["total"] for e in last]
            assert sum(e["total"] for e in ranks[rk][name]["reader_rounds"]) == exp.shape[0]


def test_sharded_find_all_bytes_widens_truncated_windows(gpu, monkeypatch):
    """an unbounded pattern whose owned match reaches past the right halo of a window that is not the last.  The
    scan of that window sees an end of text that is none (`$` fires, a greedy match stops, the next shard drops the match as not
    owned): rgx_shard_round.truncated says so and rgx_sharded_find_all_bytes scans again with a wider right halo -- the merged table
    equals ONE scan of the whole buffer, as rgx.h promises.  (The default right halo is the reference's 1 MiB leftover cap; a match
    longer than that is a megabyte without a sync point, which the scan kernels refuse as quadratic -- so the halo is shrunk to 4 KiB
    here, RGX_SHARDED_HALO_RIGHT, and the match is 40 KB.)"""
    torch = gpu
    from regengo_amd import Compiled
    from regengo_amd.sharded import Sharded
    monkeypatch.setenv("RGX_SHARDED_HALO_RIGHT", "4096")
    line = b"GET /index.html 200 alpha beta\n"
    head = (line * (300_000 // len(line) + 1))[:300_000 - 512 - 1] + b"\n"
    run = b"x" * 40_000                                  # one match of 40 KB that starts 512 bytes before the edge of the two shards
    data = head + run + b" tail\n"
    data += (line * (600_000 // len(line)))[: 600_000 - len(data)]
    buf = torch.frombuffer(bytearray(data), dtype=torch.uint8).to("cuda:0")
    for pattern in (r"(?P<w>[a-z]+)", r"(?P<l>[^\n]+)", r"(?P<k>x+)(?P<t> tail)?$|GET"):
        c = Compiled(pattern).to(0)
        assert c.info.max_match_len < 0
        whole = c.FindAllSpans(buf)[0].cpu().numpy()
        s = Sharded(c, devices=[0, 0])
        got, res = s.find_all_bytes(data)
        assert res.total == len(whole) and np.array_equal(got, whole), (pattern, res.total, len(whole))
        # the round itself reports the window as truncated (what a caller of rgx_sharded_round sees): 4 KiB of right halo inside the run
        lo, hi = 0, 300_000
        w1, w2 = buf[0:hi + 4096].clone(), buf[0:hi + 65536].clone()
        torch.cuda.synchronize()
        total, rs = s.round([dict(buf=w1, own=(lo, hi), base=0, starts_at_sync=True, last=False), None])
        assert rs[0]["truncated"], pattern
        total, rs = s.round([dict(buf=w2, own=(lo, hi), base=0, starts_at_sync=True, last=False), None])
        assert not rs[0]["truncated"], pattern
        s.close()


def test_starts_only_rounds(gpu):
    """rgx_shard_window::starts_only: a round's rows as 4-byte match starts (fixed-template programs on the exact kernel) -- two logical
    shards, asynchronous and slot paths, two rounds in flight; start + the capture template == the full records of the same window;
    the gather behind such a round and a program without a fixed template are refused."""
    torch = gpu
    from regengo_amd import Compiled, _capi, synth
    from regengo_amd.sharded import Sharded
    c = Compiled(DATE).to(0)
    tmpl, _ = c.capture_template()
    tm = torch.tensor(tmpl, dtype=torch.int32, device="cuda:0")
    buf = synth.date_log_torch(8 << 20, torch.device("cuda:0"), adversarial=True)
    full = c.FindAllSpans(buf)[0]
    s = Sharded(c, devices=[0, 0])
    plan = s.plan(buf.numel())
    outs = [[torch.empty(buf.numel() // 10 + 16, dtype=torch.int32, device="cuda:0") for _ in plan] for _ in range(2)]

    def windows(k):
        ws = _windows(torch, buf, plan, outs[k])
        for w in ws:
            w["starts_only"] = True
        return ws

    s.submit(windows(0))
    s.submit(windows(1))
    for k in range(2):
        total, rs = s.wait()
        assert total == full.shape[0] and not any(r["unsynced"] for r in rs)
        rows = []
        for i, (lo, hi, wl, wh) in enumerate(plan):
            st = outs[k][i][:rs[i]["count"]]
            rows.append(st[:, None] + tm[None, :] + wl)
        assert torch.equal(torch.cat(rows), full)
    with pytest.raises(_capi.RgxError) as ei:
        s.gather(0)
    assert ei.value.status == _capi.RGX_E_UNSUPPORTED
    total, rs = s.round(_windows(torch, buf, plan))            # a full-record round afterwards: the gather is offered again
    assert s.gather(0) == full.shape[0]
    s.close()
    # the tail window of a stream may be a few bytes long -- starts-only is a property of the program, not of the window
    # (the exact kernel wants 64 bytes; the kernel that takes a shorter window writes starts as well)
    s2 = Sharded(c, devices=[0])
    for tail in (b"x 2024-01-15 y 1999-12-31", b"2024-01-15", b"abc", b"x" * 63):
        tb = torch.frombuffer(bytearray(tail), dtype=torch.uint8).cuda()
        so = torch.empty(8, dtype=torch.int32, device="cuda")
        total, rs = s2.round([dict(buf=tb, own=(0, len(tail)), base=0, starts_at_sync=True, last=True, out=so, starts_only=True)])
        want = c.FindAllSpans(tail)[0][:, 0]
        assert total == want.shape[0] and torch.equal(so[:total], want), tail
    s2.close()
    e = Compiled(EMAIL).to(0)
    se = Sharded(e, devices=[0])
    tile = torch.frombuffer(bytearray(_tile()), dtype=torch.uint8).cuda()
    w = _windows(torch, tile, se.plan(tile.numel()))
    w[0]["starts_only"] = True
    with pytest.raises(_capi.RgxError) as ei:
        se.round(w)
    assert ei.value.status == _capi.RGX_E_UNSUPPORTED
    se.close()


def test_halo_checks_at_the_edges_of_their_ranges(gpu):
    """The two halo checks of a round (halo_sync_kernel: whole aligned 16-byte chunks, the bytes of a chunk outside the range masked
    out): a reset byte counts exactly when it lies in [0, own_lo) for the left halo and in [own_hi - 1, len) for the right one -- at the
    first byte, at the last byte, at every alignment of the range's start, and not one byte outside."""
    torch = gpu
    from regengo_amd import Compiled
    from regengo_amd.sharded import Sharded
    c = Compiled(r"(?P<w>[a-z]+)").to(0)               # unbounded; every byte that is not a lower-case letter is a reset byte
    assert c.info.max_match_len < 0
    s = Sharded(c, devices=[0])
    n = 4096
    for own_lo, own_hi in ((16, 2000), (17, 2001), (31, 2015), (1, 4000), (333, 4095)):
        for where, expect_unsynced, expect_truncated in (
                (None, True, True),                     # letters only: neither halo holds a reset byte
                (0, False, True), (own_lo - 1, False, True), (own_lo, True, True),          # left range [0, own_lo)
                (own_hi - 2, True, True), (own_hi - 1, True, False), (n - 1, True, False)):    # right range [own_hi - 1, n)
            data = bytearray(b"a" * n)
            if where is not None:
                data[where] = ord(" ")
            buf = torch.frombuffer(data, dtype=torch.uint8).to("cuda:0")
            total, rs = s.round([dict(buf=buf, own=(own_lo, own_hi), base=0, starts_at_sync=False, last=False)])
            assert bool(rs[0]["unsynced"]) == expect_unsynced, (own_lo, own_hi, where, rs[0])
            if not expect_unsynced:
                # (a window that is vouched for reports its right edge too)
                assert bool(rs[0]["truncated"]) == expect_truncated, (own_lo, own_hi, where, rs[0])
        # both halos in order: the window is neither unsynced nor truncated, and its rows are the owned matches
        data = bytearray(b"a" * n)
        data[own_lo - 1] = ord(" ")
        data[own_hi + 5 if own_hi + 5 < n else n - 1] = ord(" ")
        buf = torch.frombuffer(data, dtype=torch.uint8).to("cuda:0")
        total, rs = s.round([dict(buf=buf, own=(own_lo, own_hi), base=0, starts_at_sync=False, last=False)])
        assert not rs[0]["unsynced"] and not rs[0]["truncated"] and total == 1, (own_lo, own_hi, rs[0])
    s.close()


@pytest.mark.gpu
def test_sharded_find_all_of_random_patterns(gpu):
    """RANDOM patterns through rgx_sharded_find_all_bytes over 2, 3 and 5 logical shards of one device (windows with halos, sync points
    in the halos, windows widened where a halo holds none, matches across window edges): the rows of the stream == the rows of ONE scan
    of the same program over the same bytes (itself against the oracle's C port in the differential sweeps, tests/test_gpu_fuzz.py) --
    or both refuse."""
    torch = gpu
    import random
    from oracle import engines as E
    from regengo_amd import Compiled, _capi
    from regengo_amd.sharded import Sharded
    from tests import _fuzzgen as F
    rng = random.Random(1234)
    pats = compared = refused = rows_total = 0
    for seed in F.fuzz_seeds(1000, 1002):
        for pat in F.gen_patterns(seed, 30):
            try:
                o = E.Compiled(pat)
            except Exception:
                continue
            if F.has_empty_loop(o.prog) and not o.find_machine.memo:
                continue
            try:
                c = Compiled(pat).to(0)
            except _capi.RgxError:
                continue
            if c.info.ref_findall_offered != 1:
                continue
            parts, total = [], 0
            while total < 600_000:
                s = F.gen_input(rng, rng.choice([3, 10, 40, 120, 400])) + rng.choice([b" ", b"\n", b"", b"  "])
                parts.append(s)
                total += len(s)
            text = b"".join(parts)
            try:
                whole = c.FindAllSpans(text)[0].cpu().numpy()
            except _capi.RgxError as ex:
                assert ex.status in (_capi.RGX_E_UNSUPPORTED, _capi.RGX_E_DIVERGES), (pat, ex)
                whole = None
            pats += 1
            if whole is not None:
                # the same bytes again on the same program (what it learned in the first call picks other passes: the carry pass
                # instead of exact sync points) and on a program frozen before its first call (no learning at all): the same rows.
                # [Round 6: the rescan behind the carry pass skipped slices that had found their sync point in the far look-behind --
                # 11 of 6465 matches of `[^a][a-b0-1](a-|a|a){2}(?:1\.)*` were lost on every call but a program's first.]
                again = c.FindAllSpans(text)[0].cpu().numpy()
                assert np.array_equal(again, whole), (pat, "second call", again.shape, whole.shape)
                cf = Compiled(pat).to(0)
                cf.freeze()
                fr = cf.FindAllSpans(text)[0].cpu().numpy()
                assert np.array_equal(fr, whole), (pat, "frozen program", fr.shape, whole.shape)
            for ndev in (2, 3, 5):
                s = Sharded(c, devices=[0] * ndev)
                try:
                    rows, res = s.find_all_bytes(text)
                except _capi.RgxError as ex:
                    assert ex.status in (_capi.RGX_E_UNSUPPORTED, _capi.RGX_E_DIVERGES), (pat, ndev, ex)
                    refused += 1
                    s.close()
                    continue
                s.close()
                assert whole is not None, (pat, ndev, "the shards answered what one scan refused")
                assert res.total == whole.shape[0] and np.array_equal(rows, whole), (pat, ndev, res.total, whole.shape)
                compared += 1
                rows_total += whole.shape[0]
    print("patterns", pats, "sharded scans compared", compared, "refused", refused, "rows", rows_total)
    if F.fuzz_default():
        assert pats >= 40 and compared >= 90, (pats, compared, refused)
