"""N>1 path on CPU: world_size-2 `gloo` processes exercise the sharding logic of regengo_amd/dist.py (ownership,
halos, sync check, the one-int64 carry chain, count all_gather, variable-length span gather).  The per-shard
scanner injected here is the TEST-ONLY table walker (the HIP kernel needs a GPU); on the GPU box the same class
is driven by Compiled.FindAllSpans (bench.py)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

DATE = r"(?P<year>\d{4})-(?P<month>\d{2})-(?P<day>\d{2})"


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, pattern, data_bytes, halo_left, q, combined=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from regengo_amd.dist import ShardedFinder, plan_shards
        from tests._hosttest import HostProgram
        hp = HostProgram(pattern)
        data = np.frombuffer(data_bytes, dtype=np.uint8)
        shards = plan_shards(len(data), world, hp.info["max"], halo_left=halo_left)
        sh = shards[rank]
        window = torch.from_numpy(data[sh.win_lo:sh.win_hi].copy())

        def scan(w):
            b = bytes(w.numpy().tobytes())
            rows = hp.find_all(b)
            t = torch.tensor(rows, dtype=torch.int32).reshape(-1, hp.info["ncap"])
            return t, {"kernel_ms": 0.0, "unsynced": 0}

        def scan_owned(w, lo, hi):     # what rgx_find_all_bytes_device_owned does: whole-window chain, owned starts only
            t, info = scan(w)
            keep = (t[:, 0] >= lo) & (t[:, 0] < hi) if t.shape[0] else torch.zeros(0, dtype=torch.bool)
            return t[keep], info

        rt = torch.tensor(list(hp.reset_bytes()), dtype=torch.uint8)
        if combined:
            f = ShardedFinder(scan, rt, scan_owned_fn=scan_owned, bounded=hp.info["max"] >= 0)
            owned, cnt, info, base, total, counts = f.find_all_sharded(window, sh, "cpu")
        else:
            f = ShardedFinder(scan, rt)
            owned, cnt, info = f.find_all_local(window, sh)
            base, total, counts = f.global_row_base(cnt, "cpu")
        allrows = f.gather_spans(owned, sh, counts)
        if rank == 0:
            q.put((allrows.tolist(), total, counts, info["chained"], info.get("redone")))
        else:
            q.put((None, total, counts, info["chained"], info.get("redone")))
    finally:
        dist.destroy_process_group()


def _run(pattern, data, halo_left=4096, world=2, combined=False):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, pattern, data, halo_left, q, combined)) for r in range(world)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    rows = [o for o in outs if o[0] is not None][0]
    return rows, outs


def test_plan_shards_tiles_exactly():
    from regengo_amd.dist import plan_shards
    for total in (1, 100, 65536, (1 << 20) + 7):
        for world in (1, 2, 3, 8):
            sh = plan_shards(total, world, 10)
            assert sh[0].lo == 0 and sh[-1].hi == total
            for a, b in zip(sh, sh[1:]):
                assert a.hi == b.lo
            for s in sh:
                assert s.win_lo <= s.lo and s.win_hi >= s.hi and s.win_lo % 16 == 0
                assert s.win_hi == min(total, s.hi + 10)


def test_two_ranks_equal_single_scan(built):
    from oracle.gen_c import CMatcher
    from regengo_amd import synth
    data = synth.date_log_np(300000, adversarial=True).tobytes()
    (rows, total, counts, chained, _redone), outs = _run(DATE, data)
    exp, cnt = CMatcher(DATE).find_all_np(np.frombuffer(data, dtype=np.uint8))
    assert total == cnt and sum(counts) == cnt and not chained
    assert rows == exp.astype(np.int64).tolist()


def test_carry_chain_when_no_sync_in_halo(built):
    """A run of digits/dashes across the shard boundary longer than the left halo: ranks chain the search position."""
    from oracle.gen_c import CMatcher
    rng = np.random.default_rng(11)
    body = rng.choice(np.frombuffer(b"0123456789-", dtype=np.uint8), size=40000).tobytes()
    data = b"start " + body + b" end 2024-01-15 "
    (rows, total, counts, chained, _redone), outs = _run(DATE, data, halo_left=64)
    exp, cnt = CMatcher(DATE).find_all_np(np.frombuffer(data, dtype=np.uint8))
    assert all(o[3] for o in outs), "expected the chained path"
    assert total == cnt
    assert rows == exp.astype(np.int64).tolist()


def test_unbounded_pattern_two_ranks(built):
    from oracle.gen_c import CMatcher
    pat = r"(?P<user>\w+)@(?P<domain>\w+)"
    rng = np.random.default_rng(2)
    words = [b"bob", b"alice_1", b"@", b" ", b"host9", b"a@b", b"\n", b"x@", b"@y"]
    data = b"".join(words[i] for i in rng.integers(0, len(words), size=60000))
    (rows, total, counts, chained, _redone), outs = _run(pat, data)
    exp, cnt = CMatcher(pat).find_all_np(np.frombuffer(data, dtype=np.uint8))
    assert total == cnt
    assert rows == exp.astype(np.int64).tolist()


def test_combined_step_two_ranks(built):
    """find_all_sharded (what bench.py times for N>1): ownership-aware scan + ONE all_gather of [count, no-sync flag];
    both its fast path and its fall-back to the hand-off chain equal the single scan."""
    from oracle.gen_c import CMatcher
    from regengo_amd import synth
    data = synth.date_log_np(200000, adversarial=True).tobytes()
    (rows, total, counts, chained, redone), outs = _run(DATE, data, combined=True)
    exp, cnt = CMatcher(DATE).find_all_np(np.frombuffer(data, dtype=np.uint8))
    assert total == cnt and sum(counts) == cnt and not chained
    assert all(o[4] is False for o in outs), "the fast path must not be redone when every left halo holds a sync point"
    assert rows == exp.astype(np.int64).tolist()
    rng = np.random.default_rng(5)
    body = rng.choice(np.frombuffer(b"0123456789-", dtype=np.uint8), size=30000).tobytes()
    data = b"start " + body + b" end 2024-01-15 "
    (rows, total, counts, chained, redone), outs = _run(DATE, data, halo_left=64, combined=True)
    exp, cnt = CMatcher(DATE).find_all_np(np.frombuffer(data, dtype=np.uint8))
    assert all(o[3] for o in outs), "expected the fall-back to the chained path"
    assert total == cnt and rows == exp.astype(np.int64).tolist()


@pytest.mark.parametrize("pattern,mid", [(r"ab\b", b"abc"), (r"ab\b", b"ab c"), (r"(?m)foo$", b"foox"), (r"(?m)foo$", b"foo\nx"),
                                         (r"ab\B", b"abc"), (r"ab\B", b"ab c"), (r"\bab", b"xab"), (r"x*\b", b"a b")])
def test_lookahead_assertions_across_the_shard_boundary(built, pattern, mid):
    """A maximal-length owned match ending exactly where the right halo used to end (MaxMatchLen-1): the window's end
    was taken for the end of the text and trailing \\b / $ / \\B were evaluated without their real next byte."""
    from oracle.engines import Compiled as OC
    from regengo_amd.dist import plan_shards
    from tests._hosttest import HostProgram
    hp = HostProgram(pattern)
    o = OC(pattern)
    L = 64
    for shift in range(0, len(mid) + 1):
        at = L // 2 - shift          # the owned range of rank 0 ends at 32: slide `mid` across it
        data = bytearray(b" " * L)
        data[at:at + len(mid)] = mid
        data = bytes(data)
        want = [tuple(m[:2]) for m in o.find_machine.find_all(data)]
        got = []
        for sh in plan_shards(L, 2, hp.info["max"], halo_left=16):
            rows = hp.find_all(data[sh.win_lo:sh.win_hi])
            got += [(r[0] + sh.win_lo, r[1] + sh.win_lo) for r in rows if sh.lo <= r[0] + sh.win_lo < sh.hi]
        assert got == want, (pattern, data, got, want)
