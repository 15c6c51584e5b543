"""ShardedReader (regengo_amd/dist.py: FindReader over the ranks) on CPU: world_size-2 `gloo` processes, the per-window scan
injected from the TEST-ONLY table walker (tests/_hosttest.py; the HIP kernel needs a GPU).  Checked against one scan of the whole
stream by the same walker: rows, stream order, global row bases, count-only mode, early stop, a sequential reader as the source,
an unbounded pattern whose match crosses a window (= rank) boundary, and the widened-halo path."""
import io
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


def _host_scan(hp):
    """The per-window primitives ShardedReader needs, from the host walker."""
    queue = []
    ncap = hp.info["ncap"]

    def rows_of(window, own):
        b = bytes(window.numpy().tobytes())
        t = torch.tensor(hp.find_all(b), dtype=torch.int32).reshape(-1, ncap)
        if own is not None and t.shape[0]:
            t = t[(t[:, 0] >= own[0]) & (t[:, 0] < own[1])]
        return t

    def submit(window, own, slot):
        queue.append(rows_of(window, own))

    def wait():
        return queue.pop(0), {"kernel_ms": 0.0, "unsynced": 0}

    def count(window, own):
        return int(rows_of(window, own).shape[0])

    return {"submit": submit, "wait": wait, "count": count, "max_match_len": hp.info["max"], "ncap": ncap,
            "reset_table": torch.tensor(list(hp.reset_bytes()), dtype=torch.uint8)}


def _worker(rank, world, port, pattern, data_bytes, kw, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from regengo_amd.dist import DeviceSource, ReaderSource, ShardedReader
        from tests._hosttest import HostProgram
        hp = HostProgram(pattern)
        data = np.frombuffer(data_bytes, dtype=np.uint8)
        rd = ShardedReader(device="cpu", window_bytes=kw["W"], halo_left=kw.get("halo_left", 64), halo_max=kw.get("halo_max", 1 << 16),
                           unbounded_halo=kw.get("unbounded_halo", 4096), scan=_host_scan(hp))
        if kw.get("reader"):
            src = ReaderSource(io.BytesIO(data_bytes), "cpu", block=kw.get("block", 1000))
        else:
            src = DeviceSource(lambda lo, hi: torch.from_numpy(data[lo:hi].copy()), len(data))
        got = []
        bases = []
        stop_after = kw.get("stop_after")

        def on_rows(rows, k, base, win_lo=None):
            if win_lo is not None:          # absolute=False: int32 rows relative to the window's first byte + that byte's offset
                assert rows.dtype == torch.int32
                from regengo_amd.dist import to_global
                rows = to_global(rows, win_lo)
            got.append((k, rows.tolist()))
            bases.append(base)
            if stop_after is not None and len(got) >= stop_after:
                return False

        seen = []

        def on_match(m):
            seen.append((m["StreamOffset"], m["ChunkIndex"]))
            return True

        try:
            st = rd.find_reader(src, on_rows=None if kw.get("count_only") else on_rows, on_match=on_match if kw.get("per_match") else None,
                                gather=kw.get("gather", False), count_only=kw.get("count_only", False), absolute=not kw.get("relative", False))
        except RuntimeError as ex:
            if not kw.get("expect_error"):
                raise
            st = {"error": str(ex)}
        q.put((rank, got, bases, st, seen))
    finally:
        if world > 1:
            dist.destroy_process_group()


def _run(pattern, data, world=2, **kw):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, pattern, data, kw, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = sorted([q.get(timeout=180) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    return outs


def _expect(pattern, data):
    from tests._hosttest import HostProgram
    hp = HostProgram(pattern)
    return [list(r) for r in np.array(hp.find_all(data), dtype=np.int64).reshape(-1, hp.info["ncap"]).tolist()]


def _log(n, seed=3):
    rng = np.random.default_rng(seed)
    words = [b"2024-01-15", b"GET", b"/index.html", b"1999-12-31", b"x", b"200", b"0000-00-00 ", b"12-34", b"2024-1-15"]
    out = bytearray()
    while len(out) < n:
        out += words[int(rng.integers(len(words)))] + (b"\n" if rng.integers(6) == 0 else b" ")
    return bytes(out[:n])


def _merge(outs, world):
    """rows of all ranks in window order"""
    allw = []
    for _, got, _, _, _ in outs:
        allw += got
    allw.sort(key=lambda t: t[0])
    return [r for _, rows in allw for r in rows]


def test_two_ranks_rows_order_and_bases(built):
    data = _log(40_000)
    exp = _expect(DATE, data)
    outs = _run(DATE, data, W=4096)
    assert _merge(outs, 2) == exp
    for rank, got, bases, st, _ in outs:
        assert st["count"] == len(exp) and not st["stopped"] and st["widened_halos"] == 0
        assert all(k % 2 == rank for k, _ in got) and [k for k, _ in got] == sorted(k for k, _ in got)
        for (k, rows), base in zip(got, bases):       # the global row index of a window's first row
            if rows:
                assert exp[base] == rows[0]
    assert sum(st["bytes"] for _, _, _, st, _ in outs) == len(data)


def test_gather_delivers_in_stream_order_on_rank0(built):
    data = _log(30_000, seed=5)
    exp = _expect(DATE, data)
    outs = _run(DATE, data, W=2048, gather=True, per_match=True)
    r0, r1 = outs
    assert r1[1] == [] and r1[4] == []
    assert [k for k, _ in r0[1]] == sorted(k for k, _ in r0[1])
    assert [r for _, rows in r0[1] for r in rows] == exp
    assert [s for s, _ in r0[4]] == [r[0] for r in exp]                       # per-match callbacks: StreamOffset in order
    assert all(c == s // 2048 for s, c in r0[4])                              # ChunkIndex = the owning window


def test_count_only_and_single_rank(built):
    data = _log(25_000, seed=7)
    exp = _expect(DATE, data)
    outs = _run(DATE, data, W=4096, count_only=True)
    assert all(st["count"] == len(exp) for _, _, _, st, _ in outs)
    one = _run(DATE, data, world=1, W=4096)
    assert _merge(one, 1) == exp and one[0][3]["count"] == len(exp)


def test_sequential_reader_source(built):
    data = _log(33_333, seed=9)
    exp = _expect(DATE, data)
    for world in (1, 2):
        outs = _run(DATE, data, world=world, W=4096, reader=True, block=777)
        assert _merge(outs, world) == exp
        assert sum(st["bytes"] for _, _, _, st, _ in outs) == len(data)
    # a stream whose length is a multiple of the window: the last window ends exactly at EOF
    data = _log(4096 * 4, seed=11)
    outs = _run(DATE, data, W=4096, reader=True)
    assert _merge(outs, 2) == _expect(DATE, data)


def test_unbounded_match_across_the_rank_boundary(built):
    pat = r"(?P<w>[a-z]+)=(?P<v>\d+)"
    filler = b"key=1 " * 600
    long_word = b"q" * 3000                 # starts in window 0 (rank 0), ends in window 1 (rank 1)
    data = filler[:3000] + long_word + b"=42 " + filler
    exp = _expect(pat, data)
    assert any(r[0] < 4096 < r[1] for r in exp)
    outs = _run(pat, data, W=4096, unbounded_halo=4096)
    assert _merge(outs, 2) == exp
    assert all(st["count"] == len(exp) for _, _, _, st, _ in outs)


def test_halo_without_sync_point_is_widened(built):
    # 200 digits in front of the boundary: the 64-byte halo holds no reset byte ('\d' and '-' keep the Date DFA alive only
    # sometimes -- digits alone never kill it), the widened one does
    data = b"a 2024-01-15 b " * 700
    data = data[:4096 - 150] + b"7" * 200 + data[4096 + 50:]
    exp = _expect(DATE, data)
    outs = _run(DATE, data, W=4096, halo_left=64)
    assert _merge(outs, 2) == exp
    assert sum(st["widened_halos"] for _, _, _, st, _ in outs) >= 1


def test_early_stop_reaches_every_rank(built):
    data = _log(60_000, seed=13)
    outs = _run(DATE, data, W=4096, stop_after=2)
    for _, got, _, st, _ in outs:
        assert st["stopped"] and st["rounds"] <= 4


def test_to_global_keeps_unset_groups():
    from regengo_amd.dist import to_global
    t = torch.tensor([[5, 9, 0, 0, 6, 7], [0, 3, 0, 2, 0, 0]], dtype=torch.int32)
    assert to_global(t, 100).tolist() == [[105, 109, 0, 0, 106, 107], [100, 103, 100, 102, 0, 0]]
    # FLAG_UNMATCHED_MINUS1: an unset group is (-1, -1) and stays so (it used to become base - 1)
    t = torch.tensor([[5, 9, -1, -1, 6, 7]], dtype=torch.int32)
    assert to_global(t, 100).tolist() == [[105, 109, -1, -1, 106, 107]]


def test_a_rank_that_cannot_widen_its_halo_fails_every_rank(built):
    """widen() used to raise on ONE rank while the others were already blocked in the round's all_gather -- a hang.  The
    failure now travels with the exchange (a fourth int) and every rank raises together.  Here no halo up to halo_max holds a
    sync point for the window behind a run of 3000 digits."""
    data = b"a 2024-01-15 b " * 300
    data = data[:1000] + b"7" * 3200 + data[4200:]
    outs = _run(DATE, data, W=4096, halo_left=64, halo_max=1024, expect_error=True)
    assert all("error" in st for _, _, _, st, _ in outs), [st for _, _, _, st, _ in outs]


def test_reader_source_with_a_right_halo_longer_than_the_stream(built):
    # an unbounded pattern's right halo (here 64 KiB) sees the end of the stream from the first window on: the following
    # windows must still be produced
    pat = r"(?P<w>[a-z]+)=(?P<v>\d+)"
    data = b"key=1 other=22 x=333 " * 1500
    exp = _expect(pat, data)
    for world in (1, 2):
        outs = _run(pat, data, world=world, W=4096, reader=True, unbounded_halo=1 << 16)
        assert _merge(outs, world) == exp


def test_window_relative_rows(built):
    """absolute=False: the rows stay int32 and window-relative (what the kernel wrote), the window's stream offset comes along."""
    data = _log(30_000, seed=21)
    exp = _expect(DATE, data)
    for world in (1, 2):
        outs = _run(DATE, data, world=world, W=4096, relative=True)
        assert _merge(outs, world) == exp
