"""GPU tier: ShardedReader (FindReader over the ranks, here a group of one) through the C ABI -- windows with halos in shard mode,
submit/wait double buffering, stream-absolute rows -- against ONE FindAllSpans call over the whole stream (itself checked against
the oracle by tests/test_gpu_parity.py) and against the oracle directly on a small stream; count-only mode
(rgx_count_all_device_owned); FindReaderCount (rgx_count_chunk) against the callbacks FindReader makes; bench.py's per-config
lines at reduced size."""
import io
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DATE = r"(?P<year>\d{4})-(?P<month>\d{2})-(?P<day>\d{2})"
URL = r"(?P<full>(?P<proto>https?|ftp)://(?P<host>[\w.-]+)(?P<port>:\d+)?(?P<path>/[\w./-]*)?)"


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


@pytest.mark.parametrize("pattern", [URL, DATE, r"(?P<user>\w+)@(?P<domain>\w+)", r"\b[a-z]+\b"])
def test_windows_equal_one_scan(gpu, pattern):
    torch = gpu
    from regengo_amd import Compiled
    from regengo_amd.dist import DeviceSource, ShardedReader, to_global
    tile = _tile()
    data = (tile * 6)[: 5 * len(tile) + 12345]
    buf = torch.frombuffer(bytearray(data), dtype=torch.uint8).to("cuda:0")
    c = Compiled(pattern, stdlib=True).to(0)
    whole, _ = c.FindAllSpans(buf)
    whole = whole.to(torch.int64)
    for W in (1 << 20, 777_777):
        rd = ShardedReader(c, "cuda:0", window_bytes=W, halo_left=4096)
        got = []
        bases = []
        st = rd.find_reader(DeviceSource(lambda lo, hi: buf[lo:hi].clone(), len(data)), on_rows=lambda r, k, b: (got.append(r.clone()), bases.append(b)) and None)
        rows = torch.cat(got) if got else torch.empty((0, c.ncap), dtype=torch.int64, device="cuda:0")
        assert st["count"] == whole.shape[0] and st["truncated_windows"] == 0
        assert torch.equal(rows, whole)
        assert bases == [int(x) for x in np.cumsum([0] + [g.shape[0] for g in got])[:-1]]
        st2 = rd.find_reader(DeviceSource(lambda lo, hi: buf[lo:hi].clone(), len(data)), count_only=True)
        assert st2["count"] == whole.shape[0]


def test_reader_source_and_oracle(gpu):
    torch = gpu
    from oracle.gen_c import CMatcher
    from regengo_amd import Compiled
    from regengo_amd.dist import ReaderSource, ShardedReader
    tile = _tile()
    data = tile[:700_001]
    exp, cnt = CMatcher(URL).find_all_np(np.frombuffer(data, dtype=np.uint8))
    c = Compiled(URL).to(0)
    rd = ShardedReader(c, "cuda:0", window_bytes=100_000, halo_left=4096)
    got = []
    seen = []
    st = rd.find_reader(ReaderSource(io.BytesIO(data), "cuda:0", block=33_333), on_rows=lambda r, k, b: got.append(r.cpu()) and None,
                        on_match=lambda m: seen.append(m["StreamOffset"]) or True)
    rows = torch.cat(got).numpy()
    assert st["count"] == cnt and st["windows"] == 8
    assert np.array_equal(rows, exp.astype(np.int64))
    assert seen == exp[:, 0].tolist()


def test_find_reader_count_equals_callbacks(gpu):
    from regengo_amd import Compiled, Config
    tile = _tile()
    data = tile[:900_000]
    for pattern in (DATE, URL, r"(\d+)"):
        c = Compiled(pattern).to(0)
        if not c.info.ref_stream_offered:      # the C4 URL pattern: the reference memoises, its FindReader loop is not reproduced
            c = Compiled(pattern, stdlib=True).to(0)
        for bufsize, left in ((65536, 0), (100_000, 1024), (1 << 20, 0)):
            n = [0]

            def cb(_m):
                n[0] += 1
                return True

            c.FindReader(io.BytesIO(data), Config(bufsize, left), cb)
            assert c.FindReaderCount(io.BytesIO(data), Config(bufsize, left)) == n[0], (pattern, bufsize, left)
    assert Compiled(DATE).to(0).FindReaderCount(io.BytesIO(b""), Config(0, 0)) == 0


def _bench(args, timeout=900):
    env = dict(os.environ)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, cwd=ROOT, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1]
    return json.loads(line)


def test_bench_config_lines_at_reduced_size(gpu):
    r = _bench(["--config", "c3", "--strings", "300000", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"])
    assert r["config"]["parity_generator_truth"] and r["roofline"]["frac"] > 0 and r["config"]["workload"].startswith("C3")
    r = _bench(["--config", "c4", "--windows", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"])
    assert r["config"]["parity_oracle_fixture_periodic"] and r["config"]["matches_total"] == r["config"]["expected_matches"] > 10_000_000
    # (round 6) the line's value is the reference's FindReader over the chunk grid: fixture tiles with the grid's rule + the C port of the read loop
    rd = r["config"]["reader"]
    assert rd["parity_fixture_tiles_with_the_grid_rule"] is True and rd["parity_oracle_read_loop"]["identical"] is True and rd["parity_oracle_read_loop"]["mode"] == 1, rd
    assert rd["callbacks_total"] < r["config"]["matches_total"] and r["value_findall_semantics"]["value"] > 0 and r["roofline"]["frac"] > 0
    r = _bench(["--config", "c5", "--bytes", str(64 << 20), "--max-patterns", "60", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"])
    assert r["config"]["parity_counts_vs_oracle_fixture"], r["config"]
    assert r["config"]["patterns"] + r["config"]["patterns_skipped"] == 60
    r = _bench(["--bytes", str(1 << 26), "--steps", "3", "--warmup", "1", "--no-cpu-baseline"])
    assert r["config"]["parity_closed_form"] is True and r["repeats"] >= 1
    # (the PMC traffic is cited from profiles/ only for the kernel and size the run launched: a 64 MiB run cites a file or nothing, never a guess)
    assert r["roofline"]["traffic"] is None or (r["roofline"]["traffic_source"] or "").startswith("profiles/")
