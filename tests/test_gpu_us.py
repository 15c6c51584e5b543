"""GPU tier: the one-step-per-byte scan kernels (rgx_scan_us.hip) against the oracle's C restatement of the reference's
FindAllBytes (find.go:130-466; oracle/gen_c.py) -- each of the three variants (start registers / register-free / register-free
with two bytes per look-up), on inputs built to hit what is special about them: stretches that cross tile borders, matches that
end where a stretch ends, rewinds (a match followed by bytes that kept older threads alive), the end of the text inside a
look-up pair, texts without a single sync point, shard ownership."""
import random
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

URL = r"(?P<full>(?P<proto>https?|ftp)://(?P<host>[\w.-]+)(?P<port>:\d+)?(?P<path>/[\w./-]*)?)"
CASES = [
    # (pattern, expected rgx_info.scan_kernel, alphabet)
    (r"(?P<user>\w+)@(?P<domain>\w+)", 6, "ab_9@ .\n"),
    (r"(\d+)", 6, "0123 ab-\n"),
    (r"\b[a-z]+\b", 6, "abz_ 09.\n"),
    (URL, 6, "htps:/f.w-1 \n"),
    (r"\[(INFO|WARN)\]", 6, "[]INFOWAR x\n"),
    (r"ab+c|a", 6, "abc x"),                      # rewinds: "abbbx" ends the match [0,1) only when x arrives
    (r"(?m)^foo\d+$", 6, "fo0\n9x"),
    (r"x[a-z]*y|x", 6, "xay b\n"),                # long overshoot before the rewind
    (r"start.*middle.*end", 6, "startmidle nx\n"),
    (r"hello.*world", 6, "helowrd x\n"),
    (r"[\w.+-]+@[\w.-]+\.[a-z]{2,}", 4, "ab.@-+ z\n"),          # two start registers
    (r"(?P<major>\d+)\.(?P<minor>\d+)\.(?P<patch>\d+)", 4, "0123. a\n"),   # three
    (r"(?P<words>(?P<word>\w+\s+){5})(?P<end>\w+)", 4, "ab_ 9\n\t."),         # five: the eight-register instance (round 3)
    (r"(?:[a-c]+-){6}[a-c]+", 4, "abc- x\n"),                                  # seven
    (r"[a-q]+[0-9]|[c-z]{3}!", None, "acz09! "),
    (r"[^\s\"]+\"", None, "ab\" \n\t9"),
]


@pytest.fixture(scope="module")
def torch_dev(built):
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU; there is no CPU fallback")
    return torch


def _texts(rng, alphabet, sizes):
    out = []
    for n in sizes:
        k = rng.random()
        if k < 0.4:
            b = "".join(rng.choice(alphabet) for _ in range(n))
        elif k < 0.7:
            # long runs: matches and stretches much longer than a 64-byte slice, some longer than a tile's look-ahead
            parts = []
            size = 0
            while size < n:
                run = rng.choice(alphabet) * rng.choice([1, 3, 70, 300, 1500, 3000])
                parts.append(run)
                size += len(run)
            b = "".join(parts)[:n]
        else:
            words = ["".join(rng.choice(alphabet) for _ in range(rng.randrange(1, 9))) for _ in range(40)]
            b = "".join(rng.choice(words) for _ in range(n // 4 + 1))[:n]
        out.append(b.encode())
    return out


@pytest.mark.parametrize("pattern,kernel,alphabet", CASES)
def test_us_kernels_equal_oracle(torch_dev, pattern, kernel, alphabet):
    from oracle.gen_c import CMatcher
    from regengo_amd import Compiled, _capi
    c = Compiled(pattern, no_prefilter_scan=True).to(0)      # (the kernels under test, not rgx_scan_fc.hip: tests/test_gpu_fc.py)
    if kernel is not None:
        assert c.info.scan_kernel == kernel, (pattern, c.info.scan_kernel)
    cm = CMatcher(pattern, q8=False)
    rng = random.Random(zlib.crc32(pattern.encode()) & 0xFFFF)      # (hash() of a str changes from process to process)
    sizes = [64, 65, 127, 128, 129, 1000, 16383, 16384, 16385, 16384 + 255, 16384 + 257, 32768, 40000, 70001, 200000]
    # (runs without a sync point are kept to a few KiB: the carry pass that resolves them restates the reference's loop, which
    # is quadratic in the length of such a run -- as the reference itself is)
    for b in _texts(rng, alphabet, sizes) + [b"", b"a", (alphabet[0] * 5000).encode(), (alphabet[-1] * 3000 + alphabet[0] * 2500).encode()]:
        arr = np.frombuffer(b, dtype=np.uint8).copy() if b else np.zeros(0, dtype=np.uint8)
        exp, cnt = cm.find_all_np(arr)
        try:
            spans, res = c.FindAllSpans(b)
        except _capi.RgxError as ex:
            # the rewinding patterns are quadratic on long runs (the reference's loop too): past the walkers' step budget the call
            # is refused, never answered wrongly (tests/test_gpu_budget.py)
            assert ex.status == _capi.RGX_E_UNSUPPORTED and "quadratic" in str(ex) and "|" in pattern, (pattern, len(b), str(ex))
            continue
        got = spans.cpu().numpy()
        assert res.total == cnt and got.shape == exp.shape and np.array_equal(got, exp), (pattern, len(b), cnt, int(res.total))
        n, _ = c.CountAll(torch_dev.from_numpy(arr).cuda()) if len(b) else (0, None)
        assert n == cnt, (pattern, len(b))
        # shard ownership: only matches that START inside [lo, hi)
        if len(b) >= 1000:
            lo, hi = len(b) // 3, 2 * len(b) // 3 + 1
            own, _ = c.FindAllSpans(b, own=(lo, hi))
            keep = exp[(exp[:, 0] >= lo) & (exp[:, 0] < hi)]
            assert np.array_equal(own.cpu().numpy(), keep), (pattern, len(b), "owned range")


def test_us_kernel_matches_at_tile_and_stretch_borders(torch_dev):
    """One-byte and maximal matches placed at every offset around a tile border and around the end of the text."""
    from oracle.gen_c import CMatcher
    from regengo_amd import Compiled
    for pattern, unit in [(r"(\d+)", b"7"), (r"\w+@\w+", b"a@b"), (r"ab+c|a", b"abbb")]:
        c = Compiled(pattern, no_prefilter_scan=True).to(0)
        cm = CMatcher(pattern, q8=False)
        for border in (16384, 32768):
            for shift in range(-6, 7):
                for tail in (0, 1, 2, 5):
                    buf = bytearray(b" " * (border + 300))
                    at = border + shift
                    buf[at:at + len(unit)] = unit
                    b = bytes(buf[:at + len(unit) + tail])
                    arr = np.frombuffer(b, dtype=np.uint8).copy()
                    exp, cnt = cm.find_all_np(arr)
                    spans, res = c.FindAllSpans(b)
                    assert res.total == cnt and np.array_equal(spans.cpu().numpy(), exp), (pattern, border, shift, tail)


def test_us_kernel_without_sync_points_takes_the_carry_path(torch_dev):
    """Runs of several KiB without a reset byte (one enormous word): slices report themselves unsynced, the host resolves them
    (sync automaton / carry pass) and the result is still the oracle's.  (Kept to a few KiB per run: the carry pass restates the
    reference's loop, quadratic in the length of such a run.)"""
    from oracle.gen_c import CMatcher
    from regengo_amd import Compiled
    pattern = r"(?P<user>\w+)@(?P<domain>\w+)"
    c = Compiled(pattern).to(0)
    cm = CMatcher(pattern, q8=False)
    rng = random.Random(5)
    word = lambda n: "".join(rng.choice("abcdefgh") for _ in range(n))
    texts = [("x " + word(6000) + "@" + word(3000) + " y@z " + word(9000) + " q").encode(),
             (" ".join(word(rng.choice([5, 2500, 4000])) + rng.choice(["", "@b"]) for _ in range(40))).encode()]
    seen_unsynced = 0
    for b in texts:
        arr = np.frombuffer(b, dtype=np.uint8).copy()
        exp, cnt = cm.find_all_np(arr)
        spans, res = c.FindAllSpans(b)
        seen_unsynced += int(res.unsynced)
        assert res.total == cnt and np.array_equal(spans.cpu().numpy(), exp), (len(b), cnt, int(res.total))
    assert seen_unsynced > 0


def test_patterns_without_reset_bytes(torch_dev):
    """Patterns whose every byte keeps some thread alive (a tail that survives the newline, a negated class, (?s).*) have no
    reset-byte sync points: they are eligible for the start-tracking automaton but must take the sync automaton W (generic
    kernel).  Rows against the oracle's generated-C matcher on the web-log text, on text with very long lines, and on text
    without any sync point for kilobytes."""
    import numpy as np
    from oracle.gen_c import CMatcher
    from regengo_amd import Compiled, synth
    tile = synth.web_log_tile(1 << 19)
    long_lines = (b"x" * 700 + b" bob@example.com rest of the line " + b"y" * 900 + b"\n") * 150
    texts = [tile, long_lines, tile[:70_000] + b"a@b " + b"q" * 9000 + b"\n" + tile[:50_000]]
    pats = [r"(?P<full>(?P<name>[\w.+-]+)@(?P<host>[\w.-]+))(?P<extra>\s.*)?", r"(?P<k>[a-z]+)=(?P<v>[^;]*)", r"(?s)GET.*?\d",
            r"(?P<w>\w+)(?P<rest>\s[^\n]*)?", r"\[(?P<level>\w+)\]\s+(?P<msg>.*)"]
    for pat in pats:
        c = Compiled(pat, stdlib=True).to(0)
        cm = CMatcher(pat)
        want = [cm.find_all_np(np.frombuffer(t, dtype=np.uint8)) for t in texts]
        # twice: a program remembers how its unsynced slices were best resolved (exact sync points first, or the carry pass), so
        # the second round takes every text through the remembered strategy
        for rnd in range(2):
            for t, (exp, cnt) in zip(texts, want):
                spans, res = c.FindAllSpans(t)
                assert res.total == cnt, (pat, rnd, res.total, cnt)
                assert np.array_equal(spans.cpu().numpy(), exp), (pat, rnd)
                n, _r = c.CountAll(t)
                assert n == cnt


def test_rescan_after_the_carry_pass_survives_a_look_back_timeout(torch_dev):
    """Round 3 regression (tests/golden/regress/us_long_runs_40000.bin, `x[a-z]*y|x`): two 6004-byte runs without a sync point keep the
    workgroups of the first two tiles in their single-step walkers for so long that the third tile's look-back spin hits its bound.
    The first scan of a call repeated itself with tickets in that case; the rescan after the carry pass did not look at the flag,
    and the third tile's rows landed at offset 0 (count right, table wrong)."""
    import os
    from oracle.gen_c import CMatcher
    from regengo_amd import Compiled
    b = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "regress", "us_long_runs_40000.bin"), "rb").read()
    pattern = r"x[a-z]*y|x"
    c = Compiled(pattern).to(0)
    exp, cnt = CMatcher(pattern, q8=False).find_all_np(np.frombuffer(b, dtype=np.uint8).copy())
    for _ in range(3):
        spans, res = c.FindAllSpans(b)
        assert res.total == cnt == 4814
        assert np.array_equal(spans.cpu().numpy(), exp)


def test_rows_of_dense_matches_go_through_lds(built):
    """A wave with 128 matches or more writes its rows through LDS (rgx_scan_us.hip: UsEmitTile, rgx_kernels.hip: phase 3): (start, end)
    at the match's rank, a pass of 256 at a time, then lane j record j.  Texts where a match ends every 2-5 bytes (several passes per
    wave, tiles whose waves differ: dense next to sparse next to empty), the pair / generic / register kernels, fixed templates and
    (start, end) records, n > 0, a capacity one row short, owned ranges that cut the dense stretch -- rows == the C port of the emitted
    matcher."""
    import numpy as np
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU; there is no CPU fallback")
    from oracle.gen_c import CMatcher
    from regengo_amd import Compiled, _capi, synth
    rnd = random.Random(0xD3A5E)
    words = [b"a", b"bc", b"def", b"x1", b"42", b"7", b"hello", b"Zq"]
    dense = b" ".join(rnd.choice(words) for _ in range(60000))                      # a match every ~3 bytes
    sparse = (b"." * 5000 + b" tail ") * 8
    mixed = dense[:70000] + sparse + dense[70000:150000] + b"." * 40000 + dense[150000:]
    log = synth.web_log_tile()[:180000]
    kinds = set()
    checked = 0
    for pat in (r"\b\w+\b", r"(?P<w>\w+)", r"[a-z]+", r"\d+", r"[\p{L}\p{N}]+", r"\S+", r"(?P<a>[a-z])(?P<b>\w*)", r"\w+\s*"):
        cm = CMatcher(pat, q8=False)
        c = Compiled(pat, stdlib=True).to(0)
        kinds.add(c.info.scan_kernel)
        for text in (mixed, log, dense[:4097], dense[:300]):
            arr = np.frombuffer(text, dtype=np.uint8).copy()
            exp, cnt = cm.find_all_np(arr)
            buf = torch.from_numpy(arr).cuda()
            spans, res = c.FindAllSpans(buf)
            assert res.total == cnt and np.array_equal(spans.cpu().numpy(), exp), (pat, len(text), cnt, int(res.total))
            assert int(c.CountAll(buf)[0]) == cnt, (pat, len(text))
            checked += 1
            if cnt > 1000:
                k = cnt // 3
                assert np.array_equal(c.FindAllSpans(buf, n=k)[0].cpu().numpy(), exp[:k]), (pat, "n")
                with pytest.raises(_capi.RgxError) as ei:
                    c.FindAllSpans(buf, capacity=cnt - 1)
                assert ei.value.status == _capi.RGX_E_CAPACITY, pat
                n = len(text)
                for lo, hi in ((1, n - 1), (16383, 16385), (n // 3 + 7, 2 * n // 3 + 1), (70001, 70001 + 5100)):
                    if lo >= n:
                        continue
                    got, r = c.FindAllSpans(buf, own=(lo, min(hi, n)))
                    keep = (exp[:, 0] >= lo) & (exp[:, 0] < min(hi, n))
                    assert r.total == int(keep.sum()) and np.array_equal(got.cpu().numpy(), exp[keep]), (pat, lo, hi)
    assert checked == 32 and len(kinds) >= 2, (checked, kinds)
