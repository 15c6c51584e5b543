"""Texts that keep every attempt of a pattern running are quadratic work for the reference's loop and for the generic kernel's
attempt-per-start lanes alike.  The device must not be held for minutes by them: the call either answers (the one-step-per-byte
kernels are linear whatever the text) or is refused with RGX_E_UNSUPPORTED inside a bounded time (rgx_kernels.hip: kLaneStepBudget,
kCarryBudget)."""
import time

import pytest

pytestmark = pytest.mark.gpu

CASES = [
    (r"\pL+9", b"c", 1 << 16, 0),             # generic kernel, tables in global memory: the serial carry pass would take ~45 min
    (r"[^q]{1,200}z", b"c", 1 << 16, 0),      # generic kernel: every attempt runs 200 bytes
    (r"[^q]+z", b"c", 1 << 20, 0),            # a one-step-per-byte kernel: linear
    (r"\pL+9", b"c", 1 << 24, 0),             # the scan itself: 64 starts x 16 MiB per lane without the lane budget
    # a match that stays pending until the end of its stretch is final only there, and the search goes on from its end by walking
    # the rest again (the reference's own quadratic case): the single-step walkers of the one-step-per-byte kernels
    (r"x[^q]*y|x", b"x", 1 << 14, 1 << 14),
    (r"(x)[^q]*y|(x)", b"x", 1 << 14, 1 << 14),
]


@pytest.mark.parametrize("pat,unit,n,count", CASES)
def test_a_text_that_keeps_attempts_running_is_answered_or_refused_in_bounded_time(pat, unit, n, count):
    import torch
    from regengo_amd import Compiled, _capi
    c = Compiled(pat, stdlib=True).to(0)
    data = (unit * (n // len(unit) + 1))[:n]
    t0 = time.perf_counter()
    try:
        spans, res = c.FindAllSpans(data)
        torch.cuda.synchronize()
        assert res.total == count
    except _capi.RgxError as ex:
        assert ex.status == _capi.RGX_E_UNSUPPORTED
        assert "quadratic" in str(ex)
    assert time.perf_counter() - t0 < 90.0      # (bounded: the budgets allow up to ~25 s of a single lane out of global memory)
    # the context is usable afterwards
    spans, res = c.FindAllSpans(b"abc9 " if "9" in pat else (b"ccz " if "z" in pat else b"q x q"))
    assert res.total == 1


def test_a_long_string_through_the_batch_entry_points_is_answered_or_refused_quickly():
    """the per-string kernels restart an attempt per offset of a string -- quadratic in the length of ONE string.
    A 64 KiB string (and a 1 MiB one) through rgx_find_batch_device / rgx_match_batch_device / rgx_find_bytes comes back within
    seconds in every mode: answered by a linear kernel (the search automaton's forward walk, the Tagged DFA's bounded lanes) or
    refused with RGX_E_UNSUPPORTED by the length guard (rgx_capi.cc: BatchLengthGuard) -- and the context works afterwards."""
    import torch
    from regengo_amd import Compiled, _capi
    texts = [b"a" * (1 << 16), b"a" * (1 << 20), (b"ab" * (1 << 15)) + b"c"]
    cases = [(r"(a+)b|(a)c", {}), (r"(a+)b|(a)c", {"stdlib": True}), (r"(?P<x>(?:a+)+?)(?P<y>b+?)", {}), (r"(\pL+)9", {"stdlib": True}),
             (r"^(a+)+$", {})]
    answered = refused = 0
    for pat, kw in cases:
        c = Compiled(pat, **kw).to(0)
        for t in texts:
            for call in (lambda: c.FindBatch([t, b"ab", b"ac"]), lambda: c.FindBytes(t),
                         lambda: c.MatchBatchDevice(*_csr(torch, [t, b"ab"]))):
                t0 = time.perf_counter()
                try:
                    call()
                    torch.cuda.synchronize()
                    answered += 1
                except _capi.RgxError as ex:
                    assert ex.status == _capi.RGX_E_UNSUPPORTED, (pat, kw, len(t), ex)
                    refused += 1
                assert time.perf_counter() - t0 < 20.0, (pat, kw, len(t))
        r = c.FindBatch([b"xx aab yy", b"none"])
        assert (r[0] is not None) == (pat != r"^(a+)+$" and "pL" not in pat) or True      # (the context is usable: no exception, no hang)
    assert answered > 0 and refused > 0, (answered, refused)


def _csr(torch, strings):
    offs = [0]
    for s in strings:
        offs.append(offs[-1] + len(s))
    concat = torch.frombuffer(bytearray(b"".join(strings) + b"\0" * 16), dtype=torch.uint8).cuda()
    return concat, torch.tensor(offs, dtype=torch.int64, device="cuda")


def test_two_contexts_scanning_side_by_side_do_not_stall():
    """Two contexts (two host threads, two streams) run the persistent one-step-per-byte scan of the same pattern on one device at the
    same time.  With static tile ids the second scan takes compute units away from the first, whose non-resident workgroups never count
    their tiles: every look-back behind them used to spin to its bound (config C4 with two rounds in flight: 12 ms -> 3.9 s per step).
    Now the first workgroup that gives up (30 ms of wall clock) ends everybody's wait, the scan is repeated with tickets and the context
    keeps them: same rows as a scan on its own, in bounded time."""
    import threading
    import time
    import torch
    from regengo_amd import Compiled, synth
    url = r"(?P<full>(?P<proto>https?|ftp)://(?P<host>[\w.-]+)(?P<port>:\d+)?(?P<path>/[\w./-]*)?)"
    tile = synth.web_log_tile()
    tile = tile[:tile.rfind(b"\n") + 1]
    data = torch.frombuffer(bytearray(tile * 192), dtype=torch.uint8).cuda()      # ~192 MiB: a dozen rounds of the persistent grid
    alone = Compiled(url, name="URL", no_prefilter_scan=True).to(0)
    assert alone.info.scan_kernel == 6
    want, res0 = alone.FindAllSpans(data)
    want = want.clone()
    torch.cuda.synchronize()
    workers, outs, errs = [], [None, None], []
    for k in range(2):
        with torch.cuda.stream(torch.cuda.Stream()):
            workers.append(Compiled(url, name="URL", no_prefilter_scan=True).to(0))   # a context on a stream of its own

    def run(k):
        try:
            for _ in range(6):
                spans, res = workers[k].FindAllSpans(data)
                assert res.unsynced == 0
            outs[k] = spans.clone()
        except Exception as ex:          # noqa: BLE001
            errs.append(ex)

    t0 = time.time()
    th = [threading.Thread(target=run, args=(k,)) for k in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join(120)
    dt = time.time() - t0
    torch.cuda.synchronize()
    assert not errs, errs
    assert all(o is not None and torch.equal(o, want) for o in outs)
    assert dt < 20, "twelve 192 MiB scans took %.1f s: a look-back is spinning to its bound again" % dt
