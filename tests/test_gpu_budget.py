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
