"""Texts that keep every attempt of a pattern running are quadratic work for the reference's loop and for the generic kernel's
attempt-per-start lanes alike.  The device must not be held for minutes by them: the call either answers (the one-step-per-byte
kernels are linear whatever the text) or is refused with RGX_E_UNSUPPORTED inside a bounded time (rgx_kernels.hip: kLaneStepBudget,
kCarryBudget)."""
import time

import pytest

pytestmark = pytest.mark.gpu

CASES = [
    (r"\pL+9", b"c", 1 << 16),             # generic kernel, tables in global memory: the serial carry pass would take ~45 min
    (r"[^q]{1,200}z", b"c", 1 << 16),      # generic kernel: every attempt runs 200 bytes
    (r"[^q]+z", b"c", 1 << 20),            # a one-step-per-byte kernel: linear
    (r"\pL+9", b"c", 1 << 24),             # the scan itself: 64 starts x 16 MiB per lane without the lane budget
]


@pytest.mark.parametrize("pat,unit,n", CASES)
def test_a_text_that_keeps_attempts_running_is_answered_or_refused_in_bounded_time(pat, unit, n):
    import torch
    from regengo_amd import Compiled, _capi
    c = Compiled(pat, stdlib=True).to(0)
    data = (unit * (n // len(unit) + 1))[:n]
    t0 = time.perf_counter()
    try:
        spans, res = c.FindAllSpans(data)
        torch.cuda.synchronize()
        assert res.total == 0                  # none of these texts holds the closing byte
    except _capi.RgxError as ex:
        assert ex.status == _capi.RGX_E_UNSUPPORTED
        assert "quadratic" in str(ex)
    assert time.perf_counter() - t0 < 30.0
    # the context is usable afterwards
    spans, res = c.FindAllSpans(b"abc9 " if "9" in pat else b"ccz ")
    assert res.total == 1
