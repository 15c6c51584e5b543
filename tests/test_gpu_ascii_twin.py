"""The ASCII twin on the device (rgx_capi.cc: AsciiTwin): a program that misses the one-step-per-byte kernels scans a text without a
byte >= 0x80 through the twin built for such texts, and any other text as before -- the rows are the oracle's either way."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("pat", [r"\p{L}+", r"[\p{L}\p{N}]+", r"\p{Greek}+", r"(?P<w>\p{L}+)-(?P<n>\p{N}+)"])
def test_rows_with_and_without_a_high_byte(pat):
    from oracle.gen_c import CMatcher
    from regengo_amd import Compiled, synth
    c = Compiled(pat).to(0)
    if c.info.ref_findall_offered != 1:
        c = Compiled(pat, stdlib=True).to(0)
    cm = CMatcher(pat, q8=False)
    tile = synth.web_log_tile(1 << 20)
    tile = bytes(b if b < 0x80 else 0x2E for b in tile) + b" word-42 x"
    texts = [tile * 2,                                              # no high byte: the twin (>= 1 MiB)
             tile + "café αβγ-7 ".encode() + tile,      # high bytes in the middle: the full program
             tile[:300_000]]                                        # below the size at which the question is asked
    for t in texts:
        exp, cnt = cm.find_all_np(np.frombuffer(t, dtype=np.uint8).copy())
        spans, res = c.FindAllSpans(t)
        assert res.total == cnt
        assert np.array_equal(spans.cpu().numpy(), exp)
    # and again in the other order (the twin is made once and remembered)
    spans, res = c.FindAllSpans(texts[0])
    exp, cnt = cm.find_all_np(np.frombuffer(texts[0], dtype=np.uint8).copy())
    assert res.total == cnt and np.array_equal(spans.cpu().numpy(), exp)
