"""GPU tier: BASELINE configs C4 (URL pattern, sharded stream windows with halos, stream-absolute offsets) and C5
(the corpus patterns over a shared corpus, ^/$-anchored ones per line) through the same runner that produces the
full-size numbers (scripts/gpu_c45.py -> profiles/r01_c4_c5_fullsize.json), at reduced size.  Parity is the closed form a
periodic corpus allows: the oracle's rows on a 3-tile buffer, extended periodically, must equal the GPU's rows."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _runner():
    spec = importlib.util.spec_from_file_location("gpu_c45", os.path.join(ROOT, "scripts", "gpu_c45.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_c4_stream_windows(built):
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU; there is no CPU fallback")
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    r = _runner().run_c4(2)          # two 1 GiB owned windows with halos, shard mode
    assert r["parity"] and r["matches"] == r["expected_matches"] and r["matches"] > 10_000_000


def test_c5_pattern_suite(built):
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU; there is no CPU fallback")
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    cwd = os.getcwd()
    os.chdir(ROOT)
    try:
        r = _runner().run_c5(0, 0, mib=32)
    finally:
        os.chdir(cwd)
    assert r["bad"] == [] and r["ok"] >= 248 and r["unsupported"] <= 5 and r["oracle_timeout"] == 0
