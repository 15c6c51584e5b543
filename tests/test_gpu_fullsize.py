"""GPU tier: the BASELINE configurations at their FULL size, through bench.py's own code path (the line the driver records is the
line checked here): C3 -- 10 M strings, reference semantics, generator truth on 1 M + a 20 k sample of the whole batch against the
oracle's C port; C3 under Options.ForceTDFA ("Email TDFA with capture tags": the reference's Tagged DFA on the device) against the
C port of the emitted TDFA; C5 -- 255 patterns over 1 GiB in reference mode wherever offered, every span table and every per-line
result checksummed on the device against the oracle's rows on three tiles extended periodically (rows, not counts).  C2 and C4 at
full size: tests/test_gpu_parity.py (closed form, every row) and tests/test_gpu_configs.py."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*args, timeout=900):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(args), capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    return json.loads(line)


def test_c3_full_size_reference_mode(built):
    j = _bench("--config", "c3", "--steps", "2", "--warmup", "1", "--no-cpu-baseline")
    c = j["config"]
    assert c["strings_per_gpu"] == 10_000_000
    assert c["parity_generator_truth"] is True and c["parity_strings_checked"] == 1_000_000
    assert c["parity_rows_vs_oracle_sample"] is True and c["oracle_sample_strings"] >= 10_000
    assert j["dtype"] == "u8" and j["value"] > 0


def test_c3_full_size_forced_tdfa(built):
    j = _bench("--config", "c3", "--force-tdfa", "--steps", "2", "--warmup", "1", "--no-cpu-baseline")
    c = j["config"]
    assert c["strings_per_gpu"] == 10_000_000 and "Tagged DFA" in c["engine"]
    assert c["parity_generator_truth"] is True and c["parity_rows_vs_oracle_sample"] is True


def test_c5_full_size_rows(built):
    j = _bench("--config", "c5", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", timeout=1500)
    c = j["config"]
    assert c["corpus_bytes"] > 1 << 29 and c["patterns"] == 255 and c["patterns_skipped"] == 0
    assert c["parity_counts_vs_oracle_fixture"] is True
    assert c["parity_rows_vs_oracle_fixture"] is True and c["patterns_with_wrong_rows"] == 0
    assert c["patterns_row_checked"] >= 250, c["patterns_row_checked"]
