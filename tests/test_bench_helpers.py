"""bench.py's in-run parity helpers, checked on CPU against the oracle: the vectorised ground truth of the C3 e-mail batch must
be what the reference's FindBytes (oracle/gen_c.py: the generated-C port, restart rule included) returns per string."""
import importlib.util
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_email_truth_equals_the_oracle():
    from oracle.gen_c import CMatcher
    from regengo_amd import synth
    b = _bench()
    data, offsets = synth.email_batch_np(30000)
    found, spans = b.email_truth_np(data, offsets)
    cm = CMatcher(b.EMAIL)
    raw = data.tobytes()
    nf = 0
    for i in range(len(offsets) - 1):
        r = cm.find(raw[offsets[i]:offsets[i + 1]])
        assert (r is not None) == bool(found[i]), i
        if r is not None:
            assert r == spans[i].tolist(), (i, raw[offsets[i]:offsets[i + 1]])
            nf += 1
    assert nf > 15000
    # hand-made edge cases: '@' at the ends, adjacent strings, consecutive '@'
    strs = [b"@ab", b"ab@", b"a@b", b"a@@b", b"x a@b@c y", b"@", b"ab", b"a@b", b"_@9"]
    offs = np.zeros(len(strs) + 1, dtype=np.int64)
    offs[1:] = np.cumsum([len(s) for s in strs])
    d = np.frombuffer(b"".join(strs), dtype=np.uint8)
    f, sp = b.email_truth_np(d, offs)
    for i, s in enumerate(strs):
        r = cm.find(s)
        assert (r is not None) == bool(f[i]), s
        if r is not None:
            assert r == sp[i].tolist(), s


def test_config_fixtures_match_the_corpus_tile():
    import hashlib
    import json
    b = _bench()
    tile = b.corpus_tile()
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "c5_counts.json")))
    assert hashlib.sha256(tile).hexdigest() == fx["tile_sha256"] and len(tile) == fx["tile_len"]
    z = np.load(os.path.join(ROOT, "tests", "golden", "c4_url_rows.npz"))
    assert bytes(z["tile_sha256"]).hex() == fx["tile_sha256"]
    assert len(fx["patterns"]) == 255 and z["u"].shape[1] == 12
