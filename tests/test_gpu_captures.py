"""GPU tier: the capture pass (caps_lds_kernel, rgx_kernels.hip) against the oracle's C restatement of the reference's capture
loop (find.go:130-466, captures.go:123-158; oracle/gen_c.py) on what is special about its fast forms: matches of every length
around the 96-byte per-lane row (those that fit take the in-row walk, the others the general one with a trace in global memory),
every alignment of the match inside its first 16-byte chunk, a match at offset 0, a match that ends with the text, trailing
assertions (the byte after the match decides), automata whose cells fit a byte and automata whose cells do not."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_dev(built):
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU; there is no CPU fallback")
    return torch


CASES = [
    # (pattern, builder of one record from (length, filler byte))
    (r"(?P<k>\w+)=(?P<v>[^;\n]*)(?P<semi>;)?", lambda n, f: b"key=" + f * n + b";"),
    (r"(?P<k>[a-z]+)(?P<d>\d*)\b(?P<t>-x)?", lambda n, f: b"ab" + b"7" * n + b"-x"),              # trailing \b: lookahead form
    (r"(?P<a>[ab]{1,40})(?P<c>c)?(?P<d>d+)?", lambda n, f: (b"ab" * 20)[:max(1, min(n, 40))] + b"c" * (n % 2) + b"d" * (n // 3)),
    (r"(?P<u>[\w.+-]+)@(?P<h>[\w-]+(?:\.[\w-]+)*)(?P<p>:\d+)?", lambda n, f: b"u" * (1 + n // 2) + b"@h" + b".example" * (n // 16) + b":80"),
    (r"(?P<w>\w+)\s(?P<rest>.*)", lambda n, f: b"w " + f * n),
]


@pytest.mark.parametrize("pattern,make", CASES)
def test_capture_rows_of_every_length_and_alignment(torch_dev, pattern, make):
    from oracle.gen_c import CMatcher
    from regengo_amd import Compiled
    c = Compiled(pattern).to(0)
    if c.info.ref_findall_offered != 1:         # Tagged-DFA class: refused in reference mode (tests/test_gpu_tdfa.py); the kernels are
        c = Compiled(pattern, stdlib=True).to(0)     # what is under test here, against the leftmost-first oracle
    cm = CMatcher(pattern, q8=False)
    parts = []
    for n in list(range(0, 112)) + [127, 128, 129, 200, 1000]:
        for align in (0, 1, 3, 4, 7, 8, 15):
            parts.append(b" " * align + make(n, b"v") + b"\n")
    body = b"".join(parts)
    for text in (body, body[1:], make(5, b"v") + b"\n" + body, body + make(90, b"z"), body + make(60, b"z")):
        arr = np.frombuffer(text, dtype=np.uint8).copy()
        exp, cnt = cm.find_all_np(arr)
        assert cnt >= 64                                   # (fewer matches take the small-count kernel)
        spans, res = c.FindAllSpans(text)
        got = spans.cpu().numpy()
        assert res.total == cnt and got.shape == exp.shape
        bad = np.nonzero((got != exp).any(axis=1))[0]
        assert bad.size == 0, (pattern, len(text), int(bad[0]), got[bad[0]].tolist(), exp[bad[0]].tolist())


def test_capture_rows_with_unmatched_minus_one(torch_dev):
    from oracle.gen_c import CMatcher
    from regengo_amd import Compiled, _capi
    pattern = r"(?P<k>\w+)=(?P<v>\d+)?(?P<semi>;)?"
    c = Compiled(pattern, flags=_capi.FLAG_UNMATCHED_MINUS1).to(0)
    if c.info.ref_findall_offered != 1:
        c = Compiled(pattern, flags=_capi.FLAG_UNMATCHED_MINUS1, stdlib=True).to(0)
    text = b"".join(b" " * (i % 9) + b"k" * (1 + i % 70) + b"=" + (b"12" * (i % 5)) + (b";" if i % 3 else b"") + b"\n" for i in range(400))
    spans, res = c.FindAllSpans(text)
    got = spans.cpu().numpy()
    exp, cnt = CMatcher(pattern, q8=False).find_all_np(np.frombuffer(text, dtype=np.uint8).copy())
    assert res.total == cnt
    # the oracle reports unmatched groups the reference's way, (0, 0); the flag turns exactly those into (-1, -1)
    minus = got == -1
    assert minus.any() and np.array_equal(minus[:, 0::2], minus[:, 1::2])
    assert not exp[minus].any()
    assert np.array_equal(np.where(minus, 0, got), exp)


def test_pairs_table_and_record_slots_agree(torch_dev):
    """The scan hands (start, end) of every match to the capture pass in a table of its own (ScanParams::pairs) or -- RGX_NO_PAIRS=1,
    read once per process: a child process here -- in slots 0-1 of the records: the same rows either way, for a program of the pair
    kernel, one of the register kernels and one of the generic kernel, with a span table that is exactly as large as the matches."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = r'''
import hashlib, json, sys
sys.path.insert(0, %r)
import torch
from regengo_amd import Compiled, synth
tile = synth.web_log_tile()
buf = torch.frombuffer(bytearray(tile * 3), dtype=torch.uint8).to("cuda:0")
out = {}
for p in (r"(?P<full>(?P<proto>https?|ftp)://(?P<host>[\w.-]+)(?P<port>:\d+)?(?P<path>/[\w./-]*)?)",
          r"(?P<u>[\w.+-]+)@(?P<h>[\w-]+(?:\.[\w-]+)*)(?P<p>:\d+)?", r"(?P<w>\w+)\s(?P<rest>.*)"):
    c = Compiled(p, stdlib=True).to(0)
    n = int(c.CountAll(buf)[0])
    rows = c.FindAllSpans(buf, capacity=n)[0]
    assert rows.shape[0] == n
    out[p] = [n, hashlib.sha256(rows.cpu().numpy().tobytes()).hexdigest()]
print(json.dumps(out))
''' % root
    res = []
    for no_pairs in ("", "1"):
        env = dict(os.environ)
        env.pop("RGX_NO_PAIRS", None)
        if no_pairs:
            env["RGX_NO_PAIRS"] = "1"
        r = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        res.append(json.loads(r.stdout.strip().splitlines()[-1]))
    assert res[0] == res[1] and all(v[0] > 1000 for v in res[0].values()), res
