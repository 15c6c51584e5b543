import sys, random
sys.path.insert(0, ".")
import numpy as np, torch
from oracle.gen_c import CMatcher
from regengo_amd import Compiled
src = open('tests/test_gpu_us.py').read().replace("pytestmark = pytest.mark.gpu", "")
ns = {}
exec(compile(src, 't', 'exec'), ns)
pattern, kernel, alphabet = [c for c in ns['CASES'] if c[0].startswith("x[a-z]")][0]
c = Compiled(pattern).to(0); cm = CMatcher(pattern, q8=False)
rng = random.Random(hash(pattern) & 0xFFFF)
sizes = [64, 65, 127, 128, 129, 1000, 16383, 16384, 16385, 16384 + 255, 16384 + 257, 32768, 40000, 70001, 200000]
for b in ns['_texts'](rng, alphabet, sizes) + [b"", b"a", (alphabet[0] * 5000).encode(), (alphabet[-1] * 3000 + alphabet[0] * 2500).encode()]:
    arr = np.frombuffer(b, dtype=np.uint8).copy() if b else np.zeros(0, dtype=np.uint8)
    exp, cnt = cm.find_all_np(arr)
    spans, res = c.FindAllSpans(b)
    got = spans.cpu().numpy()
    ok = res.total == cnt and got.shape == exp.shape and np.array_equal(got, exp)
    if ok:
        print("n", len(b), "ok", cnt, "unsynced", res.unsynced); continue
    k = 0
    while k < min(len(got), len(exp)) and (got[k] == exp[k]).all(): k += 1
    print("n", len(b), "MISMATCH gpu", int(res.total), "oracle", cnt, "unsynced", res.unsynced, "first diff at match", k,
          "exp", exp[k].tolist() if k < len(exp) else None, "got", got[k].tolist() if k < len(got) else None)
    s0 = int(exp[k][0]) if k < len(exp) else 0
    print("   text around:", b[max(0, s0 - 30):s0 + 50], " prev match", exp[k-1].tolist() if k else None)
    import collections
    print("   run-length histogram of the text (top):", collections.Counter(len(x) for x in b.split(b" ")).most_common(5))
