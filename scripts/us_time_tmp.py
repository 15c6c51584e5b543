import sys, time, random
sys.path.insert(0, ".")
import numpy as np, torch
from oracle.gen_c import CMatcher
from regengo_amd import Compiled
src = open('tests/test_gpu_us.py').read().replace("pytestmark = pytest.mark.gpu", "")
ns = {}
exec(compile(src, 't', 'exec'), ns)
for pattern, kernel, alphabet in ns['CASES'][:3] + ns['CASES'][8:9]:
    c = Compiled(pattern).to(0); cm = CMatcher(pattern, q8=False)
    rng = random.Random(hash(pattern) & 0xFFFF)
    sizes = [64, 1000, 16385, 40000, 200000]
    for b in ns['_texts'](rng, alphabet, sizes) + [(alphabet[0] * 5000).encode(), (alphabet[-1] * 3000 + alphabet[0] * 2500).encode()]:
        arr = np.frombuffer(b, dtype=np.uint8).copy()
        t0 = time.time(); exp, cnt = cm.find_all_np(arr); t1 = time.time()
        spans, res = c.FindAllSpans(b); torch.cuda.synchronize(); t2 = time.time()
        own, _ = c.FindAllSpans(b, own=(len(b)//3, 2*len(b)//3+1)); torch.cuda.synchronize(); t3 = time.time()
        print("%-30s n=%6d oracle %.3fs gpu %.3fs owned %.3fs unsynced %d matches %d" % (pattern[:30], len(b), t1-t0, t2-t1, t3-t2, res.unsynced, cnt), flush=True)
