"""One-off wide sweep of the differential test (tests/test_gpu_fuzz.py runs three seeds): many seeds, ASCII and UTF-8
pattern generators, FindAllBytes on the GPU against the C oracle (Q8 off).  usage: gpu_fuzz_sweep.py [first_seed] [nseeds]
The oracle is the reference's backtracker and some random patterns are catastrophic for it (seed 420 held a 25-minute call, seed 1133
one of 101 s; the GPU side of the same pattern takes microseconds).  Its 70 KB calls therefore run in a CHILD process with a time
limit of their own (no device context in the child): the process that holds the device context is never killed from outside while
kernels are in flight -- three GPU boxes were lost to sweeps whose per-seed `timeout` fired behind such an oracle call."""
import os, random, subprocess, sys, time
sys.path.insert(0, ".")
import numpy as np
from oracle import engines as E
from oracle.gen_c import CMatcher
from regengo_amd import Compiled, _capi
from tests import _fuzzgen as F

_CHILD = r"""
import sys
sys.path.insert(0, ".")
import numpy as np
from oracle.gen_c import CMatcher
cm = CMatcher(sys.argv[1], q8=False)
b = sys.stdin.buffer.read()
arr = np.frombuffer(b, dtype=np.uint8).copy() if len(b) else np.zeros(0, dtype=np.uint8)
exp, cnt = cm.find_all_np(arr)
sys.stdout.buffer.write(np.int64(cnt).tobytes() + np.ascontiguousarray(exp, dtype=np.int32).tobytes())
"""


def oracle_bounded(p, b, ncap, limit_s=20.0):
    """(rows, count) from the C oracle in a child process, or None when it does not finish in limit_s (catastrophic backtracking)."""
    try:
        r = subprocess.run([sys.executable, "-c", _CHILD, p], input=b, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=limit_s)
    except subprocess.TimeoutExpired:
        return None
    if r.returncode != 0 or len(r.stdout) < 8:
        return None
    cnt = int(np.frombuffer(r.stdout[:8], dtype=np.int64)[0])
    return np.frombuffer(r.stdout[8:], dtype=np.int32).reshape(-1, ncap), cnt


first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
nseeds = int(sys.argv[2]) if len(sys.argv) > 2 else 20
t0 = time.time()
tot = bad = refused = 0
for seed in range(first, first + nseeds):
    rng = random.Random(seed)
    pats = [(p, False) for p in F.gen_patterns(seed, 30)] + [(p, True) for p in F.gen_patterns_u(seed, 12)]
    for p, uni in pats:
        try:
            o = E.Compiled(p)
        except Exception:
            continue
        if F.has_empty_loop(o.prog) and not o.find_machine.memo:
            continue
        try:
            c = Compiled(p).to(0)
            if c.info.ref_findall_offered != 1:          # Tagged-DFA class: refused in reference mode; the kernels are under test here
                c = Compiled(p, stdlib=True).to(0)
        except _capi.RgxError:
            refused += 1
            continue
        cm = CMatcher(p, q8=False)
        if len(sys.argv) > 3:
            print("pattern", repr(p), flush=True)
        slow_oracle = False
        for n in (0, 3, 64, 1000, 70000):
            if n >= 20000 and (cm.memo or slow_oracle):
                continue                   # (the reference's backtracker can be super-linear: keep the sweep moving)
            b = F.gen_input_u(rng, max(n // 2, 1) if n else 0) if uni else F.gen_input(rng, n)
            arr = np.frombuffer(b, dtype=np.uint8).copy() if len(b) else np.zeros(0, dtype=np.uint8)
            t1 = time.time()
            if n >= 20000:
                got_o = oracle_bounded(p, b, c.ncap)
                if got_o is None:
                    print("ORACLE-SLOW seed", seed, repr(p), "n", len(b), "(no comparison)", flush=True)
                    slow_oracle = True
                    continue
                exp, cnt = got_o
            else:
                exp, cnt = cm.find_all_np(arr)
            t2 = time.time()
            slow_oracle = slow_oracle or (t2 - t1 > 0.005 * max(1, len(b) // 1000))   # super-linear oracle: no 70 KB call
            try:
                spans, res = c.FindAllSpans(b)
            except _capi.RgxError as ex:
                if ex.status != _capi.RGX_E_UNSUPPORTED:
                    raise
                refused += 1                 # the step budgets of the fallback kernels: a refusal, not an answer
                print("REFUSED seed", seed, repr(p), "n", len(b), str(ex)[:60], flush=True)
                continue
            t3 = time.time()
            if len(sys.argv) > 3 and (t2 - t1 > 2 or t3 - t2 > 2):
                print("SLOW", repr(p), "n", len(b), "oracle %.1fs gpu %.1fs" % (t2 - t1, t3 - t2), "matches", cnt, flush=True)
            got = spans.cpu().numpy()
            tot += 1
            if not (res.total == cnt and got.shape == exp.shape and np.array_equal(got, exp)):
                bad += 1
                print("MISMATCH seed", seed, repr(p), "n", len(b), "gpu", int(res.total), "oracle", cnt, "kernel", c.info.scan_kernel, flush=True)
                m = min(len(got), len(exp))
                d = np.nonzero((got[:m] != exp[:m]).any(axis=1))[0]
                if len(d):
                    k = int(d[0])
                    print("   row", k, "gpu", got[k].tolist(), "oracle", exp[k].tolist(), "text", b[max(0, exp[k][0] - 12):exp[k][1] + 12], flush=True)
                if bad > 20:
                    sys.exit(1)
    print("seed", seed, "done; compared", tot, "bad", bad, "refused", refused, "%.0fs" % (time.time() - t0), flush=True)
print("TOTAL compared", tot, "bad", bad, "refused", refused)
