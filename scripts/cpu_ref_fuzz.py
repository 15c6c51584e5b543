"""Wide CPU sweep of tests/test_ref_engine.py::test_reference_engines_on_random_patterns: the product's reference-mode engines (host mirrors in
csrc/hosttest) against the oracle over random patterns.  usage: python scripts/cpu_ref_fuzz.py <first seed> <last seed>"""
import sys, random, time
sys.path.insert(0,'/root/repo')
from oracle import engines as E, syntax as S
from tests import _fuzzgen as F
from tests._hosttest import HostProgram
from regengo_amd import codegen
rng = random.Random(int(sys.argv[1]))
t0=time.time()
n=dict(pats=0, ref_find=0, memo_find=0, ref_match=0, memo_match=0, dead=0, high=0)
bad=[]
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    for p in F.gen_patterns(seed, 60):
        try: o = E.Compiled(p)
        except Exception: continue
        if F.has_empty_loop(o.prog) and not o.find_machine.memo: continue
        try: hp = HostProgram(p)
        except ValueError: continue
        n['pats']+=1
        info = codegen.Program(p).info
        exp = (-1, 0) if o.prog.numcap <= 2 else (1, len(o.tdfa.states)) if o.sel.find_engine == "tdfa" else (2 if o.sel.find_engine == "tnfa" else 0, 0)
        if (info.ref_find_engine, info.ref_tdfa_states) != exp: bad.append(('sel',p))
        dead = o.thompson is not None and any(i.op == S.InstEmptyWidth for i in o.prog.inst)
        if dead:
            n['dead']+=1
            if not info.ref_match_offered: bad.append(('dead-not-offered',p))
        for b in [F.gen_input(rng, rng.choice([0, 1, 5, 40, 120])) for _ in range(6)] + [b"\xc3\xa9", b"aa\xc3\xa9b", b"ab\xff."]:
            try:
                if o.tdfa is None:
                    want = o.FindBytes(b)
                    got = hp.ref_find(b)
                    if got is not NotImplemented:
                        n['ref_find']+=1
                        if got != want: bad.append(('ref_find',p,b,got,want))
                    got = hp.memo_find(b)
                    if got is not NotImplemented:
                        n['memo_find']+=1
                        if got != want: bad.append(('memo_find',p,b,got,want))
                want = o.MatchBytes(b)
                got = hp.ref_match(b)
                if got is not NotImplemented:
                    n['ref_match']+=1
                    n['high']+=int(o.thompson is not None and any(x >= 0x80 for x in b))
                    if got != want: bad.append(('ref_match',p,b,got,want))
                elif o.thompson is not None:
                    bad.append(('thompson-not-answered',p,b))
                else:
                    got = hp.memo_match(b)
                    if got is not None:
                        n['memo_match']+=1
                        if got != want: bad.append(('memo_match',p,b,got,want))
            except AssertionError as ex:
                bad.append(('assert',p,b,str(ex)))
print(n, len(bad), time.time()-t0)
for x in bad[:10]: print(x)
