"""Wide CPU sweep of tests/test_tdfa.py::test_product_tables_equal_the_oracle_on_random_patterns: the product's Tagged-DFA construction, find loop and
merged-attempts automaton against the oracle over random patterns.  usage: python scripts/cpu_tdfa_fuzz.py <first seed> <last seed>"""
import sys, random, zlib, time
sys.path.insert(0,'/root/repo')
from oracle import engines as E
from tests import _fuzzgen as F
from tests._hosttest import HostProgram
t0=time.time(); seen=finds=bad=0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    for pat in F.gen_patterns(seed, 60):
        try: o=E.Compiled(pat)
        except Exception: continue
        hp=HostProgram(pat); tb=hp.tdfa_tables()
        if (tb is None)!=(o.tdfa is None): print('CLASS', repr(pat)); bad+=1; continue
        if tb is None: continue
        seen+=1
        ob=o.tdfa.tables()
        for k in ("n_states","transitions","tag_actions","accept","accept_eot","accept_actions"):
            if tb[k]!=ob[k]: print('TABLE',k,repr(pat)); bad+=1; break
        if len(o.tdfa.states)>120: continue
        ob["start_any"]=o.tdfa.start_any
        rnd=random.Random(zlib.crc32(pat.encode()))
        for _ in range(10):
            b=F.tdfa_guided_text(ob,rnd,rnd.randint(1,60))
            want=o.tdfa.find(b)
            if hp.tdfa_find(b)!=want: print('FIND',repr(pat),b); bad+=1; break
            m=hp.tdfa_merged_find(b)
            if m is not NotImplemented and m!=(None if want is None else (want[0],want[1])): print('MERGED',repr(pat),b,m,want); bad+=1; break
            finds+=1
print(seen,finds,bad,time.time()-t0)
