import sys; sys.path.insert(0,'.')
import numpy as np, random
from regengo_amd import Compiled
from oracle.gen_c import CMatcher
from tests import _fuzzgen as F
pat=sys.argv[1]
cm=CMatcher(pat,q8=False)
c=Compiled(pat,stdlib=True).to(0)
print("kernel",c.info.scan_kernel)
rng=random.Random(3)
for n in (64,100,128,200,256,1000,16384,16500,40000):
    data=F.gen_input(rng,n)
    exp,cnt=cm.find_all_np(np.frombuffer(data,dtype=np.uint8).copy())
    sp,res=c.FindAllSpans(data)
    got=sp.cpu().numpy()
    print(n,"gpu",res.total,"oracle",cnt)
    if res.total!=cnt and n<=256:
        print("  got",got[:,:2].tolist()[:40]); print("  exp",exp[:,:2].tolist()[:40])
