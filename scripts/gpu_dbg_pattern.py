import sys; sys.path.insert(0,'.')
import numpy as np, torch
from regengo_amd import Compiled, synth
from oracle.gen_c import CMatcher
pat=sys.argv[1] if len(sys.argv)>1 else r"(?P<full>(?P<name>[\w.+-]+)@(?P<host>[\w.-]+))(?P<extra>\s.*)?"
tile=synth.web_log_tile(); tile=tile[:tile.rfind(b"\n")+1]
cm=CMatcher(pat,q8=False)
c=Compiled(pat,stdlib=True).to(0)
print("kernel", c.info.scan_kernel)
tile2=tile*3
for lo,hi in ((0,2*len(tile)),(0,len(tile)+300001),(0,len(tile)+70000),(1000000,1500000),(1048503,1048503+400000),(0,len(tile)),(0,1400000),(0,1410000),(0,1420000),(0,1500000)):
    data=tile2[lo:hi]
    exp,cnt=cm.find_all_np(np.frombuffer(data,dtype=np.uint8).copy())
    sp,res=c.FindAllSpans(data)
    got=sp.cpu().numpy()
    ok=res.total==cnt and np.array_equal(got[:,:2],exp[:,:2])
    print(lo,hi,"count",res.total,cnt,"spans_equal",ok, "captures_equal", np.array_equal(got,exp))
    if not ok:
        m=min(len(got),len(exp))
        d=np.nonzero((got[:m,:2]!=exp[:m,:2]).any(axis=1))[0]
        if len(d):
            k=int(d[0]); print("   first diff row",k,got[k,:2].tolist(),exp[k,:2].tolist(), "prev", exp[k-1,:2].tolist() if k else None)
            print("   text", data[exp[k,0]-80:exp[k,1]+10])
