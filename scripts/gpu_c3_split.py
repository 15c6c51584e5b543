"""Where the C3 batch kernel's time goes: the same call with parts switched off (RGX_C3_SKIP bits; results are wrong with 1 or 2).
python scripts/gpu_c3_split.py [nstr]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 2 and sys.argv[2] == "child":
    sys.path.insert(0, ROOT)
    import torch
    from regengo_amd import Compiled, synth
    n = int(sys.argv[1])
    data, offsets = synth.email_batch_np(n)
    concat = torch.from_numpy(data).cuda(); offs = torch.from_numpy(offsets).cuda()
    c = Compiled(r"(?P<user>\w+)@(?P<domain>\w+)").to(0)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ts = []
    for _ in range(8):
        ev[0].record(); r = c.FindBatchDevice(concat, offs); ev[1].record(); ev[1].synchronize()
        ts.append(ev[0].elapsed_time(ev[1]))
    print("RGX_C3_SKIP=%s  %.3f ms (min of 8)  found=%d" % (os.environ.get("RGX_C3_SKIP", "-"), min(ts), int((r[0] == 1).sum().item())), flush=True)
else:
    n = sys.argv[1] if len(sys.argv) > 1 else "10000000"
    for v in (None, "4", "2", "1", "3"):
        env = dict(os.environ)
        if v: env["RGX_C3_SKIP"] = v
        subprocess.run([sys.executable, os.path.abspath(__file__), n, "child"], env=env)
