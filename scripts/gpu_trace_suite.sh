#!/bin/bash
# Which patterns of the C5 suite launch the generic scan kernel: kernel trace of scripts/gpu_suite_times.py, the launches in time order
# with the pattern-specific kernels around them (the script walks the patterns in order).
export TMPDIR=/tmp
cd /root/repo
rm -rf /tmp/p/ts; mkdir -p /tmp/p gpurun_out/r03
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/p/ts -o ts -- python scripts/gpu_suite_times.py > gpurun_out/r03/c5_pattern_times_traced.txt 2>/tmp/ts.err
f=$(find /tmp/p/ts -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
out = []
for r in rows:
    n = r["Kernel_Name"]
    if "rgx" not in n: continue
    short = n.replace("void rgx::(anonymous namespace)::", "").replace("rgx::(anonymous namespace)::", "").split("(")[0]
    out.append((short, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
# compress runs
res = []
for s, ms in out:
    if res and res[-1][0] == s: res[-1][1] += 1; res[-1][2] += ms
    else: res.append([s, 1, ms])
with open("gpurun_out/r03/c5_kernel_sequence.txt", "w") as f:
    for s, n, ms in res: f.write("%-55s x%-3d %.3f ms\n" % (s, n, ms))
print(len(res), "runs")
PY
