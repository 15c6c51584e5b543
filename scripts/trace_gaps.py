"""Gaps between the kernels of a rocprofv3 --kernel-trace run (csv): python scripts/trace_gaps.py <kernel_trace.csv> [first] [n]
Prints, for n kernels from the first-th on: name, duration, idle time in front of it (us)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first = int(sys.argv[2]) if len(sys.argv) > 2 else len(rows) // 2
n = int(sys.argv[3]) if len(sys.argv) > 3 else 24
prev = None
for r in rows[first:first + n]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("rgx::(anonymous namespace)::", "").replace("void ", "")[:44]
    print("%-44s %9.1f us   gap %8.1f us" % (name, (e - s) / 1e3, (s - prev) / 1e3 if prev else 0.0))
    prev = e
