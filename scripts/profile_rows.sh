#!/bin/bash
# usage: scripts/profile_rows.sh   -- rocprofv3 --kernel-trace --stats for the rows next to the headline: generic FindAll
# kernels + capture back-trace (1 GiB web log), batch kernels (C3, 10 M strings), Replace/Transform splice (256 MiB date log).
# Output: gpurun_out/rows_kernel_stats.csv (rgx kernels only, one section per workload)
export TMPDIR=/tmp
OUT=gpurun_out/rows_kernel_stats.csv
: > $OUT
run() { # tag cmd...
  tag=$1; shift
  rm -rf /tmp/pr; mkdir -p /tmp/pr
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pr -o r -- "$@" > /tmp/pr/log.txt 2>&1
  python - "$tag" >> $OUT <<'PY'
import csv, glob, sys
tag = sys.argv[1]
rows = list(csv.DictReader(open(glob.glob("/tmp/pr/*kernel_stats.csv")[0])))
print("# workload: " + tag)
print("Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs")
for r in rows:
    if "rgx" in r["Name"] or "ROCPRIM_400200" in r["Name"]:      # (other rocprim kernels belong to torch: data generation)
        print('"%s",%s,%s,%s,%s,%s' % (r["Name"][:110], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["MinNs"], r["MaxNs"]))
PY
}
run "generic FindAll kernels, 1 GiB web log, 6 patterns (scripts/gpu_generic2.py)" python scripts/gpu_generic2.py
run "C3 batch, 10 M e-mail strings (scripts/gpu_batch.py)" python scripts/gpu_batch.py
run "Transform chunk REPLACE \$day/\$month/\$year, 256 MiB date log x6 (scripts/gpu_transform_prof.py 256 0)" python scripts/gpu_transform_prof.py 256 0
run "Transform chunk SELECT, 256 MiB date log x6 (scripts/gpu_transform_prof.py 256 1)" python scripts/gpu_transform_prof.py 256 1
cat $OUT
