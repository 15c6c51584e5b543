#!/bin/bash
# usage: scripts/pmc_one.sh <outdir under gpurun_out> <pattern>   -- SQ/GRBM counters of the scan kernel of ONE pattern over the
# 1 GiB web-log corpus (scripts/gpu_one.py: count-only launches), counter groups in separate passes, no tracing alongside
export TMPDIR=/tmp
OUT=gpurun_out/$1; PAT="$2"
mkdir -p /tmp/p $OUT
run() { # name counters...
  name=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d /tmp/p/$name -o $name -- python scripts/gpu_one.py "$PAT" > /tmp/$name.log 2>&1
  python - "$name" <<'PY' >> $OUT/summary.txt
import csv, glob, sys, collections
name = sys.argv[1]
acc = collections.defaultdict(list)
kn = set()
for f in glob.glob("/tmp/p/%s/*counter_collection.csv" % name):
    for r in csv.DictReader(open(f)):
        if "rgx::" in r["Kernel_Name"] and "scan" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            kn.add(r["Kernel_Name"][:60])
for k, v in sorted(acc.items()):
    print("%-24s n=%d mean=%.1f" % (k, len(v), sum(v) / len(v)))
if name == "a":
    print("kernel:", sorted(kn))
PY
}
echo "pattern: $PAT" > $OUT/summary.txt
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES
run b SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_WAIT_INST_LDS
run c GRBM_GUI_ACTIVE SQ_INSTS_VALU
tail -3 /tmp/a.log >> $OUT/summary.txt
cat $OUT/summary.txt
