#!/bin/bash
# usage: scripts/pmc.sh <outdir under gpurun_out> [bench args]   -- SQ/GRBM counters for the scan kernel (separate passes)
export TMPDIR=/tmp
OUT=gpurun_out/$1; shift
mkdir -p /tmp/p $OUT
run() { # name counters...
  name=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d /tmp/p/$name -o $name -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt $BARGS > /tmp/$name.log 2>&1
  python - "$name" <<'PY' >> $OUT/summary.txt
import csv, glob, sys, collections
name = sys.argv[1]
acc = collections.defaultdict(list)
for f in glob.glob("/tmp/p/%s/*counter_collection.csv" % name):
    for r in csv.DictReader(open(f)):
        if "rgx::" in r["Kernel_Name"] and "scan" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print("%-24s n=%d mean=%.1f" % (k, len(v), sum(v) / len(v)))
PY
}
BARGS="$@"
: > $OUT/summary.txt
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES
run b SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_WAIT_INST_LDS
run c GRBM_GUI_ACTIVE
cat $OUT/summary.txt
