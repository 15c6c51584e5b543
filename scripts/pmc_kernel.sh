#!/bin/bash
# usage: scripts/pmc_kernel.sh <kernel substring> <pattern>   -- SQ counters of one kernel of scripts/gpu_one_full.py (two passes)
export TMPDIR=/tmp
K="$1"; PAT="$2"
mkdir -p /tmp/pk
for grp in "a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS" "b SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM"; do
  set -- $grp; name=$1; shift
  rm -rf /tmp/pk/$name
  rocprofv3 --pmc "$@" --output-format csv -d /tmp/pk/$name -o pk -- python scripts/gpu_one_full.py "$PAT" 2 > /tmp/pk_$name.log 2>&1
done
python - "$K" <<'PY'
import csv, glob, sys, collections
k = sys.argv[1]
acc = collections.defaultdict(list)
for f in glob.glob("/tmp/pk/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if k in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for c, v in sorted(acc.items()):
    print("%-24s n=%d mean=%.0f" % (c, len(v), sum(v) / len(v)))
PY
