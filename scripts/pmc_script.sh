#!/bin/bash
# usage: scripts/pmc_script.sh <kernel substring> <python script> [args]  -- SQ counters of one kernel of a script (two --pmc passes)
export TMPDIR=/tmp
K="$1"; shift
mkdir -p /tmp/ps
for grp in "a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS" "b SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM" "c GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH"; do
  set -- $grp "$@"; name=$1; shift
  ctrs=(); while [[ "$1" == SQ_* || "$1" == GRBM_* ]]; do ctrs+=("$1"); shift; done
  rm -rf /tmp/ps/$name
  rocprofv3 --pmc "${ctrs[@]}" --output-format csv -d /tmp/ps/$name -o ps -- python "$@" > /tmp/ps_$name.log 2>&1
done
python - "$K" <<'PY'
import csv, glob, sys, collections
k = sys.argv[1]
acc = collections.defaultdict(list)
for f in glob.glob("/tmp/ps/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if k in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for c, v in sorted(acc.items()):
    print("%-24s n=%d mean=%.0f" % (c, len(v), sum(v) / len(v)))
PY
