#!/bin/bash
# usage: scripts/pmc_kernel_each.sh <kernel substring> <pattern>   -- SQ counters of one kernel of scripts/gpu_one_full.py, PER LAUNCH
# (count-only launches first, full launches behind them: the difference is the emission)
export TMPDIR=/tmp
K="$1"; PAT="$2"
mkdir -p /tmp/pk
rm -rf /tmp/pk/e
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM --output-format csv -d /tmp/pk/e -o pk -- python scripts/gpu_one_full.py "$PAT" 2 > /tmp/pk_e.log 2>&1
python - "$K" <<'PY'
import csv, glob, sys, collections
k = sys.argv[1]
acc = collections.defaultdict(dict)
for f in glob.glob("/tmp/pk/e/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if k in r["Kernel_Name"]:
            acc[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
for d in sorted(acc):
    print(d, " ".join("%s=%.3gM" % (c.replace("SQ_", ""), v / 1e6) for c, v in sorted(acc[d].items())))
PY
