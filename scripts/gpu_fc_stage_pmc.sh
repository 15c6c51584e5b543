#!/bin/bash
# experiment build: instruction counts of rgx_scan_fc.hip cut short behind a stage (see gpu_fc_stages.sh), one --pmc pass per stage
export RGX_EXTRA_FLAGS=-DRGX_EXPERIMENT TMPDIR=/tmp
PAT=${PAT:-'(?P<full>(?P<proto>https?|ftp)://(?P<host>[\w.-]+)(?P<port>:\d+)?(?P<path>/[\w./-]*)?)'}
python -c "from regengo_amd import build as b; b.build_all()" >/dev/null 2>&1
for d in ${STAGES:-2 3 4 0}; do
  rm -rf /tmp/fps
  RGX_FC_DEBUG=$d RGX_FC_FORCE=1 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d /tmp/fps -o ps -- python scripts/gpu_fc_prof.py "$PAT" > /tmp/fps.log 2>&1
  echo "== stage $d"
  python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("/tmp/fps/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "scan_fc" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("  ".join("%s n=%d mean=%.0f" % (c, len(v), sum(v) / len(v)) for c, v in sorted(acc.items())))
PY
done
