#!/bin/bash
# usage: scripts/profile.sh <tag>   -- rocprofv3 kernel stats + HBM PMC passes of `bench.py`; summaries -> gpurun_out/prof_<tag>/
export TMPDIR=/tmp
TAG=$1; shift
OUT=gpurun_out/prof_$TAG
rm -rf /tmp/p; mkdir -p /tmp/p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p/kt -o bench -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-alt "$@" > $OUT/bench_under_rocprof.log 2>&1
python - <<'PY' > $OUT/kernel_stats.csv
import csv, glob
rows = list(csv.DictReader(open(glob.glob("/tmp/p/kt/*kernel_stats.csv")[0])))
keep = [r for r in rows if "rgx::" in r["Name"]] + [r for r in rows if "rgx::" not in r["Name"]][:6]
w = csv.DictWriter(__import__("sys").stdout, fieldnames=list(rows[0].keys())); w.writeheader()
for r in keep:
    r["Name"] = r["Name"][:150]
    w.writerow(r)
PY
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/p/f -o f -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt "$@" > /tmp/f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/p/w -o w -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt "$@" > /tmp/w.log 2>&1
python - "$OUT" <<'PY'
import csv, glob, json, sys
out = sys.argv[1]
def mean(pat, counter):
    v = [float(r["Counter_Value"]) for f in glob.glob(pat) for r in csv.DictReader(open(f))
         if "rgx::" in r["Kernel_Name"] and "scan" in r["Kernel_Name"] and r["Counter_Name"] == counter]
    return (sum(v) / len(v), len(v)) if v else (None, 0)
fetch, nf = mean("/tmp/p/f/*counter_collection.csv", "FETCH_SIZE")
write, nw = mean("/tmp/p/w/*counter_collection.csv", "WRITE_SIZE")
d = {"kernel": "rgx scan kernel", "FETCH_SIZE_KB_mean": fetch, "WRITE_SIZE_KB_mean": write, "dispatches": [nf, nw],
     "note": "separate --pmc passes; gfx950 FETCH_SIZE counts 128-B requests as 64 B for wide coalesced streaming reads "
             "(MI355X_MICROARCH.md, HBM section) so the read side is doubled; WRITE_SIZE used as reported (uncalibrated)",
     "hbm_read_bytes_per_launch": None if fetch is None else fetch * 1024 * 2,
     "hbm_write_bytes_per_launch": None if write is None else write * 1024}
if fetch is not None and write is not None:
    d["hbm_bytes_per_launch"] = d["hbm_read_bytes_per_launch"] + d["hbm_write_bytes_per_launch"]
json.dump(d, open(out + "/pmc.json", "w"), indent=1)
print(json.dumps(d))
PY
tail -1 $OUT/bench_under_rocprof.log | cut -c1-400
head -3 $OUT/kernel_stats.csv
