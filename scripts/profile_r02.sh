#!/bin/bash
# Round-2 evidence in one go (run on the MI355X box from the repo root; results under gpurun_out/r02/, copy into profiles/):
#   bench lines c2..c5 (un-profiled), rocprofv3 --kernel-trace --stats of c2 and c4/c3/c5, PMC traffic of the c2 scan kernel
#   (FETCH_SIZE and WRITE_SIZE in separate passes), SQ counters of the one-step-per-byte kernels for the four VERDICT patterns.
export TMPDIR=/tmp
OUT=gpurun_out/r02
mkdir -p $OUT /tmp/p
cd /root/repo
for c in c2 c3 c4 c5; do
  timeout 600 python bench.py --config $c > $OUT/r02_bench_$c.json 2> $OUT/bench_$c.err || echo "bench $c failed"
done
# kernel traces
for c in c2 c3 c4 c5; do
  st=""; [ $c = c2 ] && st="--steps 20 --warmup 3 --no-alt"; [ $c = c5 ] && st="--steps 1 --warmup 0"; [ $c = c4 ] && st="--steps 1 --warmup 0"; [ $c = c3 ] && st="--steps 3 --warmup 1"
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p/kt_$c -o kt -- python bench.py --config $c --no-cpu-baseline $st > $OUT/r02_bench_${c}_under_rocprof.json 2> /tmp/kt_$c.err
  f=$(find /tmp/p/kt_$c -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && grep -E '^"Name"|rgx::' "$f" > $OUT/r02_kernel_stats_$c.csv
done
# HBM traffic of the c2 scan kernel: separate passes per counter
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $ctr --output-format csv -d /tmp/p/pmc_$ctr -o pmc -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt > /tmp/pmc_$ctr.log 2>&1
done
python - <<'PY' > gpurun_out/r02/r02_pmc_c2.json
import csv, glob, json
out = {"kernel": "rgx::scan_exact_kernel (bench.py default, C2)"}
tot = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    vals = []
    for f in glob.glob("/tmp/p/pmc_%s/**/*counter_collection.csv" % ctr, recursive=True):
        for r in csv.DictReader(open(f)):
            if "rgx::" in r["Kernel_Name"] and "scan_exact" in r["Kernel_Name"] and r["Counter_Name"] == ctr:
                vals.append(float(r["Counter_Value"]))
    # every launch scans the same 1 GiB: drop the launches of the warm-up's first touch by taking the median
    vals.sort()
    tot[ctr] = vals[len(vals) // 2] if vals else None
    out[ctr + "_KB_median"] = tot[ctr]
    out[ctr + "_launches"] = len(vals)
if tot.get("FETCH_SIZE") and tot.get("WRITE_SIZE"):
    rd = tot["FETCH_SIZE"] * 1024 * 2     # gfx950: FETCH_SIZE counts 128-byte requests of wide streaming reads as 64 B (MI355X_MICROARCH.md, HBM section)
    wr = tot["WRITE_SIZE"] * 1024
    out.update({"hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr,
                "note": "separate --pmc passes; FETCH_SIZE doubled per the gfx950 correction (MI355X_MICROARCH.md, HBM section); WRITE_SIZE as reported"})
print(json.dumps(out, indent=1))
PY
# SQ counters of the one-step-per-byte kernels, four patterns
: > $OUT/r02_sq_counters.txt
for pat in '(?P<user>\w+)@(?P<domain>\w+)' '(\d+)' '\b[a-z]+\b' '(?P<protocol>https?)://(?P<host>[\w\.-]+)(?::(?P<port>\d+))?(?P<path>/[\w\./]*)?'; do
  for grp in "a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES" "b SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_WAIT_INST_LDS" "c GRBM_GUI_ACTIVE"; do
    set -- $grp; name=$1; shift
    rm -rf /tmp/p/sq_$name
    timeout 300 rocprofv3 --pmc "$@" --output-format csv -d /tmp/p/sq_$name -o sq -- python scripts/gpu_one.py "$pat" > /tmp/sq_$name.log 2>&1
  done
  python - "$pat" <<'PY' >> gpurun_out/r02/r02_sq_counters.txt
import csv, glob, sys, collections
pat = sys.argv[1]
acc = collections.defaultdict(list); kn = set()
for f in glob.glob("/tmp/p/sq_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "rgx::" in r["Kernel_Name"] and "scan" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"])); kn.add(r["Kernel_Name"].split("(")[0][-40:])
print("pattern: %s   kernel: %s   (count-only launches over the 1 GiB web-log corpus, scripts/gpu_one.py)" % (pat, sorted(kn)))
for k, v in sorted(acc.items()):
    print("  %-24s n=%d mean=%.1f" % (k, len(v), sum(v) / len(v)))
n = 1 << 30
if acc.get("SQ_INSTS_VALU"):
    w = sum(acc["SQ_INSTS_VALU"]) / len(acc["SQ_INSTS_VALU"])
    print("  => VALU wave-instructions per input byte %.3f; lane-instructions per byte (x64) %.1f" % (w / n, w * 64 / n))
print()
PY
done
ls -la $OUT
