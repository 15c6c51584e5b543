#!/bin/bash
# Round-3 evidence in one go (on the MI355X box from the repo root; results under gpurun_out/r03/, copied into profiles/):
#   un-profiled bench lines c2 (+ the adversarial C2b), c3, c4, c5; rocprofv3 --kernel-trace --stats of each; PMC traffic + SQ counters
#   of the dominant kernels of c2, c3, c4 (scripts/pmc_traffic.sh: separate passes); SQ counters of the package kernel.
export TMPDIR=/tmp
OUT=gpurun_out/r03
mkdir -p $OUT /tmp/p
cd /root/repo
for c in c2 c3 c4 c5; do
  timeout 900 python bench.py --config $c > $OUT/r03_bench_$c.json 2> $OUT/bench_$c.err || echo "bench $c failed"
done
timeout 600 python bench.py --adversarial --no-alt > $OUT/r03_bench_c2b_adversarial.json 2> $OUT/bench_c2b.err || echo "bench c2b failed"
for c in c2 c3 c4 c5; do
  st=""; [ $c = c2 ] && st="--steps 20 --warmup 3 --no-alt"; [ $c = c5 ] && st="--steps 1 --warmup 0"; [ $c = c4 ] && st="--steps 1 --warmup 0"; [ $c = c3 ] && st="--steps 3 --warmup 1"
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p/kt_$c -o kt -- python bench.py --config $c --no-cpu-baseline $st > $OUT/r03_bench_${c}_under_rocprof.json 2> /tmp/kt_$c.err
  f=$(find /tmp/p/kt_$c -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && grep -E '^"Name"|rgx::' "$f" > $OUT/r03_kernel_stats_$c.csv
done
scripts/pmc_traffic.sh c2 scan_exact $OUT/r03_pmc_c2.json > /dev/null 2>&1
scripts/pmc_traffic.sh c3 batch_search $OUT/r03_pmc_c3.json > /dev/null 2>&1
scripts/pmc_traffic.sh c4 scan_us_pair $OUT/r03_pmc_c4.json > /dev/null 2>&1
scripts/pmc_script.sh batch_multi scripts/gpu_package_time.py > $OUT/r03_sq_package.txt 2>&1
ls -la $OUT
