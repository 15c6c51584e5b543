#!/bin/bash
# Round-6 evidence in one go (on the MI355X box from the repo root; results under gpurun_out/r06/, copied into profiles/):
#   PMC traffic + SQ counters of the dominant kernels of c2, c3, c4 (scripts/pmc_traffic.sh: separate passes); un-profiled bench lines c2
#   (+ the adversarial C2b), c3 (default engine and --force-tdfa), c4, c5; rocprofv3 --kernel-trace --stats of each.
export TMPDIR=/tmp
OUT=gpurun_out/r06
mkdir -p $OUT /tmp/p
cd /root/repo
which=${1:-all}
# the counter passes first, copied next to the older evidence: the bench lines below then cite THIS run's files (roofline.traffic_source)
if [ $which = all ] || [ $which = c2 ]; then scripts/pmc_traffic.sh c2 scan_exact $OUT/r06_pmc_c2.json > /dev/null 2>&1; fi
if [ $which = all ] || [ $which = c3 ]; then scripts/pmc_traffic.sh c3 batch_tiny $OUT/r06_pmc_c3.json > /dev/null 2>&1; scripts/pmc_traffic.sh c3 tdfa_batch $OUT/r06_pmc_c3t.json --force-tdfa > /dev/null 2>&1; fi
if [ $which = all ] || [ $which = c4 ]; then scripts/pmc_traffic.sh c4 scan_fc $OUT/r06_pmc_c4.json > /dev/null 2>&1; fi
cp $OUT/r06_pmc_*.json profiles/ 2>/dev/null
for c in c2 c3 c4 c5; do
  [ $which != all ] && [ $which != $c ] && continue
  timeout 900 python bench.py --config $c --no-other-configs > $OUT/r06_bench_$c.json 2> $OUT/bench_$c.err || echo "bench $c failed"
done
if [ $which = all ] || [ $which = c3 ]; then
  timeout 600 python bench.py --config c3 --force-tdfa > $OUT/r06_bench_c3_force_tdfa.json 2> $OUT/bench_c3t.err || echo "bench c3 tdfa failed"
fi
if [ $which = all ] || [ $which = c2 ]; then
  timeout 600 python bench.py --adversarial --no-alt > $OUT/r06_bench_c2b_adversarial.json 2> $OUT/bench_c2b.err || echo "bench c2b failed"
fi
for c in c2 c3 c3t c4 c5; do
  [ $which != all ] && [ $which != ${c%t} ] && continue
  st=""; cfg=$c
  [ $c = c2 ] && st="--steps 20 --warmup 3 --no-alt"; [ $c = c5 ] && st="--steps 1 --warmup 0"; [ $c = c4 ] && st="--steps 1 --warmup 0"; [ $c = c3 ] && st="--steps 3 --warmup 1"
  [ $c = c3t ] && { st="--steps 3 --warmup 1 --force-tdfa"; cfg=c3; }
  rm -rf /tmp/p/kt_$c
  # (c4: ONE round in flight under the profiler -- with two, the windows' kernels overlap and a traced duration counts the other window's share
  # of the GPU; the bench line's roofline takes its kernel time from a one-round region for the same reason)
  RGX_C4_DEPTH=1 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p/kt_$c -o kt -- python bench.py --config $cfg --no-cpu-baseline $st > $OUT/r06_bench_${c}_under_rocprof.json 2> /tmp/kt_$c.err
  f=$(find /tmp/p/kt_$c -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && grep -E '^"Name"|rgx::' "$f" > $OUT/r06_kernel_stats_$c.csv
done
# the default line as the driver runs it (other_configs legs included)
if [ $which = all ]; then timeout 900 python bench.py > $OUT/r06_bench_default.json 2> $OUT/bench_default.err || echo 'default bench failed'; fi
ls -la $OUT
