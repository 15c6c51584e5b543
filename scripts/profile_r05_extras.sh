#!/bin/bash
# Round-5 evidence beyond profile_r05.sh (on the MI355X box from the repo root; results under gpurun_out/r05/): the GPU test tier, the
# suite's scan-mode patterns one by one, the Tagged-DFA FindAll wrapper over 1 GiB, config c4 as two ranks over the CCL double with both
# gathers, and -- LAST, it rebuilds the library with -DRGX_EXPERIMENT -- the stage timings and instruction counts of the filter + candidate kernel.
export TMPDIR=/tmp
OUT=gpurun_out/r05
mkdir -p $OUT
timeout 1300 python -m pytest tests -m gpu -q > $OUT/r05_gpu_tests.txt 2>&1; tail -3 $OUT/r05_gpu_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/r05_smoke.txt 2>&1; tail -1 $OUT/r05_smoke.txt
timeout 600 python scripts/gpu_suite_times.py > $OUT/r05_c5_pattern_times_final.txt 2> /dev/null
timeout 900 python scripts/gpu_tdfa_findall.py 16 1024 2>&1 | grep -v amdgpu.ids > $OUT/r05_tdfa_findall.txt
bash scripts/gpu_c4_two_ranks.sh > $OUT/r05_bench_c4_two_ranks_double.json 2> $OUT/c4_two_ranks.err
STAGES="1 2 3 4 5 0" bash scripts/gpu_fc_stages.sh > $OUT/r05_fc_stages.txt 2>&1
bash scripts/gpu_fc_stage_pmc.sh > $OUT/r05_fc_stage_pmc.txt 2>&1
ls -la $OUT | tail -12
