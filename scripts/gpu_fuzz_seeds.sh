#!/bin/bash
# scripts/gpu_fuzz_seeds.sh first n: the differential sweep one seed per process, each under its own timeout (a seed that hangs costs
# 90 s, not the box); stops at the first seed that times out or fails.
first=${1:-1100}; n=${2:-20}
for ((s=first; s<first+n; s++)); do
  timeout 90 python scripts/gpu_fuzz_sweep.py $s 1 2>&1 | grep -v amdgpu.ids | grep -E "MISMATCH|row|REFUSED|TOTAL|Error|error" 
  rc=${PIPESTATUS[0]}
  if [ $rc -ne 0 ]; then echo "seed $s: exit $rc -- stopping"; break; fi
done
