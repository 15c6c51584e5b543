#!/bin/bash
# SQ counters of the C5 suite's slowest scan kernels, one pattern each (scripts/pmc_kernel.sh: gpu_one_full.py under
# two --pmc passes).  usage: scripts/pmc_c5_laggards.sh [outdir]   -> <outdir>/r05_pmc_c5_<tag>.txt
OUT=${1:-gpurun_out/r05}; mkdir -p $OUT
pat() { python - "$1" <<'PY'
import json, sys
print(json.load(open("tests/golden/c5_counts.json"))["patterns"][int(sys.argv[1])]["pattern"])
PY
}
while read tag kern idx; do
  p="$(pat $idx)"
  { echo "# pattern $idx: $p"; echo "# kernel: $kern"; python scripts/gpu_one_full.py "$p" 2 2>/dev/null | grep -v amdgpu.ids; scripts/pmc_kernel.sh "$kern" "$p"; } > $OUT/r05_pmc_c5_$tag.txt 2>&1
done <<'LIST'
generic_221 scan_kernel<1 221
us2_2 scan_us_kernel<2 2
us4_224 scan_us_kernel<4 224
us8_185 scan_us_kernel<8 185
us2_252 scan_us_kernel<2 252
LIST
