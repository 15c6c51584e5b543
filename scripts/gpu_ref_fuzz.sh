#!/bin/bash
# usage: scripts/gpu_ref_fuzz.sh <first seed> <last seed> [log]: the random-pattern tests of the reference-mode entry points (MatchBytes / FindBytes
# per string, FindAllBytes, FindReader, Replace, the readers, the Tagged DFA's batch, chain and FindAll wrapper) over a wider range of seeds, ONE SEED
# PER PROCESS under its own timeout: the checker is the oracle's backtracker in Python, and a random pattern can be catastrophic for it (round 5: a
# 36-seed run in one process sat in such a call until the box's time limit, 45 GPU-minutes).  SIGINT first, SIGKILL half a minute later; a seed
# that times out is logged and skipped.  Every seed leaves a line: "seed N: ok | TIMEOUT | FAILED".
lo=${1:-200}; hi=${2:-212}; log=${3:-gpurun_out/ref_fuzz.txt}
mkdir -p "$(dirname "$log")"
echo "# gpu_ref_fuzz.sh seeds $lo:$hi $(date -u +%Y-%m-%dT%H:%M:%SZ)" >> "$log"
ok=0; to=0; bad=0
for ((s=lo; s<hi; s++)); do
  tmp=$(mktemp)
  RGX_FUZZ_SEEDS=$s:$((s+1)) timeout -s INT -k 30 120 python -m pytest tests/test_gpu_reference_mode.py tests/test_gpu_replace.py tests/test_gpu_transform.py tests/test_gpu_tdfa.py -q -s -k "random" > "$tmp" 2>&1
  rc=$?
  grep -v amdgpu.ids "$tmp" | grep -E "^programs|^patterns" >> "$log"
  if [ $rc -eq 0 ]; then echo "seed $s: ok" >> "$log"; ok=$((ok+1));
  elif [ $rc -eq 124 ] || [ $rc -eq 137 ] || [ $rc -eq 2 ]; then echo "seed $s: TIMEOUT / interrupted (the oracle on a catastrophic pattern)" >> "$log"; to=$((to+1));
  else echo "seed $s: FAILED (rc $rc)" >> "$log"; grep -E "^E |Error|assert" "$tmp" | head -8 >> "$log"; bad=$((bad+1)); fi
  rm -f "$tmp"
done
echo "# done: $ok ok, $to timeouts, $bad failed of $((hi-lo)) seeds from $lo" >> "$log"
tail -6 "$log"
