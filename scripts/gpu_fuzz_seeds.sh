#!/bin/bash
# scripts/gpu_fuzz_seeds.sh first n [logfile]: the differential sweep (scripts/gpu_fuzz_sweep.py: random patterns, FindAllBytes on the
# device against the C port of the emitted matcher) ONE SEED PER PROCESS, each under its own timeout -- SIGINT first (python takes it
# between two library calls: no kernel is in flight when the process goes), SIGKILL only a minute later.  (Round 4: a plain
# `timeout 120` killed the process behind a 101-s oracle call of seed 1133 with device work in flight, and the box went with it.)  After a timeout the device is asked whether it still answers (rocm-smi + a one-kernel python probe); the sweep
# stops there.  Every seed leaves a line in the log: "seed N: ok | TIMEOUT | exit rc".
first=${1:-1100}; n=${2:-20}; log=${3:-gpurun_out/fuzz_sweep.txt}
mkdir -p "$(dirname "$log")"
echo "# gpu_fuzz_seeds.sh first=$first n=$n  $(date -u +%Y-%m-%dT%H:%M:%SZ)  $(rocm-smi --showproductname 2>/dev/null | grep -m1 -i 'card series' | sed 's/.*: *//')" >> "$log"
ok=0; to=0; bad=0
for ((s=first; s<first+n; s++)); do
  tmp=$(mktemp)
  timeout -s INT -k 60 300 python scripts/gpu_fuzz_sweep.py $s 1 > "$tmp" 2>&1
  rc=$?
  out=$(grep -v amdgpu.ids "$tmp"); rm -f "$tmp"
  echo "$out" | grep -E "MISMATCH|row|REFUSED|TOTAL|Error|error" >> "$log"
  if [ $rc -eq 0 ]; then echo "seed $s: ok" >> "$log"; ok=$((ok+1));
  elif [ $rc -eq 124 ] || [ $rc -eq 137 ]; then
    echo "seed $s: TIMEOUT (300 s)" >> "$log"; to=$((to+1))
    if timeout 60 python -c "import torch; x=torch.ones(8,device='cuda'); print(float(x.sum()))" >/dev/null 2>&1; then echo "  device answers after the timeout" >> "$log";
    else echo "  DEVICE DOES NOT ANSWER after the timeout -- stopping" >> "$log"; break; fi
  else echo "seed $s: exit $rc" >> "$log"; echo "$out" | tail -5 >> "$log"; bad=$((bad+1)); fi
done
echo "# done: $ok ok, $to timeouts, $bad failed of $n seeds from $first" >> "$log"
tail -3 "$log"
