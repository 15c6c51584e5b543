#!/bin/bash
# experiment build (RGX_EXTRA_FLAGS=-DRGX_EXPERIMENT): rgx_scan_fc.hip cut short behind a stage, timed (the results are void)
export RGX_EXTRA_FLAGS=-DRGX_EXPERIMENT
for pat in '\[(INFO|WARN|ERROR)\]' '(?P<full>(?P<proto>https?|ftp)://(?P<host>[\w.-]+)(?P<port>:\d+)?(?P<path>/[\w./-]*)?)'; do
  for d in 4 0; do
    echo "== stage $d"; RGX_FC_DEBUG=$d timeout 300 python scripts/gpu_fc_prof.py "$pat" 2>&1 | grep -v amdgpu.ids | tail -1
  done
done
