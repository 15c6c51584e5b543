#!/bin/bash
# experiment build (RGX_EXTRA_FLAGS=-DRGX_EXPERIMENT): rgx_scan_fc.hip cut short behind a stage, timed (the results are void)
# stage 1: loads + staging; 2: + filter + candidate list; 3: everything but the walks; 4: look-back bases made up; 5: no emission; 0: all
export RGX_EXTRA_FLAGS=-DRGX_EXPERIMENT
STAGES=${STAGES:-"4 0"}
for pat in '\[(INFO|WARN|ERROR)\]' '(?P<full>(?P<proto>https?|ftp)://(?P<host>[\w.-]+)(?P<port>:\d+)?(?P<path>/[\w./-]*)?)'; do
  for d in $STAGES; do
    echo "== stage $d"; RGX_FC_DEBUG=$d RGX_FC_FORCE=1 timeout 300 python scripts/gpu_fc_prof.py "$pat" 2>&1 | grep -v amdgpu.ids | tail -1
  done
done
