#!/bin/bash
# stage timings of tdfa_ends_sparse_kernel (experiment build): RGX_SP_DEBUG=1 staging only, 2 + filter, 3 + candidate list, 0 everything
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
RGX_EXTRA_FLAGS=-DRGX_EXPERIMENT python -m regengo_amd.build > /tmp/build.log 2>&1 || { tail -5 /tmp/build.log; exit 1; }
for d in 1 2 3 0; do
  rm -rf /tmp/p/st$d
  RGX_EXTRA_FLAGS=-DRGX_EXPERIMENT RGX_SP_DEBUG=$d timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p/st$d -o kt -- python scripts/gpu_tdfa_one.py ${1:-0} 1024 > /tmp/st$d.log 2>&1
  f=$(ls /tmp/p/st$d/*kernel_stats.csv /tmp/p/st$d/*/*kernel_stats.csv 2>/dev/null | head -1)
  echo "RGX_SP_DEBUG=$d: $(grep tdfa_ends_sparse $f | awk -F, '{print $(NF-4)}' | head -1) ns avg"
done
