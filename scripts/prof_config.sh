#!/bin/bash
# usage: scripts/prof_config.sh <name> <bench args...>   -- rocprofv3 kernel trace + stats of one bench.py configuration;
# the per-kernel summary lands in gpurun_out/<name>_kernel_stats.csv (copy into profiles/ to keep it)
export TMPDIR=/tmp
NAME=$1; shift
mkdir -p gpurun_out /tmp/prof_$NAME
cd /root/repo
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$NAME -o $NAME -- python bench.py --no-cpu-baseline "$@" > gpurun_out/${NAME}_bench.json 2> gpurun_out/${NAME}_bench.err
f=$(find /tmp/prof_$NAME -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" gpurun_out/${NAME}_kernel_stats.csv && head -12 gpurun_out/${NAME}_kernel_stats.csv | cut -c1-220
tail -c 600 gpurun_out/${NAME}_bench.json
