#!/bin/bash
# usage: scripts/prof_one.sh <name> <pattern>   -- rocprofv3 kernel trace + stats of scripts/gpu_one_full.py for one pattern
export TMPDIR=/tmp
NAME=$1; PAT="$2"
mkdir -p gpurun_out /tmp/prof_$NAME
cd /root/repo
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$NAME -o $NAME -- python scripts/gpu_one_full.py "$PAT" 2 > gpurun_out/${NAME}_one.txt 2>&1
f=$(find /tmp/prof_$NAME -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" gpurun_out/${NAME}_kernel_stats.csv && head -8 gpurun_out/${NAME}_kernel_stats.csv | cut -c1-200
grep -v amdgpu.ids gpurun_out/${NAME}_one.txt | tail -5
