#!/bin/bash
# Kernel statistics of one pattern's count-only and full scans over the 1 GiB web-log corpus (scripts/gpu_one_full.py under
# rocprofv3 --kernel-trace --stats).  usage: scripts/trace_pattern.sh <tag> <pattern> [iters]; output: gpurun_out/trace_<tag>.txt
set -u
tag=$1; pat=$2; iters=${3:-2}
root=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$root/gpurun_out"
out="$root/gpurun_out/trace_$tag.txt"
export TMPDIR=/tmp
d=/tmp/trace_$tag; rm -rf "$d"
(cd "$root" && timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d "$d" -o t -- python scripts/gpu_one_full.py "$pat" "$iters" 2>/dev/null | grep -E "^(count|full)") > "$out" </dev/null
f=$(find "$d" -name "*kernel_stats.csv" 2>/dev/null | head -1)
if [ -n "$f" ] && [ -f "$f" ]; then cut -c1-170 "$f" | head -12 >> "$out"; else echo "no kernel stats" >> "$out"; fi
cat "$out"
