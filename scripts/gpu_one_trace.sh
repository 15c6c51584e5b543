export TMPDIR=/tmp
cd /root/repo
mkdir -p gpurun_out/s5
for idx in ${@:-114 221 234}; do
p=$(python - $idx <<'PY'
import json, sys
print(json.load(open("tests/golden/c5_counts.json"))["patterns"][int(sys.argv[1])]["pattern"])
PY
)
rm -rf /tmp/p/one_$idx
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p/one_$idx -o kt -- python scripts/gpu_one_full.py "$p" 4 > gpurun_out/s5/one_$idx.txt 2>/dev/null
f=$(find /tmp/p/one_$idx -name "*kernel_stats.csv" | head -1)
grep -E '^"Name"|rgx::' "$f" | cut -c1-160 >> gpurun_out/s5/one_$idx.txt
done
