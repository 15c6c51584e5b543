#!/bin/bash
# usage: scripts/gpu_ref_fuzz.sh <first seed> <last seed> [log]: the random-pattern tests of the reference-mode entry points (MatchBytes / FindBytes
# per string, FindAllBytes, FindReader, Replace, the readers, the Tagged DFA's batch and FindAll wrapper) over a wider range of seeds
lo=${1:-200}; hi=${2:-212}; log=${3:-gpurun_out/ref_fuzz.txt}
mkdir -p "$(dirname "$log")"
echo "# gpu_ref_fuzz.sh seeds $lo:$hi $(date -u +%Y-%m-%dT%H:%M:%SZ)" >> "$log"
RGX_FUZZ_SEEDS=$lo:$hi timeout 3000 python -m pytest tests/test_gpu_reference_mode.py tests/test_gpu_replace.py tests/test_gpu_transform.py tests/test_gpu_tdfa.py -q -s -k "random" 2>&1 | grep -v amdgpu.ids | grep -E "programs|patterns|passed|failed|Error|assert" >> "$log"
tail -12 "$log"
