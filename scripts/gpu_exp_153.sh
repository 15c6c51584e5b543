#!/bin/bash
# experiment build: pattern 153 (matches run to the end of a line) through the pair kernel, count-only against full, under the switches
# that change how its tiles wait for one another
export RGX_EXTRA_FLAGS=-DRGX_EXPERIMENT
P='(?P<full>(?P<name>[\w.+-]+)@(?P<host>[\w.-]+))(?P<extra>\s.*)?'
for env in "A=1" "RGX_NO_US_WSYNC=1" "RGX_TICKETS=1" "RGX_US_PER_CU=2" "RGX_US_PER_CU=8"; do
  echo "== $env"
  env $env timeout 600 python scripts/gpu_one_full.py "$P" 3 2>&1 | grep -v amdgpu.ids | tail -4
done
