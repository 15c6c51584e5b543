timeout 150 scripts/trace_pattern.sh p153 '(?P<full>(?P<name>[\w.+-]+)@(?P<host>[\w.-]+))(?P<extra>\s.*)?' 3 </dev/null | head -9 | cut -c1-200
timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_configs.py -x -q 2>&1 | tail -4
