URL='(?P<full>(?P<proto>https?|ftp)://(?P<host>[\w.-]+)(?P<port>:\d+)?(?P<path>/[\w./-]*)?)'
timeout 100 python scripts/gpu_one_full.py "$URL" 3 2>&1 | grep full
RGX_CAPS_NO_PRIV=1 timeout 100 python scripts/gpu_one_full.py "$URL" 3 2>&1 | grep full
timeout 100 python scripts/gpu_one_full.py '(?P<user>[\w\.+-]+)@(?P<domain>[\w\.-]+)\.(?P<tld>[\w\.-]+)' 2 2>&1 | grep full
RGX_CAPS_NO_PRIV=1 timeout 100 python scripts/gpu_one_full.py '(?P<user>[\w\.+-]+)@(?P<domain>[\w\.-]+)\.(?P<tld>[\w\.-]+)' 2 2>&1 | grep full
timeout 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_configs.py tests/test_gpu_us.py tests/test_gpu_tdfa.py -x -q 2>&1 | tail -4
