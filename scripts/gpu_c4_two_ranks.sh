#!/bin/bash
# bench.py --config c4 --gpus 2 as two PROCESSES on one GPU over the CCL test double (tests/ccl_shim.c): the rounds' all-gather, the
# full gather (8 * ncap bytes per match) and the compact one (rgx_sharded_gather_offsets: 8 bytes per match), each timed as a pass
# with the gather minus one without.  The double moves bytes through POSIX shared memory + hipMemcpy: the figures are the protocol's,
# not xGMI's.
mkdir -p tests/_build
gcc -shared -fPIC -O1 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include tests/ccl_shim.c -L/opt/rocm/lib -lamdhip64 -lrt -Wl,-rpath,/opt/rocm/lib -o tests/_build/libccl_shim.so || exit 1
unset WORLD_SIZE RANK LOCAL_RANK MASTER_ADDR MASTER_PORT
RGX_SHARDED_CCL_LIB=$PWD/tests/_build/libccl_shim.so RGX_BENCH_ONE_DEVICE=1 timeout 600 python bench.py --config c4 --gpus 2 --windows ${WINDOWS:-2} --no-cpu-baseline "$@" | grep "^{"
