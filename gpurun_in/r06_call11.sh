#!/bin/bash
# round 6, call 11: the opt-in maximum sizes (2^30 and 2^32 leaves in one buffer) on the final library, then a 10-minute differential soak
set -u
O=gpurun_out/r06
mkdir -p $O
( time P252_TEST_HUGE=1 timeout 1200 python -m pytest "tests/test_gpu_fullsize.py::test_beyond_4GiB_leaves_in_one_buffer" -m gpu -q --durations=5 ) > $O/huge.txt 2>&1; grep "passed\|failed\|skipped\|s call\|real" $O/huge.txt
timeout 900 python bench_tools/soak_check.py --long 10 2>&1 | grep -v "amdgpu.ids\|RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl" > $O/soak_long10.txt; tail -3 $O/soak_long10.txt
