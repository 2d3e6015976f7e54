#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
O=gpurun_out/r04c6; mkdir -p $O
timeout 600 python -m pytest tests/test_openings_device.py tests/test_c_abi.py -m gpu -q > $O/tests.txt 2>&1; grep -v "amdgpu.ids\|RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl" $O/tests.txt | tail -12
timeout 300 python bench_tools/openings_extract_bench.py 2>&1 | grep -v amdgpu.ids > $O/openings_extract.txt; cat $O/openings_extract.txt
