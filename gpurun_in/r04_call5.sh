#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
O=gpurun_out/r04c5; mkdir -p $O
timeout 900 python -m pytest tests/test_comm_forest.py tests/test_c_abi.py tests/test_cpp_host_api.py tests/test_host_pipeline.py -m gpu -q > $O/tests.txt 2>&1; grep -v "amdgpu.ids\|RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl" $O/tests.txt | tail -5
timeout 400 python bench_tools/forest_bench.py --check 2>&1 | grep -v amdgpu.ids > $O/forest.txt; cat $O/forest.txt
