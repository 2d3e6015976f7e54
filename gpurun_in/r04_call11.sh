#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
O=gpurun_out/r04c11; mkdir -p $O
timeout 600 python -m pytest tests/test_bench_multiproc.py -m gpu -q -k "rccl or eight or two_ranks or watchdog" > $O/tests.txt 2>&1; grep -v "amdgpu.ids\|RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl" $O/tests.txt | tail -4
