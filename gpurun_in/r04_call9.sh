#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
ROOT=$(pwd)
O=$ROOT/gpurun_out/r04c9; mkdir -p $O
timeout 600 python -m pytest tests/test_openings_device.py -m gpu -q > $O/tests.txt 2>&1; grep -v "amdgpu.ids\|RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl" $O/tests.txt | tail -4
timeout 300 python bench_tools/openings_extract_bench.py 2>&1 | grep -v amdgpu.ids > $O/openings_extract.txt; cat $O/openings_extract.txt
bash tools/run_pmc.sh extract fetch write > $O/run_pmc.log 2>&1; tail -2 $O/run_pmc.log
cp $ROOT/gpurun_out/summaries/pmc_k_merkle4_openings.* $O/ 2>/dev/null; cat $O/pmc_k_merkle4_openings.txt
