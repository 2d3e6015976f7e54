#!/bin/bash
# round 6, call 10: the GPU suite on the final library, smoke(), the differential soak with random trims
set -u
O=gpurun_out/r06
mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=15 > $O/gputest_call10.txt 2>&1 ) 2> $O/gputest_call10.time
echo "pytest rc=$?"; tail -4 $O/gputest_call10.txt | head -3; grep real $O/gputest_call10.time
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -4
timeout 500 python bench_tools/soak_check.py --long 3 2>&1 | grep -v "amdgpu.ids\|RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl" > $O/soak_long.txt; tail -4 $O/soak_long.txt
