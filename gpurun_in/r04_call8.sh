#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
ROOT=$(pwd)
O=$ROOT/gpurun_out/r04c8; mkdir -p $O
timeout 900 python -m pytest tests/test_bench_multiproc.py -m gpu -q > $O/tests.txt 2>&1; grep -v "amdgpu.ids\|RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl" $O/tests.txt | tail -6
bash tools/run_pmc.sh extract fetch write > $O/run_pmc.log 2>&1; tail -3 $O/run_pmc.log
cp $ROOT/gpurun_out/summaries/pmc_k_merkle4_openings.* $O/ 2>/dev/null; cat $O/pmc_k_merkle4_openings.txt
timeout 300 python bench.py --workload extract --no-cpu-baseline > $O/bench_extract.json 2>/dev/null; python -c "
import json; d=json.loads([l for l in open('$O/bench_extract.json') if l.startswith('{')][0]); print(d['value'], d['unit'], d['ms_per_step'], {k:v for k,v in d['roofline'].items() if k!='note'})"
