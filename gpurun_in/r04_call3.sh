#!/bin/bash
# round 4, call 3: forest as a secondary bench workload + Merkle2 forests, the long randomized soak over every entry point (forests and the
# sharded tree through the library's RCCL communicator included), then the suite's touched files
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
O=gpurun_out/r04c3; mkdir -p $O
timeout 900 python -m pytest tests/test_comm_forest.py tests/test_bench_multiproc.py -m gpu -q > $O/tests.txt 2>&1; tail -6 $O/tests.txt
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; python - <<'P'
import json
d=json.loads([l for l in open('gpurun_out/r04c3/bench.json') if l.startswith('{')][0])
r=d['roofline']; print('value %.4g ms %.4f frac %.4f at-clock %.4f' % (d['value'], d['ms_per_step'], r['frac'], r['frac_at_measured_clock']))
for k,w in d.get('secondary',{}).items(): print(k, '%.4g' % w['value'], round(w['ms_per_step'],3), round(w['roofline']['frac'],4), round(w['roofline']['frac_at_measured_clock'],4), w['roofline']['traffic_ratio'], w['parity_sample_ok'])
P
timeout 600 python bench_tools/soak_check.py --long 5 2>&1 | grep -v amdgpu.ids > $O/soak_long.txt; tail -5 $O/soak_long.txt
