#!/bin/bash
# round 4, call 1: the new code paths first (RCCL in the library, forest, flattened roofline, 8-rank rehearsal), then the whole suite, the default bench line
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
O=gpurun_out/r04c1; mkdir -p $O
timeout 900 python -m pytest tests/test_comm_forest.py tests/test_c_abi.py tests/test_cpp_host_api.py -m gpu -x -q > $O/new_tests.txt 2>&1; tail -15 $O/new_tests.txt
timeout 1500 python -m pytest tests/test_bench_multiproc.py tests/test_kernel_resources.py tests/test_multi_device.py -m gpu -q > $O/bench_tests.txt 2>&1; tail -15 $O/bench_tests.txt
timeout 300 python bench_tools/forest_bench.py --check > $O/forest.txt 2>&1; cat $O/forest.txt | grep -v amdgpu.ids
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; python - <<'P'
import json
d=json.loads([l for l in open('gpurun_out/r04c1/bench.json') if l.startswith('{')][0])
r=d['roofline']; print('value %.4g ms %.4f' % (d['value'], d['ms_per_step'])); print({k:v for k,v in r.items() if not isinstance(v,(dict,str))})
for k,w in d.get('secondary',{}).items(): print(k, '%.4g' % w['value'], w['ms_per_step'], w['roofline']['frac'], w['roofline']['frac_at_measured_clock'], w['roofline']['traffic_ratio'])
print({k:v for k,v in d['cpu_baseline'].items() if k not in ('sample','note')})
P
timeout 1200 python -m pytest tests -m gpu -q -x --deselect tests/test_bench_multiproc.py --deselect tests/test_comm_forest.py > $O/gputest_rest.txt 2>&1; tail -5 $O/gputest_rest.txt
