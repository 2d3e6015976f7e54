#!/bin/bash
# round 4, final set: whole GPU suite, default bench line, the same under kernel trace, long soak — all on the final library
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
ROOT=$(pwd)
O=$ROOT/gpurun_out/r04c10; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/gputest.txt 2>&1; grep -v "amdgpu.ids\|RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl" $O/gputest.txt | tail -6
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; python - <<'P'
import json
d=json.loads([l for l in open('gpurun_out/r04c10/bench.json') if l.startswith('{')][0])
r=d['roofline']; print('value %.4g ms %.4f frac %.4f at-clock %.4f clock %.3f' % (d['value'], d['ms_per_step'], r['frac'], r['frac_at_measured_clock'], r['clock_ghz_measured']))
for k,w in d.get('secondary',{}).items(): print(k, '%.4g' % w['value'], round(w['ms_per_step'],3), round(w['roofline']['frac'],4), round(w['roofline']['frac_at_measured_clock'],4), w['roofline']['traffic_ratio'], w['parity_sample_ok'])
P
rm -rf $ROOT/gpurun_out/ktrace_default
(cd /tmp && TMPDIR=/tmp timeout 600 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/ktrace_default -o kt -- python $ROOT/bench.py --no-cpu-baseline > $O/bench_under_ktrace.json 2> $O/bench_under_ktrace.err)
db=$(find $ROOT/gpurun_out/ktrace_default -name "*.db" | head -1)
[ -n "$db" ] && python tools/rocprof_summary.py "$db" "python bench.py --no-cpu-baseline (the default line: configs[1], then the secondary workloads)" 38 > $O/bench_kernel_trace.txt
head -12 $O/bench_kernel_trace.txt
