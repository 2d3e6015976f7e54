#!/bin/bash
# round 4, call 2: whole GPU suite on the tree with the rccl preload fix, S-box variant A/B, host path twice on this box, default bench line + its kernel trace + counter passes
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
ROOT=$(pwd)
O=$ROOT/gpurun_out/r04c2; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/gputest.txt 2>&1; tail -4 $O/gputest.txt
timeout 200 ./bench_tools/sbox_variants > $O/sbox_variants.txt 2>&1; cat $O/sbox_variants.txt
timeout 300 python bench_tools/host_path_bench.py 2>&1 | grep -v amdgpu.ids > $O/host_path_a.txt; cat $O/host_path_a.txt
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.json
rm -rf $ROOT/gpurun_out/ktrace_default
(cd /tmp && TMPDIR=/tmp timeout 600 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/ktrace_default -o kt -- python $ROOT/bench.py --no-cpu-baseline > $O/bench_under_ktrace.json 2> $O/bench_under_ktrace.err)
db=$(find $ROOT/gpurun_out/ktrace_default -name "*.db" | head -1)
[ -n "$db" ] && python tools/rocprof_summary.py "$db" "python bench.py --no-cpu-baseline (the default line: configs[1], then the secondary workloads)" 38 > $O/bench_kernel_trace.txt
head -30 $O/bench_kernel_trace.txt
bash tools/run_pmc.sh merkle4_digests valu fetch write > $O/run_pmc.log 2>&1
bash tools/run_pmc.sh tree fetch write >> $O/run_pmc.log 2>&1
cp $ROOT/gpurun_out/summaries/* $O/ 2>/dev/null
timeout 300 python bench_tools/host_path_bench.py 2>&1 | grep -v amdgpu.ids > $O/host_path_b.txt; cat $O/host_path_b.txt
ls $O
