#!/bin/bash
# round 6, call 9: the whole GPU suite again (after the shared-GPU queue fix), then the headline's evidence for this round:
# the default bench line, the same command under rocprofv3 --kernel-trace --stats, the counter passes of k_merkle4
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/r06
mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=40 > $O/gputest_call9.txt 2>&1 ) 2> $O/gputest_call9.time
echo "pytest rc=$?"; tail -3 $O/gputest_call9.txt; cat $O/gputest_call9.time
python bench.py > $O/bench_final.json 2> $O/bench_final.err; echo "bench rc=$?"; wc -c $O/bench_final.json
cp bench_detail.json $O/bench_final_detail.json
rm -rf $ROOT/gpurun_out/ktrace_default
(cd /tmp && TMPDIR=/tmp timeout 600 rocprofv3 --kernel-trace --stats -d "$ROOT/gpurun_out/ktrace_default" -o kt -- python "$ROOT/bench.py" --no-cpu-baseline > "$O/bench_under_ktrace.json" 2> "$O/bench_under_ktrace.err")
db=$(find "$ROOT/gpurun_out/ktrace_default" -name "*.db" | head -1)
[ -n "$db" ] && python tools/rocprof_summary.py "$db" "python bench.py --no-cpu-baseline (the default line: configs[1], then the secondary workloads)" 38 50 > "$O/bench_kernel_trace.txt"
head -12 $O/bench_kernel_trace.txt
bash tools/run_pmc.sh merkle4_digests valu fetch write > $O/run_pmc_digests.log 2>&1
bash tools/run_pmc.sh tree fetch write >> $O/run_pmc_digests.log 2>&1
cp $ROOT/gpurun_out/summaries/* $O/ 2>/dev/null
ls $O
