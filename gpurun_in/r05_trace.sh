#!/bin/bash
# round 5: the default line and the SAME command under rocprofv3 --kernel-trace --stats on one more box (pairs a line with its trace)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"; O=$ROOT/gpurun_out/r05trace; mkdir -p "$O"
python bench.py > "$O/bench.json" 2> "$O/bench.err"; cp bench_detail.json "$O/bench_detail.json"; wc -c "$O/bench.json"
rm -rf "$ROOT/gpurun_out/ktrace_r05b"
(cd /tmp && TMPDIR=/tmp timeout 600 rocprofv3 --kernel-trace --stats -d "$ROOT/gpurun_out/ktrace_r05b" -o kt -- python "$ROOT/bench.py" --no-cpu-baseline > "$O/bench_under_ktrace.json" 2> "$O/bench_under_ktrace.err")
db=$(find "$ROOT/gpurun_out/ktrace_r05b" -name "*.db" | head -1)
[ -n "$db" ] && python tools/rocprof_summary.py "$db" "python bench.py --no-cpu-baseline (the default line: configs[1], then the secondary workloads)" 38 50 > "$O/bench_kernel_trace.txt"
grep "bench.py primary" "$O/bench.err"; sed -n '/PRIMARY/,/^$/p' "$O/bench_kernel_trace.txt"
