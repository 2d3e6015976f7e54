#!/bin/bash
# Everything the round's profiles/ files are made from, in one gpurun call (from the repository root on the GPU box):
#   gpurun --timeout 2400 -- 'bash tools/collect_round.sh'
# -> gpurun_out/final/: GPU test log, the default bench line, its rocprofv3 kernel trace, the counter passes of every
# kernel bench.py quotes (their JSON records the SHA-256 of the kernel sources), the other workloads' bench lines, the
# host-path sweeps.  Copy what is to be judged into profiles/ (named per round) and commit.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
O=$ROOT/gpurun_out/final
mkdir -p "$O"
python -m pytest tests -m gpu -q > "$O/gputest.txt" 2>&1
tail -2 "$O/gputest.txt"
python bench.py > "$O/bench.json" 2> "$O/bench.err"
cp "$ROOT/bench_detail.json" "$O/bench_detail.json" 2>/dev/null   # the full record behind the printed line (bench.py --detail-file)
wc -c "$O/bench.json"
rm -rf "$ROOT/gpurun_out/ktrace_default"
(cd /tmp && TMPDIR=/tmp timeout 600 rocprofv3 --kernel-trace --stats -d "$ROOT/gpurun_out/ktrace_default" -o kt -- python "$ROOT/bench.py" --no-cpu-baseline > "$O/bench_under_ktrace.json" 2> "$O/bench_under_ktrace.err")
db=$(find "$ROOT/gpurun_out/ktrace_default" -name "*.db" | head -1)
[ -n "$db" ] && python tools/rocprof_summary.py "$db" "python bench.py --no-cpu-baseline (the default line: configs[1], then the secondary workloads)" 38 50 > "$O/bench_kernel_trace.txt"
bash tools/run_pmc.sh merkle4_digests valu stall ifetch icache dcache fetch write > "$O/run_pmc.log" 2>&1
bash tools/run_pmc.sh sponge42 valu fetch write >> "$O/run_pmc.log" 2>&1
bash tools/run_pmc.sh tree fetch write >> "$O/run_pmc.log" 2>&1
bash tools/run_pmc.sh openings valu fetch write >> "$O/run_pmc.log" 2>&1
bash tools/run_pmc.sh encrypt valu fetch write >> "$O/run_pmc.log" 2>&1
bash tools/run_pmc.sh extract valu wait tcp fetch write >> "$O/run_pmc.log" 2>&1   # the extraction of openings (csrc/openings.hip): what binds it
LOG2N=12 bash tools/run_pmc.sh merkle4_digests valu >> "$O/run_pmc.log" 2>&1
cp "$ROOT"/gpurun_out/summaries/* "$O/" 2>/dev/null
for wl in tree forest sponge42 openings encrypt extract; do python bench.py --workload $wl --no-cpu-baseline > "$O/bench_$wl.json" 2>/dev/null; done
python bench_tools/forest_bench.py --check 2>&1 | grep -v amdgpu.ids > "$O/forest.txt"
python bench_tools/openings_extract_bench.py 2>&1 | grep -v amdgpu.ids > "$O/openings_extract.txt"
python bench.py --log2n 12 --no-secondary --no-cpu-baseline > "$O/bench_small4096.json" 2>/dev/null
python bench.py --log2n 14 --no-secondary --no-cpu-baseline > "$O/bench_small16384.json" 2>/dev/null
python bench.py --log2n 24 --no-secondary --no-cpu-baseline > "$O/bench_2pow24_digests.json" 2>/dev/null
for wl in sponge42 openings encrypt; do python bench.py --workload $wl --log2n 12 --no-cpu-baseline > "$O/bench_${wl}_4096.json" 2>/dev/null; done
bash bench_tools/host_multi_sweep.sh 2>&1 | grep -v amdgpu.ids > "$O/host_path_multi.txt"
python bench_tools/host_path_bench.py 2>&1 | grep -v amdgpu.ids > "$O/host_path.txt"
python bench_tools/clock_probe_check.py 2>&1 | grep -v "amdgpu.ids\|sysfs" > "$O/clock_probe_check.txt"
./bench_tools/copy_rate > "$O/copy_rate.txt" 2>&1   # the HBM yardstick on this box (16-byte copy, hipMemcpy D2D, no-arithmetic gather)
timeout 400 python bench_tools/soak_check.py --long 4 2>&1 | grep -v "amdgpu.ids\|RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl" > "$O/soak_long.txt"
tail -3 "$O/soak_long.txt"
ls "$O"
