#!/bin/bash
# round 5, second GPU call: the ONE measurement that settles the extraction kernel — a hand-written copy kernel
# on the same box as the yardstick, round 4's kernel against the FAST variant (P252_OPENINGS_FAST=0 / default), and the counter
# passes that say what binds it (VALU instructions, wait cycles, TCP->TCC request counts)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
O=$ROOT/gpurun_out/r05c2
mkdir -p "$O"
python -m pytest tests/test_openings_device.py tests/test_next_rows.py -m gpu -q -x > "$O/gputest_openings.txt" 2>&1
tail -3 "$O/gputest_openings.txt"
./bench_tools/copy_rate > "$O/copy_rate.txt" 2>&1
cat "$O/copy_rate.txt"
P252_OPENINGS_FAST=0 python bench_tools/openings_extract_bench.py 2>&1 | grep -v amdgpu.ids > "$O/extract_r04_kernel.txt"
python bench_tools/openings_extract_bench.py 2>&1 | grep -v amdgpu.ids > "$O/extract_fast_kernel.txt"
cat "$O/extract_r04_kernel.txt" "$O/extract_fast_kernel.txt"
P252_OPENINGS_FAST=0 bash tools/run_pmc.sh extract valu wait tcp fetch write > "$O/run_pmc_r04.log" 2>&1
cp "$ROOT/gpurun_out/summaries/pmc_k_merkle4_openings.txt" "$O/pmc_k_merkle4_openings_r04_kernel.txt"
rm -rf "$ROOT"/gpurun_out/pmc_extract_*
bash tools/run_pmc.sh extract valu wait tcp fetch write > "$O/run_pmc_fast.log" 2>&1
cp "$ROOT/gpurun_out/summaries/pmc_k_merkle4_openings.txt" "$O/pmc_k_merkle4_openings.txt"
cp "$ROOT/gpurun_out/summaries/pmc_k_merkle4_openings.json" "$O/pmc_k_merkle4_openings.json"
python bench.py --workload extract --no-cpu-baseline > "$O/bench_extract.json" 2>/dev/null
P252_OPENINGS_FAST=0 python bench.py --workload extract --no-cpu-baseline > "$O/bench_extract_r04_kernel.json" 2>/dev/null
cat "$O/pmc_k_merkle4_openings_r04_kernel.txt" "$O/pmc_k_merkle4_openings.txt"
