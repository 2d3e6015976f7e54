#!/bin/bash
# round 5, first GPU call: the tests touched this round, the default bench line (must parse: < 6,000 bytes), and the counter passes
# the default line quotes (the kernel sources changed: profiles/r04_pmc_*.json are stale)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
O=$ROOT/gpurun_out/r05c1
mkdir -p "$O"
python -m pytest tests/test_round5_gpu.py tests/test_bench_multiproc.py tests/test_comm_forest.py tests/test_comm_mock_ranks.py tests/test_next_rows.py \
    tests/test_c_abi.py tests/test_cpp_host_api.py tests/test_encryption.py tests/test_multi_device.py -m gpu -q -x > "$O/gputest_subset.txt" 2>&1
tail -15 "$O/gputest_subset.txt"
python bench.py > "$O/bench.json" 2> "$O/bench.err"
wc -c "$O/bench.json"; grep "bench.py primary" "$O/bench.err"
cp bench_detail.json "$O/bench_detail.json" 2>/dev/null
bash tools/run_pmc.sh merkle4_digests valu fetch write > "$O/run_pmc.log" 2>&1
bash tools/run_pmc.sh sponge42 valu fetch write >> "$O/run_pmc.log" 2>&1
bash tools/run_pmc.sh tree fetch write >> "$O/run_pmc.log" 2>&1
bash tools/run_pmc.sh extract fetch write >> "$O/run_pmc.log" 2>&1
cp "$ROOT"/gpurun_out/summaries/* "$O/" 2>/dev/null
ls "$O"
