#!/bin/bash
# one more box for the round's box-to-box table (profiles/r05_bench_boxes.txt): the default line, nothing else
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"; O=$ROOT/gpurun_out/r05box_$1; mkdir -p "$O"
python bench.py > "$O/bench.json" 2> "$O/bench.err"; cp bench_detail.json "$O/"; wc -c "$O/bench.json"; grep "bench.py primary" "$O/bench.err"
