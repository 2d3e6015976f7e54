#!/bin/bash
# round 5: the whole GPU suite + smoke + a 3-minute randomized soak on the final tree
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"; O=$ROOT/gpurun_out/r05final; mkdir -p "$O"
python -m pytest tests -m gpu -q > "$O/gputest.txt" 2>&1; tail -3 "$O/gputest.txt"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2 | tee "$O/smoke.txt"
timeout 330 python bench_tools/soak_check.py --long 3 2>&1 | grep -v "amdgpu.ids\|RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl" > "$O/soak_long.txt"; tail -2 "$O/soak_long.txt"
python bench.py > "$O/bench.json" 2> "$O/bench.err"; wc -c "$O/bench.json"; cp bench_detail.json "$O/"
