#!/bin/bash
# round 5: the whole GPU suite + smoke on the final tree
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"; O=$ROOT/gpurun_out/r05final; mkdir -p "$O"
python -m pytest tests -m gpu -q > "$O/gputest.txt" 2>&1; tail -3 "$O/gputest.txt"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2 | tee "$O/smoke.txt"
