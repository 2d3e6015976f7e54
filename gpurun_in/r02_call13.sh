#!/bin/bash
# round 2, GPU call 13: lane-group kernels for the other entry points (permute / sponge / openings) — parity and latency
cd "${GRAFT_REPO_ROOT:-.}"
ROOT=$(pwd)
O=$ROOT/gpurun_out/r02m; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
for wl in sponge42 openings; do for coop in 16384 0; do
  P252_COOP_MAX_NODES=$coop python bench.py --workload $wl --log2n 12 --steps 40 --warmup 5 --no-cpu-baseline > $O/bench_${wl}_4096_coop$coop.json 2>$O/err.txt || tail -3 $O/err.txt
done; done
for coop in 16384 0; do P252_COOP_MAX_NODES=$coop python bench.py --log2n 12 --steps 300 --warmup 30 --no-cpu-baseline > $O/bench_small4096_coop$coop.json 2>/dev/null; done
python bench.py --workload tree --no-cpu-baseline > $O/bench_tree.json 2>/dev/null
python bench.py --no-cpu-baseline > $O/bench.json 2>/dev/null
python - <<PY
import json,glob,os
for f in sorted(glob.glob("$O/bench*.json")):
    try:
        d=json.loads(open(f).readline())
        print("%-38s %.4g perm/s  %.4f ms/step  launch mean %.4f min %.4f"%(os.path.basename(f),d["value"],d["ms_per_step"],d["roofline"]["launch_ms_mean"],d["roofline"]["launch_ms_min"]))
    except Exception as e: print(f,"FAILED",e)
PY
python bench_tools/soak_check.py > $O/soak.txt 2>&1; tail -4 $O/soak.txt
du -sh $O
