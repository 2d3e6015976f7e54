#!/bin/bash
# round 2, GPU call 6: the shipped state once more — full GPU suite (incl. the padded-levels test), smoke, bench lines
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r02f; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
for wl in tree sponge42 openings encrypt; do python bench.py --workload $wl --no-cpu-baseline > $O/bench_$wl.json 2> $O/bench_$wl.err; done
P252_TREE_PAD_LANES=0 python bench.py --workload tree --no-cpu-baseline > $O/bench_tree_nopad.json 2>/dev/null
python - <<PY
import json
for f in ("bench","bench_tree","bench_tree_nopad","bench_sponge42","bench_openings","bench_encrypt"):
    try:
        d=json.loads(open("$O/%s.json"%f).readline())
        print("%-22s %.4g perm/s  %.4f ms/step  launch mean %.4f  executed.frac %s"%(f,d["value"],d["ms_per_step"],d["roofline"]["launch_ms_mean"],(d["roofline"]["executed"] or {}).get("frac")))
    except Exception as e: print(f,"FAILED",e)
PY
