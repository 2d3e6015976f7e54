#!/bin/bash
# round 2, final refresh of the committed measurements on the shipped library
cd "${GRAFT_REPO_ROOT:-.}"
ROOT=$(pwd); O=$ROOT/gpurun_out; mkdir -p $O/r02g
timeout 900 python -m pytest tests -m gpu -q > $O/r02g/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/r02g/pytest_gpu.txt
python bench.py > $O/r02g/bench.json 2> $O/r02g/bench.err; echo "bench rc=$?"
for wl in tree sponge42 openings encrypt; do python bench.py --workload $wl --no-cpu-baseline > $O/r02g/bench_$wl.json 2> $O/r02g/bench_$wl.err; done
python bench.py --log2n 12 --steps 300 --warmup 30 --no-cpu-baseline > $O/r02g/bench_small4096.json 2>/dev/null
python - <<PY
import json
for f in ("bench","bench_tree","bench_sponge42","bench_openings","bench_encrypt","bench_small4096"):
    try:
        d=json.loads(open("$O/r02g/%s.json"%f).readline())
        print("%-22s %.4g perm/s  %.4f ms/step  launch mean %.4f  executed.frac %s"%(f,d["value"],d["ms_per_step"],d["roofline"]["launch_ms_mean"],(d["roofline"]["executed"] or {}).get("frac")))
    except Exception as e: print(f,"FAILED",e)
PY
bash tools/run_pmc.sh merkle4_digests valu itype wait fetch write ktrace > $O/r02g/pmc_m4.log 2>&1
bash tools/run_pmc.sh sponge42 valu fetch write > $O/r02g/pmc_sp.log 2>&1
bash tools/run_pmc.sh tree ktrace > $O/r02g/pmc_tree.log 2>&1
tail -3 $O/r02g/pmc_m4.log; tail -14 $O/summaries/pmc_k_sponge.txt; tail -8 $O/summaries/pmc_k_merkle4.txt
find $O -name "*.db" -size +20M -delete; find $O -name "*_agent_info.csv" -delete
