#!/bin/bash
# round 2, GPU call 10: the cooperative low-latency digest kernels (k_merkle4_coop<8|4>) — parity, small-batch latency, tree
cd "${GRAFT_REPO_ROOT:-.}"
ROOT=$(pwd)
O=$ROOT/gpurun_out/r02j; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_coop.py tests/test_gpu_parity.py -m gpu -x -q > $O/pytest_coop.txt 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest_coop.txt
for l in 12 13; do
  for coop in 16384 0; do
    P252_COOP_MAX_NODES=$coop python bench.py --log2n $l --steps 300 --warmup 30 --no-cpu-baseline > $O/bench_small_${l}_coop$coop.json 2>$O/err.txt || tail -3 $O/err.txt
  done
done
# 2^14 nodes: the 4-lane cooperative kernel against one wave of the one-lane kernel
for coop in 16384 8192; do P252_COOP_MAX_NODES=$coop python bench.py --log2n 14 --steps 300 --warmup 30 --no-cpu-baseline > $O/bench_small_14_coop$coop.json 2>/dev/null; done
# two and four waves per SIMD: the 2-wave build (k_merkle4_lat) against the 3-wave build (k_merkle4)
for l in 17 18; do for m in 0x16 0x2; do
  P252_LAT_WAVES=$m python bench.py --log2n $l --steps 200 --warmup 30 --no-cpu-baseline > $O/bench_w_${l}_lat$m.json 2>/dev/null
done; done
for rep in 1 2; do
for v in "16384 0x16" "8192 0x16" "16384 0x2" "0 0x2"; do set -- $v
  P252_COOP_MAX_NODES=$1 P252_LAT_WAVES=$2 python bench.py --workload tree --no-cpu-baseline > $O/bench_tree_coop$1_lat$2_$rep.json 2>/dev/null
done; done
python bench.py --no-cpu-baseline > $O/bench.json 2>/dev/null
python - <<PY
import json,glob,os
for f in sorted(glob.glob("$O/bench*.json")):
    try:
        d=json.loads(open(f).readline())
        print("%-34s %.4g perm/s  %.4f ms/step  launch mean %.4f min %s"%(os.path.basename(f),d["value"],d["ms_per_step"],d["roofline"]["launch_ms_mean"],d["roofline"].get("launch_ms_min")))
    except Exception as e: print(f,"FAILED",e)
PY
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/ktrace_tree -o kt -- python $ROOT/bench.py --workload tree --no-cpu-baseline > $O/ktrace_tree.log 2>&1; echo "ktrace rc=$?"
cd $ROOT
python - <<PY
import sqlite3,glob
db=glob.glob("$O/ktrace_tree/**/*.db",recursive=True)[0]
cur=sqlite3.connect(db).cursor()
tabs=[r[0] for r in cur.execute("select name from sqlite_master where type='view' or type='table'")]
print("%-16s %9s %6s %11s %11s"%("kernel","grid","calls","avg_us","min_us"))
for name,g,n,a,m in cur.execute("select name, grid_x, count(*), avg(duration), min(duration) from kernels where name like '%k_merkle4%' group by name, grid_x order by grid_x desc"):
    print("%-16s %9d %6d %11.1f %11.1f"%(name.split("(")[0].split("::")[-1],g,n,a/1e3,m/1e3))
PY
find $O -name "*.db" -size +20M -delete; find $O -name "*_agent_info.csv" -delete; du -sh $O
