#!/bin/bash
# experiment: does keeping the chip loaded through the narrow levels avoid the clock dip of the next wide level?
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r02e; mkdir -p $O
for pad in 0 65536 196608 0 65536 196608 32768 16384; do
  P252_TREE_PAD_LANES=$pad python bench.py --workload tree --steps 40 --warmup 10 --no-cpu-baseline > $O/tree_pad_$pad.json 2>$O/tree_pad_$pad.err
  python -c "
import json;d=json.loads(open('$O/tree_pad_$pad.json').readline());print('pad %7d: %.4g perm/s  %.4f ms/step  ok=%s'%($pad,d['value'],d['ms_per_step'],d['self_consistency_ok']))"
done
cd /tmp; export TMPDIR=/tmp
for pad in 0 65536; do
rm -rf $OLDPWD/$O/kt_$pad
P252_TREE_PAD_LANES=$pad timeout 300 rocprofv3 --kernel-trace --stats -d $OLDPWD/$O/kt_$pad -o kt -- python $OLDPWD/bench.py --workload tree --no-cpu-baseline > $OLDPWD/$O/kt_$pad.log 2>&1
python - <<PY
import sqlite3,glob
db=glob.glob("$OLDPWD/$O/kt_$pad/**/*.db",recursive=True)[0]
cur=sqlite3.connect(db).cursor()
print("pad $pad: per level (kernel-trace)")
for name,g,n,a,m in cur.execute("select name, grid_x, count(*), avg(duration), min(duration) from kernels where name like '%k_merkle4%' group by name, grid_x order by grid_x desc"):
    print("  %-14s %9d %6d %11.1f %11.1f"%(name.split("(")[0].split("::")[-1],g,n,a/1e3,m/1e3))
PY
done
