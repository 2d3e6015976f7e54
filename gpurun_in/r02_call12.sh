#!/bin/bash
# round 2, GPU call 12: the measurement set on the library with the cooperative low-latency kernels
# (tests, bench lines of every workload, host path, kernel traces, PMC passes incl. the cooperative kernel)
cd "${GRAFT_REPO_ROOT:-.}"
ROOT=$(pwd)
O=$ROOT/gpurun_out/r02l; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
for wl in tree sponge42 openings encrypt; do python bench.py --workload $wl --no-cpu-baseline > $O/bench_$wl.json 2> $O/bench_$wl.err; echo "bench $wl rc=$?"; done
python bench.py --log2n 12 --steps 300 --warmup 30 --no-cpu-baseline > $O/bench_small4096.json 2>/dev/null
python bench.py --log2n 14 --steps 300 --warmup 30 --no-cpu-baseline > $O/bench_small16384.json 2>/dev/null
python bench.py --log2n 24 --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_2pow24_digests.json 2>/dev/null
for wl in sponge42 openings encrypt; do python bench.py --workload $wl --log2n 12 --steps 60 --warmup 6 --no-cpu-baseline > $O/bench_${wl}_4096.json 2>/dev/null; done
python - <<PY
import json
for f in ("bench","bench_tree","bench_sponge42","bench_openings","bench_encrypt","bench_small4096","bench_small16384","bench_2pow24_digests","bench_sponge42_4096","bench_openings_4096","bench_encrypt_4096"):
    try:
        d=json.loads(open("$O/%s.json"%f).readline())
        print("%-22s %.4g perm/s  %.4f ms/step  launch mean %.4f  kernel %s executed.frac %s"%(f,d["value"],d["ms_per_step"],d["roofline"]["launch_ms_mean"],d["roofline"]["kernel"],(d["roofline"]["executed"] or {}).get("frac")))
    except Exception as e: print(f,"FAILED",e)
PY
python bench_tools/host_path_bench.py > $O/host_path.txt 2>&1; tail -6 $O/host_path.txt
python bench_tools/host_tree_bench.py > $O/host_tree.txt 2>&1; tail -4 $O/host_tree.txt
cd /tmp
pass() { # tag workload extra-args counters...
  local tag=$1 wl=$2 extra=$3; shift 3
  rm -rf $O/pmc_${wl}_$tag
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/pmc_${wl}_$tag -o pmc -- \
      python $ROOT/bench.py $extra --steps 5 --warmup 1 --no-cpu-baseline > $O/pmc_${wl}_$tag.log 2>&1
  echo "pmc $wl $tag rc=$?"
}
pass valu merkle4_digests "--workload merkle4_digests" SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
pass fetch merkle4_digests "--workload merkle4_digests" FETCH_SIZE
pass write merkle4_digests "--workload merkle4_digests" WRITE_SIZE
pass valu small4096 "--log2n 12" SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
pass itype small4096 "--log2n 12" SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_SALU SQ_INSTS_SMEM
pass lds small4096 "--log2n 12" SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY
pass valu small16384 "--log2n 14" SQ_WAVES SQ_INSTS_VALU GRBM_GUI_ACTIVE
pass valu tree "--workload tree" SQ_WAVES SQ_INSTS_VALU GRBM_GUI_ACTIVE
for wl in merkle4_digests tree; do
  rm -rf $O/ktrace_$wl
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/ktrace_$wl -o kt -- python $ROOT/bench.py --workload $wl --no-cpu-baseline > $O/ktrace_$wl.log 2>&1
  echo "ktrace $wl rc=$?"
done
cd $ROOT
S=$O/summaries; mkdir -p $S
dirs=""; for d in $O/pmc_merkle4_digests_*; do [ -f $d/pmc_counter_collection.csv ] && dirs="$dirs $d"; done
python tools/pmc_summary.py k_merkle4 1048576 --bytes-per-unit 160 $dirs > $S/pmc_k_merkle4.txt 3> $S/pmc_k_merkle4.json
dirs=""; for d in $O/pmc_small4096_*; do [ -f $d/pmc_counter_collection.csv ] && dirs="$dirs $d"; done
python tools/pmc_summary.py "k_merkle4_coop<8>" 4096 --bytes-per-unit 160 $dirs > $S/pmc_k_merkle4_coop8.txt 3> $S/pmc_k_merkle4_coop8.json
python tools/pmc_summary.py "k_merkle4_coop<4>" 16384 --bytes-per-unit 160 $O/pmc_small16384_valu > $S/pmc_k_merkle4_coop4.txt 3> $S/pmc_k_merkle4_coop4.json
for wl in merkle4_digests tree; do db=$(find $O/ktrace_$wl -name "*.db" | head -1); python tools/rocprof_summary.py "$db" "bench.py --workload $wl" > $S/ktrace_$wl.txt 2>&1; done
python - <<PY > $S/tree_levels.txt
import csv,collections,sqlite3,glob
rows=list(csv.DictReader(open("$O/pmc_tree_valu/pmc_counter_collection.csv")))
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    if "k_merkle4" in r["Kernel_Name"]:
        key=(r["Kernel_Name"].split("(")[0].split("::")[-1],int(r["Grid_Size"]))
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        agg[key]["dur"].append((float(r["End_Timestamp"])-float(r["Start_Timestamp"]))/1e3)
print("# 2^24-leaf tree under rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU GRBM_GUI_ACTIVE (dispatches serialised by the counter collection): per kernel and grid")
print("%-20s %9s %6s %11s %13s %10s %14s"%("kernel","lanes","calls","median_us","cycles","clock_GHz","VALU/wave"))
for (k,g),c in sorted(agg.items(), key=lambda kv:-kv[0][1]):
    d=sorted(c["dur"]); med=d[len(d)//2]; cyc=sum(c["GRBM_GUI_ACTIVE"])/len(c["GRBM_GUI_ACTIVE"])/8
    print("%-20s %9d %6d %11.1f %13.4g %10.3f %14.0f"%(k,g,len(d)//3,med,cyc,cyc/med/1e3,sum(c["SQ_INSTS_VALU"])/max(1,sum(c["SQ_WAVES"]))))
db=glob.glob("$O/ktrace_tree/**/*.db",recursive=True)[0]
cur=sqlite3.connect(db).cursor()
print("\n# the same build under --kernel-trace only (back-to-back launches): per kernel and grid")
print("%-20s %9s %6s %11s %11s"%("kernel","lanes","calls","avg_us","min_us"))
tot=0; ncalls=None
for name,g,n,a,m in cur.execute("select name, grid_x, count(*), avg(duration), min(duration) from kernels where name like '%k_merkle4%' group by name, grid_x order by grid_x desc, avg(duration) desc"):
    print("%-20s %9d %6d %11.1f %11.1f"%(name.split("(")[0].split("::")[-1],g,n,a/1e3,m/1e3))
    if ncalls is None: ncalls=n
    if g>=65536: tot+=a/1e3*n/ncalls
print("sum over the 12 levels of one build (avg): %.1f us"%tot)
PY
cat $S/tree_levels.txt; head -40 $S/pmc_k_merkle4_coop8.txt; tail -8 $S/pmc_k_merkle4_coop4.txt; tail -12 $S/pmc_k_merkle4.txt
python bench_tools/soak_check.py > $O/soak.txt 2>&1; tail -3 $O/soak.txt
find $O -name "*.db" -size +20M -delete; find $O -name "*_agent_info.csv" -delete; du -sh $O
