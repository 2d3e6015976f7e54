#!/bin/bash
# round 2, GPU call 2: full GPU test-suite on the final kernels, A/B against the earlier builds, the bench lines of
# every workload, host-path rates, VALU rates (more occupancies) + clocks, rocprofv3 kernel trace and PMC passes.
cd "${GRAFT_REPO_ROOT:-.}"
ROOT=$(pwd)
O=$ROOT/gpurun_out/r02b; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest_gpu.txt
one() { # name lib workload
  P252_LIB_PATH=$2 python bench.py --workload $3 --steps 40 --warmup 10 --no-cpu-baseline > $O/bench_$1_$3.json 2> $O/bench_$1_$3.err
  python - "$O/bench_$1_$3.json" "$1" "$3" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).readline())
    print("%-8s %-16s %.4g perm/s  %.4f ms/step  launch mean %.4f min %.4f  ok=%s"%(sys.argv[2],sys.argv[3],d["value"],d["ms_per_step"],d["roofline"]["launch_ms_mean"],d["roofline"]["launch_ms_min"],d["self_consistency_ok"]))
except Exception as e:
    print(sys.argv[2],sys.argv[3],"FAILED",e)
PY
}
for rep in 1 2; do
  one new$rep "" merkle4_digests
  one w1_$rep variants/lib_w1.so merkle4_digests
  one r1_$rep variants/lib_r1.so merkle4_digests
done
for wl in tree sponge42 openings encrypt; do one new "" $wl; done
one w1 variants/lib_w1.so tree
python bench.py > $O/bench_full.json 2> $O/bench_full.err; echo "bench rc=$?"
python bench_tools/host_path_bench.py > $O/host_path.txt 2>&1; cat $O/host_path.txt | tail -8
./bench_tools/valu_rates --latency > $O/valu_rates.txt 2>&1; echo "valu_rates rc=$?"
# ---- profiler passes (each --pmc group its own run; never combined with sys/hip tracing)
cd /tmp
pass() { # tag workload counters...
  local tag=$1 wl=$2; shift 2
  rm -rf $O/pmc_${wl}_$tag
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/pmc_${wl}_$tag -o pmc -- \
      python $ROOT/bench.py --workload $wl --steps 5 --warmup 1 --no-cpu-baseline > $O/pmc_${wl}_$tag.log 2>&1
  echo "pmc $wl $tag rc=$?"
}
pass valu merkle4_digests SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
pass itype merkle4_digests SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_SALU SQ_INSTS_SMEM
pass wait merkle4_digests SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SMEM
for wl in merkle4_digests sponge42 openings encrypt; do
  pass fetch $wl FETCH_SIZE
  pass write $wl WRITE_SIZE
done
for wl in sponge42 openings encrypt; do pass valu $wl SQ_WAVES SQ_INSTS_VALU GRBM_GUI_ACTIVE; done
for wl in merkle4_digests tree; do
  rm -rf $O/ktrace_$wl
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/ktrace_$wl -o kt -- python $ROOT/bench.py --workload $wl --no-cpu-baseline > $O/ktrace_$wl.log 2>&1
  echo "ktrace $wl rc=$?"
done
# clocks of the microbenchmark itself: a short run under the GRBM counter
timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_valu_rates -o pmc -- $ROOT/bench_tools/valu_rates --clock > $O/valu_rates_clock.txt 2>&1; echo "valu clock rc=$?"
cd $ROOT
# summaries (the raw CSVs stay in gpurun_out; the summaries are what gets committed)
S=$O/summaries; mkdir -p $S
U20=1048576
sumk() { # kernel units bytes-per-unit workload
  local dirs=""; for d in $O/pmc_$4_*; do [ -f $d/pmc_counter_collection.csv ] && dirs="$dirs $d"; done
  [ -n "$dirs" ] && python tools/pmc_summary.py $1 $2 --bytes-per-unit $3 $dirs > $S/pmc_$1.txt 3> $S/pmc_$1.json
}
sumk k_merkle4 $U20 160 merkle4_digests
sumk k_sponge $((12*U20)) 125.3333 sponge42
sumk k_merkle4_path $((12*U20)) 99.3333 openings
sumk k_crypt $((2*U20)) 128 encrypt
for wl in merkle4_digests tree; do db=$(find $O/ktrace_$wl -name "*.db" | head -1); python tools/rocprof_summary.py "$db" "bench.py --workload $wl" > $S/ktrace_$wl.txt 2>&1; done
ls $S; head -30 $S/pmc_k_merkle4.txt; tail -12 $S/pmc_k_sponge.txt; tail -6 $S/pmc_k_merkle4_path.txt; tail -6 $S/pmc_k_crypt.txt
find $O -name "*.db" -size +20M -delete; find $O -name "*_agent_info.csv" -delete
du -sh $O
