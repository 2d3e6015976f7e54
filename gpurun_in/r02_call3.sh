#!/bin/bash
# round 2, GPU call 3: occupancy variants and small-launch latency (same box), host staging-lane sweep, new tests
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r02c; mkdir -p $O
one() { # name lib workload extra-args
  P252_LIB_PATH=$2 python bench.py --workload $3 --steps 40 --warmup 10 --no-cpu-baseline $4 > $O/bench_$1_$3.json 2> $O/bench_$1_$3.err
  python - "$O/bench_$1_$3.json" "$1" "$3$4" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).readline())
    print("%-8s %-28s %.4g perm/s  %.4f ms/step  launch mean %.4f min %.4f  ok=%s"%(sys.argv[2],sys.argv[3],d["value"],d["ms_per_step"],d["roofline"]["launch_ms_mean"],d["roofline"]["launch_ms_min"],d["self_consistency_ok"]))
except Exception as e:
    print(sys.argv[2],sys.argv[3],"FAILED",e)
PY
}
for rep in 1 2; do
  one w3_$rep "" merkle4_digests
  one w4_$rep variants/lib_w4.so merkle4_digests
  one w2_$rep variants/lib_w2.so merkle4_digests
done
one w3 "" tree; one w4 variants/lib_w4.so tree; one w2 variants/lib_w2.so tree
one w3 "" openings; one w4 variants/lib_w4.so openings
for v in "new:" "w1:variants/lib_w1.so" "r1:variants/lib_r1.so" "w4:variants/lib_w4.so" "w2:variants/lib_w2.so"; do
  name=${v%%:*}; lib=${v#*:}
  P252_LIB_PATH=$lib python bench.py --log2n 12 --steps 300 --warmup 30 --no-cpu-baseline > $O/small_$name.json 2>/dev/null
  python -c "
import json;d=json.loads(open('$O/small_$name.json').readline());print('%-4s 4096-digest launch: mean %.4f ms min %.4f ms'%('$name',d['roofline']['launch_ms_mean'],d['roofline']['launch_ms_min']))"
done
bash bench_tools/host_lanes_sweep.sh > $O/host_lanes.txt 2>&1; cat $O/host_lanes.txt
timeout 600 python -m pytest tests/test_gpu_fullsize.py tests/test_multi_device.py tests/test_host_pipeline.py -m gpu -x -q > $O/pytest_new.txt 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_new.txt
