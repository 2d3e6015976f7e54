#!/bin/bash
# round 2, GPU call 1: VALU issue rates after clock ramp, A/B of the wide Montgomery step against the round-1 library,
# the GPU test-suite on the new library, one full default bench line.
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r02a; mkdir -p $O
rocm-smi --showclocks > $O/clocks_idle.txt 2>&1
./bench_tools/valu_rates --latency > $O/valu_rates.txt 2>&1; echo "valu_rates rc=$?"
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
  one r1_$rep variants/lib_r1.so merkle4_digests
done
one new "" tree; one r1 variants/lib_r1.so tree
one new "" sponge42; one r1 variants/lib_r1.so sponge42
one new "" openings
python bench.py --log2n 12 --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_small4096.json 2>/dev/null; python -c "
import json;d=json.loads(open('$O/bench_small4096.json').readline());print('4096-digest launch: mean %.4f ms min %.4f ms'%(d['roofline']['launch_ms_mean'],d['roofline']['launch_ms_min']))"
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_gpu.txt
python bench.py > $O/bench_full.json 2> $O/bench_full.err; echo "bench rc=$?"; head -c 600 $O/bench_full.json; echo
head -40 $O/valu_rates.txt
