#!/bin/bash
# round 2, GPU call 14: k_crypt_coop — parity (both kernel families) and latency
cd "${GRAFT_REPO_ROOT:-.}"
ROOT=$(pwd)
O=$ROOT/gpurun_out/r02n; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_coop.py tests/test_encryption.py tests/test_host_pipeline.py -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_gpu.txt
for coop in 16384 0; do
  P252_COOP_MAX_NODES=$coop python bench.py --workload encrypt --log2n 12 --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_encrypt_4096_coop$coop.json 2>$O/err.txt || tail -3 $O/err.txt
done
python - <<PY
import json,glob,os
for f in sorted(glob.glob("$O/bench*.json")):
    try:
        d=json.loads(open(f).readline())
        print("%-38s %.4g perm/s  %.4f ms/step  launch mean %.4f min %.4f"%(os.path.basename(f),d["value"],d["ms_per_step"],d["roofline"]["launch_ms_mean"],d["roofline"]["launch_ms_min"]))
    except Exception as e: print(f,"FAILED",e)
PY
