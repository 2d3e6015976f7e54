#!/bin/bash
# round 2, GPU call 32: HBM traffic of the two format-conversion kernels (FETCH_SIZE / WRITE_SIZE passes)
cd "${GRAFT_REPO_ROOT:-.}"
ROOT=$(pwd)
O=$ROOT/gpurun_out/r02v; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/pmc_$c
  timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -o pmc -- python $ROOT/bench_tools/byte_format_bench.py > $O/pmc_$c.log 2>&1
  echo "pmc $c rc=$?"
done
cd $ROOT
python tools/pmc_summary.py "k_to_canonical" 67108864 --bytes-per-unit 64 $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE > $O/pmc_k_to_canonical.txt 2>&1
python tools/pmc_summary.py "k_from_canonical" 67108864 --bytes-per-unit 64 $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE > $O/pmc_k_from_canonical.txt 2>&1
cat $O/pmc_k_to_canonical.txt $O/pmc_k_from_canonical.txt
find $O -name "*_agent_info.csv" -delete
