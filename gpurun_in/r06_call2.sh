#!/bin/bash
# round 6, call 2: the changed / new GPU tests with durations; where the 8-rank rehearsal spends its 97 s
set -u
O=gpurun_out/r06
mkdir -p $O
timeout 900 python -m pytest tests/test_reference_fixtures.py tests/test_host_pipeline.py tests/test_bench_multiproc.py tests/test_round6_gpu.py -m gpu -q --durations=25 -s > $O/gputest_call2.txt 2>&1
echo "pytest rc=$?"; tail -40 $O/gputest_call2.txt
export P252_BENCH_SHARE_GPU=1 P252_BENCH_BACKEND=gloo
( time python bench.py --gpus 8 --steps 2 --warmup 1 --log2n 12 --secondary-log2n 8 > $O/eight.json 2> $O/eight.err ) 2> $O/eight.time
grep "since start\|real" $O/eight.err $O/eight.time
( time python bench.py --gpus 8 --steps 2 --warmup 1 --log2n 12 --secondary-log2n 8 > $O/eight2.json 2> $O/eight2.err ) 2> $O/eight2.time
grep "since start\|real" $O/eight2.err $O/eight2.time
( time python -c "import torch; torch.zeros(1).cuda()" ) 2>&1 | grep real
