#!/bin/bash
# round 6, call 4: why does the 8-rank rehearsal take 65-97 s inside the suite and 4.5 s on its own?
set -u
O=gpurun_out/r06
mkdir -p $O
T=tests/test_bench_multiproc.py::test_bench_eight_ranks_rehearsal_of_the_driver_command
( time python -m pytest $T -m gpu -q -p no:cacheprovider ) > $O/eight_a.txt 2>&1; grep "passed\|failed\|real\|user" $O/eight_a.txt
( time python -m pytest $T -m gpu -q -s -p no:cacheprovider ) > $O/eight_b.txt 2>&1; grep "passed\|failed\|real\|user" $O/eight_b.txt
( time python -m pytest tests/test_abi.py tests/test_bench_cli.py $T -m gpu -q -p no:cacheprovider ) > $O/eight_c.txt 2>&1; grep "passed\|failed\|real\|user" $O/eight_c.txt
( time python -m pytest tests/test_bench_multiproc.py -m gpu -q --durations=6 -p no:cacheprovider ) > $O/eight_d.txt 2>&1; grep "passed\|failed\|real\|user\|s call" $O/eight_d.txt
