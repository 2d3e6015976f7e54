#!/bin/bash
# round 6, call 3: the whole GPU suite after the trim, with durations (< 300 s)
set -u
O=gpurun_out/r06
mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=40 > $O/gputest_call3.txt 2>&1 ) 2> $O/gputest_call3.time
echo "pytest rc=$?"; tail -60 $O/gputest_call3.txt; cat $O/gputest_call3.time
