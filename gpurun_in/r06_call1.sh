#!/bin/bash
# round 6, call 1: the GPU suite with per-test durations, the default bench line, the extraction kernel's counter passes
set -u
O=gpurun_out/r06
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --durations=40 > $O/gputest.txt 2>&1
echo "pytest rc=$?"; tail -5 $O/gputest.txt
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; head -c 600 $O/bench.json
bash tools/run_pmc.sh extract valu fetch write ktrace > $O/run_pmc_extract.log 2>&1
cp gpurun_out/summaries/* $O/ 2>/dev/null
ls $O
