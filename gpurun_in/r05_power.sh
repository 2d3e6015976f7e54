#!/bin/bash
# round 5: what sets the shader clock under the digest kernel — power / thermal state sampled by rocm-smi while bench.py's primary runs in a loop
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"; O=gpurun_out/r05power; mkdir -p $O
rocm-smi --showpower --showclocks --showtemp --showperflevel --showmaxpower 2>&1 | grep -v "^$" > $O/idle.txt
(python bench.py --no-secondary --no-cpu-baseline --steps 4000 --warmup 10 > $O/bench_long.json 2> $O/bench_long.err) &
BP=$!
sleep 6
for i in 1 2 3 4 5 6; do rocm-smi --showpower --showclocks --showtemp 2>&1 | grep -E "Power|sclk|Temperature \(Sensor (junction|edge)" | head -8; echo ---; sleep 1; done > $O/under_load.txt
wait $BP
cat $O/idle.txt | head -40; echo ======; cat $O/under_load.txt | head -60; python -c "
import json; d=json.load(open('$O/bench_long.json')); print(d['value'], d['roofline']['clock_ghz_measured'], d['roofline']['frac'])"
