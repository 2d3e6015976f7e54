#!/bin/bash
# A/B-test alternative builds of the HIP library on the GPU box: bash bench_tools/ab_variants.sh [workload] lib1.so lib2.so ...
# (the default library is always measured first).  Prints one line per build.
WL=${1:-merkle4_digests}; shift
one() {
  P252_LIB_PATH=$1 python bench.py --workload $WL --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('%-28s %.4g perm/s  %.3f ms  ok=%s'%(sys.argv[1], d['value'], d['ms_per_step'], d['self_consistency_ok']))" "${1:-default}"
}
one ""
for lib in "$@"; do one "$lib"; done
