#!/bin/bash
# round 5: is one wave per SIMD still the right padding of a large tree's narrow levels? (P252_TREE_PAD_LANES sweep, same box, interleaved twice)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"; O=$ROOT/gpurun_out/r05pad; mkdir -p "$O"
for rep in 1 2; do for pad in 65536 0 32768 131072 262144; do
  P252_TREE_PAD_LANES=$pad python bench.py --workload tree --no-cpu-baseline --steps 30 2>/dev/null | python -c "
import sys, json; d=json.loads(sys.stdin.read()); print('pad %7d  rep $rep  %.4f ms/step  frac %.4f  clock %.3f' % ($pad, d['ms_per_step'], d['roofline']['frac'], d['roofline']['clock_ghz_measured']))"
done; done | tee "$O/pad_sweep.txt"
