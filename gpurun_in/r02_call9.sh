#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
ROOT=$(pwd); O=$ROOT/gpurun_out/r02i; mkdir -p $O
echo "# batch-size sweep of the Merkle4 digest path (bench.py --log2n L --no-cpu-baseline), final library" > $O/batch_sweep.txt
for L in 10 12 14 16 17 18 19 20 21 22 24; do
  steps=50; [ $L -ge 22 ] && steps=15
  python bench.py --log2n $L --steps $steps --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.readline());r=d['roofline'];print('2^%-2d digests: %.4g perm/s  launch mean %.4f ms  min %.4f ms  kernel %s'%($L,d['value'],r['launch_ms_mean'],r['launch_ms_min'],'k_merkle4_lat' if 2**$L<=65536 else 'k_merkle4'))" >> $O/batch_sweep.txt
done
cat $O/batch_sweep.txt
for wl in sponge42 openings encrypt; do bash tools/run_pmc.sh $wl ktrace > /dev/null 2>&1; sed -n '1,4p;/timed launches only/,$p' gpurun_out/summaries/ktrace_$wl.txt | grep -v "^$" | head -12; done
find gpurun_out -name "*.db" -size +20M -delete
