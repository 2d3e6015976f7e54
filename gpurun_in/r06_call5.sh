#!/bin/bash
# round 6, call 5: stamps of the 8-rank rehearsal on whatever box this is, and what 8 concurrent bare torch processes cost
set -u
O=gpurun_out/r06
mkdir -p $O
nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /proc/loadavg
export P252_BENCH_SHARE_GPU=1 P252_BENCH_BACKEND=gloo
for i in 1 2; do
  ( time python bench.py --gpus 8 --steps 2 --warmup 1 --log2n 12 --secondary-log2n 8 > $O/e$i.json 2> $O/e$i.err ) 2> $O/e$i.time
  grep "since start" $O/e$i.err | tr '\n' ';'; grep real $O/e$i.time
done
cat > /tmp/one.py <<'PY'
import time, os
t0 = time.time()
import torch
t1 = time.time()
x = torch.zeros(1, device="cuda"); torch.cuda.synchronize()
t2 = time.time()
for _ in range(200):
    x += 1
    torch.cuda.synchronize()
t3 = time.time()
print("import %.1f s, first cuda %.1f s, 200 tiny launch+sync %.2f s" % (t1 - t0, t2 - t1, t3 - t2), flush=True)
PY
echo "--- 1 process"; ( time python /tmp/one.py ) 2>&1 | grep "import\|real"
echo "--- 8 processes at once"
( time ( for r in 1 2 3 4 5 6 7 8; do python /tmp/one.py & done; wait ) ) 2>&1 | grep "import\|real\|user"
echo "--- 8 processes at once, again"
( time ( for r in 1 2 3 4 5 6 7 8; do python /tmp/one.py & done; wait ) ) 2>&1 | grep "import\|real\|user"
cat /proc/loadavg
