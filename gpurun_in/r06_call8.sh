#!/bin/bash
# round 6, call 8: 8 ranks sharing the GPU under a parent with live contexts — does capping the HW queues per process avoid the oversubscribed runlist?
set -u
python - <<'PY'
import os, subprocess, sys, time
sys.path.insert(0, ".")
import torch
import poseidon252_amd as P
x = torch.zeros(1, device="cuda"); torch.cuda.synchronize()
c = P.Context(0)
env = dict(os.environ, P252_BENCH_SHARE_GPU="1", P252_BENCH_BACKEND="gloo")
cmd = [sys.executable, "bench.py", "--gpus", "8", "--steps", "2", "--warmup", "1", "--log2n", "12", "--secondary-log2n", "8"]
def run(name, extra):
    t0 = time.time()
    subprocess.check_output(cmd, env=dict(env, **extra), timeout=1500, stderr=subprocess.DEVNULL)
    print("%-70s %.1f s" % (name, time.time() - t0), flush=True)
run("GPU_MAX_HW_QUEUES=2 for the ranks:", {"GPU_MAX_HW_QUEUES": "2"})
run("GPU_MAX_HW_QUEUES=1 for the ranks:", {"GPU_MAX_HW_QUEUES": "1"})
run("GPU_MAX_HW_QUEUES=3 for the ranks:", {"GPU_MAX_HW_QUEUES": "3"})
run("default (4) for the ranks:", {})
PY
