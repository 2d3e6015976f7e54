#!/bin/bash
# round 6, call 7: does a 9th process with a live HIP context (the pytest parent) slow the 8 ranks down?
set -u
python - <<'PY'
import os, subprocess, sys, time
env = dict(os.environ, P252_BENCH_SHARE_GPU="1", P252_BENCH_BACKEND="gloo")
cmd = [sys.executable, "bench.py", "--gpus", "8", "--steps", "2", "--warmup", "1", "--log2n", "12", "--secondary-log2n", "8"]
def run(name):
    t0 = time.time()
    subprocess.check_output(cmd, env=env, timeout=1500, stderr=subprocess.DEVNULL)
    print("%-60s %.1f s" % (name, time.time() - t0), flush=True)
run("parent without a HIP context:")
import torch
print("torch.cuda.is_available():", torch.cuda.is_available())
run("parent after torch.cuda.is_available():")
x = torch.zeros(1, device="cuda"); torch.cuda.synchronize()
run("parent holding a torch CUDA context:")
sys.path.insert(0, ".")
import poseidon252_amd as P
c = P.Context(0)
run("parent holding torch's and a p252 context:")
c.close(); del x; torch.cuda.empty_cache()
run("parent after closing the p252 context and freeing the tensor:")
cmd[3] = "7"
run("7 ranks, parent still holding torch's context:")
PY
dmesg 2>/dev/null | tail -5
