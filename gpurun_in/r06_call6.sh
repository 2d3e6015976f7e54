#!/bin/bash
# round 6, call 6: the 8-rank rehearsal — direct (file), direct (pipe), under pytest — on ONE box
set -u
O=gpurun_out/r06
mkdir -p $O
cat /proc/loadavg
export P252_BENCH_SHARE_GPU=1 P252_BENCH_BACKEND=gloo
C="python bench.py --gpus 8 --steps 2 --warmup 1 --log2n 12 --secondary-log2n 8"
echo "--- direct, stdout to a file"; ( time $C > $O/f1.json 2> $O/f1.err ) 2>&1 | grep real
echo "--- direct, stdout through a pipe"; ( time $C 2> $O/f2.err | cat > $O/f2.json ) 2>&1 | grep real
echo "--- direct, MASTER_PORT in the environment"; ( time MASTER_PORT=29777 $C > $O/f3.json 2> $O/f3.err ) 2>&1 | grep real
unset P252_BENCH_SHARE_GPU P252_BENCH_BACKEND
T=tests/test_bench_multiproc.py::test_bench_eight_ranks_rehearsal_of_the_driver_command
echo "--- pytest"; ( time python -m pytest $T -m gpu -q -p no:cacheprovider --durations=3 ) 2>&1 | grep "passed\|failed\|real\|s call\|s setup"
echo "--- python subprocess.check_output, as the test does"
python - <<'PY'
import os, subprocess, sys, time
env = dict(os.environ, P252_BENCH_SHARE_GPU="1", P252_BENCH_BACKEND="gloo")
cmd = [sys.executable, "bench.py", "--gpus", "8", "--steps", "2", "--warmup", "1", "--log2n", "12", "--secondary-log2n", "8"]
for name, extra, kw in (("stderr inherited", {}, {}), ("stderr DEVNULL", {}, {"stderr": subprocess.DEVNULL}), ("MASTER_PORT + DEVNULL", {"MASTER_PORT": "29888"}, {"stderr": subprocess.DEVNULL})):
    t0 = time.time()
    out = subprocess.check_output(cmd, env=dict(env, **extra), timeout=1500, **kw)
    print("check_output (%s): %.1f s, %d bytes" % (name, time.time() - t0, len(out)), flush=True)
PY
cat /proc/loadavg
