#!/bin/bash
# round 5: maximum sizes — 2^30 leaves (32 GiB) and, once, 2^32 leaves (128 GiB in one buffer) on one MI355X
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"; mkdir -p gpurun_out/r05big
P252_TEST_HUGE=1 timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -k "beyond_4GiB" --durations=3 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05big/huge.txt | tail -15
