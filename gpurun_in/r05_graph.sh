#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"; mkdir -p gpurun_out/r05graph
python -m pytest tests/test_round5_gpu.py -m gpu -q -x -k "graph or one_kernel" 2>&1 | tail -25
