#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
g++ -O2 -mavx2 -pthread bench_tools/ntcopy.cpp -o /tmp/ntcopy && for t in 1 4 8 12; do /tmp/ntcopy $t; done
for mode in memcpy nt memcpy nt; do echo "== P252_HOST_COPY=$mode"; P252_HOST_COPY=$mode python bench_tools/host_path_bench.py 2>&1 | grep "pageable 2^2[02]"; done
for lanes in 8 12 14; do echo "== nt lanes $lanes"; P252_HOST_COPY=nt P252_HOST_LANES=$lanes python bench_tools/host_path_bench.py 2>&1 | grep "pageable 2^22"; done
