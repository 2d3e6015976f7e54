#!/bin/bash
# pageable host->host rate of p252_hash_batch_multi (2^22 and 2^24 Merkle4 digests) over 1 / 2 / 4 / 8 contexts on device 0
# for the per-context staging-lane budget (default) and forced lane counts (P252_HOST_LANES is read once per process,
# hence one process per point).  Result kept as profiles/r03_host_path_multi.txt.
cd "$(dirname "$0")/.."
echo "# usable CPUs: $(nproc) visible, cgroup cpu.max = $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"
for nctx in ${NCTXS:-1 2 4 8}; do for lanes in ${LANESET:-budget 1 2 3}; do for chunk in ${CHUNKS:-8}; do
env $( [ $lanes != budget ] && echo P252_HOST_LANES=$lanes ) P252_HOST_CHUNK_MB=$chunk python - <<PY
import os, time, threading, numpy as np, poseidon252_amd as P
from poseidon252_amd import multi, _lib
ctxs = [P.Context(0) for _ in range($nctx)]
tag = P.merkle4_tag()
for log2n in (22, 24):
    n = 1 << log2n
    x = np.random.default_rng(1).integers(0, 2**62, size=(n, 4, 4), dtype=np.uint64)
    out = np.zeros((n, 1, 4), dtype=np.uint64)
    multi.hash_batch_multi(ctxs, tag, x, 4, 1, out=out)
    base = len(os.listdir("/proc/self/task")); peak = [0]; stop = threading.Event()
    def watch():
        while not stop.is_set():
            peak[0] = max(peak[0], len(os.listdir("/proc/self/task"))); time.sleep(0.001)
    w = threading.Thread(target=watch); w.start()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); multi.hash_batch_multi(ctxs, tag, x, 4, 1, out=out); ts.append(time.perf_counter() - t0)
    stop.set(); w.join()
    print("contexts %d lanes/ctx %-6s (=%d) chunk %2d MiB 2^%d digests: %.3f ms -> %.3e digests/s (%.1f GB/s moved), threads added %d"
          % ($nctx, "$lanes", int(os.environ.get("P252_HOST_LANES", 0)) or _lib.lib().p252_staging_lanes($nctx), $chunk, log2n, min(ts) * 1e3, n / min(ts), n * 160 / min(ts) / 1e9, peak[0] - base - 1))
PY
done; done; done
