#!/bin/bash
# pageable host->host rate of p252_hash_batch (2^22 Merkle4 digests) for several staging-lane counts and chunk sizes
# (P252_HOST_LANES / P252_HOST_CHUNK_MB are read once per process, hence one process per point)
cd "$(dirname "$0")/.."
for chunk in 4 8 16; do for lanes in 2 3 4 6 8 12; do
P252_HOST_LANES=$lanes P252_HOST_CHUNK_MB=$chunk python - <<PY
import time, numpy as np, poseidon252_amd as P
ctx = P.Context(0); hb = P.HashBatch(P.Domain.Merkle4, 4, ctx=ctx)
n = 1 << 22
x = np.random.default_rng(1).integers(0, 2**62, size=(n, 4, 4), dtype=np.uint64)
out = np.zeros((n, 1, 4), dtype=np.uint64)
hb.digest(x, out=out)
ts = []
for _ in range(6):
    t0 = time.perf_counter(); hb.digest(x, out=out); ts.append(time.perf_counter() - t0)
print("lanes %2d chunk %2d MiB: %.3f ms -> %.3e digests/s (%.1f GB/s moved)" % ($lanes, $chunk, min(ts) * 1e3, n / min(ts), n * 160 / min(ts) / 1e9))
PY
done; done
nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null
