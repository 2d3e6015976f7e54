"""PCIe-inclusive rate of the host-buffer entry point p252_hash_batch: pageable numpy memory (library-owned staging
lanes, nothing of the caller's is page-locked) vs page-locked buffers from p252_host_alloc (zero-copy DMA).
Developer tool; numbers quoted in DESIGN.md §3.5 and kept as profiles/r02_host_path.txt."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import poseidon252_amd as P
from poseidon252_amd.hash import PinnedScalars

ctx = P.Context(0)
hb = P.HashBatch(P.Domain.Merkle4, 4, ctx=ctx)
for log2n in (16, 20, 22):
    n = 1 << log2n
    rng = np.random.default_rng(1)
    x = rng.integers(0, 2 ** 62, size=(n, 4, 4), dtype=np.uint64)
    hb.digest(x[:1024])
    out = np.zeros((n, 1, 4), dtype=np.uint64)  # caller-owned, already touched: what a Rust caller re-using a Vec sees
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        hb.digest(x, out=out)
        ts.append(time.perf_counter() - t0)
    best = min(ts)
    print("pageable 2^%d digests: %.3f ms -> %.3e digests/s, %.1f GB/s moved" % (log2n, best * 1e3, n / best, n * 160 / best / 1e9))
    pin_in, pin_out = PinnedScalars(4 * n), PinnedScalars(n)
    pin_in.array[:] = x.reshape(-1, 4)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        out2 = ctx.hash_batch(hb.tag, pin_in.array, 4, 1, out=pin_out.array)
        ts.append(time.perf_counter() - t0)
    best = min(ts)
    assert np.array_equal(out2.reshape(-1, 4), out.reshape(-1, 4))
    print("pinned   2^%d digests: %.3f ms -> %.3e digests/s, %.1f GB/s moved" % (log2n, best * 1e3, n / best, n * 160 / best / 1e9))
    pin_in.free()
    pin_out.free()
