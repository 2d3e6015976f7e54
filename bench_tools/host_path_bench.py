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



def cpu_stat():
    """cgroup CPU throttling counters (v2 cpu.stat / v1 cpu/cpu.stat): the pageable path is host-memcpy work under the box's CPU
    quota — a run that was throttled says so here"""
    for path in ("/sys/fs/cgroup/cpu.stat", "/sys/fs/cgroup/cpu/cpu.stat"):
        try:
            return dict((k, int(v)) for k, v in (l.split() for l in open(path)))
        except (OSError, ValueError):
            continue
    return {}


def memcpy_gbps(threads=1, mib=256):
    """what one (or `threads`) host thread(s) move with numpy copies: the staging lanes' raw material"""
    import threading
    src = [np.ones(mib << 17, dtype=np.uint64) for _ in range(threads)]
    dst = [np.empty_like(a) for a in src]

    def work(i):
        for _ in range(4):
            np.copyto(dst[i], src[i])
    ts = [threading.Thread(target=work, args=(i,)) for i in range(threads)]
    t0 = time.perf_counter()
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    return threads * 4 * (mib << 20) / (time.perf_counter() - t0) / 1e9


ctx = P.Context(0)
hb = P.HashBatch(P.Domain.Merkle4, 4, ctx=ctx)
from poseidon252_amd import _lib
import bench
print("# host: %d CPUs usable (quota %s, %d visible), staging lanes %d; numpy copy %.1f GB/s on 1 thread, %.1f on 3, %.1f on 8"
      % (bench.usable_cpus(), bench.cpu_quota(), os.cpu_count(), _lib.lib().p252_staging_lanes(1), memcpy_gbps(1), memcpy_gbps(3), memcpy_gbps(8)))
st0 = cpu_stat()
for log2n in (16, 20, 22, 22):  # (2^22 twice: the same box, minutes apart within one run — VERDICT r3 item 6)
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
st1 = cpu_stat()
print("# cgroup cpu.stat over the run: " + ", ".join("%s +%d" % (k, st1[k] - st0.get(k, 0)) for k in sorted(st1) if "throttled" in k or k == "nr_periods"))
