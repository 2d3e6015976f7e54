"""p252_merkle4_forest_device against one p252_merkle4_tree_device call per tree: the narrow upper levels
of many small trees fill the chip together.  Prints permutations/s per shape; oracle check of a sample of roots."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import poseidon252_amd as P
from poseidon252_amd import synth

ctx = P.Context(0)
tag = P.merkle4_tag()
dev = torch.device("cuda", 0)


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return best


for n_trees, per in ((1024, 4 ** 6), (4096, 4 ** 6), (4096, 4 ** 4), (65536, 4 ** 3), (256, 4 ** 8), (16, 4 ** 10)):
    d = synth.splitmix_scalars(0xF0, n_trees * per, dev)
    roots = torch.empty((n_trees, 4), dtype=torch.int64, device=dev)
    perms = n_trees * P.levels_len(per)
    ms = timed(lambda: ctx.merkle4_forest_device(tag, d, n_trees, per, roots), 20)
    line = "forest %6d trees x 4^%d leaves: %8.3f ms  %.3e perm/s" % (n_trees, round(np.log2(per) / 2), ms, perms / (ms * 1e-3))
    if n_trees <= 1024:
        one = torch.empty(4, dtype=torch.int64, device=dev)

        def loop():
            for t in range(n_trees):
                ctx.merkle4_tree_device(tag, d[t * per:(t + 1) * per], per, one)
        ms1 = timed(loop, 2)
        line += "   | one call per tree: %8.3f ms  %.3e perm/s (%.1f x)" % (ms1, perms / (ms1 * 1e-3), ms1 / ms)
    if "--check" in sys.argv:
        import oracle
        h = d.cpu().numpy().view(np.uint64)
        r = roots.cpu().numpy().view(np.uint64)
        ok = all(np.array_equal(r[t], oracle.merkle4_tree(tag, h[t * per:(t + 1) * per])[0]) for t in range(0, n_trees, max(1, n_trees // 16)))
        line += "   oracle sample: %s" % ("ok" if ok else "MISMATCH")
    print(line, flush=True)

# host leaves -> host roots (pageable numpy memory): p252_merkle4_forest streams whole trees through the staging lanes
import time
for n_trees, per in ((4096, 4 ** 6), (65536, 4 ** 3)):
    h = synth.splitmix_scalars(0xF1, n_trees * per, dev).cpu().numpy().view(np.uint64)
    ctx.merkle4_forest(tag, h, per)
    best = 1e9
    for _ in range(4):
        t0 = time.perf_counter()
        r = ctx.merkle4_forest(tag, h, per)
        best = min(best, time.perf_counter() - t0)
    perms = n_trees * P.levels_len(per)
    line = "host forest %6d trees x 4^%d leaves (%d MiB pageable): %8.3f ms  %.3e perm/s  %.1f GB/s of leaves" % (
        n_trees, round(np.log2(per) / 2), n_trees * per * 32 >> 20, best * 1e3, perms / best, n_trees * per * 32 / best / 1e9)
    if "--check" in sys.argv:
        import oracle
        ok = all(np.array_equal(r[t], oracle.merkle4_tree(tag, h[t * per:(t + 1) * per])[0]) for t in range(0, n_trees, max(1, n_trees // 16)))
        line += "   oracle sample: %s" % ("ok" if ok else "MISMATCH")
    print(line, flush=True)
