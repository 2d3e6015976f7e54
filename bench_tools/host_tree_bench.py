"""Host-to-root time of p252_merkle4_tree on a pageable 2^24-leaf array (512 MiB), against the device-resident build."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import poseidon252_amd as P

ctx = P.Context(0)
tag = P.merkle4_tag()
for log2n in (22, 24):
    n = 1 << log2n
    lv = np.random.default_rng(3).integers(0, 2 ** 62, size=(n, 4), dtype=np.uint64)
    ctx.merkle4_tree(tag, lv)
    ts = []
    for _ in range(4):
        t0 = time.perf_counter()
        root = ctx.merkle4_tree(tag, lv)
        ts.append(time.perf_counter() - t0)
    d = torch.from_numpy(lv.view(np.int64)).cuda()
    d_root = torch.empty(4, dtype=torch.int64, device="cuda")
    ctx.merkle4_tree_device(tag, d, n, d_root, None)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(4):
        ctx.merkle4_tree_device(tag, d, n, d_root, None)
    torch.cuda.synchronize()
    t_dev = (time.perf_counter() - t0) / 4
    assert np.array_equal(root, d_root.cpu().numpy().view(np.uint64))
    tl = []
    for _ in range(3):
        t0 = time.perf_counter()
        root_l, levels = ctx.merkle4_tree(tag, lv, want_levels=True)
        tl.append(time.perf_counter() - t0)
    assert np.array_equal(root_l, root)
    print("2^%d leaves: host (pageable) -> root %.2f ms (%.1f GB/s of leaves); with all levels back on the host %.2f ms; device-resident build %.2f ms" % (
        log2n, min(ts) * 1e3, n * 32 / min(ts) / 1e9, min(tl) * 1e3, t_dev * 1e3))
