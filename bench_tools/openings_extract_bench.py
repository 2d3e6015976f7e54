"""p252_merkle4_openings_device against the HBM roofline: 2^20 openings of depth 12 out of a stored 2^24-leaf tree (pure data
movement: per (opening, level) one 128-byte line read — 96 of its bytes used — one contiguous 96-byte record + a position byte
written).  Beside it a device-to-device copy of the same number of bytes written (what this box's HBM gives a plain stream)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import poseidon252_amd as P
from poseidon252_amd import synth

ctx = P.Context(0)
dev = torch.device("cuda", 0)
n, k = 1 << 24, 1 << 20
d_lv = synth.splitmix_scalars(0xE1, n, dev)
root, d_levels = P.merkle4_tree(d_lv, ctx=ctx, want_levels=True)
g = torch.Generator(device=dev)
g.manual_seed(1)
for name, d_idx in (("random positions", torch.randint(0, n, (k,), dtype=torch.int32, device=dev, generator=g)),
                    ("consecutive positions", torch.arange(0, k, dtype=torch.int32, device=dev))):
    out, sib, pos, depth = ctx.merkle4_openings_device(d_lv, n, d_levels, d_idx, k)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(5):
        e0.record()
        for _ in range(10):
            ctx.merkle4_openings_device(d_lv, n, d_levels, d_idx, k)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 10)
    # (the wrapper allocates its outputs per call through torch's caching allocator: no device malloc in the loop after the first call)
    alg = k * (depth * (96 + 96 + 1) + 64 + 4)
    line_bytes = k * (depth * (128 + 96 + 1) + 64 + 4)
    print("%-22s %d openings of depth %d: %.3f ms  algorithmic %.2f GB -> %.0f GB/s (%.2f of 8 TB/s); whole lines fetched: %.0f GB/s"
          % (name, k, depth, best, alg / 1e9, alg / best / 1e6, alg / best / 1e6 / 8000, line_bytes / best / 1e6))
    roots = torch.empty((k, 4), dtype=torch.int64, device=dev)
    ctx.merkle4_path_batch_device(P.merkle4_tag(), out, sib, pos, depth, roots, k)
    torch.cuda.synchronize()
    print("   re-hashed: every opening gives the tree's root: %s" % bool((roots == root.view(1, 4)).all()))
src = torch.empty(k * 12 * 97 // 8 + 8, dtype=torch.int64, device=dev)
dst = torch.empty_like(src)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
dst.copy_(src)
e0.record()
for _ in range(10):
    dst.copy_(src)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print("device-to-device copy of %.2f GB: %.3f ms -> %.0f GB/s read + written" % (src.numel() * 8 / 1e9, ms, 2 * src.numel() * 8 / ms / 1e6))
