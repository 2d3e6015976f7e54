"""Quick on-GPU sanity + timing (developer tool; the judged paths are tests/ and bench.py)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle
import poseidon252_amd as P

ctx = P.Context(0)
tag = P.compute_tag(P.Domain.Merkle4, [4], 1)
ok = True
# permute
st = oracle.fill_random(1, 5 * 1000).reshape(1000, 5, 4)
r = np.array_equal(ctx.permute_batch(st), oracle.permute_batch(st)); print("permute parity", r); ok &= r
# merkle4 digest
x = oracle.fill_random(2, 4 * 4099).reshape(4099, 4, 4)
r = np.array_equal(ctx.hash_batch(tag, x, 4, 1), oracle.hash_batch(tag, x, 4, 1)); print("merkle4 parity", r); ok &= r
# sponge shapes (tests/hash.rs shapes + config 4)
for (i, o) in [(3, 1), (5, 1), (15, 1), (3, 3), (5, 2), (4, 7), (42, 5), (42, 1), (1, 1), (8, 4), (9, 9)]:
    t = P.compute_tag(P.Domain.Other, [i], o)
    m = oracle.fill_random(100 + i, i * 300).reshape(300, i, 4)
    r = np.array_equal(ctx.hash_batch(t, m, i, o), oracle.hash_batch(t, m, i, o)); print("sponge", i, o, r); ok &= r
# tree
for n in [1, 2, 4, 5, 16, 17, 64, 1000, 4096]:
    lv = oracle.fill_random(7 + n, n)
    root, levels = ctx.merkle4_tree(tag, lv, want_levels=True)
    oroot, olevels, _ = oracle.merkle4_tree(tag, lv, want_levels=True)
    r = np.array_equal(root, oroot) and np.array_equal(levels, olevels) and np.array_equal(ctx.merkle4_tree(tag, lv), oroot)
    print("tree", n, r); ok &= r
print("ALL PARITY", ok)
# timing: 2^20 merkle4 digests, device resident
n = 1 << 20
h = oracle.fill_random(0xc10d, 4 * n)
d_in = torch.from_numpy(h.view(np.int64)).cuda()
d_out = torch.empty((n, 4), dtype=torch.int64, device="cuda")
for _ in range(2):
    ctx.hash_batch_device(tag, d_in, 4, 1, d_out, n)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
K = 5
for _ in range(K):
    ctx.hash_batch_device(tag, d_in, 4, 1, d_out, n)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / K
print("2^20 merkle4 digests: %.3f ms  -> %.3e perm/s" % (ms, n / ms * 1e3))
got = d_out.cpu().numpy().view(np.uint64)
exp = oracle.hash_batch(tag, h[: 4 * 2048].reshape(2048, 4, 4), 4, 1).reshape(2048, 4)
print("spot parity at 2^20:", np.array_equal(got[:2048], exp))
