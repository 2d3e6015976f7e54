"""Root-only tree builds of ONE context on several streams (round 5: the level scratch is owned per stream): K independent 2^22- /
2^24-leaf trees issued back to back on 1, 2 and 4 streams — the narrow levels of one build (latency-bound: one wave per SIMD) overlap the
wide levels of another.  Every root is compared with the single-stream result."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import poseidon252_amd as P
from poseidon252_amd import synth

ctx = P.Context(0)
dev = torch.device("cuda", 0)
tag = P.merkle4_tag()
for log2n, K in ((24, 8), (22, 16), (18, 64)):
    n = 1 << log2n
    leaves = [synth.splitmix_scalars(0x5151 + i, n, dev) for i in range(min(K, 4))]
    roots = torch.zeros((K, 4), dtype=torch.int64, device=dev)
    ref = None
    for n_streams in (1, 2, 4):
        streams = [torch.cuda.Stream() for _ in range(n_streams)]
        def run():
            for i in range(K):
                with torch.cuda.stream(streams[i % n_streams]):
                    ctx.merkle4_tree_device(tag, leaves[i % len(leaves)], n, roots[i], None)
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(5):
            torch.cuda.synchronize()
            e0.record()
            run()
            for s in streams:
                torch.cuda.current_stream().wait_stream(s)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        if ref is None:
            ref = roots.clone()
        same = bool(torch.equal(ref, roots))
        perms = K * P.levels_len(n)
        print("%2d trees of 2^%d leaves on %d stream(s): %8.3f ms  = %6.3f ms per tree  %.3e perm/s   roots equal the single-stream ones: %s"
              % (K, log2n, n_streams, best, best / K, perms / best * 1e3, same))
