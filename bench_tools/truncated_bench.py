"""Hash::digest_truncated for a batch: the fused output stage (p252_hash_batch_truncated_device: ONE launch) against round 4's two
launches (digest, then p252_truncate250_device over the stored digests).  2^20 Merkle4 digests and 2^20 x (42 -> 5) sponges; results equal."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import poseidon252_amd as P
from poseidon252_amd import synth

ctx = P.Context(0)
dev = torch.device("cuda", 0)
reps = int(os.environ.get("REPS", "20"))
for name, hb, in_len in (("2^20 Merkle4 digests", P.HashBatch(P.Domain.Merkle4, 4, ctx=ctx), 4),
                         ("2^20 sponges 42 -> 5", P.HashBatch(P.Domain.Other, 42, output_len=5, ctx=ctx), 42)):
    n = 1 << 20
    d_in = synth.splitmix_scalars(0x7c, n * in_len, dev)
    fused = torch.empty((n, hb.out_len, 4), dtype=torch.int64, device=dev)
    two = torch.empty_like(fused)

    def run_fused():
        ctx.hash_batch_device(hb.tag, d_in, in_len, hb.out_len, fused, n, truncated=True)

    def run_two():
        ctx.hash_batch_device(hb.tag, d_in, in_len, hb.out_len, two, n)
        ctx.truncate250_device(two, two, n * hb.out_len)
    res = {}
    for label, fn in (("fused (one launch)", run_fused), ("digest + truncate (two launches)", run_two)):
        for _ in range(12):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(5):
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / reps)
        res[label] = best
        print("%-22s %-34s %8.4f ms per batch" % (name, label, best))
    print("%-22s equal: %s   fused saves %.1f us per batch (%.2f %%) and %d B of HBM traffic per output scalar"
          % (name, bool(torch.equal(fused, two)), (res["digest + truncate (two launches)"] - res["fused (one launch)"]) * 1e3,
             100 * (1 - res["fused (one launch)"] / res["digest + truncate (two launches)"]), 64))
