"""HBM rate of the two format-conversion kernels (k_to_canonical / k_from_canonical: 32 B in + 32 B out per scalar, one field
product each) on device-resident arrays.  Developer tool; numbers quoted in DESIGN.md §3.4."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import poseidon252_amd as P
from poseidon252_amd.synth import splitmix_scalars

ctx = P.Context(0)
for log2n in (20, 24, 26):
    n = 1 << log2n
    d_s = splitmix_scalars(0xb17e, n, "cuda:0")
    d_b = torch.empty((n, 32), dtype=torch.uint8, device="cuda")
    d_back = torch.empty_like(d_s)
    d_ok = torch.empty(n, dtype=torch.uint8, device="cuda")
    for name, fn in (("to_bytes  ", lambda: ctx.to_bytes_device(d_s, d_b, n)), ("from_bytes", lambda: ctx.from_bytes_device(d_b, d_back, n, d_ok))):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        print("%s 2^%d scalars: %.3f ms -> %.3e scalars/s, %.0f GB/s (64 B per scalar)" % (name, log2n, ms, n / ms * 1e3, n * 64 / ms / 1e6))
    assert torch.equal(d_back, d_s) and bool(d_ok.all())
