"""Developer check of the shader-clock probe (p252_clock_probe_device) on the GPU box: what s_memtime counts, the rate of
s_memrealtime against HIP events, the clock idle and under the digest kernel's load, and what sysfs offers beside it."""
import glob
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import poseidon252_amd as P
from poseidon252_amd import synth

dev = torch.device("cuda", 0)
ctx = P.Context(0)
hb = P.HashBatch(P.Domain.Merkle4, 4, ctx=ctx)
n = 1 << 20
d_in = synth.splitmix_scalars(0xC10D, 4 * n, dev)
d_out = torch.empty((n, 4), dtype=torch.int64, device=dev)
step = lambda: ctx.hash_batch_device(hb.tag, d_in, 4, 1, d_out, n)
side = torch.cuda.Stream()


def sysfs_clock():
    out = {}
    for p in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/freq*_input") + glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"):
        try:
            out[p] = open(p).read().strip().replace("\n", " | ")
        except OSError as e:
            out[p] = "unreadable: %s" % e
    return out


print("sysfs idle:", sysfs_clock())
# 1. rate of the real-time counter: a 20 ms probe between two HIP events
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
t = ctx.clock_probe(spin_us=20000)
e1.record()
torch.cuda.synchronize()
raw = [int(v) for v in t.cpu().tolist()]
ms = e0.elapsed_time(e1)
print("idle probe raw:", raw)
print("event time %.3f ms; realtime ticks %d -> %.2f MHz; memtime ticks %d -> %.1f MHz" % (ms, raw[4] - raw[1], (raw[4] - raw[1]) / ms / 1e3, raw[3] - raw[0], (raw[3] - raw[0]) / ms / 1e3))
print("idle:", ctx.clock_probe_result(t))
# 2. under load: probes on a side stream while digest launches run
for rep in range(3):
    for _ in range(10):
        step()
    res = []
    for k in range(4):
        for _ in range(2):
            step()
        res.append(ctx.clock_probe(spin_us=3000, stream=side))
        for _ in range(2):
            step()
    s = sysfs_clock()
    torch.cuda.synchronize()
    print("under load rep %d:" % rep, [ctx.clock_probe_result(r) for r in res])
    if rep == 0:
        print("sysfs under load:", s)
# 3. does the probe disturb the kernel?  20 launches with and without a concurrent probe
for with_probe in (False, True, False, True):
    torch.cuda.synchronize()
    e0.record()
    for i in range(20):
        step()
        if with_probe and i % 4 == 1:
            ctx.clock_probe(spin_us=3000, stream=side)
    e1.record()
    torch.cuda.synchronize()
    print("20 launches, probe=%s: %.4f ms per launch" % (with_probe, e0.elapsed_time(e1) / 20))
# 4. idle again after a pause
time.sleep(0.5)
t = ctx.clock_probe(spin_us=2000)
torch.cuda.synchronize()
print("idle after pause:", ctx.clock_probe_result(t))
