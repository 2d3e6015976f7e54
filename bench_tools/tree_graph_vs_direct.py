"""One-off (round 6): does replaying a 2^24-leaf tree build as a hipGraph (12 kernel launches + one 32-byte copy) shorten it?
The `_device` builders only enqueue, so they capture; levels in caller-owned memory as the header prescribes for captured builds."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import poseidon252_amd as P

ctx = P.Context(0)
tag = P.merkle4_tag()
n = 1 << 24
g = torch.Generator(device="cuda")
g.manual_seed(6)
d = torch.randint(0, 2 ** 62, (n, 4), dtype=torch.int64, device="cuda", generator=g)
root = torch.zeros(4, dtype=torch.int64, device="cuda")
levels = torch.empty((P.levels_len(n), 4), dtype=torch.int64, device="cuda")


def timed(f, reps=40):
    for _ in range(8):
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


for name, lv in (("root only (context scratch)", None), ("all levels (caller-owned)", levels)):
    direct = lambda: ctx.merkle4_tree_device(tag, d, n, root, lv)
    direct()
    torch.cuda.synchronize()
    ref = root.clone()
    t_direct = timed(direct)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        direct()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=s):
        ctx.merkle4_tree_device(tag, d, n, root, lv)
    root.zero_()
    gr.replay()
    torch.cuda.synchronize()
    assert torch.equal(root, ref)
    t_graph = timed(gr.replay)
    t_direct2 = timed(direct)
    print("%-30s direct %.3f ms | graph replay %.3f ms | direct again %.3f ms" % (name, t_direct, t_graph, t_direct2), flush=True)
