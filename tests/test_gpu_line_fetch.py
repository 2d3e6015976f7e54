"""Whole-cache-line fetches in the sponge and opening kernels (k_sponge_lines, k_merkle4_path_lines; VERDICT r2 item 5:
"coalesced HBM loads of batched input scalars"): taken when the layout puts every lane's data on 64- / 128-byte boundaries,
bit-identical to the block-by-block kernels and to the oracle, and NOT taken (same results) for odd message lengths, depths
that are not multiples of 4 and arrays at odd offsets.  Batches above 8,192 items: below that the lane-group kernels run."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N = 8192 + 256 + 37  # ragged: the last block is partly empty


@pytest.mark.parametrize("in_len,out_len", [(6, 1), (6, 9), (10, 3), (42, 5), (42, 1), (8, 4), (12, 2), (44, 4), (7, 2), (41, 5)])
def test_sponge_line_fetch_matches_oracle(gpu_ctx, oracle_mod, in_len, out_len):
    import torch
    # ((2, 1) is not in the list: Merkle2-shaped digests take the single-permutation kernel, tests/test_gpu_parity.py)
    tag = oracle_mod.fill_random(31 * in_len + out_len, 1)[0]
    m = oracle_mod.fill_random(7 * in_len + out_len, N * in_len).reshape(N, in_len, 4)
    exp = oracle_mod.hash_batch(tag, m, in_len, out_len, threads=8)
    # device buffers: aligned (whole-line kernel when in_len is even), then the same bytes at an offset of one scalar
    # (32 B: messages no longer start on 64-byte boundaries -> the block-by-block kernel)
    flat = torch.from_numpy(np.concatenate([np.zeros((1, 4), dtype=np.uint64), m.reshape(-1, 4)]).view(np.int64)).cuda()
    d_al = flat[1:].clone()
    # ... and at an offset of two scalars (64 B): still the whole-line kernel, with the roles of the lanes exchanged (for
    # in_len = 4 k EVERY message then starts and ends in mid-line)
    flat2 = torch.from_numpy(np.concatenate([np.zeros((2, 4), dtype=np.uint64), m.reshape(-1, 4)]).view(np.int64)).cuda()
    assert d_al.data_ptr() % 128 == 0 and flat[1:].data_ptr() % 64 == 32 and flat2[2:].data_ptr() % 128 == 64
    for d_in in (d_al, flat[1:], flat2[2:]):
        d_out = torch.zeros((N, out_len, 4), dtype=torch.int64, device="cuda")
        gpu_ctx.hash_batch_device(tag, d_in, in_len, out_len, d_out, N)
        torch.cuda.synchronize()
        assert np.array_equal(d_out.cpu().numpy().view(np.uint64), exp), (in_len, out_len, d_in.data_ptr() % 64)
    assert np.array_equal(gpu_ctx.hash_batch(tag, m, in_len, out_len), exp)


@pytest.mark.parametrize("depth", [4, 8, 12, 16, 20, 36, 3, 5, 13])
def test_openings_line_fetch_matches_oracle(gpu_ctx, oracle_mod, depth):
    import torch
    tag = oracle_mod.tag(0, [4], 1)
    n = N if depth <= 20 else 8192 + 64
    leaves = oracle_mod.fill_random(100 + depth, n)
    sib = oracle_mod.fill_random(200 + depth, n * depth * 3).reshape(n, depth, 3, 4)
    pos = np.random.default_rng(depth).integers(0, 4, size=(n, depth)).astype(np.uint8)
    idx = np.arange(0, n, 7)
    exp = oracle_mod.merkle4_path_batch(tag, leaves[idx], sib[idx], pos[idx])
    d_l = torch.from_numpy(leaves.view(np.int64)).cuda()
    d_p = torch.from_numpy(pos).cuda()
    flat = torch.from_numpy(np.concatenate([np.zeros((1, 4), dtype=np.uint64), sib.reshape(-1, 4)]).view(np.int64)).cuda()
    d_al = flat[1:].clone()
    d_pos_off = torch.cat([torch.zeros(1, dtype=torch.uint8, device="cuda"), d_p.reshape(-1)])[1:]  # positions at an odd address
    got = []
    for d_s, d_pp in ((d_al, d_p), (flat[1:], d_p), (d_al, d_pos_off)):
        d_r = torch.zeros((n, 4), dtype=torch.int64, device="cuda")
        gpu_ctx.merkle4_path_batch_device(tag, d_l, d_s, d_pp, depth, d_r, n)
        torch.cuda.synchronize()
        got.append(d_r.cpu().numpy().view(np.uint64))
        assert np.array_equal(got[-1][idx], exp), depth
    assert np.array_equal(got[0], got[1]) and np.array_equal(got[0], got[2])


def test_line_fetch_switch_gives_the_same_bytes(gpu_ctx):
    """P252_LINE_FETCH=0 (the block-by-block kernels for every layout) against the default, on config-4 and opening shapes"""
    code = r'''
import hashlib, numpy as np, torch, oracle, poseidon252_amd as P
ctx = P.Context(0)
n = 20000
h = hashlib.sha256()
for in_len, out_len in ((42, 5), (6, 2)):
    tag = oracle.fill_random(in_len, 1)[0]
    m = torch.from_numpy(oracle.fill_random(9 + in_len, n * in_len).view(np.int64)).cuda()
    out = torch.zeros((n, out_len, 4), dtype=torch.int64, device="cuda")
    ctx.hash_batch_device(tag, m, in_len, out_len, out, n)
    torch.cuda.synchronize()
    h.update(out.cpu().numpy().tobytes())
tag = oracle.tag(0, [4], 1)
for depth in (12, 8):
    lv = torch.from_numpy(oracle.fill_random(1, n).view(np.int64)).cuda()
    sb = torch.from_numpy(oracle.fill_random(2, n * depth * 3).view(np.int64)).cuda()
    ps = torch.from_numpy(np.random.default_rng(3).integers(0, 4, size=(n, depth)).astype(np.uint8)).cuda()
    r = torch.zeros((n, 4), dtype=torch.int64, device="cuda")
    ctx.merkle4_path_batch_device(tag, lv, sb, ps, depth, r, n)
    torch.cuda.synchronize()
    h.update(r.cpu().numpy().tobytes())
print("DIGEST", h.hexdigest())
'''
    outs = []
    for flag in ("1", "0"):
        env = dict(os.environ, P252_LINE_FETCH=flag)
        r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, timeout=600)
        assert r.returncode == 0, r.stderr.decode()
        outs.append([l for l in r.stdout.decode().splitlines() if l.startswith("DIGEST")][0])
    assert outs[0] == outs[1]
