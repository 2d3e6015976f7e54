"""The canonical byte format on either side of the path — BlsScalar::to_bytes / from_bytes (the reference round-trips its
round constants through the pair, src/hades/round_constants.rs:56-71, and reads its known-answer inputs with from_hex_str,
src/hades.rs:131): p252_to_bytes / p252_from_bytes (host) and their _device twins against the oracle and big integers."""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _edge_values(P):
    return [0, 1, 2, P - 1, P - 2, (P - 1) // 2, 1 << 255, (1 << 255) - 19, pow(2, 256, P), pow(3, 300, P), (1 << 200) + 7]


def _le_bytes(vals):
    return np.frombuffer(b"".join(int(v).to_bytes(32, "little") for v in vals), dtype=np.uint8).reshape(-1, 32)


def test_host_conversions_against_the_oracle(oracle_mod):
    import poseidon252_amd as P252
    P = oracle_mod.P
    vals = [v % P for v in _edge_values(P)] + [int.from_bytes(os.urandom(32), "little") % P for _ in range(500)]
    mont = np.stack([oracle_mod.mont_from_int(v) for v in vals])
    b = P252.to_bytes(mont)
    assert np.array_equal(b, _le_bytes(vals))                      # to_bytes = little-endian canonical value
    back, ok = P252.from_bytes(b)
    assert ok.all() and np.array_equal(back, mont)                 # from_bytes(to_bytes(x)) == x (round_constants.rs:66-67)
    # values that are not below p: from_bytes fails (ok False); the limbs are those of the value mod p
    big = [P, P + 1, 2 * P - 1, 2 * P + 5, (1 << 256) - 1, (1 << 256) - (1 << 200)]
    out, ok = P252.from_bytes(_le_bytes(big))
    assert not ok.any() and np.array_equal(out, np.stack([oracle_mod.mont_from_int(v % P) for v in big]))
    assert P252.to_bytes(np.zeros((0, 4), dtype=np.uint64)).shape == (0, 32)


def test_reference_round_constants_roundtrip_through_bytes(oracle_mod):
    """round_constants.rs:56-71: every round constant survives to_bytes -> from_bytes"""
    import poseidon252_amd as P252
    import ctypes
    consts = np.empty((68 * 5, 4), dtype=np.uint64)
    for r in range(68):
        for i in range(5):
            oracle_mod.lib().p252o_round_constant(r, i, consts[r * 5 + i].ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)))
    back, ok = P252.from_bytes(P252.to_bytes(consts))
    assert ok.all() and np.array_equal(back, consts)
    # arc.bin holds the raw integers, which ARE the canonical values
    raw = np.frombuffer(open(os.path.join(os.path.dirname(HERE), "poseidon252_amd", "assets", "arc.bin"), "rb").read(), dtype=np.uint8)[:68 * 5 * 32].reshape(-1, 32)
    assert np.array_equal(P252.to_bytes(consts), raw)


@pytest.mark.gpu
def test_device_conversions_and_the_kat_in_its_own_format(gpu_ctx, oracle_mod):
    """device kernels = host twins; and the reference's KAT exactly as it is written: hex inputs -> from_bytes -> sponge ->
    to_bytes -> big-endian hex (src/hades.rs:94-162)"""
    import torch
    import poseidon252_amd as P252
    P = oracle_mod.P
    vals = _edge_values(P) + [P, P + 1, 2 * P + 5, (1 << 256) - 1] + [int.from_bytes(os.urandom(32), "little") for _ in range(3000)]
    b = _le_bytes(vals)
    n = b.shape[0]
    d_b = torch.from_numpy(b.copy()).cuda()
    d_s = torch.empty((n, 4), dtype=torch.int64, device="cuda")
    d_ok = torch.zeros(n, dtype=torch.uint8, device="cuda")
    gpu_ctx.from_bytes_device(d_b, d_s, n, d_ok)
    torch.cuda.synchronize()
    h_s, h_ok = P252.from_bytes(b)
    assert np.array_equal(d_s.cpu().numpy().view(np.uint64), h_s) and np.array_equal(d_ok.cpu().numpy().astype(bool), h_ok)
    assert [bool(x) for x in h_ok[:len(_edge_values(P)) + 4]] == [v < P for v in vals[:len(_edge_values(P)) + 4]]
    d_back = torch.empty((n, 32), dtype=torch.uint8, device="cuda")
    gpu_ctx.to_bytes_device(d_s, d_back, n)
    torch.cuda.synchronize()
    assert np.array_equal(d_back.cpu().numpy(), _le_bytes([v % P for v in vals]))
    gpu_ctx.to_bytes_device(d_s, d_s, n)  # in place
    torch.cuda.synchronize()
    assert np.array_equal(d_s.cpu().numpy().view(np.uint8).reshape(n, 32), _le_bytes([v % P for v in vals]))
    # the KAT in the reference's own data format
    kat = json.load(open(os.path.join(HERE, "golden", "hades_kat.json")))
    ins, ok = P252.from_bytes(np.frombuffer(b"".join(bytes.fromhex(h) for h in kat["inputs_le_hex"]), dtype=np.uint8).reshape(-1, 32))
    assert ok.all()
    one, _ = P252.from_bytes(_le_bytes([1]))
    for k, exp in kat["expected_be_hex"].items():
        msg = np.concatenate([ins[:int(k)], one])[None]
        digest = gpu_ctx.hash_batch(np.zeros(4, dtype=np.uint64), msg, int(k) + 1, 1).reshape(1, 4)
        assert P252.to_bytes(digest)[0, ::-1].tobytes().hex() == exp
