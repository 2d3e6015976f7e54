"""The torch (device-capable) input generator of bench.py produces the oracle's `fill_random` bytes."""
import numpy as np
import pytest

import oracle
from poseidon252_amd import synth


@pytest.mark.parametrize("seed,n,chunk", [(0xc10d, 1, 1 << 22), (0xc10d, 4097, 1 << 22), (7, 100000, 1 << 22),
                                          (0xc10d + 3, 5000, 1024), (2 ** 63 + 12345, 3000, 1500)])
def test_splitmix_scalars_match_oracle(seed, n, chunk):
    got = synth.splitmix_scalars(seed, n, "cpu", chunk=chunk).numpy().view(np.uint64)
    exp = oracle.fill_random(seed, n)
    assert np.array_equal(got, exp)


def test_all_below_p():
    x = synth.splitmix_scalars(99, 20000).numpy().view(np.uint64)
    p = int("73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001", 16)
    vals = [sum(int(x[i, k]) << (64 * k) for k in range(4)) for i in range(0, 20000, 97)]
    assert all(v < p for v in vals)
    assert len({tuple(r) for r in x[:1000]}) == 1000
