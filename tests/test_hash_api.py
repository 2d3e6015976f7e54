"""Host logic of the Hash / HashBatch mirror (src/hash.rs:87-211) that needs no device: validation
happens before anything reaches the GPU and raises where the reference panics."""
import numpy as np
import pytest

import poseidon252_amd as P


def z(n):
    return np.zeros((n, 4), dtype=np.uint64)


def test_merkle_arity_panics_like_reference():
    # hash.rs:124-127 "# Panics ... using Domain::Merkle4 with an input anything other than 4 Scalar"
    for n in (0, 1, 3, 5):
        h = P.Hash.new(P.Domain.Merkle4)
        if n:
            h.update(z(n))
        with pytest.raises(P.Error):
            h.finalize()
    h = P.Hash(P.Domain.Merkle2)
    h.update(z(3))
    with pytest.raises(P.IOPatternViolation):
        h.finalize()


def test_empty_input_is_invalid_pattern():
    h = P.Hash(P.Domain.Other)
    with pytest.raises(P.InvalidIOPattern):
        h.finalize()
    h.update(z(0))
    with pytest.raises(P.InvalidIOPattern):
        h.finalize()


def test_output_len_rules():
    # hash.rs:111-115: honoured only for Domain::Other and > 0
    h = P.Hash(P.Domain.Merkle4)
    h.output_len(3)
    assert h._output_len == 1
    h = P.Hash(P.Domain.Other)
    h.output_len(0)
    assert h._output_len == 1
    h.output_len(7)
    assert h._output_len == 7
    assert P.HashBatch(P.Domain.Other, 4, output_len=7).out_len == 7
    assert P.HashBatch(P.Domain.Merkle4, 4, output_len=7).out_len == 1
    assert P.HashBatch(P.Domain.Encryption, 4, output_len=7).out_len == 1


def test_hashbatch_validates_at_construction():
    with pytest.raises(P.IOPatternViolation):
        P.HashBatch(P.Domain.Merkle4, 3)
    with pytest.raises(P.IOPatternViolation):
        P.HashBatch(P.Domain.Merkle2, 4)
    with pytest.raises(P.InvalidIOPattern):
        P.HashBatch(P.Domain.Other, 0)
    hb = P.HashBatch(P.Domain.Merkle4, 4)
    assert np.array_equal(hb.tag, P.compute_tag(P.Domain.Merkle4, [4], 1))
    assert np.array_equal(P.HashBatch(P.Domain.Other, 4, tag=z(1)).tag, z(1)[0])


def test_chunked_update_has_same_tag_as_one_shot():
    # README.md:40-44: splitting the input across update() calls does not change the digest
    assert np.array_equal(P.compute_tag(P.Domain.Other, [3, 39], 1), P.compute_tag(P.Domain.Other, [42], 1))
    assert not np.array_equal(P.compute_tag(P.Domain.Other, [4], 1), P.compute_tag(P.Domain.Merkle4, [4], 1))
