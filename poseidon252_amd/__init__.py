"""poseidon252_amd — MI355X-native batched Poseidon252 (Hades width-5 + SAFE sponge, BLS12-381 scalar
field), a drop-in for the native path of dusk-network/Poseidon252 (`dusk_poseidon::Hash` / `Domain`).

Product code path: hash.py -> ctypes -> libposeidon252_hip.so (csrc/api.cpp) -> kernels.hip (gfx950).
Nothing in this package imports oracle/ or computes hashes on the CPU.
"""
from .hash import (Context, DeviceError, Domain, Error, Hash, HashBatch, HADES_WIDTH, InvalidIOPattern,
                   IOPatternViolation, check_io_pattern, compute_tag, from_bytes, to_bytes, truncate250)
from .merkle import merkle4_tree, merkle4_forest, merkle4_tag, levels_len
from .encryption import DecryptionFailed, decrypt, decrypt_batch, encrypt, encrypt_batch, encryption_tag

__all__ = ["Context", "DeviceError", "Domain", "Error", "Hash", "HashBatch", "HADES_WIDTH", "InvalidIOPattern",
           "IOPatternViolation", "check_io_pattern", "compute_tag", "truncate250", "from_bytes", "to_bytes", "merkle4_tree", "merkle4_tag",
           "levels_len", "merkle4_forest"]
