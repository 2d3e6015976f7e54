"""Host-side mirror of the reference's public hash API over the C ABI.

Mirrors, name for name, `dusk_poseidon::{Domain, Hash}` (src/hash.rs:21-211, re-exported at
src/lib.rs:19-22) and adds the batched sibling `HashBatch` that the GPU needs.  The reference is a
Rust crate; no Rust toolchain exists in this image, so this mirror is Python over ctypes (the Rust
binding a maintainer would add is in INTEGRATION.md).  All hashing happens in
libposeidon252_hip.so on the GPU — a missing extension or GPU raises, nothing falls back to a CPU.

Scalars are numpy uint64 arrays whose last axis is the 4 little-endian limbs of a `BlsScalar`
(Montgomery form, exactly the reference's memory layout), or torch CUDA tensors of the same bytes.
"""
import ctypes
import enum

import numpy as np

from . import _lib

__all__ = ["Domain", "Hash", "HashBatch", "Context", "Error", "IOPatternViolation", "InvalidIOPattern",
           "HADES_WIDTH", "compute_tag"]

HADES_WIDTH = 5  # dusk_poseidon::HADES_WIDTH, src/lib.rs:17

_u64p = ctypes.POINTER(ctypes.c_uint64)


class Error(Exception):
    """dusk_poseidon::Error (src/error.rs:9-29) — only the variants the hash path can produce."""


class IOPatternViolation(Error):
    """Error::IOPatternViolation (src/error.rs:12-14): Merkle domain with the wrong arity (hash.rs:70-78)."""


class InvalidIOPattern(Error):
    """Error::InvalidIOPattern (src/error.rs:16-17): empty input / zero-length chunk / zero outputs."""


class DeviceError(RuntimeError):
    """HIP failure or no device: there is no CPU fallback."""


def _raise(rc, ctx_handle=None, global_err=False):
    """global_err: the failing call had no context (p252_comm_unique_id, p252_comm_backend) — its message is p252_last_error(NULL)"""
    msg = _lib.lib().p252_last_error(ctx_handle).decode() if (ctx_handle is not None or global_err) else ""
    if rc == _lib.ERR_IO_PATTERN_VIOLATION:
        raise IOPatternViolation("io-pattern should be valid: IOPatternViolation " + msg)
    if rc == _lib.ERR_INVALID_IO_PATTERN:
        raise InvalidIOPattern("at this point the io-pattern is valid: InvalidIOPattern " + msg)
    if rc == _lib.ERR_INVALID_ARGUMENT:
        raise ValueError("poseidon252_hip: invalid argument " + msg)
    raise DeviceError("poseidon252_hip error %d: %s" % (rc, msg))


class Domain(enum.IntEnum):
    """`enum Domain` (src/hash.rs:21-36), discriminants in declaration order."""
    Merkle4 = 0
    Merkle2 = 1
    Encryption = 2
    Other = 3

    def separator(self):
        """`From<Domain> for u64` (src/hash.rs:38-56)."""
        out = ctypes.c_uint64(0)
        rc = _lib.lib().p252_domain_separator(int(self), ctypes.byref(out))
        if rc:
            _raise(rc)
        return out.value

    def __int__(self):
        return self.value


def check_io_pattern(domain, absorb_lens, output_len):
    """`io_pattern()` (src/hash.rs:62-85) + dusk-safe's validation; raises like Hash::finalize panics."""
    lens = (ctypes.c_size_t * max(1, len(absorb_lens)))(*absorb_lens)
    rc = _lib.lib().p252_check_io_pattern(int(domain), lens, len(absorb_lens), output_len)
    if rc:
        _raise(rc)


def compute_tag(domain, absorb_lens, output_len):
    """Safe::tag for this io-pattern (scalar.rs:29-31).  UNPINNED recipe — see DESIGN.md; Rust callers
    pass the value from the real crates instead (every entry point accepts `tag=`)."""
    lens = (ctypes.c_size_t * max(1, len(absorb_lens)))(*absorb_lens)
    out = np.empty(4, dtype=np.uint64)
    rc = _lib.lib().p252_tag(int(domain), lens, len(absorb_lens), output_len, out.ctypes.data_as(_u64p))
    if rc:
        _raise(rc)
    return out


def _as_scalars(a):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    if a.shape[-1] != 4:
        raise ValueError("scalar arrays need a trailing axis of 4 u64 limbs")
    return a


def _is_torch(x):
    return type(x).__module__.startswith("torch")


class Context:
    """One `p252_ctx`: bound to one HIP device (one process per GPU)."""

    _default = {}

    def __init__(self, device=0):
        L = _lib.lib()
        h = ctypes.c_void_p()
        rc = L.p252_create(int(device), ctypes.byref(h))
        if rc:
            raise DeviceError("p252_create(device=%d) failed (%d): %s" % (device, rc, L.p252_last_error(None).decode()))
        self._h = h
        self.device = int(device)

    @classmethod
    def default(cls, device=None):
        if device is None:
            device = 0
            try:
                import torch
                if torch.cuda.is_available():
                    device = torch.cuda.current_device()
            except ImportError:
                pass
        if device not in cls._default:
            cls._default[device] = cls(device)
        return cls._default[device]

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().p252_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc:
            _raise(rc, self._h)

    # ---- host buffers ----
    def permute_batch(self, states):
        """n x [BlsScalar; 5] -> n permuted states (Safe::permute, scalar.rs:25-27)."""
        s = _as_scalars(states).reshape(-1, 5, 4)
        out = np.empty_like(s)
        self._check(_lib.lib().p252_permute_batch(self._h, s.ctypes.data_as(_u64p), out.ctypes.data_as(_u64p), s.shape[0]))
        return out

    def sync(self, stream=None):
        """wait for `stream` (a torch stream; default: the current one) — p252_sync; raises DeviceError if a sharded tree build of this
        context's communicator met a failed peer since the last check (comm.Comm.check)"""
        if stream is None:
            import torch
            stream = torch.cuda.current_stream()
        self._check(_lib.lib().p252_sync(self._h, ctypes.c_void_p(stream.cuda_stream)))

    def wipe(self):
        """clear every scratch buffer the context owns (p252_wipe: device scratch, level scratch, staging lanes).  The host-buffer
        encrypt / decrypt calls and close() do this themselves (the reference builds with `zeroize`, Cargo.toml:14)."""
        self._check(_lib.lib().p252_wipe(self._h))

    def trim(self):
        """give the grow-only scratch back (p252_trim: waits for the device, wipes, frees; the next call allocates again) — the
        reference holds no state at all (hash.rs:92-96)"""
        self._check(_lib.lib().p252_trim(self._h))

    def scratch_residue(self):
        """diagnostics: non-zero bytes in the context's scratch buffers (p252_scratch_residue)"""
        n = ctypes.c_uint64(0)
        self._check(_lib.lib().p252_scratch_residue(self._h, ctypes.byref(n)))
        return int(n.value)

    def hash_batch(self, tag, messages, in_len, out_len, out=None, truncated=False):
        """truncated=True: Hash::finalize_truncated's raw limbs (hash.rs:164-183), produced by the digest kernel's output stage"""
        tag = _as_scalars(tag).reshape(4)
        m = _as_scalars(messages)
        if in_len <= 0:
            self._check(_lib.ERR_INVALID_IO_PATTERN)
        m = m.reshape(-1, in_len, 4)
        if out is None:
            out = np.empty((m.shape[0], max(out_len, 0), 4), dtype=np.uint64)
        else:  # caller-provided (e.g. pinned) output buffer
            assert out.dtype == np.uint64 and out.flags.c_contiguous and out.size == m.shape[0] * out_len * 4
            out = out.reshape(m.shape[0], out_len, 4)
        fn = _lib.lib().p252_hash_batch_truncated if truncated else _lib.lib().p252_hash_batch
        self._check(fn(self._h, tag.ctypes.data_as(_u64p), m.ctypes.data_as(_u64p), in_len, out_len, out.ctypes.data_as(_u64p), m.shape[0]))
        return out

    def merkle4_tree(self, tag, leaves, want_levels=False):
        tag = _as_scalars(tag).reshape(4)
        lv = _as_scalars(leaves).reshape(-1, 4)
        n = lv.shape[0]
        root = np.empty(4, dtype=np.uint64)
        levels = np.empty((_lib.lib().p252_merkle4_levels_len(n), 4), dtype=np.uint64) if want_levels else None
        self._check(_lib.lib().p252_merkle4_tree(self._h, tag.ctypes.data_as(_u64p), lv.ctypes.data_as(_u64p), n,
                                                  root.ctypes.data_as(_u64p),
                                                  levels.ctypes.data_as(_u64p) if want_levels else None))
        return (root, levels) if want_levels else root

    def merkle2_tree(self, tag, leaves, want_levels=False):
        """arity-2 tree over Hash::digest(Domain::Merkle2, [c0, c1]) nodes (hash.rs:27-31)"""
        tag = _as_scalars(tag).reshape(4)
        lv = _as_scalars(leaves).reshape(-1, 4)
        n = lv.shape[0]
        root = np.empty(4, dtype=np.uint64)
        levels = np.empty((_lib.lib().p252_merkle2_levels_len(n), 4), dtype=np.uint64) if want_levels else None
        self._check(_lib.lib().p252_merkle2_tree(self._h, tag.ctypes.data_as(_u64p), lv.ctypes.data_as(_u64p), n,
                                                  root.ctypes.data_as(_u64p),
                                                  levels.ctypes.data_as(_u64p) if want_levels else None))
        return (root, levels) if want_levels else root

    # ---- device buffers (torch CUDA tensors; asynchronous on torch's current stream) ----
    @staticmethod
    def _stream():
        import torch
        return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    @staticmethod
    def _nbytes(t):
        return t.numel() * t.element_size()

    def permute_batch_device(self, d_states, d_out, n):
        assert d_states.is_cuda and d_out.is_cuda and d_states.is_contiguous() and d_out.is_contiguous()
        assert self._nbytes(d_states) >= n * 160 and self._nbytes(d_out) >= n * 160
        self._check(_lib.lib().p252_permute_batch_device(self._h, d_states.data_ptr(), d_out.data_ptr(), n, self._stream()))

    def hash_batch_device(self, tag, d_in, in_len, out_len, d_out, n, truncated=False):
        """truncated=True: p252_hash_batch_truncated_device — finalize_truncated's raw limbs from the SAME launch (hash.rs:164-183)"""
        tag = _as_scalars(tag).reshape(4)
        assert d_in.is_cuda and d_out.is_cuda and d_in.is_contiguous() and d_out.is_contiguous()
        assert self._nbytes(d_in) >= n * in_len * 32 and self._nbytes(d_out) >= n * out_len * 32
        fn = _lib.lib().p252_hash_batch_truncated_device if truncated else _lib.lib().p252_hash_batch_device
        self._check(fn(self._h, tag.ctypes.data_as(_u64p), d_in.data_ptr(), in_len, out_len, d_out.data_ptr(), n, self._stream()))

    def merkle4_tree_device(self, tag, d_leaves, n_leaves, d_root, d_levels=None):
        tag = _as_scalars(tag).reshape(4)
        assert d_leaves.is_cuda and d_root.is_cuda and d_leaves.is_contiguous()
        assert self._nbytes(d_leaves) >= n_leaves * 32 and self._nbytes(d_root) >= 32
        if d_levels is not None:
            assert self._nbytes(d_levels) >= _lib.lib().p252_merkle4_levels_len(n_leaves) * 32
        self._check(_lib.lib().p252_merkle4_tree_device(self._h, tag.ctypes.data_as(_u64p), d_leaves.data_ptr(), n_leaves,
                                                         d_root.data_ptr(),
                                                         d_levels.data_ptr() if d_levels is not None else None,
                                                         self._stream()))

    def merkle4_forest(self, tag, leaves, leaves_per_tree):
        """host leaves (numpy, pageable is fine) -> roots (n_trees, 4) numpy: p252_merkle4_forest hashes the first level while the
        leaves stream in through the staging lanes, then the upper levels once across all trees"""
        tag = _as_scalars(tag).reshape(4)
        lv = _as_scalars(leaves).reshape(-1, 4)
        if leaves_per_tree < 1 or lv.shape[0] % leaves_per_tree:
            raise ValueError("forest: %d leaves are not a whole number of %d-leaf trees" % (lv.shape[0], leaves_per_tree))
        n_trees = lv.shape[0] // leaves_per_tree
        roots = np.empty((n_trees, 4), dtype=np.uint64)
        self._check(_lib.lib().p252_merkle4_forest(self._h, tag.ctypes.data_as(_u64p), lv.ctypes.data_as(_u64p), n_trees, leaves_per_tree,
                                                    roots.ctypes.data_as(_u64p)))
        return roots

    def merkle4_forest_device(self, tag, d_leaves, n_trees, leaves_per_tree, d_roots, d_levels=None, arity=4):
        """n_trees independent complete 4^k-leaf trees, tree-major in d_leaves: one launch per level across ALL trees
        (p252_merkle4_forest_device); d_roots (n_trees, 4); d_levels: level-major, n_trees * levels_len(leaves_per_tree) scalars"""
        tag = _as_scalars(tag).reshape(4)
        assert d_leaves.is_cuda and d_roots.is_cuda and d_leaves.is_contiguous() and d_roots.is_contiguous()
        assert self._nbytes(d_leaves) >= n_trees * leaves_per_tree * 32 and self._nbytes(d_roots) >= n_trees * 32
        if d_levels is not None:
            ll = _lib.lib().p252_merkle4_levels_len if arity == 4 else _lib.lib().p252_merkle2_levels_len
            assert self._nbytes(d_levels) >= n_trees * ll(leaves_per_tree) * 32
        fn = _lib.lib().p252_merkle4_forest_device if arity == 4 else _lib.lib().p252_merkle2_forest_device
        self._check(fn(self._h, tag.ctypes.data_as(_u64p), d_leaves.data_ptr(), n_trees, leaves_per_tree,
                       d_roots.data_ptr(), d_levels.data_ptr() if d_levels is not None else None, self._stream()))

    # ---- truncated outputs on the device, batched Merkle openings ----
    def truncate250_device(self, d_scalars, d_out, n):
        assert d_scalars.is_cuda and d_out.is_cuda and self._nbytes(d_scalars) >= n * 32 and self._nbytes(d_out) >= n * 32
        self._check(_lib.lib().p252_truncate250_device(self._h, d_scalars.data_ptr(), d_out.data_ptr(), n, self._stream()))

    def merkle4_update_device(self, tag, d_leaves, n_leaves, d_levels, d_indices, d_new_leaves, k, d_root=None, check=False):
        """incremental update of a stored tree: d_leaves[d_indices[i]] = d_new_leaves[i] (k distinct positions, int32/uint32
        tensor) and every ancestor in d_levels (the layout merkle4_tree_device fills) re-hashed; d_root gets the new root.
        Positions >= n_leaves are skipped by the kernels.  check=True (a debugging aid: it synchronises) raises ValueError
        when the list holds an out-of-range or a repeated position."""
        tag = _as_scalars(tag).reshape(4)
        if check and k:
            import torch
            idx = d_indices[:k].to(torch.int64) & 0xFFFFFFFF
            if int(idx.max()) >= n_leaves:
                raise ValueError("merkle4_update: position %d is outside the tree (%d leaves)" % (int(idx.max()), n_leaves))
            if int(torch.unique(idx).numel()) != k:
                raise ValueError("merkle4_update: the positions are not distinct")
        assert d_leaves.is_cuda and self._nbytes(d_leaves) >= n_leaves * 32
        assert n_leaves == 1 or (d_levels.is_cuda and self._nbytes(d_levels) >= _lib.lib().p252_merkle4_levels_len(n_leaves) * 32)
        if k:
            assert d_indices.is_cuda and d_indices.element_size() == 4 and d_indices.numel() >= k
            assert d_new_leaves.is_cuda and self._nbytes(d_new_leaves) >= k * 32
        self._check(_lib.lib().p252_merkle4_update_device(
            self._h, tag.ctypes.data_as(_u64p), d_leaves.data_ptr(), n_leaves, d_levels.data_ptr() if d_levels is not None else None,
            d_indices.data_ptr() if k else None, d_new_leaves.data_ptr() if k else None, k,
            d_root.data_ptr() if d_root is not None else None, self._stream()))

    # ---- the canonical byte format (BlsScalar::to_bytes / from_bytes) on device-resident arrays ----
    def to_bytes_device(self, d_scalars, d_bytes, n):
        assert d_scalars.is_cuda and d_bytes.is_cuda and self._nbytes(d_scalars) >= n * 32 and self._nbytes(d_bytes) >= n * 32
        self._check(_lib.lib().p252_to_bytes_device(self._h, d_scalars.data_ptr(), d_bytes.data_ptr(), n, self._stream()))

    def from_bytes_device(self, d_bytes, d_scalars, n, d_ok=None):
        assert d_scalars.is_cuda and d_bytes.is_cuda and self._nbytes(d_scalars) >= n * 32 and self._nbytes(d_bytes) >= n * 32
        assert d_ok is None or (d_ok.is_cuda and self._nbytes(d_ok) >= n)
        self._check(_lib.lib().p252_from_bytes_device(self._h, d_bytes.data_ptr(), d_scalars.data_ptr(),
                                                      d_ok.data_ptr() if d_ok is not None else None, n, self._stream()))

    def merkle4_path_batch(self, tag, leaves, siblings, positions):
        """leaves (n,4) u64; siblings (n,depth,3,4) u64; positions (n,depth) u8 in 0..3 -> roots (n,4)"""
        tag = _as_scalars(tag).reshape(4)
        lv = _as_scalars(leaves).reshape(-1, 4)
        n = lv.shape[0]
        pos = np.ascontiguousarray(positions, dtype=np.uint8).reshape(n, -1)
        depth = pos.shape[1]
        sib = _as_scalars(siblings).reshape(n, depth, 3, 4) if depth else np.zeros((n, 0, 3, 4), dtype=np.uint64)
        roots = np.empty((n, 4), dtype=np.uint64)
        u8p = ctypes.POINTER(ctypes.c_uint8)
        self._check(_lib.lib().p252_merkle4_path_batch(self._h, tag.ctypes.data_as(_u64p), lv.ctypes.data_as(_u64p),
                                                        sib.ctypes.data_as(_u64p), pos.ctypes.data_as(u8p), depth,
                                                        roots.ctypes.data_as(_u64p), n))
        return roots

    def merkle4_openings_device(self, d_leaves, n_leaves, d_levels, d_indices, k, check=False, out=None, arity=4):
        """openings of a stored tree, extracted on the device (p252_merkle4_openings_device): d_indices = k leaf positions (int32 /
        uint32 torch tensor).  Returns (d_leaves_out (k,4), d_siblings (k,depth,3,4), d_positions (k,depth) uint8, depth) — what
        merkle4_path_batch_device takes.  check=True: raises if a position lies outside the tree (they yield zero openings);
        out = (d_leaves_out, d_siblings, d_positions, d_n_bad) to write into caller-owned tensors (no allocation per call).
        arity=2: a Merkle2 tree (p252_merkle2_openings_device): one sibling per level, d_siblings (k,depth,1,4)."""
        import torch
        assert d_leaves.is_cuda and d_indices.is_cuda and d_indices.element_size() == 4 and d_indices.is_contiguous() and arity in (2, 4)
        L = _lib.lib()
        depth = int((L.p252_merkle4_depth if arity == 4 else L.p252_merkle2_depth)(n_leaves))
        ll = (L.p252_merkle4_levels_len if arity == 4 else L.p252_merkle2_levels_len)(n_leaves)
        assert self._nbytes(d_leaves) >= n_leaves * 32 and (depth == 0 or self._nbytes(d_levels) >= ll * 32)
        dev = d_leaves.device
        if out is not None:
            out, sib, pos, bad = out
            assert self._nbytes(out) >= k * 32 and self._nbytes(sib) >= k * depth * 32 * (arity - 1) and self._nbytes(pos) >= k * depth and self._nbytes(bad) >= 4
        else:
            out = torch.empty((k, 4), dtype=torch.int64, device=dev)
            sib = torch.empty((k, depth, arity - 1, 4), dtype=torch.int64, device=dev)
            pos = torch.empty((k, depth), dtype=torch.uint8, device=dev)
            bad = torch.zeros(1, dtype=torch.int32, device=dev)
        fn = L.p252_merkle4_openings_device if arity == 4 else L.p252_merkle2_openings_device
        self._check(fn(self._h, d_leaves.data_ptr(), n_leaves, d_levels.data_ptr() if depth else None, d_indices.data_ptr(), k, out.data_ptr(),
                       sib.data_ptr() if depth else None, pos.data_ptr() if depth else None, bad.data_ptr(), self._stream()))
        if check and int(bad.item()):
            raise ValueError("merkle4_openings: %d position(s) outside the tree" % int(bad.item()))
        return out, sib, pos, depth

    def merkle2_path_batch_device(self, tag, d_leaves, d_siblings, d_positions, depth, d_roots, n):
        """re-hash of n arity-2 openings (Domain::Merkle2; pass the Merkle2 tag): d_siblings (n,depth[,1],4), d_positions (n,depth) in 0..1"""
        tag = _as_scalars(tag).reshape(4)
        assert d_leaves.is_cuda and d_roots.is_cuda and self._nbytes(d_leaves) >= n * 32 and self._nbytes(d_roots) >= n * 32
        assert depth == 0 or (self._nbytes(d_siblings) >= n * depth * 32 and self._nbytes(d_positions) >= n * depth)
        self._check(_lib.lib().p252_merkle2_path_batch_device(self._h, tag.ctypes.data_as(_u64p), d_leaves.data_ptr(), d_siblings.data_ptr() if depth else None,
                                                               d_positions.data_ptr() if depth else None, depth, d_roots.data_ptr(), n, self._stream()))

    def merkle4_path_batch_device(self, tag, d_leaves, d_siblings, d_positions, depth, d_roots, n):
        tag = _as_scalars(tag).reshape(4)
        assert d_leaves.is_cuda and d_roots.is_cuda and self._nbytes(d_leaves) >= n * 32 and self._nbytes(d_roots) >= n * 32
        if depth:
            assert self._nbytes(d_siblings) >= n * depth * 96 and self._nbytes(d_positions) >= n * depth
        self._check(_lib.lib().p252_merkle4_path_batch_device(
            self._h, tag.ctypes.data_as(_u64p), d_leaves.data_ptr(), d_siblings.data_ptr() if depth else None,
            d_positions.data_ptr() if depth else None, depth, d_roots.data_ptr(), n, self._stream()))

    def merkle_verify_batch_device(self, tag, d_leaves, d_siblings, d_positions, depth, d_root, d_ok, n, arity=4):
        """`Opening::verify` in bulk (the downstream poseidon-merkle verifier, AGENTS.md:62-66): d_ok[i] (uint8) = 1 iff opening i
        re-hashes to the ONE root at d_root — p252_merkle{4,2}_verify_batch_device; n bytes come back instead of n x 32"""
        tag = _as_scalars(tag).reshape(4)
        per = 3 if arity == 4 else 1
        assert d_leaves.is_cuda and d_root.is_cuda and d_ok.is_cuda and self._nbytes(d_leaves) >= n * 32 and self._nbytes(d_root) >= 32 and self._nbytes(d_ok) >= n
        if depth:
            assert self._nbytes(d_siblings) >= n * depth * per * 32 and self._nbytes(d_positions) >= n * depth
        fn = _lib.lib().p252_merkle4_verify_batch_device if arity == 4 else _lib.lib().p252_merkle2_verify_batch_device
        self._check(fn(self._h, tag.ctypes.data_as(_u64p), d_leaves.data_ptr(), d_siblings.data_ptr() if depth else None,
                       d_positions.data_ptr() if depth else None, depth, d_root.data_ptr(), d_ok.data_ptr(), n, self._stream()))

    # ---- measurement aid: the shader clock (bench.py) ----
    def clock_probe(self, spin_us=1000, stream=None):
        """launches the one-wave clock probe (p252_clock_probe_device) on `stream` (a torch.cuda.Stream; default: the current
        one) and returns the device tensor it fills; read it with `clock_probe_result` after synchronising"""
        import torch
        stream = stream if stream is not None else torch.cuda.current_stream()
        with torch.cuda.stream(stream):  # the buffer is zeroed on the probe's own stream: ordered before the kernel
            out = torch.zeros(6, dtype=torch.int64, device="cuda:%d" % self.device)
        self._check(_lib.lib().p252_clock_probe_device(self._h, out.data_ptr(), int(spin_us), ctypes.c_void_p(stream.cuda_stream)))
        return out

    @staticmethod
    def clock_probe_result(t, realtime_hz=100e6):
        """{shader_ghz, interval_us, cycles_per_dependent_add}: the shader clock over the probe's interval"""
        m0, r0, mc, m1, r1, _ = [int(v) for v in t.cpu().tolist()]
        ticks = max(1, r1 - r0)
        return {"shader_ghz": (m1 - m0) / ticks * realtime_hz / 1e9, "interval_us": ticks / realtime_hz * 1e6,
                "cycles_per_dependent_add": (mc - m0) / 1024.0}

    # ---- constant table exchange ----
    def tables_export(self):
        size = _lib.lib().p252_tables_size()
        buf = np.empty(size // 4, dtype=np.int32)
        self._check(_lib.lib().p252_tables_export(self._h, buf.ctypes.data_as(ctypes.c_void_p), size))
        return buf

    def tables_import(self, buf):
        buf = np.ascontiguousarray(buf, dtype=np.int32)
        self._check(_lib.lib().p252_tables_import(self._h, buf.ctypes.data_as(ctypes.c_void_p), buf.nbytes))


class PinnedScalars:
    """n_scalars BlsScalars in page-locked host memory (p252_host_alloc); `.array` is a numpy view
    (n_scalars, 4) uint64.  Host-buffer entry points fed from/into such buffers copy at PCIe speed."""

    def __init__(self, n_scalars):
        nbytes = max(1, int(n_scalars)) * 32
        self._ptr = _lib.lib().p252_host_alloc(nbytes)
        if not self._ptr:
            raise MemoryError("p252_host_alloc(%d) failed" % nbytes)
        buf = (ctypes.c_uint64 * (nbytes // 8)).from_address(self._ptr)
        self.array = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 4)[:n_scalars]

    def free(self):
        if getattr(self, "_ptr", None):
            self.array = None
            _lib.lib().p252_host_free(self._ptr)
            self._ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class registered:
    """`with registered(array):` page-locks a numpy array the caller owns for the duration of the block
    (p252_host_register / p252_host_unregister): host-buffer calls on it then copy at PCIe speed."""

    def __init__(self, array):
        self._a = array
        assert array.flags["C_CONTIGUOUS"]

    def __enter__(self):
        rc = _lib.lib().p252_host_register(self._a.ctypes.data, self._a.nbytes)
        if rc:
            _raise(rc)
        return self._a

    def __exit__(self, *exc):
        _lib.lib().p252_host_unregister(self._a.ctypes.data)
        return False


def truncate250(scalars):
    """finalize_truncated's post-processing (hash.rs:164-183): raw limbs for JubJubScalar::from_raw."""
    s = _as_scalars(scalars)
    out = np.empty_like(s)
    rc = _lib.lib().p252_truncate250(s.ctypes.data_as(_u64p), out.ctypes.data_as(_u64p), s.size // 4)
    if rc:
        _raise(rc)
    return out


def to_bytes(scalars):
    """`BlsScalar::to_bytes` for an array of scalars: (n,4) u64 Montgomery limbs -> (n,32) u8, the little-endian bytes of the
    canonical values (host-side; `Context.to_bytes_device` for device-resident arrays)"""
    s = _as_scalars(scalars).reshape(-1, 4)
    out = np.empty((s.shape[0], 32), dtype=np.uint8)
    rc = _lib.lib().p252_to_bytes(s.ctypes.data_as(_u64p), out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), s.shape[0])
    if rc:
        _raise(rc)
    return out


def from_bytes(data):
    """`BlsScalar::from_bytes` for n records of 32 little-endian bytes -> ((n,4) u64 Montgomery limbs, ok (n,) bool);
    ok[i] False = the value is not below p (from_bytes fails there; the limbs are those of the value mod p)"""
    b = np.ascontiguousarray(data, dtype=np.uint8).reshape(-1, 32)
    out = np.empty((b.shape[0], 4), dtype=np.uint64)
    ok = np.zeros(b.shape[0], dtype=np.uint8)
    u8p = ctypes.POINTER(ctypes.c_uint8)
    rc = _lib.lib().p252_from_bytes(b.ctypes.data_as(u8p), out.ctypes.data_as(_u64p), ok.ctypes.data_as(u8p), b.shape[0])
    if rc:
        _raise(rc)
    return out, ok.astype(bool)


class Hash:
    """`dusk_poseidon::Hash` (src/hash.rs:87-211): one message, absorbed in chunks, squeezed once.

    `tag=` overrides the capacity element (pass BlsScalar::hash_to_scalar output from the real crates)."""

    def __init__(self, domain, ctx=None, tag=None):  # Hash::new, hash.rs:98-105
        self.domain = Domain(domain)
        self.input = []
        self._output_len = 1
        self._ctx = ctx
        self._tag = tag

    @classmethod
    def new(cls, domain, **kw):
        return cls(domain, **kw)

    def output_len(self, output_len):  # hash.rs:111-115: honoured only for Domain::Other and > 0
        if self.domain == Domain.Other and output_len > 0:
            self._output_len = output_len

    def update(self, scalars):  # hash.rs:118-120
        self.input.append(_as_scalars(scalars).reshape(-1, 4))

    def finalize(self, truncated=False):  # hash.rs:128-155
        lens = [c.shape[0] for c in self.input]
        check_io_pattern(self.domain, lens, self._output_len)  # raises where the reference panics
        tag = self._tag if self._tag is not None else compute_tag(self.domain, lens, self._output_len)
        msg = np.concatenate(self.input, axis=0)
        ctx = self._ctx or Context.default()
        return ctx.hash_batch(tag, msg[None], msg.shape[0], self._output_len, truncated=truncated)[0]

    def finalize_truncated(self):  # hash.rs:164-183 — truncated by the digest kernel's output stage (one launch)
        return self.finalize(truncated=True)

    @classmethod
    def digest(cls, domain, scalars, **kw):  # hash.rs:191-195
        h = cls(domain, **kw)
        h.update(scalars)
        return h.finalize()

    @classmethod
    def digest_truncated(cls, domain, scalars, **kw):  # hash.rs:203-210
        h = cls(domain, **kw)
        h.update(scalars)
        return h.finalize_truncated()


class HashBatch:
    """Batched sibling of `Hash`: n independent messages with one io-pattern, one kernel launch.

    hb = HashBatch(Domain.Merkle4, item_len=4); digests = hb.digest(scalars)   # (n,4,4) -> (n,1,4)
    Per item the result equals Hash::digest(domain, item) — same validation, same tag, same order."""

    def __init__(self, domain, item_len, output_len=1, ctx=None, tag=None):
        self.domain = Domain(domain)
        self.item_len = int(item_len)
        self.out_len = int(output_len) if (self.domain == Domain.Other and output_len > 0) else 1  # hash.rs:111-115
        check_io_pattern(self.domain, [self.item_len], self.out_len)
        self.tag = _as_scalars(tag).reshape(4) if tag is not None else compute_tag(self.domain, [self.item_len], self.out_len)
        self._ctx = ctx

    @property
    def ctx(self):
        if self._ctx is None:
            self._ctx = Context.default()
        return self._ctx

    def digest(self, scalars, out=None, truncated=False):
        if _is_torch(scalars):
            import torch
            n = scalars.numel() * scalars.element_size() // (self.item_len * 32)
            if out is None:
                out = torch.empty((n, self.out_len, 4), dtype=torch.int64, device=scalars.device)
            self.ctx.hash_batch_device(self.tag, scalars, self.item_len, self.out_len, out, n, truncated=truncated)
            return out
        return self.ctx.hash_batch(self.tag, scalars, self.item_len, self.out_len, out=out, truncated=truncated)

    def digest_truncated(self, scalars, out=None):
        """Hash::digest_truncated (hash.rs:203-210) per item, host or device buffers: ONE kernel launch — the digest kernel's
        output stage canonicalises, masks to 250 bits and stores the raw limbs JubJubScalar::from_raw receives"""
        return self.digest(scalars, out=out, truncated=True)
