"""RCCL communicator of the library (p252_comm_*, include/poseidon252_hip.h): the multi-GPU exchange steps run INSIDE
libposeidon252_hip.so — ncclBroadcast of the constant table when the communicator is created (validated against every rank's
own derivation), ncclAllGather of the 32-byte subtree roots on the rank's stream in the sharded tree build (BASELINE
configs[4]) — so a Rust / C caller gets them without Python, and the Python driver without a host round trip.

One process per GPU: `Comm.create_rank(ctx, rank, world, exchange)` where `exchange(id_bytes_or_None) -> id_bytes` hands rank 0's
128-byte id to every rank (torch.distributed broadcast, an env store, MPI, a file: `torch_exchange` below does the first).
One process, several GPUs: `Comm.create_all(ctxs)`."""
import ctypes

import numpy as np

from . import _lib
from .hash import _as_scalars, _raise

_u64p = ctypes.POINTER(ctypes.c_uint64)


def backend():
    """path of the RCCL the library's calls go to in this process (p252_comm_backend; resolves it if nothing has yet)"""
    _lib.prefer_torch_rccl()
    buf = ctypes.create_string_buffer(4096)
    rc = _lib.lib().p252_comm_backend(buf, len(buf))
    if rc:
        _raise(rc, None, global_err=True)
    return buf.value.decode()


def unique_id():
    _lib.prefer_torch_rccl()
    buf = ctypes.create_string_buffer(_lib.COMM_ID_BYTES)
    rc = _lib.lib().p252_comm_unique_id(buf, _lib.COMM_ID_BYTES)
    if rc:
        _raise(rc, None, global_err=True)
    return buf.raw


def torch_exchange(device=None):
    """exchange function over an initialised torch.distributed process group (any backend): broadcast of rank 0's id"""
    def exchange(id_bytes):
        import torch
        import torch.distributed as dist
        t = torch.zeros(_lib.COMM_ID_BYTES, dtype=torch.uint8)
        if id_bytes is not None:
            t = torch.frombuffer(bytearray(id_bytes), dtype=torch.uint8).clone()
        if device is not None:
            t = t.to(device)
        dist.broadcast(t, src=0)
        return bytes(t.cpu().numpy().tobytes())
    return exchange


class Comm:
    """one rank of a communicator, bound to one Context"""

    def __init__(self, handle, ctx):
        self._h, self.ctx = handle, ctx

    @classmethod
    def create_rank(cls, ctx, rank, world, exchange):
        id_bytes = exchange(unique_id() if rank == 0 else None)
        assert len(id_bytes) == _lib.COMM_ID_BYTES
        h = ctypes.c_void_p()
        _lib.prefer_torch_rccl()
        rc = _lib.lib().p252_comm_create_rank(ctx._h, id_bytes, _lib.COMM_ID_BYTES, rank, world, ctypes.byref(h))
        if rc:
            _raise(rc, ctx._h)
        return cls(h, ctx)

    @classmethod
    def create_all(cls, ctxs):
        k = len(ctxs)
        arr = (ctypes.c_void_p * k)(*[c._h for c in ctxs])
        out = (ctypes.c_void_p * k)()
        _lib.prefer_torch_rccl()
        rc = _lib.lib().p252_comm_create_all(arr, k, out)
        if rc:
            _raise(rc, ctxs[0]._h)
        return [cls(ctypes.c_void_p(out[t]), ctxs[t]) for t in range(k)]

    @property
    def rank(self):
        return _lib.lib().p252_comm_rank(self._h)

    @property
    def size(self):
        return _lib.lib().p252_comm_size(self._h)

    def merkle4_tree_sharded_device(self, tag, d_leaves, n_leaves_local, d_root):
        """this rank's 4^k resident leaves -> subtree root -> ncclAllGather of the `size` roots -> top levels, all on the
        current torch stream; d_root (32 B, device) = the root over the concatenation of all ranks' leaves, on every rank"""
        import torch
        tag = _as_scalars(tag).reshape(4)
        assert d_leaves.numel() * d_leaves.element_size() >= n_leaves_local * 32 and d_root.numel() * d_root.element_size() >= 32
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        rc = _lib.lib().p252_merkle4_tree_sharded_device(self._h, tag.ctypes.data_as(_u64p), d_leaves.data_ptr(), n_leaves_local, d_root.data_ptr(), st)
        if rc:
            _raise(rc, self.ctx._h)

    def check(self):
        """wait for the current torch stream, then raise DeviceError if a sharded build of this communicator met a failed peer since
        the last check (p252_comm_check): such a build's root is all-ones on every healthy rank"""
        import torch
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        rc = _lib.lib().p252_comm_check(self._h, st)
        if rc:
            _raise(rc, self.ctx._h)

    def destroy(self):
        if self._h:
            _lib.lib().p252_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


def merkle4_tree_multi_device_resident(ctxs, tag, d_leaves, leaves_per_ctx, d_roots):
    """p252_merkle4_tree_multi_device_resident: one process, contexts on distinct devices; the root lands in d_roots[t] on
    every device, asynchronously on the default streams"""
    tag = _as_scalars(tag).reshape(4)
    k = len(ctxs)
    arr = (ctypes.c_void_p * k)(*[c._h for c in ctxs])
    ptrs = (ctypes.c_void_p * k)(*[t.data_ptr() for t in d_leaves])
    outs = (ctypes.c_void_p * k)(*[t.data_ptr() for t in d_roots])
    _lib.prefer_torch_rccl()
    rc = _lib.lib().p252_merkle4_tree_multi_device_resident(arr, k, tag.ctypes.data_as(_u64p), ptrs, leaves_per_ctx, outs, None)
    if rc:
        _raise(rc, ctxs[0]._h)
