"""Multi-device entry points of the C ABI (p252_*_multi): an array of contexts, one per GPU, sharded inside the library
 — what a Rust / C caller uses to get the 8-GPU path without Python or torch.distributed.  The Python
multi-GPU driver of bench.py is poseidon252_amd/distributed.py (one process per GPU); this module only binds the ABI."""
import ctypes

import numpy as np

from . import _lib
from .hash import _as_scalars, _raise

_u64p = ctypes.POINTER(ctypes.c_uint64)


def _ctx_array(ctxs):
    arr = (ctypes.c_void_p * len(ctxs))(*[c._h for c in ctxs])
    return arr


def _check(ctxs, rc):
    if rc:
        _raise(rc, ctxs[0]._h if ctxs else None)


def hash_batch_multi(ctxs, tag, messages, in_len, out_len, out=None):
    """n messages of in_len scalars -> (n, out_len, 4); contiguous shards over `ctxs` (p252_hash_batch_multi)"""
    x = _as_scalars(messages).reshape(-1, in_len, 4) if in_len else _as_scalars(messages).reshape(0, 1, 4)
    n = x.shape[0]
    tag = _as_scalars(tag).reshape(4)
    if out is None:
        out = np.empty((n, max(out_len, 1), 4), dtype=np.uint64)
    else:
        assert out.dtype == np.uint64 and out.flags.c_contiguous and out.size == n * out_len * 4
    _check(ctxs, _lib.lib().p252_hash_batch_multi(_ctx_array(ctxs), len(ctxs), tag.ctypes.data_as(_u64p), x.ctypes.data_as(_u64p),
                                                   in_len, out_len, out.ctypes.data_as(_u64p), n))
    return out


def merkle4_tree_multi(ctxs, tag, leaves):
    """root of the arity-4 tree over `leaves` (n_ctx * 4^k of them), one complete subtree per context"""
    lv = _as_scalars(leaves).reshape(-1, 4)
    tag = _as_scalars(tag).reshape(4)
    root = np.empty(4, dtype=np.uint64)
    _check(ctxs, _lib.lib().p252_merkle4_tree_multi(_ctx_array(ctxs), len(ctxs), tag.ctypes.data_as(_u64p), lv.ctypes.data_as(_u64p),
                                                     lv.shape[0], root.ctypes.data_as(_u64p)))
    return root


def hash_batch_multi_device(ctxs, tag, d_ins, in_len, out_len, d_outs, counts, streams=None):
    """device-resident shards (torch CUDA tensors, one per context); asynchronous"""
    tag = _as_scalars(tag).reshape(4)
    k = len(ctxs)
    ins = (ctypes.c_void_p * k)(*[t.data_ptr() for t in d_ins])
    outs = (ctypes.c_void_p * k)(*[t.data_ptr() for t in d_outs])
    cnt = (ctypes.c_size_t * k)(*counts)
    sts = (ctypes.c_void_p * k)(*streams) if streams is not None else None
    _check(ctxs, _lib.lib().p252_hash_batch_multi_device(_ctx_array(ctxs), k, tag.ctypes.data_as(_u64p), ins, in_len, out_len, outs, cnt, sts))


def merkle4_tree_multi_device(ctxs, tag, d_leaves, leaves_per_ctx):
    tag = _as_scalars(tag).reshape(4)
    k = len(ctxs)
    ptrs = (ctypes.c_void_p * k)(*[t.data_ptr() for t in d_leaves])
    root = np.empty(4, dtype=np.uint64)
    _lib.prefer_torch_rccl()  # (contexts on distinct devices exchange their roots over RCCL: one copy per process)
    _check(ctxs, _lib.lib().p252_merkle4_tree_multi_device(_ctx_array(ctxs), k, tag.ctypes.data_as(_u64p), ptrs, leaves_per_ctx,
                                                            root.ctypes.data_as(_u64p)))
    return root
