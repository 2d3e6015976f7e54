"""ctypes binding of libposeidon252_hip.so (include/poseidon252_hip.h).

The extension is mandatory: a missing library raises ImportError-like RuntimeError at first use and
a missing GPU makes Context() raise — there is no CPU path behind this package.
"""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# P252_LIB_PATH: developer switch to A/B-test alternative builds of the SAME HIP library (kernel variants)
LIB_PATH = os.environ.get("P252_LIB_PATH") or os.path.join(HERE, "libposeidon252_hip.so")

OK = 0
ERR_IO_PATTERN_VIOLATION = -1
ERR_INVALID_IO_PATTERN = -2
ERR_INVALID_ARGUMENT = -3
ERR_HIP = -4
ERR_NO_DEVICE = -5
ERR_COMM = -6
COMM_ID_BYTES = 128

# every symbol include/poseidon252_hip.h declares (tests check the .so exports all of them)
ABI_SYMBOLS = (
    "p252_device_count", "p252_create", "p252_destroy", "p252_last_error",
    "p252_permute_batch", "p252_hash_batch", "p252_merkle4_tree", "p252_merkle4_levels_len",
    "p252_permute_batch_device", "p252_hash_batch_device", "p252_merkle4_tree_device", "p252_sync",
    "p252_truncate250_device", "p252_merkle4_path_batch", "p252_merkle4_path_batch_device",
    "p252_host_alloc", "p252_host_free", "p252_host_register", "p252_host_unregister", "p252_merkle2_tree", "p252_merkle2_levels_len", "p252_merkle2_tree_device",
    "p252_encryption_tag", "p252_encrypt_batch", "p252_decrypt_batch", "p252_encrypt_batch_device",
    "p252_decrypt_batch_device",
    "p252_hash_batch_multi", "p252_hash_batch_multi_device", "p252_merkle4_tree_multi", "p252_merkle4_tree_multi_device",
    "p252_tables_size", "p252_tables_export", "p252_tables_import",
    "p252_to_bytes_device", "p252_from_bytes_device", "p252_to_bytes", "p252_from_bytes", "p252_merkle4_update_device",
    "p252_domain_separator", "p252_check_io_pattern", "p252_tag", "p252_truncate250", "p252_version",
    "p252_abi_version", "p252_merkle4_update_checked_device", "p252_clock_probe_device", "p252_staging_lanes",
    "p252_comm_unique_id", "p252_comm_create_rank", "p252_comm_create_all", "p252_comm_destroy", "p252_comm_rank", "p252_comm_size",
    "p252_merkle4_tree_sharded_device", "p252_merkle4_tree_multi_device_resident", "p252_merkle4_forest_device", "p252_merkle2_forest_device", "p252_merkle4_forest", "p252_merkle4_openings_device", "p252_merkle4_depth",
    "p252_merkle2_openings_device", "p252_merkle2_depth", "p252_merkle2_path_batch_device",
    "p252_hash_batch_truncated", "p252_hash_batch_truncated_device", "p252_wipe", "p252_scratch_residue",
    "p252_merkle4_verify_batch_device", "p252_merkle2_verify_batch_device",
    "p252_trim", "p252_comm_check", "p252_comm_backend",
)
ABI_VERSION = 8  # include/poseidon252_hip.h P252_ABI_VERSION this binding was written against

_u64p = ctypes.POINTER(ctypes.c_uint64)
_szp = ctypes.POINTER(ctypes.c_size_t)
_vp = ctypes.c_void_p
_sz = ctypes.c_size_t
_lib = None


class ExtensionMissing(RuntimeError):
    pass


def _preload_torch_hip_runtime():
    """PyTorch-ROCm wheels bundle their own copy of the HIP runtime (torch/lib/libamdhip64.so, SONAME libamdhip64.so.7 —
    the same SONAME as /opt/rocm/lib's, which libposeidon252_hip.so links to), and the first copy of a SONAME loaded into a
    process serves everyone.  Loaded after `import torch`, this library runs on torch's copy: what every test and bench
    does.  Loaded BEFORE torch it would pull in the system copy, and a later `import torch` would then run on a runtime it
    was not built against (observed: "RuntimeError: No HIP GPUs are available").  So when torch is installed but not yet
    imported, load its runtime first — the order of imports then no longer matters.  P252_SYSTEM_HIP_RUNTIME=1 opts out
    (a process that will never use torch)."""
    import sys
    if "torch" in sys.modules or os.environ.get("P252_SYSTEM_HIP_RUNTIME") == "1":
        return None
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return None
    if spec is None or not spec.origin:
        return None
    path = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if not os.path.exists(path):
        return None
    try:
        ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
    except OSError:
        return None
    # (RCCL needs no such care since ABI 8: the library does not link it and, on its first communicator call, takes the copy the
    # process already holds — torch's, once torch is imported — csrc/rccl_dyn.hpp)
    return path


def prefer_torch_rccl():
    """Called by the communicator wrappers (comm.py, multi.py) before their first call.  The library resolves RCCL on first use and
    takes a copy the process already holds (csrc/rccl_dyn.hpp) — after `import torch` that is torch's bundled one.  BEFORE torch is
    imported it would load the system copy, and a later `import torch` would then be served that copy under the same SONAME instead of
    the one it was built against; so when torch is installed but not yet imported, map its librccl first (locally, NOT RTLD_GLOBAL:
    a globally visible librccl ahead of `import torch` makes the process abort at exit, tests/test_comm_forest.py).  C and Rust
    callers need none of this: they have one RCCL, the one the resolver finds.  P252_RCCL_PATH / P252_SYSTEM_HIP_RUNTIME=1 opt out."""
    import sys
    if "torch" in sys.modules or os.environ.get("P252_RCCL_PATH") or os.environ.get("P252_SYSTEM_HIP_RUNTIME") == "1":
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if spec is None or not spec.origin:
        return
    rccl = os.path.join(os.path.dirname(spec.origin), "lib", "librccl.so")
    if os.path.exists(rccl):
        try:
            ctypes.CDLL(rccl)
        except OSError:
            pass


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ExtensionMissing(
            "%s not found: build it with `python -m poseidon252_amd.build` "
            "(hipcc --offload-arch=gfx950).  poseidon252_amd has no CPU fallback." % LIB_PATH)
    _preload_torch_hip_runtime()
    L = ctypes.CDLL(LIB_PATH)
    if not os.environ.get("P252_LIB_PATH"):
        # FIRST thing after loading: argument lists changed between versions under unchanged names, and a stale
        # build (*.so is git-ignored, so it survives a checkout) lacks the newer symbols — setting their argtypes below would
        # raise a raw "undefined symbol" AttributeError before any version check could speak
        if hasattr(L, "p252_abi_version"):
            L.p252_abi_version.restype = ctypes.c_int
            have = L.p252_abi_version()
        else:
            have = None
        if have != ABI_VERSION:
            raise ExtensionMissing("%s implements ABI version %s, this binding needs %d: rebuild it (python -m poseidon252_amd.build)"
                                   % (LIB_PATH, have, ABI_VERSION))
    if os.environ.get("P252_LIB_PATH"):
        # developer A/B switch only: an OLDER build of the library may lack entry points added since; give those a stub
        # that fails loudly when called, so that the benchmarks of the entry points it does have can still be compared
        class _Tolerant:
            def __init__(self, lib):
                self.__dict__["_l"] = lib

            def __getattr__(self, name):
                try:
                    return getattr(self._l, name)
                except AttributeError:
                    class _Missing:
                        argtypes = restype = None

                        def __call__(self, *a):
                            raise ExtensionMissing("%s is not exported by %s" % (name, LIB_PATH))
                    m = _Missing()
                    self.__dict__[name] = m
                    return m
        L = _Tolerant(L)
    L.p252_device_count.restype = ctypes.c_int
    L.p252_create.argtypes = [ctypes.c_int, ctypes.POINTER(_vp)]
    L.p252_destroy.argtypes = [_vp]
    L.p252_destroy.restype = None
    L.p252_last_error.argtypes = [_vp]
    L.p252_last_error.restype = ctypes.c_char_p
    L.p252_permute_batch.argtypes = [_vp, _u64p, _u64p, _sz]
    L.p252_hash_batch.argtypes = [_vp, _u64p, _u64p, _sz, _sz, _u64p, _sz]
    L.p252_merkle4_tree.argtypes = [_vp, _u64p, _u64p, _sz, _u64p, _u64p]
    L.p252_merkle4_levels_len.argtypes = [_sz]
    L.p252_merkle4_levels_len.restype = _sz
    L.p252_merkle2_tree.argtypes = [_vp, _u64p, _u64p, _sz, _u64p, _u64p]
    L.p252_merkle2_levels_len.argtypes = [_sz]
    L.p252_merkle2_levels_len.restype = _sz
    L.p252_merkle2_tree_device.argtypes = [_vp, _u64p, _vp, _sz, _vp, _vp, _vp]
    L.p252_permute_batch_device.argtypes = [_vp, _vp, _vp, _sz, _vp]
    L.p252_hash_batch_device.argtypes = [_vp, _u64p, _vp, _sz, _sz, _vp, _sz, _vp]
    L.p252_merkle4_tree_device.argtypes = [_vp, _u64p, _vp, _sz, _vp, _vp, _vp]
    L.p252_sync.argtypes = [_vp, _vp]
    _u8p2 = ctypes.POINTER(ctypes.c_uint8)
    L.p252_encryption_tag.argtypes = [ctypes.c_int, _sz, _u64p]
    L.p252_encrypt_batch.argtypes = [_vp, ctypes.c_int, _u64p, _u64p, _u64p, _u64p, _sz, _u64p, _sz]
    L.p252_decrypt_batch.argtypes = [_vp, ctypes.c_int, _u64p, _u64p, _u64p, _u64p, _sz, _u64p, _u8p2, _sz]
    L.p252_encrypt_batch_device.argtypes = [_vp, ctypes.c_int, _u64p, _vp, _vp, _vp, _sz, _vp, _sz, _vp]
    L.p252_decrypt_batch_device.argtypes = [_vp, ctypes.c_int, _u64p, _vp, _vp, _vp, _sz, _vp, _vp, _sz, _vp]
    _vpp = ctypes.POINTER(_vp)
    L.p252_hash_batch_multi.argtypes = [_vpp, _sz, _u64p, _u64p, _sz, _sz, _u64p, _sz]
    L.p252_hash_batch_multi_device.argtypes = [_vpp, _sz, _u64p, _vpp, _sz, _sz, _vpp, _szp, _vpp]
    L.p252_merkle4_tree_multi.argtypes = [_vpp, _sz, _u64p, _u64p, _sz, _u64p]
    L.p252_merkle4_tree_multi_device.argtypes = [_vpp, _sz, _u64p, _vpp, _sz, _u64p]
    L.p252_host_alloc.argtypes = [_sz]
    L.p252_host_alloc.restype = _vp
    L.p252_host_free.argtypes = [_vp]
    L.p252_host_free.restype = None
    L.p252_host_register.argtypes = [_vp, _sz]
    L.p252_host_unregister.argtypes = [_vp]
    L.p252_truncate250_device.argtypes = [_vp, _vp, _vp, _sz, _vp]
    _u8p = ctypes.POINTER(ctypes.c_uint8)
    L.p252_merkle4_path_batch.argtypes = [_vp, _u64p, _u64p, _u64p, _u8p, _sz, _u64p, _sz]
    L.p252_merkle4_path_batch_device.argtypes = [_vp, _u64p, _vp, _vp, _vp, _sz, _vp, _sz, _vp]
    L.p252_tables_size.restype = _sz
    L.p252_tables_export.argtypes = [_vp, _vp, _sz]
    L.p252_tables_import.argtypes = [_vp, _vp, _sz]
    L.p252_domain_separator.argtypes = [ctypes.c_int, _u64p]
    L.p252_check_io_pattern.argtypes = [ctypes.c_int, _szp, _sz, _sz]
    L.p252_tag.argtypes = [ctypes.c_int, _szp, _sz, _sz, _u64p]
    L.p252_truncate250.argtypes = [_u64p, _u64p, _sz]
    L.p252_merkle4_update_device.argtypes = [_vp, _u64p, _vp, _sz, _vp, _vp, _vp, _sz, _vp, _vp]
    L.p252_to_bytes_device.argtypes = [_vp, _vp, _vp, _sz, _vp]
    L.p252_from_bytes_device.argtypes = [_vp, _vp, _vp, _vp, _sz, _vp]
    L.p252_to_bytes.argtypes = [_u64p, _u8p, _sz]
    L.p252_from_bytes.argtypes = [_u8p, _u64p, _u8p, _sz]
    L.p252_version.restype = ctypes.c_char_p
    L.p252_merkle4_update_checked_device.argtypes = [_vp, _u64p, _vp, _sz, _vp, _vp, _vp, _sz, _vp, _vp, _vp]
    L.p252_clock_probe_device.argtypes = [_vp, _vp, ctypes.c_uint, _vp]
    L.p252_staging_lanes.argtypes = [_sz]
    L.p252_comm_unique_id.argtypes = [_vp, _sz]
    L.p252_comm_create_rank.argtypes = [_vp, _vp, _sz, ctypes.c_int, ctypes.c_int, ctypes.POINTER(_vp)]
    L.p252_comm_create_all.argtypes = [_vpp, _sz, _vpp]
    L.p252_comm_destroy.argtypes = [_vp]
    L.p252_comm_destroy.restype = None
    L.p252_comm_rank.argtypes = [_vp]
    L.p252_comm_size.argtypes = [_vp]
    L.p252_merkle4_tree_sharded_device.argtypes = [_vp, _u64p, _vp, _sz, _vp, _vp]
    L.p252_merkle4_tree_multi_device_resident.argtypes = [_vpp, _sz, _u64p, _vpp, _sz, _vpp, _vpp]
    L.p252_merkle4_forest_device.argtypes = [_vp, _u64p, _vp, _sz, _sz, _vp, _vp, _vp]
    L.p252_merkle2_forest_device.argtypes = [_vp, _u64p, _vp, _sz, _sz, _vp, _vp, _vp]
    L.p252_merkle4_forest.argtypes = [_vp, _u64p, _u64p, _sz, _sz, _u64p]
    L.p252_merkle4_openings_device.argtypes = [_vp, _vp, _sz, _vp, _vp, _sz, _vp, _vp, _vp, _vp, _vp]
    L.p252_merkle4_depth.argtypes = [_sz]
    L.p252_merkle4_depth.restype = _sz
    L.p252_merkle2_openings_device.argtypes = [_vp, _vp, _sz, _vp, _vp, _sz, _vp, _vp, _vp, _vp, _vp]
    L.p252_merkle2_depth.argtypes = [_sz]
    L.p252_merkle2_depth.restype = _sz
    L.p252_merkle2_path_batch_device.argtypes = [_vp, _u64p, _vp, _vp, _vp, _sz, _vp, _sz, _vp]
    L.p252_hash_batch_truncated.argtypes = [_vp, _u64p, _u64p, _sz, _sz, _u64p, _sz]
    L.p252_hash_batch_truncated_device.argtypes = [_vp, _u64p, _vp, _sz, _sz, _vp, _sz, _vp]
    L.p252_merkle4_verify_batch_device.argtypes = [_vp, _u64p, _vp, _vp, _vp, _sz, _vp, _vp, _sz, _vp]
    L.p252_merkle2_verify_batch_device.argtypes = [_vp, _u64p, _vp, _vp, _vp, _sz, _vp, _vp, _sz, _vp]
    L.p252_wipe.argtypes = [_vp]
    L.p252_trim.argtypes = [_vp]
    L.p252_comm_check.argtypes = [_vp, _vp]
    L.p252_comm_backend.argtypes = [ctypes.c_char_p, _sz]
    L.p252_scratch_residue.argtypes = [_vp, _u64p]
    L.p252_abi_version.restype = ctypes.c_int
    for name in ABI_SYMBOLS:
        fn = getattr(L, name)
        if fn.restype is ctypes.c_int and name not in ("p252_device_count",):
            pass
    _lib = L
    return L
