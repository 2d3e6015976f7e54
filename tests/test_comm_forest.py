"""RCCL inside the library (p252_comm_*, p252_merkle4_tree_sharded_device, the RCCL path of p252_merkle4_tree_multi_device)
and the forest entry point (p252_merkle4_forest_device).

CPU part: the library exports the communicator entry points and does NOT link librccl (resolved at run time since ABI 8).  GPU part (one GPU here: world = 1 on the
real backend — RCCL refuses two ranks on one device, so the >1-rank composition is covered by the gloo / shared-GPU bench
tests and by the driver's 8-GPU run): both ways to create a communicator, the sharded tree against the oracle, the
multi-device entry point creating its communicator itself, and forests of every shape against per-tree oracle roots."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_the_communicator_without_linking_rccl():
    from poseidon252_amd import _lib
    dyn = subprocess.check_output(["readelf", "-d", _lib.LIB_PATH]).decode()
    assert "librccl" not in dyn and "libamdhip64.so" in dyn, dyn  # RCCL is no load-time dependency
    undefined = subprocess.check_output(["nm", "-D", "--undefined-only", _lib.LIB_PATH]).decode()
    assert " nccl" not in undefined, undefined
    L = ctypes.CDLL(_lib.LIB_PATH)
    for s in ("p252_comm_unique_id", "p252_comm_create_rank", "p252_comm_create_all", "p252_comm_destroy", "p252_comm_rank",
              "p252_comm_size", "p252_comm_check", "p252_comm_backend", "p252_merkle4_tree_sharded_device",
              "p252_merkle4_tree_multi_device_resident", "p252_merkle4_forest_device"):
        assert hasattr(L, s), s
    # argument checks that need no device
    L.p252_comm_rank.argtypes = L.p252_comm_size.argtypes = [ctypes.c_void_p]
    assert L.p252_comm_rank(None) == -1 and L.p252_comm_size(None) == 0
    L.p252_comm_unique_id.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    assert L.p252_comm_unique_id(None, 128) == _lib.ERR_INVALID_ARGUMENT
    buf = ctypes.create_string_buffer(128)
    assert L.p252_comm_unique_id(buf, 64) == _lib.ERR_INVALID_ARGUMENT
    hdr = open(os.path.join(ROOT, "include", "poseidon252_hip.h")).read()
    assert "#define P252_COMM_ID_BYTES 128" in hdr and "#define P252_ERR_COMM (-6)" in hdr


def test_binding_reports_a_stale_library_as_such(tmp_path):
    """a stale build lacks newer symbols; the binding must say 'rebuild it' (ExtensionMissing), not die with a raw
    ctypes 'undefined symbol' while setting argtypes — the ABI version is checked FIRST.  A stand-in library without
    p252_abi_version (and one with an old number) plays the stale build."""
    import sys
    for body, what in (("int p252_device_count(void) { return 0; }", "None"),
                       ("int p252_abi_version(void) { return 4; } int p252_device_count(void) { return 0; }", "4")):
        src = tmp_path / "stale.c"
        src.write_text(body)
        so = tmp_path / "libposeidon252_hip.so"
        subprocess.check_call(["gcc", "-shared", "-fPIC", "-o", str(so), str(src)])
        code = ("import sys; sys.path.insert(0, %r)\n"
                "from poseidon252_amd import _lib\n"
                "_lib.LIB_PATH = %r\n"
                "try:\n    _lib.lib()\nexcept _lib.ExtensionMissing as e:\n    print('EXTMISSING', e)\n" % (ROOT, str(so)))
        env = dict(os.environ)
        env.pop("P252_LIB_PATH", None)
        out = subprocess.check_output([sys.executable, "-c", code], env=env).decode()
        assert "EXTMISSING" in out and "ABI version %s" % what in out and "rebuild" in out, out


@pytest.mark.gpu
def test_communicator_at_one_rank_on_the_real_backend(gpu_ctx, oracle_mod):
    import torch
    import poseidon252_amd as P
    from poseidon252_amd import comm as C
    tag = P.merkle4_tag()
    n = 4 ** 7
    lv = oracle_mod.fill_random(0xC0, n)
    d = torch.from_numpy(lv.view(np.int64)).to("cuda:0")
    d_root = torch.zeros(4, dtype=torch.int64, device="cuda:0")
    exp = oracle_mod.merkle4_tree(tag, lv)[0]
    # one process per GPU: the id from rank 0, every rank joins (the exchange is the identity at one rank)
    ctx = P.Context(0)
    c = C.Comm.create_rank(ctx, 0, 1, lambda b: b)
    assert c.rank == 0 and c.size == 1
    c.merkle4_tree_sharded_device(tag, d, n, d_root)
    torch.cuda.synchronize()
    assert np.array_equal(d_root.cpu().numpy().view(np.uint64), exp)
    with pytest.raises(ValueError):  # a context belongs to one communicator
        C.Comm.create_rank(ctx, 0, 1, lambda b: b)
    with pytest.raises(ValueError):  # 4^k leaves per rank
        c.merkle4_tree_sharded_device(tag, d, 100, d_root)
    # a stream other than the default one: the all-gather is enqueued on the caller's stream
    s = torch.cuda.Stream()
    d_root.zero_()
    torch.cuda.synchronize()
    with torch.cuda.stream(s):
        c.merkle4_tree_sharded_device(tag, d, n, d_root)
    s.synchronize()
    assert np.array_equal(d_root.cpu().numpy().view(np.uint64), exp)
    c.destroy()
    # after destroying it the context may join another one; one process, an array of contexts
    cs = C.Comm.create_all([ctx])
    assert len(cs) == 1 and cs[0].size == 1
    d_root.zero_()
    cs[0].merkle4_tree_sharded_device(tag, d, n, d_root)
    torch.cuda.synchronize()
    assert np.array_equal(d_root.cpu().numpy().view(np.uint64), exp)
    cs[0].destroy()
    ctx.close()


@pytest.mark.gpu
def test_multi_device_entry_point_makes_its_own_communicator(gpu_ctx, oracle_mod):
    """contexts on distinct devices (here: one context): RCCL path, created on first use, destroyed with the context;
    contexts sharing a device: the resident variant refuses (P252_ERR_COMM), the synchronous one gathers through the host"""
    import torch
    import poseidon252_amd as P
    from poseidon252_amd import multi, comm as C
    tag = P.merkle4_tag()
    per = 4 ** 6
    lv = oracle_mod.fill_random(0xC1, 2 * per)
    d = [torch.from_numpy(lv[t * per:(t + 1) * per].view(np.int64)).to("cuda:0") for t in range(2)]
    a, b = P.Context(0), P.Context(0)
    try:
        assert np.array_equal(multi.merkle4_tree_multi_device([a], tag, d[:1], per), oracle_mod.merkle4_tree(tag, lv[:per])[0])
        d_roots = [torch.zeros(4, dtype=torch.int64, device="cuda:0")]
        C.merkle4_tree_multi_device_resident([a], tag, d[:1], per, d_roots)
        torch.cuda.synchronize()
        assert np.array_equal(d_roots[0].cpu().numpy().view(np.uint64), oracle_mod.merkle4_tree(tag, lv[:per])[0])
        # ONE context has nothing to exchange — no communicator is created for it, so it is not tied to a hidden
        # one and may join the caller's; with the caller's communicator in place the same calls take the RCCL path (real backend)
        own = C.Comm.create_all([a])
        assert np.array_equal(multi.merkle4_tree_multi_device([a], tag, d[:1], per), oracle_mod.merkle4_tree(tag, lv[:per])[0])
        d_roots[0].zero_()
        C.merkle4_tree_multi_device_resident([a], tag, d[:1], per, d_roots)
        torch.cuda.synchronize()
        assert np.array_equal(d_roots[0].cpu().numpy().view(np.uint64), oracle_mod.merkle4_tree(tag, lv[:per])[0])
        own[0].destroy()
        if torch.cuda.device_count() < 2:
            c2 = P.Context(0)
            with pytest.raises(P.DeviceError):
                C.merkle4_tree_multi_device_resident([b, c2], tag, d, per, [d_roots[0], torch.zeros(4, dtype=torch.int64, device="cuda:0")])
            assert np.array_equal(multi.merkle4_tree_multi_device([b, c2], tag, d, per), oracle_mod.merkle4_tree(tag, lv)[0])
            c2.close()
    finally:
        a.close()
        b.close()


@pytest.mark.gpu
@pytest.mark.parametrize("n_trees,per", [(1024, 4 ** 6), (1, 4 ** 5), (3, 4), (7, 16), (4096, 4 ** 3), (70000, 4), (5, 1), (2, 4 ** 9), (3001, 4 ** 5)])
def test_forest_roots_equal_per_tree_oracle_roots(gpu_ctx, oracle_mod, n_trees, per):
    import torch
    import poseidon252_amd as P
    tag = P.merkle4_tag()
    lv = oracle_mod.fill_random(0xF0 + n_trees + per, n_trees * per)
    d = torch.from_numpy(lv.view(np.int64)).to("cuda:0")
    roots, levels = P.merkle4_forest(d, per, tag=tag, ctx=gpu_ctx, want_levels=True)
    torch.cuda.synchronize()
    roots = roots.cpu().numpy().view(np.uint64)
    levels = levels.cpu().numpy().view(np.uint64)
    # host leaves, roots only: p252_merkle4_forest — the first level hashed chunk by chunk while the leaves stream in through the
    # staging lanes, the upper levels once across all trees (1,024 x 4^6 and 3,001 x 4^5 leaves are several chunks, the last one
    # ragged; the small shapes take the one-upload path)
    roots2 = P.merkle4_forest(lv, per, tag=tag, ctx=gpu_ctx)
    assert np.array_equal(roots, roots2)
    idx = sorted(set([0, n_trees - 1] + list(range(0, n_trees, max(1, n_trees // 40)))))
    for t in idx:
        if per == 1:
            assert np.array_equal(roots[t], lv[t])
            continue
        r, lvls, _ = oracle_mod.merkle4_tree(tag, lv[t * per:(t + 1) * per], want_levels=True)
        assert np.array_equal(roots[t], r), t
        # level-major layout: level l of all trees is one array, tree-major inside it
        off_forest, off_tree, width = 0, 0, per // 4
        while width >= 1:
            assert np.array_equal(levels[off_forest + t * width: off_forest + (t + 1) * width], lvls[off_tree:off_tree + width]), (t, width)
            off_forest += n_trees * width
            off_tree += width
            width //= 4
    # the forest is the first log4(per) levels of the tree over the concatenation: the tree over its roots is that tree's root
    over_roots = P.merkle4_tree(torch.from_numpy(roots.view(np.int64)).to("cuda:0").contiguous(), tag=tag, ctx=gpu_ctx)
    assert np.array_equal(over_roots.cpu().numpy(), P.merkle4_tree(d, tag=tag, ctx=gpu_ctx).cpu().numpy())


@pytest.mark.gpu
def test_forest_argument_errors(gpu_ctx):
    import torch
    import poseidon252_amd as P
    d = torch.zeros((48, 4), dtype=torch.int64, device="cuda:0")
    r = torch.zeros((4, 4), dtype=torch.int64, device="cuda:0")
    with pytest.raises(ValueError):
        gpu_ctx.merkle4_forest_device(P.merkle4_tag(), d, 4, 12, r)  # 12 is not 4^k
    with pytest.raises(ValueError):
        P.merkle4_forest(d, 5)  # 48 leaves are not whole 5-leaf trees
    gpu_ctx.merkle4_forest_device(P.merkle4_tag(), d, 0, 16, r)  # an empty forest is a no-op


def test_library_before_torch_exits_cleanly():
    """Found on the GPU box in round 4: with torch's bundled librccl pre-loaded RTLD_GLOBAL ahead of `import torch` the
    process aborted at exit ("double free or corruption").  Since ABI 8 the library does not link RCCL at all; its Python
    wrappers map torch's copy LOCALLY before the first communicator call when torch is installed but not yet imported
    (_lib.prefer_torch_rccl), and the library's resolver then takes the copy the process holds: ONE RCCL per process in either
    import order.  Reproducible without a GPU: load the library first, resolve RCCL, import torch, exit."""
    import sys
    for code in ("from poseidon252_amd import _lib, comm\n_lib.lib()\nb = comm.backend()\nimport torch\n",
                 "import torch\nfrom poseidon252_amd import _lib, comm\n_lib.lib()\nb = comm.backend()\n",
                 "from poseidon252_amd import _lib, comm\n_lib.lib()\nimport torch\nb = comm.backend()\n"):
        code += "import os\nlibs = sorted(set(l.split()[-1] for l in open('/proc/self/maps') if 'librccl' in l))\nprint([libs, os.path.realpath(b)])\n"
        r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, timeout=300)
        assert r.returncode == 0, r.stdout.decode() + r.stderr.decode()[-2000:]
        libs, backend = eval(r.stdout.decode().strip().splitlines()[-1])
        assert len(libs) == 1 and os.path.realpath(libs[0]) == backend, (libs, backend)  # torch's copy serves both
        assert os.sep + "torch" + os.sep in backend, backend


@pytest.mark.gpu
@pytest.mark.parametrize("n_trees,per", [(512, 2 ** 7), (3, 2), (70000, 4), (5, 2 ** 12)])
def test_merkle2_forest_equals_per_tree_oracle_roots(gpu_ctx, oracle_mod, n_trees, per):
    """arity 2 (Domain::Merkle2 nodes): the same entry point shape, p252_merkle2_forest_device"""
    import torch
    import poseidon252_amd as P
    tag = P.compute_tag(P.Domain.Merkle2, [2], 1)
    lv = oracle_mod.fill_random(0xF2 + n_trees + per, n_trees * per)
    d = torch.from_numpy(lv.view(np.int64)).to("cuda:0")
    roots = torch.empty((n_trees, 4), dtype=torch.int64, device="cuda:0")
    n_lv = n_trees * (per - 1)  # a complete binary tree over `per` leaves has per - 1 inner nodes
    levels = torch.empty((n_lv, 4), dtype=torch.int64, device="cuda:0")
    gpu_ctx.merkle4_forest_device(tag, d, n_trees, per, roots, levels, arity=2)
    torch.cuda.synchronize()
    roots, levels = roots.cpu().numpy().view(np.uint64), levels.cpu().numpy().view(np.uint64)
    for t in sorted(set([0, n_trees - 1] + list(range(0, n_trees, max(1, n_trees // 25))))):
        r, lvls, _ = oracle_mod.merkle2_tree(tag, lv[t * per:(t + 1) * per], want_levels=True)
        assert np.array_equal(roots[t], r), t
        off_forest, off_tree, width = 0, 0, per // 2
        while width >= 1:
            assert np.array_equal(levels[off_forest + t * width: off_forest + (t + 1) * width], lvls[off_tree:off_tree + width]), (t, width)
            off_forest += n_trees * width
            off_tree += width
            width //= 2
    with pytest.raises(ValueError):
        gpu_ctx.merkle4_forest_device(tag, d, 1, 3, torch.empty((1, 4), dtype=torch.int64, device="cuda:0"), None, arity=2)  # 3 is not 2^k
