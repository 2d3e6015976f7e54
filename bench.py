#!/usr/bin/env python3
"""bench.py — BASELINE.json's metric on MI355X: width-5 Poseidon permutations/s (= Merkle4 digests/s).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one pass of the hot path over one batch of synthetic input already resident in HBM:
the default workload is BASELINE.json configs[1] — 2^20 independent Hash::digest(Domain::Merkle4,
4 random BlsScalar) per GPU, one Hades permutation each, one kernel launch (k_merkle4).  Scaling is
weak: every rank hashes its own 2^20-digest batch, no data-path collective (digests are independent);
rank 0 broadcasts the constant table over RCCL once, before the timed region.  Setup also issues 24 untimed
launches of the step (`setup.wake_up_launches`) so that an idle GPU's clocks are at steady state whatever W is;
then W untimed warm-up steps, then exactly K timed steps between barriers + synchronisation, MAX over ranks.
Other workloads: --workload tree | sponge42 | openings | encrypt (BASELINE configs[2], [3]).

Prints ONE JSON line (rank 0) with `roofline` (VALU int32-MAC bound — DESIGN.md §3; HBM figures are
included, the path is not HBM- or MFMA-bound) and, at N=1, `cpu_baseline` (the C oracle, kind "port",
timed on the host cores on a bounded sample).  oracle/ is touched ONLY inside that cpu_baseline leg,
where it is timed and, as a by-product, checks a sample of the GPU output just measured; every run
(any N) additionally performs an oracle-free GPU self-consistency check.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic figures
MACS_PER_PERM_REFERENCE = 256_000      # 2000 field mults x 128 32x32->64 MACs (reference schedule)
BYTES_PER_PERM = {"merkle4_digests": 160.0, "tree": 96.0, "sponge42": 1504.0 / 12.0,
                  "openings": (32 + 12 * 96 + 12 + 32) / 12.0,  # leaf + 12 x 3 siblings + 12 position bytes + root
                  "encrypt": (5 * 32 + 3 * 32) / 2.0}             # 2 message + 2 secret + 1 nonce scalars in, 3 cipher scalars out
KERNEL_OF = {"merkle4_digests": "k_merkle4", "tree": "k_merkle4", "sponge42": "k_sponge", "openings": "k_merkle4_path", "encrypt": "k_crypt"}
# VALU issue peak of the chip (the binding roofline): a wave64 v_mad_i64_i32 occupies its SIMD for 4
# cycles, so 1024 SIMDs x 2.4 GHz / 4 = 614.4 G wave-instructions/s = 39.3 T lane-MACs/s.  The clock is the one
# GRBM_GUI_ACTIVE / duration shows during this very kernel (profiles/r02_pmc_k_merkle4.txt: 2.40 GHz).  For reference,
# the best a pure dependent-free multiply-add stream was MEASURED to sustain after clock ramp is lower
# (profiles/r02_valu_rates_gfx950.txt: 531 G/s at 4 waves per SIMD) — reported beside it, never used as the peak,
# because a peak the kernel's own instruction mix can exceed is not a peak.
SIMDS, NOMINAL_CLOCK_HZ = 1024, 2.4e9
PEAK_WAVE_INST_4CYCLE_PER_S = SIMDS * NOMINAL_CLOCK_HZ / 4.0
PEAK_INT32_MAC_PER_S = PEAK_WAVE_INST_4CYCLE_PER_S * 64
MEASURED_MAD_STREAM_WAVE_INST_PER_S = 531.2e9
PEAK_HBM_GBPS = 8000.0                 # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def _latest(pattern):
    import glob
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
    return paths[-1] if paths else None


def isa_counts(kernel):
    """instructions one lane executes per pass of `kernel`, derived from the ISA of the committed sources by
    tools/isa_count.py (scalar control flow interpreted, vector instructions counted; tests/test_isa_counts.py keeps the
    file in step with the kernels).  k_sponge / k_merkle4_path / k_crypt run the same permutation body as k_permute
    (all five lanes kept) once per permutation, so k_permute's figures stand for them."""
    path = _latest("r*_isa_counts.json")
    if not path:
        return None
    d = json.load(open(path))
    key = kernel if kernel in d else "k_permute"
    c = dict((k, v) for k, v in d[key].items() if k != "by_mnemonic")
    c["source"] = os.path.relpath(path, ROOT) + ":" + key
    return c


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)  # the clocks need ~10 launches (25 ms) to ramp from idle
    ap.add_argument("--workload", default="merkle4_digests", choices=["merkle4_digests", "tree", "sponge42", "openings", "encrypt"])
    ap.add_argument("--log2n", type=int, default=None, help="log2 of units per GPU per step (default: 20; tree: 24 leaves)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-check", action="store_true")
    return ap.parse_args()


def pmc_profile(kernel):
    """the committed rocprofv3 --pmc summary of `kernel` from the latest round (tools/run_pmc.sh + tools/pmc_summary.py:
    counters collected in separate passes; FETCH_SIZE / WRITE_SIZE in KB, FETCH_SIZE doubled on gfx950 as
    MI355X_MICROARCH.md §HBM prescribes) — bench.py cannot run the profiler on itself"""
    path = _latest("r*_pmc_%s.json" % kernel)
    if not path:
        return None
    try:
        d = json.load(open(path))
    except (OSError, ValueError):
        return None
    d["source"] = os.path.relpath(path, ROOT)
    return d


def pmc_traffic(kernel, workload, units_per_launch):
    """HBM bytes per launch of the dominant kernel from the committed --pmc passes, scaled to this launch size"""
    d = pmc_profile(kernel)
    if not d or "hbm_bytes_per_launch" not in d or workload == "tree":  # (a tree is 12 launches of different sizes)
        return None
    scale = units_per_launch / d["units_per_launch"]
    return {"bytes": d["hbm_bytes_per_launch"] * scale, "algorithmic_bytes": BYTES_PER_PERM[workload] * units_per_launch,
            "ratio": d["hbm_bytes_per_launch"] * scale / (BYTES_PER_PERM[workload] * units_per_launch), "source": d["source"]}


def pmc_valu(kernel):
    """what the counters say about the same kernel: VALU instructions per wave (must equal the ISA-derived count) and
    the clock during the kernel (GRBM_GUI_ACTIVE / 8 XCDs / duration)"""
    d = pmc_profile(kernel)
    if not d or "valu_insts_per_wave" not in d:
        return None
    out = {"valu_insts_per_wave": d["valu_insts_per_wave"], "source": d["source"]}
    if d.get("clock_ghz"):
        out["clock_ghz"] = d["clock_ghz"]  # GRBM_GUI_ACTIVE / 8 XCDs / median dispatch duration of the same pass
    return out


def usable_cpus():
    """threads worth starting: min(affinity mask, cgroup CPU quota) — the GPU box reports 256 logical CPUs
    but a container quota may allow far fewer"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                quota = int(txt[0])
                period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if quota > 0:
                    n = min(n, max(1, quota // period))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)


def cpu_baseline(tag, gpu_sample=None):
    """The oracle (C restatement of the reference CPU path, reference schedule: 2000 mults/perm) on the
    host cores.  Bounded sample: 2^14 digests on 1 thread, then 2^14 per thread on all threads.
    gpu_sample = (kind, inputs, in_len, out_len, gpu_output): inputs of the run just timed with the GPU's
    answers — recomputed here on the CPU and compared (reported as parity_sample_ok)."""
    import oracle
    try:  # rebuild for this host's ISA when a compiler is present (mulx/adx); fall back to the shipped build
        import subprocess
        native = os.path.join(ROOT, "oracle", "libp252_oracle_native.so")
        subprocess.check_call(["gcc", "-O3", "-march=native", "-fPIC", "-shared", "-std=gnu11", "-o", native,
                               os.path.join(ROOT, "oracle", "p252_oracle.c"), "-lpthread"],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        oracle._LIB_PATH = native
        oracle._lib = None
        oracle.lib()
        flags = "-O3 -march=native"
    except Exception:
        flags = "-O3 (shipped build)"
    threads = usable_cpus()
    n1 = 1 << 12
    x1 = oracle.fill_random(0xc10d, 4 * n1).reshape(n1, 4, 4)
    oracle.hash_batch(tag, x1, 4, 1)  # warm-up
    one = []
    for _ in range(10):  # criterion-style: 10 samples (benches/hash.rs:95), median and min reported
        t0 = time.perf_counter()
        oracle.hash_batch(tag, x1, 4, 1)
        one.append(time.perf_counter() - t0)
    t1 = float(np.median(one))
    # the quota is not always visible: probe a few thread counts on a small sample and keep the fastest
    cands = sorted(set([threads] + [c for c in (8, 16, 32, 64, 128) if c <= (os.cpu_count() or 1)]))
    probe = {}
    for c in cands:
        xs = np.tile(x1[: 1 << 11], (c, 1, 1))
        t0 = time.perf_counter()
        oracle.hash_batch(tag, xs, 4, 1, threads=c)
        probe[c] = xs.shape[0] / (time.perf_counter() - t0)
    threads = max(probe, key=probe.get)
    nall = 4 * n1 * threads
    xall = np.tile(x1, (4 * threads, 1, 1))
    oracle.hash_batch(tag, xall, 4, 1, threads=threads)  # warm-up
    allt = []
    for _ in range(10):
        t0 = time.perf_counter()
        oracle.hash_batch(tag, xall, 4, 1, threads=threads)
        allt.append(time.perf_counter() - t0)
    best = float(np.median(allt))
    cpu_model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu_model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    parity = None
    if gpu_sample is not None:
        kind, inp, in_len, out_len, got = gpu_sample
        if kind == "tree":
            exp = oracle.merkle4_tree(tag, inp)[0]
        elif kind == "encrypt":
            exp = oracle.encrypt_batch(tag, np.ascontiguousarray(inp[0]), np.ascontiguousarray(inp[1]), np.ascontiguousarray(inp[2]))
        elif kind == "paths":
            exp = oracle.merkle4_path_batch(tag, np.ascontiguousarray(inp[0]), np.ascontiguousarray(inp[1]), np.ascontiguousarray(inp[2]))
        else:
            exp = oracle.hash_batch(tag, inp, in_len, out_len)
        parity = bool(np.array_equal(np.asarray(got).reshape(-1), np.asarray(exp).reshape(-1)))
    return {"parity_sample_ok": parity,
            "value": nall / best, "unit": "permutations/s", "cores": threads, "kind": "port",
            "sample": "Hash::digest(Merkle4, 4 scalars): %d digests per sample on %d threads, 1 thread: %d digests per sample; "
                      "warm-up + 10 samples each, median reported (min in *_min)" % (nall, threads, n1),
            "value_min_time": nall / float(np.min(allt)), "value_1core": n1 / t1, "value_1core_min_time": n1 / float(np.min(one)),
            "cpu": cpu_model, "compiler": "gcc " + flags,
            "note": "C restatement of the reference CPU path (oracle/p252_oracle.c, reference schedule); "
                    "the Rust reference cannot be built here (no cargo; un-vendored crates)"}


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch(n):
    """`python bench.py --gpus N` (N > 1) WITHOUT a launcher: re-execute this command line under
    torch.distributed.run with N ranks on this node (exactly how the driver launches the N > 1 runs) and
    hand its output and exit status through.  --gpus is therefore never silently ignored."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: required for RCCL across processes on this driver
    sys.exit(subprocess.call(cmd, env=env))


def main():
    args = parse()
    if args.gpus < 1:
        print("bench.py: --gpus must be >= 1", file=sys.stderr)
        sys.exit(2)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        self_launch(args.gpus)
    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        print("bench.py: --gpus %d but WORLD_SIZE=%d — launch as `python -m torch.distributed.run --nproc-per-node %d ... "
              "bench.py --gpus %d` (or plain `python bench.py --gpus %d`, which does that itself)"
              % (args.gpus, world, args.gpus, args.gpus, args.gpus), file=sys.stderr)
        sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py needs a GPU: the product has no CPU path", file=sys.stderr)
        sys.exit(2)
    # test-only switches (tests/test_bench_multiproc.py): run N ranks on ONE GPU with gloo collectives,
    # to exercise the N > 1 code path where only a single GPU exists.  The driver never sets them.
    share_gpu = os.environ.get("P252_BENCH_SHARE_GPU") == "1"
    backend = os.environ.get("P252_BENCH_BACKEND", "nccl")
    if share_gpu:
        local_rank = 0
    elif world > torch.cuda.device_count():
        print("bench.py: --gpus %d but only %d HIP device(s) visible" % (world, torch.cuda.device_count()), file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    coll_dev = dev if backend == "nccl" else torch.device("cpu")
    import torch.distributed as dist
    force_dist = os.environ.get("P252_BENCH_FORCE_DIST") == "1"  # test-only: init the process group at N=1 too
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    import poseidon252_amd as P
    from poseidon252_amd import distributed as D
    ctx = P.Context(local_rank)
    tables_identical = D.broadcast_tables(ctx, device=coll_dev)  # RCCL broadcast of the constants (no-op at N=1)

    wl = args.workload
    if wl == "merkle4_digests":
        log2n = args.log2n or 20
        n = 1 << log2n
        hb = P.HashBatch(P.Domain.Merkle4, 4, ctx=ctx)
        in_scalars, perms_per_step = 4 * n, n
        name = "2^%d independent Merkle4 digests per GPU (BASELINE configs[1])" % log2n
    elif wl == "tree":
        log2n = args.log2n or 24
        n = 1 << log2n
        hb = P.HashBatch(P.Domain.Merkle4, 4, ctx=ctx)
        in_scalars, perms_per_step = n, P.levels_len(n)
        name = "2^%d-leaf arity-4 Merkle tree per GPU, all levels (BASELINE configs[2])" % log2n
    elif wl == "encrypt":
        log2n = args.log2n or 20
        n = 1 << log2n
        hb = P.HashBatch(P.Domain.Merkle4, 4, ctx=ctx)  # (only for the context; the tag below is the encryption tag)
        in_scalars, perms_per_step = 5 * n, 2 * n
        name = "2^%d encryptions of 2-scalar messages per GPU (2 permutations each, SURVEY §8 f4; recipe unpinned)" % log2n
    elif wl == "openings":
        log2n = args.log2n or 20
        n = 1 << log2n
        depth = 12
        hb = P.HashBatch(P.Domain.Merkle4, 4, ctx=ctx)
        in_scalars, perms_per_step = n * (1 + 3 * depth), depth * n
        name = "2^%d Merkle4 openings of depth %d per GPU (branch re-hash, SURVEY §8 f3)" % (log2n, depth)
    else:
        log2n = args.log2n or 20
        n = 1 << log2n
        hb = P.HashBatch(P.Domain.Other, 42, output_len=5, ctx=ctx)
        in_scalars, perms_per_step = 42 * n, 12 * n
        name = "Domain::Other sponge, 2^%d messages x 42 scalars -> 5 outputs per GPU (BASELINE configs[3])" % log2n
    tag = hb.tag
    if wl == "encrypt":
        from poseidon252_amd import encryption as E
        tag = E.encryption_tag(2)

    # synthetic input generated ON the device by SURVEY §8(d)'s generator (splitmix64 stream, rejection-sampled below p:
    # uniform field elements; poseidon252_amd/synth.py, byte-identical to the oracle's fill_random).  Rank 0's
    # configs[1] batch (seed 0xc10d) is exactly the one tests/test_gpu_fullsize.py verifies digest by digest.
    from poseidon252_amd import synth
    g = torch.Generator(device=dev)
    g.manual_seed(0xC10D + rank)
    d_in = synth.splitmix_scalars(0xC10D + rank, in_scalars, dev)
    if wl == "merkle4_digests":
        d_out = torch.empty((n, 4), dtype=torch.int64, device=dev)
        step = lambda: ctx.hash_batch_device(tag, d_in, 4, 1, d_out, n)
    elif wl == "tree":
        d_out = torch.empty(4, dtype=torch.int64, device=dev)
        if world == 1:
            step = lambda: ctx.merkle4_tree_device(tag, d_in, n, d_out, None)
        else:
            # BASELINE configs[4] structure: every rank reduces its complete subtree, the W roots (32 B each)
            # are all-gathered (the path's only exchange step) and the top levels are hashed on every rank
            d_roots = torch.empty(world * 4, dtype=torch.int64, device=coll_dev)  # flat: gloo and nccl both accept it
            d_top = torch.empty(4, dtype=torch.int64, device=dev)

            def step():
                ctx.merkle4_tree_device(tag, d_in, n, d_out, None)
                dist.all_gather_into_tensor(d_roots, d_out if coll_dev == dev else d_out.cpu())
                roots_dev = d_roots if coll_dev == dev else d_roots.to(dev)
                ctx.merkle4_tree_device(tag, roots_dev.contiguous(), world, d_top, None)
            perms_per_step += P.levels_len(world)
            name += " + all-gather of %d subtree roots and top levels" % world
            if world == 8 and log2n == 24:
                name += " = 2^27-leaf tree sharded across 8 GPUs (BASELINE configs[4])"
        step()  # allocate the context-owned level scratch outside the timed region
    elif wl == "encrypt":
        d_out = torch.empty((n, 3, 4), dtype=torch.int64, device=dev)
        d_msgs, d_secrets, d_nonces = d_in[:2 * n], d_in[2 * n:4 * n], d_in[4 * n:]
        step = lambda: E.encrypt_batch_device(d_msgs, d_secrets, d_nonces, 2, d_out, n, ctx=ctx, tag=tag)
    elif wl == "openings":
        d_out = torch.empty((n, 4), dtype=torch.int64, device=dev)
        d_leaves, d_sibs = d_in[:n], d_in[n:]
        d_pos = torch.randint(0, 4, (n, depth), dtype=torch.uint8, device=dev, generator=g)
        step = lambda: ctx.merkle4_path_batch_device(tag, d_leaves, d_sibs, d_pos, depth, d_out, n)
    else:
        d_out = torch.empty((n, 5, 4), dtype=torch.int64, device=dev)
        step = lambda: ctx.hash_batch_device(tag, d_in, 42, 5, d_out, n)

    def barrier():
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    # Device wake-up (setup, reported in the JSON line): an idle MI355X needs ~25 ms of activity before its clocks
    # reach the steady state a running service sees (profiles/r01_bench_kernel_trace_v9.txt: the first ten launches are up
    # to 25 % slower).  Done here so that the measurement does not depend on how many warm-up steps the caller asks
    # for; the W warm-up steps and the K timed steps below are untouched.
    WAKE_UP_LAUNCHES = 24  # a fixed count (not a time): every rank must issue the same collectives at N > 1
    for _ in range(WAKE_UP_LAUNCHES):
        step()
    torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    barrier()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    evs[0].record()
    for i in range(args.steps):
        step()  # launched on torch's current stream; the events below are recorded on that same stream
        evs[i + 1].record()
    barrier()
    elapsed = time.perf_counter() - t0
    launch_ms = [evs[i].elapsed_time(evs[i + 1]) for i in range(args.steps)]
    if dist.is_initialized():
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # correctness of what was just timed, WITHOUT the oracle (every rank, every N): a slice of the batch is
    # hashed again on its own and must reproduce the same digests (shard consistency + determinism)
    self_ok = None
    if not args.no_check:
        if wl == "merkle4_digests":
            lo, cnt = n // 3, min(n - n // 3, 4096)
            again = torch.empty((cnt, 4), dtype=torch.int64, device=dev)
            ctx.hash_batch_device(tag, d_in[lo * 4:(lo + cnt) * 4], 4, 1, again, cnt)
            torch.cuda.synchronize()
            self_ok = bool(torch.equal(again, d_out[lo:lo + cnt]))
        elif wl == "tree":
            # subtree composition: the tree over the 4 quarter-tree roots is the tree's root
            quarters = torch.stack([P.merkle4_tree(d_in[i * (n // 4):(i + 1) * (n // 4)], tag=tag, ctx=ctx) for i in range(4)])
            top = P.merkle4_tree(quarters.contiguous(), tag=tag, ctx=ctx)
            ref = torch.empty(4, dtype=torch.int64, device=dev)
            ctx.merkle4_tree_device(tag, d_in, n, ref, None)
            torch.cuda.synchronize()
            self_ok = bool(torch.equal(top, ref))
        elif wl == "encrypt":
            # decrypting what was just produced gives the messages back, with every authentication flag set
            back = torch.empty((n, 2, 4), dtype=torch.int64, device=dev)
            ok = torch.zeros(n, dtype=torch.uint8, device=dev)
            E.decrypt_batch_device(d_out, d_secrets, d_nonces, 2, back, ok, n, ctx=ctx, tag=tag)
            torch.cuda.synchronize()
            self_ok = bool(torch.equal(back.view(-1, 4), d_msgs) and bool(ok.all()))
        elif wl == "openings":
            lo, cnt = n // 3, min(n - n // 3, 4096)
            again = torch.empty((cnt, 4), dtype=torch.int64, device=dev)
            ctx.merkle4_path_batch_device(tag, d_leaves[lo:lo + cnt], d_sibs[lo * 3 * depth:(lo + cnt) * 3 * depth], d_pos[lo:lo + cnt].contiguous(),
                                          depth, again, cnt)
            torch.cuda.synchronize()
            self_ok = bool(torch.equal(again, d_out[lo:lo + cnt]))
        else:
            lo, cnt = n // 3, min(n - n // 3, 1024)
            again = torch.empty((cnt, 5, 4), dtype=torch.int64, device=dev)
            ctx.hash_batch_device(tag, d_in[lo * 42:(lo + cnt) * 42], 42, 5, again, cnt)
            torch.cuda.synchronize()
            self_ok = bool(torch.equal(again, d_out[lo:lo + cnt]))
        if not self_ok:
            print("SELF-CONSISTENCY FAILURE on rank %d" % rank, file=sys.stderr)
            sys.exit(3)

    if rank == 0:
        total_perms = perms_per_step * args.steps * world
        value = total_perms / elapsed
        k_ms = float(np.mean(launch_ms))
        per_gpu_rate = perms_per_step / (k_ms * 1e-3)
        achieved_mac = per_gpu_rate * MACS_PER_PERM_REFERENCE
        hbm_gbps = per_gpu_rate * BYTES_PER_PERM[wl] / 1e9
        kern = KERNEL_OF[wl]
        lanes_per_perm, isa_key, pmc_key = 1, None, None
        coop_max = int(os.environ.get("P252_COOP_MAX_NODES", "16384"))
        items = n  # independent states per launch (digests, messages, openings)
        if wl == "merkle4_digests" and 8192 < items <= min(coop_max, 16384):
            kern, lanes_per_perm, isa_key, pmc_key = "k_merkle4_coop<4>", 4, "k_merkle4_coop<4>", "k_merkle4_coop4"
        elif wl in ("merkle4_digests", "sponge42", "openings", "encrypt") and items <= min(coop_max, 8192):
            # batches this small run the lane-group kernels: eight lanes per state (csrc/coop29.hpp); the permutation body is
            # the one of k_merkle4_coop<8>, whose ISA counts and counter passes stand for all of them
            kern = {"merkle4_digests": "k_merkle4_coop<8>", "sponge42": "k_sponge_coop", "openings": "k_merkle4_path_coop", "encrypt": "k_crypt_coop"}[wl]
            lanes_per_perm, isa_key, pmc_key = 8, "k_merkle4_coop<8>", "k_merkle4_coop8"
        isa = isa_counts(isa_key or kern)
        executed = issue = None
        if isa:
            mac_rate = per_gpu_rate * lanes_per_perm * isa["v_mad_i64_i32"]
            executed = {"macs_per_perm": isa["v_mad_i64_i32"] * lanes_per_perm, "achieved": mac_rate / 1e12, "peak": PEAK_INT32_MAC_PER_S / 1e12,
                        "frac": mac_rate / PEAK_INT32_MAC_PER_S, "unit": "TMAC/s", "source": isa["source"],
                        "frac_of_measured_mad_stream": mac_rate / (MEASURED_MAD_STREAM_WAVE_INST_PER_S * 64),
                        "note": "the fraction of the hardware: multiply-adds actually issued (counted in the ISA) / (1024 SIMDs x 2.4 GHz / 4 cycles x 64 lanes)"}
            # every VALU instruction priced at its issue cost (4 cycles for multiply-adds, 64-bit adds / shifts and VOP3
            # 3-operand forms, 2 for plain 32-bit ops — profiles/r02_valu_rates_gfx950.txt): share of all SIMD cycles
            cyc = per_gpu_rate * lanes_per_perm / 64.0 * isa["valu_issue_cycles"]
            issue = {"valu_insts_per_perm": isa["valu_total"], "lanes_per_perm": lanes_per_perm, "issue_cycles_per_perm": isa["valu_issue_cycles"],
                     "frac": cyc / (SIMDS * NOMINAL_CLOCK_HZ), "pmc": pmc_valu(pmc_key or kern),
                     "note": "SIMD cycles spent issuing VALU work under the 4-/2-cycle model / all SIMD cycles at 2.4 GHz"}
        roofline = {
            "bound": "valu-int32-mac", "kernel": kern,
            "achieved": achieved_mac / 1e12, "peak": PEAK_INT32_MAC_PER_S / 1e12, "unit": "TMAC/s",
            "frac": achieved_mac / PEAK_INT32_MAC_PER_S,
            "note": "SURVEY §8d figure: 256,000 MACs per permutation (reference schedule) x permutations per launch / mean launch time "
                    "(HIP events on the launch stream) against the VALU issue peak.  The kernel runs an algebraically equivalent schedule with "
                    "4x fewer multiply-adds, so this exceeds 1 and says nothing about the hardware; `executed.frac` does.",
            "executed": executed, "valu_issue": issue,
            "launch_ms_mean": k_ms, "launch_ms_min": float(np.min(launch_ms)),
            "hbm": {"achieved": hbm_gbps, "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": hbm_gbps / PEAK_HBM_GBPS,
                    "algorithmic_bytes_per_perm": BYTES_PER_PERM[wl]},
            "traffic": pmc_traffic(pmc_key or kern, wl, perms_per_step),
        }
        line = {
            "metric": "Poseidon width-5 permutations/s (= Merkle4 digests/s), bit-exact",
            "value": value, "unit": "permutations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "int64", "data": "synthetic",
            "dtype_note": "255-bit field elements as 9 x 29-bit limbs (int32), products accumulated in signed 64-bit columns (v_mad_i64_i32)",
            "config": {"workload": name, "units_per_gpu_per_step": perms_per_step, "sharding": "independent batches per GPU, no data-path collective",
                       "ranks": dist.get_world_size() if dist.is_initialized() else 1,
                       "collective_backend": (dist.get_backend() if dist.is_initialized() else None),
                       "input": "splitmix64 seed 0xc10d + rank, uniform mod p (SURVEY §8d)",
                       "constants": "RCCL broadcast from rank 0 (identical to local derivation: %s)" % tables_identical},
            "roofline": roofline,
            # the same kernel priced against the HBM roofline in the contract's shape (NOT the binding bound here)
            "roofline_hbm": {"bound": "hbm", "achieved": hbm_gbps, "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                             "frac": hbm_gbps / PEAK_HBM_GBPS, "traffic": roofline["traffic"]},
            "self_consistency_ok": self_ok,
            "setup": {"wake_up_launches": WAKE_UP_LAUNCHES,
                      "note": "untimed launches of the same step before the W warm-up steps: brings an idle GPU's clocks to steady state"},
        }
        if world == 1 and not args.no_cpu_baseline:
            # the cpu_baseline leg is the ONLY place bench.py touches oracle/: it times the CPU restatement and,
            # as a by-product, checks a strided sample of the GPU output just measured against it
            sample = None
            if not args.no_check:
                h_in = d_in.cpu().numpy().view(np.uint64)
                if wl == "merkle4_digests":
                    idx = np.arange(0, n, max(1, n // 512))
                    sample = ("hash", h_in.reshape(n, 4, 4)[idx], 4, 1, d_out.cpu().numpy().view(np.uint64)[idx].reshape(-1, 1, 4))
                elif wl == "tree":
                    sub = 1 << 12
                    got = P.merkle4_tree(d_in[:sub].contiguous(), tag=tag, ctx=ctx).cpu().numpy().view(np.uint64)
                    sample = ("tree", h_in[:sub], None, None, got)
                elif wl == "encrypt":
                    idx = np.arange(0, n, max(1, n // 128))
                    sample = ("encrypt", (h_in[:2 * n].reshape(n, 2, 4)[idx], h_in[2 * n:4 * n].reshape(n, 2, 4)[idx], h_in[4 * n:][idx]), None, None,
                              d_out.cpu().numpy().view(np.uint64)[idx])
                elif wl == "openings":
                    idx = np.arange(0, n, max(1, n // 128))
                    sample = ("paths", (h_in[:n][idx], h_in[n:].reshape(n, depth, 3, 4)[idx], d_pos.cpu().numpy()[idx]), None, None,
                              d_out.cpu().numpy().view(np.uint64)[idx])
                else:
                    idx = np.arange(0, n, max(1, n // 128))
                    sample = ("hash", h_in.reshape(n, 42, 4)[idx], 42, 5, d_out.cpu().numpy().view(np.uint64).reshape(n, 5, 4)[idx])
            line["cpu_baseline"] = cpu_baseline(tag, sample)
            if line["cpu_baseline"].get("parity_sample_ok") is False:
                print("PARITY FAILURE: GPU output differs from the oracle", file=sys.stderr)
                sys.exit(3)
        print(json.dumps(line))
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
