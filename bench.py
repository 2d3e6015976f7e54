#!/usr/bin/env python3
"""bench.py — BASELINE.json's metric on MI355X: width-5 Poseidon permutations/s (= Merkle4 digests/s).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one pass of the hot path over one batch of synthetic input already resident in HBM:
the PRIMARY workload (the top-level keys of the JSON line) is BASELINE.json configs[1] — 2^20 independent
Hash::digest(Domain::Merkle4, 4 random BlsScalar) per GPU, one Hades permutation each, one kernel launch
(k_merkle4).  Scaling is weak: every rank hashes its own 2^20-digest batch, no data-path collective (digests
are independent); rank 0 broadcasts the constant table over RCCL once, before the timed region.  Per workload:
untimed wake-up launches of the step (so that an idle GPU's clocks are at steady state whatever W is), W untimed
warm-up steps, then exactly K timed steps between barriers + synchronisation, MAX over ranks.

The same command then times the other BASELINE configs as SECONDARY workloads, reported under "secondary" with the
same discipline (their own wake-up, warm-up, barriers, max over ranks) — so that every config is measured by whoever
runs this one command:
  secondary.tree      BASELINE configs[2]: 2^24-leaf arity-4 Merkle tree per GPU, all levels; at N > 1 every rank
                      reduces its subtree, the N roots (32 B each) are all-gathered — the path's only exchange step —
                      and the top levels are hashed on every rank: at N = 8 that IS configs[4] (2^27 leaves)
  secondary.forest    the same 2^24 leaves per GPU as 4,096 independent trees of 4^6 leaves (p252_merkle4_forest_device: one launch per
                      level across all trees — the downstream poseidon-merkle shape; no narrow levels left to wait for)
  secondary.sponge42  BASELINE configs[3]: Domain::Other sponge, 2^20 messages x 42 scalars -> 5 outputs per GPU
  secondary.openings  SURVEY §8 f3: 2^20 Merkle4 openings of depth 12 per GPU (branch re-hash, k_merkle4_path_lines)
  secondary.extract   SURVEY §8 f3: 2^20 openings of depth 12 EXTRACTED from a stored 2^24-leaf tree per GPU (k_merkle4_openings: data movement,
                      no hashing) — the one workload priced against the HBM roofline ("bound": "hbm", GB/s against 8 TB/s)
  secondary.encrypt   SURVEY §8 f4: 2^20 encryptions of 2-scalar messages per GPU (k_crypt; construction unpinned)
(--no-secondary skips them; --workload X makes X the primary and runs no secondary; --log2n scales the primary.)

The shader clock is MEASURED inside the run: a one-wave probe kernel (p252_clock_probe_device: s_memtime
against the 100 MHz s_memrealtime) runs on a second stream beside extra untimed steps immediately before and after
each timed region — the clock under this very load — and every roofline fraction is given at the nominal 2.4 GHz
AND at the measured clock.

Prints ONE JSON line (rank 0) of SCALARS ONLY, under 6,000 bytes (the driver keeps an 8 KB tail of stdout; round 4's 26 KB
line could not be parsed): the contract's keys, `roofline`, `cpu_baseline` and each secondary as a handful of scalars.  The full
record — nested models (`executed`, `valu_issue`, `clock`, `hbm`, `traffic_detail`), per-rank arrays, sources and notes — is written
to bench_detail.json next to this file (`detail_file` on the line; --detail-file).  As soon as the primary is measured a minimal
record of it also goes to stderr ("bench.py primary: {...}").

The line carries `roofline` (VALU int32-MAC bound — DESIGN.md §3; HBM figures are
included, the path is not HBM- or MFMA-bound) and, at N=1, `cpu_baseline` (the C oracle, kind "port",
timed on the host cores on a bounded sample).  oracle/ is touched ONLY inside that cpu_baseline leg,
where it is timed and, as a by-product, checks a sample of the GPU output of every workload just measured; every
run (any N) additionally performs an oracle-free GPU self-consistency check per workload.
"""
import argparse
import hashlib
import json
import os
import sys
import time

T_START = time.time()  # (stderr progress stamps: "N s since start")

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic figures
MACS_PER_PERM_REFERENCE = 256_000      # 2000 field mults x 128 32x32->64 MACs (reference schedule)
BYTES_PER_PERM = {"merkle4_digests": 160.0, "tree": 96.0, "forest": (4096 * 32 + 32) / 1365.0, "sponge42": 1504.0 / 12.0,
                  "openings": (32 + 12 * 96 + 12 + 32) / 12.0,  # leaf + 12 x 3 siblings + 12 position bytes + root
                  "encrypt": (5 * 32 + 3 * 32) / 2.0}             # 2 message + 2 secret + 1 nonce scalars in, 3 cipher scalars out
KERNEL_OF = {"merkle4_digests": "k_merkle4", "tree": "k_merkle4", "forest": "k_merkle4", "extract": "k_merkle4_openings", "sponge42": "k_sponge", "openings": "k_merkle4_path", "encrypt": "k_crypt"}
# VALU issue peak of the chip (the binding roofline): a wave64 v_mad_i64_i32 occupies its SIMD for 4
# cycles, so 1024 SIMDs x clock / 4 wave-instructions/s; x 64 lanes = lane-MACs/s.  At the nominal 2.4 GHz that is
# 614.4 G wave-instructions/s = 39.3 T lane-MACs/s — `peak`.  The clock the chip actually holds under this load is lower
# and differs from box to box (2.26-2.37 GHz seen): it is measured in the run and the fractions are given at it as well.
# For reference, the best a pure dependent-free multiply-add stream was MEASURED to sustain after clock ramp
# (profiles/r02_valu_rates_gfx950.txt: 531 G/s at 4 waves per SIMD) is reported beside it, never used as the peak,
# because a peak the kernel's own instruction mix can exceed is not a peak.
SIMDS, NOMINAL_CLOCK_HZ = 1024, 2.4e9
PEAK_WAVE_INST_4CYCLE_PER_S = SIMDS * NOMINAL_CLOCK_HZ / 4.0
PEAK_INT32_MAC_PER_S = PEAK_WAVE_INST_4CYCLE_PER_S * 64
MEASURED_MAD_STREAM_WAVE_INST_PER_S = 531.2e9
PEAK_HBM_GBPS = 8000.0                 # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
# the kernel sources a counter pass or an ISA count belongs to (profiles/*.json carry the digest of these files)
KERNEL_SOURCES = ("kernels.hip", "fr29.hpp", "hades29.hpp", "coop29.hpp", "tables.hpp", "kernels.h")


def kernel_sources_sha256(kernel=None):
    """digest of the sources `kernel` is compiled from: the hashing kernels' files; the extraction kernel's own file on top for it
    (a change of csrc/openings.hip makes only ITS counter passes stale)"""
    h = hashlib.sha256()
    for f in KERNEL_SOURCES + (("openings.hip", "fastdiv.hpp") if kernel and "openings" in kernel else ()):
        h.update(open(os.path.join(ROOT, "poseidon252_amd", "csrc", f), "rb").read())
    return h.hexdigest()


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
    ap.add_argument("--workload", default=None, choices=["merkle4_digests", "tree", "forest", "sponge42", "openings", "encrypt", "extract"],
                    help="primary workload (default merkle4_digests = BASELINE configs[1], followed by the secondary workloads)")
    ap.add_argument("--log2n", type=int, default=None, help="log2 of units per GPU per step of the primary (default: 20; tree: 24 leaves)")
    ap.add_argument("--no-secondary", action="store_true", help="do not time the secondary workloads (tree, sponge42)")
    ap.add_argument("--secondary-log2n", type=int, default=None, help="(tests) scale the secondary workloads: 2^k sponge messages, 2^(k+4) tree leaves")
    ap.add_argument("--secondary-timeout", type=int, default=240, help="N > 1: seconds a secondary workload may take before the line is printed without it (0 = no watchdog)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--detail-file", default=os.path.join(ROOT, "bench_detail.json"),
                    help="where rank 0 writes the full record (nested models, sources, per-rank arrays, notes); the printed line names it")
    return ap.parse_args()


def pmc_profile(kernel):
    """the committed rocprofv3 --pmc summary of `kernel` from the latest round (tools/run_pmc.sh + tools/pmc_summary.py:
    counters collected in separate passes; FETCH_SIZE / WRITE_SIZE in KB, FETCH_SIZE doubled on gfx950 as
    MI355X_MICROARCH.md §HBM prescribes) — bench.py cannot run the profiler on itself.  A summary records the digest of
    the kernel sources it was collected from; one that belongs to other sources is reported as stale, never silently used
    (tests/test_profiles_fresh.py keeps the default line's files fresh)."""
    path = _latest("r*_pmc_%s.json" % kernel)
    if not path:
        return None
    try:
        d = json.load(open(path))
    except (OSError, ValueError):
        return None
    d["source"] = os.path.relpath(path, ROOT)
    d["stale"] = d.get("kernel_sources_sha256") != kernel_sources_sha256(kernel)
    return d


def pmc_traffic(kernel, workload, units_per_launch):
    """HBM bytes per step from the committed --pmc passes of the kernel(s) the workload runs, scaled to this size"""
    # (a forest is built level by level like a tree: per permutation it moves what the tree's passes show)
    d = pmc_profile(kernel if workload not in ("tree", "forest") else "tree")
    if not d or "hbm_bytes_per_launch" not in d:
        return None
    if d["stale"]:
        return {"bytes": None, "stale_source": d["source"], "note": "the committed counter pass belongs to other kernel sources"}
    scale = units_per_launch / d["units_per_launch"]
    out = {"bytes": d["hbm_bytes_per_launch"] * scale, "algorithmic_bytes": BYTES_PER_PERM[workload] * units_per_launch,
           "ratio": d["hbm_bytes_per_launch"] * scale / (BYTES_PER_PERM[workload] * units_per_launch), "source": d["source"]}
    if workload in ("tree", "forest"):
        # SURVEY §8d's figure counts the leaves in and the root out; a level-by-level build also writes every level and reads it
        # back (nodes x 64 B): that is what the counters must be compared with to see wasted re-reads
        lbl = BYTES_PER_PERM[workload] * units_per_launch + 64.0 * (units_per_launch - 1)
        out["level_by_level_bytes"] = lbl
        out["ratio_level_by_level"] = out["bytes"] / lbl
        out["note"] = "ratio is against SURVEY's leaves-in + root-out figure; the build is level by level, which writes and re-reads every level"
    return out


def pmc_valu(kernel):
    """what the counters say about the same kernel: VALU instructions per wave (must equal the ISA-derived count) and
    the clock during the kernel (GRBM_GUI_ACTIVE / 8 XCDs / duration)"""
    d = pmc_profile(kernel)
    if not d or "valu_insts_per_wave" not in d or d["stale"]:
        return None
    out = {"valu_insts_per_wave": d["valu_insts_per_wave"], "source": d["source"]}
    if d.get("clock_ghz"):
        out["clock_ghz"] = d["clock_ghz"]  # GRBM_GUI_ACTIVE / 8 XCDs / median dispatch duration of the same pass
    return out


def cpu_quota():
    """CPUs the cgroup grants (None = unlimited)"""
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                return None if txt[0] == "max" else int(txt[0]) / int(txt[1])
            quota = int(txt[0])
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            return quota / period if quota > 0 else None
        except (OSError, ValueError, IndexError):
            continue
    return None


def usable_cpus():
    """threads worth starting: min(affinity mask, cgroup CPU quota) — the GPU box reports 256 logical CPUs
    but a container quota may allow far fewer"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    q = cpu_quota()
    if q:
        n = min(n, max(1, int(q)))
    return max(1, n)


def reference_cargo_bench():
    """the reference's OWN CPU figure: `cargo bench --features=zk --bench hash -- "hash 4 BlsScalar"`
    (/root/reference/benches/hash.rs:68-72,93-97; Cargo.toml:50-53) — probed at run time on this box, not assumed: it needs cargo +
    rustc, the reference tree (P252_REFERENCE_DIR, default /root/reference; it does not travel to the GPU box) and the dusk
    crates resolvable offline (a vendored registry).  Returns the criterion figure when all of that is present, otherwise what
    was missing."""
    import shutil
    import subprocess
    probe = {"cargo": shutil.which("cargo"), "rustc": shutil.which("rustc")}
    ref = os.environ.get("P252_REFERENCE_DIR", "/root/reference")
    probe["reference_dir"] = ref if os.path.exists(os.path.join(ref, "Cargo.toml")) else None
    reg = [d for d in (os.path.join(os.path.expanduser("~"), ".cargo", "registry"), os.path.join(ref, "vendor"),
                       os.path.join(ROOT, "bindings", "rust", "vendor")) if os.path.isdir(d)]
    probe["crate_registry"] = reg[0] if reg else None
    missing = [k for k, v in probe.items() if not v]
    out = {"available": False, "probe": probe, "command": 'cargo bench --features=zk --bench hash -- "hash 4 BlsScalar"'}
    if missing:
        out["why"] = "not on this box: " + ", ".join(missing)
        return out
    try:
        r = subprocess.run(["cargo", "bench", "--offline", "--features=zk", "--bench", "hash", "--", "hash 4 BlsScalar"], cwd=ref,
                           capture_output=True, timeout=900, env=dict(os.environ, CARGO_TARGET_DIR=os.path.join("/tmp", "p252_ref_target")))
    except (OSError, subprocess.TimeoutExpired) as e:
        out["why"] = "cargo bench did not finish: %r" % (e,)
        return out
    import re
    m = re.search(r"time:\s*\[\s*([0-9.]+)\s*(ns|µs|us|ms)\s+([0-9.]+)\s*(ns|µs|us|ms)\s+([0-9.]+)\s*(ns|µs|us|ms)", r.stdout.decode(errors="replace"))
    if r.returncode != 0 or not m:
        out["why"] = "cargo bench failed (rc %d): %s" % (r.returncode, r.stderr.decode(errors="replace")[-300:])
        return out
    scale = {"ns": 1e-9, "us": 1e-6, "µs": 1e-6, "ms": 1e-3}
    med = float(m.group(3)) * scale[m.group(4)]
    return {"available": True, "probe": probe, "command": out["command"], "seconds_per_hash_median": med, "value": 1.0 / med,
            "unit": "permutations/s", "cores": 1, "kind": "reference"}


def cpu_baseline(tag, gpu_samples=()):
    """The oracle (C restatement of the reference CPU path, reference schedule: 2000 mults/perm) on the
    host cores.  Bounded sample: 2^12 digests on 1 thread, then 2^14 per thread on all threads.
    gpu_samples = [(name, kind, tag, inputs, in_len, out_len, gpu_output)]: inputs of the runs just timed with the GPU's
    answers — recomputed here on the CPU and compared (reported as parity_sample_ok, per workload in parity_samples)."""
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
    parity = {}
    for name, kind, stag, inp, in_len, out_len, got in gpu_samples:
        if kind == "tree":
            exp = oracle.merkle4_tree(stag, inp)[0]
        elif kind == "forest":  # inp = (trees, leaves_per_tree, 4): one oracle tree each
            exp = np.stack([oracle.merkle4_tree(stag, t)[0] for t in inp])
        elif kind == "encrypt":
            exp = oracle.encrypt_batch(stag, np.ascontiguousarray(inp[0]), np.ascontiguousarray(inp[1]), np.ascontiguousarray(inp[2]))
        elif kind == "paths":
            exp = oracle.merkle4_path_batch(stag, np.ascontiguousarray(inp[0]), np.ascontiguousarray(inp[1]), np.ascontiguousarray(inp[2]))
        else:
            exp = oracle.hash_batch(stag, inp, in_len, out_len)
        parity[name] = bool(np.array_equal(np.asarray(got).reshape(-1), np.asarray(exp).reshape(-1)))
    quota = cpu_quota()
    granted = usable_cpus()  # CPUs this process can actually run on at once: min(affinity mask, cgroup quota)
    return {"parity_sample_ok": (all(parity.values()) if parity else None), "parity_samples": parity,
            "value": nall / best, "unit": "permutations/s",
            # `cores` (the contract's key) = the CPUs the box GRANTS this process (cgroup quota / affinity): the figure is bound by
            # them, not by the `threads` the fastest run started (oversubscribing a quota hides its throttling pauses) nor by the
            # cpus_visible; value_per_quota_cpu = value / cores, to be compared with value_1core (one unthrottled core)
            "cores": granted, "threads": threads, "cpu_quota": quota, "cpus_visible": os.cpu_count(), "kind": "port",
            "value_per_quota_cpu": nall / best / granted,
            "reference_cargo_bench": reference_cargo_bench(),
            "sample": "Hash::digest(Merkle4, 4 scalars): %d digests per sample on %d threads, 1 thread: %d digests per sample; "
                      "warm-up + 10 samples each, median reported (min in *_min)" % (nall, threads, n1),
            "value_min_time": nall / float(np.min(allt)), "value_1core": n1 / t1, "value_1core_min_time": n1 / float(np.min(one)),
            "cpu": cpu_model, "compiler": "gcc " + flags,
            "note": "C restatement of the reference CPU path (oracle/p252_oracle.c, reference schedule: 2,000 field multiplications per "
                    "permutation); the reference's own cargo bench is probed at run time: see reference_cargo_bench"}


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


class Env:
    """what every workload needs: the context, the device, the process group"""

    def __init__(self, torch, dist, ctx, dev, coll_dev, rank, world):
        self.torch, self.dist, self.ctx, self.dev, self.coll_dev, self.rank, self.world = torch, dist, ctx, dev, coll_dev, rank, world
        self.side = torch.cuda.Stream(device=dev)  # the clock probe's stream
        self._comm, self.library_comm_error = None, None

    def library_comm(self):
        """this rank's p252_comm over all ranks (created once: ncclCommInitRank inside the library, rank 0's id handed round
        through the torch.distributed process group; the creation itself broadcasts and validates the constant table over
        RCCL) — only with the nccl backend, i.e. one rank per GPU; None with gloo (ranks sharing a GPU in the tests)"""
        if self._comm is None and self.library_comm_error is None:
            if not self.dist.is_initialized() or self.dist.get_backend() != "nccl" or os.environ.get("P252_BENCH_TORCH_GATHER") == "1":
                self.library_comm_error = ""
                return None
            from poseidon252_amd import comm as C
            try:
                self._comm = C.Comm.create_rank(self.ctx, self.rank, self.world, C.torch_exchange(self.coll_dev))
            except Exception as e:  # reported in the JSON line; the torch.distributed (also RCCL) exchange takes over
                self.library_comm_error = repr(e)
                print("bench.py rank %d: library communicator failed: %r" % (self.rank, e), file=sys.stderr)
            # every rank must take the same path: one failure anywhere sends all of them to the torch exchange
            ok = self.torch.tensor([1 if self._comm is not None else 0], dtype=self.torch.int32, device=self.coll_dev)
            self.dist.all_reduce(ok, op=self.dist.ReduceOp.MIN)
            if int(ok.item()) == 0 and self._comm is not None:
                self._comm.destroy()
                self._comm, self.library_comm_error = None, "another rank failed to create its communicator"
        return self._comm

    def barrier(self):
        if self.dist.is_initialized():
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def max_over_ranks(self, seconds):
        if not self.dist.is_initialized():
            return seconds
        t = self.torch.tensor([seconds], dtype=self.torch.float64, device=self.coll_dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def per_rank(self, seconds):
        """every rank's own wall time of the timed region (the same barriers bracket all of them, so the spread is the
        ranks' imbalance: a scaling curve can be read for balance, not only for the slowest rank)"""
        if not self.dist.is_initialized():
            return [seconds]
        t = self.torch.zeros(self.world, dtype=self.torch.float64, device=self.coll_dev)
        t[self.rank] = seconds
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return [float(v) for v in t.tolist()]


class Workload:
    """one BASELINE config as synthetic device-resident input + a step; see make_workload"""
    pass


def make_workload(E, wl, log2n):
    import poseidon252_amd as P
    from poseidon252_amd import synth
    torch, dist, ctx, dev, rank, world = E.torch, E.dist, E.ctx, E.dev, E.rank, E.world
    W = Workload()
    W.key, W.depth = wl, 12
    if wl == "merkle4_digests":
        log2n = log2n or 20
        n = 1 << log2n
        hb = P.HashBatch(P.Domain.Merkle4, 4, ctx=ctx)
        in_scalars, W.perms_per_step = 4 * n, n
        W.name = "2^%d independent Merkle4 digests per GPU (BASELINE configs[1])" % log2n
        W.wake = 24
    elif wl == "tree":
        log2n = log2n or 24
        n = 1 << log2n
        hb = P.HashBatch(P.Domain.Merkle4, 4, ctx=ctx)
        in_scalars, W.perms_per_step = n, P.levels_len(n)
        W.name = "2^%d-leaf arity-4 Merkle tree per GPU, all levels (BASELINE configs[2])" % log2n
        W.wake = 6
    elif wl == "forest":
        log2n = log2n or 24
        n = 1 << log2n
        hb = P.HashBatch(P.Domain.Merkle4, 4, ctx=ctx)
        W.per_tree = 4 ** 6 if n >= 4 ** 8 else 4 ** 2
        W.n_trees = n // W.per_tree
        in_scalars, W.perms_per_step = n, W.n_trees * P.levels_len(W.per_tree)
        BYTES_PER_PERM["forest"] = (W.per_tree * 32 + 32) / float(P.levels_len(W.per_tree))  # leaves in, root out, per tree
        W.name = ("forest of %d independent arity-4 Merkle trees of 4^%d leaves per GPU (2^%d leaves; p252_merkle4_forest_device: one launch per level "
                  "across all trees; the poseidon-merkle shape, AGENTS.md:62-66)" % (W.n_trees, 6 if W.per_tree == 4 ** 6 else 2, log2n))
        W.wake = 6
    elif wl == "extract":
        # the path's one HBM-bound kernel with a workload of its own: 2^log2n openings extracted (no hashing) out of a STORED tree of
        # 2^(log2n + 4) leaves — units are (opening, level) records of 96 sibling bytes; priced against the HBM roofline
        log2n = log2n or 20
        n = 1 << log2n
        hb = P.HashBatch(P.Domain.Merkle4, 4, ctx=ctx)
        W.tree_leaves = n << 4
        W.depth = (log2n + 4 + 1) // 2  # levels above 2^(log2n + 4) leaves
        in_scalars, W.perms_per_step = W.tree_leaves, n * W.depth
        W.unit = "opening levels/s"
        W.bytes_per_unit = 96.0 + 96.0 + 1.0 + (64.0 + 4.0) / W.depth  # siblings read + written, position byte; per opening: leaf in + out, index
        W.name = ("2^%d openings of depth %d extracted on the device from a stored 2^%d-leaf tree (p252_merkle4_openings_device: data movement "
                  "only, SURVEY §8 f3) — HBM roofline" % (log2n, W.depth, log2n + 4))
        W.wake = 6
    elif wl == "encrypt":
        log2n = log2n or 20
        n = 1 << log2n
        hb = P.HashBatch(P.Domain.Merkle4, 4, ctx=ctx)  # (only for the context; the tag below is the encryption tag)
        in_scalars, W.perms_per_step = 5 * n, 2 * n
        W.name = "2^%d encryptions of 2-scalar messages per GPU (2 permutations each, SURVEY §8 f4; recipe unpinned)" % log2n
        W.wake = 12
    elif wl == "openings":
        log2n = log2n or 20
        n = 1 << log2n
        hb = P.HashBatch(P.Domain.Merkle4, 4, ctx=ctx)
        in_scalars, W.perms_per_step = n * (1 + 3 * W.depth), W.depth * n
        W.name = "2^%d Merkle4 openings of depth %d per GPU (branch re-hash, SURVEY §8 f3)" % (log2n, W.depth)
        W.wake = 4
    else:
        log2n = log2n or 20
        n = 1 << log2n
        hb = P.HashBatch(P.Domain.Other, 42, output_len=5, ctx=ctx)
        in_scalars, W.perms_per_step = 42 * n, 12 * n
        W.name = "Domain::Other sponge, 2^%d messages x 42 scalars -> 5 outputs per GPU (BASELINE configs[3])" % log2n
        W.wake = 4
    W.n, W.log2n = n, log2n
    tag = hb.tag
    if wl == "encrypt":
        from poseidon252_amd import encryption as Enc
        tag = Enc.encryption_tag(2)
    W.tag = tag
    # synthetic input generated ON the device by SURVEY §8(d)'s generator (splitmix64 stream, rejection-sampled below p:
    # uniform field elements; poseidon252_amd/synth.py, byte-identical to the oracle's fill_random).  Rank 0's
    # configs[1] batch (seed 0xc10d) is exactly the one tests/test_gpu_fullsize.py verifies digest by digest.
    g = torch.Generator(device=dev)
    g.manual_seed(0xC10D + rank)
    d_in = synth.splitmix_scalars(0xC10D + rank, in_scalars, dev)
    W.d_in = d_in
    depth = W.depth
    if wl == "merkle4_digests":
        d_out = torch.empty((n, 4), dtype=torch.int64, device=dev)
        W.step = lambda: ctx.hash_batch_device(tag, d_in, 4, 1, d_out, n)
    elif wl == "tree":
        d_out = torch.empty(4, dtype=torch.int64, device=dev)
        if not dist.is_initialized():
            W.step = lambda: ctx.merkle4_tree_device(tag, d_in, n, d_out, None)
        else:  # (also at one rank when a process group exists — P252_BENCH_FORCE_DIST: the RCCL all-gather on a 1-GPU box)
            # BASELINE configs[4] structure: every rank reduces its complete subtree, the W roots (32 B each)
            # are all-gathered (the path's only exchange step) and the top levels are hashed on every rank
            coll_dev = E.coll_dev
            d_top = torch.empty(4, dtype=torch.int64, device=dev)
            W.d_top = d_top
            comm = E.library_comm()
            if comm is not None:
                # RCCL INSIDE the library (p252_merkle4_tree_sharded_device): subtree -> ncclAllGather of the roots on the
                # launch stream -> top levels, one call, no host round trip and no Python between the launches
                W.exchange_impl = "ncclAllGather inside libposeidon252_hip.so (p252_merkle4_tree_sharded_device), on the launch stream"
                W.step = lambda: comm.merkle4_tree_sharded_device(tag, d_in, n, d_top)
            else:
                # torch.distributed collective between two library calls: the gloo test configuration (ranks sharing one
                # GPU — RCCL refuses two ranks on a device), or a library communicator that could not be created (reported)
                W.exchange_impl = "torch.distributed.all_gather_into_tensor (%s)%s" % (
                    dist.get_backend(), "; library communicator unavailable: " + E.library_comm_error if E.library_comm_error else "")
                d_roots = torch.empty(world * 4, dtype=torch.int64, device=coll_dev)  # flat: gloo and nccl both accept it

                def step():
                    ctx.merkle4_tree_device(tag, d_in, n, d_out, None)
                    dist.all_gather_into_tensor(d_roots, d_out if coll_dev == dev else d_out.cpu())
                    roots_dev = d_roots if coll_dev == dev else d_roots.to(dev)
                    ctx.merkle4_tree_device(tag, roots_dev.contiguous(), world, d_top, None)
                W.step = step
            # whole-job units of one step: every rank's subtree + the top levels ONCE (every rank hashes the same
            # levels_len(world) top nodes redundantly; they are not counted `world` times)
            W.job_units_per_step = world * W.perms_per_step + P.levels_len(world)
            W.perms_per_step += P.levels_len(world)
            W.name += " + all-gather of %d subtree roots and top levels" % world
            if world == 8:
                W.name += (" = 2^27-leaf tree sharded across 8 GPUs (BASELINE configs[4])" if log2n == 24 else
                           " = the BASELINE configs[4] composition (8 subtrees, roots gathered, top [n0, n1, 0, 0]) scaled down to 8 x 2^%d leaves" % log2n)
            W.root_hex = lambda: "".join("%016x" % (int(v) & 0xFFFFFFFFFFFFFFFF) for v in reversed(W.d_top.cpu().tolist()))
        W.step()  # allocate the context-owned level scratch outside the timed region
    elif wl == "forest":
        d_out = torch.empty((W.n_trees, 4), dtype=torch.int64, device=dev)
        W.step = lambda: ctx.merkle4_forest_device(tag, d_in, W.n_trees, W.per_tree, d_out)
        W.step()  # (context-owned level scratch, outside the timed region)
    elif wl == "extract":
        W.d_root, W.d_levels = P.merkle4_tree(d_in, tag=tag, ctx=ctx, want_levels=True)  # the stored tree (built once, outside any timed region)
        d_idx = torch.randint(0, W.tree_leaves, (n,), dtype=torch.int32, device=dev, generator=g)  # random paths: the gather's worst case
        d_out = torch.empty((n, 4), dtype=torch.int64, device=dev)
        W.d_sib = torch.empty((n, W.depth, 3, 4), dtype=torch.int64, device=dev)
        W.d_pos = torch.empty((n, W.depth), dtype=torch.uint8, device=dev)
        W.d_bad = torch.zeros(1, dtype=torch.int32, device=dev)
        W.step = lambda: ctx.merkle4_openings_device(d_in, W.tree_leaves, W.d_levels, d_idx, n, out=(d_out, W.d_sib, W.d_pos, W.d_bad))
    elif wl == "encrypt":
        d_out = torch.empty((n, 3, 4), dtype=torch.int64, device=dev)
        d_msgs, d_secrets, d_nonces = d_in[:2 * n], d_in[2 * n:4 * n], d_in[4 * n:]
        W.step = lambda: Enc.encrypt_batch_device(d_msgs, d_secrets, d_nonces, 2, d_out, n, ctx=ctx, tag=tag)
    elif wl == "openings":
        d_out = torch.empty((n, 4), dtype=torch.int64, device=dev)
        d_leaves, d_sibs = d_in[:n], d_in[n:]
        d_pos = torch.randint(0, 4, (n, depth), dtype=torch.uint8, device=dev, generator=g)
        W.step = lambda: ctx.merkle4_path_batch_device(tag, d_leaves, d_sibs, d_pos, depth, d_out, n)
    else:
        d_out = torch.empty((n, 5, 4), dtype=torch.int64, device=dev)
        W.step = lambda: ctx.hash_batch_device(tag, d_in, 42, 5, d_out, n)
    W.d_out = d_out

    def self_check():
        """correctness of what was just timed, WITHOUT the oracle (every rank, every N): a slice of the batch is
        hashed again on its own and must reproduce the same digests (shard consistency + determinism)"""
        if wl == "merkle4_digests":
            lo, cnt = n // 3, min(n - n // 3, 4096)
            again = torch.empty((cnt, 4), dtype=torch.int64, device=dev)
            ctx.hash_batch_device(tag, d_in[lo * 4:(lo + cnt) * 4], 4, 1, again, cnt)
            torch.cuda.synchronize()
            return bool(torch.equal(again, d_out[lo:lo + cnt]))
        if wl == "tree":
            # subtree composition: the tree over the roots of the 4 quarter trees (4^k leaves: complete subtrees; 2 half
            # trees when the leaf count is 2 x 4^k) is the tree's root
            parts = 4 if log2n % 2 == 0 else 2
            quarters = torch.stack([P.merkle4_tree(d_in[i * (n // parts):(i + 1) * (n // parts)], tag=tag, ctx=ctx) for i in range(parts)])
            top = P.merkle4_tree(quarters.contiguous(), tag=tag, ctx=ctx)
            ref = torch.empty(4, dtype=torch.int64, device=dev)
            ctx.merkle4_tree_device(tag, d_in, n, ref, None)
            torch.cuda.synchronize()
            if not dist.is_initialized():
                return bool(torch.equal(top, ref) and torch.equal(ref, d_out))
            # sharded build: the timed step's root (d_top: through the library's ncclAllGather, or the torch exchange) must be the
            # tree over ALL ranks' subtree roots — re-gathered here through torch.distributed, an independent exchange
            mine = ref if E.coll_dev == dev else ref.cpu()
            gathered = torch.empty(world * 4, dtype=torch.int64, device=E.coll_dev)
            dist.all_gather_into_tensor(gathered, mine)
            again = P.merkle4_tree(gathered.to(dev).view(world, 4).contiguous(), tag=tag, ctx=ctx)
            torch.cuda.synchronize()
            return bool(torch.equal(top, ref) and torch.equal(again, W.d_top))
        if wl == "extract":  # every extracted opening re-hashes to the stored tree's root; no position was out of range
            roots = torch.empty((n, 4), dtype=torch.int64, device=dev)
            ctx.merkle4_path_batch_device(tag, d_out, W.d_sib, W.d_pos, W.depth, roots, n)
            torch.cuda.synchronize()
            return bool((roots == W.d_root.view(1, 4)).all()) and int(W.d_bad.item()) == 0
        if wl == "forest":  # a few trees built on their own by the single-tree entry point
            pick = sorted(set([0, W.n_trees // 3, W.n_trees - 1]))
            alone = torch.stack([P.merkle4_tree(d_in[t * W.per_tree:(t + 1) * W.per_tree], tag=tag, ctx=ctx) for t in pick])
            torch.cuda.synchronize()
            return bool(torch.equal(alone, d_out[pick]))
        if wl == "encrypt":
            # decrypting what was just produced gives the messages back, with every authentication flag set
            back = torch.empty((n, 2, 4), dtype=torch.int64, device=dev)
            ok = torch.zeros(n, dtype=torch.uint8, device=dev)
            Enc.decrypt_batch_device(d_out, d_secrets, d_nonces, 2, back, ok, n, ctx=ctx, tag=tag)
            torch.cuda.synchronize()
            return bool(torch.equal(back.view(-1, 4), d_msgs) and bool(ok.all()))
        if wl == "openings":
            lo, cnt = n // 3, min(n - n // 3, 4096)
            again = torch.empty((cnt, 4), dtype=torch.int64, device=dev)
            ctx.merkle4_path_batch_device(tag, d_leaves[lo:lo + cnt], d_sibs[lo * 3 * depth:(lo + cnt) * 3 * depth], d_pos[lo:lo + cnt].contiguous(),
                                          depth, again, cnt)
            torch.cuda.synchronize()
            return bool(torch.equal(again, d_out[lo:lo + cnt]))
        lo, cnt = n // 3, min(n - n // 3, 1024)
        again = torch.empty((cnt, 5, 4), dtype=torch.int64, device=dev)
        ctx.hash_batch_device(tag, d_in[lo * 42:(lo + cnt) * 42], 42, 5, again, cnt)
        torch.cuda.synchronize()
        return bool(torch.equal(again, d_out[lo:lo + cnt]))
    W.self_check = self_check

    def oracle_sample():
        """(kind, tag, inputs, in_len, out_len, gpu_output) of a strided sample of what was just timed — handed to the
        cpu_baseline leg, the only place the oracle runs"""
        if wl == "tree":  # a 4^6-leaf subtree at the front of the same leaves, built by the same entry point
            sub = min(n, 1 << 12)
            got = P.merkle4_tree(d_in[:sub].contiguous(), tag=tag, ctx=ctx).cpu().numpy().view(np.uint64)
            return ("tree", tag, d_in[:sub].cpu().numpy().view(np.uint64), None, None, got)
        if wl == "extract":  # a sample of the extracted openings: the oracle re-hashes them and must arrive at the GPU tree's root
            idx = torch.arange(0, n, max(1, n // 128), device=dev)
            return ("paths", tag, (d_out[idx].cpu().numpy().view(np.uint64), W.d_sib[idx].cpu().numpy().view(np.uint64), W.d_pos[idx].cpu().numpy()),
                    None, None, W.d_root.cpu().numpy().view(np.uint64).reshape(1, 4).repeat(idx.numel(), axis=0))
        if wl == "forest":  # four trees of what was just timed, leaves and roots
            pick = sorted(set([0, W.n_trees // 2, W.n_trees - 2, W.n_trees - 1]))
            leaves = torch.stack([d_in[t * W.per_tree:(t + 1) * W.per_tree] for t in pick]).cpu().numpy().view(np.uint64)
            return ("forest", tag, leaves, None, None, d_out[pick].cpu().numpy().view(np.uint64))
        if wl == "merkle4_digests":
            idx = torch.arange(0, n, max(1, n // 512), device=dev)
            return ("hash", tag, d_in.view(n, 4, 4)[idx].cpu().numpy().view(np.uint64), 4, 1, d_out[idx].cpu().numpy().view(np.uint64).reshape(-1, 1, 4))
        idx = torch.arange(0, n, max(1, n // 128), device=dev)
        if wl == "encrypt":
            return ("encrypt", tag, tuple(t.cpu().numpy().view(np.uint64) for t in (d_msgs.view(n, 2, 4)[idx], d_secrets.view(n, 2, 4)[idx], d_nonces[idx])),
                    None, None, d_out[idx].cpu().numpy().view(np.uint64))
        if wl == "openings":
            return ("paths", tag, (d_leaves[idx].cpu().numpy().view(np.uint64), d_sibs.view(n, depth, 3, 4)[idx].cpu().numpy().view(np.uint64),
                                   d_pos[idx].cpu().numpy()), None, None, d_out[idx].cpu().numpy().view(np.uint64))
        return ("hash", tag, d_in.view(n, 42, 4)[idx].cpu().numpy().view(np.uint64), 42, 5, d_out[idx].cpu().numpy().view(np.uint64))
    W.oracle_sample = oracle_sample
    return W


def probe_under_load(E, W, step_ms, n_steps=4):
    """the shader clock while `n_steps` untimed steps of the workload run: the one-wave probe (p252_clock_probe_device) is
    launched on a second stream after the first of them and sleeps for about 1.5 steps"""
    torch = E.torch
    spin_us = int(min(60000.0, max(400.0, 1500.0 * step_ms)))
    W.step()
    t = E.ctx.clock_probe(spin_us=spin_us, stream=E.side)
    for _ in range(n_steps - 1):
        W.step()
    torch.cuda.synchronize()
    r = E.ctx.clock_probe_result(t)
    return r if 0.3 < r["shader_ghz"] < 4.0 else None


def run_timed(E, W, steps, warmup):
    """wake-up launches, `warmup` untimed steps, exactly `steps` timed steps between barriers; HIP events per step on the
    launch stream; the clock probe beside extra untimed steps immediately before and after the timed region"""
    torch = E.torch
    # Device wake-up (setup, reported in the JSON line): an idle MI355X needs ~25 ms of activity before its clocks
    # reach the steady state a running service sees (profiles/r01_bench_kernel_trace_v9.txt: the first ten launches are up
    # to 25 % slower).  A fixed count per workload (not a time): every rank must issue the same collectives at N > 1.
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for i in range(W.wake):
        if i == W.wake - 1:
            e0.record()
        W.step()
    e1.record()
    torch.cuda.synchronize()
    step_ms_est = e0.elapsed_time(e1)
    for _ in range(warmup):
        W.step()
    clk_before = probe_under_load(E, W, step_ms_est)
    E.barrier()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    t0 = time.perf_counter()
    evs[0].record()
    for i in range(steps):
        W.step()  # launched on torch's current stream; the events below are recorded on that same stream
        evs[i + 1].record()
    E.torch.cuda.synchronize()
    own = time.perf_counter() - t0  # this rank's own work done (before the closing barrier: the balance figure)
    E.barrier()
    elapsed = time.perf_counter() - t0
    launch_ms = [evs[i].elapsed_time(evs[i + 1]) for i in range(steps)]
    clk_after = probe_under_load(E, W, step_ms_est)
    elapsed = E.max_over_ranks(elapsed)
    W.rank_ms_per_step = [v / steps * 1e3 for v in E.per_rank(own)]
    return elapsed, launch_ms, clk_before, clk_after


def sysfs_sclk_mhz(torch, local_rank):
    """the driver's own reading of this GPU's shader clock (hwmon freq1_input), where the box exposes it — a cross-check
    of the probe, sampled outside the timed region"""
    try:
        p = torch.cuda.get_device_properties(local_rank)
        bdf = "%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
        import glob
        paths = glob.glob("/sys/bus/pci/devices/%s/hwmon/hwmon*/freq1_input" % bdf)
        return int(open(paths[0]).read()) / 1e6 if paths else None
    except Exception:
        return None


def roofline_hbm_of(W, launch_ms, clk_before, clk_after):
    """the contract's HBM shape for a workload whose kernel moves bytes and computes nothing (extract): achieved = ALGORITHMIC bytes
    per launch / mean launch time (HIP events on the launch stream) against the 8 TB/s HBM3E peak; traffic = HBM bytes per launch
    from the committed FETCH_SIZE x 2 + WRITE_SIZE counter passes of that kernel (a random gather fetches whole 128-byte lines for
    the 96 bytes it uses: ratio ~1.17 by construction)"""
    k_ms = float(np.mean(launch_ms))
    alg = W.bytes_per_unit * W.perms_per_step
    gbps = alg / (k_ms * 1e-3) / 1e9
    kern = KERNEL_OF[W.key]
    d = pmc_profile(kern)
    traffic = None
    if d and "hbm_bytes_per_launch" in d and not d["stale"]:
        traffic = d["hbm_bytes_per_launch"] * W.perms_per_step / d["units_per_launch"]
    clocks = [c["shader_ghz"] for c in (clk_before, clk_after) if c]
    return {"bound": "hbm", "kernel": kern, "achieved": gbps, "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": gbps / PEAK_HBM_GBPS,
            "traffic": traffic, "traffic_algorithmic_bytes": alg, "traffic_ratio": (traffic / alg) if traffic else None,
            "traffic_source": d["source"] if d else None,
            "algorithmic_bytes_per_unit": W.bytes_per_unit, "units_per_launch": W.perms_per_step,
            "launch_ms_mean": k_ms, "launch_ms_min": float(np.min(launch_ms)),
            "clock_ghz_measured": float(np.mean(clocks)) if clocks else None,
            "note": "HBM-bound data movement (no arithmetic): algorithmic bytes = per (opening, level) 96 sibling bytes read + 96 written + "
                    "1 position byte, per opening the leaf in and out and its index; random positions (a gather of one 128-byte line per record)"}


def roofline_of(W, launch_ms, clk_before, clk_after, sclk_sysfs=None):
    if W.key == "extract":
        return roofline_hbm_of(W, launch_ms, clk_before, clk_after)
    wl, n = W.key, W.n
    k_ms = float(np.mean(launch_ms))
    per_gpu_rate = W.perms_per_step / (k_ms * 1e-3)
    achieved_mac = per_gpu_rate * MACS_PER_PERM_REFERENCE
    hbm_gbps = per_gpu_rate * BYTES_PER_PERM[wl] / 1e9
    kern = KERNEL_OF[wl]
    lanes_per_perm, isa_key, pmc_key = 1, None, None
    if wl in ("sponge42", "openings") and os.environ.get("P252_LINE_FETCH", "1")[:1] != "0":
        # 42-scalar messages / depth-12 paths in line-aligned arrays: the whole-line-fetch builds (kernels.hip); the counter
        # passes and ISA counts are filed under the base name
        isa_key, pmc_key, kern = None, kern, kern + "_lines"
    coop_max = int(os.environ.get("P252_COOP_MAX_NODES", "16384"))
    items = n  # independent states per launch (digests, messages, openings)
    if wl == "merkle4_digests" and 8192 < items <= min(coop_max, 16384):
        kern, lanes_per_perm, isa_key, pmc_key = "k_merkle4_coop<4>", 4, "k_merkle4_coop<4>", "k_merkle4_coop4"
    elif wl in ("merkle4_digests", "sponge42", "openings", "encrypt") and items <= min(coop_max, 8192):
        # batches this small run the lane-group kernels: eight lanes per state (csrc/coop29.hpp); the permutation body is
        # the one of k_merkle4_coop<8>, whose ISA counts and counter passes stand for all of them
        kern = {"merkle4_digests": "k_merkle4_coop<8>", "sponge42": "k_sponge_coop", "openings": "k_merkle4_path_coop", "encrypt": "k_crypt_coop"}[wl]
        lanes_per_perm, isa_key, pmc_key = 8, "k_merkle4_coop<8>", "k_merkle4_coop8"
    clocks = [c["shader_ghz"] for c in (clk_before, clk_after) if c]
    ghz = float(np.mean(clocks)) if clocks else None
    clock = {"ghz_measured": ghz, "before": clk_before, "after": clk_after, "nominal_ghz": NOMINAL_CLOCK_HZ / 1e9,
             "sysfs_sclk_mhz": sclk_sysfs,
             "method": "one-wave probe kernel (s_memtime / s_memrealtime at 100 MHz) on a second stream beside untimed steps of this "
                       "workload, immediately before and after the timed region"}
    peak_meas = SIMDS * ghz * 1e9 / 4.0 * 64 if ghz else None
    isa = isa_counts(isa_key or kern)
    executed = issue = None
    if isa:
        mac_rate = per_gpu_rate * lanes_per_perm * isa["v_mad_i64_i32"]
        executed = {"macs_per_perm": isa["v_mad_i64_i32"] * lanes_per_perm, "achieved": mac_rate / 1e12, "peak": PEAK_INT32_MAC_PER_S / 1e12,
                    "frac": mac_rate / PEAK_INT32_MAC_PER_S, "unit": "TMAC/s", "source": isa["source"],
                    "peak_at_measured_clock": peak_meas / 1e12 if peak_meas else None,
                    "frac_at_measured_clock": mac_rate / peak_meas if peak_meas else None,
                    "frac_of_measured_mad_stream": mac_rate / (MEASURED_MAD_STREAM_WAVE_INST_PER_S * 64),
                    "note": "the fraction of the hardware: multiply-adds actually issued (counted in the ISA) / (1024 SIMDs x clock / 4 cycles x 64 lanes), "
                            "at the nominal 2.4 GHz (frac) and at the clock measured in this run (frac_at_measured_clock)"}
        # every VALU instruction priced at its issue cost (4 cycles for multiply-adds, 64-bit adds / shifts and VOP3
        # 3-operand forms, 2 for plain 32-bit ops — profiles/r02_valu_rates_gfx950.txt): share of all SIMD cycles
        cyc = per_gpu_rate * lanes_per_perm / 64.0 * isa["valu_issue_cycles"]
        issue = {"valu_insts_per_perm": isa["valu_total"], "lanes_per_perm": lanes_per_perm, "issue_cycles_per_perm": isa["valu_issue_cycles"],
                 "frac": cyc / (SIMDS * NOMINAL_CLOCK_HZ), "frac_at_measured_clock": cyc / (SIMDS * ghz * 1e9) if ghz else None,
                 "pmc": pmc_valu(pmc_key or kern),
                 "note": "SIMD cycles spent issuing VALU work under the 4-/2-cycle model / all SIMD cycles (at 2.4 GHz; at the measured clock)"}
    traffic = pmc_traffic(pmc_key or kern, wl, W.perms_per_step)
    exe_frac = executed["frac"] if executed else None
    return {
        # ---- top-level SCALARS (the driver's record keeps only these): the HARDWARE fraction first ----
        # achieved = multiply-adds the kernel actually issues (ISA-derived count x lanes x permutations per launch / mean launch
        # time, HIP events on the launch stream); peak = 1024 SIMDs x 2.4 GHz / 4 cycles x 64 lanes; frac = achieved / peak <= 1
        "bound": "valu-int32-mac", "kernel": kern,
        "achieved": executed["achieved"] if executed else None, "peak": PEAK_INT32_MAC_PER_S / 1e12, "unit": "TMAC/s",
        "frac": exe_frac,
        "frac_at_measured_clock": executed["frac_at_measured_clock"] if executed else None,
        "frac_of_measured_mad_stream": executed["frac_of_measured_mad_stream"] if executed else None,
        "frac_valu_issue": issue["frac"] if issue else None,
        "frac_valu_issue_at_measured_clock": issue["frac_at_measured_clock"] if issue else None,
        "macs_per_perm_executed": executed["macs_per_perm"] if executed else None,
        "valu_insts_per_perm": issue["valu_insts_per_perm"] * lanes_per_perm if issue else None,
        "clock_ghz_measured": ghz,
        "launch_ms_mean": k_ms, "launch_ms_min": float(np.min(launch_ms)), "units_per_launch": W.perms_per_step,
        # HBM bytes per step from the PMC passes (FETCH_SIZE x 2 + WRITE_SIZE, MI355X_MICROARCH.md) next to the algorithmic bytes
        "traffic": traffic.get("bytes") if traffic else None,
        "traffic_algorithmic_bytes": BYTES_PER_PERM[wl] * W.perms_per_step,
        "traffic_ratio": traffic.get("ratio") if traffic else None,
        "hbm_achieved_gbps": hbm_gbps, "hbm_frac": hbm_gbps / PEAK_HBM_GBPS,
        # SURVEY §8d's pricing: the REFERENCE schedule's 256,000 MACs per permutation against the same peak.  The kernel runs an
        # algebraically equivalent schedule with 4x fewer multiply-adds (bit-exact), so this is an algorithmic speed-up times a
        # hardware fraction — it exceeds 1 and says nothing about the hardware; `frac` above does.
        "macs_per_perm_reference_schedule": MACS_PER_PERM_REFERENCE,
        "achieved_reference_schedule": achieved_mac / 1e12,
        "frac_reference_schedule": achieved_mac / PEAK_INT32_MAC_PER_S,
        "note": "frac = multiply-adds actually issued (counted in the ISA, equal to SQ_INSTS_VALU-derived counts) / VALU issue peak at the nominal "
                "2.4 GHz; *_at_measured_clock = the same at the shader clock measured in this run; frac_valu_issue = SIMD cycles issuing any VALU "
                "instruction (4-/2-cycle model) / all SIMD cycles; *_reference_schedule = SURVEY §8d's 256,000-MAC figure (not a hardware fraction)",
        # ---- nested detail (sources, probes, per-model notes) ----
        "executed": executed, "valu_issue": issue, "clock": clock,
        "hbm": {"achieved": hbm_gbps, "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": hbm_gbps / PEAK_HBM_GBPS,
                "algorithmic_bytes_per_perm": BYTES_PER_PERM[wl]},
        "traffic_detail": traffic,
    }


# ---- the printed line -------------------------------------------------------------------------------------
# The driver keeps an 8 KB tail of stdout; round 4's line had grown to 26 KB and could no longer be parsed.  bench.py therefore
# builds the full record as before (every source, probe, nested model and note: `detail`), writes it to a side file next to
# bench.py (--detail-file) and PRINTS a line of scalars only, whose length is bounded and asserted.
LINE_BUDGET_BYTES = 6000
# stdout carries exactly ONE line.  Libraries loaded into the ranks print to fd 1 on their own (RCCL's version banner at communicator
# creation, HIP warnings): main() points fd 1 at stderr for the whole run and keeps the real stdout for the line (emit()).
_REAL_STDOUT_FD = None


def keep_stdout_for_the_line():
    global _REAL_STDOUT_FD
    if _REAL_STDOUT_FD is None:
        sys.stdout.flush()
        _REAL_STDOUT_FD = os.dup(1)
        os.dup2(2, 1)

ROOFLINE_SCALARS = ("bound", "kernel", "achieved", "peak", "unit", "frac", "frac_at_measured_clock", "frac_valu_issue", "macs_per_perm_executed",
                    "clock_ghz_measured", "launch_ms_mean", "units_per_launch", "traffic", "traffic_ratio", "frac_reference_schedule")
SECONDARY_SCALARS = ("value", "unit", "units_per_gpu_per_step", "ms_per_step", "steps", "ms_per_step_rank_min", "ms_per_step_rank_max",
                     "self_consistency_ok", "parity_sample_ok")


def _sig(v, digits=7):
    """floats at `digits` significant digits (a 17-digit repr is 2.5 x the bytes and carries no information here)"""
    if isinstance(v, bool) or not isinstance(v, float):
        return v
    if v != v or v in (float("inf"), float("-inf")):
        return None
    return float("%.*g" % (digits, v))


def _short(text, limit):
    return text if text is None or len(text) <= limit else text[:limit - 3] + "..."


def compact_roofline(r):
    return {k: _sig(r.get(k)) for k in ROOFLINE_SCALARS}


def compact_line(detail, detail_file=None):
    """the ONE line printed on stdout: the contract's keys, `roofline` / `cpu_baseline` / each secondary as scalars only.  Everything
    else of `detail` (executed / valu_issue / clock / hbm / traffic_detail, per-rank arrays, sources, notes) lives in the side file."""
    c = detail["config"]
    line = {k: _sig(detail.get(k), 9) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "ms_per_step_rank_min",
                                                 "ms_per_step_rank_max", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    line["config"] = {"workload": _short(c["workload"], 160), "units_per_gpu_per_step": c["units_per_gpu_per_step"],
                      "units_whole_job_per_step": c.get("units_whole_job_per_step"), "ranks": c.get("ranks"),
                      "collective_backend": c.get("collective_backend"), "exchange_impl": _short(c.get("exchange_impl"), 100),
                      "constants_identical_on_all_ranks": c.get("constants_identical_on_all_ranks")}
    line["roofline"] = compact_roofline(detail["roofline"])
    rh = detail.get("roofline_hbm")
    if rh and detail["roofline"].get("bound") != "hbm":  # the same kernel in the contract's HBM shape (not the binding bound)
        line["roofline_hbm"] = {k: _sig(rh.get(k)) for k in ("bound", "achieved", "peak", "unit", "frac")}
    line["self_consistency_ok"] = detail.get("self_consistency_ok")
    if "secondary" in detail:
        sec = {}
        for key, w in detail["secondary"].items():
            s = {k: _sig(w.get(k), 9 if k in ("value", "ms_per_step") else 7) for k in SECONDARY_SCALARS}
            r = w.get("roofline") or {}
            s.update({"bound": r.get("bound"), "kernel": r.get("kernel"), "frac": _sig(r.get("frac")), "frac_at_measured_clock": _sig(r.get("frac_at_measured_clock")),
                      "achieved": _sig(r.get("achieved")), "traffic_ratio": _sig(r.get("traffic_ratio"))})
            if w.get("exchange_impl"):
                s["exchange_impl"] = _short(w["exchange_impl"], 60)
                s["units_whole_job_per_step"] = w.get("units_whole_job_per_step")
            sec[key] = s
        line["secondary"] = sec
    if "secondary_timeout" in detail:
        line["secondary_timeout"] = {k: detail["secondary_timeout"][k] for k in ("workload", "seconds")}
    cb = detail.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = {"value": _sig(cb["value"]), "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"],
                                "sample": _short(cb["sample"], 120), "threads": cb.get("threads"), "value_1core": _sig(cb.get("value_1core")),
                                "cpu": cb.get("cpu"), "parity_sample_ok": cb.get("parity_sample_ok"),
                                "reference_cargo_bench": {"available": bool((cb.get("reference_cargo_bench") or {}).get("available")),
                                                          "value": _sig((cb.get("reference_cargo_bench") or {}).get("value"))}}
    line["detail_file"] = detail_file
    return line


def emit(detail, detail_path):
    """write the full record to the side file, print the bounded line; a line over budget is a bug (tests assert the budget on
    full synthetic and real lines) — should it ever happen the secondaries are cut to their headline pair rather than the line lost"""
    written = None
    for path in (detail_path, os.path.join("/tmp", "bench_detail.json")):
        try:
            with open(path, "w") as f:
                json.dump(detail, f, indent=1)
            written = path
            break
        except OSError:
            continue
    line = compact_line(detail, os.path.relpath(written, ROOT) if written and written.startswith(ROOT + os.sep) else written)
    text = json.dumps(line, separators=(",", ":"))
    if len(text) > LINE_BUDGET_BYTES and "secondary" in line:
        line["secondary"] = {k: {"value": v["value"], "ms_per_step": v["ms_per_step"], "frac": v["frac"]} for k, v in line["secondary"].items()}
        line["line_trimmed"] = True
        text = json.dumps(line, separators=(",", ":"))
    if _REAL_STDOUT_FD is None:
        print(text, flush=True)
    else:
        sys.stdout.flush()
        os.write(_REAL_STDOUT_FD, (text + "\n").encode())
    return line


def primary_record_to_stderr(detail):
    """the scaling curve is made of the primary alone: as soon as it is measured a minimal record goes to STDERR, so that a
    driver-side kill during a secondary workload (not a hang the watchdog sees) still leaves the number in the captured tail"""
    r = detail["roofline"]
    rec = {"bench_primary": True, "metric": detail["metric"], "value": _sig(detail["value"], 9), "unit": detail["unit"], "n_gpus": detail["n_gpus"],
           "steps": detail["steps"], "ms_per_step": _sig(detail["ms_per_step"], 9), "roofline_frac": _sig(r.get("frac")), "kernel": r.get("kernel")}
    print("bench.py primary: " + json.dumps(rec, separators=(",", ":")), file=sys.stderr, flush=True)


def main():
    args = parse()
    if args.gpus < 1:
        print("bench.py: --gpus must be >= 1", file=sys.stderr)
        sys.exit(2)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        self_launch(args.gpus)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC (RCCL across processes on this driver); before the runtime loads
    if os.environ.get("P252_BENCH_SHARE_GPU") == "1":
        # test-only configuration (N ranks on ONE GPU): one hardware queue per rank.  With the runtime's default of 4, eight ranks plus the
        # test runner's own context exceed the chip's hardware queue slots and the driver time-slices oversubscribed runlists at a
        # granularity of seconds — measured on one box (profiles/r06_shared_gpu_queues.txt): the 8-rank rehearsal 3.8 s with 1 queue per
        # rank, 95-143 s with 2-4.  One rank per GPU (the driver's run) is not affected and not touched.
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "1")
    keep_stdout_for_the_line()
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
    E = Env(torch, dist, ctx, dev, coll_dev, rank, world)

    primary_key = args.workload or "merkle4_digests"
    # BASELINE configs[2] / [3] (configs[4] at 8 ranks), then the SURVEY §8(f) rows that have kernels of their own
    secondary_keys = [] if (args.workload or args.no_secondary) else ["tree", "forest", "sponge42", "openings", "encrypt", "extract"]
    sclk0 = sysfs_sclk_mhz(torch, local_rank)

    def measure(key, log2n, steps, warmup):
        W = make_workload(E, key, log2n)
        elapsed, launch_ms, cb, ca = run_timed(E, W, steps, warmup)
        ok = None
        if not args.no_check:
            ok = W.self_check()
            if not ok:
                print("SELF-CONSISTENCY FAILURE on rank %d (%s)" % (rank, key), file=sys.stderr)
                sys.exit(3)
        sample = W.oracle_sample() if (rank == 0 and world == 1 and not args.no_cpu_baseline and not args.no_check) else None
        return W, elapsed, launch_ms, cb, ca, ok, sample

    if rank == 0:
        print("bench.py: set up (imports, process group, constants) %.1f s since start" % (time.time() - T_START), file=sys.stderr, flush=True)
    W, elapsed, launch_ms, cb, ca, self_ok, sample = measure(primary_key, args.log2n, args.steps, args.warmup)
    if rank == 0:
        print("bench.py: primary measured, %.1f s since start" % (time.time() - T_START), file=sys.stderr, flush=True)
    samples = [(primary_key,) + sample] if sample else []
    line = None
    if rank == 0:
        total_perms = getattr(W, "job_units_per_step", W.perms_per_step * world) * args.steps
        roofline = roofline_of(W, launch_ms, cb, ca, {"idle_before_run": sclk0, "after_timed_region": sysfs_sclk_mhz(torch, local_rank)})
        line = {
            "metric": ("Poseidon width-5 permutations/s (= Merkle4 digests/s), bit-exact" if W.key != "extract" else
                       "Merkle opening levels extracted/s (data movement, no hashing; --workload extract)"),
            "value": total_perms / elapsed, "unit": getattr(W, "unit", "permutations/s"), "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            # every rank's own time per step up to its last launch completing (before the closing barrier): min / max = balance
            "ms_per_step_rank_min": min(W.rank_ms_per_step), "ms_per_step_rank_max": max(W.rank_ms_per_step), "ms_per_step_per_rank": W.rank_ms_per_step,
            "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "int64", "data": "synthetic",
            "dtype_note": "255-bit field elements as 9 x 29-bit limbs (int32), products accumulated in signed 64-bit columns (v_mad_i64_i32)",
            "config": {"workload": W.name, "units_per_gpu_per_step": W.perms_per_step,
                       "units_whole_job_per_step": getattr(W, "job_units_per_step", W.perms_per_step * world),
                       "exchange_impl": getattr(W, "exchange_impl", None), "root_mont_hex": W.root_hex() if hasattr(W, "root_hex") else None,
                       "sharding": "independent batches per GPU, no data-path collective",
                       "ranks": dist.get_world_size() if dist.is_initialized() else 1,
                       "collective_backend": (dist.get_backend() if dist.is_initialized() else None),
                       "input": "splitmix64 seed 0xc10d + rank, uniform mod p (SURVEY §8d)",
                       "constants": "RCCL broadcast from rank 0 (identical to local derivation: %s)" % tables_identical,
                       "constants_identical_on_all_ranks": bool(tables_identical)},
            "roofline": roofline,
            # the same kernel priced against the HBM roofline in the contract's shape (NOT the binding bound here)
            "roofline_hbm": ({"bound": "hbm", "achieved": roofline["hbm"]["achieved"], "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                              "frac": roofline["hbm"]["frac"], "traffic": roofline["traffic"], "traffic_ratio": roofline["traffic_ratio"]}
                             if roofline["bound"] != "hbm" else
                             {k: roofline[k] for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_ratio")}),
            "self_consistency_ok": self_ok,
            "setup": {"wake_up_launches": W.wake,
                      "note": "untimed launches of the same step before the W warm-up steps: brings an idle GPU's clocks to steady state; "
                              "4 + 4 more untimed steps carry the clock probe before and after the timed region"},
        }
        primary_record_to_stderr(line)
    del W
    torch.cuda.empty_cache()

    # ---- secondary workloads: the other BASELINE configs, same discipline, same command ----
    secondary = {}
    # Watchdog (N > 1 only): the primary line above is what the scaling curve is made of.  A secondary workload that HANGS — a
    # collective that never completes on a node this code has never seen (no 8-GPU box was ever available to the builder) — must
    # cost that workload, not the run: past the deadline rank 0 prints the line with what is measured so far plus
    # `secondary_timeout`, and every rank leaves.  The deadline is per workload and generous (the largest takes < 10 s at full size).
    import threading
    watch = {"deadline": None, "key": None}

    def watchdog():
        while True:
            time.sleep(1.0)
            dl = watch["deadline"]
            if dl is not None and time.time() > dl:
                if rank == 0 and line is not None:
                    line["secondary"] = secondary
                    line["secondary_timeout"] = {"workload": watch["key"], "seconds": args.secondary_timeout,
                                                 "note": "this secondary workload did not finish; the primary figures above are complete"}
                    emit(line, args.detail_file)
                print("bench.py rank %d: secondary workload %r exceeded %d s — leaving" % (rank, watch["key"], args.secondary_timeout), file=sys.stderr, flush=True)
                os._exit(0)
    if world > 1 and secondary_keys and args.secondary_timeout > 0:
        threading.Thread(target=watchdog, daemon=True).start()
    for key in secondary_keys:
        watch["key"], watch["deadline"] = key, time.time() + args.secondary_timeout
        if os.environ.get("P252_BENCH_TEST_HANG") == key:  # test-only: a workload that never returns (tests the watchdog)
            time.sleep(10 ** 6)
        s_steps = max(2, min(args.steps, 20))
        s_warm = min(args.warmup, 5)
        s_log2n = None
        if args.secondary_log2n is not None:
            s_log2n = args.secondary_log2n + (4 if key in ("tree", "forest") else 0)
        W2, el2, lm2, cb2, ca2, ok2, sample2 = measure(key, s_log2n, s_steps, s_warm)
        if rank == 0:  # (wall-clock progress on stderr: where a slow run — 8 ranks on a cold node — spends its time)
            print("bench.py: secondary %s measured, %.1f s since start" % (key, time.time() - T_START), file=sys.stderr, flush=True)
        if sample2:
            samples.append((key,) + sample2)
        if rank == 0:
            r2 = roofline_of(W2, lm2, cb2, ca2)
            secondary[key] = {
                "workload": W2.name, "units_per_gpu_per_step": W2.perms_per_step, "steps": s_steps, "warmup": s_warm,
                "wake_up_launches": W2.wake, "ms_per_step": el2 / s_steps * 1e3,
                "units_whole_job_per_step": getattr(W2, "job_units_per_step", W2.perms_per_step * world),
                "value": getattr(W2, "job_units_per_step", W2.perms_per_step * world) * s_steps / el2,
                "ms_per_step_rank_min": min(W2.rank_ms_per_step), "ms_per_step_rank_max": max(W2.rank_ms_per_step), "ms_per_step_per_rank": W2.rank_ms_per_step,
                "unit": getattr(W2, "unit", "permutations/s"), "n_gpus": world,
                "ranks": dist.get_world_size() if dist.is_initialized() else 1,
                "collective_backend": (dist.get_backend() if dist.is_initialized() else None),
                "exchange": ("all-gather of %d x 32-byte subtree roots per step" % world) if (key == "tree" and dist.is_initialized()) else None,
                "exchange_impl": getattr(W2, "exchange_impl", None),
                # (Montgomery limbs of the root the last timed step left on rank 0, most significant first: the N > 1 rehearsal test
                # compares it with the oracle tree over the concatenation of all ranks' leaves)
                "root_mont_hex": W2.root_hex() if hasattr(W2, "root_hex") else None,
                "roofline": r2, "self_consistency_ok": ok2, "parity_sample_ok": None,
            }
        del W2
        torch.cuda.empty_cache()
    watch["deadline"] = None

    if rank == 0:
        if secondary_keys:
            line["secondary"] = secondary
        if world == 1 and not args.no_cpu_baseline:
            # the cpu_baseline leg is the ONLY place bench.py touches oracle/: it times the CPU restatement and,
            # as a by-product, checks a strided sample of the GPU output of every workload just measured against it
            import poseidon252_amd as P2
            line["cpu_baseline"] = cpu_baseline(P2.merkle4_tag(), samples)
            for key, okp in line["cpu_baseline"]["parity_samples"].items():
                if key in secondary:
                    secondary[key]["parity_sample_ok"] = okp
            if line["cpu_baseline"].get("parity_sample_ok") is False:
                print("PARITY FAILURE: GPU output differs from the oracle: %s" % line["cpu_baseline"]["parity_samples"], file=sys.stderr)
                sys.exit(3)
        emit(line, args.detail_file)
    if E._comm is not None:  # the library's communicator goes first, while every rank is still here
        torch.cuda.synchronize()
        E._comm.destroy()
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
