"""bench.py quotes HBM traffic and VALU counts from counter passes committed under profiles/.  Every such summary records
the SHA-256 of the kernel sources it was collected from (tools/pmc_summary.py); bench.py reports a summary of other sources
as stale instead of using it, and this test keeps the files behind the DEFAULT bench line (configs[1] + the secondary tree
and sponge42) fresh: changing a kernel without re-running tools/run_pmc.sh on the GPU fails here."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_hash_functions_agree():
    import bench
    import pmc_summary
    assert bench.KERNEL_SOURCES == pmc_summary.KERNEL_SOURCES
    assert bench.kernel_sources_sha256() == pmc_summary.kernel_sources_sha256()
    # the extraction kernel has a source file of its own: only ITS counter passes go stale with it
    assert bench.kernel_sources_sha256("k_merkle4_openings") == pmc_summary.kernel_sources_sha256("k_merkle4_openings") != bench.kernel_sources_sha256()


def test_default_line_reads_counter_passes_of_the_current_sources():
    import bench
    now = bench.kernel_sources_sha256()
    for kernel in ("k_merkle4", "k_sponge", "tree"):
        d = bench.pmc_profile(kernel)
        assert d is not None, "no profiles/r*_pmc_%s.json" % kernel
        assert d.get("kernel_sources_sha256") == now and d["stale"] is False, \
            "%s was collected from other kernel sources: re-run `bash tools/run_pmc.sh <workload>` on the GPU and commit the summaries" % d["source"]
    t = bench.pmc_traffic("k_merkle4", "merkle4_digests", 1 << 20)
    assert t and 0.95 < t["ratio"] < 1.10, t
    t = bench.pmc_traffic("k_merkle4", "tree", 5592405)  # (level by level: every level is written and read back)
    assert t and 1.5 < t["ratio"] < 1.9 and 0.95 < t["ratio_level_by_level"] < 1.10, t


def test_a_stale_summary_is_reported_not_used(tmp_path, monkeypatch):
    import bench
    d = bench.pmc_profile("k_merkle4")
    assert d is not None
    fake = dict(d, kernel_sources_sha256="0" * 64)
    fake.pop("source", None)
    fake.pop("stale", None)
    prof = tmp_path / "profiles"
    prof.mkdir()
    (prof / "r99_pmc_k_merkle4.json").write_text(json.dumps(fake))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    os.makedirs(tmp_path / "poseidon252_amd" / "csrc")
    for f in bench.KERNEL_SOURCES:
        (tmp_path / "poseidon252_amd" / "csrc" / f).write_bytes(open(os.path.join(ROOT, "poseidon252_amd", "csrc", f), "rb").read())
    assert bench.pmc_profile("k_merkle4")["stale"] is True
    assert bench.pmc_valu("k_merkle4") is None
    t = bench.pmc_traffic("k_merkle4", "merkle4_digests", 1 << 20)
    assert t["bytes"] is None and "stale_source" in t
