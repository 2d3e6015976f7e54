"""profiles/r02_isa_counts.json — the executed-instruction counts bench.py's roofline quotes — is what tools/isa_count.py
derives from the CURRENT kernel sources (the file cannot silently go stale), and the counts have the structure the
schedule predicts."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_committed_isa_counts_match_the_sources():
    import glob
    import isa_count as ic
    path = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_isa_counts.json")))[-1]
    committed = json.load(open(path))
    text = open(ic.compile_asm()).read()
    for kernel in ("k_merkle4", "k_permute", "k_merkle4_coop<8>", "k_merkle4_coop<4>"):
        now = ic.count_kernel(text, kernel)
        for key in ("valu_total", "v_mad_i64_i32", "valu_4cycle_class", "valu_2cycle_class", "valu_issue_cycles"):
            assert now[key] == committed[kernel][key], (kernel, key, now[key], committed[kernel][key])
    # round 5: the truncating build of the digest kernel (finalize_truncated in the output stage) costs next to nothing — one product and
    # one reduction more, three conditional subtractions fewer: +14 VALU instructions on 76,983 when this was written
    tr = ic.count_kernel(text, "k_merkle4_trunc")
    assert 0 <= tr["valu_total"] - committed["k_merkle4"]["valu_total"] < 200 and 0 < tr["v_mad_i64_i32"] - committed["k_merkle4"]["v_mad_i64_i32"] < 200
    m4, pm = committed["k_merkle4"], committed["k_permute"]
    # one v_mad_i64_i32 per digit product: 100 S-boxes minus the hoisted one, 60 G-products, ...
    assert 59_000 < m4["v_mad_i64_i32"] < 62_000 and m4["v_mad_i64_i32"] < pm["v_mad_i64_i32"]
    assert m4["valu_total"] < 77_500  # round 1: 84,606; round 2: 80,386
    # the cooperative digest (eight lanes per node): about two thirds of the one-lane instruction stream per lane
    co, c4 = committed["k_merkle4_coop<8>"], committed["k_merkle4_coop<4>"]
    assert co["valu_total"] < 0.67 * m4["valu_total"] and co["v_mad_i64_i32"] < 0.65 * m4["v_mad_i64_i32"]
    assert co["valu_total"] < c4["valu_total"] < 0.75 * m4["valu_total"]
    # 45 ds_bpermute per full round (8 lanes) / 36 DPP broadcasts (4 lanes), 9 DPP moves per partial round
    assert co["lds"] == 8 * 45 and co["by_mnemonic"]["v_mov_b32_dpp"] == 60 * 9
    assert c4["lds"] == 0 and c4["by_mnemonic"]["v_mov_b32_dpp"] == 60 * 9 + 8 * 36
    assert m4["lds"] == 0
    # nothing but multiply-adds, the carry chains' shift / add / mask and the steps' xor in any quantity
    top = list(m4["by_mnemonic"].items())[:6]
    assert top[0][0] == "v_mad_i64_i32" and {k for k, _ in top[1:]} <= {"v_and_b32_e32", "v_lshl_add_u64", "v_xor_b32_e32", "v_ashrrev_i64",
                                                                       "v_add_u32", "v_mov_b32_e32", "v_alignbit_b32"}


def test_bench_reads_the_counts():
    sys.path.insert(0, ROOT)
    import bench
    c = bench.isa_counts("k_merkle4")
    assert c["v_mad_i64_i32"] > 60_000 and "_isa_counts.json:k_merkle4" in c["source"]
    assert bench.isa_counts("k_sponge")["source"].endswith("k_permute")
    assert abs(bench.PEAK_INT32_MAC_PER_S - 39.32e12) < 0.01e12
