#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV passes (gpurun_out/pmc_*/pmc_counter_collection.csv) for one kernel.
usage: pmc_summary.py <kernel-substring> <units_per_launch> [--bytes-per-unit B] [--per-step] <dir> [<dir> ...]
-> text on stdout, JSON on fd 3 if open.  units = permutations per launch; B = algorithmic bytes per permutation.
--per-step: a step is SEVERAL launches of different sizes (a tree build: one launch per level, several kernel builds):
every dispatch whose name contains the substring is summed, and the sum is divided by the number of steps (= dispatches of
the largest grid, the first level); run the workload with --no-check so that nothing but full steps is in the trace.
The JSON records the SHA-256 of the kernel sources the pass was collected from (bench.py refuses a file that belongs to
other sources)."""
import collections
import csv
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL_SOURCES = ("kernels.hip", "fr29.hpp", "hades29.hpp", "coop29.hpp", "tables.hpp", "kernels.h")  # = bench.py's


def kernel_sources_sha256(kernel=None):
    h = hashlib.sha256()
    for f in KERNEL_SOURCES + (("openings.hip", "fastdiv.hpp") if kernel and "openings" in kernel else ()):  # = bench.py's rule
        h.update(open(os.path.join(ROOT, "poseidon252_amd", "csrc", f), "rb").read())
    return h.hexdigest()


def main():
    kern, units = sys.argv[1], float(sys.argv[2])
    rest = sys.argv[3:]
    bpu = 160.0
    if rest and rest[0] == "--bytes-per-unit":
        bpu = float(rest[1])
        rest = rest[2:]
    per_step = False
    if rest and rest[0] == "--per-step":
        per_step = True
        rest = rest[1:]
    agg = collections.defaultdict(list)
    durs = []
    rows = []
    for d in rest:
        rows += [r for r in csv.DictReader(open(os.path.join(d, "pmc_counter_collection.csv"))) if kern in r["Kernel_Name"]]
    full = max(int(r["Grid_Size"]) for r in rows)  # the timed launches; smaller self-check launches of the same kernel are left out
    n_full = collections.Counter()
    for r in rows:
        if int(r["Grid_Size"]) == full:
            n_full[r["Counter_Name"]] += 1
        if int(r["Grid_Size"]) == full or per_step:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            if int(r["Grid_Size"]) == full:
                if r.get("Start_Timestamp") and r.get("End_Timestamp"):
                    durs.append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
                meta = (r["Grid_Size"], r["Workgroup_Size"], r["VGPR_Count"], r["SGPR_Count"], r["Scratch_Size"], r["LDS_Block_Size"])
    if per_step:
        avg = {k: sum(v) / n_full[k] for k, v in agg.items()}
        print("# rocprofv3 --pmc summary for the kernels *%s* of one STEP (all launches of a step summed; %d steps; separate passes per counter group)" % (kern, next(iter(n_full.values()))))
    else:
        avg = {k: sum(v) / len(v) for k, v in agg.items()}
        print("# rocprofv3 --pmc summary for kernel *%s* (avg per dispatch over %d dispatches, separate passes per counter group)" % (kern, len(next(iter(agg.values())))))
    print("# grid=%s wg=%s vgpr=%s sgpr=%s scratch=%s lds=%s ; units (permutations) per launch = %d" % (meta + (units,)))
    for k in sorted(avg):
        print("%-24s %.6g" % (k, avg[k]))
    out = {"kernel": kern, "units_per_launch": units, "counters": avg, "algorithmic_bytes_per_unit": bpu, "per_step": per_step,
           "kernel_sources_sha256": kernel_sources_sha256(kern)}
    print("# kernel sources sha256 %s" % out["kernel_sources_sha256"][:16])
    if durs:
        durs.sort()
        out["avg_duration_us"] = sum(durs) / len(durs)
        out["median_duration_us"] = durs[len(durs) // 2]
        print("%-24s %.6g   (median %.6g; under counter collection, dispatches are serialised)" % ("avg_duration_us", out["avg_duration_us"], out["median_duration_us"]))
    if "FETCH_SIZE" in avg and "WRITE_SIZE" in avg:
        # MI355X_MICROARCH.md §HBM: FETCH_SIZE/WRITE_SIZE are in KB; on gfx950 FETCH_SIZE reports exactly half
        # of the bytes of a wide (16 B/lane) streaming read -> doubled; WRITE_SIZE taken as is (it equals the
        # output size to the byte here, which calibrates it for this access pattern).
        rd = 2.0 * avg["FETCH_SIZE"] * 1024.0
        wr = avg["WRITE_SIZE"] * 1024.0
        out["hbm_bytes_per_launch"] = rd + wr
        print("hbm_read_bytes (2 x FETCH_SIZE KB)   %.6g" % rd)
        print("hbm_write_bytes (WRITE_SIZE KB)      %.6g" % wr)
        print("hbm_bytes_per_launch                 %.6g   (algorithmic: %.6g = %.4g B x units)" % (rd + wr, bpu * units, bpu))
        print("traffic / algorithmic                %.3f" % ((rd + wr) / (bpu * units)))
        out["traffic_ratio"] = (rd + wr) / (bpu * units)
    if "SQ_INSTS_VALU" in avg and "SQ_WAVES" in avg and not per_step:
        per_wave = avg["SQ_INSTS_VALU"] / avg["SQ_WAVES"]
        out["valu_insts_per_wave"] = per_wave
        print("VALU instructions per wave (= per 64 permutations)   %.0f" % per_wave)
        if "GRBM_GUI_ACTIVE" in avg:
            cyc = avg["GRBM_GUI_ACTIVE"] / 8.0  # summed over the 8 XCDs
            out["gpu_cycles_per_launch"] = cyc
            print("GPU cycles per launch (GRBM_GUI_ACTIVE / 8 XCDs)     %.4g" % cyc)
            if durs:
                print("shader clock during the kernel (cycles / median duration)   %.3f GHz" % (cyc / (out["median_duration_us"] * 1e3)))
                out["clock_ghz"] = cyc / (out["median_duration_us"] * 1e3)
            print("VALU instructions per SIMD-cycle (1024 SIMDs)        %.3f   (a 4-cycle-class stream saturates at 0.25, 2-cycle at 0.5)" % (avg["SQ_INSTS_VALU"] / (1024.0 * cyc)))
    try:
        os.write(3, json.dumps(out).encode())
    except OSError:
        pass


if __name__ == "__main__":
    main()
