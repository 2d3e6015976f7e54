#!/usr/bin/env python3
"""Summarise a rocprofv3 results.db (rocpd sqlite) into the text table kept under profiles/.
usage: rocprof_summary.py <results.db> [title] [warmup_launches=10] [primary_steps=50]"""
import sqlite3
import sys


def short_name(name, width):
    """kernel name without its argument list; '(anonymous namespace)::' is part of the NAME, not the start of the arguments"""
    return name.replace("(anonymous namespace)::", "").split("(")[0][-width:]


def primary_region(cur, steps):
    """bench.py's PRIMARY workload is the first thing a run launches: wake-up, warm-up, 4 probe steps, `steps` TIMED launches, 4 probe
    steps — all of one (kernel, grid) — and then its self-check, which launches something else.  Later launches of the same kernel and
    grid (a 2^24-leaf tree's level of 2^20 nodes in the secondary workloads) do NOT belong to it: the per-grid average above mixes them
    in, this section does not.  Returns (name, grid, durations of the timed launches) or None."""
    rows = list(cur.execute("select name, grid_x, duration from kernels where name like '%p252::%' and name not like '%k_clock_probe%' order by start"))
    if not rows:
        return None
    name, grid = rows[0][0], rows[0][1]
    region = []
    for n, g, d in rows:
        if (n, g) != (name, grid):
            break
        region.append(d)
    if len(region) < steps + 8:
        return None
    return name, grid, region[-(steps + 4):-4]


def main(path, title="", warm=10, steps=50):
    db = sqlite3.connect(path)
    cur = db.cursor()
    print("# rocprofv3 --kernel-trace --stats summary%s" % ((": " + title) if title else ""))
    print("# source db: %s" % path.split("/")[-1])
    print("%-60s %8s %14s %12s %12s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct"))
    rows = list(cur.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by name order by sum(duration) desc"))
    tot = sum(r[2] for r in rows) or 1
    for name, calls, total, avg, mn, mx in rows:
        short = short_name(name, 58)
        print("%-60s %8d %14.1f %12.1f %12.1f %12.1f %6.2f%%" % (short, calls, total / 1e3, avg / 1e3, mn / 1e3, mx / 1e3, 100.0 * total / tot))
    print()
    print("# timed launches only: grid == the kernel's largest grid (self-check launches of other sizes left out) and the")
    print("# first %d such launches skipped (bench.py's warm-up: the clocks ramp from idle over ~10 launches)" % warm)
    print("%-60s %8s %12s %12s %12s %14s" % ("kernel", "calls", "avg_us", "min_us", "max_us", "warmup_avg_us"))
    for (name,) in list(cur.execute("select distinct name from kernels")):
        gmax = cur.execute("select max(grid_x) from kernels where name = ?", (name,)).fetchone()[0]
        durs = [r[0] for r in cur.execute("select duration from kernels where name = ? and grid_x = ? order by start", (name, gmax))]
        if len(durs) > warm + 1:
            rest, head = durs[warm:], durs[:warm]
            print("%-60s %8d %12.1f %12.1f %12.1f %14.1f" % (short_name(name, 58), len(rest), sum(rest) / len(rest) / 1e3, min(rest) / 1e3,
                                                             max(rest) / 1e3, sum(head) / len(head) / 1e3))
    print()
    print("# the library's kernels per launch size (one default bench.py run launches several: configs[1]'s 2^20 digests, the tree's")
    print("# twelve levels, the sponge).  `steady` = the launches after the first %d of that size (wake-up, warm-up, probe steps)." % warm)
    print("%-44s %10s %7s %12s %12s %12s %12s" % ("kernel", "grid", "calls", "avg_us", "min_us", "steady_calls", "steady_avg_us"))
    for name, grid in list(cur.execute("select name, grid_x from kernels where name like '%p252::%' group by name, grid_x order by name, grid_x desc")):
        durs = [r[0] for r in cur.execute("select duration from kernels where name = ? and grid_x = ? order by start", (name, grid))]
        rest = durs[warm:] if len(durs) > warm + 1 else []
        print("%-44s %10d %7d %12.1f %12.1f %12s %12s" % (short_name(name, 42), grid, len(durs), sum(durs) / len(durs) / 1e3, min(durs) / 1e3,
                                                         len(rest) if rest else "-", ("%.1f" % (sum(rest) / len(rest) / 1e3)) if rest else "-"))
    pr = primary_region(cur, steps)
    if pr:
        name, grid, durs = pr
        print()
        print("# the PRIMARY's timed region alone: the %d launches bench.py times (HIP events: roofline.launch_ms_mean), i.e. the launches of the run's first"
              % steps)
        print("# kernel and grid up to its self-check minus wake-up / warm-up / probe steps; later launches of the same grid (tree levels) excluded")
        print("%-44s %10s %7s %12s %12s %12s" % ("kernel", "grid", "calls", "avg_us", "min_us", "max_us"))
        print("%-44s %10d %7d %12.1f %12.1f %12.1f" % (short_name(name, 42), grid, len(durs), sum(durs) / len(durs) / 1e3, min(durs) / 1e3, max(durs) / 1e3))
    print()
    print("# per-kernel launch geometry / registers (first dispatch)")
    for r in cur.execute("select name, grid_x, workgroup_x, vgpr_count, accum_vgpr_count, sgpr_count, lds_size, scratch_size from kernels group by name"):
        print("%-60s grid=%d wg=%d vgpr=%s agpr=%s sgpr=%s lds=%s scratch=%s" % ((r[0].split("(")[0][-58:],) + tuple(r[1:])))
    try:
        rows = list(cur.execute("select name, counter_name, avg(value), sum(value), count(*) from counters_collection group by name, counter_name"))
        if rows:
            print()
            print("# PMC counters (avg per dispatch, sum, dispatches)")
            for name, cname, avg, sm, n in rows:
                print("%-40s %-28s avg=%.4g sum=%.4g n=%d" % (short_name(name, 38), cname, avg, sm, n))
    except sqlite3.Error:
        pass


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "", int(sys.argv[3]) if len(sys.argv) > 3 else 10,
         int(sys.argv[4]) if len(sys.argv) > 4 else 50)
