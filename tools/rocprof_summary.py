#!/usr/bin/env python3
"""Summarise a rocprofv3 results.db (rocpd sqlite) into the text table kept under profiles/.
usage: rocprof_summary.py <results.db> [title] [warmup_launches=10]"""
import sqlite3
import sys


def short_name(name, width):
    """kernel name without its argument list; '(anonymous namespace)::' is part of the NAME, not the start of the arguments"""
    return name.replace("(anonymous namespace)::", "").split("(")[0][-width:]


def main(path, title="", warm=10):
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
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "", int(sys.argv[3]) if len(sys.argv) > 3 else 10)
