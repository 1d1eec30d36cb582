#!/usr/bin/env python3
"""Turn rocprofv3's rocpd sqlite output (what `rocprofv3 --kernel-trace --stats` / `--pmc X` write on this ROCm 7.2
image) into the small text summaries committed under profiles/.

  python profiles/summarize.py stats  gpurun_out/prof_stats/<host>/<pid>_results.db  > profiles/rNN_kernel_stats.md
  python profiles/summarize.py pmc    gpurun_out/prof_fetch/<host>/<pid>_results.db  > profiles/rNN_pmc_fetch.md
"""
import sqlite3
import sys


def stats(path):
    c = sqlite3.connect(path).cursor()
    print("| kernel | calls | total ms | avg us | % | vgpr | sgpr | lds B | scratch B | grid | wg |")
    print("|---|---|---|---|---|---|---|---|---|---|---|")
    rows = c.execute(
        "select name, count(*), sum(duration), avg(duration), max(vgpr_count), max(sgpr_count), max(lds_size), "
        "max(scratch_size), max(grid_x), max(workgroup_x) from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    for r in rows:
        print("| `%s` | %d | %.3f | %.1f | %.1f | %d | %d | %d | %d | %d | %d |" % (
            r[0][:90], r[1], r[2] / 1e6, r[3] / 1e3, 100.0 * r[2] / total, r[4], r[5], r[6], r[7], r[8], r[9]))


def pmc(path):
    c = sqlite3.connect(path).cursor()
    print("| kernel | counter | dispatches | avg value | sum value | avg dispatch us |")
    print("|---|---|---|---|---|---|")
    rows = c.execute(
        "select kernel_name, counter_name, count(*), avg(value), sum(value), avg(duration) from counters_collection "
        "group by kernel_name, counter_name order by sum(value) desc").fetchall()
    for r in rows:
        print("| `%s` | %s | %d | %.1f | %.1f | %.1f |" % (r[0][:90], r[1], r[2], r[3], r[4], r[5] / 1e3))


if __name__ == "__main__":
    {"stats": stats, "pmc": pmc}[sys.argv[1]](sys.argv[2])
