#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace results .db (rocpd sqlite) into a per-kernel table:
   python tools/prof_summary.py gpurun_out/prof/x_results.db [out.csv]"""
import re
import sqlite3
import sys


def short(name):
    m = re.search(r"(k_[a-z0-9_]+|__amd_rocclr_\w+|rocprim::\w+[^<(]*)(<[^(]*>)?", name)
    if not m:
        return name[:70]
    t = m.group(2) or ""
    t = re.sub(r"\s+", "", t)
    return (m.group(1) + t)[:70]


def main():
    c = sqlite3.connect(sys.argv[1])
    rows = c.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                     "from kernels group by name order by 3 desc").fetchall()
    tot = float(sum(r[2] for r in rows))
    lines = ["kernel,calls,total_us,avg_us,min_us,max_us,percent"]
    for r in rows:
        lines.append("%s,%d,%.1f,%.1f,%.1f,%.1f,%.2f" % (short(r[0]).replace(",", ";"), r[1], r[2] / 1e3, r[3] / 1e3,
                                                        r[4] / 1e3, r[5] / 1e3, 100.0 * r[2] / tot))
    out = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out)
    sys.stdout.write(out)


if __name__ == "__main__":
    main()
