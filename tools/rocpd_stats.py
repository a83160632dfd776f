"""Per-kernel summary (calls, total/avg/min/max duration, share) from a rocprofv3 rocpd sqlite database
(`rocprofv3 --kernel-trace --stats` default output on ROCm 7.2).  usage: rocpd_stats.py results.db [> profiles/x.txt]"""
import re
import sqlite3
import sys


def short(name):
    """kernel name without the argument list (cut at the first '(' outside template brackets)"""
    name = name.replace("(anonymous namespace)::", "")
    depth = 0
    for i, ch in enumerate(name):
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            return name[:i]
    return name

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else "kernel_name"
rows = db.execute("select %s, start, end from kernels" % name_col).fetchall()
agg = {}
for name, s, e in rows:
    a = agg.setdefault(short(name), [0, 0, 10 ** 18, 0])
    d = e - s
    a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
total = sum(a[1] for a in agg.values())
print("%-72s %7s %12s %10s %10s %10s %6s" % ("kernel", "calls", "total_ms", "avg_us", "min_us", "max_us", "%"))
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-72s %7d %12.3f %10.1f %10.1f %10.1f %6.2f" % (k[:72], a[0], a[1] / 1e6, a[1] / a[0] / 1e3, a[2] / 1e3, a[3] / 1e3, 100.0 * a[1] / total))
print("%-72s %7d %12.3f" % ("TOTAL", sum(a[0] for a in agg.values()), total / 1e6))
