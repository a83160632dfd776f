"""Dispatch timeline of the LAST step in a rocprofv3 rocpd database: every kernel in start order with its duration, the gap to the
previous kernel's end and the grid size (to tell the GEMM shapes apart).  usage: rocpd_timeline.py results.db [first_kernel_substring]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
marker = sys.argv[2] if len(sys.argv) > 2 else "knn_kernel<9>"
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else "kernel_name"
grid = [c for c in ("grid_x", "grid_size_x", "grid_size") if c in cols]
sel = "%s, start, end%s" % (name_col, "".join(", " + g for g in grid[:1]))
rows = sorted(db.execute("select %s from kernels" % sel).fetchall(), key=lambda r: r[1])
starts = [i for i, r in enumerate(rows) if marker in r[0]]
rows = rows[starts[-1]:] if starts else rows
t0, prev_end = rows[0][1], rows[0][1]
for r in rows:
    name = r[0].replace("(anonymous namespace)::", "").split("(")[0][:60]
    print("%9.1f us  +%7.1f  dur %8.1f  grid %-8s %s" % ((r[1] - t0) / 1e3, (r[1] - prev_end) / 1e3, (r[2] - r[1]) / 1e3, r[3] if len(r) > 3 else "", name))
    prev_end = max(prev_end, r[2])
print("span %.1f us" % ((prev_end - t0) / 1e3))
