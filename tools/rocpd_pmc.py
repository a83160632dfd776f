"""Per-kernel PMC summary from a rocprofv3 rocpd database collected with --pmc: for every kernel name, calls, mean duration and the
per-dispatch mean of each counter (summed over the instances/XCDs reported per dispatch).  usage: rocpd_pmc.py results.db"""
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
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, dispatch_id, duration, counter_name, counter_value from pmc_events").fetchall()
per_dispatch = defaultdict(lambda: defaultdict(float))
dur, names = {}, {}
for name, did, d, cn, cv in rows:
    per_dispatch[did][cn] += cv
    dur[did] = d
    names[did] = short(name)
agg = defaultdict(lambda: {"n": 0, "dur": 0.0, "c": defaultdict(float)})
for did, cs in per_dispatch.items():
    a = agg[names[did]]
    a["n"] += 1
    a["dur"] += dur[did]
    for k, v in cs.items():
        a["c"][k] += v
counters = sorted({k for a in agg.values() for k in a["c"]})
print("%-60s %6s %10s " % ("kernel", "calls", "avg_us") + " ".join("%22s" % c for c in counters))
for k, a in sorted(agg.items(), key=lambda kv: -kv[1]["dur"])[:25]:
    print("%-60s %6d %10.1f " % (k[:60], a["n"], a["dur"] / a["n"] / 1e3) + " ".join("%22.4g" % (a["c"][c] / a["n"]) for c in counters))
