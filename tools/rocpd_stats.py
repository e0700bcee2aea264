#!/usr/bin/env python3
"""rocprofv3 (rocpd sqlite output) -> the per-kernel stats CSV that `--stats` prints with the csv output format.
usage: rocpd_stats.py results.db out.csv"""
import csv
import math
import sqlite3
import sys
con = sqlite3.connect(sys.argv[1])
rows = {}
for name, dur in con.execute("select name, duration from kernels"):
    rows.setdefault(name, []).append(dur)
total = sum(sum(v) for v in rows.values())
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
    for name, v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
        n, s = len(v), sum(v)
        mean = s / n
        sd = math.sqrt(sum((d - mean) ** 2 for d in v) / (n - 1)) if n > 1 else 0.0
        w.writerow([name, n, s, round(mean, 6), round(100.0 * s / total, 2), min(v), max(v), round(sd, 6)])
