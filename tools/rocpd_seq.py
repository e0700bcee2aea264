#!/usr/bin/env python3
"""ordered kernel sequence of the LAST step in a rocprofv3 rocpd database (kernel-trace): start offset, duration, queue,
name, grid.   python tools/rocpd_seq.py results.db [kernels_per_step]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "info_kernel_symbol" in t][0]
rows = list(cur.execute(f"select d.start, d.end, d.queue_id, s.display_name, d.grid_size_x, d.grid_size_y, d.grid_size_z, d.workgroup_size_x "
                        f"from {kd} d join {ks} s on d.kernel_id = s.id order by d.start"))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 0
if not n:
    # FlagStep (round 5): the optimizer runs at the HEAD of the next replay -- a step is the stretch between two adam_prepare launches
    # (the replay in front of the last one: the run ends with a flush that is only the optimizer); else a step ends with the optimizer kernel
    heads = [i for i, r in enumerate(rows) if "adam_prepare" in r[3]]
    if len(heads) >= 3:
        rows = rows[heads[-3]:heads[-2]]
    else:
        ends = [i for i, r in enumerate(rows) if "adam" in r[3]]
        rows = rows[ends[-2] + 1:ends[-1] + 1]
else:
    rows = rows[-n:]
t0 = rows[0][0]
for r in rows:
    print(f"{(r[0]-t0)/1e3:8.1f} {(r[1]-r[0])/1e3:7.1f} q{r[2]} {r[3][:100]} g={r[4]//max(r[7],1)}x{r[5]}x{r[6]}")
print("launches", len(rows), "span us", (rows[-1][1] - t0) / 1e3, "sum us", sum(r[1] - r[0] for r in rows) / 1e3)
