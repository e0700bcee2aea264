#!/usr/bin/env python3
"""Kernel timeline of the last step in a rocprofv3 rocpd database: one line per kernel with its queue, start offset and
duration.  usage: trace_timeline.py results.db"""
import sqlite3
import sys
con = sqlite3.connect(sys.argv[1])
rows = list(con.execute("select start, end, name, grid_x, grid_y, grid_z, queue_id, workgroup_x from kernels order by start"))
idx = [i for i, r in enumerate(rows) if 'adam_kernel' in r[2]]
a, b = idx[-2] + 1, idx[-1] + 1
t0 = rows[a][0]
busy = {}
for r in rows[a:b]:
    nm = r[2].replace('(anonymous namespace)::', '').replace('void ', '')
    for junk in ('at::native::', 'vectorized_elementwise_kernel', 'std::array'):
        nm = nm.replace(junk, '')
    q = r[6]
    busy[q] = busy.get(q, 0) + (r[1] - r[0]) / 1e3
    pad = '' if q == min(x[6] for x in rows[a:b]) else ' ' * 60
    print(f"{(r[0]-t0)/1e3:8.1f} {(r[1]-r[0])/1e3:6.1f} {pad}q{q} {nm[:52]} wg={r[3]//max(r[7],1)}x{r[4]}x{r[5]}")
print('launches', b - a, 'span us', (rows[b - 1][1] - t0) / 1e3, 'busy per queue', busy)
