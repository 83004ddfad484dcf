#!/usr/bin/env python3
"""Print the kernel timeline of one full-batch step from a rocprofv3 rocpd database (development aid)."""
import sqlite3
import sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name,start,end,stream_id from kernels order by start").fetchall()
# a step starts with k_ed_dedup; print the step before the last one
# (a context computes its dummy record with ONE small launch when it is created: a run of small-path steps has more k_tiny launches than that)
# (the small path: a step starts with k_tiny, its key pipeline -- k_ed_dedup ... -- follows on the side stream)
tiny = [i for i, r in enumerate(rows) if "k_tiny" in r[0] and "k_tiny_tail" not in r[0]]
ded = (tiny if len(tiny) >= 3 else None) or [i for i, r in enumerate(rows) if "k_ed_dedup" in r[0]] or [i for i, r in enumerate(rows) if "k_ed_keys" in r[0]]
i1, i2 = ded[-2] - 1, ded[-1] - 1
t0 = None
for r in rows[i1 + 1:i2 + 1]:
    if t0 is None:
        t0 = r[1]
    print(f"{r[0][:30]:30s} start {((r[1]-t0)/1e3):9.1f} us  dur {((r[2]-r[1])/1e3):8.1f} us  stream {r[3]}")
