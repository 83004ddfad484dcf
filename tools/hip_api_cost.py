#!/usr/bin/env python3
"""Host cost of one call from a rocprofv3 --hip-trace database: the HIP API calls of the last isolated batch (between two synchronizes), how
long the host spent inside them and between them.
  cd /tmp && rocprofv3 --hip-trace -d /tmp/hip -o h -- python $GRAFT_REPO_ROOT/tools/enqueue_cost.py; python tools/hip_api_cost.py /tmp/hip"""
import collections
import glob
import sqlite3
import sys

db = glob.glob(sys.argv[1] + "/**/*.db", recursive=True)[0]
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
reg = [t for t in tabs if t.startswith("rocpd_region_")][0]
st = [t for t in tabs if t.startswith("rocpd_string_")][0]
rows = list(c.execute(f"select r.start, r.end, s.string from {reg} r join {st} s on r.name_id=s.id order by r.start"))
rows = [r for r in rows if not r[2].startswith("__hip")]
blocks, cur = [], []
for s0, e0, nm in rows:
    if nm in ("hipDeviceSynchronize", "hipStreamSynchronize"):
        if cur:
            blocks.append(cur)
        cur = []
    else:
        cur.append((s0, e0, nm))
if cur:
    blocks.append(cur)
for b in blocks:
    if len(b) < 20:
        continue
    span = (b[-1][1] - b[0][0]) / 1e3
    api = sum(e - s for s, e, _ in b) / 1e3
    print(f"block of {len(b):5d} HIP calls: span {span:9.1f} us, inside HIP {api:9.1f} us, between calls {span - api:8.1f} us (incl. ~0.7 us of tracer per call)")
one = [b for b in blocks if 40 < len(b) < 200]
if one:
    b = one[-1]
    by = collections.defaultdict(lambda: [0, 0.0])
    for s0, e0, nm in b:
        by[nm][0] += 1
        by[nm][1] += (e0 - s0) / 1e3
    print("the last isolated batch:")
    for nm, (k, us) in sorted(by.items(), key=lambda kv: -kv[1][1]):
        print(f"  {nm:28s} x{k:3d}  {us:7.1f} us")
