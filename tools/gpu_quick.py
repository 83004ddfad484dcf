#!/usr/bin/env python3
"""Quick GPU-vs-oracle parity + timing loop over tests/golden/cases.json (development aid; the real tests are in tests/)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle", "py"))
import oracle_c as oc  # noqa: E402
from tendermintx_amd import Context  # noqa: E402

cases = json.load(open(os.path.join(ROOT, "tests", "golden", "cases.json")))
bad = 0
for name, c in sorted(cases.items()):
    kind, n = c["kind"], c["n"]
    proof, target = bytes.fromhex(c["proof"]), bytes.fromhex(c["target"])
    trusted = bytes.fromhex(c["trusted"]) if c["trusted"] else None
    t0 = time.time()
    ctx = Context(n, c["chain_id"].encode(), c["skip_max"], max_batch=2)
    t1 = time.time()
    elems, reps = ctx.witness_batch(kind, proof, target, trusted)
    t2 = time.time()
    ms = ctx.last_kernel_ms()
    ow, orep = oc.witness(kind, proof, target, trusted, c["chain_id"].encode(), c["skip_max"])
    same = np.array_equal(elems[0], ow)
    rep = reps[0]
    rep_ok = all(rep[k] == orep[k] for k in ("header", "all_ok", "fail_mask", "first_bad_sig", "gt_target", "gt_trusted", "dist_ok"))
    print(f"{name:34s} n={n:4d} elems_equal={same} report_equal={rep_ok} ctx={t1-t0:.2f}s call={1e3*(t2-t1):.1f}ms kernels={ms}")
    if not same:
        diff = np.nonzero(elems[0] != ow)[0]
        print("   first diffs at", diff[:20], "count", len(diff), "of", len(ow))
        print("   gpu", elems[0][diff[:8]], "oracle", ow[diff[:8]])
        bad += 1
    if not rep_ok:
        print("   gpu rep", rep, "\n   oracle ", orep)
        bad += 1
    ctx.close()
print("FAILURES:", bad)
sys.exit(1 if bad else 0)
