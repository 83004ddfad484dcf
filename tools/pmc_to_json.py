#!/usr/bin/env python3
"""Per-batch PMC figures from the databases tools/collect_profiles.sh wrote: pmc_to_json.py gpurun_out/<tag> > pmc.json
The batches are told apart by dispatch id (k_proof is the first kernel a batch enqueues); the WARM batches in front -- the first of them
builds every key table, the second still runs the cold schedule -- are left out, so that a per-batch figure is that of the warm steps
bench.py times.  `cold_batch` lists the first batch on its own."""
import glob
import json
import os
import sqlite3
import sys

out = sys.argv[1]
WARM, STEPS = int(os.environ.get("WARM", "2")), int(os.environ.get("STEPS", "6"))
nb = WARM + STEPS


def group(name):
    if "k_serialize" in name:
        return "k_serialize"
    if "k_proof" in name:
        return "k_proof"
    if "k_verdict" in name:
        return "k_verdict"
    if "k_init_base" in name:
        return "setup"  # built once per context (the table of B): not part of a batch
    if "tmx::" in name:
        return "k_eddsa"
    return None


res = {"source": f"tools/collect_profiles.sh: rocprofv3 --pmc passes (separate runs) over tools/profile_step.py, {nb} full batches each; per-batch = sum over the last {STEPS} (key cache warm) / {STEPS}",
       "config": {"n_max": int(os.environ.get("N", "128")), "proofs_per_gpu": int(os.environ.get("P", "256")), "workload": os.environ.get("WORKLOAD", "survey8d")},
       "unit_note": "fetch_kb / write_kb: FETCH_SIZE / WRITE_SIZE (KB) per batch, uncorrected (MI355X_MICROARCH.md: FETCH_SIZE under-reports wide coalesced reads by 2x; "
                    "the reads here are mostly 1- and 4-byte accesses).  valu_insts / salu_insts: wave-level instructions per batch.",
       "setup_note": "`setup` = k_init_base / k_init_base_quad: run once per context, listed per profiled run / batches, NOT part of a batch's k_eddsa sum "
                     "(until r02d these 9e6 instructions per profiled batch were counted into k_eddsa)",
       "kernels": {}, "per_kernel": {}}
KEYS = {"FETCH_SIZE": "fetch_kb", "WRITE_SIZE": "write_kb", "SQ_INSTS_VALU": "valu_insts", "SQ_INSTS_SALU": "salu_insts", "SQ_WAVES": "waves",
        "GRBM_GUI_ACTIVE": "gui_active_cycles_sum_xcd", "SQ_BUSY_CYCLES": "sq_busy_cycles"}
res["cold_batch"] = {}
for db_path in sorted(glob.glob(os.path.join(out, "pmc_*", "**", "*.db"), recursive=True)):
    db = sqlite3.connect(db_path)
    starts = sorted(r[0] for r in db.execute("select distinct dispatch_id from counters_collection where kernel_name like '%k_proof%'"))
    if len(starts) != nb:
        print(f"{db_path}: {len(starts)} batches found, {nb} expected", file=sys.stderr)
    first_timed = starts[WARM] if len(starts) > WARM else 0
    second = starts[1] if len(starts) > 1 else 1 << 62
    for name, counter, did, value in db.execute("select kernel_name, counter_name, dispatch_id, sum(value) from counters_collection group by kernel_name, counter_name, dispatch_id"):
        g, key = group(name), KEYS.get(counter)
        if g is None or key is None:
            continue
        short = name.split("(")[0].replace("void ", "").replace("tmx::", "")
        if g == "setup" or did >= first_timed:
            div = nb if g == "setup" else STEPS
            res["kernels"].setdefault(g, {})
            res["kernels"][g][key] = res["kernels"][g].get(key, 0) + value / div
            res["per_kernel"].setdefault(short, {})
            res["per_kernel"][short][key] = res["per_kernel"][short].get(key, 0) + value / div
        if g != "setup" and starts and starts[0] <= did < second:
            res["cold_batch"].setdefault(short, {})
            res["cold_batch"][short][key] = res["cold_batch"][short].get(key, 0) + value
for tab in (res["kernels"], res["per_kernel"], res["cold_batch"]):
    for v in tab.values():
        for k in v:
            v[k] = round(v[k], 1)
# concurrent durations of the same script (kernel-trace only run)
for db_path in sorted(glob.glob(os.path.join(out, "trace_step", "**", "*.db"), recursive=True)):
    db = sqlite3.connect(db_path)
    t_first = sorted(r[0] for r in db.execute("select start from kernels where name like '%k_proof%'"))
    t0 = t_first[WARM] if len(t_first) > WARM else 0
    for name, total in db.execute("select name, sum(end - start) from kernels where start >= ? group by name", (t0,)):
        if "tmx::" not in name:
            continue
        short = name.split("(")[0].replace("void ", "").replace("tmx::", "")
        res["per_kernel"].setdefault(short, {})["us_per_batch_concurrent"] = round(total / STEPS / 1e3, 1)
print(json.dumps(res, indent=1))
