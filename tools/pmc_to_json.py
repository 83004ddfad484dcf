#!/usr/bin/env python3
"""Per-batch PMC figures from the databases tools/collect_profiles.sh wrote: pmc_to_json.py gpurun_out/<tag> > pmc.json"""
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


res = {"source": f"tools/collect_profiles.sh: rocprofv3 --pmc passes (separate runs) over tools/profile_step.py, {nb} full batches each; per-batch = sum / {nb}",
       "config": {"n_max": int(os.environ.get("N", "128")), "proofs_per_gpu": int(os.environ.get("P", "256")), "workload": os.environ.get("WORKLOAD", "survey8d")},
       "unit_note": "fetch_kb / write_kb: FETCH_SIZE / WRITE_SIZE (KB) per batch, uncorrected (MI355X_MICROARCH.md: FETCH_SIZE under-reports wide coalesced reads by 2x; "
                    "the reads here are mostly 1- and 4-byte accesses).  valu_insts / salu_insts: wave-level instructions per batch.",
       "setup_note": "`setup` = k_init_base / k_init_base_quad: run once per context, listed per profiled run / batches, NOT part of a batch's k_eddsa sum "
                     "(until r02d these 9e6 instructions per profiled batch were counted into k_eddsa)",
       "kernels": {}, "per_kernel": {}}
for db_path in sorted(glob.glob(os.path.join(out, "pmc_*", "**", "*.db"), recursive=True)):
    db = sqlite3.connect(db_path)
    for name, counter, total in db.execute("select kernel_name, counter_name, sum(value) from counters_collection group by kernel_name, counter_name"):
        g = group(name)
        if g is None:
            continue
        key = {"FETCH_SIZE": "fetch_kb", "WRITE_SIZE": "write_kb", "SQ_INSTS_VALU": "valu_insts", "SQ_INSTS_SALU": "salu_insts", "SQ_WAVES": "waves",
               "GRBM_GUI_ACTIVE": "gui_active_cycles_sum_xcd", "SQ_BUSY_CYCLES": "sq_busy_cycles"}.get(counter)
        if key is None:
            continue
        res["kernels"].setdefault(g, {})
        res["kernels"][g][key] = round(res["kernels"][g].get(key, 0) + total / nb, 1)
        short = name.split("(")[0].replace("void ", "").replace("tmx::", "")
        res["per_kernel"].setdefault(short, {})
        res["per_kernel"][short][key] = round(res["per_kernel"][short].get(key, 0) + total / nb, 1)
# concurrent durations of the same script (kernel-trace only run)
for db_path in sorted(glob.glob(os.path.join(out, "trace_step", "**", "*.db"), recursive=True)):
    db = sqlite3.connect(db_path)
    for name, total in db.execute("select name, sum(end - start) from kernels group by name"):
        if "tmx::" not in name:
            continue
        short = name.split("(")[0].replace("void ", "").replace("tmx::", "")
        res["per_kernel"].setdefault(short, {})["us_per_batch_concurrent"] = round(total / nb / 1e3, 1)
print(json.dumps(res, indent=1))
