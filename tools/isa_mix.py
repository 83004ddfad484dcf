#!/usr/bin/env python3
"""Static instruction mix of the kernels of a gfx950 assembly listing (hipcc -S --cuda-device-only): per kernel, the number of VALU
instructions by issue class, as measured by tools/microbench/valu_isa.hip, and the register / scratch / LDS figures of its metadata.
usage: isa_mix.py kernels.s [name-substring ...]"""
import collections
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
want = sys.argv[2:]
labels = [(m.start(), m.group(1)) for m in re.finditer(r'^(_Z[A-Za-z0-9_]+):', txt, re.M)]
# opcodes that tools/microbench/valu_isa.hip measured at ~2.1-2.4 cycles per wave64 instruction per SIMD once two waves share the SIMD
# (profiles/r03_valu_isa.txt); everything else these kernels use costs 4.04-4.08 (v_permlane*_swap 8.05)
FAST = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_lshrrev_b32", "v_ashrrev_i32", "v_mov_b32", "v_not_b32", "v_fma_f32"}
JSON_OUT = None
if "--json" in sys.argv:
    i = sys.argv.index("--json")
    JSON_OUT = sys.argv[i + 1]
    del sys.argv[i:i + 2]
    want = sys.argv[2:]
summary = {}
for i, (pos, name) in enumerate(labels):
    end = labels[i + 1][0] if i + 1 < len(labels) else len(txt)
    body = txt[pos:end]
    if ".amdhsa_kernel" not in body:
        continue
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    short = dem.split("(")[0].replace("tmx::", "").replace("void ", "")
    if want and not any(w in short for w in want):
        continue
    code = body[:body.find("s_endpgm") + 8]
    ops = collections.Counter()
    for line in code.split("\n"):
        m = re.match(r"^\s+([vs]_\w+|ds_\w+|global_\w+|buffer_\w+|scratch_\w+|flat_\w+)", line)
        if m:
            ops[m.group(1)] += 1
    valu = {o: c for o, c in ops.items() if o.startswith("v_")}
    meta = {k: re.search(r"\.amdhsa_" + k + r"\s+(\S+)", body) for k in ("next_free_vgpr", "next_free_sgpr", "group_segment_fixed_size", "private_segment_fixed_size")}
    meta = {k: (v.group(1) if v else "?") for k, v in meta.items()}
    tot = sum(valu.values())
    mad64 = sum(c for o, c in valu.items() if o.startswith("v_mad_i64") or o.startswith("v_mad_u64"))
    dpp = len(re.findall(r"_dpp|quad_perm|row_", code))
    cnd = sum(c for o, c in valu.items() if o.startswith("v_cndmask"))
    perm = sum(c for o, c in valu.items() if "permlane" in o)
    print(f"{short[:44]:44s} VALU {tot:6d} (mad64 {mad64:5d}, cndmask {cnd:5d}, dpp {dpp:5d}, permlane-swap {perm:4d})  SALU {sum(c for o, c in ops.items() if o.startswith('s_')):5d} "
          f"mem {sum(c for o, c in ops.items() if o.split('_')[0] in ('global', 'buffer', 'flat')):4d} lds {sum(c for o, c in ops.items() if o.startswith('ds_')):4d} "
          f"scratch {sum(c for o, c in ops.items() if o.startswith('scratch_')):4d} | vgpr {meta['next_free_vgpr']} sgpr {meta['next_free_sgpr']} lds {meta['group_segment_fixed_size']} B scratch {meta['private_segment_fixed_size']} B")
    fast = sum(c for o, c in valu.items() if o.split("_e32")[0].split("_e64")[0] in FAST and "_dpp" not in o and "_sdwa" not in o)
    summary[short.split("<")[0]] = {"static_valu": tot, "fast_class_share": round(fast / max(tot, 1), 4), "mad64": mad64, "permlane_swap": perm,
                                    "vgpr": meta["next_free_vgpr"], "scratch_bytes": meta["private_segment_fixed_size"], "lds_bytes": meta["group_segment_fixed_size"]}
    if want:
        print("     ", ", ".join(f"{o} {c}" for o, c in sorted(valu.items(), key=lambda kv: -kv[1])[:16]))

if JSON_OUT:
    import json
    json.dump({"source": "tools/isa_mix.py over hipcc -S --cuda-device-only of tendermintx_amd/csrc/kernels.hip: STATIC instruction mix per kernel; "
                         "fast_class_share = share of VALU instructions in the opcode class measured at ~2.3 cycles (profiles/r03_valu_isa.txt)",
               "kernels": summary}, open(JSON_OUT, "w"), indent=1)
