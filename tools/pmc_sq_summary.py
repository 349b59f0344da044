#!/usr/bin/env python3
"""Per-kernel means of the SQ / GRBM counters of the rocprofv3 --pmc passes under <dir>/sq*/ for the persistent kernels.

    python tools/pmc_sq_summary.py <dir> <out.json>

SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles per wave summed over the waves; SQ_BUSY_CYCLES per SE;
SQ_VALU_MFMA_BUSY_CYCLES in cycles (guide: per-instruction cycle constants).  Derived ratios are per kernel dispatch means."""
import csv
import glob
import json
import sys
from collections import defaultdict

root, out_path = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(f"{root}/sq*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "pool_collapse_kernel" in k or "pipe_kernel" in k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for k, ctr in acc.items():
    m = {c: sum(v) / len(v) for c, v in ctr.items()}
    d = {"dispatches": max(len(v) for v in ctr.values()), "counters": m}
    wc = m.get("SQ_WAVE_CYCLES")
    if wc:
        for name, key in (("wait_any_frac", "SQ_WAIT_ANY"), ("wait_inst_any_frac", "SQ_WAIT_INST_ANY"), ("active_inst_any_frac", "SQ_ACTIVE_INST_ANY"),
                          ("active_valu_frac", "SQ_ACTIVE_INST_VALU"), ("active_lds_frac", "SQ_ACTIVE_INST_LDS"), ("active_scalar_frac", "SQ_ACTIVE_INST_SCA")):
            if key in m:
                d[name + "_of_wave_cycles"] = m[key] / wc
    if m.get("SQ_LDS_IDX_ACTIVE"):
        d["lds_bank_conflict_frac_of_lds_cycles"] = m.get("SQ_LDS_BANK_CONFLICT", 0.0) / m["SQ_LDS_IDX_ACTIVE"]
    if m.get("SQ_INSTS_MFMA") and m.get("GRBM_GUI_ACTIVE"):
        # every v_mfma_f32_32x32x16_bf16 holds its SIMD's matrix pipe for 32 cycles; 1024 SIMDs; GRBM_GUI_ACTIVE is summed over 8 XCDs
        d["matrix_pipe_busy_frac"] = m["SQ_INSTS_MFMA"] * 32.0 / (1024.0 * m["GRBM_GUI_ACTIVE"] / 8.0)
    res[k] = d
json.dump({"source": "rocprofv3 --pmc passes (tools/refresh_profiles.sh), means per dispatch", "kernels": res}, open(out_path, "w"), indent=1)
print(json.dumps({k[:60]: {x: round(y, 3) for x, y in v.items() if isinstance(y, float)} for k, v in res.items()}, indent=1)[:3000])
