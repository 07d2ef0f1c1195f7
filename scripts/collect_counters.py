#!/usr/bin/env python3
"""Aggregate the PMC passes of scripts/pmc_kernels.sh (gpurun_out/<dir>/set*/) into profiles/<tag>_kernels_mfma_lds_counters.json:
per kernel, per-launch averages of every counter, the LDS conflict share, the LDS-issue-wait share and the MFMA-busy share of the
SIMD cycles.   python scripts/collect_counters.py <dir> <tag>"""
import collections, csv, glob, json, sys
src, tag = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"gpurun_out/{src}/set*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
keep = ("linear_pp_kernel", "linear_kernel", "bilinear", "fusion_attention", "prep_operands", "msd_")
out = {"command": "rocprofv3 --pmc <one set per pass> --output-format csv -- python3 scripts/kernels_once.py  (scripts/pmc_kernels.sh)",
       "note": "per-launch averages summed over the chip as rocprofv3 reports them; lds_conflict_share = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE; "
               "lds_issue_wait_share = SQ_WAIT_INST_LDS / SQ_BUSY_CYCLES (the r02 verdict's figure); mfma_busy_share_of_simd_cycles = "
               "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs)",
       "kernels": {}}
for k, cs in agg.items():
    if not any(s in k for s in keep):
        continue
    d = {c: sum(v) / len(v) for c, v in cs.items()}
    if d.get("SQ_LDS_IDX_ACTIVE"):
        d["lds_conflict_share"] = d.get("SQ_LDS_BANK_CONFLICT", 0.0) / d["SQ_LDS_IDX_ACTIVE"]
    if d.get("SQ_BUSY_CYCLES") and "SQ_WAIT_INST_LDS" in d:
        d["lds_issue_wait_share"] = d["SQ_WAIT_INST_LDS"] / d["SQ_BUSY_CYCLES"]
    if d.get("GRBM_GUI_ACTIVE") and "SQ_VALU_MFMA_BUSY_CYCLES" in d:
        d["mfma_busy_share_of_simd_cycles"] = d["SQ_VALU_MFMA_BUSY_CYCLES"] / (d["GRBM_GUI_ACTIVE"] / 8 * 1024)
    out["kernels"][k] = d
json.dump(out, open(f"profiles/{tag}_kernels_mfma_lds_counters.json", "w"), indent=1)
for k, d in out["kernels"].items():
    print(k[:90], {c: round(v, 3) for c, v in d.items() if "share" in c})
