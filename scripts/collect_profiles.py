#!/usr/bin/env python3
"""Copy the rocprofv3 summaries of a gpurun_out/<dir> (kt = --kernel-trace --stats, pmc_w / pmc_r = WRITE_SIZE /
FETCH_SIZE passes of `bench.py`) into profiles/ under a round tag.   python scripts/collect_profiles.py prof3 r01"""
import collections, csv, glob, json, os, shutil, sys
src, tag = sys.argv[1], sys.argv[2]
root = os.path.join("gpurun_out", src)
os.makedirs("profiles", exist_ok=True)
shutil.copy(glob.glob(f"{root}/kt/*/*_kernel_stats.csv")[0], f"profiles/{tag}_bench_wholejob_bf16x3_kernel_stats.csv")
# (the kernel-trace run is the FULL default bench command: inference headline + finetune legs + cfg5 row statistics)
for f in glob.glob(f"{root}/bench_line*.json"):
    shutil.copy(f, f"profiles/{tag}_{os.path.basename(f)}")
out = {"command": "rocprofv3 --pmc WRITE_SIZE (and, separately, --pmc FETCH_SIZE) --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --finetune-steps 0 --stress-drugs 0",
       "workload": {"drugs": 4096, "outcomes": 896, "precision": "bf16x3"},
       "note": "per-dispatch averages; counters are in KiB; gfx950: FETCH_SIZE tallies 128-B requests at 64 B, so read bytes = 2 x FETCH_SIZE (MI355X_MICROARCH.md, HBM); WRITE_SIZE is exact for these stores.  "
               "FETCH_SIZE / WRITE_SIZE count requests that leave the L2 for the fabric: bytes the 256-MB Infinity Cache serves are INCLUDED, so 'past_l2_tb_per_s' is traffic past the L2, an upper bound on HBM bytes -- "
               "a gather kernel whose table fits that cache (hgt_attention_kernel: k' | v' rows of ~500 MB re-read ~9 x) shows a rate above the 8 TB/s of HBM; that rate is cache-served, not HBM",
       "kernels": {}}
keep = ("bilinear", "linear_kernel", "linear_pp_kernel", "hgt_attention", "fusion_attention", "csr_aggregate", "prep_operands")
kt = {r["Name"]: r for r in csv.DictReader(open(glob.glob(f"{root}/kt/*/*_kernel_stats.csv")[0]))}
for sub, cname in (("pmc_w", "WRITE_SIZE"), ("pmc_r", "FETCH_SIZE")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(glob.glob(f"{root}/{sub}/*/*_counter_collection.csv")[0])):
        agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        if any(s in k for s in keep):
            out["kernels"].setdefault(k, {})[cname + "_KiB_avg"] = sum(v) / len(v)
for k, d in out["kernels"].items():
    w, r = d.get("WRITE_SIZE_KiB_avg", 0) * 1024, d.get("FETCH_SIZE_KiB_avg", 0) * 1024 * 2
    d.update(write_bytes=w, read_bytes_corrected=r, past_l2_bytes_per_launch_corrected=w + r)
    if k in kt:
        avg_ns = float(kt[k]["AverageNs"])
        d.update(kernel_trace_avg_us=avg_ns / 1e3, kernel_trace_calls=int(kt[k]["Calls"]), past_l2_tb_per_s=(w + r) / avg_ns / 1e3)
json.dump(out, open(f"profiles/{tag}_bench_wholejob_bf16x3_pmc_traffic.json", "w"), indent=1)
print(open(f"profiles/{tag}_bench_wholejob_bf16x3_kernel_stats.csv").read()[:1500])
