#!/bin/bash
# HBM traffic per kernel of the finetune step (run on the GPU box):  bash scripts/pmc_train.sh <outdir> <kernel-trace dir of profile_train.sh>
#   separate --pmc passes (WRITE_SIZE, FETCH_SIZE) as the guide prescribes; durations come from the kernel trace of profile_train.sh
set -e
out=$1; kt=$2
export TMPDIR=/tmp
mkdir -p gpurun_out/$out
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/$out/pmc_w -- python3 scripts/train_bench.py --precision bf16 --steps 2 --warmup 1 > gpurun_out/$out/pmc_w.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/$out/pmc_r -- python3 scripts/train_bench.py --precision bf16 --steps 2 --warmup 1 > gpurun_out/$out/pmc_r.log 2>&1
python3 - <<PY
import collections, csv, glob, json
root = "gpurun_out/$out"
kt = {r["Name"]: r for r in csv.DictReader(open(glob.glob("gpurun_out/$kt/kt/*/*_kernel_stats.csv")[0]))}
out = {"command": "rocprofv3 --pmc WRITE_SIZE (and, separately, --pmc FETCH_SIZE) --output-format csv -- python3 scripts/train_bench.py --precision bf16 --steps 2 --warmup 1",
       "note": "per-launch averages; counters in KiB; gfx950: read bytes = 2 x FETCH_SIZE (MI355X_MICROARCH.md, HBM); durations: kernel trace of scripts/profile_train.sh.  FETCH_SIZE / WRITE_SIZE count requests past the L2 (Infinity-Cache hits included): past_l2_* is an upper bound on HBM bytes, and a gather kernel whose table fits that cache shows a rate above 8 TB/s that is cache-served",
       "kernels": {}}
for sub, cname in (("pmc_w", "WRITE_SIZE"), ("pmc_r", "FETCH_SIZE")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(glob.glob(f"{root}/{sub}/*/*_counter_collection.csv")[0])):
        agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        out["kernels"].setdefault(k, {})[cname + "_KiB_avg"] = sum(v) / len(v)
        out["kernels"][k]["launches_seen"] = len(v)
rows = []
for k, d in out["kernels"].items():
    w, r = d.get("WRITE_SIZE_KiB_avg", 0) * 1024, d.get("FETCH_SIZE_KiB_avg", 0) * 1024 * 2
    d.update(write_bytes=w, read_bytes_corrected=r, past_l2_bytes_per_launch_corrected=w + r)
    if k in kt:
        avg_ns = float(kt[k]["AverageNs"])
        d.update(kernel_trace_avg_us=avg_ns / 1e3, past_l2_tb_per_s=(w + r) / avg_ns / 1e3)
        rows.append((float(kt[k]["TotalDurationNs"]), k, d))
json.dump(out, open(f"{root}/finetune_step_pmc_traffic.json", "w"), indent=1)
for _, k, d in sorted(rows, reverse=True)[:24]:
    print(f"{k[:70]:70s} read {d['read_bytes_corrected']/1e6:9.1f} MB write {d['write_bytes']/1e6:9.1f} MB  {d['kernel_trace_avg_us']:9.1f} us  {d['past_l2_tb_per_s']:5.2f} TB/s")
PY
