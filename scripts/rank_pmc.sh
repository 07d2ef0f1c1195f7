#!/bin/bash
# HBM traffic of every rank-normalisation kernel (run on the GPU box): bash scripts/rank_pmc.sh <outdir> <round tag> [N] [L]
# Three separate rocprofv3 passes of scripts/rank_bench.py: --kernel-trace --stats, --pmc WRITE_SIZE, --pmc FETCH_SIZE (the guide's HBM
# recipe: separate passes, counters in KiB, read bytes = 2 x FETCH_SIZE on gfx950); summary -> profiles/<tag>_rank_normalize_*.{csv,json}
export TMPDIR=/tmp
out=gpurun_out/$1; tag=$2; N=${3:-4096}; L=${4:-64}
mkdir -p $out profiles
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- python3 scripts/rank_bench.py $N $L --no-oracle > $out/kt.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_w -- python3 scripts/rank_bench.py $N $L --no-oracle > $out/pmc_w.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_r -- python3 scripts/rank_bench.py $N $L --no-oracle > $out/pmc_r.log 2>&1
python3 - $out $tag $N $L <<'PY'
import collections, csv, glob, hashlib, json, os, re, shutil, sys
def code_sha16(path):
    src = open(path, encoding="utf-8").read()
    code = "\n".join(l for l in (re.sub(r"\s*//.*$", "", ln).rstrip() for ln in src.splitlines()) if l.strip())
    return hashlib.sha256(code.encode()).hexdigest()[:16]
root, tag, N, L = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
ktf = glob.glob(f"{root}/kt/*/*_kernel_stats.csv")[0]
name = os.environ.get("RANK_PROFILE_NAME", "rank_normalize")
shutil.copy(ktf, f"profiles/{tag}_{name}_kernel_stats.csv")
kt = {r["Name"]: r for r in csv.DictReader(open(ktf))}
keep = ("msd_", "rank_block", "scatter_kernel", "extract_keys", "histogram_kernel", "scan_kernel", "rank_blocks", "zero_diag", "fillBuffer")
calls_per_run = 4                                   # rank_bench.py: one warm-up call on 4 outcomes + three timed calls on L outcomes
out = {"command": f"rocprofv3 --pmc WRITE_SIZE (and, separately, --pmc FETCH_SIZE) --output-format csv -- python3 scripts/rank_bench.py {N} {L} --no-oracle",
       "workload": {"drugs": N, "outcomes_per_call": L, "what": "ops.rank_normalize on randn scores, MDG_RANKS_MSD=" + os.environ.get("MDG_RANKS_MSD", "1 (default: the MSD path)"),
                    "msd_path": os.environ.get("MDG_RANKS_MSD", "1") != "0", "ranks_hip_sha16": code_sha16("madrigal_amd/csrc/ranks.hip")},      # the code only: // comments and blank lines do not count (bench.py)
       "note": "per-dispatch averages over the run (one warm-up call on 4 outcomes + three calls on L outcomes); counters in KiB; gfx950: read bytes = 2 x FETCH_SIZE "
               "(MI355X_MICROARCH.md, HBM); FETCH / WRITE count fabric requests: bytes served by the Infinity Cache are included, so this is traffic past the L2, "
               "an upper bound on HBM bytes", "kernels": {}}
tot_w = tot_r = 0.0
for sub, cname in (("pmc_w", "WRITE_SIZE"), ("pmc_r", "FETCH_SIZE")):
    agg = collections.defaultdict(list)
    for f in glob.glob(f"{root}/{sub}/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        if any(s in k for s in keep):
            d = out["kernels"].setdefault(k, {})
            d[cname + "_KiB_avg"] = sum(v) / len(v)
            d[cname + "_KiB_total"] = sum(v)
            d["dispatches"] = len(v)
for k, d in out["kernels"].items():
    w, r = d.get("WRITE_SIZE_KiB_avg", 0) * 1024, d.get("FETCH_SIZE_KiB_avg", 0) * 1024 * 2
    d.update(write_bytes=w, read_bytes_corrected=r, bytes_per_launch_corrected=w + r)
    tot_w += d.get("WRITE_SIZE_KiB_total", 0) * 1024
    tot_r += d.get("FETCH_SIZE_KiB_total", 0) * 1024 * 2
    if k in kt:
        avg_ns = float(kt[k]["AverageNs"])
        d.update(kernel_trace_avg_us=avg_ns / 1e3, kernel_trace_calls=int(kt[k]["Calls"]), tb_per_s_past_l2=(w + r) / avg_ns / 1e3)
outcomes = 4 + 3 * L
M = N * (N - 1) // 2
out["hbm_bytes_per_outcome_corrected"] = (tot_w + tot_r) / outcomes
out["write_bytes_per_outcome"] = tot_w / outcomes
out["read_bytes_per_outcome_corrected"] = tot_r / outcomes
out["algorithmic_bytes_per_outcome"] = M * 4.0 + N * N * 4.0
out["traffic_over_algorithmic"] = out["hbm_bytes_per_outcome_corrected"] / out["algorithmic_bytes_per_outcome"]
json.dump(out, open(f"profiles/{tag}_{name}_pmc_traffic.json", "w"), indent=1)
print(json.dumps({k: out[k] for k in ("hbm_bytes_per_outcome_corrected", "write_bytes_per_outcome", "read_bytes_per_outcome_corrected", "algorithmic_bytes_per_outcome", "traffic_over_algorithmic")}))
for k, d in sorted(out["kernels"].items(), key=lambda kv: -kv[1].get("kernel_trace_avg_us", 0) * kv[1].get("kernel_trace_calls", 0)):
    print(f"{k[:60]:60s} avg {d.get('kernel_trace_avg_us', 0):8.1f} us x {d.get('kernel_trace_calls', 0):4d}  write {d['write_bytes'] / 1e6:8.1f} MB  read {d['read_bytes_corrected'] / 1e6:8.1f} MB  {d.get('tb_per_s_past_l2', 0):5.2f} TB/s")
PY
