#!/bin/bash
# PMC passes over the head kernel at the bench shape (run on the GPU box):  bash scripts/pmc_head.sh <outdir> "<counters>" [more sets...]
# env: PRECS (default bf16x3), VARS (default 0), SYM (default 1)
set -e
out=$1; shift
export TMPDIR=/tmp
mkdir -p gpurun_out/$out
i=0
for set in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d gpurun_out/$out/set$i -- python3 scripts/head_variants.py --precisions ${PRECS:-bf16x3} --variants ${VARS:-0} --sym ${SYM:-1} --reps 2 > gpurun_out/$out/set$i.log 2>&1
  python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/$out/set$i/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "bilinear_allpairs" in r["Kernel_Name"]:
            agg[r["Kernel_Name"][:100]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("gpurun_out/$out/summary.txt", "a") as o:
    for k, d in agg.items():
        for c, v in d.items():
            line = f"{k} | {c} | n={len(v)} avg={sum(v)/len(v):.6g}"
            print(line); o.write(line + "\n")
PY
done
