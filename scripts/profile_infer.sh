#!/bin/bash
# kernel statistics of the inference step only (encode+fuse + head), run on the GPU box: bash scripts/profile_infer.sh <outdir>
set -e
out=$1; export TMPDIR=/tmp; mkdir -p gpurun_out/$out
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$out/kt -- python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --finetune-steps 0 --pretrain-steps 0 --stress-drugs 0 --rank-outcomes 0 > gpurun_out/$out/run.log 2>&1
python3 - <<PY
import csv, glob
rows = list(csv.DictReader(open(glob.glob("gpurun_out/$out/kt/*/*_kernel_stats.csv")[0])))
steps = 10
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"kernel time per step {tot/steps/1e6:.2f} ms, launches per step {sum(int(r['Calls']) for r in rows)/steps:.0f}")
for r in rows[:32]:
    print(f"{r['Name'][:100]:100s} {int(r['Calls'])/steps:7.1f} calls  {float(r['TotalDurationNs'])/steps/1e6:7.3f} ms  avg {float(r['AverageNs'])/1e3:8.1f} us")
PY
