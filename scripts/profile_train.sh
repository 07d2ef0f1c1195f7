#!/bin/bash
# rocprofv3 kernel statistics of the finetune step (run on the GPU box):  bash scripts/profile_train.sh <outdir> <precision> [steps]
set -e
out=$1; prec=${2:-bf16}; steps=${3:-4}
export TMPDIR=/tmp
mkdir -p gpurun_out/$out
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$out/kt -- python3 scripts/train_bench.py --precision $prec --steps $steps --warmup 0 > gpurun_out/$out/run.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("gpurun_out/$out/kt/*/*_kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
steps = $steps
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"kernel time per step {tot/steps/1e6:.2f} ms, launches per step {sum(int(r['Calls']) for r in rows)/steps:.0f}")
for r in rows[:45]:
    print(f"{r['Name'][:100]:100s} {int(r['Calls'])/steps:7.1f} calls  {float(r['TotalDurationNs'])/steps/1e6:7.3f} ms  avg {float(r['AverageNs'])/1e3:8.1f} us")
PY
grep ms_per_step gpurun_out/$out/run.log | cut -c1-200
