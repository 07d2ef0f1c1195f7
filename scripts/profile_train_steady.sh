#!/bin/bash
# steady-state kernel statistics of the finetune step (run on the GPU box):  bash scripts/profile_train_steady.sh <outdir> [precision]
# two traces of the same loop (8 and 24 steps, no warm-up): their difference is 16 steady-state steps (scripts/steady_state_stats.py)
export TMPDIR=/tmp
out=gpurun_out/$1; prec=${2:-bf16}
mkdir -p $out
for n in 8 24; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt$n -- python3 scripts/train_bench.py --precision $prec --steps $n --warmup 0 > $out/run$n.log 2>&1
  cp "$(find $out/kt$n -name '*kernel_stats.csv' | head -n 1)" $out/stats$n.csv
done
python3 scripts/steady_state_stats.py $out/stats8.csv 8 $out/stats24.csv 24 $out/steady_state_kernel_stats.csv
