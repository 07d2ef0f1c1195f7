#!/bin/bash
# every rocprofv3 summary of a round in one call (run on the GPU box):  bash scripts/profile_round.sh <outdir>
# <outdir>/prof = whole-job kernel stats + head PMC traffic + bench line; infer / train / pretrain / ranks = per-step kernel stats
export TMPDIR=/tmp
out=$1
mkdir -p gpurun_out/$out
bash scripts/profile_bench.sh $out/prof > gpurun_out/$out/prof.txt 2>&1
bash scripts/profile_infer.sh $out/infer > gpurun_out/$out/infer.txt 2>&1
bash scripts/profile_train.sh $out/train bf16 16 > gpurun_out/$out/train.txt 2>&1     # 16 steps: the first step's one-time plan building is 1/16 of the average
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$out/pretrain/kt -- python3 scripts/pretrain_bench.py --steps 5 --warmup 2 > gpurun_out/$out/pretrain.log 2>&1
bash scripts/rank_pmc.sh $out/ranks ${2:-rXX} 4096 64 > gpurun_out/$out/ranks.txt 2>&1 && cp profiles/${2:-rXX}_rank_normalize_* gpurun_out/$out/    # kernel stats + PMC traffic of every rank kernel
python3 bench.py --drugs 4003 --outcomes 901 --steps 5 --warmup 2 --finetune-steps 0 --pretrain-steps 0 --stress-drugs 0 --rank-outcomes 32 > gpurun_out/$out/bench_line_ragged_4003x901.json 2> gpurun_out/$out/ragged.err
echo done
