#!/bin/bash
# every rocprofv3 summary of a round in one call (run on the GPU box):  bash scripts/profile_round.sh <outdir>
# <outdir>/prof = whole-job kernel stats + head PMC traffic + bench line; infer / train / pretrain / ranks = per-step kernel stats
export TMPDIR=/tmp
out=$1
mkdir -p gpurun_out/$out
bash scripts/profile_bench.sh $out/prof > gpurun_out/$out/prof.txt 2>&1
bash scripts/profile_infer.sh $out/infer > gpurun_out/$out/infer.txt 2>&1
bash scripts/profile_train.sh $out/train bf16 4 > gpurun_out/$out/train.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$out/pretrain/kt -- python3 scripts/pretrain_bench.py --steps 5 --warmup 2 > gpurun_out/$out/pretrain.log 2>&1
bash scripts/rank_profile.sh $out/ranks > gpurun_out/$out/ranks.txt 2>&1
python3 bench.py --drugs 4003 --outcomes 901 --steps 5 --warmup 2 --finetune-steps 0 --pretrain-steps 0 --stress-drugs 0 --rank-outcomes 32 > gpurun_out/$out/bench_line_ragged_4003x901.json 2> gpurun_out/$out/ragged.err
echo done
