#!/bin/bash
# rocprofv3 evidence for the headline bench (run on the GPU box):  bash scripts/profile_bench.sh <outdir>
#   kt    = --kernel-trace --stats of `bench.py --steps 5 --warmup 2 --no-cpu-baseline`
#   pmc_w = --pmc WRITE_SIZE, pmc_r = --pmc FETCH_SIZE (separate passes) of the inference part of the same command
set -e
out=$1
export TMPDIR=/tmp
mkdir -p gpurun_out/$out
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$out/kt -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/$out/bench_line_under_rocprof.json 2> gpurun_out/$out/kt.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/$out/pmc_w -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --finetune-steps 0 --pretrain-steps 0 --stress-drugs 0 --rank-outcomes 0 > gpurun_out/$out/pmc_w.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/$out/pmc_r -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --finetune-steps 0 --pretrain-steps 0 --stress-drugs 0 --rank-outcomes 0 > gpurun_out/$out/pmc_r.log 2>&1
python3 bench.py --steps 10 --warmup 3 > gpurun_out/$out/bench_line.json 2> gpurun_out/$out/bench_line.err
echo done
