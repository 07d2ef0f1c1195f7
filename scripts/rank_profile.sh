#!/bin/bash
# per-kernel times of the rank normalisation (run on the GPU box): bash scripts/rank_profile.sh <outdir>
export TMPDIR=/tmp
out=gpurun_out/$1
mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- python3 scripts/rank_bench.py > $out/rank_bench.log 2>&1
f=$(find $out/kt -name "*kernel_stats.csv" | head -n 1)
cp "$f" $out/rank_kernel_stats.csv
head -n 14 $out/rank_kernel_stats.csv | cut -c1-200
