#!/bin/bash
# per-kernel times of the rank normalisation (run on the GPU box): bash scripts/rank_profile.sh <outdir> [N] [L]
export TMPDIR=/tmp
out=gpurun_out/$1
mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- python3 scripts/rank_bench.py ${2:-4096} ${3:-64} --no-oracle > $out/rank_bench.log 2>&1
f=$(find $out/kt -name "*kernel_stats.csv" | head -n 1)
cp "$f" $out/rank_kernel_stats.csv
tail -n 2 $out/rank_bench.log
head -n 16 $out/rank_kernel_stats.csv | cut -c1-150 | awk -F, '{print $1, $2, $3, $4}' | sed -e 's/(anonymous namespace):://g' | cut -c1-170
