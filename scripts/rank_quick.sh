#!/bin/bash
# rank path, quick GPU check: parity tests, per-kernel times, per-outcome time.   bash scripts/rank_quick.sh <tag>
mkdir -p gpurun_out/r5
timeout -k 10 600 python -m pytest tests/test_losses_ranks_gpu.py -m gpu -q -x > gpurun_out/r5/ranks_tests_$1.log 2>&1; tail -2 gpurun_out/r5/ranks_tests_$1.log
bash scripts/rank_prof.sh $1 4096 32 > gpurun_out/r5/prof_$1.log 2>&1; head -5 gpurun_out/r5/prof_$1.log | cut -c1-60,112-
cd $GRAFT_REPO_ROOT
python scripts/rank_bench.py 4096 32 --no-oracle 2>&1 | grep "HIP ("
