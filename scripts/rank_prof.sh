#!/bin/bash
# per-kernel times of the rank normalisation (rocprofv3 kernel stats of scripts/rank_bench.py): bash scripts/rank_prof.sh <tag> [N] [L]
set -e
TAG=${1:-rank}; N=${2:-4096}; L=${3:-32}
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5/prof_$TAG
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o p -- python3 $GRAFT_REPO_ROOT/scripts/rank_bench.py $N $L --no-oracle > $OUT/run.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:16]:
    print(f"{r['Name'][:110]:110s} calls {r['Calls']:>5s} total_us {float(r['TotalDurationNs'])/1e3:10.1f} avg_us {float(r['AverageNs'])/1e3:9.1f} {r['Percentage']}%")
PY
tail -3 $OUT/run.log
