#!/bin/bash
# per-kernel times of the rank normalisation on the bench's own score tensor (8 outcomes): bash scripts/rank_real_prof.sh <tag>
TAG=${1:-real}
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5/prof_$TAG
rm -rf $OUT && mkdir -p $OUT
DBG_MODEL=0 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o p -- python3 $GRAFT_REPO_ROOT/scripts/rank_msd_debug.py > $OUT/run.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "msd_" in r["Name"] or "scatter" in r["Name"] or "rank_block" in r["Name"]:
        print(f"{r['Name'][:70]:70s} calls {r['Calls']:>5s} total_us {float(r['TotalDurationNs'])/1e3:10.1f} avg_us {float(r['AverageNs'])/1e3:9.1f}")
PY
