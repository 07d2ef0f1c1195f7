#!/bin/bash
# Steady-state launch census per step of the three timed loops (run on the GPU box):  bash scripts/profile_steady.sh <outdir> <round tag>
# For each loop two rocprofv3 traces (kernels + memory copies) with 8 and 24 steps and no warm-up; their difference / 16 is one
# steady-state step (scripts/steady_state_stats.py): own kernels, torch / rocBLAS / rocprim kernels, copies and fills per step.
#   inference step   bench.py headline (encode+fuse 4096 drugs, score 4096^2 x 896)
#   finetune step    scripts/train_bench.py (fixed batch; bench.py's fresh-batch step adds the mask / triple plans)
#   contrastive step scripts/pretrain_bench.py (batch 2048)
export TMPDIR=/tmp
out=gpurun_out/$1; tag=$2
mkdir -p $out profiles
run() {   # name, command...
  name=$1; shift
  for n in 8 24; do
    rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $out/$name$n -- "$@" --steps $n --warmup 0 > $out/$name$n.log 2>&1
    cp "$(find $out/$name$n -name '*kernel_stats.csv' | head -n 1)" $out/${name}_stats$n.csv
    cp "$(find $out/$name$n -name '*memory_copy_stats.csv' | head -n 1)" $out/${name}_copies$n.csv 2>/dev/null
  done
  python3 scripts/steady_state_stats.py $out/${name}_stats8.csv 8 $out/${name}_stats24.csv 24 profiles/${tag}_${name}_steady_state_kernel_stats.csv > $out/${name}_steady.txt
  python3 - $out $name $tag <<'PY'
import csv, sys
out, name, tag = sys.argv[1:4]
rows = list(csv.DictReader(open(f"profiles/{tag}_{name}_steady_state_kernel_stats.csv")))
own = [r for r in rows if "(anonymous namespace)" in r["Name"] or r["Name"].startswith("mdg_")]
lib = [r for r in rows if r not in own]
def tot(rs): return sum(float(r["CallsPerStep"]) for r in rs), sum(float(r["DurationNsPerStep"]) for r in rs) / 1e6
cop = {}
try:
    a = {r["Name"]: r for r in csv.DictReader(open(f"{out}/{name}_copies8.csv"))}
    b = {r["Name"]: r for r in csv.DictReader(open(f"{out}/{name}_copies24.csv"))}
    for k, r in b.items():
        cop[k] = (int(r["Calls"]) - int(a.get(k, {"Calls": 0})["Calls"])) / 16.0
except Exception as e:
    cop = {"(no memory-copy stats)": str(e)}
with open(f"profiles/{tag}_{name}_launch_census.txt", "w") as f:
    f.write(f"{name} step, steady state = (24-step trace - 8-step trace) / 16; rocprofv3 --kernel-trace --memory-copy-trace --stats\n")
    f.write("own kernels (libmadrigal_hip.so): %.1f launches, %.2f ms of kernel time per step\n" % tot(own))
    f.write("other kernels (torch elementwise / index / fill, rocBLAS, rocprim): %.1f launches, %.2f ms per step\n" % tot(lib))
    f.write("memory copies per step: " + ", ".join(f"{k} {v:.1f}" if isinstance(v, float) else f"{k} {v}" for k, v in cop.items()) + "\n\nother kernels by time:\n")
    for r in sorted(lib, key=lambda r: -float(r["DurationNsPerStep"]))[:25]:
        f.write(f"  {float(r['CallsPerStep']):7.2f} x {float(r['AverageNs'])/1e3:8.1f} us  {r['Name'][:110]}\n")
print(open(f"profiles/{tag}_{name}_launch_census.txt").read()[:1500])
PY
}
run inference python3 bench.py --no-cpu-baseline --finetune-steps 0 --pretrain-steps 0 --stress-drugs 0 --rank-outcomes 0 --no-f32-exact
run finetune python3 scripts/train_bench.py --precision bf16
run contrastive python3 scripts/pretrain_bench.py
cp profiles/${tag}_* $out/ 2>/dev/null
