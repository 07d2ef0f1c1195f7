"""Per-step kernel statistics of the STEADY STATE from two rocprofv3 --stats runs of the same loop with different step counts:
(stats of the long run - stats of the short run) / (difference of the step counts).  One-time work (plans, first-step allocations,
graph capture) appears in both runs and cancels.

    python scripts/steady_state_stats.py <short_kernel_stats.csv> <n_short> <long_kernel_stats.csv> <n_long> <out.csv>"""
import csv, sys
short, n_short, long_, n_long, out = sys.argv[1], int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), sys.argv[5]
a = {r["Name"]: r for r in csv.DictReader(open(short))}
b = {r["Name"]: r for r in csv.DictReader(open(long_))}
steps = n_long - n_short
rows = []
for name, r in b.items():
    calls = int(r["Calls"]) - int(a.get(name, {"Calls": 0})["Calls"])
    ns = float(r["TotalDurationNs"]) - float(a.get(name, {"TotalDurationNs": 0})["TotalDurationNs"])
    if calls <= 0:
        continue
    rows.append((name, calls / steps, ns / steps, ns / calls))
rows.sort(key=lambda x: -x[2])
tot_calls, tot_ns = sum(r[1] for r in rows), sum(r[2] for r in rows)
with open(out, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "CallsPerStep", "DurationNsPerStep", "AverageNs", "Percentage"])
    for name, c, ns, avg in rows:
        w.writerow([name, f"{c:.3f}", f"{ns:.0f}", f"{avg:.0f}", f"{100 * ns / tot_ns:.2f}"])
print(f"steady state over {steps} steps: {tot_calls:.1f} launches per step, {tot_ns / 1e6:.2f} ms of kernel time per step")
lib = [r for r in rows if "rocprim" in r[0] or r[0].startswith("Cijk")]
print("rocprim / rocBLAS rows:", [(r[0][:60], round(r[1], 2)) for r in lib] or "none")
