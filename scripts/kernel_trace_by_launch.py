"""Per-launch durations of the kernels whose name contains a pattern, from a rocprofv3 kernel trace (csv), in launch order:
    python scripts/kernel_trace_by_launch.py <trace.csv> hgt_ [--last N]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
pat = sys.argv[2]
last = int(sys.argv[sys.argv.index("--last") + 1]) if "--last" in sys.argv else 60
sel = [r for r in rows if pat in r["Kernel_Name"]][-last:]
for r in sel:
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][:40]
    print(f"{n:40s} grid {int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1):7d} wg  {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:8.1f} us")
