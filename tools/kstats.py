"""Print a rocprofv3 kernel_stats.csv compactly:  python tools/kstats.py <csv> [steps] [rows]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
n = int(sys.argv[3]) if len(sys.argv) > 3 else 25
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total {tot / 1e3:.1f} us, per step {tot / 1e3 / steps:.1f} us")
for r in rows[:n]:
    print(f"{r['Name'][:100]:100s} {r['Calls']:>5s} {float(r['AverageNs']) / 1e3:9.1f} us {float(r['Percentage']):6.2f} %")
