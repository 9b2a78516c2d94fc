"""Summarise a rocprofv3 kernel trace (…_kernel_trace.csv) per step: steps are delimited by k_ingest launches.
usage: trace_steps.py trace.csv [step indices from the end, default -1]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ing = [i for i, r in enumerate(rows) if "k_ingest" in r["Kernel_Name"]] + [len(rows)]
which = [int(a) for a in sys.argv[2:]] or [-1]
for w in which:
    a, b = ing[w - 1], ing[w]
    seg = rows[a:b]
    wall = (int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])) / 1e3
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg) / 1e3
    print(f"step {w}: wall {wall:.1f} us  busy {busy:.1f} us  kernels {len(seg)}")
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in seg:
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "")[:60]
        agg[n][0] += 1
        agg[n][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    for n, (c, t) in sorted(agg.items(), key=lambda x: -x[1][1])[:12]:
        print(f"  {t:9.1f} us {c:4d} x {t / c:8.1f}  {n}")
