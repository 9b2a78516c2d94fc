"""Per-kernel means of rocprofv3 --pmc counter collections.

    python3 tools/pmc_sq.py OUT.csv pass1_counter_collection.csv [pass2_counter_collection.csv ...]

Every input is the `*_counter_collection.csv` of one `rocprofv3 --pmc ... --output-format csv` pass; launches smaller than
1 % of a kernel's largest grid (warm-up calls) are left out.  Output: one row per (kernel, counter) with calls and mean.
"""
import csv
import sys
from collections import defaultdict


def main():
    out, ins = sys.argv[1], sys.argv[2:]
    acc = defaultdict(list)
    for path in ins:
        with open(path) as f:
            for r in csv.DictReader(f):
                name = r["Kernel_Name"].split("(")[0]
                acc[(name, r["Counter_Name"])].append((int(r["Grid_Size"]), float(r["Counter_Value"])))
    rows = []
    for (name, ctr), v in sorted(acc.items()):
        gmax = max(g for g, _ in v)
        keep = [x for g, x in v if g * 100 >= gmax]
        rows.append((name, ctr, len(keep), sum(keep) / len(keep)))
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "counter", "launches", "mean_per_launch"])
        w.writerows(rows)
    for r in rows:
        if r[0].startswith("void k_") or r[0].startswith("k_"):
            print("%-60s %-28s %4d %16.1f" % (r[0][:60], r[1], r[2], r[3]))


if __name__ == "__main__":
    main()
