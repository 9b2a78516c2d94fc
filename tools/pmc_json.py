"""rocprofv3 --pmc counter_collection CSVs -> one JSON of per-kernel means for bench.py (profiles/r02_pmc_*.json).

    python3 tools/pmc_json.py OUT.json steps_per_launch NAME=pass_counter_collection.csv [...]

NAME is one of valu (SQ_INSTS_VALU[, SQ_INSTS_VALU_MFMA_F64, ...]), fetch (FETCH_SIZE), write (WRITE_SIZE), mfma
(SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE).  Only the full-size launches of a kernel (largest grid) are averaged.
HBM bytes = 2 x FETCH_SIZE KB + WRITE_SIZE KB (gfx950 tallies 128-byte read requests at 64 B: MI355X_MICROARCH.md, HBM).
MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs)."""
import collections
import csv
import json
import re
import sys


def short(n):
    n = n.replace("(anonymous namespace)::", "")
    n = re.sub(r"\(.*", "", n).replace("void ", "").strip()
    return re.sub(r"<.*", "", n)


def load(path):
    by = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        by[short(r["Kernel_Name"])][r["Counter_Name"]].append((int(r["Grid_Size"]), float(r["Counter_Value"])))
    out = {}
    for k, ctrs in by.items():
        out[k] = {}
        for c, rows in ctrs.items():
            gmax = max(g for g, _ in rows)
            vals = [v for g, v in rows if g == gmax]
            out[k][c] = (sum(vals) / len(vals), len(vals), len(rows))
    return out


out_path, spl = sys.argv[1], int(sys.argv[2])
res = {"steps_per_launch": spl, "kernels": {}}
sets = 1
for arg in sys.argv[3:]:
    name, path = arg.split("=", 1)
    if name == "sets":                                   # launch sets (or iterations) the profiled command ran
        sets = int(path)
        continue
    for k, ctrs in load(path).items():
        if not k.startswith("k_"):
            continue
        rec = res["kernels"].setdefault(k, {})
        for c, (mean, n_full, n_all) in ctrs.items():
            rec[c] = round(mean, 3)
            rec["full_size_launches"] = n_full
res["sets"] = sets
for k, rec in res["kernels"].items():
    rec["launches_per_set"] = round(rec.get("full_size_launches", 0) / sets, 3)
    if "SQ_INSTS_VALU" in rec:
        rec["valu_insts"] = rec["SQ_INSTS_VALU"]
    if "FETCH_SIZE" in rec or "WRITE_SIZE" in rec:
        rec["hbm_bytes"] = round(2.0 * rec.get("FETCH_SIZE", 0.0) * 1024 + rec.get("WRITE_SIZE", 0.0) * 1024, 1)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in rec and rec.get("GRBM_GUI_ACTIVE"):
        rec["mfma_util"] = round(rec["SQ_VALU_MFMA_BUSY_CYCLES"] / (rec["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0), 4)
json.dump(res, open(out_path, "w"), indent=1, sort_keys=True)
for k, rec in sorted(res["kernels"].items()):
    print(f"{k:28s} " + "  ".join(f"{a}={rec[a]}" for a in ("valu_insts", "hbm_bytes", "mfma_util", "launches_per_set") if a in rec))
