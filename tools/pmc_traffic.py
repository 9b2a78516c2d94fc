"""HBM traffic per launch from two rocprofv3 PMC passes of bench.py (FETCH_SIZE, WRITE_SIZE: they do not fit one pass).

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -o f -- python3 bench.py --steps 10 --warmup 10 --no-cpu-baseline
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -o w -- python3 bench.py --steps 10 --warmup 10 --no-cpu-baseline
    python3 tools/pmc_traffic.py gpurun_out/pmc_fetch/f_counter_collection.csv gpurun_out/pmc_write/w_counter_collection.csv OUT.json OUT.csv

Counter values are KB.  gfx950: FETCH_SIZE tallies 128-byte requests at 64 B (MI355X_MICROARCH.md, "HBM"), so it is doubled;
the calibration rows below (kernels whose byte counts are known exactly) check both corrections.  Only the full-size launches
(the largest grid of each kernel name) are averaged."""
import collections
import csv
import json
import re
import sys


def load(path):
    by = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").strip()
        by[name].append((int(r["Grid_Size"]), float(r["Counter_Value"])))
    out = {}
    for name, rows in by.items():
        gmax = max(g for g, _ in rows)
        vals = [v for g, v in rows if g == gmax]
        out[name] = (gmax, len(vals), sum(vals) / len(vals))
    return out


fetch, write = load(sys.argv[1]), load(sys.argv[2])
rows, js = [], {}
for name in sorted(set(fetch) | set(write)):
    if not name.startswith("k_"):
        continue
    g, nf, f = fetch.get(name, (0, 0, 0.0))
    _, nw, w = write.get(name, (0, 0, 0.0))
    total = 2.0 * f * 1024 + w * 1024
    rows.append((name, g, nf, f, w, total))
    js[name] = {"grid_threads": g, "launches": nf, "FETCH_SIZE_KB": round(f, 3), "WRITE_SIZE_KB": round(w, 3),
                "bytes_per_launch": round(total, 1)}
with open(sys.argv[4], "w") as fh:
    fh.write("kernel,grid_threads,launches,FETCH_SIZE_KB_avg,WRITE_SIZE_KB_avg,hbm_bytes_per_launch(2*fetch+write)\n")
    for r in rows:
        fh.write(f"\"{r[0]}\",{r[1]},{r[2]},{r[3]:.3f},{r[4]:.3f},{r[5]:.0f}\n")
json.dump(js, open(sys.argv[3], "w"), indent=1)
for r in rows:
    print(f"{r[0][:48]:48s} grid {r[1]:9d} x{r[2]:3d}  fetch {r[3]:12.1f} KB  write {r[4]:12.1f} KB  -> {r[5] / 1e6:9.2f} MB")
