#!/bin/bash
# The profile set of one round, on the GPU box:   bash tools/profile_round.sh r01_v13
# Writes gpurun_out/<tag>_*: bench line, rocprofv3 kernel stats, per-launch list of the roofline kernels, PMC HBM traffic.
# (Copy what should be judged into profiles/.)  Counters run in their own passes, without tracing.
set -e
TAG=${1:-round}
OUT=$PWD/gpurun_out
mkdir -p $OUT
python3 bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_kt -o k -- python3 bench.py --no-cpu-baseline > $OUT/${TAG}_kt.log 2>&1
cp $OUT/${TAG}_kt/k_kernel_stats.csv $OUT/${TAG}_bench_kernel_stats.csv
python3 - $OUT/${TAG}_kt/k_kernel_trace.csv $OUT/${TAG}_roofline_kernel_launches.csv <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
with open(sys.argv[2], "w") as f:
    f.write("# rocprofv3 --kernel-trace of `python3 bench.py --no-cpu-baseline`: every launch of the flow kernels (apply = roofline kernel).\n")
    f.write("kernel,grid_x_wg,grid_y,duration_us\n")
    for r in rows:
        n = r["Kernel_Name"]
        if not re.search(r"k_cell_apply|k_bn_stats|k_ingest", n):
            continue
        name = re.sub(r"\(.*", "", n).replace("void ", "")
        wg = int(r["Workgroup_Size_X"]) if "Workgroup_Size_X" in r else 256
        gx = int(r["Grid_Size_X"]) // max(wg, 1) if "Grid_Size_X" in r else 0
        gy = int(r.get("Grid_Size_Y", 1))
        f.write('"%s",%d,%d,%.1f\n' % (name, gx, gy, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
PY
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_pmc_fetch -o f -- python3 bench.py --steps 10 --warmup 10 --no-cpu-baseline > $OUT/${TAG}_pmc.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_pmc_write -o w -- python3 bench.py --steps 10 --warmup 10 --no-cpu-baseline >> $OUT/${TAG}_pmc.log 2>&1
python3 tools/pmc_traffic.py $OUT/${TAG}_pmc_fetch/f_counter_collection.csv $OUT/${TAG}_pmc_write/w_counter_collection.csv $OUT/${TAG}_hbm_traffic.json $OUT/${TAG}_pmc_hbm_traffic.csv
cat $OUT/${TAG}_bench.json
