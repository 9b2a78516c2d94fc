#!/bin/bash
# Round-3 profile set, on the GPU box:   bash tools/profile_round3.sh [what]     what: all | bench | kernels | train | trainpmc | c2 | rambo | wide
# Writes gpurun_out/r03_*; copy what should be judged into profiles/.  Counters run in their own passes, without tracing
# (rocprofv3 --pmc only).  `kernels`: ONE config per rocprofv3 --kernel-trace --stats run, so that per-config kernel timings
# (apply / statistics / ingest / tail split) are artefacts: r03_kernels_<cfg>.csv.
set -e
WHAT=${1:-all}
OUT=$PWD/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
pmc() {   # pmc NAME COUNTERS -- command...
  local name=$1 ctr=$2; shift 2
  rocprofv3 --pmc $ctr --output-format csv -d $OUT/r03_pmc_$name -o p -- "$@" > $OUT/r03_pmc_$name.log 2>&1
}
if [ $WHAT = all ] || [ $WHAT = bench ]; then
  python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r03_bench_driver.json 2> $OUT/r03_bench_driver.err
  python3 bench.py --no-cpu-baseline > $OUT/r03_bench_default.json 2> $OUT/r03_bench_default.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r03_kt -o k -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/r03_kt.log 2>&1
  cp $OUT/r03_kt/k_kernel_stats.csv $OUT/r03_bench_driver_kernel_stats.csv
fi
if [ $WHAT = all ] || [ $WHAT = kernels ]; then
  for c in C3 C4 C5; do
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r03_kt_$c -o k -- python3 tools/run_config_step.py $c 6 > $OUT/r03_kt_$c.log 2>&1
    cp $OUT/r03_kt_$c/k_kernel_stats.csv $OUT/r03_kernels_$c.csv
  done
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r03_kt_C2 -o k -- python3 tools/run_c2_set.py 20 4 > $OUT/r03_kt_C2.log 2>&1
  cp $OUT/r03_kt_C2/k_kernel_stats.csv $OUT/r03_kernels_C2.csv
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r03_kt_rambo -o k -- python3 tools/run_rambo.py > $OUT/r03_kt_rambo.log 2>&1
  cp $OUT/r03_kt_rambo/k_kernel_stats.csv $OUT/r03_kernels_rambo.csv
fi
if [ $WHAT = all ] || [ $WHAT = train ]; then
  # training legs of bench.py (one optimizer step each): C2 at mbs 1e6, C5 at 2e5, the reference's default regime (5 x 2000 as grid.y)
  for c in C2 C5 C1_epoch C2_mbs4e6 C2_graph; do
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r03_kt_train_$c -o k -- python3 tools/run_train_leg.py $c 5 > $OUT/r03_kt_train_$c.log 2>&1
    cp $OUT/r03_kt_train_$c/k_kernel_stats.csv $OUT/r03_kernels_train_$c.csv
  done
  python3 tools/train_probe.py C2 1e6 1 20 both 2>&1 | grep "^C" > $OUT/r03_train_probe.txt
  python3 tools/train_probe.py C2 4e6 1 10 batched 2>&1 | grep "^C" >> $OUT/r03_train_probe.txt
  python3 tools/train_probe.py C2 2000 5 50 both 2>&1 | grep "^C" >> $OUT/r03_train_probe.txt
  python3 tools/train_probe.py C5 2e5 1 5 both 2>&1 | grep "^C" >> $OUT/r03_train_probe.txt
  python3 tools/train_probe.py C5 2000 5 20 both 2>&1 | grep "^C" >> $OUT/r03_train_probe.txt
  python3 tools/host_sections.py 2000 5 2>&1 | grep -v "Warn\|amdgpu" >> $OUT/r03_train_probe.txt
  tools/probe/rambo_traffic 1000000 40 > $OUT/r03_rambo_traffic_probe.txt 2>&1 || true
fi
if [ $WHAT = all ] || [ $WHAT = trainpmc ]; then
  # counters of the narrow training kernels at a kernel-bound minibatch (C2 model, one minibatch of 4e6): instructions and HBM bytes
  pmc train_valu "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" python3 tools/train_probe.py C2 4e6 1 3 batched
  pmc train_fetch FETCH_SIZE python3 tools/train_probe.py C2 4e6 1 3 batched
  pmc train_write WRITE_SIZE python3 tools/train_probe.py C2 4e6 1 3 batched
  python3 tools/pmc_json.py $OUT/r03_pmc_train_c2.json 1 sets=6 valu=$OUT/r03_pmc_train_valu/p_counter_collection.csv fetch=$OUT/r03_pmc_train_fetch/p_counter_collection.csv write=$OUT/r03_pmc_train_write/p_counter_collection.csv > $OUT/r03_pmc_train_c2.txt
fi
if [ $WHAT = all ] || [ $WHAT = c2 ]; then
  pmc c2_valu "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" python3 tools/run_c2_set.py 20 4
  pmc c2_fetch FETCH_SIZE python3 tools/run_c2_set.py 20 4
  pmc c2_write WRITE_SIZE python3 tools/run_c2_set.py 20 4
  python3 tools/pmc_json.py $OUT/r03_pmc_c2.json 20 sets=4 valu=$OUT/r03_pmc_c2_valu/p_counter_collection.csv fetch=$OUT/r03_pmc_c2_fetch/p_counter_collection.csv write=$OUT/r03_pmc_c2_write/p_counter_collection.csv > $OUT/r03_pmc_c2.txt
fi
if [ $WHAT = all ] || [ $WHAT = rambo ]; then
  pmc rambo_fetch FETCH_SIZE python3 tools/run_rambo.py
  pmc rambo_write WRITE_SIZE python3 tools/run_rambo.py
  pmc rambo_sq "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS" python3 tools/run_rambo.py
  python3 tools/pmc_json.py $OUT/r03_pmc_rambo.json 1 valu=$OUT/r03_pmc_rambo_sq/p_counter_collection.csv fetch=$OUT/r03_pmc_rambo_fetch/p_counter_collection.csv write=$OUT/r03_pmc_rambo_write/p_counter_collection.csv > $OUT/r03_pmc_rambo.txt
fi
if [ $WHAT = all ] || [ $WHAT = wide ]; then
  for c in C3 C4 C5; do
    pmc wide_${c}_mfma "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_F64 SQ_WAVE_CYCLES SQ_BUSY_CYCLES" python3 tools/run_config_step.py $c 3
    pmc wide_${c}_fetch FETCH_SIZE python3 tools/run_config_step.py $c 3
    pmc wide_${c}_write WRITE_SIZE python3 tools/run_config_step.py $c 3
    python3 tools/pmc_json.py $OUT/r03_pmc_wide_$c.json 1 sets=3 mfma=$OUT/r03_pmc_wide_${c}_mfma/p_counter_collection.csv fetch=$OUT/r03_pmc_wide_${c}_fetch/p_counter_collection.csv write=$OUT/r03_pmc_wide_${c}_write/p_counter_collection.csv > $OUT/r03_pmc_wide_$c.txt
  done
  python3 - <<'PY'
import json
out = {}
for c in ("C3", "C4", "C5"):
    d = json.load(open(f"gpurun_out/r03_pmc_wide_{c}.json"))
    ks = d["kernels"]
    ap = ks.get("k_cell_apply_mfma", {})
    out[c] = {"mfma_util": ap.get("mfma_util"),
              "hbm_bytes_per_step": round(sum(k.get("hbm_bytes", 0) * k.get("launches_per_set", 0) for k in ks.values()), 1),
              "kernels": {k: {a: v[a] for a in ("mfma_util", "hbm_bytes", "launches_per_set", "valu_insts") if a in v} for k, v in ks.items()}}
json.dump(out, open("gpurun_out/r03_pmc_wide.json", "w"), indent=1, sort_keys=True)
print(json.dumps({c: (out[c]["mfma_util"], out[c]["hbm_bytes_per_step"]) for c in out}))
PY
fi
ls $OUT | grep r03_ | head -80
