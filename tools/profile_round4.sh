#!/bin/bash
# Round-4 profile set, on the GPU box:   bash tools/profile_round4.sh [what]     what: all | bench | kernels | train | trainpmc | trainpmc5 | c2 | rambo | wide
# Writes gpurun_out/r04_*; copy what should be judged into profiles/.  Counters run in their own passes, without tracing
# (rocprofv3 --pmc only).  `kernels`: ONE config per rocprofv3 --kernel-trace --stats run, so that per-config kernel timings
# (apply / statistics / ingest / tail split) are artefacts: r04_kernels_<cfg>.csv.
set -e
WHAT=${1:-all}
OUT=$PWD/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
pmc() {   # pmc NAME COUNTERS -- command...
  local name=$1 ctr=$2; shift 2
  rocprofv3 --pmc $ctr --output-format csv -d $OUT/r04_pmc_$name -o p -- "$@" > $OUT/r04_pmc_$name.log 2>&1
}
if [ $WHAT = all ] || [ $WHAT = bench ]; then
  python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r04_bench_driver.json 2> $OUT/r04_bench_driver.err
  python3 bench.py --no-cpu-baseline > $OUT/r04_bench_default.json 2> $OUT/r04_bench_default.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r04_kt -o k -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/r04_kt.log 2>&1
  cp $OUT/r04_kt/k_kernel_stats.csv $OUT/r04_bench_driver_kernel_stats.csv
fi
if [ $WHAT = all ] || [ $WHAT = kernels ]; then
  for c in C3 C4 C5; do
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r04_kt_$c -o k -- python3 tools/run_config_step.py $c 6 > $OUT/r04_kt_$c.log 2>&1
    cp $OUT/r04_kt_$c/k_kernel_stats.csv $OUT/r04_kernels_$c.csv
  done
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r04_kt_C2 -o k -- python3 tools/run_c2_set.py 20 4 > $OUT/r04_kt_C2.log 2>&1
  cp $OUT/r04_kt_C2/k_kernel_stats.csv $OUT/r04_kernels_C2.csv
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r04_kt_rambo -o k -- python3 tools/run_rambo.py > $OUT/r04_kt_rambo.log 2>&1
  cp $OUT/r04_kt_rambo/k_kernel_stats.csv $OUT/r04_kernels_rambo.csv
fi
if [ $WHAT = all ] || [ $WHAT = train ]; then
  # training legs of bench.py (one optimizer step each): C2 at mbs 1e6, C5 at 2e5, the reference's default regime (5 x 2000 as grid.y)
  for c in C2 C5 C1_epoch C2_mbs4e6 C2_graph; do
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r04_kt_train_$c -o k -- python3 tools/run_train_leg.py $c 5 > $OUT/r04_kt_train_$c.log 2>&1
    cp $OUT/r04_kt_train_$c/k_kernel_stats.csv $OUT/r04_kernels_train_$c.csv
  done
  python3 tools/train_probe.py C2 1e6 1 20 both 2>&1 | grep "^C" > $OUT/r04_train_probe.txt
  python3 tools/train_probe.py C2 4e6 1 10 batched 2>&1 | grep "^C" >> $OUT/r04_train_probe.txt
  python3 tools/train_probe.py C2 2000 5 50 both 2>&1 | grep "^C" >> $OUT/r04_train_probe.txt
  python3 tools/train_probe.py C5 2e5 1 5 both 2>&1 | grep "^C" >> $OUT/r04_train_probe.txt
  python3 tools/train_probe.py C5 2000 5 20 both 2>&1 | grep "^C" >> $OUT/r04_train_probe.txt
  # (launches of one tile per wave: the step's time moves by +-8 % from run to run on one box -- three more runs of 150 steps)
  for i in 1 2 3; do python3 tools/train_probe.py C5 2000 5 150 batched 2>&1 | grep "^C" >> $OUT/r04_train_probe.txt; done
  # the same steps as two hipGraph replays (flow.TrainPlan, opt-in NIS_TRAIN_GRAPH=1): what the kernel boundaries of the eager chain cost
  for i in 1 2 3; do NIS_TRAIN_GRAPH=1 python3 tools/train_probe.py C5 2000 5 150 batched 2>&1 | grep "^C" | sed 's/batched /captured/' >> $OUT/r04_train_probe.txt; done
  NIS_TRAIN_GRAPH=1 python3 tools/train_probe.py C5 2e5 1 8 batched 2>&1 | grep "^C" | sed 's/batched /captured/' >> $OUT/r04_train_probe.txt
  python3 tools/host_sections.py 2000 5 2>&1 | grep -v "Warn\|amdgpu" >> $OUT/r04_train_probe.txt
  tools/probe/rambo_traffic 1000000 40 > $OUT/r04_rambo_traffic_probe.txt 2>&1 || true
fi
if [ $WHAT = all ] || [ $WHAT = trainpmc ]; then
  # counters of the narrow training kernels at a kernel-bound minibatch (C2 model, one minibatch of 4e6): instructions and HBM bytes
  pmc train_valu "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" python3 tools/train_probe.py C2 4e6 1 3 batched
  pmc train_fetch FETCH_SIZE python3 tools/train_probe.py C2 4e6 1 3 batched
  pmc train_write WRITE_SIZE python3 tools/train_probe.py C2 4e6 1 3 batched
  python3 tools/pmc_json.py $OUT/r04_pmc_train_c2.json 1 sets=6 valu=$OUT/r04_pmc_train_valu/p_counter_collection.csv fetch=$OUT/r04_pmc_train_fetch/p_counter_collection.csv write=$OUT/r04_pmc_train_write/p_counter_collection.csv > $OUT/r04_pmc_train_c2.txt
fi
if [ $WHAT = all ] || [ $WHAT = trainpmc5 ]; then
  # counters of the matrix-core training kernels (C5 topology, one minibatch of 2e5, 3 optimizer steps after warm-up): MFMA busy cycles,
  # instruction counts, waits, HBM bytes -- per kernel; what `attainable` of the adjoint's head is computed from
  pmc train5_mfma "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_F64 SQ_WAVE_CYCLES SQ_BUSY_CYCLES" python3 tools/train_probe.py C5 2e5 1 3 batched
  pmc train5_wait "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_WAVE_CYCLES" python3 tools/train_probe.py C5 2e5 1 3 batched
  pmc train5_fetch FETCH_SIZE python3 tools/train_probe.py C5 2e5 1 3 batched
  pmc train5_write WRITE_SIZE python3 tools/train_probe.py C5 2e5 1 3 batched
  python3 tools/pmc_json.py $OUT/r04_pmc_train_c5.json 1 sets=6 mfma=$OUT/r04_pmc_train5_mfma/p_counter_collection.csv coex=$OUT/r04_pmc_train5_wait/p_counter_collection.csv fetch=$OUT/r04_pmc_train5_fetch/p_counter_collection.csv write=$OUT/r04_pmc_train5_write/p_counter_collection.csv > $OUT/r04_pmc_train_c5.txt
fi
if [ $WHAT = all ] || [ $WHAT = c2 ]; then
  pmc c2_valu "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" python3 tools/run_c2_set.py 20 4
  pmc c2_fetch FETCH_SIZE python3 tools/run_c2_set.py 20 4
  pmc c2_write WRITE_SIZE python3 tools/run_c2_set.py 20 4
  python3 tools/pmc_json.py $OUT/r04_pmc_c2.json 20 sets=4 valu=$OUT/r04_pmc_c2_valu/p_counter_collection.csv fetch=$OUT/r04_pmc_c2_fetch/p_counter_collection.csv write=$OUT/r04_pmc_c2_write/p_counter_collection.csv > $OUT/r04_pmc_c2.txt
fi
if [ $WHAT = all ] || [ $WHAT = rambo ]; then
  pmc rambo_fetch FETCH_SIZE python3 tools/run_rambo.py
  pmc rambo_write WRITE_SIZE python3 tools/run_rambo.py
  pmc rambo_sq "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS" python3 tools/run_rambo.py
  python3 tools/pmc_json.py $OUT/r04_pmc_rambo.json 1 valu=$OUT/r04_pmc_rambo_sq/p_counter_collection.csv fetch=$OUT/r04_pmc_rambo_fetch/p_counter_collection.csv write=$OUT/r04_pmc_rambo_write/p_counter_collection.csv > $OUT/r04_pmc_rambo.txt
fi
if [ $WHAT = all ] || [ $WHAT = wide ]; then
  for c in C3 C4 C5; do
    pmc wide_${c}_mfma "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_F64 SQ_WAVE_CYCLES SQ_BUSY_CYCLES" python3 tools/run_config_step.py $c 3
    pmc wide_${c}_fetch FETCH_SIZE python3 tools/run_config_step.py $c 3
    pmc wide_${c}_write WRITE_SIZE python3 tools/run_config_step.py $c 3
    # co-execution / wait counters asked for in VERDICT r3 item 2 (their own pass; a counter this build of rocprofv3 does not know fails the pass, not the script)
    pmc wide_${c}_coex "SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" python3 tools/run_config_step.py $c 3 || echo "coex pass failed for $c (see $OUT/r04_pmc_wide_${c}_coex.log)"
    COEX=""; [ -f $OUT/r04_pmc_wide_${c}_coex/p_counter_collection.csv ] && COEX="coex=$OUT/r04_pmc_wide_${c}_coex/p_counter_collection.csv"
    python3 tools/pmc_json.py $OUT/r04_pmc_wide_$c.json 1 sets=3 mfma=$OUT/r04_pmc_wide_${c}_mfma/p_counter_collection.csv fetch=$OUT/r04_pmc_wide_${c}_fetch/p_counter_collection.csv write=$OUT/r04_pmc_wide_${c}_write/p_counter_collection.csv $COEX > $OUT/r04_pmc_wide_$c.txt
  done
  python3 - <<'PY'
import json
out = {}
for c in ("C3", "C4", "C5"):
    d = json.load(open(f"gpurun_out/r04_pmc_wide_{c}.json"))
    ks = d["kernels"]
    ap = ks.get("k_cell_apply_mfma", {})
    per_launch = {"C3": 3 * 4_000_000, "C4": 3 * 500_000, "C5": 3 * 1_000_000}[c]      # run_config_step.py c 3: ONE launch set of three iterations
    keep = ("mfma_util", "hbm_bytes", "launches_per_set", "valu_insts", "mfma_insts", "SQ_VALU_MFMA_COEXEC_CYCLES", "SQ_INST_CYCLES_VMEM", "SQ_WAIT_INST_LDS",
            "SQ_INSTS_LDS", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE")
    for v in ks.values():
        if "SQ_INSTS_VALU_MFMA_F64" in v:
            v["mfma_insts"] = v["SQ_INSTS_VALU_MFMA_F64"]
    out[c] = {"mfma_util": ap.get("mfma_util"),
              "hbm_bytes_per_step": round(sum(k.get("hbm_bytes", 0) * k.get("launches_per_set", 0) for k in ks.values()), 1),
              "kernels": {k: dict({a: v[a] for a in keep if a in v}, samples_per_launch=per_launch) for k, v in ks.items()}}
    if ap.get("mfma_insts") and ap.get("GRBM_GUI_ACTIVE"):
        simd_cycles = ap["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0
        out[c]["apply_t_mfma_frac"] = round(64.0 * ap["mfma_insts"] / simd_cycles, 4)
        out[c]["apply_t_valu_frac"] = round(4.0 * (ap["valu_insts"] - ap["mfma_insts"]) / simd_cycles, 4)
json.dump(out, open("gpurun_out/r04_pmc_wide.json", "w"), indent=1, sort_keys=True)
print(json.dumps({c: (out[c]["mfma_util"], out[c]["hbm_bytes_per_step"]) for c in out}))
PY
fi
ls $OUT | grep r04_ | head -80
