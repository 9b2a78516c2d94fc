#!/bin/bash
# Per-phase cost of the two-role head: average k_bwd_head2_w launch time of the full kernel and of timing-only builds that skip one phase
# (tools/build_variant.sh h2skip_X "-DNIS_WM=32 -DH2_SKIP_X" flow_bwd_wm flow_bwd_wm32; results of those builds are wrong).
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r04
mkdir -p $OUT
for v in full P2 P3 DA DW; do
  if [ $v = full ]; then unset NIS_LIB; else export NIS_LIB=$PWD/nf_amd/lib/variants/h2skip_$v.so; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/h2_$v -o k -- python3 tools/train_probe.py C5 2e5 1 5 batched > $OUT/h2_$v.log 2>&1
  echo "$v: $(grep '^C5' $OUT/h2_$v.log | tail -1)"
  python3 tools/kstats.py $OUT/h2_$v/k_kernel_stats.csv 2>/dev/null | grep -E "head|layer|xty|reduce" | head -6
done
