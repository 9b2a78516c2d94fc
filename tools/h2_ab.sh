#!/bin/bash
# A/B of the adjoint's head on the C5 training step (one minibatch of 2e5): two-role head with all four tiles in step (shipped), with two tile
# pairs one slot apart (variant h2pair2: tools/build_variant.sh h2pair2 "-DNIS_WM=32 -DH2_NPAIR=2" flow_bwd_wm flow_bwd_wm32), and the one-role
# head + k_bwd_xty_w (NIS_BWD_HEAD2=0).  Per-kernel times from rocprofv3 --kernel-trace --stats.
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r04
mkdir -p $OUT
run() {  # name, env...
  local name=$1; shift
  env "$@" python3 tools/train_probe.py C5 2e5 1 5 batched 2>&1 | grep "^C5" | sed "s/^/$name: /"
}
run "two-role head, tiles in step  " NIS_X=0
[ -f nf_amd/lib/variants/h2pair2.so ] && run "two-role head, pairs pipelined" NIS_LIB=$PWD/nf_amd/lib/variants/h2pair2.so
run "one-role head + GEMM launch   " NIS_BWD_HEAD2=0
for v in step pair2 old; do
  case $v in
    step) unset NIS_LIB NIS_BWD_HEAD2;;
    pair2) [ -f nf_amd/lib/variants/h2pair2.so ] || continue; export NIS_LIB=$PWD/nf_amd/lib/variants/h2pair2.so;;
    old) unset NIS_LIB; export NIS_BWD_HEAD2=0;;
  esac
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/h2ab_$v -o k -- python3 tools/train_probe.py C5 2e5 1 5 batched > $OUT/h2ab_$v.log 2>&1
  echo "-- $v"
  python3 tools/kstats.py $OUT/h2ab_$v/k_kernel_stats.csv 2>/dev/null | grep -E "head|layer|xty|reduce" | head -6
done
