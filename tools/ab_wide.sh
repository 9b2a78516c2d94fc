# on-box A/B of the matrix-core forward: bash tools/ab_wide.sh "new nopf" "C3 C5"
timeout -k 10 700 python3 -m pytest tests -m gpu -q -x -k "wide or mfma or matrix or fused or C3 or C4 or C5 or c3 or c5 or configs" > gpurun_out/gputest_w.log 2>&1; echo rc=$?; tail -3 gpurun_out/gputest_w.log
for v in ${1:-new}; do
  if [ $v = new ]; then unset NIS_LIB; else export NIS_LIB=$PWD/nf_amd/lib/variants/$v.so; fi
  for c in ${2:-C3 C5}; do
    echo "== $v $c"; python3 tools/wide_steps_time.py $c 2>&1 | grep "G=3\|plain"
  done
done
