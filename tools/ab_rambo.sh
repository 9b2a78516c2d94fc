# on-box A/B of the 2 -> 4 map: bash tools/ab_rambo.sh "new rbw1"
timeout -k 10 700 python3 -m pytest tests -m gpu -q -x -k "rambo or Rambo or c4 or C4 or phase" > gpurun_out/gputest_r.log 2>&1; echo rc=$?; tail -3 gpurun_out/gputest_r.log
for v in ${1:-new}; do
  if [ $v = new ]; then unset NIS_LIB; else export NIS_LIB=$PWD/nf_amd/lib/variants/$v.so; fi
  for m in ${2:-1 2}; do
    echo "== $v mult $m"; NIS_RAMBO_GRID_MULT=$m python3 tools/rambo_time.py 2>&1 | tail -1
  done
done
