# on-box A/B of library variants (tools/build_variant.sh): bash tools/ab_exp.sh "new wd0 wd4"
run() { python3 bench.py "$@" --no-cpu-baseline --no-configs 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('flow %.3e rambo %.3e (%.1f us) frac %s'%(d['value'], d['rambo']['value'], 1e12/d['rambo']['value'], d['roofline'].get('frac')))"; }
timeout -k 10 700 python3 -m pytest tests -m gpu -q -x > gpurun_out/gputest_a.log 2>&1; echo rc=$?; tail -3 gpurun_out/gputest_a.log
for v in ${1:-new}; do
  if [ $v = new ]; then unset NIS_LIB; else export NIS_LIB=$PWD/nf_amd/lib/variants/$v.so; fi
  echo "== $v driver"; run --steps 20 --warmup 5; run --steps 20 --warmup 5
  echo "== $v default"; run
done
