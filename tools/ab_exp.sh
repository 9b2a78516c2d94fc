run() { python3 bench.py "$@" --no-cpu-baseline --no-configs 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('flow %.3e rambo %.3e frac %s'%(d['value'], d['rambo']['value'], d['roofline'].get('frac')))"; }
timeout -k 10 700 python3 -m pytest tests -m gpu -q > gpurun_out/gputest_a.log 2>&1; echo rc=$?; tail -3 gpurun_out/gputest_a.log
for v in new exp6 occ1; do
  if [ $v = new ]; then unset NIS_LIB; else export NIS_LIB=$PWD/nf_amd/lib/variants/$v.so; fi
  echo "== $v driver"; run --steps 20 --warmup 5; run --steps 20 --warmup 5
  echo "== $v default"; run
done
unset NIS_LIB
echo "== wide new"; python3 tools/wide_steps_time.py C3 2>&1 | grep "C3"; python3 tools/wide_steps_time.py C5 2>&1 | grep "C5"
export NIS_LIB=$PWD/nf_amd/lib/variants/exp6w.so
echo "== wide exp6"; python3 tools/wide_steps_time.py C3 2>&1 | grep "C3"; python3 tools/wide_steps_time.py C5 2>&1 | grep "C5"
