#!/bin/bash
# A/B of library variants on the C2 headline step:  tools/c2_ab.sh nf_amd/lib/variants/A.so [B.so ...]
for v in base "$@"; do
  if [ $v = base ]; then unset NIS_LIB; else export NIS_LIB=$PWD/$v; fi
  for r in 1 2; do python bench.py --gpus 1 --steps 20 --warmup 5 --no-configs --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v driver', '%.4e' % d['value'], d['roofline']['avg_launch_us'])"; done
  python bench.py --no-configs --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v default', '%.4e' % d['value'], d['roofline']['avg_launch_us'])"
done
