for a in 2 3 4 5 6 8 12 16; do for st in 2; do
  echo -n "APPLY_BPC=$a STATS_BPC=$st: "; NIS_APPLY_BPC=$a NIS_STATS_BPC=$st timeout -k 10 120 python bench.py --steps 20 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e9,3), d['ms_per_step'], d['roofline']['avg_launch_us'])"
done; done
for st in 1 2 4 8; do echo -n "STATS_BPC=$st (apply 5): "; NIS_APPLY_BPC=5 NIS_STATS_BPC=$st timeout -k 10 120 python bench.py --steps 20 --warmup 10 --no-cpu-baseline --roofline-kernel stats 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e9,3), d['ms_per_step'], d['roofline']['avg_launch_us'])"; done
