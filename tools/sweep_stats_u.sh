# On the GPU box: rebuild the narrow-width unit with NIS_STATS_U in {4, 8, 12, 16} and run the bench (statistics grid 1 and 2 workgroups/CU).
set -e
python -m nf_amd.build > /dev/null
cd nf_amd/csrc
for U in 4 8 12 16; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -DNIS_WM=4 -DNIS_STATS_U=$U -I../../include -c flow_wm.hip -o build/flow_wm4.o 2>/dev/null
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../lib/libnis_hip.so build/*.o
  for st in 1 2; do
    echo -n "U=$U STATS_BPC=$st: "
    (cd ../.. && NIS_STATS_BPC=$st timeout -k 10 120 python bench.py --steps 200 --warmup 50 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e9,3), d['ms_per_step'])")
  done
done
