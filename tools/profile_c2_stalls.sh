#!/bin/bash
# Where do the C2 kernels wait?  bash tools/profile_c2_stalls.sh  -> gpurun_out/r02_c2stall_*.csv + summary on stdout
set -e
OUT=$PWD/gpurun_out
export TMPDIR=/tmp
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY" \
           "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM SQ_INSTS_BRANCH GRBM_GUI_ACTIVE SQ_INST_LEVEL_LDS"; do
  n=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --output-format csv -d $OUT/r02_c2stall_$n -o p -- python3 tools/run_c2_set.py 20 4 > $OUT/r02_c2stall_$n.log 2>&1 || echo "pass $n failed"
done
python3 - <<PY
import csv, collections, re, glob
def short(n): return re.sub(r'\(.*','',n).replace('void ','')[:64]
ctr=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('$OUT/r02_c2stall_*/p_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        ctr[short(r['Kernel_Name'])][r['Counter_Name']].append((int(r['Grid_Size']), float(r['Counter_Value'])))
for k in sorted(ctr):
    if not k.startswith('k_cell_apply') and not k.startswith('k_bn_stats_fold') and not k.startswith('k_ingest2'): continue
    print(k)
    for c,rows in sorted(ctr[k].items()):
        g=max(x for x,_ in rows); vals=[v for x,v in rows if x==g]
        print('   %-28s %16.0f'%(c,sum(vals)/len(vals)))
PY
