#!/bin/bash
# Where does the matrix-core forward wait?  bash tools/profile_wide_stalls.sh C5   -> gpurun_out/r02_stall_<cfg>_*.csv
set -e
C=${1:-C5}
OUT=$PWD/gpurun_out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/r02_stall_${C}_kt -o k -- python3 tools/run_config_step.py $C 3 > $OUT/r02_stall_${C}_kt.log 2>&1
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY" \
           "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE"; do
  n=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --output-format csv -d $OUT/r02_stall_${C}_$n -o p -- python3 tools/run_config_step.py $C 3 > $OUT/r02_stall_${C}_$n.log 2>&1 || echo "pass $n failed"
done
python3 - <<PY
import csv, collections, re, glob
def short(n): return re.sub(r'<.*','',re.sub(r'\(.*','',n).replace('void ',''))
rows=list(csv.DictReader(open(glob.glob('$OUT/r02_stall_${C}_kt/k_kernel_trace.csv')[0])))
agg=collections.defaultdict(list)
for r in rows: agg[short(r['Kernel_Name'])].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
tot=sum(sum(v) for v in agg.values())
for k,v in sorted(agg.items(), key=lambda kv:-sum(kv[1])):
    if sum(v)/tot>0.005: print('%-40s n=%4d avg %9.1f us  total %8.2f ms (%.1f %%)'%(k,len(v),sum(v)/len(v),sum(v)/1e3,100*sum(v)/tot))
ctr=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('$OUT/r02_stall_${C}_SQ*/p_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        ctr[short(r['Kernel_Name'])][r['Counter_Name']].append((int(r['Grid_Size']), float(r['Counter_Value'])))
for k in ('k_cell_apply_mfma','k_bn_stats_mfma','k_bn_stats'):
    if k not in ctr: continue
    print(k)
    for c,rows in sorted(ctr[k].items()):
        g=max(x for x,_ in rows); vals=[v for x,v in rows if x==g]
        print('   %-34s %16.0f'%(c,sum(vals)/len(vals)))
PY
