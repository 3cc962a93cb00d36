O=gpurun_out/r02ay; mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
Q="--kernel-trace --output-format csv"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE $Q -d $O/pmc_mfma -- python3 tools/prof_dominant.py 2 full > $O/pmc.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES $Q -d $O/pmc_wait -- python3 tools/prof_dominant.py 2 full >> $O/pmc.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA $Q -d $O/pmc_lds -- python3 tools/prof_dominant.py 2 full >> $O/pmc.log 2>&1
echo "pass,kernel,counter,launches,mean_value" > $O/r02_pmc_mfma_lds.csv
python3 tools/pmc_summary.py $O/pmc_mfma $O/pmc_wait $O/pmc_lds | grep "gemm_tn_q\|gemm_nt_kernel" >> $O/r02_pmc_mfma_lds.csv
python3 - <<'PY'
import csv, glob
from collections import defaultdict
# kernel durations from the traces of the first pass
dur = defaultdict(list)
for f in glob.glob('gpurun_out/r02ay/pmc_mfma/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r['Kernel_Name'][:60]].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
for k, v in dur.items():
    if 'gemm_t' in k or 'gemm_nt' in k:
        print(k, len(v), sum(v) / len(v) / 1e3, 'us')
PY
cat $O/r02_pmc_mfma_lds.csv | cut -c1-200
find $O -name "*kernel_trace.csv" -size +2M -delete; find $O -name "*counter_collection.csv" -size +8M -delete
