O=gpurun_out/r02ah; mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python tools/sweep_nt_group.py 37748 2>&1 | grep "^M=" | tee $O/sweep_37748.log
timeout 600 python tools/sweep_nt_group.py 10917 2>&1 | grep "^M=" | tee $O/sweep_10917.log
PMC_ONCE=1 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 tools/sweep_nt_group.py 37748 > $O/pmc.log 2>&1
python3 - <<'PY'
import csv, glob
rows=[]
for f in glob.glob('gpurun_out/r02ah/pmc_fetch/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if r['Counter_Name']=='FETCH_SIZE' and 'gemm_nt_kernel' in r['Kernel_Name']:
            rows.append((int(r['Dispatch_Id']), r['Kernel_Name'][40:70], float(r['Counter_Value'])))
rows.sort()
print(len(rows))
for i in range(0, len(rows), 9):
    print(rows[i][1], ' '.join('%6.0f' % (2*x[2]*1024/1e6) for x in rows[i:i+9]), 'MB fetched (x2 calibration)')
PY
find $O -name "*kernel_trace.csv" -size +2M -delete
