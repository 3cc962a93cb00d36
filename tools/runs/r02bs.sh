O=gpurun_out/r02bs; mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/tr -o tr -- python3 bench.py --steps 4 --warmup 2 --no-extras > $O/bench.log 2>&1
python3 - <<'PY'
import csv, glob
from collections import defaultdict
f = glob.glob('gpurun_out/r02bs/tr/**/*kernel_trace.csv', recursive=True)[0]
acc = defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name']
    if 'gemm_nt_kernel' in n or 'gemm_tn_q' in n:
        key = (n.split('gemm_')[1][:34], int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']) if 'Workgroup_Size_X' in r else int(r['Grid_Size_X']))
        acc[key].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k in sorted(acc, key=lambda k: -sum(acc[k])):
    v = acc[k]
    if len(v) >= 6:
        print("%-36s wgs %5d  x%4d  mean %7.1f us  min %7.1f  max %7.1f" % (k[0], k[1], len(v), sum(v) / len(v), min(v), max(v)))
PY
find $O -name "*kernel_trace.csv" -size +2M -delete
