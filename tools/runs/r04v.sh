O=gpurun_out/r04v; mkdir -p $O
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/p -o p -- python3 bench.py --steps 10 --warmup 3 --no-extras > $O/log.txt 2>&1
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open("gpurun_out/r04v/p/p_kernel_stats.csv")))
for r in rows:
    n=r['Name']
    if any(k in n for k in ('tap_','FillFunctor','bfloat16_copy','scatter_add','gather_rows','Memset','fillBuffer')):
        print("%-90s %5s %9.3f ms/step avg %7.1f us"%(n[:90], r['Calls'], float(r['TotalDurationNs'])/13e6, float(r['AverageNs'])/1e3))
PY
find $O -name "*kernel_trace.csv" -delete
