O=gpurun_out/r03h; mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "cast_pack or gemm_nt" > $O/ops.log 2>&1; echo "rc=$?" >> $O/ops.log; tail -3 $O/ops.log
timeout 900 python tools/exp_rowtile.py > $O/rowtile.txt 2>&1; cat $O/rowtile.txt | cut -c1-260
P="--kernel-trace --stats --output-format csv"
rocprofv3 $P -d $O/feat -o feat -- python3 tools/prof_features.py 8 > $O/feat.log 2>&1
python3 - <<'PY'
import csv
for r in csv.DictReader(open('gpurun_out/r03h/feat/feat_kernel_stats.csv')):
    if 'cast_rows' in r['Name'] or 'gemm_nt' in r['Name']:
        print(r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3, "us")
PY
find $O -name "*kernel_trace.csv" -size +2M -delete
