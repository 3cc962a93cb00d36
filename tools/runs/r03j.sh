O=gpurun_out/r03j; mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -m gpu -q -x -k "gemm_nt or decoder or bi_pretrain_parity or single_pretrain" > $O/ops.log 2>&1; echo "rc=$?" >> $O/ops.log; tail -3 $O/ops.log
for i in 1 2 3; do
python bench.py --steps 10 --warmup 3 --no-extras > $O/bench_new_$i.json 2> $O/bench_new_$i.err; python -c "import json;d=json.load(open('$O/bench_new_$i.json'));print('split-K decoder dgrad',d['ms_per_step'])"
MVPTR_NO_SPLITK=1 python bench.py --steps 10 --warmup 3 --no-extras > $O/bench_prev_$i.json 2> $O/bench_prev_$i.err; python -c "import json;d=json.load(open('$O/bench_prev_$i.json'));print('one launch',d['ms_per_step'])"
done
P="--kernel-trace --stats --output-format csv"
rocprofv3 $P -d $O/packed -o packed -- python3 bench.py --steps 10 --warmup 3 --no-extras > $O/bench_packed_under_rocprof.log 2>&1
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r03j/packed/packed_kernel_stats.csv')))
for r in rows:
    if 'gemm_nt_kernel<5' in r['Name'] or 'gemm_nt_kernel<4, 32' in r['Name'] or 'reduce' in r['Name'].lower()[:80] and 'sum' in r['Name']:
        print("%-110s calls/step %6.1f  total/step %7.3f ms  avg %8.1f us" % (r['Name'][:110], int(r['Calls'])/13, float(r['TotalDurationNs'])/1e6/13, float(r['AverageNs'])/1e3))
PY
find $O -name "*kernel_trace.csv" -size +2M -delete
