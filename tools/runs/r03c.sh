O=gpurun_out/r03c; mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_ops_gpu.py -m gpu -q -x -k "packed_pipeline or stream_placement or unpadded_equals or gather_and" > $O/packed.log 2>&1; echo "rc=$?" >> $O/packed.log
tail -8 $O/packed.log | cut -c1-400
timeout 900 python -m pytest tests/test_dp_gpu.py -m gpu -q -x -s > $O/dp.log 2>&1; echo "rc=$?" >> $O/dp.log
grep -v Gloo $O/dp.log | tail -12 | cut -c1-700
P="--kernel-trace --stats --output-format csv"
rocprofv3 $P -d $O/packed -o packed -- python3 bench.py --steps 10 --warmup 3 --no-extras > $O/bench_packed_under_rocprof.log 2>&1
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r03c/packed/packed_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("total kernel ms per 13 steps", tot/1e6, "launches", sum(int(r['Calls']) for r in rows))
for r in rows[:45]:
    print("%-90s calls %6s  total %9.3f ms  avg %8.1f us" % (r['Name'][:90], r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3))
PY
find $O -name "*kernel_trace.csv" -size +2M -delete
echo "--- sweep hot"
python tools/sweep_gemm_cfg.py t256k,r4,r5,t256 37748,10917 2>&1 | tail -16
echo "--- sweep cold"
SWEEP_COLD=1 python tools/sweep_gemm_cfg.py t256k,r4,r5,t256 37748,10917 2>&1 | tail -16
