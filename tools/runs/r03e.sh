O=gpurun_out/r03e; mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > $O/gputest.log 2>&1; echo "pytest rc=$?" >> $O/gputest.log
grep -v Gloo $O/gputest.log | tail -8 | cut -c1-400
for i in 1 2 3; do
python bench.py --steps 10 --warmup 3 --no-extras > $O/bench_packed_$i.json 2> $O/bench_packed_$i.err; python -c "import json;d=json.load(open('$O/bench_packed_$i.json'));print('packed pipeline',d['ms_per_step'])"
MVPTR_PACKED_PIPELINE=0 python bench.py --steps 10 --warmup 3 --no-extras --no-arena > $O/bench_general_$i.json 2> $O/bench_general_$i.err; python -c "import json;d=json.load(open('$O/bench_general_$i.json'));print('general path, no arena',d['ms_per_step'])"
MVPTR_LIB=diag MVPTR_NT_EXP=512 python bench.py --steps 10 --warmup 3 --no-extras > $O/bench_nt_$i.json 2> $O/bench_nt_$i.err; python -c "import json;d=json.load(open('$O/bench_nt_$i.json'));print('packed + nt stash stores',d['ms_per_step'])"
done
python bench.py --steps 10 --warmup 3 --no-extras --fixed-length > $O/bench_fixed.json 2> $O/bench_fixed.err; python -c "import json;d=json.load(open('$O/bench_fixed.json'));print('fixed',d['ms_per_step'])"
MVPTR_LIB=diag MVPTR_NT_EXP=512 python bench.py --steps 10 --warmup 3 --no-extras --fixed-length > $O/bench_fixed_nt.json 2> $O/bench_fixed_nt.err; python -c "import json;d=json.load(open('$O/bench_fixed_nt.json'));print('fixed + nt',d['ms_per_step'])"
python bench.py --steps 10 --warmup 3 --no-extras --with-input-pipeline > $O/bench_piped.json 2> $O/bench_piped.err; python -c "import json;d=json.load(open('$O/bench_piped.json'));print('with input pipeline',d['ms_per_step'], d['config']['with_input_pipeline'])"
tail -3 $O/bench_piped.err
P="--kernel-trace --stats --output-format csv"
rocprofv3 $P -d $O/packed -o packed -- python3 bench.py --steps 10 --warmup 3 --no-extras > $O/bench_packed_under_rocprof.log 2>&1
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r03e/packed/packed_kernel_stats.csv')))
for r in rows:
    if any(k in r['Name'] for k in ('sgemm','cast_rows','cast_pack','scatter','adamw')):
        print("%-70s calls/step %6.1f  total/step %7.3f ms  avg %8.1f us" % (r['Name'][:70], int(r['Calls'])/13, float(r['TotalDurationNs'])/1e6/13, float(r['AverageNs'])/1e3))
print("launches/step", sum(int(r['Calls']) for r in rows)/13, "kernel ms/step", sum(float(r['TotalDurationNs']) for r in rows)/1e6/13)
PY
find $O -name "*kernel_trace.csv" -size +2M -delete
