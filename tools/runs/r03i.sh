O=gpurun_out/r03i; mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "gemm_nt or decoder" > $O/ops.log 2>&1; echo "rc=$?" >> $O/ops.log; tail -3 $O/ops.log
for i in 1 2 3; do
python bench.py --steps 10 --warmup 3 --no-extras > $O/bench_new_$i.json 2> $O/bench_new_$i.err; python -c "import json;d=json.load(open('$O/bench_new_$i.json'));print('aux prefetch',d['ms_per_step'])"
MVPTR_LIB=prev python bench.py --steps 10 --warmup 3 --no-extras > $O/bench_prev_$i.json 2> $O/bench_prev_$i.err; python -c "import json;d=json.load(open('$O/bench_prev_$i.json'));print('previous epilogue',d['ms_per_step'])"
done
for i in 1 2; do
python bench.py --steps 10 --warmup 3 --no-extras --fixed-length > $O/bench_fixed_new_$i.json 2> $O/bench_fixed_new_$i.err; python -c "import json;d=json.load(open('$O/bench_fixed_new_$i.json'));print('fixed, aux prefetch',d['ms_per_step'])"
MVPTR_LIB=prev python bench.py --steps 10 --warmup 3 --no-extras --fixed-length > $O/bench_fixed_prev_$i.json 2> $O/bench_fixed_prev_$i.err; python -c "import json;d=json.load(open('$O/bench_fixed_prev_$i.json'));print('fixed, previous epilogue',d['ms_per_step'])"
done
