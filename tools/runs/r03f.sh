O=gpurun_out/r03f; mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_dp_gpu.py -m gpu -q -x -s > $O/dp.log 2>&1; echo "rc=$?" >> $O/dp.log
grep -v Gloo $O/dp.log | grep -E "one-rank|passed|failed|^E " | cut -c1-500
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_ops_gpu.py tests/test_pipeline_gpu.py -m gpu -q -rP > $O/prints.log 2>&1; echo "pytest rc=$?" >> $O/prints.log
tail -4 $O/prints.log | cut -c1-300
grep -E "sim_mat max abs err|hard indices|losses|rel L2|worst|grad-norm|adamw probe|max abs diff|err " $O/prints.log | cut -c1-260 > $O/parity_values.txt
wc -l $O/parity_values.txt
for i in 1 2 3; do
python bench.py --steps 10 --warmup 3 --no-extras --fixed-length > $O/bench_fixed_$i.json 2> $O/bench_fixed_$i.err; python -c "import json;d=json.load(open('$O/bench_fixed_$i.json'));print('fixed',d['ms_per_step'])"
MVPTR_LIB=diag MVPTR_NT_EXP=512 python bench.py --steps 10 --warmup 3 --no-extras --fixed-length > $O/bench_fixed_nt_$i.json 2> $O/bench_fixed_nt_$i.err; python -c "import json;d=json.load(open('$O/bench_fixed_nt_$i.json'));print('fixed + nt',d['ms_per_step'])"
done
