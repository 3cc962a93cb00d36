O=gpurun_out/r03a; mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "adamw or sgemm or l2norm or gather_and or fused_adamw" > $O/ops.log 2>&1; echo "rc=$?" >> $O/ops.log
tail -15 $O/ops.log | cut -c1-400
timeout 900 python -m pytest tests/test_dp_gpu.py -m gpu -q -x -s > $O/dp.log 2>&1; echo "rc=$?" >> $O/dp.log
grep -v Gloo $O/dp.log | tail -25 | cut -c1-600
timeout 1500 python -m pytest tests -m gpu -q -x --deselect tests/test_dp_gpu.py > $O/gputest.log 2>&1; echo "pytest rc=$?" >> $O/gputest.log
tail -8 $O/gputest.log | cut -c1-400
for i in 1 2; do
python bench.py --steps 10 --warmup 3 --no-extras > $O/bench_arena_$i.json 2> $O/bench_arena_$i.err; python -c "import json;d=json.load(open('$O/bench_arena_$i.json'));print('arena',d['ms_per_step'])"
python bench.py --steps 10 --warmup 3 --no-extras --no-arena > $O/bench_noarena_$i.json 2> $O/bench_noarena_$i.err; python -c "import json;d=json.load(open('$O/bench_noarena_$i.json'));print('noarena',d['ms_per_step'])"
done
tail -3 $O/bench_arena_1.err
