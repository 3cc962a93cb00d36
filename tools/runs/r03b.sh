O=gpurun_out/r03b; mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "adamw or sgemm or l2norm or gather_and or pack_maps" > $O/ops.log 2>&1; echo "rc=$?" >> $O/ops.log
tail -12 $O/ops.log | cut -c1-400
timeout 900 python -m pytest tests/test_model_gpu.py -m gpu -q -x -s -k "packed_pipeline or stream_placement" > $O/packed.log 2>&1; echo "rc=$?" >> $O/packed.log
tail -25 $O/packed.log | cut -c1-600
timeout 1200 python -m pytest tests/test_dp_gpu.py -m gpu -q -x -s > $O/dp.log 2>&1; echo "rc=$?" >> $O/dp.log
grep -v Gloo $O/dp.log | tail -25 | cut -c1-600
timeout 1500 python -m pytest tests -m gpu -q --deselect tests/test_dp_gpu.py > $O/gputest.log 2>&1; echo "pytest rc=$?" >> $O/gputest.log
tail -12 $O/gputest.log | cut -c1-400
for i in 1 2; do
python bench.py --steps 10 --warmup 3 --no-extras > $O/bench_packed_$i.json 2> $O/bench_packed_$i.err; python -c "import json;d=json.load(open('$O/bench_packed_$i.json'));print('packed pipeline',d['ms_per_step'])"
MVPTR_PACKED_PIPELINE=0 python bench.py --steps 10 --warmup 3 --no-extras > $O/bench_general_$i.json 2> $O/bench_general_$i.err; python -c "import json;d=json.load(open('$O/bench_general_$i.json'));print('general path',d['ms_per_step'])"
done
python bench.py --steps 10 --warmup 3 --no-extras --fixed-length > $O/bench_fixed.json 2> $O/bench_fixed.err; python -c "import json;d=json.load(open('$O/bench_fixed.json'));print('fixed',d['ms_per_step'])"
tail -3 $O/bench_packed_1.err
