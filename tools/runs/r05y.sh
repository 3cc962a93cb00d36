# Round 5: bias-gradient column sums shared out over all tiles / waves of a problem (tn stack): tests, time, FETCH_SIZE, step
O=gpurun_out/r05y; mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python3 -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "gemm_tn" > $O/pytest_tn.log 2>&1; tail -2 $O/pytest_tn.log
python3 -m pytest tests/test_model_gpu.py -x -q -m gpu -k "deferred or packed_training or pretrain_step or dropout" > $O/pytest_model.log 2>&1; tail -2 $O/pytest_model.log
python3 tools/bench_tn_stack.py --reps 8 2>/dev/null | grep "M=" | cut -c1-110
python3 tools/bench_tn_stack.py --reps 8 --no-colsum 2>/dev/null | grep "M=" | cut -c1-110
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python3 tools/prof_tn_shapes.py 10917 > $O/shapes.log 2>&1
python3 tools/pmc_kernel.py $O/fetch FETCH_SIZE gemm_tn_sk_kernel
find $O -name "*counter_collection.csv" -size +8M -delete
for r in 1 2; do python3 bench.py --steps 30 --warmup 8 --no-extras 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('packed', d['ms_per_step'], d['value'])"; done
python3 bench.py --steps 30 --warmup 8 --no-extras --fixed-length 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print('fixed', d['ms_per_step'], d['value'])"
