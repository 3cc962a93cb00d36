# Round 5: stack-wide weight gradients — op tests, model tests that cover the encoder backward, same-box A/B of the step
O=gpurun_out/r05c; mkdir -p $O
cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_ops_gpu.py -x -q -k "gemm_tn" > $O/pytest_tn.log 2>&1; tail -3 $O/pytest_tn.log
python3 -m pytest tests/test_model_gpu.py -x -q -k "deferred or device_side or packed_training or packed_pipeline or bi_pretrain_parity" > $O/pytest_model.log 2>&1; tail -3 $O/pytest_model.log
for i in 1 2; do
python3 bench.py --no-extras --no-cpu-baseline --steps 20 > $O/bench_stack_$i.log 2>&1; tail -1 $O/bench_stack_$i.log | cut -c1-200
python3 bench.py --no-extras --no-cpu-baseline --steps 20 --wgrad-per-layer > $O/bench_layer_$i.log 2>&1; tail -1 $O/bench_layer_$i.log | cut -c1-200
done
python3 bench.py --no-extras --no-cpu-baseline --steps 20 --fixed-length > $O/bench_fixed_stack.log 2>&1; tail -1 $O/bench_fixed_stack.log | cut -c1-200
python3 bench.py --no-extras --no-cpu-baseline --steps 20 --fixed-length --wgrad-per-layer > $O/bench_fixed_layer.log 2>&1; tail -1 $O/bench_fixed_layer.log | cut -c1-200
python3 bench.py --no-extras --no-cpu-baseline --steps 20 --one-stream > $O/bench_one_stack.log 2>&1; tail -1 $O/bench_one_stack.log | cut -c1-200
python3 bench.py --no-extras --no-cpu-baseline --steps 20 --one-stream --wgrad-per-layer > $O/bench_one_layer.log 2>&1; tail -1 $O/bench_one_layer.log | cut -c1-200
