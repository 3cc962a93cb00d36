O=gpurun_out/r04ag; mkdir -p $O
timeout 2400 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py tests/test_dp_gpu.py -x -q -m gpu -k "compact or glue or parity or pretrain or packed or configs1 or dp or rccl or launcher or sync_free" > $O/tests.txt 2>&1; tail -3 $O/tests.txt
timeout 600 python bench.py --steps 20 --warmup 5 --no-extras > $O/bench.txt 2>&1; echo "bench $(grep -o '"ms_per_step": [0-9.]*' $O/bench.txt | head -1)"
