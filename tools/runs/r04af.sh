O=gpurun_out/r04af; mkdir -p $O
timeout 2400 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -x -q -m gpu -k "glue or decoder or mlm or parity or pretrain or packed or finetune or single or branches or vqa" > $O/tests.txt 2>&1; tail -3 $O/tests.txt
for i in 1 2; do timeout 600 python bench.py --steps 20 --warmup 5 --no-extras > $O/bench_$i.txt 2>&1; echo "bench $(grep -o '"ms_per_step": [0-9.]*' $O/bench_$i.txt | head -1)"; done
