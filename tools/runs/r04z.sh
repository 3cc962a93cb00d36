O=gpurun_out/r04z; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu -k "layer or encoder or parity or packed or layernorm or pretrain" > $O/tests.txt 2>&1; tail -3 $O/tests.txt
for i in 1 2; do timeout 600 python bench.py --steps 20 --warmup 5 --no-extras > $O/bench_$i.txt 2>&1; echo "bench $(grep -o '"ms_per_step": [0-9.]*' $O/bench_$i.txt | head -1)"; done
