# full GPU suite + bench of the round-4 state (8-bit gelu' stash, ADVICE fixes, new parity tests)
O=gpurun_out/r04g; mkdir -p $O
timeout 1500 python3 -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.txt
tail -25 $O/pytest_gpu.txt; grep PARITY $O/pytest_gpu.txt > $O/parity_values.txt
timeout 600 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err; cut -c1-1500 $O/bench.json
