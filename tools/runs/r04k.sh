# whole GPU suite (parity values printed), then bench
O=gpurun_out/r04k; mkdir -p $O
timeout 2400 python3 -m pytest tests -m gpu -q -rP > $O/pytest_gpu.txt 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.txt
grep -E "passed|failed|rc " $O/pytest_gpu.txt | tail -5; grep "^PARITY" $O/pytest_gpu.txt > $O/parity_values.txt; grep "FAILED\|Error" $O/pytest_gpu.txt | head
timeout 600 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err; cut -c1-600 $O/bench.json
