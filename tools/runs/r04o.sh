O=gpurun_out/r04o; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu -k "attention or parity or packed or encoder" > $O/tests.txt 2>&1; tail -3 $O/tests.txt
timeout 600 python bench.py --steps 20 --warmup 5 --no-extras > $O/bench.txt 2>&1; tail -1 $O/bench.txt | cut -c1-300
