O=gpurun_out/r04x; mkdir -p $O
timeout 1800 python -m pytest tests/test_dp_gpu.py -x -q -m gpu > $O/test_dp.txt 2>&1; tail -4 $O/test_dp.txt
