O=gpurun_out/r04s; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu > $O/tests_gpu.txt 2>&1; tail -4 $O/tests_gpu.txt
bash tools/runs/r04_profile.sh > $O/profile.log 2>&1; tail -30 $O/profile.log
timeout 900 python bench.py > $O/bench_default.txt 2>&1; tail -1 $O/bench_default.txt | cut -c1-1500
