O=gpurun_out/r04n; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "attention" > $O/test_attn.txt 2>&1; tail -3 $O/test_attn.txt
cd tools
timeout 600 python3 bench_attn.py > ../$O/bench_attn2.txt 2>&1; cat ../$O/bench_attn2.txt
