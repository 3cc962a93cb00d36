O=gpurun_out/r04w; mkdir -p $O
cd tools; timeout 600 python3 bench_ln.py > ../$O/bench_ln.txt 2>&1; cat ../$O/bench_ln.txt
