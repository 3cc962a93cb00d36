O=gpurun_out/r04y; mkdir -p $O
cd tools; timeout 1500 python3 soak.py 400 256 > ../$O/soak.txt 2>&1; tail -6 ../$O/soak.txt
timeout 900 python3 grad_repro.py > ../$O/grad_repro.txt 2>&1; tail -8 ../$O/grad_repro.txt
