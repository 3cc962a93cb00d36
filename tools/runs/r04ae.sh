O=gpurun_out/r04ae; mkdir -p $O
cd tools; timeout 600 python3 torch_launch_sites.py > ../$O/sites.txt 2>&1; tail -80 ../$O/sites.txt
