O=gpurun_out/r04p; mkdir -p $O
cd tools
timeout 600 python3 fill_sites.py > ../$O/fill_sites.txt 2>&1; tail -90 ../$O/fill_sites.txt
