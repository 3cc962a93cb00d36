O=gpurun_out/r04ae; mkdir -p $O
cd tools; timeout 600 python3 torch_launch_stacks.py > ../$O/stacks.txt 2>&1; tail -110 ../$O/stacks.txt
