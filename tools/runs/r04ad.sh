O=gpurun_out/r04ad; mkdir -p $O
cd tools; timeout 600 python3 stash_policy.py > ../$O/stash.txt 2>&1; cat ../$O/stash.txt
