O=gpurun_out/r04c; mkdir -p $O
timeout 300 python3 tools/debug_ntp.py > $O/debug.txt 2>&1; cat $O/debug.txt | head -80
