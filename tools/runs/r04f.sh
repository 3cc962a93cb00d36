O=gpurun_out/r04f; mkdir -p $O
cd tools && timeout 600 python3 exp_ntp_stores.py > ../$O/stores.txt 2>&1; cd ..; cat $O/stores.txt
