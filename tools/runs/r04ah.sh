O=gpurun_out/r04ah; mkdir -p $O
cd tools; MVPTR_LIB=diag timeout 900 python3 blas_table.py --ms 37748,10917,64000 --ab --cfg f > ../$O/f.txt 2>&1; cat ../$O/f.txt | cut -c1-110
MVPTR_LIB=diag MVPTR_NT_EXP=1024 timeout 900 python3 blas_table.py --ms 37748 --ab --cfg f > ../$O/f_loop.txt 2>&1; echo LOOP-ONLY; cat ../$O/f_loop.txt | cut -c1-110
