O=gpurun_out/r04m; mkdir -p $O
cd tools; MVPTR_LIB=diag timeout 900 python3 pad_operands.py 37748 64 > ../$O/pad64.txt 2>&1; cat ../$O/pad64.txt
MVPTR_LIB=diag timeout 900 python3 pad_operands.py 37748 192 > ../$O/pad192.txt 2>&1; cat ../$O/pad192.txt
MVPTR_LIB=diag timeout 900 python3 pad_operands.py 10917 64 > ../$O/pad64_small.txt 2>&1; cat ../$O/pad64_small.txt
