O=gpurun_out/r04q; mkdir -p $O
timeout 1200 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "gemm_nt" > $O/test_nt.txt 2>&1; tail -3 $O/test_nt.txt
timeout 1200 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "device_side_row_count" > $O/test_rows.txt 2>&1; tail -3 $O/test_rows.txt
cd tools
MVPTR_LIB=diag MVPTR_NT_EXP=65536 timeout 900 python3 blas_table.py --ms 37748,10917,64000 --ab --cfg m6 > ../$O/nosplit_m6.txt 2>&1
MVPTR_LIB=diag timeout 900 python3 blas_table.py --ms 37748,10917,64000 --ab --cfg m4 > ../$O/split_m4.txt 2>&1
MVPTR_LIB=diag timeout 900 python3 blas_table.py --ms 37748,10917,64000 --ab --cfg m2 > ../$O/split_m2.txt 2>&1
paste <(cut -c1-62 ../$O/nosplit_m6.txt) <(cut -c41-52,84- ../$O/split_m4.txt) <(cut -c84- ../$O/split_m2.txt)
