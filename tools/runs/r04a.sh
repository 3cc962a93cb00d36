# round 4, first call: cold table of every step GEMM against hipBLASLt + the bench line of the round-3 code on this box
O=gpurun_out/r04a; mkdir -p $O
python3 tools/blas_table.py > $O/blas_table.txt 2>&1
python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
tail -40 $O/blas_table.txt; cut -c1-600 $O/bench.json
