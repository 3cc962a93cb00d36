O=gpurun_out/r04aa; mkdir -p $O
run() { MVPTR_LIB=diag MVPTR_NT_GROUP=$1 timeout 600 python bench.py --steps 20 --warmup 5 --no-extras $2 > $O/b_$1_$3.txt 2>&1; echo "group=$1 $2 $(grep -o '"ms_per_step": [0-9.]*' $O/b_$1_$3.txt | head -1)"; }
for rep in 1 2; do
for g in "" "4,4" "8,4" "6,6" "8,6" "4,3" "8,3" "12,4" "6,2"; do run "$g" "" $rep; done
done
for g in "" "8,4" "6,6" "8,6" "12,4"; do run "$g" "--fixed-length" f; done
