# Round-5 baseline: default bench line of the round-4 code on this box + which hipBLASLt kernels serve the step's GEMM shapes
# (kernel names / grid / LDS / VGPRs from a kernel trace of tools/blas_table.py).  Output: gpurun_out/r05a/
O=gpurun_out/r05a; mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python3 bench.py > $O/bench_default.log 2>&1
tail -1 $O/bench_default.log | cut -c1-600
rocprofv3 --kernel-trace --output-format csv -d $O/blas -o blas -- python3 tools/blas_table.py --ms 37748,10917 --reps 2 > $O/blas_table.log 2>&1
python3 - <<'PY' > gpurun_out/r05a/blas_kernels.txt 2>&1
import csv, collections, glob
f = glob.glob('gpurun_out/r05a/blas/**/*kernel_trace.csv', recursive=True)[0]
agg = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name']
    if 'Cijk' not in n and 'gemm_nt' not in n: continue
    key = (n[:200], r.get('Grid_Size_X') or r.get('Grid_Size'), r.get('Workgroup_Size_X') or r.get('Workgroup_Size'), r.get('LDS_Block_Size'), r.get('VGPR_Count'), r.get('Accum_VGPR_Count'), r.get('SGPR_Count'))
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    agg.setdefault(key, []).append(d)
for k, v in agg.items():
    print('%8.1f us x%d  grid %s wg %s lds %s vgpr %s agpr %s sgpr %s\n    %s' % (sum(v)/len(v), len(v), k[1], k[2], k[3], k[4], k[5], k[6], k[0]))
PY
cat $O/blas_table.log | tail -22
head -c 6000 $O/blas_kernels.txt
find $O -name "*kernel_trace.csv" -size +2M -delete
