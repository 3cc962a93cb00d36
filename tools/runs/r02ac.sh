O=gpurun_out/r02ac; mkdir -p $O
cd $GRAFT_REPO_ROOT
for lib in libmvptr_hip.so libmvptr_hip_exp2.so libmvptr_hip_exp3.so libmvptr_hip_exp4.so libmvptr_hip_exp6.so; do
echo "== $lib"
MVPTR_TOOL_LIB=$lib timeout 300 python tools/sweep_tn_group.py h 0 64000 0 2>&1 | grep "^M=" | tee -a $O/ablate_h.log
done
