cd $GRAFT_REPO_ROOT
O=gpurun_out/r06g; mkdir -p $O
python3 -X faulthandler -m pytest tests -x -q -m gpu -k "graphed_step" > $O/graph.log 2>&1; grep -v amdgpu $O/graph.log | tail -60 | cut -c1-400
