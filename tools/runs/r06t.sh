cd $GRAFT_REPO_ROOT
O=gpurun_out/r06t; mkdir -p $O
python3 -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; grep -v amdgpu $O/pytest.log | grep -E "^FAILED|^ERROR|passed|failed|^E  " | tail -8 | cut -c1-300
python3 - <<'PY'
import torch
dev = torch.device("cuda:0")
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    torch.manual_seed(5)
    out = torch.empty(3, 8, dtype=torch.long, device=dev)
    g = torch.cuda.CUDAGraph()
    static = torch.empty(8, dtype=torch.long, device=dev)
    with torch.cuda.graph(g, stream=s):
        static.copy_(torch.randperm(8, device=dev))
    for i in range(3):
        g.replay(); out[i] = static
torch.cuda.synchronize()
print("randperm under replay (three replays must differ):", out.tolist())
PY
