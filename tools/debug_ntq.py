import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvp_pytorch_amd import hip
dev = torch.device("cuda:0")
torch.manual_seed(0)
for (M, N, K) in [(256, 256, 32), (256, 256, 64), (256, 256, 128), (256, 256, 160), (256, 256, 256), (512, 512, 768)]:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    b = torch.randn(N, K, device=dev).to(torch.bfloat16)
    ref = a.float() @ b.float().t()
    hip.set_knob("MVPTR_GEMM_CFG", "q")
    out = hip.gemm_nt(a, b, hip.EPI_F32)
    hip.set_knob("MVPTR_GEMM_CFG", "")
    err = ((out - ref).norm() / ref.norm()).item()
    bad = (out - ref).abs() > 1e-2 * ref.abs().max()
    print(M, N, K, "rel err", err, "bad frac", bad.float().mean().item())
    if bad.any():
        rows = bad.any(1).nonzero().flatten()
        cols = bad.any(0).nonzero().flatten()
        print("  bad rows", rows[:16].tolist(), len(rows), " bad cols", cols[:16].tolist(), len(cols))
# identity test: A = I(256x256 as K=256), B[n][k] = n*1000 + k  -> out[m][n] = B[n][m]
K = 256
a = torch.eye(256, K, device=dev).to(torch.bfloat16)
b = (torch.arange(256, device=dev).float()[:, None] * 0 + torch.arange(K, device=dev).float()[None, :]).to(torch.bfloat16)  # B[n][k] = k
hip.set_knob("MVPTR_GEMM_CFG", "q")
out = hip.gemm_nt(a, b, hip.EPI_F32)   # expect out[m][n] = m
print("k-index seen by row m (expect m):", out[:40, 0].tolist())
b2 = (torch.arange(256, device=dev).float()[:, None] + 0 * torch.arange(K, device=dev).float()[None, :]).to(torch.bfloat16)  # B[n][k] = n
out = hip.gemm_nt(a, b2, hip.EPI_F32)  # expect out[m][n] = n
print("n-index seen at col n (expect n):", out[0, :40].tolist())
print("n-index row 35:", out[35, 120:140].tolist())
hip.set_knob("MVPTR_GEMM_CFG", "")
