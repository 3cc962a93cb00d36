#!/usr/bin/env python
"""A few launches of each GEMM shape of the step, for rocprofv3 --pmc runs (counter collection
replays kernels, so keep it short)."""
import os
os.environ.setdefault("MVPTR_LIB", "diag")   # kernel-configuration knobs live in the diagnostic build only (make -C mvp_pytorch_amd/csrc diag)
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvp_pytorch_amd import hip  # noqa: E402

dev = torch.device("cuda:0")
M, H, I = 64000, 768, 3072


def rnd(*s):
    return torch.randn(*s, device=dev).to(torch.bfloat16)


x, xi = rnd(M, H), rnd(M, I)
w_qkv, w_i, w_out = rnd(3 * H, H), rnd(I, H), rnd(H, I)
b_qkv, b_i, b_out = torch.zeros(3 * H, device=dev), torch.zeros(I, device=dev), torch.zeros(H, device=dev)
for _ in range(3):
    hip.gemm_nt(x, w_qkv, hip.EPI_BIAS, bias=b_qkv)            # gemm_nt<0>: K=768, N=2304
    hip.gemm_nt(xi, w_out, hip.EPI_BIAS_RESID, bias=b_out, aux=x)  # gemm_nt<2>: K=3072, N=768
    hip.gemm_nt(x, w_i, hip.EPI_BIAS_GELU, bias=b_i)           # gemm_nt<1>: K=768, N=3072
    dw = torch.zeros(I, H, device=dev)
    hip.set_knob("MVPTR_GEMM_TN", "32")
    hip.gemm_tn(xi, x, dw)                                     # gemm_tn<32,1,3>: N=3072, K=768
    hip.set_knob("MVPTR_GEMM_TN", "K")
    hip.gemm_tn(xi, x, dw)                                     # gemm_tn<64,2,2>
    os.environ.pop("MVPTR_GEMM_TN")
B, L, heads = 512, 125, 12
qkv = rnd(B * L, 3 * H)
mask = torch.zeros(B, L, device=dev)
dctx = rnd(B * L, H)
for _ in range(3):
    ctx, lse = hip.attention_fwd(qkv, mask, B, L, heads)
    hip.attention_bwd(qkv, mask, ctx, dctx, lse, B, L, heads)
torch.cuda.synchronize()
print("done")
