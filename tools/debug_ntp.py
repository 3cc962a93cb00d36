#!/usr/bin/env python
"""Exact layout check of the persistent ring experiment (diagnostic library, MVPTR_GEMM_CFG=p): A = tiled identity,
B = integers; prints where the output differs.  MVPTR_LIB=diag python tools/debug_ntp.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvp_pytorch_amd import hip  # noqa: E402

dev = torch.device("cuda:0")
CFG = os.environ.get("NTP_CFG", "p")
if os.environ.get("MVPTR_LIB") == "diag":
    hip.set_knob("MVPTR_GEMM_CFG", CFG)


def run(M, N, K):
    a = torch.zeros(M, K, dtype=torch.bfloat16, device=dev)
    r = torch.arange(M, device=dev)
    a[r, r % K] = 1
    b = ((torch.arange(N * K, device=dev, dtype=torch.float32).reshape(N, K) * 7) % 251 - 125).to(torch.bfloat16)
    ref = b.float().t()[r % K]            # [M, N]
    out = hip.gemm_nt(a, b, hip.EPI_BIAS).float()
    bad = out != ref
    print("M %d N %d K %d: %d / %d elements differ" % (M, N, K, int(bad.sum()), M * N))
    if bad.any():
        rows = bad.any(1).nonzero().view(-1)
        cols = bad.any(0).nonzero().view(-1)
        print("  bad rows: %d (first %s)  bad cols: %d (first %s)" % (rows.numel(), rows[:8].tolist(), cols.numel(), cols[:8].tolist()))
        r0 = int(rows[0])
        print("  row %d out[:32]  %s" % (r0, out[r0, :32].int().tolist()))
        print("  row %d ref[:32]  %s" % (r0, ref[r0, :32].int().tolist()))
        # where does each output value of row r0 come from in ref (same row)?
        src = []
        for c in range(32):
            m = (ref[r0] == out[r0, c]).nonzero().view(-1)
            src.append(int(m[0]) if m.numel() else -1)
        print("  row %d: out col c holds ref col %s" % (r0, src))
        # per 256x256 tile error map
        tm, tn = (M + 255) // 256, N // 256
        cnt = torch.zeros(tm, tn, dtype=torch.long)
        for i in range(tm):
            for j in range(tn):
                cnt[i, j] = int(bad[i * 256:(i + 1) * 256, j * 256:(j + 1) * 256].sum())
        print("  bad elements per tile:\n%s" % cnt)


for M, N, K in ((4200, 2304, 768), (256 * 70, 768, 256), (70000, 768, 768)):
    run(M, N, K)


def rel(a, b):
    a, b = a.float(), b.float()
    return ((a - b).norm() / (b.norm() + 1e-12)).item()


def epilogues(M, N, K):
    """every epilogue of the experiment kernel against f32 torch (several tiles per workgroup)"""
    g = torch.Generator(device="cpu").manual_seed(M + N)
    a = (torch.randn(M, K, generator=g)).to(torch.bfloat16).to(dev)
    b = (torch.randn(N, K, generator=g) * 0.05).to(torch.bfloat16).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    aux = torch.randn(M, N, generator=g).to(torch.bfloat16).to(dev)
    base = a.float() @ b.float().t()
    errs = {}
    errs["bias"] = rel(hip.gemm_nt(a, b, hip.EPI_BIAS, bias=bias), base + bias)
    dq, act = hip.gemm_nt(a, b, hip.EPI_BIAS_GELU, bias=bias)
    u = (base + bias).requires_grad_(True)
    ar = torch.nn.functional.gelu(u)
    ar.sum().backward()
    errs["gelu"] = rel(act, ar.detach())
    errs["dgelu_abs"] = (hip.dgelu_decode(dq) - u.grad).abs().max().item()
    errs["resid"] = rel(hip.gemm_nt(a, b, hip.EPI_BIAS_RESID, bias=bias, aux=aux), base + bias + aux.float())
    errs["add"] = rel(hip.gemm_nt(a, b, hip.EPI_ADD, aux=aux), base + aux.float())
    gq = hip.dgelu_encode(torch.rand(M, N, generator=g) * 1.25 - 0.125).to(dev)
    vec = torch.zeros(N, device=dev)
    errs["gelu_bwd"] = rel(hip.gemm_nt(a, b, hip.EPI_GELU_BWD, aux=gq, vec_out=vec), base * hip.dgelu_decode(gq))
    errs["colsum"] = rel(vec, (base * hip.dgelu_decode(gq)).sum(0))
    print("cfg %s M %d N %d K %d:" % (CFG, M, N, K), {k: "%.2e" % v for k, v in errs.items()})
    assert all(v < 5e-3 for v in errs.values()), errs


epilogues(70000, 768, 768)
epilogues(33100, 2304, 768)
epilogues(20000, 768, 3072)
