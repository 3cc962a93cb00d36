#!/usr/bin/env python
"""Which weight-gradient shapes of the stack launch re-fetch their operands?  One mvptr_gemm_tn_stack launch per problem TYPE (24
problems of the same shape, separate operands) for rocprofv3 --pmc FETCH_SIZE; prints the operand bytes of every launch in
dispatch order (tools/pmc_kernel.py lists the counter in the same order).   prof_tn_shapes.py [M]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvp_pytorch_amd import hip  # noqa: E402

dev = torch.device("cuda:0")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 10917
H, I = 768, 3072
TYPES = [("ffn2 dW [768 x 3072]  (3 x 12 tiles)", H, I), ("ffn1 dW [3072 x 768]  (12 x 3 tiles)", I, H), ("attn-out dW [768 x 768]  (3 x 3)", H, H),
         ("qkv dW [2304 x 768]  (9 x 3)", 3 * H, H)]
for name, N, K in TYPES:
    probs = []
    for _ in range(24):
        dy = (torch.randn(M, N, device=dev) * 0.5).to(torch.bfloat16)
        x = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16)
        probs.append((dy, x, torch.zeros(N, K, device=dev), None))
    hip.gemm_tn_stack(probs)
    torch.cuda.synchronize()
    tiles = 24 * ((N + 255) // 256) * ((K + 255) // 256)
    print("%-40s operands %.1f MB, results %.1f MB, %d tiles = %.2f rounds of 256" % (name, 24 * M * (N + K) * 2 / 1e6, 24 * N * K * 4 / 1e6, tiles, tiles / 256.0), flush=True)
    del probs
    torch.cuda.empty_cache()
# the step's own mix (6 layers x the four types), without and with the bias-gradient column sums of FFN1 and Q|K|V, then one type with them
for name, with_cs in (("mix of the four types, no column sums", False), ("mix, column sums on ffn1 / qkv (the step)", True)):
    probs = []
    for _ in range(6):
        for (_, N, K), cs in zip(TYPES, (False, True, False, True)):
            dy = (torch.randn(M, N, device=dev) * 0.5).to(torch.bfloat16)
            x = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16)
            probs.append((dy, x, torch.zeros(N, K, device=dev), torch.zeros(N, device=dev) if (cs and with_cs) else None))
    hip.gemm_tn_stack(probs)
    torch.cuda.synchronize()
    print("%-40s operands %.1f MB" % (name, 6 * M * 12288 * 2 / 1e6), flush=True)
    del probs
    torch.cuda.empty_cache()
probs = []
for _ in range(24):
    dy = (torch.randn(M, I, device=dev) * 0.5).to(torch.bfloat16)
    x = (torch.randn(M, H, device=dev) * 0.5).to(torch.bfloat16)
    probs.append((dy, x, torch.zeros(I, H, device=dev), torch.zeros(I, device=dev)))
hip.gemm_tn_stack(probs)
torch.cuda.synchronize()
print("%-40s operands %.1f MB" % ("ffn1 dW x 24 WITH column sums", 24 * M * (I + H) * 2 / 1e6), flush=True)
