import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvp_pytorch_amd import hip
dev = torch.device("cuda:0")
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
def rnd(*s): return (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)
for M in (64000, 256 * 256):
  for (N, K, epi, name) in ((2304, 768, hip.EPI_BIAS, "BIAS"), (768, 768, hip.EPI_BIAS, "BIAS"), (768, 3072, hip.EPI_BIAS, "BIAS")):
    a, b = rnd(M, K), rnd(N, K)
    bias = torch.zeros(N, device=dev)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    out1 = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    line = "M=%d N=%d K=%d %s" % (M, N, K, name)
    for cfg, exp in (("qp", 0), ("qp", 8), ("qp", 16), ("qp", 24), ("q", 2)):
        hip.set_knob("MVPTR_GEMM_CFG", cfg); hip.set_knob("MVPTR_NT_EXP", str(exp))
        us = timeit(lambda: hip.gemm_nt(a, b, epi, bias=bias, out=out, out1=out1))
        line += "  %s/%d %.1fus %.0fTF" % (cfg, exp, us, 2.0 * M * N * K / us / 1e6)
    print(line, flush=True)
hip.set_knob("MVPTR_GEMM_CFG", ""); hip.set_knob("MVPTR_NT_EXP", "0")
