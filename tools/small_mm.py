#!/usr/bin/env python
"""How long the step's small f32 matrix products take through torch (hipBLASLt picks one 256x256 tile
for a [256,768]x[768,256] product: a single workgroup), and what the alternatives cost."""
import torch

dev = torch.device("cuda:0")


def timeit(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


x = torch.randn(256, 768, device=dev)
w = torch.randn(768, 256, device=dev)
g = torch.randn(256, 256, device=dev)
ref = (x.double() @ w.double()).float()
forms = {
    "x @ w": lambda: x @ w,
    "(w.t() @ x.t()).t()": lambda: (w.t() @ x.t()).t(),
    "split-K 6 bmm + sum": lambda: torch.bmm(x.view(256, 6, 128).transpose(0, 1), w.view(6, 128, 256)).sum(0),
    "split-K 12 bmm + sum": lambda: torch.bmm(x.view(256, 12, 64).transpose(0, 1), w.view(12, 64, 256)).sum(0),
    "row-split 8 bmm": lambda: torch.bmm(x.view(8, 32, 768), w.expand(8, 768, 256)).view(256, 256),
    "dX: g @ w.t()": lambda: g @ w.t(),
    "dW: x.t() @ g": lambda: x.t() @ g,
    "dW split-M 8 bmm + sum": lambda: torch.bmm(x.view(8, 32, 768).transpose(1, 2), g.view(8, 32, 256)).sum(0),
    "sim: g @ g.t()": lambda: g @ g.t(),
    "pooler [512,768]x[768,768]": lambda: torch.randn(1, device=dev) if False else (torch.empty(512, 768, device=dev) @ torch.empty(768, 768, device=dev)),
}
for lib in ("default", "cublas", "cublaslt"):
    if lib != "default":
        try:
            torch.backends.cuda.preferred_blas_library(lib)
        except Exception as e:  # noqa: BLE001
            print("preferred_blas_library(%s): %s" % (lib, e))
            continue
    print("== blas library:", lib)
    for name, fn in forms.items():
        out = fn()
        err = (out - ref).abs().max().item() if out.shape == ref.shape and name.split()[0] in ("x", "(w.t()", "split-K", "row-split") else float("nan")
        print("  %-28s %7.1f us   max err vs f64 %.2e" % (name, timeit(fn), err))
