"""Op-level parity of the HIP kernels (through the C ABI) against plain PyTorch fp32 on the same
bf16-rounded inputs.  Tolerances are for bf16 outputs with f32 accumulation."""
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a = a.float()
    b = b.float()
    return ((a - b).norm() / (b.norm() + 1e-12)).item()


def _bf(x):
    return x.to(torch.bfloat16)


# shapes chosen to reach every tile configuration of the product's shape rule (gemm_nt.hip launch()):
# 128x128 tiles (few tiles), 256x256 / BK 64 (default), 256x128 two workgroups per CU (narrow short-K with many tiles)
@pytest.mark.parametrize("M,N,K", [(256, 768, 768), (200, 2304, 768), (160, 768, 3072), (37, 100, 64), (130, 3129, 768), (500, 768, 2056),
                                   (4200, 2304, 768), (5000, 3072, 136), (70000, 768, 768), (66000, 512, 520)])
def test_gemm_nt_bias(dev, M, N, K):
    from mvp_pytorch_amd import hip
    g = torch.Generator(device="cpu").manual_seed(1)
    a = _bf(torch.randn(M, K, generator=g)).to(dev)
    b = _bf(torch.randn(N, K, generator=g) * 0.05).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    ref = a.float() @ b.float().t() + bias
    out = hip.gemm_nt(a, b, hip.EPI_BIAS, bias=bias)
    err = _rel(out, ref)
    print("gemm_nt bias", M, N, K, err)
    assert err < 4e-3
    out32 = hip.gemm_nt(a, b, hip.EPI_F32, bias=bias)
    assert _rel(out32, ref) < 1e-5


def test_gemm_nt_identity_layout(dev):
    """A = I against an asymmetric B catches transposed / permuted fragment maps exactly."""
    from mvp_pytorch_amd import hip
    K = 128
    a = torch.eye(K, dtype=torch.bfloat16, device=dev)
    b = (torch.arange(192 * K, device=dev, dtype=torch.float32).reshape(192, K) % 251 - 125).to(torch.bfloat16)
    out = hip.gemm_nt(a, b, hip.EPI_F32)
    assert torch.equal(out, b.float().t().contiguous())


def test_gemm_nt_epilogues(dev):
    from mvp_pytorch_amd import hip
    g = torch.Generator(device="cpu").manual_seed(2)
    M, N, K = 300, 768, 256
    a = _bf(torch.randn(M, K, generator=g)).to(dev)
    b = _bf(torch.randn(N, K, generator=g) * 0.1).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    aux = _bf(torch.randn(M, N, generator=g)).to(dev)
    base = a.float() @ b.float().t()
    dq, act = hip.gemm_nt(a, b, hip.EPI_BIAS_GELU, bias=bias)
    uref = (base + bias).requires_grad_(True)
    aref = torch.nn.functional.gelu(uref)
    aref.sum().backward()
    assert _rel(act, aref.detach()) < 4e-3
    # gelu'(u) comes back as 8-bit fixed point (step 1/200) with DITHERED rounding (round 5): |error| < one step, zero mean
    assert dq.dtype == torch.uint8
    err = hip.dgelu_decode(dq) - uref.grad
    assert err.abs().max() < 0.005 + 1e-4
    assert abs(float(err.mean())) < 1e-4, float(err.mean())                      # unbiased over the 230 k elements (sigma / sqrt(n) = 6e-6)
    small = uref.grad.abs() < 0.0025                                             # round-to-nearest stored all of these as exactly 0
    assert abs(float(err[small].mean())) < 3e-4 and float((hip.dgelu_decode(dq)[small] != 0).float().mean()) > 0.05
    assert _rel(hip.dgelu_decode(dq), uref.grad) < 6e-3
    # bf16 rounding is the only error: the erfc form is accurate to 1.5e-7
    assert (act.float() - aref.detach()).abs().max() < 2.0 ** -8 * max(1.0, aref.abs().max().item())
    z = hip.gemm_nt(a, b, hip.EPI_BIAS_RESID, bias=bias, aux=aux)
    assert _rel(z, base + bias + aux.float()) < 4e-3
    t = hip.gemm_nt(a, b, hip.EPI_BIAS_TANH, bias=bias)
    assert _rel(t, torch.tanh(base + bias)) < 4e-3
    ad = hip.gemm_nt(a, b, hip.EPI_ADD, aux=aux)
    assert _rel(ad, base + aux.float()) < 4e-3
    ad0 = hip.gemm_nt(a, b, hip.EPI_ADD)
    assert _rel(ad0, base) < 4e-3
    # gelu backward epilogue: out = acc * aux (aux = the gelu'(u) the forward epilogue saved), colsum
    vec = torch.zeros(N, device=dev)
    gq = hip.dgelu_encode(torch.rand(M, N, generator=g) * 1.25 - 0.125).to(dev)
    du = hip.gemm_nt(a, b, hip.EPI_GELU_BWD, aux=gq, vec_out=vec)
    assert _rel(du, base * hip.dgelu_decode(gq)) < 4e-3
    assert _rel(vec, (base * hip.dgelu_decode(gq)).sum(0)) < 1e-3
    # exact end points of the stash: 0 and 1 survive the round trip
    assert torch.equal(hip.dgelu_decode(hip.dgelu_encode(torch.tensor([0.0, 1.0]))), torch.tensor([0.0, 1.0]))
    # ABI 5: the same pair with the stash in bf16 (config.gelu_stash = "bf16"): same gelu(u), derivative to bf16 rounding
    d16, act16 = hip.gemm_nt(a, b, hip.EPI_BIAS_GELU_BF16, bias=bias)
    assert d16.dtype == torch.bfloat16 and torch.equal(act16, act)
    assert _rel(d16, uref.grad) < 3e-3 and (d16.float() - uref.grad).abs().max() < 2.0 ** -8 * 1.2
    g16 = _bf(torch.rand(M, N, generator=g) * 1.25 - 0.125).to(dev)
    vec16 = torch.zeros(N, device=dev)
    du16 = hip.gemm_nt(a, b, hip.EPI_GELU_BWD_BF16, aux=g16, vec_out=vec16)
    assert _rel(du16, base * g16.float()) < 4e-3 and _rel(vec16, (base * g16.float()).sum(0)) < 1e-3
    dyh = _bf(torch.randn(M, N, generator=g)).to(dev)
    assert _rel(hip.dgelu_mul(dyh, g16), dyh.float() * g16.float()) < 3e-3
    assert _rel(hip.dgelu_mul(dyh, gq), dyh.float() * hip.dgelu_decode(gq)) < 3e-3


def test_gelu_epilogue_function_values(dev):
    """Round 6: the GELU pair of the FFN1 epilogue is a degree-11 minimax polynomial for Phi - 1/2 on |x| <= 4.9 with ONE exp
    (common.h gelu_pair) instead of A&S 7.1.26 (exp + rcp).  A zero product + a bias that sweeps [-9, 9] puts known pre-activations
    through the epilogue: gelu(x) and gelu'(x) against float64 erf (modeling_bert.py:142-148) to far below the output's bf16
    rounding, and the exact tails the `clamp` bit of the last v_pk_fma_f32 provides (Phi = 0 / 1 beyond |x| = 4.9)."""
    import math
    from mvp_pytorch_amd import hip
    N, K, M = 8192, 64, 32
    x = torch.linspace(-9.0, 9.0, N, dtype=torch.float64)
    x[N // 2] = 0.0
    bias = x.float().to(dev)
    a = torch.zeros(M, K, dtype=torch.bfloat16, device=dev)
    b = torch.zeros(N, K, dtype=torch.bfloat16, device=dev)
    d16, act = hip.gemm_nt(a, b, hip.EPI_BIAS_GELU_BF16, bias=bias)
    xf = bias.double().cpu()
    cdf = 0.5 * (1.0 + torch.erf(xf / math.sqrt(2.0)))
    want = xf * cdf
    dwant = cdf + xf * torch.exp(-0.5 * xf * xf) / math.sqrt(2.0 * math.pi)
    got, dgot = act[0].double().cpu(), d16[0].double().cpu()
    assert torch.equal(act, act[0:1].expand_as(act)) and torch.equal(d16, d16[0:1].expand_as(d16))
    # function error (1e-6 |x| at most) + half a bf16 ulp of the result (2^-8 relative just above a power of two)
    assert ((got - want).abs() <= 2.0 ** -8 * want.abs() + 2e-6 * xf.abs() + 1e-9).all(), float((got - want).abs().max())
    assert ((dgot - dwant).abs() <= 2.0 ** -8 * dwant.abs() + 2e-5).all(), float((dgot - dwant).abs().max())
    hi, lo = xf >= 4.9001, xf <= -4.9001
    assert torch.equal(got[hi], xf[hi].float().to(torch.bfloat16).double())       # Phi clamps to exactly 1 ...
    assert (got[lo] == 0).all()                                                   # ... and exactly 0
    assert (dgot[hi] - 1.0).abs().max() < 4e-3 and dgot[lo].abs().max() < 2e-5
    assert float(got[N // 2]) == 0.0 and abs(float(dgot[N // 2]) - 0.5) < 1e-6
    # the 8-bit stash of the same derivative: one grid step
    dq, act8 = hip.gemm_nt(a, b, hip.EPI_BIAS_GELU, bias=bias)
    assert torch.equal(act8, act)
    assert (hip.dgelu_decode(dq)[0].double().cpu() - dwant).abs().max() < 0.005 + 2e-5


@pytest.mark.parametrize("M,N,K", [(16500, 768, 768), (16700, 2304, 768), (17000, 3072, 768), (16641, 768, 3072), (16900, 768, 2304),
                                   (70000, 768, 768), (33000, 3072, 256), (37748, 768, 2304), (10917, 768, 768),
                                   (11143, 3072, 768), (10917, 768, 3072)])
def test_gemm_nt_encoder_shapes_every_epilogue(dev, M, N, K):
    """The encoder-layer GEMM shapes at step-sized row counts (more than 64 tiles of 256 x 256, rows not a multiple of the
    tile): every epilogue against f32 torch, the 8-bit gelu' stash included, twice in a row.  Round 4: the 10917 / 11143-row
    cases (the text and visual stacks of the timed batch) take the 192-row tiles of launch()'s tile-height rule, the others
    256-row ones, 70000 x 768 x 768 the 256 x 128 configuration."""
    from mvp_pytorch_amd import hip
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    a = _bf(torch.randn(M, K, generator=g)).to(dev)
    b = _bf(torch.randn(N, K, generator=g) * 0.05).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    aux = _bf(torch.randn(M, N, generator=g)).to(dev)
    base = a.float() @ b.float().t()
    for rep in range(2):
        out = hip.gemm_nt(a, b, hip.EPI_BIAS, bias=bias)
        assert _rel(out, base + bias) < 4e-3
        # exact layout check on a slice: every element within bf16 rounding of the f32 product
        ref = base + bias
        assert ((out.float() - ref).abs() <= 2.0 ** -7 * ref.abs() + 1e-2).all()
        dq, act = hip.gemm_nt(a, b, hip.EPI_BIAS_GELU, bias=bias)
        uref = (base + bias).requires_grad_(True)
        aref = torch.nn.functional.gelu(uref)
        aref.sum().backward()
        assert _rel(act, aref.detach()) < 4e-3
        assert (hip.dgelu_decode(dq) - uref.grad).abs().max() < 0.005 + 1e-4
        z = hip.gemm_nt(a, b, hip.EPI_BIAS_RESID, bias=bias, aux=aux)
        assert _rel(z, base + bias + aux.float()) < 4e-3
        ad = hip.gemm_nt(a, b, hip.EPI_ADD, aux=aux)
        assert _rel(ad, base + aux.float()) < 4e-3
        ad0 = hip.gemm_nt(a, b, hip.EPI_ADD)
        assert _rel(ad0, base) < 4e-3
        vec = torch.zeros(N, device=dev)
        gq = hip.dgelu_encode(torch.rand(M, N, generator=g) * 1.25 - 0.125).to(dev)
        du = hip.gemm_nt(a, b, hip.EPI_GELU_BWD, aux=gq, vec_out=vec)
        assert _rel(du, base * hip.dgelu_decode(gq)) < 4e-3
        assert _rel(vec, (base * hip.dgelu_decode(gq)).sum(0)) < 1e-3
    # dropout of the residual epilogue reproduces the documented mask
    drop = hip.make_dropout(0.1, 0x1234567890 + M)
    zero = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
    zd = hip.gemm_nt(a, b, hip.EPI_BIAS_RESID, aux=zero, drop=drop)
    keep = hip.dropout_mask(drop, M * N, dev).reshape(M, N).float()
    assert _rel(zd, base * keep * (65536.0 / (65536.0 - drop.thresh16))) < 4e-3


def test_gemm_nt_identity_layout_many_tiles(dev):
    """A = I (tiled) against an asymmetric integer B over 210 tiles: every output element where it belongs, exactly."""
    from mvp_pytorch_amd import hip
    K, N, M = 256, 768, 256 * 70
    a = torch.eye(K, dtype=torch.bfloat16, device=dev).repeat(M // K, 1)
    b = ((torch.arange(N * K, device=dev, dtype=torch.float32).reshape(N, K) * 7) % 251 - 125).to(torch.bfloat16)
    out = hip.gemm_nt(a, b, hip.EPI_BIAS)
    assert torch.equal(out.float(), b.float().t().contiguous().repeat(M // K, 1))


def test_gemm_nt_dropout_matches_mask(dev):
    from mvp_pytorch_amd import hip
    g = torch.Generator(device="cpu").manual_seed(3)
    M, N, K = 128, 256, 64
    a = _bf(torch.randn(M, K, generator=g)).to(dev)
    b = _bf(torch.randn(N, K, generator=g)).to(dev)
    aux = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
    drop = hip.make_dropout(0.1, 0x1234567890)
    z = hip.gemm_nt(a, b, hip.EPI_BIAS_RESID, aux=aux, drop=drop)
    keep = hip.dropout_mask(drop, M * N, dev).reshape(M, N).float()
    scale = 65536.0 / (65536.0 - drop.thresh16)
    ref = (a.float() @ b.float().t()) * keep * scale
    assert _rel(z, ref) < 4e-3
    frac = 1.0 - keep.mean().item()
    assert abs(frac - 0.1) < 0.01


@pytest.mark.parametrize("M,N,K", [(1024, 768, 768), (700, 256, 3072), (333, 2304, 768), (64, 128, 128), (515, 1000, 136),
                                   (6400, 2304, 768), (7000, 1000, 136), (20000, 768, 3072)])
def test_gemm_tn(dev, M, N, K):
    """Below 6 000 token rows the planner picks a 256x128-tile configuration, from there on the 256x256 "Q" kernel."""
    from mvp_pytorch_amd import hip
    g = torch.Generator(device="cpu").manual_seed(4)
    ldn = (N + 7) // 8 * 8
    dy = torch.zeros(M, ldn, dtype=torch.bfloat16)
    dy[:, :N] = _bf(torch.randn(M, N, generator=g))
    dy = dy.to(dev)
    x = _bf(torch.randn(M, K, generator=g)).to(dev)
    dw = torch.zeros(N, K, device=dev)
    cs = torch.zeros(N, device=dev)
    hip.gemm_tn(dy, x, dw, n=N, colsum=cs)
    ref = dy[:, :N].float().t() @ x.float()
    err = _rel(dw, ref)
    print("gemm_tn", M, N, K, err)
    assert err < 1e-5
    assert _rel(cs, dy[:, :N].float().sum(0)) < 1e-5
    hip.gemm_tn(dy, x, dw, n=N)  # accumulates
    assert _rel(dw, 2 * ref) < 1e-5


@pytest.mark.parametrize("M", [700, 6400, 10917, 37748])
def test_gemm_tn_multi(dev, M):
    """Grouped launch: four problems sharing M (an encoder layer's weight gradients) and one with a
    different M (split into its own launch) give the same results as separate calls."""
    from mvp_pytorch_amd import hip
    g = torch.Generator(device="cpu").manual_seed(14)
    shapes = [(768, 3072), (3072, 768), (768, 768), (2304, 768)]
    probs, refs = [], []
    for N, K in shapes:
        dy = _bf(torch.randn(M, N, generator=g)).to(dev)
        x = _bf(torch.randn(M, K, generator=g)).to(dev)
        dw = torch.zeros(N, K, device=dev)
        cs = torch.zeros(N, device=dev) if N == 2304 else None
        probs.append((dy, x, dw, cs))
        refs.append(dy.float().t() @ x.float())
    dy5 = _bf(torch.randn(130, 100 + 4, generator=g)).to(dev)
    x5 = _bf(torch.randn(130, 72, generator=g)).to(dev)
    dw5 = torch.zeros(104, 72, device=dev)
    probs.append((dy5, x5, dw5, None))
    refs.append(dy5.float().t() @ x5.float())
    hip.gemm_tn_multi(probs)
    for (dy, x, dw, cs), ref in zip(probs, refs):
        assert _rel(dw, ref) < 1e-5
        if cs is not None:
            assert _rel(cs, dy.float().sum(0)) < 1e-5


@pytest.mark.parametrize("M", [6400, 10917, 20011])
def test_gemm_tn_slab_write_out_is_exact_and_reproducible(dev, M):
    """Few-row grouped launches (6 000 .. 24 000 rows) write per-split partial tiles into the caller's workspace and add
    them in split order (mvptr_gemm_tn_multi_ws): same values as the atomic write-out to f32 rounding, ACCUMULATED into
    dW like it, bitwise identical from run to run, and odd shapes (partial tiles at the N / K edge) included."""
    from mvp_pytorch_amd import hip
    g = torch.Generator(device="cpu").manual_seed(M)
    shapes = [(768, 768), (2304, 768), (520, 1000)]
    data = []
    for N, K in shapes:
        dy = _bf(torch.randn(M, N, generator=g)).to(dev)
        x = _bf(torch.randn(M, K, generator=g)).to(dev)
        data.append((dy, x, torch.randn(N, K, generator=g).to(dev)))
    outs = []
    for slab in (True, True, False):
        probs = [(dy, x, init.clone(), torch.zeros(dy.shape[1], device=dev)) for dy, x, init in data]
        arr_need = None
        if slab:
            arr = (hip.TnProblem * len(probs))()
            for q, (dy, x, dw, cs) in zip(arr, probs):
                q.A, q.lda, q.B, q.ldb, q.M, q.N, q.K = dy.data_ptr(), dy.stride(0), x.data_ptr(), x.stride(0), M, dy.shape[1], x.shape[1]
                q.dW, q.ldw, q.colsum = dw.data_ptr(), dw.stride(0), cs.data_ptr()
            arr_need = int(hip.load().mvptr_gemm_tn_ws_bytes(arr, len(probs)))
            assert arr_need > 0, "these row counts are in the slab range"
        hip.gemm_tn_multi(probs, slab_workspace=slab)
        torch.cuda.synchronize()
        outs.append([(dw, cs) for _, _, dw, cs in probs])
    # a workspace that is too small (or NULL) is not an error: the launch falls back to the atomic write-out
    probs = [(dy, x, init.clone(), None) for dy, x, init in data]
    arr = (hip.TnProblem * len(probs))()
    for q, (dy, x, dw, cs) in zip(arr, probs):
        q.A, q.lda, q.B, q.ldb, q.M, q.N, q.K = dy.data_ptr(), dy.stride(0), x.data_ptr(), x.stride(0), M, dy.shape[1], x.shape[1]
        q.dW, q.ldw, q.colsum = dw.data_ptr(), dw.stride(0), None
    small = torch.empty(4096, device=dev, dtype=torch.uint8)
    hip._check(hip.load().mvptr_gemm_tn_multi_ws(arr, len(probs), hip._p(small), small.numel(), hip._stream()))
    for (dy, x, init), (_, _, dw, _) in zip(data, probs):
        assert _rel(dw, init + dy.float().t() @ x.float()) < 1e-5
    for (dy, x, init), (dw_a, cs_a), (dw_b, _), (dw_c, cs_c) in zip(data, outs[0], outs[1], outs[2]):
        ref = init + dy.float().t() @ x.float()
        assert _rel(dw_a, ref) < 1e-5 and _rel(dw_c, ref) < 1e-5
        assert torch.equal(dw_a, dw_b)                 # fixed summation order
        assert _rel(cs_a, dy.float().sum(0)) < 1e-5 and _rel(cs_c, dy.float().sum(0)) < 1e-5


@pytest.mark.parametrize("M,layers,max_wg", [(700, 2, 0), (6400, 3, 0), (10917, 6, 0), (3001, 9, 0), (5000, 2, 40), (97, 1, 0), (33, 1, 7), (2, 1, 0)])
def test_gemm_tn_stack(dev, M, layers, max_wg):
    """One balanced launch for the weight gradients of a whole stack (mvptr_gemm_tn_stack): whole tiles per workgroup in
    the full rounds, equal runs of 32-row steps over the left-over tiles.  Checked against f32 products: odd shapes (partial
    tiles at the N / K edge), lists longer than MVPTR_TN_STACK_MAX (9 layers x 4 + 1 = 37 problems), accumulation into dW,
    column sums, a workgroup cap, and a device-side row count."""
    from mvp_pytorch_amd import hip
    g = torch.Generator(device="cpu").manual_seed(M + layers)
    shapes = [(768, 3072, False), (3072, 768, True), (768, 768, False), (2304, 768, True)]
    probs, refs = [], []
    for li in range(layers):
        for N, K, cs in shapes:
            dy = _bf(torch.randn(M, N, generator=g)).to(dev)
            x = _bf(torch.randn(M, K, generator=g)).to(dev)
            init = torch.randn(N, K, generator=g).to(dev)
            probs.append((dy, x, init.clone(), torch.zeros(N, device=dev) if cs else None))
            refs.append(init)
    dy5 = torch.zeros(M, 528, dtype=torch.bfloat16)
    dy5[:, :520] = _bf(torch.randn(M, 520, generator=g))
    dy5 = dy5.to(dev)[:, :520]            # lda = 528: row padding behind the 520 columns
    x5 = _bf(torch.randn(M, 1000, generator=g)).to(dev)
    probs.append((dy5, x5, torch.zeros(520, 1000, device=dev), torch.zeros(520, device=dev)))
    refs.append(torch.zeros(520, 1000, device=dev))
    hip.gemm_tn_stack(probs, max_workgroups=max_wg)
    for (dy, x, dw, cs), init in zip(probs, refs):
        assert _rel(dw, init + dy.float().t() @ x.float()) < 1e-5
        if cs is not None:
            assert _rel(cs, dy.float().sum(0)) < 1e-5
    # device-side row count: only the first Mv rows take part
    Mv = max(1, M - 333)
    rd = torch.tensor([Mv], device=dev, dtype=torch.int32)
    probs2 = [(dy, x, torch.zeros_like(dw), torch.zeros_like(cs) if cs is not None else None) for dy, x, dw, cs in probs]
    hip.gemm_tn_stack(probs2, rows_dev=rd, max_workgroups=max_wg)
    for dy, x, dw, cs in probs2:
        assert _rel(dw, dy[:Mv].float().t() @ x[:Mv].float()) < 1e-5
        if cs is not None:
            assert _rel(cs, dy[:Mv].float().sum(0)) < 1e-5
    # problems of one call share M
    bad = probs[:1] + [(dy5[:M - 1], x5[:M - 1], torch.zeros(520, 1000, device=dev), None)]
    with pytest.raises(RuntimeError, match="share M"):
        hip.gemm_tn_stack(bad)


def test_gemm_tn_layout_exact(dev):
    from mvp_pytorch_amd import hip
    M, N, K = 64, 128, 128
    dy = torch.zeros(M, N)
    dy[torch.arange(M), torch.arange(M)] = 1.0  # dy[m, n=m] = 1 -> dW[n] = x[n] for n < M
    x = (torch.arange(M * K).reshape(M, K) % 251 - 125).float()
    dw = torch.zeros(N, K, device=dev)
    hip.gemm_tn(_bf(dy).to(dev), _bf(x).to(dev), dw)
    ref = torch.zeros(N, K)
    ref[:M] = x
    assert torch.equal(dw.cpu(), ref)


def _attn_ref(qkv, mask, B, L, heads, keep=None, scale=1.0):
    H = heads * 64
    q, k, v = qkv.float().reshape(B, L, 3, heads, 64).permute(2, 0, 3, 1, 4)
    s = q @ k.transpose(-1, -2) / 8.0 + mask[:, None, None, :]
    p = torch.softmax(s, -1)
    if keep is not None:
        p = p * keep * scale
    ctx = (p @ v).permute(0, 2, 1, 3).reshape(B * L, H)
    return ctx, torch.logsumexp(s, -1)


@pytest.mark.parametrize("L", [5, 64, 70, 125, 256])
def test_attention_probs_tensor(dev, L):
    """mvptr_attention_probs (config.output_attentions, vl:85,100): softmax(QK^T/8 + mask) of the bf16 rows in f32, and
    consistent with the fused kernel: probs @ V is its context output."""
    g = torch.Generator().manual_seed(L)
    B, heads = 3, 12
    H = heads * 64
    qkv = (torch.randn(B * L, 3 * H, generator=g) * 0.8).to(torch.bfloat16).to(dev)
    lens = torch.tensor([L, max(1, L // 2), max(1, L - 3)])
    mask = ((torch.arange(L)[None, :] >= lens[:, None]).float() * -10000.0).to(dev)
    from mvp_pytorch_amd import hip
    probs = hip.attention_probs(qkv, mask, B, L, heads)
    q, k, v = (t.float().view(B, L, heads, 64).permute(0, 2, 1, 3) for t in qkv.split(H, dim=1))
    ref = torch.softmax(q @ k.transpose(-1, -2) * 0.125 + mask[:, None, None, :], dim=-1)
    assert probs.shape == (B, heads, L, L)
    assert float((probs - ref).abs().max()) < 1e-5
    ctx, _ = hip.attention_fwd(qkv, mask, B, L, heads)
    mine = (probs @ v).permute(0, 2, 1, 3).reshape(B * L, H)
    assert float((mine - ctx.float()).abs().max()) < 2e-2
    with pytest.raises(RuntimeError):
        hip.attention_probs(qkv, None, B, L, heads)


@pytest.mark.parametrize("B,L,heads", [(2, 32, 2), (3, 45, 2), (2, 125, 12), (2, 70, 12), (1, 193, 4), (2, 256, 2), (2, 5, 1)])
def test_attention_fwd_bwd(dev, B, L, heads):
    from mvp_pytorch_amd import hip
    g = torch.Generator(device="cpu").manual_seed(5)
    H = heads * 64
    qkv = _bf(torch.randn(B * L, 3 * H, generator=g)).to(dev)
    mask = torch.zeros(B, L)
    for b in range(B):
        nvalid = max(1, L - 3 * b - (L // 4) * (b % 2))
        mask[b, nvalid:] = -10000.0
    mask = mask.to(dev)
    ctx, lse = hip.attention_fwd(qkv, mask, B, L, heads)
    qr = qkv.float().clone().requires_grad_(True)
    ref, lse_ref = _attn_ref(qr, mask, B, L, heads)
    e1 = _rel(ctx, ref)
    e2 = (lse - lse_ref).abs().max().item()
    print("attn fwd", B, L, heads, e1, e2)
    assert e1 < 6e-3 and e2 < 2e-3
    dctx = _bf(torch.randn(B * L, H, generator=g)).to(dev)
    ref.backward(dctx.float())
    dqkv = hip.attention_bwd(qkv, mask, ctx, dctx, lse, B, L, heads)
    d = dqkv.float().reshape(B * L, 3, H)
    r = qr.grad.reshape(B * L, 3, H)
    errs = [_rel(d[:, i], r[:, i]) for i in range(3)]
    print("attn bwd dq,dk,dv", errs)
    assert max(errs) < 1.5e-2


def test_attention_dropout(dev):
    from mvp_pytorch_amd import hip
    g = torch.Generator(device="cpu").manual_seed(6)
    B, L, heads = 2, 75, 2
    H = heads * 64
    qkv = _bf(torch.randn(B * L, 3 * H, generator=g)).to(dev)
    mask = torch.zeros(B, L, device=dev)
    drop = hip.make_dropout(0.1, 987654321)
    Lp = (L + 31) // 32 * 32  # element index = ((b*heads+h)*L + q)*Lp + key
    keep = hip.dropout_mask(drop, B * heads * L * Lp, dev).reshape(B, heads, L, Lp)[..., :L].float()
    scale = 65536.0 / (65536.0 - drop.thresh16)
    ctx, lse = hip.attention_fwd(qkv, mask, B, L, heads, drop=drop)
    qr = qkv.float().clone().requires_grad_(True)
    ref, _ = _attn_ref(qr, mask, B, L, heads, keep, scale)
    assert _rel(ctx, ref) < 6e-3
    dctx = _bf(torch.randn(B * L, H, generator=g)).to(dev)
    ref.backward(dctx.float())
    dqkv = hip.attention_bwd(qkv, mask, ctx, dctx, lse, B, L, heads, drop=drop)
    assert _rel(dqkv, qr.grad) < 1.5e-2


@pytest.mark.parametrize("B,Lmax,heads", [(5, 125, 12), (7, 70, 2), (3, 256, 1), (4, 33, 2)])
def test_attention_packed_equals_dense_on_valid_rows(dev, B, Lmax, heads):
    """Row-packed (unpadded) attention: ragged lengths incl. 1 and the maximum, forward + backward equal
    the dense call whose padded keys carry the reference's -10000 mask; with dropout the packed call
    reproduces the mask of mvptr_dropout_mask at its documented element index."""
    from mvp_pytorch_amd import hip
    g = torch.Generator(device="cpu").manual_seed(50 + B)
    H = heads * 64
    lens = torch.randint(1, Lmax + 1, (B,), generator=g)
    lens[0], lens[-1] = Lmax, 1
    qkv_d = _bf(torch.randn(B * Lmax, 3 * H, generator=g)).to(dev)
    dctx_d = _bf(torch.randn(B * Lmax, H, generator=g)).to(dev)
    valid = (torch.arange(Lmax)[None, :] < lens[:, None])
    mask = ((~valid).float() * -10000.0).to(dev)
    idx = torch.nonzero(valid.reshape(-1)).reshape(-1).to(dev)
    starts = (torch.cumsum(lens, 0) - lens).to(torch.int32).to(dev)
    lens_d = lens.to(torch.int32).to(dev)
    ctx_d, lse_d = hip.attention_fwd(qkv_d, mask, B, Lmax, heads)
    dq_d = hip.attention_bwd(qkv_d, mask, ctx_d, dctx_d, lse_d, B, Lmax, heads)
    qkv_p, dctx_p = qkv_d.index_select(0, idx).contiguous(), dctx_d.index_select(0, idx).contiguous()
    ctx_p, lse_p = hip.attention_fwd_packed(qkv_p, starts, lens_d, B, Lmax, heads)
    dq_p = hip.attention_bwd_packed(qkv_p, starts, lens_d, ctx_p, dctx_p, lse_p, B, Lmax, heads)
    assert torch.equal(ctx_p, ctx_d.index_select(0, idx))
    # the dense backward also spends gradient on nothing but the valid rows (dctx of padded queries is
    # not zero in this test, so compare after removing their contribution: run dense with it zeroed)
    dctx_z = torch.zeros_like(dctx_d).index_copy(0, idx, dctx_p)
    dq_dz = hip.attention_bwd(qkv_d, mask, ctx_d, dctx_z, lse_d, B, Lmax, heads)
    assert _rel(dq_p, dq_dz.index_select(0, idx)) < 1e-6 and torch.allclose(dq_p.float(), dq_dz.index_select(0, idx).float(), atol=2e-2)
    v = valid.to(dev)
    assert torch.allclose(lse_p[v[:, None, :].expand(-1, heads, -1)], lse_d[v[:, None, :].expand(-1, heads, -1)])
    # dropout in packed mode: element index ((b*heads + h)*Lmax + q)*Lp + key
    drop = hip.make_dropout(0.2, 4242)
    Lp = (Lmax + 31) // 32 * 32
    keep = hip.dropout_mask(drop, B * heads * Lmax * Lp, dev).reshape(B, heads, Lmax, Lp)[:, :, :, :Lmax].float()
    ctx_pd, lse_pd = hip.attention_fwd_packed(qkv_p, starts, lens_d, B, Lmax, heads, drop=drop)
    ref, _ = _attn_ref(qkv_d.cpu(), mask.cpu(), B, Lmax, heads, keep=keep.cpu(), scale=65536.0 / (65536.0 - drop.thresh16))
    assert _rel(ctx_pd.cpu(), ref.index_select(0, idx.cpu())) < 1.5e-2


@pytest.mark.parametrize("heads,Lmax", [(12, 96), (12, 128), (6, 75), (2, 64), (3, 40)])
def test_attention_every_block_count_per_head(dev, heads, Lmax):
    """Round 4 (outputs leave through an LDS transpose as whole 128-byte rows; delta from 8 lanes per row): lengths on both sides
    of every 32-row block boundary, every (sequence, head) slice on its own against the fp32 torch reference, forward +
    backward, with and without dropout (keep mask index ((b*heads + h)*Lmax + q)*Lp + key)."""
    from mvp_pytorch_amd import hip
    g = torch.Generator(device="cpu").manual_seed(77 + heads)
    H = heads * 64
    lens = torch.tensor([n for n in (1, 17, 32, 33, 50, 64, 65, 96, 8, 31, 128, 97) if n <= Lmax] + [Lmax])
    B = lens.numel()
    qkv_d = _bf(torch.randn(B * Lmax, 3 * H, generator=g))
    dctx_d = _bf(torch.randn(B * Lmax, H, generator=g))
    valid = (torch.arange(Lmax)[None, :] < lens[:, None])
    mask = (~valid).float() * -10000.0
    idx = torch.nonzero(valid.reshape(-1)).reshape(-1)
    starts = (torch.cumsum(lens, 0) - lens).to(torch.int32).to(dev)
    lens_d = lens.to(torch.int32).to(dev)
    qkv_p, dctx_p = qkv_d.index_select(0, idx).contiguous().to(dev), dctx_d.index_select(0, idx).contiguous().to(dev)
    for pdrop in (0.0, 0.15):
        drop = hip.make_dropout(pdrop, 99) if pdrop > 0 else None
        keep, scale = None, 1.0
        if drop is not None:
            Lp = (Lmax + 31) // 32 * 32
            keep = hip.dropout_mask(drop, B * heads * Lmax * Lp, dev).reshape(B, heads, Lmax, Lp)[:, :, :, :Lmax].float().cpu()
            scale = 65536.0 / (65536.0 - drop.thresh16)
        ctx_p, lse_p = hip.attention_fwd_packed(qkv_p, starts, lens_d, B, Lmax, heads, drop=drop)
        dq_p = hip.attention_bwd_packed(qkv_p, starts, lens_d, ctx_p, dctx_p, lse_p, B, Lmax, heads, drop=drop)
        qr = qkv_d.float().clone().requires_grad_(True)
        ref, lse_ref = _attn_ref(qr, mask, B, Lmax, heads, keep, scale)
        (ref * dctx_d.float() * valid.reshape(-1, 1).float()).sum().backward()
        assert _rel(ctx_p.cpu(), ref.detach().index_select(0, idx)) < 6e-3
        vv = valid[:, None, :].expand(-1, heads, -1)
        assert torch.allclose(lse_p.cpu()[vv], lse_ref[vv], atol=2e-3, rtol=1e-4)
        assert _rel(dq_p.cpu(), qr.grad.index_select(0, idx)) < 1.5e-2
        # per sequence too: a wrong head or tile offset inside ONE group must not hide in the aggregate
        for b in range(B):
            sl = slice(int(starts[b]), int(starts[b]) + int(lens[b]))
            rows = idx[sl]
            for h in range(heads):
                cs = slice(h * 64, h * 64 + 64)
                assert _rel(ctx_p[sl, cs].cpu(), ref.detach()[rows][:, cs]) < 2e-2, (b, h)
                for part in range(3):
                    cg = slice(part * H + h * 64, part * H + h * 64 + 64)
                    # dQ, dK of a one-row sequence are exactly zero in fp32; the kernel's delta is taken from the bf16 forward
                    # output, which leaves |dS| ~ 2^-9 |dO.V| (|dO.V| ~ 8 here): an absolute floor of that size per 64-element slice
                    want = qr.grad[rows][:, cg]
                    assert (dq_p[sl, cg].cpu().float() - want).norm() < 4e-2 * want.norm() + 0.3, (b, h, part)


@pytest.mark.parametrize("rows,rows2,H,ns", [(900, 700, 768, (300, 1200, 64, 5)), (64, 0, 128, (40, 200)), (5000, 3000, 768, (9000, 1)),
                                            (33, 0, 2048, (10,)), (700, 41, 260, (1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 900))])
def test_tap_rows_bwd_against_index_add(dev, rows, rows2, H, ns):
    """mvptr_tap_rows_bwd (the backward of engine.MultiTapFn): every destination row = the f32 sum of the rows that tap it,
    rounded once; bf16 and f32 taps, -1 entries, repeated rows inside a tap and across taps, one row tapped by more than 64
    entries, untapped rows (must come out as zero rows from uninitialised memory), one and two destinations; twice in a row
    with the same result bit for bit wherever a row has at most 64 contributions."""
    from mvp_pytorch_amd import hip
    g = torch.Generator(device="cpu").manual_seed(rows + H)
    R = rows + rows2
    taps, ref = [], torch.zeros(R, H, dtype=torch.float64)
    for k, n in enumerate(ns):
        idx = torch.randint(-1, R, (n,), generator=g, dtype=torch.int32)
        if k == 1 and n >= 200:
            idx[:150] = 7                      # one hot row: more than 64 contributions
        if k == 0:
            idx[: n // 4] = idx[n // 4: 2 * (n // 4)]     # repeats inside a tap
        gr = torch.randn(n, H, generator=g)
        gr = gr if (k % 2) else _bf(gr)        # odd taps f32, even taps bf16
        ok = idx >= 0
        ref.index_add_(0, idx[ok].long(), gr[ok].double())
        taps.append(((gr if (k % 2) else gr.to(torch.bfloat16)).to(dev), idx.to(dev)))
    # poison the allocator's next blocks so that "untapped rows are zero" is a statement about the kernel
    junk = torch.full((R + 8, H), float("nan"), dtype=torch.bfloat16, device=dev)
    del junk
    d, d2 = hip.tap_rows_bwd(taps, rows, rows2, H)
    got = torch.cat([d, d2], 0) if rows2 else d
    assert (d2 is None) == (rows2 == 0)
    assert torch.isfinite(got.float()).all()
    want = ref.float()
    assert ((got.float().cpu() - want).abs() <= 2.0 ** -8 * want.abs() + 1e-6).all()
    untouched = torch.ones(R, dtype=torch.bool)
    for _, idx in taps:
        ic = idx.cpu()
        untouched[ic[ic >= 0].long()] = False
    assert untouched.any() and (got[untouched.to(dev)] == 0).all()
    again, again2 = hip.tap_rows_bwd(taps, rows, rows2, H)
    got2 = torch.cat([again, again2], 0) if rows2 else again
    cnt = torch.zeros(R, dtype=torch.long)
    for _, idx in taps:
        ic = idx.cpu()
        cnt.index_add_(0, ic[ic >= 0].long(), torch.ones(int((ic >= 0).sum()), dtype=torch.long))
    few = (cnt <= 64).to(dev)
    assert torch.equal(got[few], got2[few])


def test_multi_tap_fn_gradients(dev):
    """engine.MultiTapFn end to end: three taps of a two-buffer source (one with -1 slots, two read through .float(): autograd
    hands their gradients back in the taps' own bf16) against torch.index_select autograd on the concatenated buffers."""
    from mvp_pytorch_amd import engine
    g = torch.Generator(device="cpu").manual_seed(11)
    a, b = _bf(torch.randn(500, 768, generator=g)), _bf(torch.randn(300, 768, generator=g))
    i1 = torch.randint(0, 800, (256,), generator=g, dtype=torch.int32)
    i2 = torch.randint(-1, 800, (1000,), generator=g, dtype=torch.int32)
    i3 = torch.arange(0, 800, 7, dtype=torch.int32)
    w = [torch.randn(n, 768, generator=g) for n in (256, 1000, i3.numel())]
    ad, bd = a.to(dev).requires_grad_(True), b.to(dev).requires_grad_(True)
    outs = engine.MultiTapFn.apply(ad, bd, i1.to(dev), i2.to(dev), i3.to(dev))
    loss = (outs[0].float() * w[0].to(dev)).sum() + (outs[1] * _bf(w[1]).to(dev)).sum() + (outs[2].float() * w[2].to(dev)).sum()
    loss.backward()
    ref = torch.cat([a, b], 0).double().requires_grad_(True)
    z = torch.zeros(1, 768, dtype=torch.double)
    pick = lambda i: torch.cat([ref, z], 0).index_select(0, torch.where(i >= 0, i, torch.full_like(i, 800)).long())   # noqa: E731
    (pick(i1) * _bf(w[0]).double()).sum().add((pick(i2) * _bf(w[1]).double()).sum()).add((pick(i3) * _bf(w[2]).double()).sum()).backward()
    got = torch.cat([ad.grad, bd.grad], 0).float().cpu()
    assert ad.grad.dtype == torch.bfloat16 and _rel(got, ref.grad.float()) < 3e-3
    assert ((got - ref.grad.float()).abs() <= 2.0 ** -7 * ref.grad.float().abs() + 1e-5).all()


def test_dropout_kernels_follow_the_salt_word(dev):
    """ABI 7 (mvptr_set_dropout_salt): dropout seeds are launch arguments, so a replayed HIP graph would repeat its masks; every
    dropout-applying kernel mixes in a device-side salt word first.  With the word at 7 the mask helper gives another mask of the
    same rate, the GEMM residual epilogue, the attention kernels (forward + both backward kernels) and the LayerNorm kernels
    reproduce exactly that mask (their own tests, re-run under the salt), and the word back at 0 restores the documented masks."""
    from mvp_pytorch_amd import hip
    w = hip.dropout_salt(dev)
    w.zero_()                     # (a captured step run earlier in the process leaves its replay count in the word)
    drop = hip.make_dropout(0.1, 0xABCDEF12345)
    m0 = hip.dropout_mask(drop, 1 << 18, dev).clone()
    try:
        w.fill_(7)
        m7 = hip.dropout_mask(drop, 1 << 18, dev).clone()
        assert not torch.equal(m7, m0) and abs(float(m7.float().mean()) - 0.9) < 5e-3
        assert abs(float((m7 == m0).float().mean()) - 0.82) < 0.01          # independent masks agree on 0.9^2 + 0.1^2 of the elements
        test_gemm_nt_dropout_matches_mask(dev)
        test_attention_dropout(dev)
        test_layernorm_remap_and_dropout(dev, 768)
        test_layernorm_bwd_dropout_outputs(dev, 768)
        test_layernorm_bwd_dropout_outputs(dev, 128)
    finally:
        w.zero_()
    assert torch.equal(hip.dropout_mask(drop, 1 << 18, dev), m0)


def test_compact_scored_rows(dev):
    """mvptr_compact_scored against the chain it replaces (labels > -1 -> nonzero -> two index_selects), with a row map wider
    and taller than the label matrix (the packed joint map), without one, and with a shortfall of scored slots."""
    from mvp_pytorch_amd import hip
    g = torch.Generator(device="cpu").manual_seed(31)
    for B, L, ld, rows in ((256, 75, 125, 512), (7, 20, 20, 7), (1, 5, 9, 3), (300, 70, 70, 300), (600, 70, 72, 600)):      # the last: > 32 labels per thread (the loop path)
        labels = torch.randint(0, 30522, (B, L), generator=g)
        labels[torch.rand(B, L, generator=g) < 0.85] = -1
        pos = torch.randint(-1, 40000, (rows, ld), generator=g, dtype=torch.int32)
        keep = (labels > -1).reshape(-1)
        idx = torch.nonzero(keep).reshape(-1)
        want_l = labels.reshape(-1)[idx]
        want_r = pos[:B, :L].reshape(-1)[idx]
        n = idx.numel()
        ol, orow = hip.compact_scored(labels.to(dev), pos.to(dev), n)
        assert torch.equal(ol.cpu(), want_l) and torch.equal(orow.cpu(), want_r)
        ol, orow = hip.compact_scored(labels.to(dev), None, n)
        assert torch.equal(ol.cpu(), want_l) and torch.equal(orow.cpu(), idx.to(torch.int32))
        # fewer output slots than scored rows is a caller bug: since ABI 7 the kernel cuts the surplus, stays in bounds and
        # reports through the device error word; the host raises at its next read-back (ABI 5-6 trapped: the process died)
        if n > 3:
            ol, orow = hip.compact_scored(labels.to(dev), pos.to(dev), n - 3)
            assert torch.equal(ol.cpu(), want_l[:n - 3]) and torch.equal(orow.cpu(), want_r[:n - 3])
            with pytest.raises(RuntimeError, match=r"more scored .* \(%d scored rows, %d slots\)" % (n, n - 3)):
                hip.check_device_errors(dev)
            hip.check_device_errors(dev)          # the word was cleared by the raise
            from mvp_pytorch_amd import engine
            hip.compact_scored(labels.to(dev), pos.to(dev), n - 1)
            cnt = engine.AsyncCounts([torch.tensor(5, device=dev), torch.tensor(7, device=dev)])      # any count copy carries the word
            with pytest.raises(RuntimeError, match="more scored"):
                cnt.get()
            assert engine.AsyncCounts([torch.tensor(5, device=dev), torch.tensor(7, device=dev)]).get() == [5, 7]
        ol, orow = hip.compact_scored(labels.to(dev), pos.to(dev), n + 5)                # shortfall padded with -1 / -1
        assert torch.equal(ol.cpu()[:n], want_l) and (ol.cpu()[n:] == -1).all() and (orow.cpu()[n:] == -1).all()


def test_head_glue_kernels(dev):
    """mvptr_masked_mean (mean of the decoder-CE row losses over the rows with label >= 0) and mvptr_dgelu_mul (GELU backward
    of a head transform from the 8-bit stash, pad columns zeroed) against the torch expressions they replace."""
    from mvp_pytorch_amd import hip
    g = torch.Generator(device="cpu").manual_seed(21)
    for M in (1, 37, 2900, 5000):
        labels = torch.randint(-1, 30522, (M,), generator=g)
        labels[torch.rand(M, generator=g) < 0.4] = -1
        loss_row = torch.rand(M, generator=g) * (labels >= 0).float()
        loss, n = hip.masked_mean(loss_row.to(dev), labels.to(dev))
        want_n = max(int((labels >= 0).sum()), 1)
        assert float(n) == float(want_n)
        assert abs(float(loss) - float(loss_row.double().sum()) / want_n) < 1e-5 * max(1.0, float(loss_row.sum()) / want_n)
    none = torch.full((9,), -1, dtype=torch.int64)
    loss, n = hip.masked_mean(torch.zeros(9, device=dev), none.to(dev))
    assert float(loss) == 0.0 and float(n) == 1.0
    for M, N, Npad in ((2900, 768, 768), (33, 100, 104), (5, 8, 8)):
        dy = _bf(torch.randn(M, N, generator=g)).to(dev)
        gq = hip.dgelu_encode(torch.rand(M, N, generator=g) * 1.25 - 0.125).to(dev)
        out = hip.dgelu_mul(dy, gq, Npad)
        want = (dy.float() * hip.dgelu_decode(gq)).to(torch.bfloat16)
        assert out.shape == (M, Npad) and (out[:, N:] == 0).all() and torch.equal(out[:, :N], want)


@pytest.mark.parametrize("M,H", [(1000, 768), (77, 128), (5, 1024)])
def test_layernorm(dev, M, H):
    from mvp_pytorch_amd import hip
    g = torch.Generator(device="cpu").manual_seed(7)
    z = _bf(torch.randn(M, H, generator=g) * 2 + 0.5).to(dev)
    gamma = (1 + 0.1 * torch.randn(H, generator=g)).to(dev)
    beta = (0.1 * torch.randn(H, generator=g)).to(dev)
    y, mean, rstd = hip.layernorm_fwd(z, gamma, beta, 1e-12)
    zr = z.float().clone().requires_grad_(True)
    gr = gamma.clone().requires_grad_(True)
    br = beta.clone().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(zr, (H,), gr, br, 1e-12)
    assert _rel(y, ref) < 4e-3
    dy = _bf(torch.randn(M, H, generator=g)).to(dev)
    ref.backward(dy.float())
    dg = torch.zeros(H, device=dev)
    db = torch.zeros(H, device=dev)
    dbias = torch.zeros(H, device=dev)
    dz, dd = hip.layernorm_bwd(dy, z, mean, rstd, gamma, dg, db, dbias)
    assert dd is None
    assert _rel(dz, zr.grad) < 6e-3
    assert _rel(dg, gr.grad) < 1e-3 and _rel(db, br.grad) < 1e-3
    assert _rel(dbias, dz.float().sum(0)) < 5e-3


@pytest.mark.parametrize("H", [128, 768])
def test_layernorm_remap_and_dropout(dev, H):
    from mvp_pytorch_amd import hip
    g = torch.Generator(device="cpu").manual_seed(8)
    B, R, G = 3, 10, 6
    z = _bf(torch.randn(B * R, H, generator=g)).to(dev)
    gamma = torch.ones(H, device=dev)
    beta = torch.zeros(H, device=dev)
    out = torch.zeros(B * (G + R), H, dtype=torch.bfloat16, device=dev)
    drop = hip.make_dropout(0.25, 42)
    hip.layernorm_fwd(z, gamma, beta, 1e-12, out=out, rows_per_group=R, group_stride=G + R, row_offset=G, drop=drop)
    keep = hip.dropout_mask(drop, B * R * H, dev).reshape(B * R, H).float()
    scale = 65536.0 / (65536.0 - drop.thresh16)
    ref = torch.nn.functional.layer_norm(z.float(), (H,)) * keep * scale
    o = out.reshape(B, G + R, H)
    assert torch.all(o[:, :G] == 0)
    assert _rel(o[:, G:].reshape(B * R, H), ref) < 4e-3


@pytest.mark.parametrize("H", [128, 768])
def test_layernorm_bwd_dropout_outputs(dev, H):
    """Backward with (a) dropout on the LayerNorm output (embeddings: mb:276) regenerated from the
    counter hash, (b) the masked copy for the preceding dense layer (mb:350,409), (c) remapped
    gradient rows, (d) identity mode (gamma = NULL: dropout only)."""
    from mvp_pytorch_amd import hip
    g = torch.Generator(device="cpu").manual_seed(18)
    B, R, G = 5, 9, 4
    M = B * R
    z = _bf(torch.randn(M, H, generator=g) * 1.5).to(dev)
    gamma = (1 + 0.1 * torch.randn(H, generator=g)).to(dev)
    beta = (0.1 * torch.randn(H, generator=g)).to(dev)
    ydrop, ddrop = hip.make_dropout(0.2, 77), hip.make_dropout(0.1, 99)
    y, mean, rstd = hip.layernorm_fwd(z, gamma, beta, 1e-12)
    dy_full = _bf(torch.randn(B * (G + R), H, generator=g)).to(dev)       # gradient in the remapped layout
    dy_rows = dy_full.reshape(B, G + R, H)[:, G:].reshape(M, H)
    ky = hip.dropout_mask(ydrop, M * H, dev).reshape(M, H).float() * (65536.0 / (65536.0 - ydrop.thresh16))
    kd = hip.dropout_mask(ddrop, M * H, dev).reshape(M, H).float() * (65536.0 / (65536.0 - ddrop.thresh16))
    zr = z.float().clone().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(zr, (H,), gamma, beta, 1e-12)
    ref.backward(dy_rows.float() * ky)
    dg, db, dbias = (torch.zeros(H, device=dev) for _ in range(3))
    dz, dd = hip.layernorm_bwd(dy_full, z, mean, rstd, gamma, dg, db, dbias, rows_per_group=R, group_stride=G + R,
                               row_offset=G, y_drop=ydrop, dense_drop=ddrop)
    assert _rel(dz, zr.grad) < 6e-3
    assert _rel(dd, zr.grad * kd) < 6e-3
    assert _rel(dbias, (zr.grad * kd).sum(0)) < 5e-3
    assert _rel(db, (dy_rows.float() * ky).sum(0)) < 1e-3
    # identity mode: y = dropout(z), dz = dy * mask
    yi, _, _ = hip.layernorm_fwd(z, None, None, 0.0, drop=ydrop, save_stats=False)
    assert _rel(yi, z.float() * ky) < 4e-3
    dzi, _ = hip.layernorm_bwd(dy_rows.contiguous(), z, None, None, None, None, None, None, y_drop=ydrop)
    assert _rel(dzi, dy_rows.float() * ky) < 4e-3


def test_embed(dev):
    from mvp_pytorch_amd import hip
    g = torch.Generator(device="cpu").manual_seed(9)
    V, P, T, H, rows = 500, 64, 2, 128, 300
    word = torch.randn(V, H, generator=g).to(dev)
    pos = torch.randn(P, H, generator=g).to(dev)
    typ = torch.randn(T, H, generator=g).to(dev)
    ids = torch.randint(0, V, (rows,), generator=g).to(dev)
    ids[:20] = 0
    pids = torch.randint(0, P, (rows,), generator=g).to(dev)
    tids = torch.randint(0, T, (rows,), generator=g).to(dev)
    z = hip.embed_fwd(ids, pids, tids, word, pos, typ)
    ref = word[ids] + pos[pids] + typ[tids]
    assert _rel(z, ref) < 4e-3
    dz = _bf(torch.randn(rows, H, generator=g)).to(dev)
    dw = torch.zeros_like(word)
    dp = torch.zeros_like(pos)
    dt = torch.zeros_like(typ)
    hip.embed_bwd(ids, pids, tids, dz, dw, dp, dt)
    rw = torch.zeros_like(word).index_add_(0, ids, dz.float())
    rw[0] = 0  # padding_idx
    rp = torch.zeros_like(pos).index_add_(0, pids, dz.float())
    rt = torch.zeros_like(typ).index_add_(0, tids, dz.float())
    assert _rel(dw, rw) < 1e-5 and _rel(dp, rp) < 1e-5 and _rel(dt, rt) < 1e-5


def test_cast_pack(dev):
    from mvp_pytorch_amd import hip
    g = torch.Generator(device="cpu").manual_seed(10)
    src = torch.randn(70, 2054, generator=g).to(dev)
    dst = torch.full((70, 2056), 7.0, dtype=torch.bfloat16, device=dev)
    dst_t = torch.zeros(2054, 200, dtype=torch.bfloat16, device=dev)
    hip.cast_pack(src, dst=dst, dst_t=dst_t, col_off_t=100)
    assert torch.equal(dst[:, :2054], src.to(torch.bfloat16))
    assert torch.all(dst[:, 2054:] == 0)
    assert torch.equal(dst_t[:, 100:170], src.to(torch.bfloat16).t())
    assert torch.all(dst_t[:, :100] == 0) and torch.all(dst_t[:, 170:] == 0)
    assert torch.equal(hip.cast_f32(dst, cols=2054), src.to(torch.bfloat16).float())
    # row-major only (the region features): vectorised row cast, pad columns zeroed, odd column counts, unaligned rows
    for cols in (2054, 2053, 2049):
        d2 = torch.full((70, 2056), 7.0, dtype=torch.bfloat16, device=dev)
        hip.cast_pack(src[:, :cols], dst=d2)
        assert torch.equal(d2[:, :cols], src[:, :cols].to(torch.bfloat16)) and torch.all(d2[:, cols:] == 0)
    odd = torch.randn(33, 2055, generator=g).to(dev)          # 4-byte-aligned rows only
    d3 = torch.full((33, 2056), 7.0, dtype=torch.bfloat16, device=dev)
    hip.cast_pack(odd, dst=d3)
    assert torch.equal(d3[:, :2055], odd.to(torch.bfloat16)) and torch.all(d3[:, 2055:] == 0)


def test_cross_entropy(dev):
    from mvp_pytorch_amd import hip
    g = torch.Generator(device="cpu").manual_seed(11)
    M, V, Vpad = 50, 30522, 30528
    logits = torch.zeros(M, Vpad)
    logits[:, :V] = torch.randn(M, V, generator=g) * 3
    logits = logits.to(dev)
    labels = torch.randint(0, V, (M,), generator=g)
    labels[::7] = -1
    labels = labels.to(dev)
    loss, lse = hip.ce_fwd(logits, labels, V=V)
    lr = logits[:, :V].clone().requires_grad_(True)
    ref = torch.nn.functional.cross_entropy(lr, labels, ignore_index=-1, reduction="none")
    assert (loss - ref).abs().max().item() < 1e-4
    nvalid = (labels >= 0).sum()
    ref.sum().div(nvalid).backward()
    scale = (1.0 / nvalid.float()).reshape(1)
    d = hip.ce_bwd(logits, labels, lse, scale, V, Vpad)
    assert torch.all(d[:, V:] == 0)
    assert _rel(d[:, :V], lr.grad) < 4e-3


@pytest.mark.parametrize("M,V,K", [(50, 30522, 768), (3000, 30522, 768), (300, 1000, 136), (7, 37, 64), (513, 3129, 768)])
def test_fused_decoder_cross_entropy(dev, M, V, K):
    """mvptr_decoder_ce_fwd / _bwd (no logits tensor) against the materialising path and torch f32."""
    from mvp_pytorch_amd import hip
    g = torch.Generator(device="cpu").manual_seed(21)
    Vp = (V + 7) // 8 * 8
    h = _bf(torch.randn(M, K, generator=g)).to(dev)
    w = _bf(torch.randn(V, K, generator=g) * 0.08).to(dev)
    bias = torch.randn(V, generator=g).to(dev)
    labels = torch.randint(0, V, (M,), generator=g)
    labels[::5] = -1
    labels = labels.to(dev)
    loss, lse = hip.decoder_ce_fwd(h, w, bias, labels, V)
    logits = (h.float() @ w.float().t() + bias).requires_grad_(True)
    ref = torch.nn.functional.cross_entropy(logits, labels, ignore_index=-1, reduction="none")
    assert (loss - ref.detach()).abs().max().item() < 2e-3
    assert (lse - torch.logsumexp(logits.detach(), 1)).abs().max().item() < 2e-3
    assert torch.all(loss[labels < 0] == 0)
    nvalid = (labels >= 0).sum()
    ref.sum().div(nvalid).backward()
    scale = (1.0 / nvalid.float()).reshape(1)
    d = hip.decoder_ce_bwd(h, w, bias, labels, lse, scale, V, Vp)
    assert d.shape == (M, Vp)
    assert torch.all(d[:, V:] == 0)
    assert torch.all(d[labels < 0] == 0)
    assert _rel(d[:, :V], logits.grad) < 6e-3
    # the materialising kernels on f32 logits of the same GEMM give the same numbers up to bf16 rounding
    lg = torch.zeros(M, Vp, device=dev)
    hip.gemm_nt(h, w, hip.EPI_F32, bias=bias, out=lg, n=V)
    loss2, lse2 = hip.ce_fwd(lg, labels, V=V)
    assert (loss - loss2).abs().max().item() < 1e-4 and (lse - lse2).abs().max().item() < 1e-4
    d2 = hip.ce_bwd(lg, labels, lse2, scale, V, Vp)
    assert _rel(d, d2) < 1e-3


def test_fused_adamw_matches_oracle(dev):
    """mvptr_adamw_multi (through mvp_pytorch_amd.optimization.AdamW) against the oracle's
    restatement of optimization.py:131-187, three steps, two weight-decay groups, ragged sizes."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import mvptr_oracle as orc
    from mvp_pytorch_amd.optimization import AdamW
    g = torch.Generator(device="cpu").manual_seed(12)
    shapes = {"a.weight": (300, 257), "a.bias": (300,), "b.LayerNorm.weight": (131075,), "c.weight": (7,), "d": ()}
    ref = {k: torch.randn(s, generator=g) for k, s in shapes.items()}
    params = {k: torch.nn.Parameter(v.clone().to(dev)) for k, v in ref.items()}
    no_decay = ["bias", "LayerNorm.weight"]
    groups = [{"params": [p for n, p in params.items() if not any(x in n for x in no_decay)], "weight_decay": 0.01},
              {"params": [p for n, p in params.items() if any(x in n for x in no_decay)], "weight_decay": 0.0}]
    opt = AdamW(groups, lr=1e-2, eps=1e-8)
    state = {}
    for step in range(3):
        grads = {k: torch.randn(s, generator=g) for k, s in shapes.items()}
        for k, p in params.items():
            p.grad = grads[k].to(dev)
        v0 = params["a.weight"]._version
        opt.step()
        assert params["a.weight"]._version > v0
        orc.adamw_step(ref, grads, state, lr=1e-2, eps=1e-8, weight_decay=lambda n: 0.0 if any(x in n for x in no_decay) else 0.01)
    for k in shapes:
        assert torch.allclose(params[k].detach().cpu(), ref[k], rtol=2e-5, atol=1e-6), k


# ------------------------------------------------------------------------------ round-3 additions
def test_adamw_mirror_clip_matches_oracle(dev):
    """SURVEY §8 f1 in one pass: global-norm clip (mvptr_sumsq_partial + mvptr_clip_coef) -> AdamW that also writes
    the bf16 working copies (mvptr_adamw_mirror_multi: row-major with K padding, transposed at a column offset,
    packed f32 bias copy) against the oracle's clip_grad_norm + adamw_step and a plain cast of its result."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import mvptr_oracle as orc
    from mvp_pytorch_amd import engine, hip
    from mvp_pytorch_amd.optimization import AdamW
    g = torch.Generator(device="cpu").manual_seed(5)
    H = 64
    shapes = {"q.weight": (H, H), "k.weight": (H, H), "v.weight": (H, H), "q.bias": (H,), "k.bias": (H,), "v.bias": (H,),
              "img.weight": (72, 2054), "ffn.weight": (200, 136), "plain.weight": (33, 17), "plain.LayerNorm.weight": (70001,)}
    ref = {k: torch.randn(s, generator=g) for k, s in shapes.items()}
    params = {k: torch.nn.Parameter(v.clone().to(dev)) for k, v in ref.items()}
    bf = dict(device=dev, dtype=torch.bfloat16)
    w_qkv, w_qkv_t = torch.zeros((3 * H, H), **bf), torch.zeros((H, 3 * H), **bf)
    b_qkv = torch.zeros(3 * H, device=dev)
    img_w = torch.ones((72, 2056), **bf)
    ffn_w, ffn_t = torch.zeros((200, 136), **bf), torch.zeros((136, 200), **bf)
    cache = engine.WeightCache()
    for i, n in enumerate("qkv"):
        engine.register_mirror(params[n + ".weight"], cache, dst=w_qkv[i * H:(i + 1) * H], dst_t=w_qkv_t, col_off_t=i * H)
        engine.register_mirror(params[n + ".bias"], cache, dst_f32=b_qkv[i * H:(i + 1) * H])
    engine.register_mirror(params["img.weight"], cache, dst=img_w)
    engine.register_mirror(params["ffn.weight"], cache, dst=ffn_w, dst_t=ffn_t)
    cache.stale(list(params.values()), force=False)
    no_decay = ["bias", "LayerNorm.weight"]
    groups = [{"params": [p for n, p in params.items() if not any(x in n for x in no_decay)], "weight_decay": 0.01},
              {"params": [p for n, p in params.items() if any(x in n for x in no_decay)], "weight_decay": 0.0}]
    opt = AdamW(groups, lr=1e-2, eps=1e-8)
    total = sum(int(np.prod(s)) for s in shapes.values())
    flat = torch.zeros(total, device=dev)
    state = {}
    for step in range(3):
        grads = {k: torch.randn(s, generator=g) * (3.0 if step == 1 else 0.01) for k, s in shapes.items()}   # step 1 is clipped, 0 and 2 are not
        o = 0
        for k, p in params.items():
            n = p.numel()
            flat[o:o + n].copy_(grads[k].reshape(-1))
            p.grad = flat[o:o + n].view_as(p)
            o += n
        norm, coef, _ = hip.grad_clip_coef([flat[:70000], flat[70000:]], 10.0)
        opt.step(grad_scale=coef)
        ref_norm = orc.clip_grad_norm(grads, 10.0)
        assert abs(float(norm) - float(ref_norm)) < 1e-4 * float(ref_norm)
        assert (float(coef) < 1.0) == (step == 1)
        orc.adamw_step(ref, grads, state, lr=1e-2, eps=1e-8, weight_decay=lambda n: 0.0 if any(x in n for x in no_decay) else 0.01)
        for k in shapes:
            assert torch.allclose(params[k].detach().cpu(), ref[k], rtol=2e-5, atol=1e-6), (step, k)
        # working copies = the updated masters, rounded once
        for i, n in enumerate("qkv"):
            w = params[n + ".weight"].detach()
            assert torch.equal(w_qkv[i * H:(i + 1) * H], w.to(torch.bfloat16))
            assert torch.equal(w_qkv_t[:, i * H:(i + 1) * H], w.t().to(torch.bfloat16))
            assert torch.equal(b_qkv[i * H:(i + 1) * H], params[n + ".bias"].detach())
        assert torch.equal(img_w[:, :2054], params["img.weight"].detach().to(torch.bfloat16)) and float(img_w[:, 2054:].float().abs().sum()) == 0.0
        assert torch.equal(ffn_w, params["ffn.weight"].detach().to(torch.bfloat16))
        assert torch.equal(ffn_t, params["ffn.weight"].detach().t().to(torch.bfloat16))
        assert cache._fresh_once       # marked fresh after the version bumps
        assert cache.stale(list(params.values())) is False and cache.stale(list(params.values()), force=True) is True


@pytest.mark.parametrize("M,N,K", [(256, 768, 768), (37, 2, 768), (512, 256, 100), (5, 5, 5)])
def test_sgemm_small_variants(dev, M, N, K):
    """mvptr_sgemm_small against f32 torch: both transposes, bf16 operands, gathered A rows, bias + tanh, accumulate."""
    from mvp_pytorch_amd import hip
    g = torch.Generator(device="cpu").manual_seed(M * 7 + N)
    a = torch.randn(M, K, generator=g).to(dev)
    b = torch.randn(K, N, generator=g).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    ref = a.double() @ b.double()
    tol = 2e-5 * float(ref.abs().max())
    assert float((hip.sgemm_small(a, b).double() - ref).abs().max()) < tol
    assert float((hip.sgemm_small(a.t().contiguous(), b, trans_a=True).double() - ref).abs().max()) < tol
    assert float((hip.sgemm_small(a, b.t().contiguous(), trans_b=True).double() - ref).abs().max()) < tol
    got = hip.sgemm_small(a, b, bias=bias, act="tanh", alpha=0.05)
    assert float((got.double() - torch.tanh(0.05 * ref + bias.double())).abs().max()) < 2e-5
    # gathered bf16 rows out of a larger buffer (the [CLS] rows of a sequence buffer), accumulate into C
    big = torch.randn(3 * M + 1, K + 8, generator=g).to(dev).to(torch.bfloat16)
    rows = torch.arange(M, device=dev, dtype=torch.int32) * 3 + 1
    c0 = torch.randn(M, N, generator=g).to(dev)
    got = hip.sgemm_small(big[:, :K], b, a_rows=rows, out=c0.clone(), accumulate=True)
    ref2 = c0.double() + big[rows.long(), :K].double() @ b.double()
    assert float((got.double() - ref2).abs().max()) < 2e-5 * float(ref2.abs().max())
    # dW = X^T dY with X gathered (trans_a + a_rows)
    dy = torch.randn(M, N, generator=g).to(dev)
    got = hip.sgemm_small(big[:, :K], dy, trans_a=True, a_rows=rows)
    ref3 = big[rows.long(), :K].double().t() @ dy.double()
    assert got.shape == (K, N) and float((got.double() - ref3).abs().max()) < 2e-5 * float(ref3.abs().max())


def test_l2norm_and_clip_contrastive_loss(dev):
    """mvptr_l2norm_fwd/bwd and mvptr_clip_ce_fwd/bwd against torch autograd (vl:525-527, 1238-1241)."""
    import torch.nn.functional as F
    from mvp_pytorch_amd import hip
    g = torch.Generator(device="cpu").manual_seed(3)
    n, H = 37, 768
    y1 = torch.randn(n, H, generator=g).to(dev).requires_grad_(True)
    y2 = torch.randn(n, H, generator=g).to(dev).requires_grad_(True)
    ls = torch.tensor(float(np.log(1 / 0.07)), device=dev, requires_grad=True)
    gt, gi = F.normalize(y1, p=2, dim=-1), F.normalize(y2, p=2, dim=-1)
    sim = gt @ gi.t()
    logits = sim * ls.exp()
    pseudo = torch.arange(n, device=dev)
    loss = (F.cross_entropy(logits, pseudo) + F.cross_entropy(logits.t(), pseudo)) / 2
    (loss * 1.7).backward()
    g1, inv1 = hip.l2norm_fwd(y1.detach())
    g2, inv2 = hip.l2norm_fwd(y2.detach())
    assert torch.allclose(g1, gt.detach(), atol=1e-6) and torch.allclose(g2, gi.detach(), atol=1e-6)
    sim_h = hip.sgemm_small(g1, g2, trans_b=True)
    assert torch.allclose(sim_h, sim.detach(), atol=2e-6)
    loss_h, lse = hip.clip_ce_fwd(sim_h, ls.detach().reshape(1))
    assert abs(float(loss_h) - float(loss)) < 1e-5 * abs(float(loss))
    dls = torch.zeros(1, device=dev)
    dsim = hip.clip_ce_bwd(sim_h, ls.detach().reshape(1), lse, torch.tensor([1.7], device=dev), dls)
    assert abs(float(dls) - float(ls.grad)) < 1e-4 * abs(float(ls.grad)) + 1e-6
    dgt = hip.sgemm_small(dsim, g2)
    dgi = hip.sgemm_small(dsim, g1, trans_a=True)
    dy1, dy2 = hip.l2norm_bwd(g1, inv1, dgt), hip.l2norm_bwd(g2, inv2, dgi)
    for got, want in ((dy1, y1.grad), (dy2, y2.grad)):
        assert float((got - want).abs().max()) < 1e-4 * float(want.abs().max()) + 1e-8


def test_gather_and_scatter_add_rows(dev):
    from mvp_pytorch_amd import hip
    g = torch.Generator(device="cpu").manual_seed(9)
    src = torch.randn(50, 128, generator=g).to(dev).to(torch.bfloat16)
    idx = torch.tensor([3, 49, -1, 3, 0, 17], device=dev, dtype=torch.int32)
    out = hip.gather_rows(src, idx)
    ref = src[idx.clamp(min=0).long()].clone()
    ref[2] = 0
    assert torch.equal(out, ref)
    dst = torch.zeros(50, 128, device=dev, dtype=torch.bfloat16)
    upd = (torch.randint(-8, 8, (6, 128), generator=g).float() * 0.25).to(dev)   # exactly representable: order-independent sums
    hip.scatter_add_rows(upd.to(torch.bfloat16), idx, dst)
    hip.scatter_add_rows(upd, idx, dst)                                          # f32 source, rounded on the way
    want = torch.zeros(50, 128, device=dev)
    for i, d in enumerate(idx.tolist()):
        if d >= 0:
            want[d] += 2 * upd[i]
    assert torch.equal(dst.float(), want)
    dst32 = torch.zeros(50, 128, device=dev)
    hip.scatter_add_rows(upd.to(torch.bfloat16), idx, dst32)
    hip.scatter_add_rows(upd, idx, dst32)
    assert torch.equal(dst32, want)


def test_pack_maps_against_host_walk(dev):
    """mvptr_pack_maps: packed positions / source rows / starts / lengths of a one-segment pass (uni-modal stack, padded
    source, interior padding) and of a two-segment pass built on it (joint sequences = text rows + region rows of two packed
    outputs, hard negatives through `sel`), each against a plain host loop; > 1024 sequences exercises the chunked scan."""
    from mvp_pytorch_amd import hip
    g = torch.Generator(device="cpu").manual_seed(4)
    for B, La, Lb, cut in ((7, 9, 8, 3), (1300, 5, 6, 2)):
        ma = (torch.rand(B, La, generator=g) < 0.6).float()
        mb = (torch.rand(B, Lb, generator=g) < 0.5).float()
        ma[:, 0] = 1
        add_a, add_b = ((1 - ma) * -10000.0).to(dev), ((1 - mb) * -10000.0).to(dev)

        def walk1(m, L):
            pos, idx, st, ln, r = -np.ones((B, L), np.int64), [], [], [], 0
            for b in range(B):
                st.append(r)
                for l in range(L):
                    if m[b, l]:
                        pos[b, l] = r
                        idx.append(b * L + l)
                        r += 1
                ln.append(r - st[-1])
            return pos, np.array(idx), np.array(st), np.array(ln)

        pa, ia, sa, la, ca = hip.pack_maps([dict(mask=add_a, len=La, src_seq_stride=La)], B)
        pb, ib, sb, lb, cb = hip.pack_maps([dict(mask=add_b, len=Lb, src_seq_stride=Lb)], B)
        for (p, i, s_, l_, c), (m, L) in (((pa, ia, sa, la, ca), (ma.numpy(), La)), ((pb, ib, sb, lb, cb), (mb.numpy(), Lb))):
            rp, ri, rs, rl = walk1(m, L)
            assert np.array_equal(p.cpu().numpy(), rp) and np.array_equal(s_.cpu().numpy(), rs) and np.array_equal(l_.cpu().numpy(), rl)
            assert c.tolist() == [len(ri), int(rl.max())] and np.array_equal(i.cpu().numpy()[:len(ri)], ri)
        ra = int(ca[0])
        sel_t = torch.randint(0, B, (2 * B,), generator=g)
        sel_i = torch.randint(0, B, (2 * B,), generator=g)
        pj, ij, sj, lj, cj = hip.pack_maps([dict(mask=add_a, sel=sel_t.to(dev), len=La, pos=pa),
                                            dict(mask=add_b, sel=sel_i.to(dev), col0=cut, len=Lb - cut, pos=pb, src_base=ra)], 2 * B)
        rpa, rpb = pa.cpu().numpy(), pb.cpu().numpy()
        pos, idx, st, ln, r = -np.ones((2 * B, La + Lb - cut), np.int64), [], [], [], 0
        for s_ in range(2 * B):
            t, i = int(sel_t[s_]), int(sel_i[s_])
            st.append(r)
            for l in range(La):
                if ma[t, l]:
                    pos[s_, l] = r
                    idx.append(rpa[t, l])
                    r += 1
            for l in range(cut, Lb):
                if mb[i, l]:
                    pos[s_, La + l - cut] = r
                    idx.append(ra + rpb[i, l])
                    r += 1
            ln.append(r - st[-1])
        assert np.array_equal(pj.cpu().numpy(), pos) and np.array_equal(sj.cpu().numpy(), np.array(st)) and np.array_equal(lj.cpu().numpy(), np.array(ln))
        assert cj.tolist() == [r, max(ln)] and np.array_equal(ij.cpu().numpy()[:r], np.array(idx))


@pytest.mark.parametrize("M,N,K,splits", [(2828, 768, 30528, 5), (800, 768, 30528, 8), (300, 520, 1000, 3), (129, 768, 4096, 2)])
def test_gemm_nt_splitk(dev, M, N, K, splits):
    """mvptr_gemm_nt_splitk: the f32 slabs of the K slices add up to the full product (the decoder's data gradient)."""
    from mvp_pytorch_amd import hip
    g = torch.Generator(device="cpu").manual_seed(M + K)
    a = _bf(torch.randn(M, K, generator=g) * 0.1).to(dev)
    b = _bf(torch.randn(N, K, generator=g) * 0.1).to(dev)
    slabs = hip.gemm_nt_splitk(a, b, splits)
    assert slabs.shape == (splits, M, N)
    ref = a.float() @ b.float().t()
    assert _rel(slabs.sum(0), ref) < 1e-5
    # every slab is the product over its own slice: the slices partition K
    klen = ((K + splits - 1) // splits + 63) // 64 * 64
    for z in range(splits):
        lo, hi = z * klen, min(K, (z + 1) * klen)
        part = a[:, lo:hi].float() @ b[:, lo:hi].float().t() if lo < hi else torch.zeros(M, N, device=dev)
        assert (slabs[z] - part).abs().max() < 1e-4 * max(1.0, float(part.abs().max()))


@pytest.mark.parametrize("n,Pw,Rw,H", [(16, 5, 12, 768), (9, 20, 7, 128), (64, 70, 50, 768)])
def test_wra_loss_kernel_against_torch_and_oracle(dev, n, Pw, Rw, H):
    """csrc/wra.hip (forward + gradient) against the torch restatement of vl:1285-1300 / 1553-1596 on the same rows and
    draws, and against the oracle's per-sample walk; samples without phrases and with few regions included."""
    from mvp_pytorch_amd import engine, hip
    from mvp_pytorch_amd.modeling import modeling_vlbert as mv
    from oracle import mvptr_oracle as orc
    g = torch.Generator(device="cpu").manual_seed(n + Pw)
    La = Pw + 3
    np_ = torch.randint(0, Pw + 1, (n,), generator=g)
    np_[0], np_[1] = 0, Pw
    nr_ = torch.randint(3, Rw + 1, (n,), generator=g)
    p0 = torch.randint(1, 3, (n,), generator=g)
    phrase_index = torch.stack([p0, p0 + np_], 1)
    img_index = torch.stack([torch.full((n,), La), La + nr_], 1)
    txt = _bf(torch.randn(n, Pw, H, generator=g)).to(dev)
    reg = _bf(torch.randn(n, Rw, H, generator=g) + 0.3 * torch.randn(n, 1, H, generator=g)).to(dev)
    pos_pick = torch.randint(0, 3, (n, Pw), generator=g).to(dev)
    neg_pick = torch.randint(0, 3, (n, Pw), generator=g).to(dev)
    neg_img = ((torch.arange(n) + 1 + torch.randint(0, n - 1, (n,), generator=g)) % n).to(dev)
    pi, ii = phrase_index.to(dev), img_index.to(dev)
    valid_p = torch.arange(Pw, device=dev)[None, :] < (pi[:, 1] - pi[:, 0])[:, None]
    valid_r = torch.arange(Rw, device=dev)[None, :] < (ii[:, 1] - ii[:, 0])[:, None]

    def torch_loss(t, r):
        tz = torch.where(valid_p[:, :, None], t.float(), torch.zeros((), device=dev))   # the gathers deliver zero rows there
        tn = torch.nn.functional.normalize(tz, p=2, dim=-1)
        rn = torch.nn.functional.normalize(r.float(), p=2, dim=-1)
        pos, neg = mv._wra_from_rows(tn, rn, valid_p, valid_r, draws=(pos_pick, neg_pick, neg_img))
        valid = (pi[:, 1] - pi[:, 0]) > 0
        hinge = torch.clamp(neg + 0.2 - pos, min=0)
        return torch.where(valid, hinge, torch.zeros_like(hinge)).sum() / valid.sum().to(hinge.dtype), pos, neg

    t1, r1 = txt.clone().requires_grad_(True), reg.clone().requires_grad_(True)
    loss = engine.WraLossFn.apply(t1, r1, pi, ii, pos_pick, neg_pick, neg_img)
    loss.backward()
    t2, r2 = txt.clone().float().requires_grad_(True), reg.clone().float().requires_grad_(True)
    ref, pos_ref, neg_ref = torch_loss(t2, r2)
    ref.backward()
    print("wra", n, Pw, Rw, H, float(loss.detach()), float(ref.detach()))
    assert abs(float(loss) - float(ref)) < 2e-6 + 1e-5 * abs(float(ref))
    # rows beyond a sample's counts receive exactly zero; the rest agree to bf16 rounding of the gradient rows
    assert float(t1.grad.float()[~valid_p].abs().max()) == 0.0
    assert float(r1.grad.float()[~valid_r].abs().max()) == 0.0
    assert _rel(t1.grad, t2.grad) < 4e-3 and _rel(r1.grad, r2.grad) < 4e-3
    # run-to-run identical (fixed summation order)
    t3, r3 = txt.clone().requires_grad_(True), reg.clone().requires_grad_(True)
    engine.WraLossFn.apply(t3, r3, pi, ii, pos_pick, neg_pick, neg_img).backward()
    assert torch.equal(t3.grad, t1.grad) and torch.equal(r3.grad, r1.grad)
    # the oracle's walk over samples (vl:1553-1596) on the joint layout these rows come from
    Lj = La + Rw
    seq = torch.zeros(n, Lj, H)
    for i in range(n):
        seq[i, int(p0[i]):int(p0[i]) + int(np_[i])] = txt[i, :int(np_[i])].float().cpu()
        seq[i, La:La + int(nr_[i])] = reg[i, :int(nr_[i])].float().cpu()
    picks = []
    for i in range(n):      # the oracle draws per sample: ranks for the own image, the other image, ranks for it
        if int(np_[i]):
            picks += [pos_pick[i, :int(np_[i])].tolist(), neg_pick[i, :int(np_[i])].tolist()]
    loss_o = orc.wra_loss_sample(seq, phrase_index, img_index, orc.Draws(randint3=picks, choice=neg_img.tolist()))
    assert abs(float(loss) - float(loss_o)) < 1e-5


def test_wra_rows_against_torch(dev):
    from mvp_pytorch_amd import hip
    g = torch.Generator(device="cpu").manual_seed(3)
    n, Lj, Pw, Rw = 11, 40, 6, 9
    pos = torch.randint(-1, 1000, (2 * n, Lj), generator=g, dtype=torch.int32).to(dev)
    p0 = torch.randint(0, 5, (n,), generator=g)
    i0 = torch.randint(20, 30, (n,), generator=g)
    phrase_index = torch.stack([p0, p0 + torch.randint(0, Pw + 1, (n,), generator=g)], 1).to(dev)
    img_index = torch.stack([i0, i0 + torch.randint(0, Rw + 1, (n,), generator=g)], 1).to(dev)
    rows_p, rows_r = hip.wra_rows(pos, phrase_index, img_index, n, Pw, Rw)
    for rows, index, W in ((rows_p, phrase_index, Pw), (rows_r, img_index, Rw)):
        ar = torch.arange(W, device=dev)
        valid = ar[None, :] < (index[:, 1] - index[:, 0])[:, None]
        want = torch.where(valid, pos[:n].gather(1, (index[:, :1] + ar[None, :]).clamp(max=Lj - 1)), torch.full((), -1, dtype=torch.int32, device=dev))
        assert torch.equal(rows, want)


@pytest.mark.parametrize("n", [4, 64, 256, 300])
def test_hard_negative_mine_against_torch_and_oracle(dev, n):
    """mvptr_hard_negative_mine (vl:529-566, hn_mod 'hard') against the reference's formulation (the one the oracle
    restates, oracle/mvptr_oracle.py:242-244): argmax of sim - 2 I along rows / columns, the permutation split, the
    `sel` vectors.  (The model-level fixtures assert the same indices against the reference's own.)"""
    from mvp_pytorch_amd import hip
    g = torch.Generator(device="cpu").manual_seed(n)
    f = torch.nn.functional.normalize(torch.randn(n, 32, generator=g), dim=-1)
    h = torch.nn.functional.normalize(f + 0.3 * torch.randn(n, 32, generator=g), dim=-1)
    sim = (f @ h.t()).contiguous()
    perm = torch.randperm(n, generator=g)
    masked = sim - 2 * torch.eye(n)
    ref_img, ref_txt = masked.max(1)[1], masked.max(0)[1]
    hi, ht = hip.hard_negative_mine(sim.to(dev))
    assert torch.equal(hi.cpu(), ref_img) and torch.equal(ht.cpu(), ref_txt)
    hi, ht, htf, hif, st, si = hip.hard_negative_mine(sim.to(dev), perm.to(dev), want_sel=True)
    first, second = perm[: n // 2], perm[n // 2:]
    ar = torch.arange(n)
    ref_tf = torch.cat([ar[first], ref_txt[second]])
    ref_if = torch.cat([ref_img[first], ar[second]])
    assert torch.equal(htf.cpu(), ref_tf) and torch.equal(hif.cpu(), ref_if)
    assert torch.equal(st.cpu(), torch.cat([ar, ref_tf])) and torch.equal(si.cpu(), torch.cat([ar, ref_if]))
    # a tie goes to the lowest index (torch.max on the CPU does the same)
    tie = torch.zeros(8, 8)
    hi2, ht2 = hip.hard_negative_mine(tie.to(dev))
    assert hi2.cpu().tolist() == [1, 0, 0, 0, 0, 0, 0, 0] and ht2.cpu().tolist() == [1, 0, 0, 0, 0, 0, 0, 0]


@pytest.mark.parametrize("rows,cols", [(6, 3129), (64, 3129), (512, 17)])
def test_bce_logits_against_torch(dev, rows, cols):
    """mvptr_bce_logits = instance_bce_with_logits (vl:878-883): loss and gradient against torch autograd."""
    from mvp_pytorch_amd import engine
    g = torch.Generator(device="cpu").manual_seed(rows + cols)
    x = (torch.randn(rows, cols, generator=g) * 3).requires_grad_(True)
    y = (torch.rand(rows, cols, generator=g) < 0.01).float() * torch.rand(rows, cols, generator=g)
    ref = torch.nn.functional.binary_cross_entropy_with_logits(x, y, reduction="mean") * cols
    ref.backward()
    xd = x.detach().to(dev).requires_grad_(True)
    got = engine.BceLogitsFn.apply(xd, y.to(dev))
    (got * 1.5).backward()
    assert abs(got.item() - ref.item()) < 2e-6 * abs(ref.item()) + 1e-6
    assert _rel(xd.grad.cpu(), 1.5 * x.grad) < 1e-5
    # bitwise reproducible (ordered partial sums)
    got2 = engine.BceLogitsFn.apply(xd.detach(), y.to(dev))
    assert got2.item() == got.item()


# ----------------------------------------------------------------- LayerNorm folded into the neighbouring GEMMs (NS-1, inference path)
@pytest.mark.parametrize("M,N,K,gelu", [(1000, 2304, 768, False), (777, 3072, 768, True), (300, 768, 3072, False)])
def test_gemm_nt_ln_fold_matches_layernorm_then_linear(dev, M, N, K, gelu):
    """mvptr_gemm_nt_ln FOLD modes: the GEMM runs on the PRE-LayerNorm rows with the gamma-scaled weight and finishes the
    LayerNorm in its epilogue; reference = BertLayerNorm (mb:242-246) in f32, then nn.Linear (+ erf-GELU, mb:394-397) in f32."""
    from mvp_pytorch_amd import hip
    g = torch.Generator().manual_seed(5)
    z = (torch.randn(M, K, generator=g) * 0.8 + 0.3 * torch.randn(M, 1, generator=g)).to(torch.bfloat16).to(dev)
    gamma = (torch.rand(K, generator=g) + 0.5).to(dev)
    beta = (torch.randn(K, generator=g) * 0.2).to(dev)
    W = (torch.randn(N, K, generator=g) * 0.03).to(dev)
    b = (torch.randn(N, generator=g) * 0.1).to(dev)
    zf = z.float()
    mu = zf.mean(1, keepdim=True)
    var = ((zf - mu) ** 2).mean(1, keepdim=True)
    x = (zf - mu) / torch.sqrt(var + 1e-12) * gamma + beta
    ref = x @ W.t() + b
    if gelu:
        ref = torch.nn.functional.gelu(ref)
    stats = torch.cat([mu, 1.0 / torch.sqrt(var + 1e-12)], 1).contiguous()
    wf = (W * gamma[None, :]).to(torch.bfloat16).contiguous()
    out = hip.gemm_nt_ln(z, wf, hip.LN_FOLD_GELU if gelu else hip.LN_FOLD_BIAS, (W @ beta + b).contiguous(), stats=stats,
                         colsum=wf.float().sum(1).contiguous())
    # the unfused product path for scale: LayerNorm kernel (bf16 out), then the plain GEMM
    y = hip.layernorm_fwd(z, gamma, beta, 1e-12)[0] if K <= 1024 else x.to(torch.bfloat16)      # the LayerNorm kernel takes rows of <= 1024
    unf = hip.gemm_nt(y, W.to(torch.bfloat16).contiguous(), hip.EPI_BIAS, bias=b).float()
    if gelu:
        unf = torch.nn.functional.gelu(unf)
    e_fold, e_unf = _rel(out.float(), ref), _rel(unf, ref)
    print("LN fold M=%d N=%d K=%d gelu=%s: folded rel L2 %.2e, unfused %.2e" % (M, N, K, gelu, e_fold, e_unf))
    assert e_fold < 6e-3 and e_fold < 1.5 * e_unf + 1e-3, (e_fold, e_unf)


@pytest.mark.parametrize("M,K,inline_ln", [(900, 768, False), (1111, 3072, True)])
def test_gemm_nt_ln_residual_and_row_statistics(dev, M, K, inline_ln):
    """mvptr_gemm_nt_ln RESID_STATS: z = A W^T + b + r with r the residual rows or LayerNorm(residual rows) computed on the fly;
    the partial sums it leaves + mvptr_ln_stats_finalize = mean / rstd of the STORED rows (what a LayerNorm kernel reading z would
    compute, mb:242-246)."""
    from mvp_pytorch_amd import hip
    N = 768
    g = torch.Generator().manual_seed(6)
    a = (torch.randn(M, K, generator=g) * 0.5).to(torch.bfloat16).to(dev)
    W = (torch.randn(N, K, generator=g) * 0.03).to(torch.bfloat16).to(dev)
    b = (torch.randn(N, generator=g) * 0.1).to(dev)
    r = (torch.randn(M, N, generator=g) * 0.7 + 0.2).to(torch.bfloat16).to(dev)
    gamma = (torch.rand(N, generator=g) + 0.5).to(dev)
    beta = (torch.randn(N, generator=g) * 0.2).to(dev)
    rf = r.float()
    if inline_ln:
        mu = rf.mean(1, keepdim=True)
        var = ((rf - mu) ** 2).mean(1, keepdim=True)
        st_r = torch.cat([mu, 1.0 / torch.sqrt(var + 1e-12)], 1).contiguous()
        res = (rf - mu) / torch.sqrt(var + 1e-12) * gamma + beta
        z, part = hip.gemm_nt_ln(a, W, hip.LN_RESID_STATS, b, aux=r, stats=st_r, gamma=gamma, beta=beta)
    else:
        res = rf
        z, part = hip.gemm_nt_ln(a, W, hip.LN_RESID_STATS, b, aux=r)
    ref = a.float() @ W.float().t() + b + res
    assert _rel(z.float(), ref) < 4e-3
    stats = hip.ln_stats_finalize(part, N, 1e-12)
    zf = z.float()
    mu = zf.mean(1)
    rstd = 1.0 / torch.sqrt(((zf - mu[:, None]) ** 2).mean(1) + 1e-12)
    assert float((stats[:, 0] - mu).abs().max()) < 2e-5 * max(1.0, float(mu.abs().max()))
    assert float(((stats[:, 1] - rstd) / rstd).abs().max()) < 2e-4


def test_gemm_nt_ln_rejects_shapes_outside_the_256_tile_kernel(dev):
    from mvp_pytorch_amd import hip
    z = torch.zeros(64, 192, device=dev, dtype=torch.bfloat16)
    w = torch.zeros(320, 192, device=dev, dtype=torch.bfloat16)
    with pytest.raises(RuntimeError):
        hip.gemm_nt_ln(z, w, hip.LN_FOLD_BIAS, torch.zeros(320, device=dev), stats=torch.zeros(64, 2, device=dev), colsum=torch.zeros(320, device=dev))
