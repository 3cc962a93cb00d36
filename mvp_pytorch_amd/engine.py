"""Autograd glue between the `oscar.modeling`-style modules and the HIP kernels.

Everything on the encoder path runs in bf16 with f32 accumulation through the C ABI
(mvp_pytorch_amd.hip); the nn.Parameters stay f32 master weights with the reference's names
and shapes, and kernel-friendly bf16 copies (Q|K|V packed, plus transposed copies for the
data-gradient GEMMs) are rebuilt whenever a parameter's version counter changes.

Functions
  InputEmbedFn   BertEmbeddings (+ image embedding + concat)   modeling_bert.py:262-277,
                                                                modeling_vlbert.py:328-336,498-506
  EncoderFn      CaptionBertEncoder (n layers)                  modeling_vlbert.py:134-199
  LinearFn       nn.Linear (+gelu) on bf16 rows                 modeling_bert.py:484-488
  LayerNormFn    BertLayerNorm on bf16 rows                     modeling_bert.py:242-246
  DecoderCEFn    vocabulary decoder + CrossEntropyLoss          modeling_bert.py:513-516,
                                                                modeling_vlbert.py:1228-1249
"""
import ctypes
import os
import threading

import torch

from . import hip

# nn.DataParallel (oscar/run_retrieval.py:577-578,1125 — that script's only multi-GPU mode) runs one
# THREAD per device over replicas whose non-tensor attributes are shared by reference
# (torch.nn.parallel.replicate copies module.__dict__ shallowly): every piece of module-level state
# below is therefore either guarded by this lock or kept per device (WeightCache.for_device,
# PackList.for_device, side_stream).
_state_lock = threading.Lock()
_tls = threading.local()


class GradAwareFunction(torch.autograd.Function):
    """autograd.Function whose forward can see the CALLER's grad mode: PyTorch runs Function.forward with grad mode
    off, and the weight caches decide "parameters are being trained -> rebuild the bf16 copies at every forward pass"
    from it (WeightCache.stale)."""

    @classmethod
    def apply(cls, *args, **kwargs):
        prev = getattr(_tls, "outer_grad", None)
        _tls.outer_grad = torch.is_grad_enabled()
        try:
            return super().apply(*args, **kwargs)
        finally:
            _tls.outer_grad = prev


def _caller_grad_enabled():
    g = getattr(_tls, "outer_grad", None)
    return torch.is_grad_enabled() if g is None else g
_seed_counter = [0x5DEECE66D]


def next_seed():
    """Fresh 64-bit dropout seed; derived from torch's RNG so torch.manual_seed controls it."""
    with _state_lock:
        _seed_counter[0] = (_seed_counter[0] * 6364136223846793005 + 1442695040888963407) & 0xFFFFFFFFFFFFFFFF
        base = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
        return (_seed_counter[0] ^ (base << 20)) & 0xFFFFFFFFFFFFFFFF


def _dev_key(device):
    d = torch.device(device)
    if d.type != "cuda":
        return -1
    return d.index if d.index is not None else torch.cuda.current_device()


def _f32(p):
    t = p.detach()
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def pad8(n):
    return (n + 7) // 8 * 8


class WeightCache:
    """bf16 working copies of a group of f32 parameters.

    Inference (grad mode off, or frozen parameters): refreshed when a parameter's (data_ptr, version
    counter) changes.  In-place updates through `.data` do NOT bump the version counter (the
    reference's own AdamW does `p.data.addcdiv_`, optimization.py:176,187; so do EMA / master-weight
    copies), so while the parameters are being trained — grad mode on and any of them requires grad —
    the copies are rebuilt at EVERY forward pass: the weights change every step anyway and the fused
    cast of an encoder stack is one ~0.1 ms launch.  For inference after an update through `.data`
    call invalidate_weight_caches(model)."""

    def __init__(self):
        self._key = None
        self._fresh_once = False  # set by a prefetch: the next stale() of the same parameters answers "fresh" once
        self.t = {}
        self._dev = None        # device this object serves (set at first use)
        self._children = {}     # other devices' caches (replicas of the owning module under DataParallel)

    def for_device(self, device):
        """The cache to use for parameters living on `device`: this object for the first device seen,
        a per-device child otherwise (replicas made by torch.nn.parallel.replicate share this object
        by reference while their parameters live on different GPUs and are used from different threads)."""
        k = _dev_key(device)
        if self._dev is None:
            with _state_lock:
                if self._dev is None:
                    self._dev = k
        if k == self._dev:
            return self
        with _state_lock:
            c = self._children.get(k)
            if c is None:
                c = self._children[k] = WeightCache()
                c._dev = k
        return c

    def stale(self, params, force=None):
        if force is None:
            force = _caller_grad_enabled() and any(p.requires_grad for p in params)
        key = tuple((p.data_ptr(), p._version) for p in params)
        if self._fresh_once:
            # the copies were rebuilt earlier in THIS forward pass (EncoderPacks.prefetch, on the side stream beside the
            # embedding kernels): do not rebuild them again
            self._fresh_once = False
            if key == self._key:
                return False
        if force or key != self._key:
            self._key = key
            return True
        return False

    def invalidate(self):
        self._key = None
        for c in self._children.values():
            c.invalidate()


def invalidate_weight_caches(model):
    """Drop every bf16 working copy held by `model`'s modules (they are rebuilt at the next forward
    pass).  Needed only in inference after parameters were modified through `.data` / a raw pointer,
    which PyTorch's version counters do not see."""
    n = 0
    for m in model.modules():
        for v in list(m.__dict__.values()):
            if isinstance(v, WeightCache):
                v.invalidate()
                n += 1
            elif isinstance(v, PackList):
                for pl in [v] + list(v._children.values()):
                    pl.group.cache.invalidate()
                    for pk in pl:
                        pk.cache.invalidate()
                n += 1
    return n


def cast_weight(w, want_t=True, kpad=None):
    """f32 [N,K] -> (bf16 [N,Kp], bf16 [K,Np] or None).  Leading dims padded to multiples of 8."""
    w = _f32(w)
    N, K = w.shape
    Kp = pad8(K) if kpad is None else kpad
    dst = torch.empty((N, Kp), device=w.device, dtype=torch.bfloat16)
    alloc_t = torch.empty if N % 8 == 0 else torch.zeros  # only pad columns need the zeros
    dst_t = alloc_t((K, pad8(N)), device=w.device, dtype=torch.bfloat16) if want_t else None
    hip.cast_pack(w, dst=dst, dst_t=dst_t)
    return dst, dst_t


# ---------------------------------------------------------------------------------------------
class LayerPack:
    """Kernel-side view of one CaptionBertLayer's parameters."""
    NAMES = ["attention.self.query.weight", "attention.self.query.bias",
             "attention.self.key.weight", "attention.self.key.bias",
             "attention.self.value.weight", "attention.self.value.bias",
             "attention.output.dense.weight", "attention.output.dense.bias",
             "attention.output.LayerNorm.weight", "attention.output.LayerNorm.bias",
             "intermediate.dense.weight", "intermediate.dense.bias",
             "output.dense.weight", "output.dense.bias",
             "output.LayerNorm.weight", "output.LayerNorm.bias"]

    def __init__(self):
        self.cache = WeightCache()
        self.w = None
        self.keep = None

    def refresh(self, params):
        if not self.cache.stale(params):
            return self.w
        (qw, qb, kw, kb, vw, vb, ow, ob, g1, b1, iw, ib, pw, pb, g2, b2) = params
        dev = qw.device
        H = qw.shape[1]
        I = iw.shape[0]
        bf = torch.bfloat16
        w_qkv = torch.empty((3 * H, H), device=dev, dtype=bf)
        w_qkv_t = torch.empty((H, 3 * H), device=dev, dtype=bf)
        for i, w in enumerate((qw, kw, vw)):
            hip.cast_pack(_f32(w), dst=w_qkv[i * H:(i + 1) * H], dst_t=w_qkv_t, col_off_t=i * H)
        b_qkv = torch.cat([_f32(qb), _f32(kb), _f32(vb)])
        w_o, w_o_t = cast_weight(ow)
        w_i, w_i_t = cast_weight(iw)
        w_out, w_out_t = cast_weight(pw)
        f = [_f32(x) for x in (ob, g1, b1, ib, pb, g2, b2)]
        self.keep = [w_qkv, w_qkv_t, b_qkv, w_o, w_o_t, w_i, w_i_t, w_out, w_out_t] + f
        lw = hip.LayerWeights()
        lw.w_qkv, lw.w_qkv_t, lw.b_qkv = w_qkv.data_ptr(), w_qkv_t.data_ptr(), b_qkv.data_ptr()
        lw.w_o, lw.w_o_t, lw.b_o = w_o.data_ptr(), w_o_t.data_ptr(), f[0].data_ptr()
        lw.ln1_g, lw.ln1_b = f[1].data_ptr(), f[2].data_ptr()
        lw.w_i, lw.w_i_t, lw.b_i = w_i.data_ptr(), w_i_t.data_ptr(), f[3].data_ptr()
        lw.w_out, lw.w_out_t, lw.b_out = w_out.data_ptr(), w_out_t.data_ptr(), f[4].data_ptr()
        lw.ln2_g, lw.ln2_b = f[5].data_ptr(), f[6].data_ptr()
        self.w = lw
        self.dims = (H, I)
        return lw


class PackList(list):
    """The LayerPack objects of one encoder stack + their shared EncoderPacks refresher."""

    def __init__(self, it):
        super().__init__(it)
        self.group = EncoderPacks(self)
        self._dev = None
        self._children = {}

    def for_device(self, device):
        """Per-device working copies (see WeightCache.for_device): this list for the first device that
        uses it, a lazily built twin for every other device."""
        k = _dev_key(device)
        if self._dev is None:
            with _state_lock:
                if self._dev is None:
                    self._dev = k
        if k == self._dev:
            return self
        with _state_lock:
            c = self._children.get(k)
            if c is None:
                c = self._children[k] = PackList(LayerPack() for _ in range(len(self)))
                c._dev = k
        return c


class EncoderPacks:
    """bf16 working copies of ALL layers of one encoder stack, refreshed with one mvptr_cast_multi
    launch when any parameter version changes (after every optimizer step): buffers and the device
    task table are built once, so a refresh is a single kernel instead of 6 casts + 4 fills per
    layer.  Falls back to the per-layer LayerPack path for parameters that are not contiguous f32."""

    def __init__(self, packs):
        self.packs = packs          # per-layer LayerPack objects (fallback path, and owners of .w)
        self.cache = WeightCache()
        self.ptr_key = None
        self.plan = None

    def _build(self, params):
        n = len(self.packs)
        dev = params[0].device
        bf = torch.bfloat16
        plan = hip.CastPlan(dev)
        for li in range(n):
            (qw, qb, kw, kb, vw, vb, ow, ob, g1, b1, iw, ib, pw, pb, g2, b2) = params[16 * li:16 * (li + 1)]
            H, I = qw.shape[1], iw.shape[0]
            assert H % 8 == 0 and I % 8 == 0
            w_qkv = torch.empty((3 * H, H), device=dev, dtype=bf)
            w_qkv_t = torch.empty((H, 3 * H), device=dev, dtype=bf)
            b_qkv = torch.empty(3 * H, device=dev, dtype=torch.float32)
            for i, (w, b) in enumerate(((qw, qb), (kw, kb), (vw, vb))):
                plan.add(w.data, dst=w_qkv[i * H:(i + 1) * H], dst_t=w_qkv_t, col_off_t=i * H)
                plan.add(b.data, dst_f32=b_qkv[i * H:(i + 1) * H])
            w_o, w_o_t = torch.empty((H, H), device=dev, dtype=bf), torch.empty((H, H), device=dev, dtype=bf)
            w_i, w_i_t = torch.empty((I, H), device=dev, dtype=bf), torch.empty((H, I), device=dev, dtype=bf)
            w_out, w_out_t = torch.empty((H, I), device=dev, dtype=bf), torch.empty((I, H), device=dev, dtype=bf)
            plan.add(ow.data, dst=w_o, dst_t=w_o_t)
            plan.add(iw.data, dst=w_i, dst_t=w_i_t)
            plan.add(pw.data, dst=w_out, dst_t=w_out_t)
            lw = hip.LayerWeights()
            lw.w_qkv, lw.w_qkv_t, lw.b_qkv = w_qkv.data_ptr(), w_qkv_t.data_ptr(), b_qkv.data_ptr()
            lw.w_o, lw.w_o_t, lw.b_o = w_o.data_ptr(), w_o_t.data_ptr(), ob.data_ptr()
            lw.ln1_g, lw.ln1_b = g1.data_ptr(), b1.data_ptr()
            lw.w_i, lw.w_i_t, lw.b_i = w_i.data_ptr(), w_i_t.data_ptr(), ib.data_ptr()
            lw.w_out, lw.w_out_t, lw.b_out = w_out.data_ptr(), w_out_t.data_ptr(), pb.data_ptr()
            lw.ln2_g, lw.ln2_b = g2.data_ptr(), b2.data_ptr()
            pk = self.packs[li]
            pk.keep = [w_qkv, w_qkv_t, b_qkv, w_o, w_o_t, w_i, w_i_t, w_out, w_out_t]
            pk.w = lw
            pk.dims = (H, I)
            pk.cache._key = None
        plan.build()
        self.plan = plan

    def prefetch(self, params):
        """Rebuild the bf16 copies now, on the current stream; the stack's own refresh in the same forward pass
        then finds them fresh.  Only for the one-launch path (contiguous f32 parameters)."""
        if all(p.dtype == torch.float32 and p.is_contiguous() for p in params):
            self.refresh(params)       # training: rebuilt unconditionally (WeightCache.stale)
            self.cache._fresh_once = True

    def refresh(self, params):
        """-> list of LayerWeights, one per layer."""
        if not all(p.dtype == torch.float32 and p.is_contiguous() for p in params):
            return [self.packs[li].refresh(params[16 * li:16 * (li + 1)]) for li in range(len(self.packs))]
        if self.cache.stale(params):
            ptr_key = tuple(p.data_ptr() for p in params)
            if ptr_key != self.ptr_key:
                self._build(params)
                self.ptr_key = ptr_key
            self.plan.run()
        return [pk.w for pk in self.packs]


class EncoderMeta:
    """Static description handed to EncoderFn (not a tensor)."""

    def __init__(self, packs, B, L, H, heads, I, eps, training, p_hidden, p_attn, seq_start=None, seq_len=None, rows=0):
        """seq_start / seq_len (device int32 [B]) + rows: row-packed mode — x holds `rows` valid token
        rows, sequence b at [seq_start[b], +seq_len[b]), L = the longest sequence."""
        self.packs, self.B, self.L, self.H, self.heads, self.I = packs, B, L, H, heads, I
        self.seq_start, self.seq_len, self.rows = seq_start, seq_len, rows
        self.group = getattr(packs, "group", None)
        self.eps, self.training, self.p_hidden, self.p_attn = eps, training, p_hidden, p_attn


def _thresh(p):
    return int(round(float(p) * 65536.0))


# Side streams used for independent sub-networks (text / visual stack); mvp_pytorch_amd.dp waits on
# them before a gradient bucket is handed to RCCL.
SIDE_STREAMS = {}
# Weight gradients of an encoder stack on a stream of their own (EncoderFn.backward, mvptr_encoder_layer_bwd2): opt-in
# (MVPTR_WGRAD_ASIDE=1, optionally only for stacks of at most MVPTR_WGRAD_ASIDE_MAX_ROWS rows).  Measured on the
# training step: -0.3 ms on one box and nothing on another for the variable-length batch, +0.4 ms (worse) with all
# slots valid — the weight-gradient kernel owns a CU's whole register file and LDS, so "beside" means fewer CUs for
# the data-gradient chain, which only pays where that chain leaves CUs idle (profiles/r02_experiments.txt).
WGRAD_ASIDE = os.environ.get("MVPTR_WGRAD_ASIDE", "0") == "1"
WGRAD_ASIDE_MAX_ROWS = int(os.environ.get("MVPTR_WGRAD_ASIDE_MAX_ROWS", "1000000"))


WGRAD_STREAMS = {}


def wgrad_stream(device, parent):
    """The stream on which a stack's weight-gradient GEMMs run beside the rest of its backward pass: one per
    (device, stream the backward pass itself runs on) — the text and visual stacks run their backward passes on
    two streams at once."""
    dev = torch.device(device)
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), parent.cuda_stream)
    with _state_lock:
        if key not in WGRAD_STREAMS:
            WGRAD_STREAMS[key] = torch.cuda.Stream(device=key[0])
        return WGRAD_STREAMS[key]


def side_stream(device):
    """One extra HIP stream per device, created on first use."""
    key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    with _state_lock:
        if key not in SIDE_STREAMS:
            SIDE_STREAMS[key] = torch.cuda.Stream(device=key)
        return SIDE_STREAMS[key]


class small_f32_blas:
    """Scope in which torch's matrix products go to rocBLAS.  hipBLASLt, torch's default BLAS on this ROCm
    build, serves the step's [256, 768] x [768, 256] f32 global projections and the 256 x 256
    similarity matrix with ONE 256 x 256 tile = one workgroup (173 us and 63 us; rocBLAS: 7 and 8 us,
    tools/small_mm.py).  The switch is process-wide in torch, so the scope is reference-counted under
    the module lock: the previous setting comes back when the last concurrent user (nn.DataParallel
    replicas run their forward passes in threads) leaves.  MVPTR_KEEP_BLAS=1 disables it."""
    _depth = 0
    _saved = None

    def __enter__(self):
        if os.environ.get("MVPTR_KEEP_BLAS") == "1" or not getattr(torch.version, "hip", None):
            return self
        with _state_lock:
            if small_f32_blas._depth == 0:
                small_f32_blas._saved = torch.backends.cuda.preferred_blas_library()
                torch.backends.cuda.preferred_blas_library("cublas")      # = rocBLAS on ROCm
            small_f32_blas._depth += 1
        self._entered = True
        return self

    def __exit__(self, *exc):
        if getattr(self, "_entered", False):
            with _state_lock:
                small_f32_blas._depth -= 1
                if small_f32_blas._depth == 0:
                    torch.backends.cuda.preferred_blas_library(small_f32_blas._saved)
        return False


class AsyncCounts:
    """A few device-side integers (row counts that size later tensors) on their way to the host.
    `tolist()` on the device tensor would wait for EVERYTHING queued on the stream and hand the GPU an
    empty queue; here the copy goes to pinned memory right away and is awaited through an event, so
    the kernels queued after it (embeddings, index_select / cat of the hard batch) keep the GPU busy
    while the host reads the counts and starts queueing the next stage."""
    _pins = {}

    def __init__(self, values):
        t = torch.stack([v.reshape(()) for v in values]).to(torch.int64)
        if not t.is_cuda:
            self._host, self._event = t, None
            return
        # one small ring of pinned buffers per (device, count): a device is driven by one thread at a time (its
        # nn.DataParallel replica), and those threads are new ones in every forward pass, so the thread is not part of the key
        key = (t.device.index, t.numel())
        with _state_lock:
            bufs = AsyncCounts._pins.setdefault(key, [torch.empty(t.numel(), dtype=torch.int64).pin_memory() for _ in range(8)])
            bufs.append(bufs.pop(0))       # rotate: a buffer is reused eight requests later at the earliest
            self._host = bufs[-1]
        self._host.copy_(t, non_blocking=True)
        self._event = torch.cuda.Event()
        self._event.record()

    def get(self):
        if self._event is not None:
            self._event.synchronize()
        return [int(v) for v in self._host.tolist()]


class PackRows(torch.autograd.Function):
    """x [n, H] -> x[idx] for a strictly increasing idx (the valid rows of a padded batch).  idx has no
    duplicates, so the backward pass is a plain scatter into zeros (index_copy), not the atomic
    index_add autograd derives for index_select."""

    @staticmethod
    def forward(ctx, x, idx):
        ctx.idx, ctx.n = idx, x.shape[0]
        return x.index_select(0, idx)

    @staticmethod
    def backward(ctx, g):
        out = torch.zeros((ctx.n, g.shape[1]), dtype=g.dtype, device=g.device)
        return out.index_copy_(0, ctx.idx, g), None


class UnpackRows(torch.autograd.Function):
    """inverse of PackRows: y [rows, H] scattered to row idx[i] of a zero [n, H] tensor."""

    @staticmethod
    def forward(ctx, y, idx, n):
        ctx.idx = idx
        out = torch.zeros((n, y.shape[1]), dtype=y.dtype, device=y.device)
        return out.index_copy_(0, idx, y)

    @staticmethod
    def backward(ctx, g):
        return g.index_select(0, ctx.idx), None, None


class EncoderFn(GradAwareFunction):
    """n stacked encoder layers: x bf16 [B*L,H], additive mask f32 [B,L] -> bf16 [B*L,H]."""

    @staticmethod
    def forward(ctx, x, mask_add, meta, *params):
        lib = hip.load()
        n = len(meta.packs)
        assert len(params) == 16 * n
        need_grad = meta.training or any(p.requires_grad for p in params) or x.requires_grad
        descs, stashes, xs = [], [], [x]
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        cur = x
        scratch = None
        if meta.group is not None:
            lws = meta.group.refresh(params)
        else:
            lws = [meta.packs[li].refresh(params[16 * li:16 * (li + 1)]) for li in range(n)]
        for li in range(n):
            lw = lws[li]
            d = hip.LayerDesc(meta.B, meta.L, meta.H, meta.heads, meta.I, meta.eps,
                              1 if meta.training else 0, _thresh(meta.p_hidden), _thresh(meta.p_attn),
                              next_seed() if meta.training else 0, meta.rows, 0,
                              meta.seq_start.data_ptr() if meta.rows else None,
                              meta.seq_len.data_ptr() if meta.rows else None)
            nbytes = lib.mvptr_layer_saved_bytes(ctypes.byref(d))
            if nbytes < 0:
                hip._check(-1)
            if need_grad or scratch is None:
                stash = torch.empty(nbytes, device=x.device, dtype=torch.uint8)
                scratch = stash
            else:
                stash = scratch
            y = torch.empty_like(cur)
            hip._check(lib.mvptr_encoder_layer_fwd(ctypes.byref(d), ctypes.byref(lw), hip._p(cur), hip._p(mask_add),
                                                   hip._p(y), hip._p(stash), None, 0, stream))
            descs.append(d)
            stashes.append(stash)
            cur = y
            xs.append(y)
        ctx.meta, ctx.descs, ctx.stashes, ctx.xs, ctx.mask = meta, descs, stashes, xs, mask_add
        ctx.params = params
        return cur

    @staticmethod
    def backward(ctx, dy):
        lib = hip.load()
        meta = ctx.meta
        n = len(meta.packs)
        dev = dy.device
        H, I = meta.H, meta.I
        main = torch.cuda.current_stream()
        stream = ctypes.c_void_p(main.cuda_stream)
        ws_bytes = lib.mvptr_layer_workspace_bytes(ctypes.byref(ctx.descs[0]))
        # Weight gradients beside the data-gradient chain: a layer's two grouped weight-gradient launches only feed the
        # optimizer, so they go to a stream of their own (mvptr_encoder_layer_bwd2 orders them behind the kernels
        # that produce their operands) and run beside the next kernels of the chain — LayerNorm / attention backward,
        # which leave the matrix pipes idle, and the partly filled last rounds of the N = 768 data-gradient GEMMs.
        # Their operands live in the workspace and the stash: two workspaces alternate, and a layer waits for the
        # weight gradients of the layer two before it (same workspace); the streams join at the end.
        aside = WGRAD_ASIDE and n > 1 and dy.shape[0] <= WGRAD_ASIDE_MAX_ROWS
        aux = wgrad_stream(dev, main) if aside else None
        wss = [torch.empty(ws_bytes, device=dev, dtype=torch.uint8) for _ in range(2 if aside else 1)]
        done = []
        sizes = [3 * H * H, 3 * H, H * H, H, H, H, I * H, I, H * I, H, H, H]
        total = sum(sizes)
        grads = [None] * (16 * n)
        d_cur = dy.contiguous()
        for k, li in enumerate(reversed(range(n))):
            ws = wss[k % len(wss)]
            if aside and k >= 2:
                main.wait_event(done[k - 2])
            flat = torch.zeros(total, device=dev, dtype=torch.float32)
            parts, o = [], 0
            for s in sizes:
                parts.append(flat[o:o + s])
                o += s
            g = hip.LayerGrads()
            (g.w_qkv, g.b_qkv, g.w_o, g.b_o, g.ln1_g, g.ln1_b, g.w_i, g.b_i, g.w_out, g.b_out, g.ln2_g,
             g.ln2_b) = [p.data_ptr() for p in parts]
            dx = torch.empty_like(d_cur)
            hip._check(lib.mvptr_encoder_layer_bwd2(ctypes.byref(ctx.descs[li]), ctypes.byref(meta.packs[li].w),
                                                    hip._p(ctx.xs[li]), hip._p(ctx.mask), hip._p(ctx.stashes[li]),
                                                    hip._p(d_cur), hip._p(dx), ctypes.byref(g), hip._p(ws), ws_bytes, stream,
                                                    ctypes.c_void_p(aux.cuda_stream) if aside else None))
            if aside:
                ev = torch.cuda.Event()
                ev.record(aux)
                done.append(ev)
                for t in (flat, ws, ctx.xs[li], ctx.stashes[li]):
                    t.record_stream(aux)
            wq = parts[0].view(3, H, H)
            bq = parts[1].view(3, H)
            gl = [wq[0], bq[0], wq[1], bq[1], wq[2], bq[2], parts[2].view(H, H), parts[3], parts[4], parts[5],
                  parts[6].view(I, H), parts[7], parts[8].view(H, I), parts[9], parts[10], parts[11]]
            for j in range(16):
                p = ctx.params[16 * li + j]
                if p.requires_grad:
                    grads[16 * li + j] = gl[j].to(p.dtype)
            d_cur = dx
        if aside:
            main.wait_stream(aux)      # the gradients handed back below are complete on this stream
        ctx.stashes = ctx.xs = None
        return (d_cur, None, None) + tuple(grads)


# ---------------------------------------------------------------------------------------------
class InputEmbedFn(GradAwareFunction):
    """Token embeddings (+ region-feature embedding) -> concatenated bf16 [B, Lt+R, H].

    args: ids/type_ids/pos_ids int64 [B,Lt]; img_feats f32 [B,R,D] or None; meta dict with
    eps, img_eps, use_img_ln, p (hidden dropout), training, cache (WeightCache-like dict);
    params: word, pos, type, ln_w, ln_b, img_w, img_b, img_ln_w, img_ln_b (missing -> None)."""

    @staticmethod
    def forward(ctx, ids, type_ids, pos_ids, img_feats, meta, word, pos, typ, ln_w, ln_b, img_w, img_b,
                img_ln_w, img_ln_b):
        B, Lt = ids.shape
        H = word.shape[1]
        R = img_feats.shape[1] if img_feats is not None else 0
        Ltot = Lt + R
        dev = word.device
        training = meta["training"]
        p = meta["p"] if training else 0.0
        out = torch.empty((B * Ltot, H), device=dev, dtype=torch.bfloat16)
        idf, tyf, pof = ids.reshape(-1).contiguous(), type_ids.reshape(-1).contiguous(), pos_ids.reshape(-1).contiguous()
        wordf, posf, typf = _f32(word), _f32(pos), _f32(typ)
        z = hip.embed_fwd(idf, pof, tyf, wordf, posf, typf)
        drop_t = hip.make_dropout(p, next_seed()) if p > 0 else None
        lnw, lnb = _f32(ln_w), _f32(ln_b)
        _, mean, rstd = hip.layernorm_fwd(z, lnw, lnb, meta["eps"], out=out, rows_per_group=Lt,
                                          group_stride=Ltot, row_offset=0, drop=drop_t)
        ctx.txt = (idf, tyf, pof, z, mean, rstd, lnw, drop_t)
        ctx.img = None
        if R > 0:
            D = img_feats.shape[2]
            Dp = pad8(D)
            feats = img_feats.reshape(B * R, D)
            if feats.dtype != torch.float32:
                feats = feats.float()
            fb = torch.empty((B * R, Dp), device=dev, dtype=torch.bfloat16)
            hip.cast_pack(feats.contiguous(), dst=fb)
            cache = meta["cache"]
            cache = cache.for_device(img_w.device)
            if cache.stale([img_w]):
                cache.t["img_w"] = cast_weight(img_w, want_t=False)[0]
            zi = hip.gemm_nt(fb, cache.t["img_w"], hip.EPI_BIAS, bias=_f32(img_b))
            p_img = meta.get("p_img", meta["p"]) if training else 0.0
            drop_i = hip.make_dropout(p_img, next_seed()) if p_img > 0 else None
            use_ln = img_ln_w is not None and meta["use_img_ln"]
            g = _f32(img_ln_w) if use_ln else None
            bta = _f32(img_ln_b) if use_ln else None
            _, mi, ri = hip.layernorm_fwd(zi, g, bta, meta["img_eps"], out=out, rows_per_group=R,
                                          group_stride=Ltot, row_offset=Lt, drop=drop_i, save_stats=use_ln)
            ctx.img = (fb, zi, mi, ri, g, drop_i, D)
        ctx.dims = (B, Lt, R, H)
        ctx.share = meta.get("share")
        ctx.shapes = (word.shape, pos.shape, typ.shape)
        ctx.needs = [t is not None and t.requires_grad for t in (word, pos, typ, ln_w, ln_b, img_w, img_b, img_ln_w, img_ln_b)]
        return out.view(B, Ltot, H)

    @staticmethod
    def backward(ctx, dout):
        B, Lt, R, H = ctx.dims
        Ltot = Lt + R
        dev = dout.device
        dout = dout.contiguous().view(B * Ltot, H)
        idf, tyf, pof, z, mean, rstd, lnw, drop_t = ctx.txt
        f32 = dict(device=dev, dtype=torch.float32)
        dg, db = torch.zeros(H, **f32), torch.zeros(H, **f32)
        dz, _ = hip.layernorm_bwd(dout, z, mean, rstd, lnw, dg, db, None, rows_per_group=Lt,
                                  group_stride=Ltot, row_offset=0, y_drop=drop_t)
        # The word table (86 051 x 768 f32 = 264 MB) is looked up twice per forward pass (text ids, tag
        # ids).  Both backward calls scatter into ONE zero-filled buffer (meta["share"], created per
        # forward pass by the backbone): the second call returns no gradient of its own, which saves
        # a 264-MB fill and the 0.8-GB add autograd would use to sum two dense gradients.
        share = ctx.share
        dword = share.get("dword") if share is not None else None
        first = dword is None
        if first:
            dword = torch.zeros(ctx.shapes[0], **f32)
            if share is not None:
                share["dword"] = dword
        dpos = torch.zeros(ctx.shapes[1], **f32)
        dtyp = torch.zeros(ctx.shapes[2], **f32)
        hip.embed_bwd(idf, pof, tyf, dz, dword, dpos, dtyp)
        if not first:
            dword = None
        gi_w = gi_b = gi_lw = gi_lb = None
        if ctx.img is not None:
            fb, zi, mi, ri, g, drop_i, D = ctx.img
            dbias = torch.zeros(H, **f32)
            gi_lw = torch.zeros(H, **f32) if g is not None else None
            gi_lb = torch.zeros(H, **f32) if g is not None else None
            dzi, _ = hip.layernorm_bwd(dout, zi, mi, ri, g, gi_lw, gi_lb, dbias, rows_per_group=R,
                                       group_stride=Ltot, row_offset=Lt, y_drop=drop_i)
            dw = torch.zeros((H, fb.shape[1]), **f32)
            hip.gemm_tn(dzi, fb, dw)
            gi_w, gi_b = dw[:, :D].contiguous(), dbias
        outs = [dword, dpos, dtyp, dg, db, gi_w, gi_b, gi_lw, gi_lb]
        outs = [o if need else None for o, need in zip(outs, ctx.needs)]
        ctx.txt = ctx.img = ctx.share = None
        return (None, None, None, None, None) + tuple(outs)


# ---------------------------------------------------------------------------------------------
class LinearFn(GradAwareFunction):
    """y = act(x W^T + b) on bf16 rows; act in {None, 'gelu'}; W f32 [N,K] master weight."""

    @staticmethod
    def forward(ctx, x, weight, bias, act, cache):
        x2 = x.reshape(-1, x.shape[-1])
        if x2.stride(1) != 1 or (x2.stride(0) % 8) or (x2.data_ptr() % 16):
            x2 = x2.contiguous()
        cache = cache.for_device(weight.device)
        if cache.stale([weight]):
            cache.t["w"], cache.t["wt"] = cast_weight(weight)
        b = _f32(bias) if bias is not None else None
        N = weight.shape[0]
        if act == "gelu":
            u, y = hip.gemm_nt(x2, cache.t["w"], hip.EPI_BIAS_GELU, bias=b, n=N)
        else:
            u, y = None, hip.gemm_nt(x2, cache.t["w"], hip.EPI_BIAS, bias=b, n=N)
        ctx.save = (x2, u, cache, weight.shape, bias is not None)
        ctx.needs = (x.requires_grad, weight.requires_grad, bias is not None and bias.requires_grad)
        ctx.xshape = x.shape
        return y.view(x.shape[:-1] + (N,))

    @staticmethod
    def backward(ctx, dy):
        x2, u, cache, wshape, has_bias = ctx.save
        N, K = wshape
        dy2 = dy.reshape(-1, N)
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        dev = dy.device
        db = torch.zeros(N, device=dev, dtype=torch.float32) if has_bias else None
        wt = cache.t["wt"]  # bf16 [K, pad8(N)]
        Np = wt.shape[1]
        if Np != N:
            pad = torch.zeros((dy2.shape[0], Np), device=dev, dtype=torch.bfloat16)
            pad[:, :N] = dy2
            dy2 = pad
        if u is not None:
            # dy2 is d(gelu(u)); u holds gelu'(u) saved by the forward epilogue (head-sized rows:
            # a torch multiply is enough here, the encoder layers use the fused GEMM epilogue)
            du = torch.zeros((dy2.shape[0], Np), device=dev, dtype=torch.bfloat16)
            du[:, :N] = dy2[:, :N] * u
            dy2 = du
        dx = hip.gemm_nt(dy2, wt, hip.EPI_ADD, n=K) if ctx.needs[0] else None
        dw = None
        if ctx.needs[1]:
            dw = torch.zeros((N, x2.shape[1]), device=dev, dtype=torch.float32)
            hip.gemm_tn(dy2, x2, dw, n=N, colsum=db)   # bias gradient rides on the weight gradient
            dw = dw[:, :K]
        elif db is not None:
            hip.colsum(dy2, db, n=N)
        if dx is not None:
            dx = dx.view(ctx.xshape)
        return dx, dw, (db if ctx.needs[2] else None), None, None


class LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, weight, bias, eps):
        z2 = z.reshape(-1, z.shape[-1]).contiguous()
        w, b = _f32(weight), _f32(bias)
        y, mean, rstd = hip.layernorm_fwd(z2, w, b, eps)
        ctx.save = (z2, mean, rstd, w)
        return y.view(z.shape)

    @staticmethod
    def backward(ctx, dy):
        z2, mean, rstd, w = ctx.save
        H = z2.shape[1]
        dg = torch.zeros(H, device=dy.device, dtype=torch.float32)
        db = torch.zeros(H, device=dy.device, dtype=torch.float32)
        dz, _ = hip.layernorm_bwd(dy.reshape(-1, H).contiguous(), z2, mean, rstd, w, dg, db)
        return dz.view(dy.shape), dg, db, None


class DecoderCEFn(GradAwareFunction):
    """loss = mean_{labels>=0} CE(h W^T + b, labels); W f32 [V,H] (vocabulary decoder).
    Returns (loss, logits f32 [M,V] view) — logits are a non-differentiable side output.

    want_scores=False (the training path when nobody reads prediction_scores): the fused kernels
    mvptr_decoder_ce_fwd / _bwd — the [M,V] f32 logits never reach HBM; forward keeps per-strip
    log-sum-exp partials, backward recomputes the logits inside the GEMM whose epilogue writes
    softmax - onehot in bf16.  The second output is then an empty [0, V] tensor."""

    @staticmethod
    def forward(ctx, h, weight, bias, labels, cache, want_scores=True):
        M, H = h.shape
        V = weight.shape[0]
        Vp = pad8(V)
        cache = cache.for_device(weight.device)
        if cache.stale([weight]):
            cache.t["w"], cache.t["wt"] = cast_weight(weight)
        h = h.contiguous()
        labels = labels.contiguous()
        nvalid = (labels >= 0).sum().clamp(min=1).to(torch.float32)
        ctx.needs = (h.requires_grad, weight.requires_grad, bias.requires_grad)
        if not want_scores:
            b32 = _f32(bias)
            loss_row, lse = hip.decoder_ce_fwd(h, cache.t["w"], b32, labels, V)
            loss = loss_row.sum() / nvalid
            ctx.save = (h, None, labels, lse, nvalid, cache, V, Vp, b32)
            out_logits = torch.zeros((0, V), device=h.device, dtype=torch.float32)
            ctx.mark_non_differentiable(out_logits)
            return loss, out_logits
        logits = torch.empty((M, Vp), device=h.device, dtype=torch.float32)
        hip.gemm_nt(h, cache.t["w"], hip.EPI_F32, bias=_f32(bias), out=logits, n=V)
        loss_row, lse = hip.ce_fwd(logits, labels, V=V)
        loss = loss_row.sum() / nvalid
        ctx.save = (h, logits, labels, lse, nvalid, cache, V, Vp, None)
        out_logits = logits[:, :V]
        ctx.mark_non_differentiable(out_logits)
        return loss, out_logits

    @staticmethod
    def backward(ctx, gloss, _glogits):
        h, logits, labels, lse, nvalid, cache, V, Vp, b32 = ctx.save
        scale = (gloss.to(torch.float32) / nvalid).reshape(1).contiguous()
        if logits is None:
            d = hip.decoder_ce_bwd(h, cache.t["w"], b32, labels, lse, scale, V, Vp)
        else:
            d = hip.ce_bwd(logits, labels, lse, scale, V, Vp)
        H = h.shape[1]
        dh = hip.gemm_nt(d, cache.t["wt"], hip.EPI_ADD, n=H) if ctx.needs[0] else None
        dw = db = None
        if ctx.needs[2]:
            db = torch.zeros(V, device=h.device, dtype=torch.float32)
        if ctx.needs[1]:
            dw = torch.zeros((V, H), device=h.device, dtype=torch.float32)
            hip.gemm_tn(d, h, dw, n=V, colsum=db)
        elif db is not None:
            hip.colsum(d, db, n=V)
        ctx.save = None
        return dh, dw, db, None, None, None
