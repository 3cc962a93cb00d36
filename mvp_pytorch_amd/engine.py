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


# Gradient arena (mvp_pytorch_amd.dp.GradSync): the autograd functions below let the kernels accumulate weight gradients
# straight into the arena views that are the parameters' .grad and report `delivered(p)` instead of handing autograd a
# fresh tensor per parameter (a zero fill, a copy and an AccumulateGrad add per gradient otherwise).  The arena is found
# PER PARAMETER (each GradSync registers the parameters it lays out), so several models with their own GradSync can live
# in one process; a parameter no GradSync has claimed takes the plain autograd path.  Held weakly: a sink dies with its
# owner, an entry with its parameter.
_SINKS = {}        # id(parameter) -> (weakref(parameter), weakref(sink))
_last_sink_ref = [None]


def set_grad_sink(sink, params=None):
    """Register `sink` (a dp.GradSync) as the gradient arena of `params` (default: sink.params); sink=None drops every
    registration (tests)."""
    import weakref
    if sink is None:
        _SINKS.clear()
        _last_sink_ref[0] = None
        return
    sref = weakref.ref(sink)
    for p in (sink.params if params is None else params):
        key = id(p)
        _SINKS[key] = (weakref.ref(p, lambda _r, k=key: _SINKS.pop(k, None)), sref)
    _last_sink_ref[0] = sref


def grad_sink(p=None):
    """The arena that lays out parameter p, or None.  Without an argument: the most recently registered live arena
    (kept for tools and tests that have exactly one)."""
    if p is None:
        r = _last_sink_ref[0]
        return None if r is None else r()
    ent = _SINKS.get(id(p))
    if ent is None or ent[0]() is not p:
        return None
    return ent[1]()


def note_uses(ctx, params, first_index):
    """Called from an autograd function's forward: tell the gradient arena which parameters this call will deliver a
    gradient for (params[i] is input first_index + i of the function; ctx.needs_input_grad says whether autograd wants
    it — all False under no_grad), see dp.GradSync.note_use."""
    needs = ctx.needs_input_grad
    for i, p in enumerate(params):
        if p is not None and needs[first_index + i]:
            sink = grad_sink(p)
            if sink is not None:
                sink.note_use(p)


def grad_buffer(p, shape=None):
    """-> (f32 buffer the kernels ACCUMULATE p's gradient into, True when it is the arena view p.grad)."""
    sink = grad_sink(p) if p is not None else None
    if sink is not None and p.requires_grad and p.dtype == torch.float32:
        g = sink.direct(p)
        if g is not None and g.is_contiguous() and (shape is None or tuple(g.shape) == tuple(shape)):
            return g, True
    return torch.zeros(tuple(p.shape) if shape is None else tuple(shape), device=p.device, dtype=torch.float32), False


def grad_result(p, buf, direct):
    """What backward returns for p: None when the kernels wrote into the arena (the sink is told), else the buffer."""
    if direct:
        grad_sink(p).delivered(p)
        return None
    return buf


# bf16 working copies that the fused optimizer keeps current (optimization.AdamW writes them in the same pass
# that updates the f32 master weight, mvptr_adamw_mirror_multi): parameter -> where its copies live and which
# WeightCache they belong to.  Keyed by id() (tensors do not work as weak dictionary keys: == is elementwise).
_MIRRORS = {}


def register_mirror(p, cache, dst=None, dst_t=None, col_off_t=0, dst_f32=None):
    import weakref
    if not (isinstance(p, torch.Tensor) and p.dtype == torch.float32 and p.is_contiguous() and p.dim() in (1, 2)):
        return
    key = id(p)
    _MIRRORS[key] = dict(ref=weakref.ref(p, lambda _r, k=key: _MIRRORS.pop(k, None)), cache=weakref.ref(cache),
                         dst=dst, dst_t=dst_t, col_off_t=col_off_t, dst_f32=dst_f32)
    cache._mirror_ids.add(key)


def mirror_of(p):
    m = _MIRRORS.get(id(p))
    if m is None or m["ref"]() is not p:
        return None
    return m


_seed_counter = [0x5DEECE66D]


def next_seed():
    """Fresh 64-bit dropout seed; derived from torch's RNG so torch.manual_seed controls it."""
    with _state_lock:
        _seed_counter[0] = (_seed_counter[0] * 6364136223846793005 + 1442695040888963407) & 0xFFFFFFFFFFFFFFFF
        base = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
        return (_seed_counter[0] ^ (base << 20)) & 0xFFFFFFFFFFFFFFFF


_ARANGES = {}


def arange(n, device, dtype=torch.long, step=1, floor_div=1):
    """(torch.arange(n) // floor_div) * step on `device`, made once per (device, n, dtype, step, floor_div) and shared: position
    ids, [CLS] row indices, pseudo-labels of the contrastive loss, index vectors of the WRA draws, the 0 / 1 labels of the
    matched / hard pairs (floor_div = n over 2 n entries) were a dozen tiny launches per step.  READ-ONLY for the callers
    (every use in this package indexes with it or feeds it to an out-of-place op)."""
    d = torch.device(device)
    if int(floor_div) <= 0:
        raise ValueError("engine.arange: floor_div must be positive (got %r)" % (floor_div,))
    key = (d.type, d.index if d.index is not None or d.type != "cuda" else torch.cuda.current_device(), int(n), dtype, int(step),
           int(floor_div))
    t = _ARANGES.get(key)
    if t is None:
        with _state_lock:
            t = _ARANGES.get(key)
            if t is None:
                # made outside inference mode (an entry first created under torch.inference_mode() could not be saved for
                # backward by a later training step) and complete before any stream may read it: the tensor is created on
                # whichever stream is current and then shared with the side stream
                with torch.inference_mode(False), torch.no_grad():
                    t = torch.arange(int(n), device=d, dtype=dtype)
                    if floor_div != 1:
                        t = torch.div(t, int(floor_div), rounding_mode="floor")
                    if step != 1:
                        t = t * step
                if t.is_cuda:
                    torch.cuda.current_stream(t.device).synchronize()     # once per entry and process
                _ARANGES[key] = t
    return t


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
        self._fresh_once = False  # set by a prefetch / the fused optimizer: the next stale() of the same parameters answers "fresh" once
        self._prefs = None        # weak references to the parameters of the last stale() call (mark_fresh recomputes the key)
        self._mirror_ids = set()  # parameters whose copies in this cache the fused optimizer can write (register_mirror)
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
        if self._prefs is None or len(self._prefs) != len(params) or any(r() is not p for r, p in zip(self._prefs, params)):
            import weakref
            self._prefs = [weakref.ref(p) for p in params]
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

    def mark_fresh(self):
        """The copies were just rewritten from the current parameter values by the optimizer kernel: the next
        stale() of the same parameters answers "fresh" once (later ones fall back to the usual rules)."""
        if self._prefs is None:
            return
        ps = [r() for r in self._prefs]
        if any(p is None for p in ps):
            return
        self._key = tuple((p.data_ptr(), p._version) for p in ps)
        self._fresh_once = True

    def weight_copies(self, weight, want_t=True, kpad=None):
        """bf16 [N, Kp] (+ transposed [K, Np]) copies of one f32 [N, K] weight in PERSISTENT buffers (registered as
        mirrors, so the fused optimizer refreshes them in its own pass) -> (w, wt or None)."""
        if self.stale([weight]):
            N, K = weight.shape
            Kp = pad8(K) if kpad is None else kpad
            w, wt = self.t.get("w"), self.t.get("wt")
            if w is None or w.shape != (N, Kp) or w.device != weight.device:
                w = torch.empty((N, Kp), device=weight.device, dtype=torch.bfloat16)
                alloc_t = torch.empty if N % 8 == 0 else torch.zeros   # only pad columns need the zeros
                wt = alloc_t((K, pad8(N)), device=weight.device, dtype=torch.bfloat16) if want_t else None
                self.t["w"], self.t["wt"] = w, wt
                if weight.dtype == torch.float32 and weight.is_contiguous():
                    register_mirror(weight, self, dst=w, dst_t=wt)
            hip.cast_pack(_f32(weight), dst=w, dst_t=wt)
        return self.t["w"], self.t.get("wt")

    def invalidate(self):
        self._key = None
        self._fresh_once = False
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
                register_mirror(w, self.cache, dst=w_qkv[i * H:(i + 1) * H], dst_t=w_qkv_t, col_off_t=i * H)
                register_mirror(b, self.cache, dst_f32=b_qkv[i * H:(i + 1) * H])
            w_o, w_o_t = torch.empty((H, H), device=dev, dtype=bf), torch.empty((H, H), device=dev, dtype=bf)
            w_i, w_i_t = torch.empty((I, H), device=dev, dtype=bf), torch.empty((H, I), device=dev, dtype=bf)
            w_out, w_out_t = torch.empty((H, I), device=dev, dtype=bf), torch.empty((I, H), device=dev, dtype=bf)
            plan.add(ow.data, dst=w_o, dst_t=w_o_t)
            plan.add(iw.data, dst=w_i, dst_t=w_i_t)
            plan.add(pw.data, dst=w_out, dst_t=w_out_t)
            register_mirror(ow, self.cache, dst=w_o, dst_t=w_o_t)
            register_mirror(iw, self.cache, dst=w_i, dst_t=w_i_t)
            register_mirror(pw, self.cache, dst=w_out, dst_t=w_out_t)
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

    def fold_tables(self, params):
        """Operands of the LayerNorm-folded inference path (encoder_infer_folded), per layer: the weight whose input is a LayerNorm
        output with that LayerNorm's gamma multiplied into its columns (bf16), c = its row sums, d = W beta + b — Q|K|V of layer
        l >= 1 against LayerNorm 2 of layer l - 1, FFN1 against LayerNorm 1 of its own layer.  Rebuilt when a parameter changes."""
        key = tuple((p.data_ptr(), p._version) for p in params)
        if getattr(self, "_fold_key", None) != key:
            bf = torch.bfloat16
            tabs = []
            with torch.no_grad():
                for li in range(len(self.packs)):
                    (qw, qb, kw, kb, vw, vb, _ow, _ob, g1, b1, iw, ib) = params[16 * li:16 * li + 12]
                    t = {}
                    if li > 0:
                        g_prev, b_prev = params[16 * (li - 1) + 14].float(), params[16 * (li - 1) + 15].float()
                        w = torch.cat([qw, kw, vw], 0).float()
                        wf = (w * g_prev[None, :]).to(bf).contiguous()
                        t["w_qkv"], t["c_qkv"] = wf, wf.float().sum(1).contiguous()
                        t["d_qkv"] = (w @ b_prev + torch.cat([qb, kb, vb]).float()).contiguous()
                    w = iw.float()
                    wf = (w * g1.float()[None, :]).to(bf).contiguous()
                    t["w_i"], t["c_i"], t["d_i"] = wf, wf.float().sum(1).contiguous(), (w @ b1.float() + ib.float()).contiguous()
                    tabs.append(t)
            self._fold, self._fold_key = tabs, key
        return self._fold

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

    def __init__(self, packs, B, L, H, heads, I, eps, training, p_hidden, p_attn, seq_start=None, seq_len=None, rows=0,
                 first=0, count=None, all_params=None, rows_dev=None, rows_plan=0):
        """seq_start / seq_len (device int32 [B]) + rows: row-packed mode — x holds `rows` valid token
        rows, sequence b at [seq_start[b], +seq_len[b]), L = the longest sequence.
        first / count: run layers [first, first + count) of the stack only (return_at_layer, vl:162-163); the
        parameters handed to EncoderFn are then those layers' 16 * count, and all_params names the WHOLE stack's
        parameters: a segment refreshes the stack-wide working copies (the buffers the fused optimizer writes) and
        uses its slice of them — never per-layer copies of its own, which a later whole-stack call would not see
        (ADVICE r03)."""
        whole = first == 0 and (count is None or count == len(packs))
        self.group = getattr(packs, "group", None)
        self.segment = None
        if not whole:
            n_seg = len(packs) - first if count is None else count
            if self.group is None or all_params is None:
                self.group = None                                           # per-layer LayerPack path
            else:
                self.segment = (first, n_seg, list(all_params))
            packs = list(packs)[first:first + n_seg]
        self.packs, self.B, self.L, self.H, self.heads, self.I = packs, B, L, H, heads, I
        self.seq_start, self.seq_len, self.rows = seq_start, seq_len, rows
        # rows_dev (device integer tensor, first 4 bytes = the count as int32): the rows actually present; `rows` is then
        # the bound the buffers are sized for and rows_plan the host-side planning hint (mvptr_layer_desc.rows_dev / M_plan)
        self.rows_dev, self.rows_plan = rows_dev, int(rows_plan)
        self.eps, self.training, self.p_hidden, self.p_attn = eps, training, p_hidden, p_attn
        # weight gradients of the whole stack in ONE balanced launch at the end of its backward pass (mvptr_gemm_tn_stack)
        # instead of two grouped launches per layer; False restores the per-layer launches
        self.defer_wgrad = DEFER_WGRAD
        # gelu'(u) stash of the FFN: 8-bit fixed point (default) or bf16 (config.gelu_stash = "bf16": the format of rounds 1-3,
        # for reference-numerics runs and A/B runs; set by the encoder module)
        self.stash_bf16 = GELU_STASH_BF16
        # another stack's GEMMs run beside this one on a second stream (the two uni-modal stacks of the two-stage model): the
        # GEMM tiles keep their full height (mvptr_layer_desc.beside); set by BiBertImgModel
        self.beside = False


# gelu'(u) stash format where no config says otherwise (LinearFn of the head transforms reads it too): False = 8-bit fixed
# point (dithered rounding, |error| < 0.005 with zero mean: round 5), True = bf16.  config.gelu_stash = "bf16" | "u8" sets it per model (modeling_utils).
GELU_STASH_BF16 = False
# Default of EncoderMeta.defer_wgrad (A/B switch: bench.py --wgrad-per-layer, tests).
DEFER_WGRAD = True
# Upper bound on the bytes the deferred mode may keep alive per stack (None: only an allocation failure ends it).
DEFER_WGRAD_MAX_BYTES = None
# CUs the stack-wide weight-gradient launch leaves free (0: it takes every CU).  Its workgroups keep their CU for the whole
# launch (0.6 - 3 ms): dp.GradSync sets this in multi-rank RCCL jobs so that the collectives' kernels find a CU meanwhile.
WGRAD_RESERVE_CUS = 0


def _thresh(p):
    return int(round(float(p) * 65536.0))


# Side streams used for independent sub-networks (text / visual stack); mvp_pytorch_amd.dp waits on
# them before a gradient bucket is handed to RCCL.
SIDE_STREAMS = {}
def side_stream(device):
    """One extra HIP stream per device, created on first use."""
    key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    with _state_lock:
        if key not in SIDE_STREAMS:
            SIDE_STREAMS[key] = torch.cuda.Stream(device=key)
        return SIDE_STREAMS[key]


class AsyncCounts:
    """A few device-side integers (row counts that size later tensors) on their way to the host.
    `tolist()` on the device tensor would wait for EVERYTHING queued on the stream and hand the GPU an
    empty queue; here the copy goes to pinned memory right away and is awaited through an event, so
    the kernels queued after it (embeddings, index_select / cat of the hard batch) keep the GPU busy
    while the host reads the counts and starts queueing the next stage."""
    _pins = {}

    def __init__(self, values):
        self._n = len(values)
        vals = [v.reshape(()) for v in values]
        if vals and vals[0].is_cuda and torch.cuda.is_current_stream_capturing():
            raise RuntimeError("a device -> host count read-back cannot be part of a captured HIP graph (train.GraphedStep needs batches "
                               "that carry host_counts)")
        if vals and vals[0].is_cuda:
            # every count copy carries the device error word along (hip.device_error_word): the host's next look at the device
            # is also its look at the data-dependent checks queued since the last one
            vals.append(hip.device_error_word(vals[0].device)[0])
        t = torch.stack([v.to(torch.int64) for v in vals])
        if not t.is_cuda:
            self._host, self._event = t, None
            return
        self._device = t.device
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

    def ready(self):
        """True once the copy has landed (get() then returns without waiting)."""
        return self._event is None or self._event.query()

    def get(self):
        if self._event is not None:
            self._event.synchronize()
        out = [int(v) for v in self._host.tolist()]
        if len(out) > self._n:
            if out[self._n]:
                hip.raise_device_error(out[self._n], self._device)
            out = out[:self._n]
        return out


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


def fold_eligible(meta, params):
    """the LayerNorm-folded inference path takes whole stacks of contiguous f32 parameters whose GEMMs fit mvptr_gemm_nt_ln"""
    return (meta.group is not None and meta.segment is None and len(meta.packs) >= 1 and hip.gemm_nt_ln_eligible(meta.H, meta.H) and
            hip.gemm_nt_ln_eligible(meta.I, meta.H) and hip.gemm_nt_ln_eligible(meta.H, meta.I) and
            all(p.dtype == torch.float32 and p.is_contiguous() for p in params))


@torch.no_grad()
def encoder_infer_folded(x, mask_add, meta, params):
    """Forward of a whole encoder stack with every LayerNorm folded into the GEMMs around it (inference only; north_star's "fused
    LayerNorm + QKV projection", mvptr_gemm_nt_ln): the GEMM in front of a LayerNorm writes the pre-LayerNorm rows z and the partial
    sums of their statistics, the GEMM behind it (Q|K|V of the next layer, FFN1) multiplies z by the gamma-scaled weight and applies
    (mean, rstd) in its epilogue; the residual adds normalise their rows on the fly.  Per stack one LayerNorm launch is left (the
    output) instead of 2 per layer, and no normalised tensor goes through HBM in between.  x: bf16 [rows, H] (row-packed with
    meta.seq_start / seq_len, or B * L padded rows with mask_add) -> bf16 [rows, H].  Equal to the unfused path up to bf16
    rounding (the normalised rows are not rounded to bf16 on their way into the GEMM; W' = bf16(gamma o W) instead of bf16(W))."""
    n = len(meta.packs)
    meta.group.refresh(params)                  # bf16 working copies current (pk.keep)
    tabs = meta.group.fold_tables(params)
    H, heads, eps = meta.H, meta.heads, meta.eps
    z_prev = st_prev = g_prev = b_prev = None
    for li in range(n):
        w_qkv, _, b_qkv, w_o, _, _w_i, _, w_out, _ = meta.packs[li].keep
        prm = params[16 * li:16 * (li + 1)]
        ob, g1, b1, pb, g2, b2 = prm[7], prm[8], prm[9], prm[13], prm[14], prm[15]
        t = tabs[li]
        if z_prev is None:
            qkv = hip.gemm_nt(x, w_qkv, hip.EPI_BIAS, bias=b_qkv)
        else:
            qkv = hip.gemm_nt_ln(z_prev, t["w_qkv"], hip.LN_FOLD_BIAS, t["d_qkv"], stats=st_prev, colsum=t["c_qkv"])
        if meta.rows:
            ctx, _ = hip.attention_fwd_packed(qkv, meta.seq_start, meta.seq_len, meta.B, meta.L, heads, need_lse=False)
        else:
            ctx, _ = hip.attention_fwd(qkv, mask_add, meta.B, meta.L, heads, need_lse=False)
        if z_prev is None:
            z1, part = hip.gemm_nt_ln(ctx, w_o, hip.LN_RESID_STATS, ob, aux=x)
        else:
            z1, part = hip.gemm_nt_ln(ctx, w_o, hip.LN_RESID_STATS, ob, aux=z_prev, stats=st_prev, gamma=g_prev, beta=b_prev)
        st1 = hip.ln_stats_finalize(part, H, eps)
        a = hip.gemm_nt_ln(z1, t["w_i"], hip.LN_FOLD_GELU, t["d_i"], stats=st1, colsum=t["c_i"])
        z2, part = hip.gemm_nt_ln(a, w_out, hip.LN_RESID_STATS, pb, aux=z1, stats=st1, gamma=g1, beta=b1)
        st_prev = hip.ln_stats_finalize(part, H, eps)
        z_prev, g_prev, b_prev = z2, g2, b2
    return hip.layernorm_fwd(z_prev, g_prev, b_prev, eps, save_stats=False)[0]


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
        if meta.group is not None and meta.segment is not None:
            first, n_seg, all_params = meta.segment
            lws = meta.group.refresh(all_params)[first:first + n_seg]
        elif meta.group is not None:
            lws = meta.group.refresh(params)
        else:
            lws = [meta.packs[li].refresh(params[16 * li:16 * (li + 1)]) for li in range(n)]
        for li in range(n):
            lw = lws[li]
            d = hip.LayerDesc(meta.B, meta.L, meta.H, meta.heads, meta.I, meta.eps,
                              1 if meta.training else 0, _thresh(meta.p_hidden), _thresh(meta.p_attn),
                              next_seed() if meta.training else 0, meta.rows, min(meta.rows_plan, meta.rows) if meta.rows_dev is not None else 0,
                              meta.seq_start.data_ptr() if meta.rows else None,
                              meta.seq_len.data_ptr() if meta.rows else None,
                              meta.rows_dev.data_ptr() if (meta.rows and meta.rows_dev is not None) else None,
                              1 if getattr(meta, "stash_bf16", False) else 0, 1 if getattr(meta, "beside", False) else 0)
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
        note_uses(ctx, params, 3)
        return cur

    # order of a layer's sixteen parameters (LayerPack.NAMES positions) inside its gradient arena = the order of
    # mvptr_layer_grads: [Wq;Wk;Wv] [bq;bk;bv] Wo bo ln1.g ln1.b Wi bi Wout bout ln2.g ln2.b
    ARENA_ORDER = (0, 2, 4, 1, 3, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15)

    @staticmethod
    def backward(ctx, dy):
        lib = hip.load()
        meta = ctx.meta
        n = len(meta.packs)
        dev = dy.device
        H, I = meta.H, meta.I
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        ws_bytes = lib.mvptr_layer_workspace_bytes(ctypes.byref(ctx.descs[0]))
        # Deferred weight gradients: every layer keeps its dY operands in a workspace of its own and the whole stack's
        # problems (4 per layer) go out as one balanced launch after the last layer (mvptr_gemm_tn_stack)
        defer = bool(getattr(meta, "defer_wgrad", False)) and n >= 1
        ws_step = (ws_bytes + 255) // 256 * 256
        # one workspace per layer stays alive until the stack launch (~18 KB per row per layer: 4 GB for six layers at
        # M = 37 748).  A stack that does not get that memory takes the per-layer path, as before round 5.
        ws = None
        if defer and DEFER_WGRAD_MAX_BYTES is not None and ws_step * n > DEFER_WGRAD_MAX_BYTES:
            defer = False
        if defer:
            try:
                ws = torch.empty(ws_step * n, device=dev, dtype=torch.uint8)
            except torch.OutOfMemoryError:
                if torch.cuda.is_current_stream_capturing():
                    raise
                defer = False
        if ws is None:
            ws = torch.empty(ws_bytes, device=dev, dtype=torch.uint8)
        EncoderFn.last_backward_deferred = defer
        probs = (hip.TnProblem * (4 * n))() if defer else None
        n_probs = 0
        delivered_later = []
        convert_later = []
        sizes = [3 * H * H, 3 * H, H * H, H, H, H, I * H, I, H * I, H, H, H]
        total = sum(sizes)
        grads = [None] * (16 * n)
        sink = grad_sink(ctx.params[0]) if ctx.params else None
        d_cur = dy.contiguous()
        for li in reversed(range(n)):
            ps = ctx.params[16 * li:16 * (li + 1)]
            arena_ps = [ps[j] for j in EncoderFn.ARENA_ORDER]
            # the layer's gradient arena: the sixteen .grad views back to back (dp.GradSync lays encoder layers out
            # that way) -> the kernels accumulate into the communication / optimizer buffer itself; otherwise a
            # zero-filled scratch arena whose pieces are handed to autograd
            flat = None
            if sink is not None and all(p.requires_grad and p.dtype == torch.float32 for p in ps):
                flat = sink.arena(arena_ps)
            direct = flat is not None
            if not direct:
                flat = torch.zeros(total, device=dev, dtype=torch.float32)
            base = flat.data_ptr()
            g = hip.LayerGrads()
            offs, o = [], 0
            for sz in sizes:
                offs.append(base + 4 * o)
                o += sz
            (g.w_qkv, g.b_qkv, g.w_o, g.b_o, g.ln1_g, g.ln1_b, g.w_i, g.b_i, g.w_out, g.b_out, g.ln2_g, g.ln2_b) = offs
            dx = torch.empty_like(d_cur)
            if defer:
                cnt = ctypes.c_int(0)
                sub = ctypes.cast(ctypes.byref(probs, n_probs * ctypes.sizeof(hip.TnProblem)), ctypes.POINTER(hip.TnProblem))
                hip._check(lib.mvptr_encoder_layer_bwd_defer(ctypes.byref(ctx.descs[li]), ctypes.byref(meta.packs[li].w),
                                                             hip._p(ctx.xs[li]), hip._p(ctx.mask), hip._p(ctx.stashes[li]),
                                                             hip._p(d_cur), hip._p(dx), ctypes.byref(g),
                                                             ctypes.c_void_p(ws.data_ptr() + li * ws_step), ws_bytes, sub,
                                                             ctypes.byref(cnt), stream))
                n_probs += cnt.value
            else:
                hip._check(lib.mvptr_encoder_layer_bwd(ctypes.byref(ctx.descs[li]), ctypes.byref(meta.packs[li].w),
                                                       hip._p(ctx.xs[li]), hip._p(ctx.mask), hip._p(ctx.stashes[li]),
                                                       hip._p(d_cur), hip._p(dx), ctypes.byref(g), hip._p(ws), ws_bytes, stream))
            if direct and defer:
                delivered_later.extend(ps)     # complete only after the stack-wide launch below
            elif direct:
                for p in ps:
                    sink.delivered(p)
            else:
                parts, o = [], 0
                for sz in sizes:
                    parts.append(flat[o:o + sz])
                    o += sz
                wq = parts[0].view(3, H, H)
                bq = parts[1].view(3, H)
                gl = [wq[0], bq[0], wq[1], bq[1], wq[2], bq[2], parts[2].view(H, H), parts[3], parts[4], parts[5],
                      parts[6].view(I, H), parts[7], parts[8].view(H, I), parts[9], parts[10], parts[11]]
                for j in range(16):
                    p = ps[j]
                    if p.requires_grad:
                        # deferred mode: the four weight gradients and two column-sum bias gradients of this scratch arena are only
                        # written by the stack-wide launch below, so a dtype conversion must wait for it
                        grads[16 * li + j] = gl[j]
                        if p.dtype != torch.float32:
                            convert_later.append((16 * li + j, p.dtype))
            d_cur = dx
        if defer and n_probs > 0:
            rd = meta.rows_dev.data_ptr() if (meta.rows and meta.rows_dev is not None) else None
            ncu = torch.cuda.get_device_properties(dev).multi_processor_count
            max_wg = max(8, ncu - WGRAD_RESERVE_CUS) if WGRAD_RESERVE_CUS > 0 else 0
            hip._check(lib.mvptr_gemm_tn_stack(probs, n_probs, ctypes.c_void_p(rd) if rd else None, max_wg, stream))
        for p in delivered_later:
            sink.delivered(p)
        for k, dt in convert_later:
            grads[k] = grads[k].to(dt)
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
            D = img_w.shape[1]
            Dp = pad8(D)
            if img_feats.dtype == torch.bfloat16 and img_feats.shape[2] == Dp and Dp != D and img_feats.is_contiguous():
                # already the K-padded bf16 GEMM operand (input_pipeline.PretrainBatchStager, features="bf16"): no cast pass
                fb = img_feats.view(B * R, Dp)
            else:
                if img_feats.shape[2] != D:
                    raise RuntimeError("img_feats has %d columns, img_embedding expects %d" % (img_feats.shape[2], D))
                feats = img_feats.reshape(B * R, D)
                if feats.dtype != torch.float32:
                    feats = feats.float()
                fb = torch.empty((B * R, Dp), device=dev, dtype=torch.bfloat16)
                hip.cast_pack(feats.contiguous(), dst=fb)
            cache = meta["cache"]
            cache = cache.for_device(img_w.device)
            w_img, _ = cache.weight_copies(img_w, want_t=False)
            zi = hip.gemm_nt(fb, w_img, hip.EPI_BIAS, bias=_f32(img_b))
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
        ctx.ps = (word, pos, typ, ln_w, ln_b, img_w, img_b, img_ln_w, img_ln_b)
        note_uses(ctx, (word, pos, typ, ln_w, ln_b, None, img_b if R > 0 else None, img_ln_w if (R > 0 and use_ln) else None,
                        img_ln_b if (R > 0 and use_ln) else None), 5)
        return out.view(B, Ltot, H)

    @staticmethod
    def backward(ctx, dout):
        B, Lt, R, H = ctx.dims
        Ltot = Lt + R
        dev = dout.device
        dout = dout.contiguous().view(B * Ltot, H)
        idf, tyf, pof, z, mean, rstd, lnw, drop_t = ctx.txt
        f32 = dict(device=dev, dtype=torch.float32)
        word, pos, typ, ln_w, ln_b, img_w, img_b, img_ln_w, img_ln_b = ctx.ps
        needs = ctx.needs

        def buf(p, need, shape=None):
            # arena view (accumulated into by the kernels) or a zero-filled scratch tensor handed to autograd
            return grad_buffer(p, shape) if need else (torch.zeros(tuple(p.shape) if shape is None else shape, **f32), False)

        (dg, dg_d), (db, db_d) = buf(ln_w, needs[3]), buf(ln_b, needs[4])
        dz, _ = hip.layernorm_bwd(dout, z, mean, rstd, lnw, dg, db, None, rows_per_group=Lt,
                                  group_stride=Ltot, row_offset=0, y_drop=drop_t)
        # The word table (86 051 x 768 f32 = 264 MB) is looked up twice per forward pass (text ids, tag
        # ids).  With a gradient arena both backward calls scatter into the table's arena view; without
        # one they share ONE zero-filled buffer (meta["share"], created per forward pass by the backbone)
        # and the second call returns no gradient of its own, which saves a 264-MB fill and the 0.8-GB add
        # autograd would use to sum two dense gradients.
        share = ctx.share
        dword, dword_d = grad_buffer(word) if needs[0] else (None, False)
        first = True
        if not dword_d:
            dword = share.get("dword") if share is not None else None
            first = dword is None
            if first:
                dword = torch.zeros(ctx.shapes[0], **f32)
                if share is not None:
                    share["dword"] = dword
        (dpos, dpos_d), (dtyp, dtyp_d) = buf(pos, needs[1]), buf(typ, needs[2])
        hip.embed_bwd(idf, pof, tyf, dz, dword, dpos, dtyp)
        if not first:
            dword = None
        gi_w = gi_b = gi_lw = gi_lb = None
        gb_d = glw_d = glb_d = False
        if ctx.img is not None:
            fb, zi, mi, ri, g, drop_i, D = ctx.img
            gi_b, gb_d = buf(img_b, needs[6])
            if g is not None:
                (gi_lw, glw_d), (gi_lb, glb_d) = buf(img_ln_w, needs[7]), buf(img_ln_b, needs[8])
            dzi, _ = hip.layernorm_bwd(dout, zi, mi, ri, g, gi_lw, gi_lb, gi_b, rows_per_group=R,
                                       group_stride=Ltot, row_offset=Lt, y_drop=drop_i)
            dw = torch.zeros((H, fb.shape[1]), **f32)      # K padded to a multiple of 8: not the parameter's layout
            hip.gemm_tn(dzi, fb, dw)
            gi_w = dw[:, :D].contiguous()
        outs = [grad_result(word, dword, dword_d) if needs[0] else None,
                grad_result(pos, dpos, dpos_d) if needs[1] else None, grad_result(typ, dtyp, dtyp_d) if needs[2] else None,
                grad_result(ln_w, dg, dg_d) if needs[3] else None, grad_result(ln_b, db, db_d) if needs[4] else None,
                gi_w if needs[5] else None,
                grad_result(img_b, gi_b, gb_d) if (needs[6] and gi_b is not None) else None,
                grad_result(img_ln_w, gi_lw, glw_d) if (needs[7] and gi_lw is not None) else None,
                grad_result(img_ln_b, gi_lb, glb_d) if (needs[8] and gi_lb is not None) else None]
        ctx.txt = ctx.img = ctx.share = ctx.ps = None
        return (None, None, None, None, None) + tuple(outs)


# ---------------------------------------------------------------------------------------------
class LinearFn(GradAwareFunction):
    """y = act(x W^T + b) on bf16 rows; act in {None, 'gelu', 'gelu16'}; W f32 [N,K] master weight."""

    @staticmethod
    def forward(ctx, x, weight, bias, act, cache):
        x2 = x.reshape(-1, x.shape[-1])
        if x2.stride(1) != 1 or (x2.stride(0) % 8) or (x2.data_ptr() % 16):
            x2 = x2.contiguous()
        cache = cache.for_device(weight.device)
        cache.weight_copies(weight)
        b = _f32(bias) if bias is not None else None
        N = weight.shape[0]
        if act in ("gelu", "gelu16"):       # "gelu16": gelu'(u) stashed in bf16 instead of 8-bit fixed point (config.gelu_stash)
            u, y = hip.gemm_nt(x2, cache.t["w"], hip.EPI_BIAS_GELU_BF16 if (act == "gelu16" or GELU_STASH_BF16) else hip.EPI_BIAS_GELU, bias=b, n=N)
        else:
            u, y = None, hip.gemm_nt(x2, cache.t["w"], hip.EPI_BIAS, bias=b, n=N)
        ctx.save = (x2, u, cache, weight.shape, bias is not None)
        ctx.needs = (x.requires_grad, weight.requires_grad, bias is not None and bias.requires_grad)
        ctx.ps = (weight, bias)
        note_uses(ctx, (weight if x2.shape[1] == weight.shape[1] else None, bias), 1)
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
        weight, bias = ctx.ps
        db, db_d = grad_buffer(bias) if ctx.needs[2] else ((torch.zeros(N, device=dev, dtype=torch.float32) if has_bias else None), False)
        wt = cache.t["wt"]  # bf16 [K, pad8(N)]
        Np = wt.shape[1]
        if Np != N:
            pad = torch.zeros((dy2.shape[0], Np), device=dev, dtype=torch.bfloat16)
            pad[:, :N] = dy2
            dy2 = pad
        if u is not None:
            # dy2 is d(gelu(u)); u holds gelu'(u) saved by the forward epilogue as 8-bit fixed point: one launch (the encoder
            # layers have the multiply in their GEMM epilogue; here the producer of dy2 is a LayerNorm backward)
            if dy2.dtype != torch.bfloat16:
                dy2 = dy2.to(torch.bfloat16)
            dy2 = hip.dgelu_mul(dy2[:, :N], u, Np)
        dx = hip.gemm_nt(dy2, wt, hip.EPI_ADD, n=K) if ctx.needs[0] else None
        dw, dw_d = None, False
        if ctx.needs[1]:
            if x2.shape[1] == K:
                dw, dw_d = grad_buffer(weight, (N, K))
            else:
                dw = torch.zeros((N, x2.shape[1]), device=dev, dtype=torch.float32)
            hip.gemm_tn(dy2, x2, dw, n=N, colsum=db)   # bias gradient rides on the weight gradient
            dw = dw[:, :K] if not dw_d else dw
        elif db is not None:
            hip.colsum(dy2, db, n=N)
        if dx is not None:
            dx = dx.view(ctx.xshape)
        ctx.ps = None
        return (dx, grad_result(weight, dw, dw_d) if ctx.needs[1] else None,
                grad_result(bias, db, db_d) if ctx.needs[2] else None, None, None)


class LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, weight, bias, eps):
        z2 = z.reshape(-1, z.shape[-1]).contiguous()
        w, b = _f32(weight), _f32(bias)
        y, mean, rstd = hip.layernorm_fwd(z2, w, b, eps)
        ctx.save = (z2, mean, rstd, w)
        ctx.ps = (weight, bias)
        note_uses(ctx, (weight, bias), 1)
        return y.view(z.shape)

    @staticmethod
    def backward(ctx, dy):
        z2, mean, rstd, w = ctx.save
        H = z2.shape[1]
        weight, bias = ctx.ps
        (dg, dg_d), (db, db_d) = grad_buffer(weight), grad_buffer(bias)
        dz, _ = hip.layernorm_bwd(dy.reshape(-1, H).contiguous(), z2, mean, rstd, w, dg, db)
        ctx.ps = None
        return dz.view(dy.shape), grad_result(weight, dg, dg_d), grad_result(bias, db, db_d), None


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
        cache.weight_copies(weight)
        h = h.contiguous()
        labels = labels.contiguous()
        nvalid = (labels >= 0).sum().clamp(min=1).to(torch.float32) if want_scores else None
        ctx.needs = (h.requires_grad, weight.requires_grad, bias.requires_grad)
        ctx.ps = (weight, bias)
        note_uses(ctx, (weight, bias), 1)
        if not want_scores:
            b32 = _f32(bias)
            loss_row, lse = hip.decoder_ce_fwd(h, cache.t["w"], b32, labels, V)
            loss, nvalid = hip.masked_mean(loss_row, labels)       # sum / max(#labels >= 0, 1) in one launch
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
        dh = None
        if ctx.needs[0]:
            M = d.shape[0]
            tiles = ((M + 127) // 128) * ((H + 127) // 128)
            if Vp >= 8192 and tiles < 256 and M > 128:
                # few scored rows x the whole vocabulary: 128 x 128 tiles cannot fill the chip and each loops over Vp / 32
                # K-steps -> cut the reduction into slices (one workgroup per tile and slice), add the f32 slabs in order
                splits = max(2, min(8, 768 // tiles))
                dh = hip.gemm_nt_splitk(d, cache.t["wt"], splits, n=H).sum(0).to(torch.bfloat16)
            else:
                dh = hip.gemm_nt(d, cache.t["wt"], hip.EPI_ADD, n=H)
        weight, bias = ctx.ps
        dw = db = None
        dw_d = db_d = False
        if ctx.needs[2]:
            db, db_d = grad_buffer(bias, (V,))
        if ctx.needs[1]:
            dw, dw_d = grad_buffer(weight, (V, H))
            hip.gemm_tn(d, h, dw, n=V, colsum=db)
        elif db is not None:
            hip.colsum(d, db, n=V)
        ctx.save = ctx.ps = None
        return (dh, grad_result(weight, dw, dw_d) if ctx.needs[1] else None,
                grad_result(bias, db, db_d) if ctx.needs[2] else None, None, None, None)


# ---------------------------------------------------------------------------------------------
# Row taps and the f32 B-row heads (heads.hip)
class MultiTapFn(torch.autograd.Function):
    """outs[k] = rows idx_k of a bf16 [R, H] buffer (or of two buffers addressed as one: indices >= src.shape[0]
    read src2), idx_k int32 (negative index = zero row).  ALL the rows that later stages read from one stack
    output go through ONE call — the packed joint input gathered from the packed text and visual outputs, [CLS]
    states, masked-LM rows, phrase / region rows of the word-region alignment — so the backward pass builds each
    buffer's gradient in ONE pass over its rows (hip.tap_rows_bwd: the taps inverted by a counting sort, every row the
    f32 sum of its contributions rounded to bf16 once, untapped rows zero) instead of a full-size zero-filled tensor +
    add per consumer.  Incoming gradients may be bf16 or f32."""

    @staticmethod
    def forward(ctx, src, src2, *idxs):
        ctx.set_materialize_grads(False)
        ctx.idxs = idxs
        ctx.shapes = (src.shape, None if src2 is None else src2.shape)
        return tuple(hip.gather_rows(src, i, src2=src2) for i in idxs)

    @staticmethod
    def backward(ctx, *gs):
        s1, s2 = ctx.shapes
        # one pass over the destination rows: the taps are inverted on the device and every source row sums its own
        # contributions in f32 and is written once, rounded once (hip.tap_rows_bwd); rows nobody tapped become zero rows
        taps = []
        for g, i in zip(gs, ctx.idxs):
            if g is None or i.numel() == 0:
                continue
            g = g.contiguous()
            if g.dtype not in (torch.bfloat16, torch.float32):
                g = g.float()
            taps.append((g, i))
        dev = ctx.idxs[0].device
        if not taps:
            d = torch.zeros(s1, device=dev, dtype=torch.bfloat16)
            d2 = torch.zeros(s2, device=dev, dtype=torch.bfloat16) if s2 is not None else None
        elif len(taps) <= hip.TAP_MAX and s1[1] <= 2048 and s1[1] % 4 == 0:
            d, d2 = hip.tap_rows_bwd(taps, s1[0], s2[0] if s2 is not None else 0, s1[1])
        else:
            # more taps than one call takes (or an odd width): f32 accumulation by atomics, one rounding at the end
            d = torch.zeros(s1, device=dev, dtype=torch.float32)
            d2 = torch.zeros(s2, device=dev, dtype=torch.float32) if s2 is not None else None
            for g, i in taps:
                hip.scatter_add_rows(g, i, d, d2)
            d, d2 = d.to(torch.bfloat16), (d2.to(torch.bfloat16) if d2 is not None else None)
        return (d, d2) + (None,) * len(ctx.idxs)


def tap_rows(src, idx, src2=None):
    """One-output form of MultiTapFn."""
    return MultiTapFn.apply(src, src2, idx)[0]


class SmallLinearFn(GradAwareFunction):
    """y = act(x W^T + b) in f32 on a few hundred rows: x bf16 or f32 [n, K], W f32 [N, K] (w_kn=False, nn.Linear
    layout) or [K, N] (w_kn=True: `x @ W`, the CLIP projections vl:525-526), act in {None, 'tanh'} -> f32 [n, N].
    Pooler (modeling_bert.py:468-474), seq_relationship (vl:975-979), txt_proj / vis_proj."""

    @staticmethod
    def forward(ctx, x, weight, bias, act, w_kn):
        x = x.contiguous()
        w = _f32(weight)
        y = hip.sgemm_small(x, w, trans_b=not w_kn, bias=_f32(bias) if bias is not None else None, act=act)
        ctx.save = (x, w, y if act == "tanh" else None)
        ctx.ps = (weight, bias)
        ctx.w_kn = w_kn
        ctx.needs = (x.requires_grad, weight.requires_grad, bias is not None and bias.requires_grad)
        note_uses(ctx, (weight, bias), 1)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.save
        weight, bias = ctx.ps
        du = dy.contiguous().float()
        if y is not None:
            du = du * (1.0 - y * y)
        dx = dw = db = None
        dw_d = db_d = False
        if ctx.needs[0]:
            dx = hip.sgemm_small(du, w, trans_b=ctx.w_kn)          # [n, K] f32
        if ctx.needs[1]:
            dw, dw_d = grad_buffer(weight)
            if ctx.w_kn:
                hip.sgemm_small(x, du, trans_a=True, out=dw, accumulate=True)      # [K, N] += x^T du
            else:
                hip.sgemm_small(du, x, trans_a=True, out=dw, accumulate=True)      # [N, K] += du^T x
        if ctx.needs[2]:
            db, db_d = grad_buffer(bias)
            db += du.sum(0)
        ctx.save = ctx.ps = None
        return (dx, grad_result(weight, dw, dw_d) if ctx.needs[1] else None,
                grad_result(bias, db, db_d) if ctx.needs[2] else None, None, None)


class L2NormFn(torch.autograd.Function):
    """F.normalize(y, p=2, dim=-1) on f32 rows (vl:525-526)."""

    @staticmethod
    def forward(ctx, y):
        g, inv = hip.l2norm_fwd(y.contiguous())
        ctx.save = (g, inv)
        return g

    @staticmethod
    def backward(ctx, dg):
        g, inv = ctx.save
        return hip.l2norm_bwd(g, inv, dg.contiguous().float())


class SimFn(torch.autograd.Function):
    """sim = gt gi^T in exact f32 (vl:527; feeds the hard-negative argmax :531-534)."""

    @staticmethod
    def forward(ctx, gt, gi):
        gt, gi = gt.contiguous(), gi.contiguous()
        ctx.save = (gt, gi)
        return hip.sgemm_small(gt, gi, trans_b=True)

    @staticmethod
    def backward(ctx, dsim):
        gt, gi = ctx.save
        dsim = dsim.contiguous().float()
        return hip.sgemm_small(dsim, gi), hip.sgemm_small(dsim, gt, trans_a=True)


class ContrastiveLossFn(GradAwareFunction):
    """(CE(sim * exp(logit_scale), arange) + CE(its transpose, arange)) / 2 (vl:1238-1241) -> scalar."""

    @staticmethod
    def forward(ctx, sim, logit_scale):
        sim = sim.contiguous()
        ls = _f32(logit_scale).reshape(1)
        loss, lse = hip.clip_ce_fwd(sim, ls)
        ctx.save = (sim, ls, lse)
        ctx.ps = logit_scale
        ctx.needs = (sim.requires_grad, logit_scale.requires_grad)
        note_uses(ctx, (logit_scale,), 1)
        return loss

    @staticmethod
    def backward(ctx, gloss):
        sim, ls, lse = ctx.save
        p = ctx.ps
        dls, dls_d = (grad_buffer(p) if ctx.needs[1] else (None, False))
        dsim = hip.clip_ce_bwd(sim, ls, lse, gloss.reshape(1).float().contiguous(), dls.view(1) if dls is not None else None)
        ctx.save = ctx.ps = None
        return (dsim if ctx.needs[0] else None), (grad_result(p, dls, dls_d) if ctx.needs[1] else None)


class WraLossFn(torch.autograd.Function):
    """Word-region alignment loss on gathered rows (vl:1285-1300 + get_pos_neg_sims vl:1553-1596 + t2i_sim vl:1543-1550):
    txt bf16 [n, Pw, H] phrase rows, reg bf16 [n, Rw, H] region rows, the reference's draws -> the mean hinge, f32 [] —
    two launches forward, one backward (csrc/wra.hip)."""

    @staticmethod
    def forward(ctx, txt, reg, phrase_index, img_index, pos_pick, neg_pick, neg_img):
        txt, reg = txt.contiguous(), reg.contiguous()
        neg_img = neg_img.contiguous()
        loss, saved = hip.wra_fwd(txt, reg, phrase_index.contiguous(), img_index.contiguous(), pos_pick.contiguous(),
                                  neg_pick.contiguous(), neg_img)
        ctx.save = (txt, reg, neg_img, saved)
        return loss.view(())

    @staticmethod
    def backward(ctx, gloss):
        txt, reg, neg_img, saved = ctx.save
        ctx.save = None
        d_txt, d_reg = hip.wra_bwd(txt, reg, neg_img, saved, gloss)
        return d_txt, d_reg, None, None, None, None, None


class CeMeanFn(torch.autograd.Function):
    """CrossEntropyLoss(ignore_index=-1) over a few classes (the 2-way ITM loss vl:1247-1251): one launch computes the
    mean loss and d loss / d logits."""

    @staticmethod
    def forward(ctx, logits, labels):
        loss, d = hip.ce_mean_small(logits.contiguous().float(), labels.contiguous())
        ctx.save = d
        return loss

    @staticmethod
    def backward(ctx, gloss):
        return ctx.save * gloss, None


class BceLogitsFn(torch.autograd.Function):
    """instance_bce_with_logits (vl:878-883; the VQA loss): binary_cross_entropy_with_logits(mean) * n_classes on f32
    [rows, classes] logits — loss and gradient in one pass over the logits (two launches) instead of torch's chain."""

    @staticmethod
    def forward(ctx, logits, labels):
        loss, d = hip.bce_logits(logits.contiguous().float(), labels.contiguous().float(), want_grad=logits.requires_grad)
        ctx.save = d
        return loss

    @staticmethod
    def backward(ctx, gloss):
        d = ctx.save
        ctx.save = None
        return (d * gloss if d is not None else None), None
