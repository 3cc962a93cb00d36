"""MVPTR model classes with the class / config / checkpoint surface of
oscar/modeling/modeling_vlbert.py, running on the HIP encoder.

  CaptionBertEncoder           vl:123-178      BertImgModel               vl:202-352
  BiBertImgModel               vl:354-874      BertPreTrainingHeads       vl:970-980
  BertImgForPreTraining        vl:1024-1130    BiBertImgForPreTraining    vl:1133-1311
  BiImageBertForRetrieval      vl:1598-1712    BiImageBertForSequenceClassification vl:1715-1798
  BiImageBertForVQA            vl:1801-1870    WRA helpers                vl:1502-1596

(`vl` = oscar/modeling/modeling_vlbert.py in the reference tree.)  Keyword arguments, return
tuples, state_dict keys and config attributes follow the reference; encoder activations are
bf16 tensors, pooled outputs / logits / losses are f32.
"""
import copy
import os
import logging
import math
import random

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn
from torch.nn import CrossEntropyLoss, MSELoss

from .. import engine, hip
from .modeling_bert import (BertConfig, BertEmbeddings, BertLayer, BertLayerNorm, BertLMPredictionHead,
                            BertPooler, BertPreTrainedModel, BertQAPredictionHead, HeadLinear, embed_inputs)
from .modeling_utils import ImgPreTrainedModel

logger = logging.getLogger(__name__)


def soft_cross_entropy(target, input_prob, reduction="mean"):
    """vl:27-40."""
    logprobs = F.log_softmax(input_prob, dim=1)
    target = target.float()
    target = torch.stack([1 - target, target], dim=1)
    batchloss = -torch.sum(target.view(target.shape[0], -1) * logprobs, dim=1)
    if reduction == "none":
        return batchloss
    if reduction == "mean":
        return torch.mean(batchloss)
    if reduction == "sum":
        return torch.sum(batchloss)
    raise NotImplementedError("Unsupported reduction mode.")


def instance_bce_with_logits(logits, labels, reduction="mean", pos_weight=None):
    """vl:878-883."""
    assert logits.dim() == 2
    if (reduction == "mean" and pos_weight is None and logits.is_cuda and logits.dtype == torch.float32 and labels.shape == logits.shape and
            not labels.requires_grad):
        return engine.BceLogitsFn.apply(logits, labels.to(torch.float32))     # mvptr_bce_logits: loss (x classes) + gradient
    # torch path, on purpose: other reductions / pos_weight (not used by run_vqa.py), fp16 logits of model.half() inference, CPU
    # tensors
    loss = F.binary_cross_entropy_with_logits(logits, labels, reduction=reduction, pos_weight=pos_weight)
    if reduction == "mean":
        loss = loss * labels.size(1)
    return loss


def _streams_allowed(setting):
    """Two-stream execution of the uni-modal stacks (and heads beside the joint stack).  Policy since round 4:
      * single-rank process: yes;
      * multi-rank job over RCCL (backend "nccl"): yes — the same compute schedule at every N, so that a weak-scaling series
        compares like with like (one stream costs 2.3 ms of the 26.3-ms step at N = 1).  Nothing about the gradient exchange
        depends on stream timing: the autograd engine walks the backward graph in one fixed CPU-side order on every rank, so
        the buckets' collectives are issued in the same order everywhere, and mvp_pytorch_amd.dp.GradSync makes a bucket wait
        for every stream that produced one of its gradients before it is handed to the collective (tests/test_dp_gpu.py:
        one-rank RCCL group with hooks, bucket launches and two compute streams);
      * multi-rank job over gloo: no — the bucketed exchange slows down badly beside a side stream (host-side copies that
        synchronise the device, tools/dp_gloo_check.py);
      * config.parallel_stacks = "always" forces two streams, "single_rank" restores the round-3 policy (multi-rank jobs on one
        stream whatever the backend; bench.py --dp-one-stream), False keeps one stream everywhere."""
    if setting == "always":
        return True
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return True
    return setting != "single_rank" and dist.get_backend() == "nccl"


def additive_mask(attention_mask):
    """vl:278-292 / vl:430-460 — [B,L] 0/1 -> f32 [B,L] additive (0 / -10000); the kernel
    broadcasts it over heads and query positions like the reference's [B,1,1,L] tensor."""
    if attention_mask.dim() != 2:
        raise NotImplementedError("only 2-D attention masks are supported by the HIP attention kernel")
    return ((1.0 - attention_mask.to(torch.float32)) * -10000.0).contiguous()



# engine.AsyncCounts (pinned buffer + event) of the joint passes still in flight, per BiBertImgModel instance; weak: an entry
# dies with its module and is never copied or pickled with it
import weakref as _weakref
_JOINT_PENDING = _weakref.WeakKeyDictionary()

class CaptionBertLayer(BertLayer):
    pass


class CaptionBertEncoder(nn.Module):
    """vl:123-178.  forward(hidden [B,L,H] bf16, additive mask f32 [B,L]) -> (hidden,)"""

    def __init__(self, config):
        super().__init__()
        # vl:131-132.  Both are inspection outputs: a stack that has to hand them back runs layer by layer on padded tensors
        # (_forward_inspect); the step's fused path never materialises them.
        self.output_attentions = config.output_attentions
        self.output_hidden_states = config.output_hidden_states
        self.num_layers = config.num_hidden_layers
        self.layer = nn.ModuleList([CaptionBertLayer(config) for _ in range(config.num_hidden_layers)])
        self._packs = engine.PackList(engine.LayerPack() for _ in range(config.num_hidden_layers))
        self._dims = (config.hidden_size, config.num_attention_heads, config.intermediate_size, config.layer_norm_eps)
        # skip padded rows (see forward): "train" (default) = in training mode only, where nothing but
        # the losses is consumed; True = always; False = never (the reference's padded execution)
        self.unpad = getattr(config, "unpad", "train")
        # gelu'(u) stash of the FFN: "u8" (default: 8-bit fixed point, dithered rounding: |error| < 0.005, zero mean) or "bf16" (rounds 1-3; reference-numerics
        # runs and A/B runs of the 8-bit stash: tools/soak.py --gelu-stash)
        self.gelu_stash_bf16 = str(getattr(config, "gelu_stash", "u8")).lower() == "bf16"
        # config.fold_layernorm: True = every no-grad eval forward of the stack folds its LayerNorms into the neighbouring GEMMs
        # (engine.encoder_infer_folded; equal to the unfused path up to bf16 rounding), False = never, None (default) = only inside the
        # cached retrieval engine (encode_text / encode_image / fuse_pairs with packed=True, which promise no more than that)
        self.fold_layernorm = getattr(config, "fold_layernorm", None)
        self._fold_now = False

    def _apply(self, fn, *args, **kwargs):
        self.__dict__.pop("_flat_cache", None)      # .to() / .half() may replace Parameter objects
        return super()._apply(fn, *args, **kwargs)

    def _flat_params(self):
        """The stack's parameters, sixteen per layer in LayerPack.NAMES order.  Walking the module tree costs ~130 us of
        host time per call (six calls per training step), so the list is kept until a parameter object is replaced."""
        cached = self.__dict__.get("_flat_cache")
        if cached is not None and cached[0] is self.layer[0].attention.self.query.weight and \
                cached[-1] is self.layer[-1].output.LayerNorm.bias:
            return cached
        out = []
        for layer in self.layer:
            a = layer.attention
            out += [a.self.query.weight, a.self.query.bias, a.self.key.weight, a.self.key.bias,
                    a.self.value.weight, a.self.value.bias, a.output.dense.weight, a.output.dense.bias,
                    a.output.LayerNorm.weight, a.output.LayerNorm.bias, layer.intermediate.dense.weight,
                    layer.intermediate.dense.bias, layer.output.dense.weight, layer.output.dense.bias,
                    layer.output.LayerNorm.weight, layer.output.LayerNorm.bias]
        self.__dict__["_flat_cache"] = out
        return out

    def grad_arena_units(self):
        """Per layer, its sixteen parameters in the order the backward kernels write their gradients
        (mvptr_layer_grads): mvp_pytorch_amd.dp.GradSync lays each list out back to back so that
        engine.EncoderFn.backward accumulates straight into the gradient arena."""
        flat = self._flat_params()
        return [[flat[16 * li + j] for j in engine.EncoderFn.ARENA_ORDER] for li in range(len(self.layer))]

    def forward(self, hidden_states, attention_mask, head_mask=None, encoder_history_states=None,
                return_at_layer=None, pack_hint=None):
        if encoder_history_states is not None:
            raise NotImplementedError("history states are outside the accelerated path")
        if head_mask is not None and any(h is not None for h in head_mask):
            raise NotImplementedError("head_mask must stay None")
        if isinstance(attention_mask, (list, tuple)) or self.output_hidden_states or self.output_attentions:
            return self._forward_inspect(hidden_states, attention_mask, return_at_layer)
        if return_at_layer is not None:
            # vl:162-163,176-177: also hand back the hidden states after layer `return_at_layer` -> ((final,), mid).
            # Two segments of the stack on padded tensors (the mid output is read position by position).
            k = int(return_at_layer)
            n = len(self.layer)
            if not 0 <= k < n:
                return self.forward(hidden_states, attention_mask, head_mask, None, None, pack_hint), None   # the reference leaves mid_output None
            mid = self._run_layers(hidden_states, attention_mask, 0, k + 1)
            out = mid if k + 1 == n else self._run_layers(mid, attention_mask, k + 1, n - k - 1)
            return (out,), mid
        B, L, H = hidden_states.shape
        Hc, heads, I, eps = self._dims
        l0 = self.layer[0]
        x = hidden_states.to(torch.bfloat16).contiguous().view(B * L, H)
        mask = attention_mask.contiguous()
        if self.unpad is True or (self.unpad == "train" and self.training):
            # Row-packed execution: the reference pushes every padded slot through all layers and
            # only masks it as a key (vl:430-460).  A padded row influences nothing that reaches a loss
            # (its output is read by nobody, as a key it weighs exp(-10000) = 0), so only the valid
            # rows are gathered, run through the stack and scattered back; padded rows of the returned
            # tensor are zero instead of the reference's unused values (hence off in eval mode by
            # default: inference outputs stay position-for-position what the reference returns).
            valid = (mask == 0).view(-1)
            lens = (mask == 0).sum(1, dtype=torch.int32)
            if pack_hint is not None:   # (rows, longest) already fetched by the caller with other counts
                rows, lmax = pack_hint
            else:
                rows, lmax = (int(v) for v in torch.stack([lens.sum(), lens.max()]).tolist())   # one host sync
            if rows < B * L and lmax > 0:
                idx = torch.nonzero_static(valid, size=rows).view(-1)
                starts = (torch.cumsum(lens, 0, dtype=torch.int32) - lens).contiguous()
                meta = engine.EncoderMeta(self._packs.for_device(hidden_states.device), B, lmax, Hc, heads, I, eps, self.training, l0.output.dropout.p,
                                          l0.attention.self.dropout.p, seq_start=starts, seq_len=lens.contiguous(), rows=rows)
                meta.stash_bf16 = self.gelu_stash_bf16
                meta.beside = bool(self.__dict__.get("_beside", False))      # set by BiBertImgModel while its two uni-modal stacks share the GPU
                if self._folds(meta):
                    y = engine.encoder_infer_folded(engine.PackRows.apply(x, idx), None, meta, self._flat_params())
                else:
                    y = engine.EncoderFn.apply(engine.PackRows.apply(x, idx), None, meta, *self._flat_params())
                return (engine.UnpackRows.apply(y, idx, B * L).view(B, L, H),)
        meta = engine.EncoderMeta(self._packs.for_device(hidden_states.device), B, L, Hc, heads, I, eps, self.training,
                                  l0.output.dropout.p, l0.attention.self.dropout.p)
        meta.stash_bf16 = self.gelu_stash_bf16
        meta.beside = bool(self.__dict__.get("_beside", False))      # set by BiBertImgModel while its two uni-modal stacks share the GPU
        if self._folds(meta):
            return (engine.encoder_infer_folded(x, mask, meta, self._flat_params()).view(B, L, H),)
        y = engine.EncoderFn.apply(x, mask, meta, *self._flat_params())
        return (y.view(B, L, H),)

    def _folds(self, meta):
        """LayerNorms folded into the GEMMs for this call?  Only without autograd, in eval mode, for shapes mvptr_gemm_nt_ln takes."""
        want = self.fold_layernorm is True or (self.fold_layernorm is None and self._fold_now)
        return (want and not self.training and not torch.is_grad_enabled() and engine.fold_eligible(meta, self._flat_params()))


def _run_layers(self, hidden_states, attention_mask, first, count):
    """Layers [first, first + count) of the stack on a padded [B, L, H] tensor (padded execution, as the reference)."""
    B, L, H = hidden_states.shape
    Hc, heads, I, eps = self._dims
    l0 = self.layer[0]
    x = hidden_states.to(torch.bfloat16).contiguous().view(B * L, H)
    meta = engine.EncoderMeta(self._packs.for_device(hidden_states.device), B, L, Hc, heads, I, eps, self.training,
                              l0.output.dropout.p, l0.attention.self.dropout.p, first=first, count=count,
                              all_params=self._flat_params())
    meta.stash_bf16 = self.gelu_stash_bf16
    meta.beside = bool(self.__dict__.get("_beside", False))      # set by BiBertImgModel while its two uni-modal stacks share the GPU
    y = engine.EncoderFn.apply(x, attention_mask.contiguous(), meta, *self._flat_params()[16 * first:16 * (first + count)])
    return y.view(B, L, H)


CaptionBertEncoder._run_layers = _run_layers


def _forward_inspect(self, hidden_states, attention_mask, return_at_layer=None):
    """vl:134-178 with the outputs the fused call does not produce: `all_hidden_states` (config.output_hidden_states: the input
    of every layer + the last output), `all_attentions` (config.output_attentions: softmax(QK^T/8 + mask) per layer, f32
    [B, heads, L, L], detached), and a LIST of masks (one per phase of ceil(n / len) layers; the first phase's output is handed
    back as `stage_output`).  Same tuple order as the reference: (hidden, [all_hidden], [all_attentions], [stage_output]), and
    (outputs, mid_output) with return_at_layer.  Padded execution, runs of layers that need nothing collected stay one launch
    sequence (_run_layers)."""
    n = len(self.layer)
    phased = isinstance(attention_mask, (list, tuple))
    masks = list(attention_mask) if phased else [attention_mask]
    per = -(-n // len(masks))
    heads = self._dims[1]
    if self.output_attentions and self.training and self.layer[0].attention.self.dropout.p > 0:
        raise NotImplementedError("output_attentions with active attention dropout: the kernels keep no L x L mask to hand back "
                                  "(use eval mode or attention_probs_dropout_prob = 0)")
    every = self.output_hidden_states or self.output_attentions
    cuts = set(range(1, n)) if every else set()
    if phased:
        cuts.update(range(per, n, per))
    if return_at_layer is not None and 0 <= int(return_at_layer) < n - 1:
        cuts.add(int(return_at_layer) + 1)
    bounds = [0] + sorted(cuts) + [n]
    all_h, all_a = (), ()
    stage = mid = None
    h = hidden_states.to(torch.bfloat16)
    for first, last in zip(bounds[:-1], bounds[1:]):
        m = masks[first // per].contiguous()
        if self.output_hidden_states:
            all_h += (h,)
        if self.output_attentions:
            all_a += (self._attention_probs(h, m, first, heads),)
        h = self._run_layers(h, m, first, last - first)
        if phased and last == per:
            stage = h
        if return_at_layer is not None and last - 1 == int(return_at_layer):
            mid = h
    outputs = (h,)
    if self.output_hidden_states:
        outputs += (all_h + (h,),)
    if self.output_attentions:
        outputs += (all_a,)
    if stage is not None:
        outputs += (stage,)
    return (outputs, mid) if return_at_layer is not None else outputs


def _attention_probs(self, hidden_states, mask_add, li, heads):
    """Layer li's attention probabilities for its input `hidden_states`: Q | K projection of the same bf16 rows (mvptr_gemm_nt,
    bias epilogue) + mvptr_attention_probs."""
    B, L, H = hidden_states.shape
    a = self.layer[li].attention.self
    with torch.no_grad():
        w = torch.cat([a.query.weight, a.key.weight, a.value.weight]).to(torch.bfloat16)
        b = torch.cat([a.query.bias, a.key.bias, a.value.bias]).float()
        qkv = hip.gemm_nt(hidden_states.detach().contiguous().view(B * L, H), w, hip.EPI_BIAS, bias=b)
        return hip.attention_probs(qkv, mask_add, B, L, heads)


CaptionBertEncoder._forward_inspect = _forward_inspect
CaptionBertEncoder._attention_probs = _attention_probs


def _forward_rows(self, x_rows, seq_start, seq_len, n_seq, lmax, rows_dev=None, rows_plan=0):
    """The layer stack on row-packed input: x_rows bf16 [rows, H] holds the valid token rows of n_seq sequences
    back to back (sequence b at [seq_start[b], +seq_len[b]), device int32; lmax = the longest) -> [rows, H].
    The caller owns the packing (hip.pack_maps + engine.MultiTapFn).  rows_dev: device-side count of the rows that are
    really present (x_rows then has the BOUND's rows, the tail is never read; lmax an upper bound): no host read-back."""
    Hc, heads, I, eps = self._dims
    l0 = self.layer[0]
    meta = engine.EncoderMeta(self._packs.for_device(x_rows.device), n_seq, lmax, Hc, heads, I, eps, self.training,
                              l0.output.dropout.p, l0.attention.self.dropout.p, seq_start=seq_start, seq_len=seq_len,
                              rows=x_rows.shape[0], rows_dev=rows_dev, rows_plan=rows_plan)
    meta.stash_bf16 = self.gelu_stash_bf16
    meta.beside = bool(self.__dict__.get("_beside", False))      # set by BiBertImgModel while its two uni-modal stacks share the GPU
    return engine.EncoderFn.apply(x_rows, None, meta, *self._flat_params())


CaptionBertEncoder.forward_rows = _forward_rows


def _prefetch_weights(self, device):
    """bf16 working copies of this stack rebuilt now, on the current stream (see engine.EncoderPacks.prefetch)."""
    self._packs.for_device(device).group.prefetch(self._flat_params())


CaptionBertEncoder.prefetch_weights = _prefetch_weights


def _check_img_type(config):
    if getattr(config, "img_feature_type", "faster_r-cnn") in ("dis_code", "dis_code_t", "dis_code_scale"):
        raise NotImplementedError("discrete-code image features are outside the accelerated path")


class _ImgBackboneMixin:
    def _init_img(self, config):
        _check_img_type(config)
        self.img_dim = config.img_feature_dim
        self.img_feature_type = config.img_feature_type
        self.use_img_layernorm = getattr(config, "use_img_layernorm", None)
        self.img_embedding = nn.Linear(self.img_dim, config.hidden_size, bias=True)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)
        if self.use_img_layernorm:
            self.LayerNorm = BertLayerNorm(config.hidden_size, eps=config.img_layer_norm_eps)

    def _resize_token_embeddings(self, new_num_tokens):
        self.embeddings.word_embeddings = self._get_resized_embeddings(self.embeddings.word_embeddings, new_num_tokens)
        return self.embeddings.word_embeddings

    @property
    def dtype(self):
        return next(self.parameters()).dtype


class BertImgModel(_ImgBackboneMixin, BertPreTrainedModel):
    """vl:202-352 — single-stream backbone over [text ; regions]."""

    def __init__(self, config):
        super().__init__(config)
        self.embeddings = BertEmbeddings(config)
        self.encoder = CaptionBertEncoder(config)
        self.pooler = BertPooler(config)
        self._init_img(config)
        self.apply(self.init_weights)

    def forward(self, input_ids, token_type_ids=None, attention_mask=None, position_ids=None,
                head_mask=None, img_feats=None, encoder_history_states=None, pack_hint=None):
        """pack_hint (optional, not a reference argument): (valid rows, longest sequence) of the batch from where it was built —
        the row-packed stack then reads nothing back (checked on the device, mvptr_check_counts)."""
        if head_mask is not None or encoder_history_states:
            raise NotImplementedError("head_mask / encoder_history_states are outside the accelerated path")
        if attention_mask is None:
            n = input_ids.shape[1] + (img_feats.shape[1] if img_feats is not None else 0)
            attention_mask = torch.ones((input_ids.shape[0], n), dtype=torch.long, device=input_ids.device)
        if token_type_ids is None:
            token_type_ids = torch.zeros_like(input_ids)
        phased = isinstance(attention_mask, (list, tuple))      # vl:265-276: one mask per phase of the stack
        mask = [additive_mask(m) for m in attention_mask] if phased else additive_mask(attention_mask)
        x = embed_inputs(self.embeddings, input_ids, token_type_ids, position_ids, img_feats, self)
        if pack_hint is not None and not phased and x.is_cuda and (self.encoder.unpad is True or (self.encoder.unpad == "train" and self.training)):
            ln = attention_mask.sum(1)
            cnt = torch.stack([ln.sum(), ln.max()]).to(torch.int64)
            hip.check_counts(cnt, cnt, tuple(int(v) for v in pack_hint) * 2)
        else:
            pack_hint = None
        encoder_outputs = self.encoder(x, mask, pack_hint=pack_hint)
        sequence_output = encoder_outputs[0]
        # vl:346-347: hidden states / attentions / stage output follow when the configuration asks for them
        return (sequence_output, self.pooler(sequence_output)) + tuple(encoder_outputs[1:])


class BiBertImgModel(_ImgBackboneMixin, BertPreTrainedModel):
    """vl:354-874 — MVPTR two-stage backbone: txt_encoder / vis_encoder (uni-modal) and
    mul_encoder (joint), CLIP-style global similarity and in-batch hard negatives."""

    def __init__(self, config):
        super().__init__(config)
        self.embeddings = BertEmbeddings(config)
        half = copy.deepcopy(config)
        half.num_hidden_layers = half.num_hidden_layers // 2
        self.vis_encoder = CaptionBertEncoder(half)
        self.txt_encoder = CaptionBertEncoder(half)
        self.mul_encoder = CaptionBertEncoder(half)
        self.pooler = BertPooler(config)
        scale = config.hidden_size ** -0.5
        self.txt_proj = nn.Parameter(scale * torch.randn(config.hidden_size, config.hidden_size))
        self.vis_proj = nn.Parameter(scale * torch.randn(config.hidden_size, config.hidden_size))
        self._init_img(config)
        # see _uni: True = whenever this process is not part of a multi-rank job, "always" = also under
        # torch.distributed, False = never
        self.parallel_stacks = getattr(config, "parallel_stacks", True)
        # rebuild the three stacks' bf16 weight copies on the second stream beside the embedding kernels (first step / foreign
        # optimizers; the fused AdamW keeps them current afterwards); config.prefetch_weights = False: in front of each stack
        self.prefetch_weights = bool(getattr(config, "prefetch_weights", True))
        # the joint + hard-negative pass of the packed pipeline without reading its row count back (mvptr_layer_desc.rows_dev);
        # config.sync_free_joint = False: wait for the count and size the pass exactly
        self.sync_free_joint = bool(getattr(config, "sync_free_joint", True))
        # (the engine.AsyncCounts of recent steps' joint row counts live in _JOINT_PENDING, a weak side table: CUDA events
        #  and pinned buffers on the module itself would break copy.deepcopy(model) / torch.save(model))
        self._joint_plan = 0               # the latest count that has landed: the planning hint (0: none yet -> the bound)
        self.apply(self.init_weights)

    # -- stage 1: uni-modal encoders (vl:479-513)
    def _uni(self, input_ids_a, token_type_ids_a, attention_mask_a, position_ids_a, input_ids_b,
             token_type_ids_b, attention_mask_b, position_ids_b, img_feats, pack_hints=None):
        if attention_mask_a is None:
            attention_mask_a = torch.ones_like(input_ids_a)
        if attention_mask_b is None:
            attention_mask_b = torch.ones_like(input_ids_b)
        if token_type_ids_a is None:
            token_type_ids_a = torch.zeros_like(input_ids_a)
        if token_type_ids_b is None:
            token_type_ids_b = torch.zeros_like(input_ids_b)
        mask_a = additive_mask(attention_mask_a)
        mask_b = additive_mask(attention_mask_b)
        two_streams = bool(self.parallel_stacks) and input_ids_a.is_cuda and _streams_allowed(self.parallel_stacks)
        self.txt_encoder._beside = self.vis_encoder._beside = two_streams      # tile-height hint of their GEMMs (mvptr_layer_desc.beside)
        prefetch = two_streams and self.training and torch.is_grad_enabled() and self.prefetch_weights
        if prefetch:
            # training rebuilds the bf16 weight copies of every stack each forward pass (one ~0.1-ms launch per stack):
            # all three go to the side stream now, beside the embedding kernels, instead of in front of each stack
            main = torch.cuda.current_stream(input_ids_a.device)
            side = engine.side_stream(input_ids_a.device)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                for enc in (self.txt_encoder, self.vis_encoder, self.mul_encoder):
                    enc.prefetch_weights(input_ids_a.device)
        share = {}   # one word-table gradient buffer for both lookups of this forward pass
        xa = embed_inputs(self.embeddings, input_ids_a, token_type_ids_a, position_ids_a, None, self, share)
        xb = embed_inputs(self.embeddings, input_ids_b, token_type_ids_b, position_ids_b, img_feats, self, share)
        if prefetch:
            main.wait_stream(side)      # the copies are read by the stacks below
        hint_a = hint_b = None
        if pack_hints is not None:
            # fetched by the caller together with its other counts; an AsyncCounts is awaited only now,
            # with the embedding kernels already queued behind the copy
            if isinstance(pack_hints, engine.AsyncCounts):
                c = pack_hints.get()
                pack_hints = ((c[2], c[3]), (c[4], c[5]))
            hint_a, hint_b = pack_hints
        elif self.txt_encoder.unpad is True or (self.txt_encoder.unpad == "train" and self.training):
            # the valid-row counts of both uni-modal passes in ONE device->host copy, issued before any
            # encoder work is queued (a sync in the middle of the forward pass drains the launch queue)
            la, lb = attention_mask_a.sum(1), attention_mask_b.sum(1)
            c = torch.stack([la.sum(), la.max(), lb.sum(), lb.max()]).tolist()
            hint_a, hint_b = (int(c[0]), int(c[1])), (int(c[2]), int(c[3]))
        if two_streams:
            # The two uni-modal stacks are independent networks: the visual one runs on a second HIP
            # stream beside the text one.  At ~11 k rows per stack a 256x256-tile GEMM with N = 768
            # occupies half of the CUs, so the two stacks' kernels fill each other's idle CUs
            # (tools/overlap_stacks.py: 3.35 -> 2.60 ms for 6+6 forward layers); autograd replays
            # each stack's backward on the stream its forward ran on.
            main = torch.cuda.current_stream(xa.device)
            side = engine.side_stream(xa.device)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                vis = self.vis_encoder(xb, mask_b, pack_hint=hint_b)[0]
            txt = self.txt_encoder(xa, mask_a, pack_hint=hint_a)[0]
            main.wait_stream(side)
            for t in (xb, mask_b):
                t.record_stream(side)      # allocated on the main stream, read on the side stream
            vis.record_stream(main)        # and the other way round
        else:
            txt = self.txt_encoder(xa, mask_a, pack_hint=hint_a)[0]
            vis = self.vis_encoder(xb, mask_b, pack_hint=hint_b)[0]
        return txt, vis, mask_a, mask_b

    @staticmethod
    def mine_hard_negatives(sim_mat, hn_mod="hard", logit=None):
        """vl:531-542 — per text the most similar OTHER image and per image the most similar other
        text ('hard': argmax of sim - 2 I; 'sample': one multinomial draw from softmax(logit * sim)
        with the diagonal at -10000) -> (hard_img_index [n], hard_txt_index [n]) int64."""
        n = sim_mat.shape[0]
        if hn_mod == "hard" and sim_mat.is_cuda and sim_mat.dtype == torch.float32:
            from .. import hip
            return hip.hard_negative_mine(sim_mat.detach().contiguous())          # mvptr_hard_negative_mine: two argmax passes, one launch
        eye = torch.eye(n, dtype=sim_mat.dtype, device=sim_mat.device)
        if hn_mod == "hard":
            masked = sim_mat - 2 * eye
            return torch.max(masked, dim=1)[1], torch.max(masked, dim=0)[1]
        if hn_mod == "sample":
            masked = (logit * sim_mat) - 10000 * eye
            return (torch.multinomial(F.softmax(masked, dim=1), num_samples=1).squeeze(),
                    torch.multinomial(F.softmax(masked.t(), dim=1), num_samples=1).squeeze())
        raise NotImplementedError

    @staticmethod
    def _cls_rows(seq):
        """[CLS] rows of a [B, L, H] sequence tensor, tapped by row index when it is a contiguous bf16 device buffer."""
        if seq.is_cuda and seq.dtype == torch.bfloat16 and seq.is_contiguous():
            B, L, H = seq.shape
            return engine.tap_rows(seq.view(B * L, H), engine.arange(B, seq.device, torch.int32, step=L))
        return seq[:, 0, :]

    def _project(self, cls_rows, proj):
        """normalize(cls @ proj) in f32 (vl:525-526; feeds the hard-negative argmax: kept out of bf16)."""
        if cls_rows.is_cuda and proj.dtype == torch.float32:
            return engine.L2NormFn.apply(engine.SmallLinearFn.apply(cls_rows, proj, None, None, True))
        # documented torch paths only: fp16 parameters of model.half() inference, CPU tensors
        return F.normalize(cls_rows.float() @ proj.float(), p=2, dim=-1)

    @staticmethod
    def _sim(global_txt, global_img):
        """vl:527 — exact f32 similarity matrix."""
        if global_txt.is_cuda and global_txt.dtype == torch.float32 and global_img.dtype == torch.float32:
            return engine.SimFn.apply(global_txt, global_img)
        return global_txt @ global_img.t()

    def _globals(self, txt, vis):
        return self._project(self._cls_rows(txt), self.txt_proj), self._project(self._cls_rows(vis), self.vis_proj)

    def forward(self, input_ids_a, token_type_ids_a=None, attention_mask_a=None, max_tag_length=None,
                use_b=False, position_ids_a=None, input_ids_b=None, token_type_ids_b=None,
                attention_mask_b=None, phrase_layer=None, position_ids_b=None, head_mask=None,
                img_feats=None, encoder_history_states=None, encode_hn=False, hn_mod="hard", logit=None,
                pack_hints=None, beside=None, host_counts=None):
        """host_counts (optional, encode_hn=False paths; synthetic.finetune_host_counts): rows / longest sequence of the three
        row-packed passes computed where the batch was built — no read-back inside the step (checked on the device against the
        masks, mvptr_check_counts: a mismatch traps, every buffer behind it is sized from these numbers).
        beside: optional callable(txt, vis, sim_mat) with work that only needs the uni-modal outputs (the visual
        MLM head and the contrastive loss of the pre-training model).  It is queued on the side stream before the
        joint stack, so its small kernels — and, in the backward pass, their gradients — run beside the joint
        stack's GEMMs instead of after them; the caller waits for engine.side_stream before using its results."""
        if head_mask is not None or encoder_history_states:
            raise NotImplementedError("head_mask / encoder_history_states are outside the accelerated path")
        joint_hint = None
        packing = input_ids_a.is_cuda and (self.txt_encoder.unpad is True or (self.txt_encoder.unpad == "train" and self.training))
        if host_counts is not None and not encode_hn and pack_hints is None and packing and attention_mask_a is not None and attention_mask_b is not None:
            hc = {k: int(v) for k, v in host_counts.items()}
            pack_hints = ((hc["rows_a"], hc["lmax_a"]), (hc["rows_b"], hc["lmax_b"]))
            joint_hint = (hc["rows_j"], hc["lmax_j"])
            cut0 = 1 if use_b else max_tag_length
            la, lb = attention_mask_a.sum(1), attention_mask_b.sum(1)
            lj = la + attention_mask_b[:, cut0:].sum(1)
            cnts = torch.stack([la.sum(), la.max(), lb.sum(), lb.max(), lj.sum(), lj.max()]).to(torch.int64)
            hip.check_counts(cnts[0:2], cnts[2:4], pack_hints[0] + pack_hints[1])
            hip.check_counts(cnts[4:6], cnts[4:6], joint_hint + joint_hint)
        txt, vis, mask_a, mask_b = self._uni(input_ids_a, token_type_ids_a, attention_mask_a, position_ids_a,
                                             input_ids_b, token_type_ids_b, attention_mask_b, position_ids_b, img_feats,
                                             pack_hints)
        cut = 1 if use_b else max_tag_length
        only_vis = vis[:, cut:, :]
        only_vis_mask = mask_b[:, cut:]
        global_txt, global_img = self._globals(txt, vis)
        sim_mat = self._sim(global_txt, global_img)

        hard_out = hard_pooled = hard_txt_full = hard_img_full = None
        if encode_hn:
            n = sim_mat.shape[0]
            dev = sim_mat.device
            hard_img, hard_txt = self.mine_hard_negatives(sim_mat, hn_mod, logit)
            dice = torch.randperm(n, device=dev)
            first, second = dice[: n // 2], dice[n // 2:]
            ar = engine.arange(n, dev)
            # rows of the hard batch: (text i, image hard_img[i]) for i in first,
            #                         (text hard_txt[j], image j)  for j in second       (vl:544-566)
            hard_txt_full = torch.cat([ar.index_select(0, first), hard_txt.index_select(0, second)], 0)
            hard_img_full = torch.cat([hard_img.index_select(0, first), ar.index_select(0, second)], 0)
            hard_mask = torch.cat([mask_a.index_select(0, hard_txt_full), only_vis_mask.index_select(0, hard_img_full)], 1)
        joint_mask = torch.cat([mask_a, only_vis_mask], dim=-1)
        cnt = None
        enc = self.mul_encoder
        if encode_hn and txt.is_cuda and (enc.unpad is True or (enc.unpad == "train" and self.training)):
            # rows / longest sequence of the 2n-row pass depend on the mined hard negatives: request them
            # from the (small) masks now, queue the gathers and concatenations of the activations behind
            # the copy, and read the two numbers while the GPU is still busy with those
            both_mask = torch.cat([joint_mask, hard_mask], 0)
            lens = (both_mask == 0).sum(1)
            cnt = engine.AsyncCounts([lens.sum(), lens.max()])
        if beside is not None:
            if txt.is_cuda and self.parallel_stacks and _streams_allowed(self.parallel_stacks):
                main = torch.cuda.current_stream(txt.device)
                side = engine.side_stream(txt.device)
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    beside(txt, vis, sim_mat)
                for t in (txt, vis, sim_mat):
                    t.record_stream(side)      # produced on the main stream, read on the side stream
            else:
                beside(txt, vis, sim_mat)
        if encode_hn:
            hard_seqs = torch.cat([txt.index_select(0, hard_txt_full), only_vis.index_select(0, hard_img_full)], 1)
        joint = torch.cat([txt, only_vis], dim=1)
        if encode_hn:
            # The reference runs mul_encoder twice (hard batch vl:567, matched batch vl:577); the two
            # passes share weights and do not depend on each other, so they go through the layer
            # stack as ONE 2n-row batch: at configs[1] that is 250 row tiles of 256 instead of
            # 2 x 125, which fills whole rounds of the 256 CUs (375 tiles = 1.46 rounds before).
            n = joint.shape[0]
            both_in = torch.cat([joint, hard_seqs], 0)
            if cnt is None:
                both_mask, hint = torch.cat([joint_mask, hard_mask], 0), None
            else:
                hint = tuple(cnt.get())
            mid_joint = mid_hard = None
            if phrase_layer is not None:      # vl:570-572,589-592: also the hidden states after layer `phrase_layer`
                (both,), mid = self.mul_encoder(both_in, both_mask, return_at_layer=phrase_layer)
                if mid is not None:
                    mid_joint, mid_hard = mid[:n], mid[n:]
            else:
                both = self.mul_encoder(both_in, both_mask, pack_hint=hint)[0]
            sequence_output, hard_out = both[:n], both[n:]
            hard_pooled = self.pooler(hard_out)
        else:
            mid_joint = mid_hard = None
            if phrase_layer is not None:
                (sequence_output,), mid_joint = self.mul_encoder(joint, joint_mask, return_at_layer=phrase_layer)
            else:
                sequence_output = self.mul_encoder(joint, joint_mask, pack_hint=joint_hint)[0]
        pooled_output = self.pooler(sequence_output)
        outputs = (sequence_output, pooled_output, hard_out, hard_pooled)
        if phrase_layer is not None:          # vl:605-608: a fourth element
            return outputs, (txt, vis, sim_mat), (hard_txt_full, hard_img_full), (mid_joint, mid_hard)
        return outputs, (txt, vis, sim_mat), (hard_txt_full, hard_img_full)

    # -- row-packed pipeline (training fast path) ------------------------------------------------------
    def packed_ok(self, attention_mask_a, attention_mask_b, input_ids_a):
        enc = self.txt_encoder
        return (input_ids_a.is_cuda and attention_mask_a is not None and attention_mask_b is not None and
                attention_mask_a.dim() == 2 and attention_mask_b.dim() == 2 and
                (enc.unpad is True or (enc.unpad == "train" and self.training)) and
                self.embeddings.word_embeddings.weight.dtype == torch.float32)

    def forward_packed(self, input_ids_a, token_type_ids_a, attention_mask_a, max_tag_length, input_ids_b, token_type_ids_b,
                       attention_mask_b, img_feats, position_ids_a=None, position_ids_b=None, use_b=False, hn_mod="hard",
                       logit=None, uni_taps=None, beside=None, host_counts=None):
        """The two-stage backbone of `forward(encode_hn=True)` without ever materialising a padded activation tensor
        between the stacks (vl:479-600): the valid rows of the text / visual inputs are gathered once, each stack
        runs on packed rows, the packed joint + hard-negative input is gathered straight from the two packed outputs
        through index maps built on the device (hip.pack_maps), and everything later stages read — [CLS] states,
        masked rows, phrase / region rows — is tapped from packed buffers by row index (engine.MultiTapFn).
        Same arithmetic on the same rows as the row-packed `forward`.

        uni_taps(pos_a, pos_b) -> (list of int32 row vectors into the packed text output, same for the visual
        output): extra rows the caller wants from the uni-modal outputs; beside(taps_txt, taps_vis, sim_mat): work
        that only needs those (queued on the side stream beside the joint stack).
        -> dict: both (packed joint output [rows_j, H]; sequences 0..n-1 matched, n..2n-1 hard), pos_j int32 [2n, Lj]
        (packed row of every slot of the unpadded joint layout, -1 = padded), seq_start_j, sim_mat, hard_txt_full,
        hard_img_full, pos_a, pos_b, n_txt_rows."""
        from .. import hip
        dev = input_ids_a.device
        B, La = input_ids_a.shape
        Lb = attention_mask_b.shape[1]
        H = self.config.hidden_size
        if token_type_ids_a is None:
            token_type_ids_a = torch.zeros_like(input_ids_a)
        if token_type_ids_b is None:
            token_type_ids_b = torch.zeros_like(input_ids_b)
        mask_a, mask_b = additive_mask(attention_mask_a), additive_mask(attention_mask_b)
        pos_a, idx_a, st_a, ln_a, cnt_a = hip.pack_maps([dict(mask=mask_a, len=La, src_seq_stride=La)], B)
        pos_b, idx_b, st_b, ln_b, cnt_b = hip.pack_maps([dict(mask=mask_b, len=Lb, src_seq_stride=Lb)], B)
        # rows / longest sequence of the two uni-modal passes: from the caller (host_counts, computed where the batch was
        # built) or read back from the device behind the embedding kernels
        counts = None if host_counts is not None else engine.AsyncCounts([cnt_a[0], cnt_a[1], cnt_b[0], cnt_b[1]])
        two_streams = bool(self.parallel_stacks) and _streams_allowed(self.parallel_stacks)
        self.txt_encoder._beside = self.vis_encoder._beside = two_streams
        main = torch.cuda.current_stream(dev)
        side = engine.side_stream(dev) if two_streams else None
        prefetch = two_streams and torch.is_grad_enabled() and self.prefetch_weights
        if prefetch:
            side.wait_stream(main)
            with torch.cuda.stream(side):
                for enc in (self.txt_encoder, self.vis_encoder, self.mul_encoder):
                    enc.prefetch_weights(dev)
        share = {}
        xa = embed_inputs(self.embeddings, input_ids_a, token_type_ids_a, position_ids_a, None, self, share)
        xb = embed_inputs(self.embeddings, input_ids_b, token_type_ids_b, position_ids_b, img_feats, self, share)
        if prefetch:
            main.wait_stream(side)
        if host_counts is not None:
            ra, la_max, rb, lb_max = (int(host_counts[k]) for k in ("rows_a", "lmax_a", "rows_b", "lmax_b"))
            if getattr(self, "verify_host_counts", False):
                # bring-up mode of a data pipeline: one read-back, a catchable error instead of the device-side trap
                got = [int(v) for v in torch.cat([cnt_a, cnt_b]).tolist()]
                if got != [ra, la_max, rb, lb_max]:
                    raise ValueError("host_counts (rows_a, lmax_a, rows_b, lmax_b) = %r, the masks give %r" % ([ra, la_max, rb, lb_max], got))
            else:
                hip.check_counts(cnt_a, cnt_b, (ra, la_max, rb, lb_max))      # device-side: host_counts must describe this batch's masks
        else:
            ra, la_max, rb, lb_max = counts.get()
        xa_p = engine.tap_rows(xa.view(B * La, H), idx_a[:ra])
        if two_streams:
            side.wait_stream(main)
            with torch.cuda.stream(side):
                xb_p = engine.tap_rows(xb.view(B * Lb, H), idx_b[:rb])
                vis_p = self.vis_encoder.forward_rows(xb_p, st_b, ln_b, B, lb_max)
            txt_p = self.txt_encoder.forward_rows(xa_p, st_a, ln_a, B, la_max)
            main.wait_stream(side)
            for t in (xb, idx_b, st_b, ln_b):
                t.record_stream(side)
            vis_p.record_stream(main)
        else:
            xb_p = engine.tap_rows(xb.view(B * Lb, H), idx_b[:rb])
            txt_p = self.txt_encoder.forward_rows(xa_p, st_a, ln_a, B, la_max)
            vis_p = self.vis_encoder.forward_rows(xb_p, st_b, ln_b, B, lb_max)
        # [CLS] states + whatever else the caller reads from the uni-modal outputs: one tap call
        extra_t, extra_v = uni_taps(pos_a, pos_b) if uni_taps is not None else ([], [])
        cls_t, cls_v = pos_a[:, 0].contiguous(), pos_b[:, 0].contiguous()
        # rows of the second source are addressed at offset ra; a -1 entry (padded slot) must stay a zero row
        off = lambda v: torch.where(v >= 0, v + ra, v)       # noqa: E731
        taps = engine.MultiTapFn.apply(txt_p, vis_p, cls_t, off(cls_v), *extra_t, *[off(v) for v in extra_v])
        global_txt, global_img = self._project(taps[0], self.txt_proj), self._project(taps[1], self.vis_proj)
        sim_mat = self._sim(global_txt, global_img)
        n = B
        own_mining = "mine_hard_negatives" not in self.__dict__        # tests inject captured indices on the instance
        if own_mining and hn_mod == "hard" and sim_mat.dtype == torch.float32:
            # argmax of sim - 2 I along rows and columns, the permutation split and the `sel` vectors of the joint maps in
            # two launches (mvptr_hard_negative_mine); the permutation itself stays torch's draw
            dice = torch.randperm(n, device=dev)
            _, _, hard_txt_full, hard_img_full, sel_txt, sel_img = hip.hard_negative_mine(sim_mat.detach().contiguous(), dice, want_sel=True)
        else:
            hard_img, hard_txt = self.mine_hard_negatives(sim_mat.detach(), hn_mod, logit)
            dice = torch.randperm(n, device=dev)
            first, second = dice[: n // 2], dice[n // 2:]
            ar = engine.arange(n, dev)
            hard_txt_full = torch.cat([ar.index_select(0, first), hard_txt.index_select(0, second)], 0)
            hard_img_full = torch.cat([hard_img.index_select(0, first), ar.index_select(0, second)], 0)
            sel_txt, sel_img = torch.cat([ar, hard_txt_full]), torch.cat([ar, hard_img_full])
        cut = 1 if use_b else max_tag_length
        # the sync-free pass is sized for the padded bound La + Lb - cut per sequence; the attention kernels hold at most 256
        # rows per sequence, so batches whose PADDED widths exceed that (while every real joint sequence fits) take the
        # pass that waits for its exact size instead of failing
        sync_free = self.sync_free_joint and (La + Lb - cut) <= 256
        pos_j, idx_j, st_j, ln_j, cnt_j = hip.pack_maps(
            [dict(mask=mask_a, sel=sel_txt, len=La, pos=pos_a),
             dict(mask=mask_b, sel=sel_img, col0=cut, len=Lb - cut, pos=pos_b, src_base=ra)], 2 * n, fill_idx=sync_free)
        # (inside a HIP-graph capture — train.GraphedStep — nothing can be read back: the pass keeps the plan of the eager warm-up steps)
        capturing = torch.cuda.is_current_stream_capturing()
        cj = None if capturing else engine.AsyncCounts([cnt_j[0], cnt_j[1]])
        if beside is not None:
            if two_streams:
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    beside(taps[2:2 + len(extra_t)], taps[2 + len(extra_t):], sim_mat)
                for t in taps[2:] + (sim_mat,):
                    t.record_stream(side)
            else:
                beside(taps[2:2 + len(extra_t)], taps[2 + len(extra_t):], sim_mat)
        if sync_free:
            # The row count of this pass depends on the mined hard negatives.  Instead of waiting for it the pass is
            # sized for its bound (every slot valid): the gather yields zero rows past the count (idx = -1), every layer
            # kernel clamps to the device-side count (workgroups past it return at once) and the launches are planned
            # for the previous step's count, which has long landed.
            # the latest count that has LANDED is the hint (the host may run more than a step ahead of the device: a copy that
            # is still in flight is simply left for a later step — nothing here waits)
            if not capturing:
                pend = _JOINT_PENDING.setdefault(self, [])
                while pend and pend[0].ready():
                    self._joint_plan = pend.pop(0).get()[0]
                pend.append(cj)
                del pend[:-8]
            xj_p = engine.MultiTapFn.apply(txt_p, vis_p, idx_j)[0]
            both = self.mul_encoder.forward_rows(xj_p, st_j, ln_j, 2 * n, La + Lb - cut, rows_dev=cnt_j, rows_plan=self._joint_plan)
        else:
            if capturing:
                raise RuntimeError("the joint pass sized from a count read-back (config.sync_free_joint = False, or padded widths over 256) "
                                   "cannot be captured in a HIP graph")
            rj, lj_max = cj.get()
            xj_p = engine.MultiTapFn.apply(txt_p, vis_p, idx_j[:rj])[0]
            both = self.mul_encoder.forward_rows(xj_p, st_j, ln_j, 2 * n, lj_max)
        return dict(both=both, pos_j=pos_j, seq_start_j=st_j, sim_mat=sim_mat, hard_txt_full=hard_txt_full,
                    hard_img_full=hard_img_full, pos_a=pos_a, pos_b=pos_b, n_txt_rows=ra, text_len=La)

    def forward_single(self, input_ids_a, token_type_ids_a=None, attention_mask_a=None, max_tag_length=None,
                       position_ids_a=None, input_ids_b=None, token_type_ids_b=None, attention_mask_b=None,
                       position_ids_b=None, head_mask=None, img_feats=None, encoder_history_states=None):
        """vl:611-723 — uni-modal encoders + projections only (retrieval coarse stage)."""
        txt, vis, _, _ = self._uni(input_ids_a, token_type_ids_a, attention_mask_a, position_ids_a,
                                   input_ids_b, token_type_ids_b, attention_mask_b, position_ids_b, img_feats)
        return self._globals(txt, vis)


    # -- two-stage retrieval with cached uni-modal outputs ------------------------
    # The reference's evaluation (run_retrieval.py:694-826) re-runs txt_encoder and vis_encoder for
    # every (caption, image) pair of the re-ranking stage; their outputs only depend on the caption
    # or on the image, so they are computed once here and the pair stage runs mul_encoder alone
    # (9.1 of the 20.9 GFLOP per pair at the README's retrieval shapes).  Same kernels on the same
    # rows: the scores equal forward(..., encode_hn=False) on the materialised pairs.
    class _packed_stacks:
        """context: run the three stacks row-packed whatever the mode (the retrieval engine only reads
        valid rows — [CLS] / pooled outputs).  Bit-identical to the padded execution where the padding
        sits at the end (text stack); in the visual and joint sequences the padding is interior, the
        valid keys move to other 32-key blocks and sums are taken in another order: equal up to bf16
        rounding.  packed=False in encode_text / encode_image / rerank keeps the padded execution."""

        def __init__(self, bert, on=True):
            self.encs = (bert.txt_encoder, bert.vis_encoder, bert.mul_encoder)
            self.on = on

        def __enter__(self):
            self.saved = [e.unpad for e in self.encs]
            for e in self.encs:
                if self.on and e.unpad is not False:
                    e.unpad = True
                e._fold_now = bool(self.on)

        def __exit__(self, *exc):
            for e, u in zip(self.encs, self.saved):
                e.unpad = u
                e._fold_now = False

    @torch.no_grad()
    def encode_text(self, input_ids_a, token_type_ids_a=None, attention_mask_a=None, position_ids_a=None, packed=True):
        """-> dict(seq bf16 [N, La, H], mask additive f32 [N, La], glob f32 [N, H] unit norm)."""
        if attention_mask_a is None:
            attention_mask_a = torch.ones_like(input_ids_a)
        if token_type_ids_a is None:
            token_type_ids_a = torch.zeros_like(input_ids_a)
        mask_a = additive_mask(attention_mask_a)
        xa = embed_inputs(self.embeddings, input_ids_a, token_type_ids_a, position_ids_a, None, self)
        with self._packed_stacks(self, packed):
            txt = self.txt_encoder(xa, mask_a)[0]
        glob = self._project(self._cls_rows(txt), self.txt_proj)
        return dict(seq=txt, mask=mask_a, glob=glob)

    @torch.no_grad()
    def encode_image(self, input_ids_b, img_feats, token_type_ids_b=None, attention_mask_b=None, position_ids_b=None,
                     max_tag_length=20, use_b=False, packed=True):
        """-> dict(seq bf16 [N, R', H] (tag rows cut as in vl:516-519), mask [N, R'], glob f32 [N, H])."""
        if attention_mask_b is None:
            attention_mask_b = torch.ones((input_ids_b.shape[0], input_ids_b.shape[1] + img_feats.shape[1]),
                                          dtype=torch.long, device=input_ids_b.device)
        if token_type_ids_b is None:
            token_type_ids_b = torch.zeros_like(input_ids_b)
        mask_b = additive_mask(attention_mask_b)
        xb = embed_inputs(self.embeddings, input_ids_b, token_type_ids_b, position_ids_b, img_feats, self)
        with self._packed_stacks(self, packed):
            vis = self.vis_encoder(xb, mask_b)[0]
        cut = 1 if use_b else max_tag_length
        glob = self._project(self._cls_rows(vis), self.vis_proj)
        return dict(seq=vis[:, cut:, :].contiguous(), mask=mask_b[:, cut:].contiguous(), glob=glob)

    @torch.no_grad()
    def fuse_pairs(self, text, image, txt_idx, img_idx, packed=True, pack_hint=None):
        """mul_encoder + pooler on the pairs (text[txt_idx[i]], image[img_idx[i]]) -> (seq, pooled).
        packed=False keeps the padded execution (bit-identical to forward(..., encode_hn=False)).
        pack_hint: (valid rows, longest joint sequence) of these pairs when the caller already has them on the host."""
        joint = torch.cat([text["seq"].index_select(0, txt_idx), image["seq"].index_select(0, img_idx)], dim=1)
        mask = torch.cat([text["mask"].index_select(0, txt_idx), image["mask"].index_select(0, img_idx)], dim=-1)
        with self._packed_stacks(self, packed):
            seq = self.mul_encoder(joint, mask, pack_hint=pack_hint if packed else None)[0]
        return seq, self.pooler(seq)


# ------------------------------------------------------------------------------------------ heads
class BertPreTrainingHeads(nn.Module):
    """vl:970-980."""

    def __init__(self, config, only_vocab=False):
        super().__init__()
        self.predictions = BertLMPredictionHead(config, only_vocab=only_vocab)
        n = config.num_contrast_classes if hasattr(config, "num_contrast_classes") else 2
        self.seq_relationship = nn.Linear(config.hidden_size, n)

    def forward(self, sequence_output, pooled_output):
        return self.predictions(sequence_output), self.seq_relationship(pooled_output.to(self.seq_relationship.weight.dtype))


class BertVQAHeads(nn.Module):
    """vl:983-990."""

    def __init__(self, config):
        super().__init__()
        self.predictions = BertQAPredictionHead(config)

    def forward(self, sequence_output):
        return self.predictions(sequence_output)


def _masked_rows(seq, labels, idx=None):
    """rows of seq [B,L,H] whose label > -1 (vl:1231-1234 masked_select + reshape) and their labels.
    idx: their flat positions when the caller has them already (torch.nonzero is a host sync)."""
    if idx is None:
        keep = (labels > -1).reshape(-1)
        idx = torch.nonzero(keep).squeeze(1)
    rows = seq.reshape(-1, seq.shape[-1]).index_select(0, idx)
    return rows, labels.reshape(-1).index_select(0, idx)


# ------------------------------------------------------------------------------------------- WRA
def _flat_rows(index, seq_len):
    """host-side: [B,2] (start, end) ranges -> flat row ids into [B*L], owner sample per row,
    and per-sample (start offset, count) in the stacked order (vl:1502-1508 semantics)."""
    rows, owner, starts, counts = [], [], [], []
    for i, (s, e) in enumerate(index.tolist()):
        starts.append(len(rows))
        counts.append(max(0, e - s))
        for r in range(s, e):
            rows.append(i * seq_len + r)
            owner.append(i)
    return rows, owner, starts, counts


def mask_slice_and_stack(features, valid_index):
    """vl:1502-1508 — one index_select instead of a per-sample loop (a per-sample slice costs a
    full-size zero-filled gradient buffer each in backward)."""
    B, L, H = features.shape
    rows, _, _, _ = _flat_rows(valid_index, L)
    idx = torch.tensor(rows, dtype=torch.long, device=features.device)
    return features.reshape(B * L, H).index_select(0, idx)


def _draw_top3_picks(counts):
    """The reference draws, per sample t: randint(0,3,(n_t,)) for the positive pair, then
    random.choice of a negative image, then randint again (vl:1566-1576, 1547-1549); samples
    without phrases draw no randint.  Same order here, on the host."""
    n = len(counts)
    pos_pick, neg_pick, neg_img = [], [], []
    for t in range(n):
        if counts[t] > 0:
            pos_pick.append(torch.randint(0, 3, (counts[t],)).cpu())
        j = random.choice(list(range(0, t)) + list(range(t + 1, n)))
        neg_img.append(j)
        if counts[t] > 0:
            neg_pick.append(torch.randint(0, 3, (counts[t],)).cpu())
    cat = lambda xs: torch.cat(xs) if xs else torch.zeros(0, dtype=torch.long)  # noqa: E731
    return cat(pos_pick), cat(neg_pick), neg_img


def _segment_top3_mean(sims, row_owner, col_start, col_count, picks, n):
    """mean over a sample's phrase rows of a random one of the top-3 similarities inside the
    column range [col_start, col_start+col_count) of each row (t2i_sim, vl:1543-1550)."""
    dev = sims.device
    rmax = int(col_count.max().item()) if col_count.numel() else 0
    ar = engine.arange(max(rmax, 3), dev)
    cols = col_start[:, None] + ar[None, :]
    valid = ar[None, :] < col_count[:, None]
    vals = sims.gather(1, cols.clamp(max=sims.shape[1] - 1)).masked_fill(~valid, float("-inf"))
    top = vals.topk(3, dim=1)[0]
    picked = top.gather(1, picks.to(dev)[:, None]).squeeze(1)
    sums = torch.zeros(n, dtype=sims.dtype, device=dev).index_add_(0, row_owner, picked)
    cnt = torch.zeros(n, dtype=sims.dtype, device=dev).index_add_(0, row_owner, torch.ones_like(picked))
    return sums / cnt.clamp(min=1.0)


def get_pos_neg_sims(sims, text_index, img_index, draws=None):
    """vl:1553-1596, vectorised: per sample the mean top-3-pick similarity of its phrases against
    its own regions (pos) and against one randomly chosen other image (neg).  draws: optional
    (pos_pick, neg_pick, neg_img) in the stacked-phrase order instead of fresh host draws."""
    dev = sims.device
    n = text_index.shape[0]
    _, owner, _, tcounts = _flat_rows(text_index, 1 << 20)
    _, _, istarts, icounts = _flat_rows(img_index, 1 << 20)
    if any(c < 3 for c in icounts) and len(owner):
        raise RuntimeError("selected index k out of range: every image needs >= 3 valid regions (topk(3), vl:1547)")
    pos_pick, neg_pick, neg_img = _draw_top3_picks(tcounts) if draws is None else draws
    owner_t = torch.tensor(owner, dtype=torch.long, device=dev)
    istarts_t = torch.tensor(istarts, dtype=torch.long, device=dev)
    icounts_t = torch.tensor(icounts, dtype=torch.long, device=dev)
    neg_t = torch.tensor(neg_img, dtype=torch.long, device=dev)
    pos = _segment_top3_mean(sims, owner_t, istarts_t[owner_t], icounts_t[owner_t], pos_pick, n)
    nj = neg_t[owner_t]
    neg = _segment_top3_mean(sims, owner_t, istarts_t[nj], icounts_t[nj], neg_pick, n)
    return pos, neg


def wra_sample_on_device(seq, phrase_index, img_index, text_len, draws=None, max_phrases=None):
    """The phrase_mod='sample' branch of vl:1285-1300 + get_pos_neg_sims (vl:1553-1596) with every
    shape fixed by the tensor shapes — no `.tolist()`, no host loops, no device->host copy: the
    host version costs ~2 ms of Python per 256-pair step during which the GPU queue runs dry.
    Phrase rows are rows [p0, p1) of the joint sequence, gathered into a [B, Pw, H] grid with
    Pw = max_phrases (config.max_phrases, the data pipeline's --max_phrases) or, when that is not
    known, text_len; region rows the `Lj - text_len` rows from i0.  Only the gathered rows are
    converted to f32 and normalised.  The random draws (one of the top-3 regions per phrase,
    vl:1547-1549; one other image per sample, vl:1572-1573) come from the device generator: same
    distributions as the reference, different stream.  draws: optional (pos_pick [B, Pw],
    neg_pick [B, Pw], neg_img [B]) for tests, pick j belonging to phrase row p0 + j.
    -> pos_sims [B], neg_sims [B] (0 where a sample has no phrase, as the reference)."""
    B, Lj, H = seq.shape
    dev = seq.device
    Rw = Lj - text_len
    Pw = int(max_phrases) if max_phrases else text_len
    p0, p1, i0, i1 = phrase_index[:, 0], phrase_index[:, 1], img_index[:, 0], img_index[:, 1]
    if max_phrases:
        hip.flag_device_error_if(((p1 - p0) > Pw).any(), hip.DEV_ERR_PHRASES)      # raised at the host's next read-back (hip.py)
    ar_p = engine.arange(Pw, dev)
    rows_p = (p0[:, None] + ar_p[None, :]).clamp(max=Lj - 1)                              # [B, Pw]
    valid_p = ar_p[None, :] < (p1 - p0)[:, None]
    ar_r = engine.arange(Rw, dev)
    rows_r = (i0[:, None] + ar_r[None, :]).clamp(max=Lj - 1)                              # [B, Rw]
    valid_r = ar_r[None, :] < (i1 - i0)[:, None]
    txt_n = F.normalize(seq.gather(1, rows_p[:, :, None].expand(-1, -1, H)).float(), p=2, dim=-1)
    reg_n = F.normalize(seq.gather(1, rows_r[:, :, None].expand(-1, -1, H)).float(), p=2, dim=-1)
    if draws is None:
        pos_pick = torch.randint(0, 3, (B, Pw), device=dev)
        neg_pick = torch.randint(0, 3, (B, Pw), device=dev)
        neg_img = (engine.arange(B, dev) + 1 + torch.randint(0, max(B - 1, 1), (B,), device=dev)) % B
    else:
        pos_pick, neg_pick, neg_img = (d.to(dev) for d in draws)
    cnt = valid_p.sum(1).clamp(min=1).to(txt_n.dtype)

    def mean_top3(regions, rvalid, pick):
        sims = torch.bmm(txt_n, regions.transpose(1, 2)).masked_fill(~rvalid[:, None, :], float("-inf"))
        top = sims.topk(3, dim=2)[0]                                                      # [B, Pw, 3]
        picked = top.gather(2, pick[:, :, None]).squeeze(2)
        return torch.where(valid_p, picked, torch.zeros_like(picked)).sum(1) / cnt

    pos = mean_top3(reg_n, valid_r, pos_pick)
    neg = mean_top3(reg_n.index_select(0, neg_img), valid_r.index_select(0, neg_img), neg_pick)
    return pos, neg


def _wra_from_rows(txt_n, reg_n, valid_p, valid_r, draws=None):
    """pos / neg similarities of wra_sample_on_device from already gathered, normalised rows: txt_n f32 [B, Pw, H]
    (phrase rows, zero rows where invalid), reg_n f32 [B, Rw, H], validity masks; device draws (vl:1547-1549,
    1572-1573: one of the top-3 regions per phrase, one other image per sample)."""
    B, Pw, _ = txt_n.shape
    dev = txt_n.device
    if draws is None:
        pos_pick = torch.randint(0, 3, (B, Pw), device=dev)
        neg_pick = torch.randint(0, 3, (B, Pw), device=dev)
        neg_img = (engine.arange(B, dev) + 1 + torch.randint(0, max(B - 1, 1), (B,), device=dev)) % B
    else:
        pos_pick, neg_pick, neg_img = (d.to(dev) for d in draws)
    cnt = valid_p.sum(1).clamp(min=1).to(txt_n.dtype)

    def mean_top3(regions, rvalid, pick):
        sims = torch.bmm(txt_n, regions.transpose(1, 2)).masked_fill(~rvalid[:, None, :], float("-inf"))
        top = sims.topk(3, dim=2)[0]
        picked = top.gather(2, pick[:, :, None]).squeeze(2)
        return torch.where(valid_p, picked, torch.zeros_like(picked)).sum(1) / cnt

    pos = mean_top3(reg_n, valid_r, pos_pick)
    neg = mean_top3(reg_n.index_select(0, neg_img), valid_r.index_select(0, neg_img), neg_pick)
    return pos, neg


def t2i_sim(sim_matrix):
    """vl:1543-1550 (single matrix form, used by phrase_mod='hard')."""
    if sim_matrix.shape[0] == 0:
        return torch.zeros((), dtype=sim_matrix.dtype, device=sim_matrix.device)
    f_sim = sim_matrix.topk(3, dim=1)[0]
    rand_index = torch.randint(0, 3, (f_sim.shape[0],)).to(f_sim.device)
    return f_sim[engine.arange(f_sim.shape[0], f_sim.device), rand_index].mean()


def get_pos_sims(sequence_output, text_index, img_index):
    """vl:1510-1527."""
    out = []
    ti, ii = text_index.tolist(), img_index.tolist()
    for i in range(len(ti)):
        f = sequence_output[i].float()
        t = F.normalize(f[ti[i][0]:ti[i][1]], p=2, dim=-1)
        v = F.normalize(f[ii[i][0]:ii[i][1]], p=2, dim=-1)
        out.append(t2i_sim(t @ v.t()) if t.shape[0] else torch.zeros((), dtype=v.dtype, device=v.device))
    return torch.stack(out)


# ------------------------------------------------------------------------------------ task models
class BertImgForPreTraining(ImgPreTrainedModel):
    """vl:1024-1130 — single-stream MLM + ITM."""
    config_class = BertConfig
    base_model_prefix = "bert"

    def __init__(self, config):
        super().__init__(config)
        self.bert = BertImgModel(config)
        self.cls = BertPreTrainingHeads(config)
        self.num_seq_relations = config.num_contrast_classes if hasattr(config, "num_contrast_classes") else 2
        self.max_text_seq_length = config.max_text_seq_length if hasattr(config, "max_text_seq_length") else None
        # True (default): the reference's output tuple, prediction_scores [B, T, V] included.  False:
        # loss-only training — masked rows through the fused decoder + cross-entropy kernels
        self.return_prediction_scores = True
        self.apply(self.init_weights)
        self.tie_weights()

    def tie_weights(self):
        self._tie_or_clone_weights(self.cls.predictions.decoder, self.bert.embeddings.word_embeddings)

    def forward(self, input_ids, token_type_ids=None, attention_mask=None, masked_lm_labels=None,
                next_sentence_label=None, position_ids=None, head_mask=None, img_feats=None, host_counts=None):
        """host_counts (optional, not a reference argument): dict(rows, lmax, scored) computed where the batch was built
        (synthetic.synthetic_batch(single_stream=True)) — the loss-only training step then reads nothing back from the device."""
        hint = (int(host_counts["rows"]), int(host_counts["lmax"])) if host_counts is not None else None
        outputs = self.bert(input_ids, position_ids=position_ids, token_type_ids=token_type_ids,
                            attention_mask=attention_mask, head_mask=head_mask, img_feats=img_feats, pack_hint=hint)
        sequence_output, pooled_output = outputs[:2]
        extras = tuple(outputs[2:])       # vl:1116: hidden states / attentions when the configuration asks for them
        T = self.max_text_seq_length
        text = sequence_output[:, :T, :] if T is not None else sequence_output
        seq_relationship_score = self.cls.seq_relationship(pooled_output)
        if masked_lm_labels is None or next_sentence_label is None:
            return (self.cls.predictions(text), seq_relationship_score) + extras
        labels = masked_lm_labels[:, :T].contiguous() if T is not None else masked_lm_labels
        if not self.return_prediction_scores:
            # training loops that read outputs[0] only (run_oscarplus_pretrain-style): the head runs on
            # the scored rows alone and the logits never reach HBM; prediction_scores comes back empty
            flat = labels.reshape(-1)
            if host_counts is not None and flat.is_cuda:
                # the scored-row count came with the batch: no read-back; a wrong count shows in the device error word
                n_sc = int(host_counts["scored"])
                hip.flag_device_error_if((flat >= 0).sum() != n_sc, hip.DEV_ERR_SCORED_ROWS)
                idx = torch.nonzero_static(flat >= 0, size=n_sc, fill_value=0).squeeze(1)
            else:
                idx = torch.nonzero(flat >= 0).squeeze(1)
            rows = text.reshape(-1, text.shape[-1]).index_select(0, idx)
            masked_lm_loss, scores = self.cls.predictions.loss_and_scores(rows, flat.index_select(0, idx), want_scores=False)
            prediction_scores = scores
        else:
            masked_lm_loss, scores = self.cls.predictions.loss_and_scores(text.reshape(-1, text.shape[-1]), labels.reshape(-1))
            prediction_scores = scores.reshape(text.shape[0], text.shape[1], -1)
        loss_fct = CrossEntropyLoss(ignore_index=-1)
        next_sentence_loss = loss_fct(seq_relationship_score.view(-1, self.num_seq_relations), next_sentence_label.view(-1))
        total_loss = masked_lm_loss + next_sentence_loss
        return (total_loss, prediction_scores, seq_relationship_score) + extras + (masked_lm_loss,)


class BiBertImgForPreTraining(ImgPreTrainedModel):
    """vl:1133-1311 — the model oscar/run_pretrain_ml.py trains (:25,324): masked-concept MLM on
    tags, CLIP-style retrieval loss, text MLM, ITM on [matched ; hard-negative] pairs, WRA."""
    config_class = BertConfig
    base_model_prefix = "bert"

    def __init__(self, config):
        super().__init__(config)
        self.bert = BiBertImgModel(config)
        self.cls = BertPreTrainingHeads(config, only_vocab=True)
        self.half_mlm = BertLMPredictionHead(config, only_vocab=True)
        self.qa_head = HeadLinear(config.hidden_size, config.qa_answer_size)
        self.only_vocab_size = config.only_word_size
        self.num_seq_relations = config.num_contrast_classes if hasattr(config, "num_contrast_classes") else 2
        self.max_text_seq_length = config.max_text_seq_length if hasattr(config, "max_text_seq_length") else None
        self.logit_scale = nn.Parameter(torch.ones([]) * np.log(1 / 0.07))
        # WRA random draws on the device (no host round trip); False = host draws in the reference's
        # order, which the parity tests replay from the golden fixtures
        self.wra_on_device = True
        # Heads on the second stream (2, default): the visual-MLM and contrastive heads only need the uni-modal outputs and
        # are queued beside the joint stack; ITM / QA / WRA (chains of small kernels) run beside the text MLM head's
        # vocabulary GEMMs.  config.heads_beside: 1 = only the first pair, 0 = everything on one stream.
        self.heads_beside = int(getattr(config, "heads_beside", 2))
        # training steps run the row-packed pipeline (BiBertImgModel.forward_packed + tapped rows for every head);
        # config.packed_pipeline = False: the general path through padded tensors (same results to rounding)
        self.packed_pipeline = bool(getattr(config, "packed_pipeline", True))
        self.apply(self.init_weights)
        self.tie_weights()

    def _forward_packed(self, input_ids_a, token_type_ids_a, attention_mask_a, masked_lm_labels_a, qa_ans, input_ids_b,
                        token_type_ids_b, attention_mask_b, masked_lm_labels_b, max_tag_length, position_ids_a, position_ids_b,
                        img_feats, img_index, phrase_index, host_counts=None):
        """forward() for a training step, on packed rows end to end (vl:1218-1311): the backbone hands back the packed
        joint output and row maps; every head reads its rows through ONE engine.MultiTapFn call per stack output.
        Same losses as forward() (tests/test_model_gpu.py::test_packed_pipeline_equals_general_path)."""
        dev = input_ids_a.device
        n, La = input_ids_a.shape
        from .. import hip
        if host_counts is not None:
            n_scored = (int(host_counts["scored_a"]), int(host_counts["scored_b"]))
        else:
            scored = engine.AsyncCounts([(masked_lm_labels_a > -1).sum(), (masked_lm_labels_b > -1).sum()])
            n_scored = None
        early = {}

        def uni_taps(pos_a, pos_b):
            # labels and packed rows of the scored tag slots, ascending, in one launch (mvptr_compact_scored)
            early["labels_b"], rows_b = hip.compact_scored(masked_lm_labels_b.contiguous(), pos_b, (n_scored or scored.get())[1])
            return [], [rows_b]      # masked tag rows of the packed visual output

        def uni_heads(taps_txt, taps_vis, sim_mat):
            early["vis_mlm"], _ = self.half_mlm.loss_and_scores(taps_vis[0], early["labels_b"], want_scores=False)
            early["retrieval"] = engine.ContrastiveLossFn.apply(sim_mat, self.logit_scale)

        bb = self.bert
        out = bb.forward_packed(input_ids_a, token_type_ids_a, attention_mask_a, max_tag_length, input_ids_b, token_type_ids_b,
                                attention_mask_b, img_feats, position_ids_a=position_ids_a, position_ids_b=position_ids_b,
                                uni_taps=uni_taps, beside=uni_heads, host_counts=host_counts)
        two_streams = bool(bb.parallel_stacks) and _streams_allowed(bb.parallel_stacks)
        main = torch.cuda.current_stream(dev)
        side = engine.side_stream(dev) if two_streams else None
        if two_streams:
            main.wait_stream(side)
            for t in (early["vis_mlm"], early["retrieval"]):
                t.record_stream(main)
        both, pos_j = out["both"], out["pos_j"]
        Lj = pos_j.shape[1]
        # rows of the packed joint output the heads read: [CLS] of the 2n sequences, scored text rows, phrase / region rows
        # scored text slots of the n matched pairs: their labels and their rows in the packed joint output (pos_j[:n, :La])
        labels_a, rows_mlm = hip.compact_scored(masked_lm_labels_a.contiguous(), pos_j, (n_scored or scored.get())[0])
        idxs = [pos_j[:, 0].contiguous(), rows_mlm]
        wra = phrase_index is not None
        if wra:
            Pw = int(getattr(self.config, "max_phrases", None) or La)
            Rw = Lj - La
            if getattr(self.config, "max_phrases", None):
                hip.flag_device_error_if(((phrase_index[:, 1] - phrase_index[:, 0]) > Pw).any(), hip.DEV_ERR_PHRASES)
            # topk(3) of the reference (vl:1547) raises — catchably, and only for a sample that HAS phrases (t2i_sim returns 0
            # for an empty phrase set) — when an image has fewer than 3 regions.  A device-side assert (rounds 3-5) aborted the whole
            # process on ROCm for any such image: the kernel clamps the drawn rank to the regions that exist
            # (csrc/wra.hip) unless config.wra_strict asks for the reference's failure, restricted to samples with phrases.
            if getattr(self.config, "wra_strict", False):
                few = ((img_index[:, 1] - img_index[:, 0]) < 3) & ((phrase_index[:, 1] - phrase_index[:, 0]) > 0)
                hip.flag_device_error_if(few.any(), hip.DEV_ERR_FEW_REGIONS)
            rows_p, rows_r = hip.wra_rows(pos_j, phrase_index, img_index, n, Pw, Rw)
            idxs += [rows_p.view(-1), rows_r.view(-1)]
        taps = engine.MultiTapFn.apply(both, None, *idxs)
        late = {}

        def small_heads():
            pooled = bb.pooler.forward_rows(taps[0])                     # [2n, H]: matched then hard pairs
            score = engine.SmallLinearFn.apply(pooled, self.cls.seq_relationship.weight, self.cls.seq_relationship.bias, None, False)
            label = engine.arange(2 * n, dev, torch.long, floor_div=n)     # n zeros (matched pairs) then n ones (hard pairs), cached
            late["itm"] = engine.CeMeanFn.apply(score.view(-1, self.num_seq_relations), label)
            if qa_ans is not None:
                late["qa"] = CrossEntropyLoss(ignore_index=-1)(self.qa_head(pooled[:n]), qa_ans)
            if wra:
                H = both.shape[1]
                # the reference's draws (vl:1547-1549 one of the top-3 regions per phrase, vl:1572-1573 one other image
                # per sample) from the device generator, in wra_sample_on_device's order
                pos_pick = torch.randint(0, 3, (n, Pw), device=dev)
                neg_pick = torch.randint(0, 3, (n, Pw), device=dev)
                neg_img = (engine.arange(n, dev) + 1 + torch.randint(0, max(n - 1, 1), (n,), device=dev)) % n
                late["wra"] = engine.WraLossFn.apply(taps[2].view(n, Pw, H), taps[3].view(n, Rw, H), phrase_index, img_index,
                                                     pos_pick, neg_pick, neg_img)

        use_side = self.heads_beside >= 2 and two_streams
        if use_side:
            side.wait_stream(main)
            with torch.cuda.stream(side):
                small_heads()
            for t in taps:
                t.record_stream(side)
        else:
            small_heads()
        masked_lm_loss, _ = self.cls.predictions.loss_and_scores(taps[1], labels_a, want_scores=False)
        if use_side:
            main.wait_stream(side)
            for t in late.values():
                t.record_stream(main)
        vis_mlm_loss, retrieval_loss, next_sentence_loss = early["vis_mlm"], early["retrieval"], late["itm"]
        total_loss = vis_mlm_loss + retrieval_loss + masked_lm_loss + next_sentence_loss
        outs = (vis_mlm_loss, retrieval_loss, masked_lm_loss, next_sentence_loss)
        if qa_ans is not None:
            total_loss = total_loss + late["qa"]
            outs = outs + (late["qa"],)
        if wra:
            total_loss = total_loss + late["wra"]
            return (total_loss,) + outs + (late["wra"],)
        return (total_loss,) + outs

    def tie_weights(self):
        emb = self.bert.embeddings.word_embeddings
        self._tie_or_clone_weights(self.cls.predictions.decoder, emb, only_vocab=True, only_word_size=self.only_vocab_size)
        self._tie_or_clone_weights(self.half_mlm.decoder, emb, only_vocab=True, only_word_size=self.only_vocab_size)

    def forward(self, input_ids_a, token_type_ids_a=None, attention_mask_a=None, masked_lm_labels_a=None,
                qa_ans=None, input_ids_b=None, token_type_ids_b=None, attention_mask_b=None,
                masked_lm_labels_b=None, max_tag_length=20, position_ids_a=None, position_ids_b=None,
                head_mask=None, img_feats=None, is_img_match=None, img_index=None, phrase_index=None,
                phrase_mod="sample", host_counts=None):
        """host_counts (optional, not a reference argument): synthetic.host_counts(batch) — the input-only counts a step
        needs on the host (valid rows / longest sequence of both inputs, scored MLM rows), computed where the batch was
        built, so that the row-packed training step does not read them back from the device.  They MUST describe this
        batch: the device checks them against the masks and the labels (mvptr_check_counts: rows / longest sequence;
        mvptr_compact_scored: no more scored rows than slots).  Wrong row / longest-sequence counts TRAP (every buffer behind
        them is sized from the host's numbers: the process aborts with both sets printed; `model.bert.verify_host_counts = True`
        checks on the host with one read-back and raises ValueError instead); too many scored rows raise RuntimeError at the
        host's next count read-back (hip.device_error_word); train.model_inputs rejects counts that cannot fit the batch's shapes
        before anything is queued.
        Recompute them (synthetic.host_counts) whenever a collated batch is edited, or leave the argument out."""
        if (self.packed_pipeline and self.training and masked_lm_labels_a is not None and masked_lm_labels_b is not None and
                head_mask is None and phrase_mod == "sample" and (phrase_index is None or (img_index is not None and self.wra_on_device)) and
                self.bert.packed_ok(attention_mask_a, attention_mask_b, input_ids_a)):
            return self._forward_packed(input_ids_a, token_type_ids_a, attention_mask_a, masked_lm_labels_a, qa_ans, input_ids_b,
                                        token_type_ids_b, attention_mask_b, masked_lm_labels_b, max_tag_length, position_ids_a,
                                        position_ids_b, img_feats, img_index, phrase_index, host_counts)
        # Every data-dependent COUNT that only depends on the inputs (scored rows of the two MLM heads,
        # valid rows / longest sequence of the two uni-modal stacks) is fetched in ONE device->host copy
        # here, before any encoder work is queued: a sync in the middle of the step drains the launch
        # queue and leaves the GPU idle until the host has caught up (four of them: ~1 ms per step).
        idx_a = idx_b = pack_hints = None
        enc = self.bert.txt_encoder
        if (masked_lm_labels_a is not None and masked_lm_labels_b is not None and masked_lm_labels_a.is_cuda and
                attention_mask_a is not None and attention_mask_b is not None and
                (enc.unpad is True or (enc.unpad == "train" and self.training))):
            keep_a, keep_b = (masked_lm_labels_a > -1).reshape(-1), (masked_lm_labels_b > -1).reshape(-1)
            la, lb = attention_mask_a.sum(1), attention_mask_b.sum(1)
            pack_hints = engine.AsyncCounts([keep_a.sum(), keep_b.sum(), la.sum(), la.max(), lb.sum(), lb.max()])
        ce_loss = CrossEntropyLoss(ignore_index=-1)
        early = {}

        def uni_heads(txt_out, vis_out, sim_mat):
            """The two losses that only need the uni-modal outputs: masked-concept / visual MLM and the contrastive
            loss.  Run from inside the backbone, beside the joint stack (BiBertImgModel.forward, `beside`)."""
            ib = None
            if pack_hints is not None:
                ib = torch.nonzero_static(keep_b, size=pack_hints.get()[1]).view(-1)      # counts landed in the backbone
            vis_rows, vis_labels = _masked_rows(vis_out, masked_lm_labels_b, ib)
            early["vis_mlm"], _ = self.half_mlm.loss_and_scores(vis_rows, vis_labels, want_scores=False)
            logits = sim_mat * self.logit_scale.exp()
            pseudo = engine.arange(sim_mat.shape[0], sim_mat.device)
            early["retrieval"] = (ce_loss(logits, pseudo) + ce_loss(logits.t(), pseudo)) / 2

        outputs, single, hard_indexes = self.bert(
            input_ids_a=input_ids_a, position_ids_a=position_ids_a, token_type_ids_a=token_type_ids_a,
            attention_mask_a=attention_mask_a, head_mask=head_mask, img_feats=img_feats, input_ids_b=input_ids_b,
            position_ids_b=position_ids_b, token_type_ids_b=token_type_ids_b, attention_mask_b=attention_mask_b,
            max_tag_length=max_tag_length, encode_hn=True, pack_hints=pack_hints,
            beside=uni_heads if self.heads_beside >= 1 else None)
        txt_out, vis_out, sim_mat = single
        if pack_hints is not None:
            idx_a = torch.nonzero_static(keep_a, size=pack_hints.get()[0]).view(-1)      # landed long ago (awaited in the backbone)
        if not early:
            uni_heads(txt_out, vis_out, sim_mat)
        elif sim_mat.is_cuda:
            main = torch.cuda.current_stream(sim_mat.device)
            main.wait_stream(engine.side_stream(sim_mat.device))
            for t in early.values():
                t.record_stream(main)
        vis_mlm_loss, retrieval_loss = early["vis_mlm"], early["retrieval"]

        sequence_output, pooled_output, hard_sequence_output, hard_pooled_output = outputs
        late = {}

        def small_heads():
            """ITM, QA and word-region alignment: chains of small kernels on the joint output; run beside the text MLM
            head's vocabulary GEMMs (second stream) when streams are in use."""
            both_pooled = torch.cat([pooled_output, hard_pooled_output], dim=0)
            n = pooled_output.shape[0]
            dev = both_pooled.device      # built on the device: a pageable host->device copy is a sync
            next_sentence_label = torch.cat([torch.zeros(n, dtype=torch.long, device=dev), torch.ones(n, dtype=torch.long, device=dev)])
            sr = self.cls.seq_relationship
            if both_pooled.is_cuda and sr.weight.dtype == torch.float32 and self.num_seq_relations <= 64:
                # the 2-way head and its mean cross entropy on the f32 HIP kernels, as the packed pipeline runs them
                score = engine.SmallLinearFn.apply(both_pooled, sr.weight, sr.bias, None, False)
                late["itm"] = engine.CeMeanFn.apply(score.view(-1, self.num_seq_relations), next_sentence_label)
            else:
                late["itm"] = ce_loss(sr(both_pooled).view(-1, self.num_seq_relations), next_sentence_label.view(-1))
            if qa_ans is not None:
                late["qa"] = ce_loss(self.qa_head(pooled_output), qa_ans)
            if phrase_index is None:
                return
            if phrase_mod == "hard":
                hard_txt_index, hard_img_index = hard_indexes
                hard_phrase_index = phrase_index.index_select(0, hard_txt_index)
                hard_object_index = img_index.index_select(0, hard_img_index)
                pos_sims = get_pos_sims(sequence_output, phrase_index, img_index)
                neg_sims = get_pos_sims(hard_sequence_output, hard_phrase_index, hard_object_index)
                valid = ((phrase_index[:, 1] - phrase_index[:, 0]) > 0) & ((hard_phrase_index[:, 1] - hard_phrase_index[:, 0]) > 0)
            elif phrase_mod == "sample":
                if self.wra_on_device:
                    pos_sims, neg_sims = wra_sample_on_device(sequence_output, phrase_index, img_index, input_ids_a.shape[1],
                                                              max_phrases=getattr(self.config, "max_phrases", None))
                else:  # host draws in the reference's order (replayable: parity tests)
                    seq32 = sequence_output.float()
                    valid_phrases = F.normalize(mask_slice_and_stack(seq32, phrase_index), p=2, dim=-1)
                    valid_images = F.normalize(mask_slice_and_stack(seq32, img_index), p=2, dim=-1)
                    pos_sims, neg_sims = get_pos_neg_sims(valid_phrases @ valid_images.t(), phrase_index, img_index)
                valid = (phrase_index[:, 1] - phrase_index[:, 0]) > 0
            else:
                raise NotImplementedError
            hinge = torch.clamp(neg_sims + 0.2 - pos_sims, min=0)
            # mean over the samples that have phrases (vl:1298-1300) without a data-dependent shape
            late["wra"] = torch.where(valid, hinge, torch.zeros_like(hinge)).sum() / valid.sum().to(hinge.dtype)

        bb = self.bert
        use_side = (self.heads_beside >= 2 and sequence_output.is_cuda and bool(bb.parallel_stacks) and _streams_allowed(bb.parallel_stacks))
        if use_side:
            main = torch.cuda.current_stream(sequence_output.device)
            side = engine.side_stream(sequence_output.device)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                small_heads()
            for t in (sequence_output, pooled_output, hard_pooled_output, hard_sequence_output):
                if t is not None:
                    t.record_stream(side)
        else:
            small_heads()
        rows, labels = _masked_rows(sequence_output[:, :input_ids_a.shape[1], :], masked_lm_labels_a, idx_a)
        masked_lm_loss, _ = self.cls.predictions.loss_and_scores(rows, labels, want_scores=False)
        if use_side:
            main.wait_stream(side)
            for t in late.values():
                t.record_stream(main)
        next_sentence_loss = late["itm"]

        total_loss = vis_mlm_loss + retrieval_loss + masked_lm_loss + next_sentence_loss
        outs = (vis_mlm_loss, retrieval_loss, masked_lm_loss, next_sentence_loss)
        if qa_ans is not None:
            total_loss = total_loss + late["qa"]
            outs = outs + (late["qa"],)
        if phrase_index is not None:
            total_loss = total_loss + late["wra"]
            return (total_loss,) + outs + (late["wra"],)
        return (total_loss,) + outs


def _make_classifier(config, num_labels):
    if hasattr(config, "classifier"):
        if not hasattr(config, "cls_hidden_scale"):
            config.cls_hidden_scale = 2
        if config.classifier == "linear":
            return HeadLinear(config.hidden_size, num_labels)
        if config.classifier == "mlp":
            return nn.Sequential(HeadLinear(config.hidden_size, config.hidden_size * config.cls_hidden_scale), nn.ReLU(),
                                 HeadLinear(config.hidden_size * config.cls_hidden_scale, num_labels))
    return HeadLinear(config.hidden_size, num_labels)


def _bi_kwargs(kw):
    return dict(input_ids_a=kw["input_ids_a"], position_ids_a=kw.get("position_ids_a"),
                token_type_ids_a=kw.get("token_type_ids_a"), attention_mask_a=kw.get("attention_mask_a"),
                head_mask=kw.get("head_mask"), img_feats=kw.get("img_feats"), input_ids_b=kw.get("input_ids_b"),
                position_ids_b=kw.get("position_ids_b"), token_type_ids_b=kw.get("token_type_ids_b"),
                attention_mask_b=kw.get("attention_mask_b"), max_tag_length=kw.get("max_tag_length", 20))


class BiImageBertForRetrieval(BertPreTrainedModel):
    """vl:1598-1712 — `forward_mod` in {'train','coarse','fine'} (run_retrieval.py:598-601)."""

    def __init__(self, config):
        super().__init__(config)
        self.num_labels = 2
        self.loss_type = config.loss_type
        self.config = config
        if config.img_feature_dim <= 0:
            raise NotImplementedError("text-only BertModel is outside the accelerated path")
        self.bert = BiBertImgModel(config)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)
        self.logit_scale = nn.Parameter(torch.ones([]) * np.log(1 / 0.07))
        self.forward_mod = "train"
        self.classifier = _make_classifier(config, self.config.num_labels)
        self.apply(self.init_weights)

    def reinit_cls_head(self):
        self.classifier.apply(self.init_weights)

    def forward(self, input_ids_a, token_type_ids_a=None, attention_mask_a=None, input_ids_b=None,
                token_type_ids_b=None, attention_mask_b=None, max_tag_length=20, position_ids_a=None,
                position_ids_b=None, head_mask=None, img_feats=None):
        kw = dict(input_ids_a=input_ids_a, token_type_ids_a=token_type_ids_a, attention_mask_a=attention_mask_a,
                  input_ids_b=input_ids_b, token_type_ids_b=token_type_ids_b, attention_mask_b=attention_mask_b,
                  img_feats=img_feats, max_tag_length=max_tag_length, position_ids_a=position_ids_a,
                  position_ids_b=position_ids_b, head_mask=head_mask)
        if self.forward_mod == "train":
            return self.forward_train(**kw)
        if self.forward_mod == "coarse":
            return self.forward_emb(**kw)
        if self.forward_mod == "fine":
            return self.forward_fine(**kw)
        raise NotImplementedError

    def forward_train(self, **kw):
        outputs, single, _ = self.bert(encode_hn=True, **_bi_kwargs(kw))
        sim_mat = single[2]
        ce_loss = CrossEntropyLoss(ignore_index=-1)
        logits = sim_mat * self.logit_scale.exp()
        pseudo = engine.arange(sim_mat.shape[0], sim_mat.device)
        retrieval_loss = (ce_loss(logits, pseudo) + ce_loss(logits.t(), pseudo)) / 2
        _, pooled, _, hard_pooled = outputs
        score = self.classifier(self.dropout(torch.cat([pooled, hard_pooled], dim=0)))
        n = pooled.shape[0]
        label = torch.cat([torch.ones(n, dtype=torch.long, device=score.device), torch.zeros(n, dtype=torch.long, device=score.device)])
        itm_loss = ce_loss(score.view(-1, self.num_labels), label.view(-1))
        return (retrieval_loss + itm_loss, score, retrieval_loss, itm_loss, label)

    def forward_emb(self, **kw):
        k = _bi_kwargs(kw)
        return tuple(self.bert.forward_single(**k))

    def forward_fine(self, **kw):
        outputs, _, _ = self.bert(encode_hn=False, **_bi_kwargs(kw))
        return self.classifier(outputs[1])

    # cached two-stage evaluation: encode every caption / image once, then score pairs
    def encode_text(self, **kw):
        return self.bert.encode_text(**kw)

    def encode_image(self, **kw):
        return self.bert.encode_image(**kw)

    @torch.no_grad()
    def coarse_scores(self, text, image):
        """[n_text, n_image] cosine similarities of the global embeddings (forward_mod 'coarse' +
        the matrix product of run_retrieval.py:741-745)."""
        return self.bert._sim(text["glob"], image["glob"])

    @torch.no_grad()
    def rerank(self, text, image, txt_idx, img_idx, chunk=4096, packed=True):
        """ITM logits [n_pairs, num_labels] of forward_mod 'fine' for the listed pairs, computed from
        the cached uni-modal outputs.  packed=True skips the padded slots of the joint sequences
        (equal to 'fine' up to bf16 rounding); packed=False reproduces 'fine' bit for bit."""
        out = []
        n = txt_idx.numel()
        hints = None
        if packed and n > 0:
            # (rows, longest joint sequence) of every chunk in ONE read-back for the whole call: the row-packed stack would
            # otherwise fetch its two counts chunk by chunk, and the host could never queue a chunk ahead of the GPU
            # (configs[3]: 110 chunks; 147 k pairs/s on an idle host against 91 k on a loaded one before this)
            lens = (text["mask"] == 0).sum(1, dtype=torch.int32).index_select(0, txt_idx) + \
                   (image["mask"] == 0).sum(1, dtype=torch.int32).index_select(0, img_idx)
            nch = (n + chunk - 1) // chunk
            lens = torch.nn.functional.pad(lens, (0, nch * chunk - n)).view(nch, chunk)
            hints = torch.stack([lens.sum(1), lens.max(1).values], 1).tolist()
        for c, s0 in enumerate(range(0, n, chunk)):
            _, pooled = self.bert.fuse_pairs(text, image, txt_idx[s0:s0 + chunk], img_idx[s0:s0 + chunk], packed=packed,
                                             pack_hint=None if hints is None else tuple(hints[c]))
            out.append(self.classifier(pooled))
        return torch.cat(out, 0)


def _cls_loss(self, logits, labels, soft_label):
    """shared by the VE / VQA wrappers (vl:1777-1797, vl:1849-1869)."""
    if self.num_labels == 1:
        return MSELoss()(logits.view(-1), labels.to(torch.float).view(-1))
    if soft_label:
        return soft_cross_entropy(labels, logits)
    if self.loss_type == "kl":
        log_p = torch.nn.LogSoftmax(dim=-1)(logits.contiguous().view(-1, 3129))
        return torch.nn.KLDivLoss(reduction="batchmean")(log_p, labels.contiguous())
    if self.loss_type == "bce":
        return instance_bce_with_logits(logits, labels)
    return CrossEntropyLoss()(logits.view(-1, self.num_labels), labels.view(-1))


class _FreezeMixin:
    def freeze_backbone(self):
        for p in self.bert.parameters():
            p.requires_grad = False

    def unfreeze_backbone(self):
        for p in self.bert.parameters():
            p.requires_grad = True


class BiImageBertForSequenceClassification(_FreezeMixin, BertPreTrainedModel):
    """vl:1715-1798 — VE (and VQA without --use_pretrain)."""

    def __init__(self, config):
        super().__init__(config)
        self.num_labels = config.num_labels
        self.loss_type = config.loss_type
        self.config = config
        if config.img_feature_dim <= 0:
            raise NotImplementedError("text-only BertModel is outside the accelerated path")
        self.bert = BiBertImgModel(config)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)
        self.classifier = _make_classifier(config, self.config.num_labels)
        self.apply(self.init_weights)

    def reinit_cls_head(self):
        self.classifier.apply(self.init_weights)

    def forward(self, input_ids_a, token_type_ids_a=None, attention_mask_a=None, labels=None, input_ids_b=None,
                token_type_ids_b=None, attention_mask_b=None, max_tag_length=20, use_b=False,
                position_ids_a=None, position_ids_b=None, head_mask=None, img_feats=None, soft_label=False):
        outputs, _, _ = self.bert(input_ids_a=input_ids_a, position_ids_a=position_ids_a, token_type_ids_a=token_type_ids_a,
                                  attention_mask_a=attention_mask_a, head_mask=head_mask, img_feats=img_feats, use_b=use_b,
                                  input_ids_b=input_ids_b, position_ids_b=position_ids_b, token_type_ids_b=token_type_ids_b,
                                  attention_mask_b=attention_mask_b, max_tag_length=max_tag_length, encode_hn=False)
        logits = self.classifier(self.dropout(outputs[1]))
        out = (logits,) + outputs[2:]
        if labels is not None:
            out = (_cls_loss(self, logits, labels, soft_label),) + out
        return out


class BiImageBertForVQA(_FreezeMixin, BertPreTrainedModel):
    """vl:1801-1870 — answer classifier on the raw [CLS] state of the joint encoder."""

    def __init__(self, config):
        super().__init__(config)
        self.num_labels = config.num_labels
        self.loss_type = config.loss_type
        self.config = config
        if config.img_feature_dim <= 0:
            raise NotImplementedError("text-only BertModel is outside the accelerated path")
        self.bert = BiBertImgModel(config)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)
        self.cls = BertVQAHeads(config)
        self.apply(self.init_weights)

    def forward(self, input_ids_a, token_type_ids_a=None, attention_mask_a=None, labels=None, input_ids_b=None,
                token_type_ids_b=None, attention_mask_b=None, max_tag_length=20, position_ids_a=None,
                position_ids_b=None, head_mask=None, img_feats=None, soft_label=False, host_counts=None):
        """host_counts (optional, not a reference argument): synthetic.finetune_host_counts(batch, max_tag_length) — the step then
        reads nothing back from the device (and can be captured as a HIP graph, train.GraphedStep)."""
        outputs, _, _ = self.bert(input_ids_a=input_ids_a, position_ids_a=position_ids_a, token_type_ids_a=token_type_ids_a,
                                  attention_mask_a=attention_mask_a, head_mask=head_mask, img_feats=img_feats,
                                  input_ids_b=input_ids_b, position_ids_b=position_ids_b, token_type_ids_b=token_type_ids_b,
                                  attention_mask_b=attention_mask_b, max_tag_length=max_tag_length, encode_hn=False,
                                  host_counts=host_counts)
        logits = self.cls(self.dropout(outputs[0][:, 0]))
        out = (logits,) + outputs[2:]
        if labels is not None:
            out = (_cls_loss(self, logits, labels, soft_label),) + out
        return out
