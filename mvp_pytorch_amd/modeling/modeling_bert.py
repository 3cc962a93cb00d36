"""BERT building blocks with the reference's module / parameter tree
(transformers/pytorch_transformers/modeling_bert.py:158-533) so checkpoints round-trip.
The modules own f32 master parameters; their forward goes through the HIP kernels
(mvp_pytorch_amd.engine) on bf16 activations — there is no CPU forward.
"""
import json
import sys

import torch
import torch.nn.functional as F
from torch import nn

from .. import engine
from .modeling_utils import PretrainedConfig, PreTrainedModel


class BertConfig(PretrainedConfig):
    """Attribute bag of modeling_bert.py:189-225 (int vocab size or path to a JSON file)."""

    def __init__(self, vocab_size_or_config_json_file=30522, hidden_size=768, num_hidden_layers=12,
                 num_attention_heads=12, intermediate_size=3072, hidden_act="gelu",
                 hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1,
                 max_position_embeddings=512, type_vocab_size=2, initializer_range=0.02,
                 layer_norm_eps=1e-12, **kwargs):
        super().__init__(**kwargs)
        if isinstance(vocab_size_or_config_json_file, str):
            with open(vocab_size_or_config_json_file, "r", encoding="utf-8") as reader:
                self.__dict__.update(json.loads(reader.read()))
        elif isinstance(vocab_size_or_config_json_file, int):
            self.vocab_size = vocab_size_or_config_json_file
            self.hidden_size = hidden_size
            self.num_hidden_layers = num_hidden_layers
            self.num_attention_heads = num_attention_heads
            self.hidden_act = hidden_act
            self.intermediate_size = intermediate_size
            self.hidden_dropout_prob = hidden_dropout_prob
            self.attention_probs_dropout_prob = attention_probs_dropout_prob
            self.max_position_embeddings = max_position_embeddings
            self.type_vocab_size = type_vocab_size
            self.initializer_range = initializer_range
            self.layer_norm_eps = layer_norm_eps
        else:
            raise ValueError("First argument must be either a vocabulary size (int)"
                             "or the path to a pretrained model config file (str)")


def _require_gelu(config):
    if config.hidden_act != "gelu":
        raise NotImplementedError("the HIP encoder implements hidden_act='gelu' (erf) only, got %r" % (config.hidden_act,))


def bf16_rows(x):
    return x if x.dtype == torch.bfloat16 else x.to(torch.bfloat16)


class BertLayerNorm(nn.Module):
    """modeling_bert.py:233-246 (TF style, eps inside the sqrt)."""

    def __init__(self, hidden_size, eps=1e-12):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(hidden_size))
        self.bias = nn.Parameter(torch.zeros(hidden_size))
        self.variance_epsilon = eps

    def forward(self, x):
        return engine.LayerNormFn.apply(bf16_rows(x), self.weight, self.bias, self.variance_epsilon)


class BertEmbeddings(nn.Module):
    """modeling_bert.py:248-277.  forward() is driven by the backbone through
    engine.InputEmbedFn (gather + add + LayerNorm + dropout, optionally fused with the region
    embedding and the concat)."""

    def __init__(self, config):
        super().__init__()
        self.word_embeddings = nn.Embedding(config.vocab_size, config.hidden_size, padding_idx=0)
        self.position_embeddings = nn.Embedding(config.max_position_embeddings, config.hidden_size)
        self.token_type_embeddings = nn.Embedding(config.type_vocab_size, config.hidden_size)
        self.LayerNorm = BertLayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)

    def forward(self, input_ids, token_type_ids=None, position_ids=None):
        return embed_inputs(self, input_ids, token_type_ids, position_ids, None, None)


def embed_inputs(emb, input_ids, token_type_ids, position_ids, img_feats, owner, share=None):
    """Shared driver: `owner` is the backbone holding img_embedding / LayerNorm / dropout.  share: a
    dict common to the embedding calls of one forward pass (see InputEmbedFn.backward)."""
    L = input_ids.size(1)
    if position_ids is None:
        position_ids = engine.arange(L, input_ids.device).unsqueeze(0).expand_as(input_ids)
    if token_type_ids is None:
        token_type_ids = torch.zeros_like(input_ids)
    cache = emb.__dict__.setdefault("_img_cache", engine.WeightCache())
    meta = dict(eps=emb.LayerNorm.variance_epsilon, training=emb.training, p=emb.dropout.p, cache=cache,
                use_img_ln=False, img_eps=1e-12, share=share)
    img_w = img_b = ln_w = ln_b = None
    if img_feats is not None:
        img_w, img_b = owner.img_embedding.weight, owner.img_embedding.bias
        if owner.use_img_layernorm:
            ln_w, ln_b = owner.LayerNorm.weight, owner.LayerNorm.bias
            meta["use_img_ln"] = True
            meta["img_eps"] = owner.LayerNorm.variance_epsilon
        meta["p_img"] = owner.dropout.p
    return engine.InputEmbedFn.apply(input_ids, token_type_ids, position_ids, img_feats, meta,
                                     emb.word_embeddings.weight, emb.position_embeddings.weight,
                                     emb.token_type_embeddings.weight, emb.LayerNorm.weight,
                                     emb.LayerNorm.bias, img_w, img_b, ln_w, ln_b)


class BertSelfAttention(nn.Module):
    """Parameter container of modeling_bert.py:280-297 (query / key / value + dropout)."""

    def __init__(self, config):
        super().__init__()
        if config.hidden_size % config.num_attention_heads != 0:
            raise ValueError("The hidden size (%d) is not a multiple of the number of attention "
                             "heads (%d)" % (config.hidden_size, config.num_attention_heads))
        self.output_attentions = config.output_attentions
        self.num_attention_heads = config.num_attention_heads
        self.attention_head_size = int(config.hidden_size / config.num_attention_heads)
        self.all_head_size = self.num_attention_heads * self.attention_head_size
        self.query = nn.Linear(config.hidden_size, self.all_head_size)
        self.key = nn.Linear(config.hidden_size, self.all_head_size)
        self.value = nn.Linear(config.hidden_size, self.all_head_size)
        self.dropout = nn.Dropout(config.attention_probs_dropout_prob)


class BertSelfOutput(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.hidden_size, config.hidden_size)
        self.LayerNorm = BertLayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)


class BertAttention(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.self = BertSelfAttention(config)
        self.output = BertSelfOutput(config)


class BertIntermediate(nn.Module):
    def __init__(self, config):
        super().__init__()
        _require_gelu(config)
        self.dense = nn.Linear(config.hidden_size, config.intermediate_size)


class BertOutput(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.intermediate_size, config.hidden_size)
        self.LayerNorm = BertLayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)


class BertLayer(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.attention = BertAttention(config)
        self.intermediate = BertIntermediate(config)
        self.output = BertOutput(config)


class BertPooler(nn.Module):
    """modeling_bert.py:462-474 — tanh(dense(h[:, 0])): the [CLS] rows are tapped from the sequence buffer by row
    index (no strided copy) and go through the f32 HIP GEMM with a bias + tanh epilogue."""

    def __init__(self, config):
        super().__init__()
        self.dense = nn.Linear(config.hidden_size, config.hidden_size)
        self.activation = nn.Tanh()

    def forward_rows(self, cls_rows):
        """tanh(dense(rows)) on [CLS] rows already gathered ([n, H] bf16 / f32): f32 HIP GEMM with the bias + tanh
        epilogue (mvptr_sgemm_small) for f32 master weights on a HIP device; after model.half() (or on the CPU)
        the tail runs in torch in the parameters' dtype like the reference."""
        w = self.dense.weight
        if cls_rows.is_cuda and w.dtype == torch.float32:
            return engine.SmallLinearFn.apply(cls_rows, w, self.dense.bias, "tanh", False)
        return self.activation(self.dense(cls_rows.to(w.dtype)))

    def forward(self, hidden_states):
        if hidden_states.is_cuda and hidden_states.dtype == torch.bfloat16 and hidden_states.is_contiguous() and hidden_states.dim() == 3:
            B, L, H = hidden_states.shape
            rows = engine.arange(B, hidden_states.device, torch.int32, step=L)
            return self.forward_rows(engine.tap_rows(hidden_states.view(B * L, H), rows))
        return self.forward_rows(hidden_states[:, 0])


class BertPredictionHeadTransform(nn.Module):
    """modeling_bert.py:477-491 — dense + gelu + LayerNorm on the selected rows (HIP GEMM with
    the gelu epilogue, HIP LayerNorm)."""

    def __init__(self, config):
        super().__init__()
        _require_gelu(config)
        self.dense = nn.Linear(config.hidden_size, config.hidden_size)
        self.LayerNorm = BertLayerNorm(config.hidden_size, eps=config.layer_norm_eps)
        self._cache = engine.WeightCache()
        self._act = "gelu16" if str(getattr(config, "gelu_stash", "u8")).lower() == "bf16" else "gelu"   # format of the gelu' stash

    def forward(self, hidden_states):
        h = engine.LinearFn.apply(bf16_rows(hidden_states), self.dense.weight, self.dense.bias, self._act, self._cache)
        return self.LayerNorm(h)


class BertLMPredictionHead(nn.Module):
    """modeling_bert.py:494-516 (only_vocab = MVPTR edit: decoder over the first
    `only_word_size` word pieces)."""

    def __init__(self, config, only_vocab=False):
        super().__init__()
        self.transform = BertPredictionHeadTransform(config)
        n_out = config.only_word_size if only_vocab else config.vocab_size
        self.decoder = nn.Linear(config.hidden_size, n_out, bias=False)
        self.bias = nn.Parameter(torch.zeros(n_out))
        self._cache = engine.WeightCache()

    def loss_and_scores(self, hidden_states, labels, want_scores=True):
        """Fused decoder + CrossEntropyLoss(ignore_index=-1): (mean loss, f32 logits [M,V]).
        want_scores=False: the loss only — the logits stay inside the GEMM epilogues
        (mvptr_decoder_ce_fwd / _bwd) and the second result is an empty [0, V] tensor."""
        if hidden_states.reshape(-1, hidden_states.shape[-1]).shape[0] == 0:
            # no scored row in this batch / data-parallel shard (the dataset masks 15 % of the tokens
            # with no guarantee of one per shard, oscar_tsv4.py:782-893): a zero that is still connected
            # to every parameter of the head, so each rank produces the same set of gradients (the
            # reference's CrossEntropyLoss over zero rows would give NaN; a skipped shard must not
            # poison the averaged gradient)
            zero = hidden_states.float().sum() * 0.0
            for p in (self.transform.dense.weight, self.transform.dense.bias, self.transform.LayerNorm.weight,
                      self.transform.LayerNorm.bias, self.decoder.weight, self.bias):
                zero = zero + p.float().sum() * 0.0
            return zero, torch.zeros((0, self.decoder.weight.shape[0]), dtype=torch.float32, device=hidden_states.device)
        h = self.transform(hidden_states)
        return engine.DecoderCEFn.apply(h.reshape(-1, h.shape[-1]), self.decoder.weight, self.bias,
                                        labels.reshape(-1), self._cache, want_scores)

    def forward(self, hidden_states):
        h = self.transform(hidden_states)
        dummy = torch.full((h.reshape(-1, h.shape[-1]).shape[0],), -1, dtype=torch.long, device=h.device)
        _, scores = engine.DecoderCEFn.apply(h.reshape(-1, h.shape[-1]), self.decoder.weight, self.bias, dummy, self._cache)
        return scores.reshape(h.shape[:-1] + (scores.shape[-1],))


class HeadLinear(nn.Linear):
    """nn.Linear for the B-row f32 heads (fine-tune classifiers vl:1628-1640,1745-1756, the 3129-way VQA decoder
    modeling_bert.py:518-533, the QA head vl:1207): on a HIP device the product runs on the f32 MFMA kernel
    (engine.SmallLinearFn -> mvptr_sgemm_small: exact f32 operands, bias fused, gradients into the arena) instead of a
    BLAS call that serves these few-row shapes with a single tile.  Same parameters / state_dict as nn.Linear."""

    def forward(self, x, bias=None):
        b = self.bias if bias is None else bias
        if x.is_cuda and self.weight.dtype == torch.float32:
            # the product path: f32 parameters on a HIP device.  Anything it does not take raises instead of silently running a
            # torch op; inputs of more than two dimensions are rows of a 2-D problem
            if x.dtype not in (torch.float32, torch.bfloat16):
                raise TypeError("HeadLinear: the HIP path takes f32 or bf16 rows, got %s (f32 parameters on a HIP device)" % x.dtype)
            if x.dim() != 2:
                return engine.SmallLinearFn.apply(x.reshape(-1, x.shape[-1]), self.weight, b, None, False).reshape(x.shape[:-1] + (self.weight.shape[0],))
            return engine.SmallLinearFn.apply(x, self.weight, b, None, False)
        # documented torch paths: model.half() inference (fp16 parameters) and CPU tensors (checkpoint
        # surgery, host-logic tests) — never taken by f32 parameters on a HIP device
        return F.linear(x.to(self.weight.dtype), self.weight, b)


class BertQAPredictionHead(nn.Module):
    """modeling_bert.py:518-533 — transform + Linear(hidden -> num_labels, no bias) + bias."""

    def __init__(self, config, only_vocab=False):
        super().__init__()
        self.transform = BertPredictionHeadTransform(config)
        self.decoder = HeadLinear(config.hidden_size, config.num_labels, bias=False)
        self.bias = nn.Parameter(torch.zeros(config.num_labels))

    def forward(self, hidden_states):
        h = self.transform(hidden_states)
        return self.decoder(h, bias=self.bias)


class BertPreTrainedModel(PreTrainedModel):
    config_class = BertConfig
    base_model_prefix = "bert"
