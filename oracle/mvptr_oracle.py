"""CPU oracle for the MVPTR cross-modal encoder path — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A plain-PyTorch fp32, *functional* restatement (no nn.Module, parameters come from a
state_dict-shaped mapping) of the reference algorithm.  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import this file; mvp_pytorch_amd never does.

Parity pinning: the reference's own tests hold no numeric vectors for this path, so
the oracle is pinned against outputs of the reference itself, imported in the build container by
tools/gen_golden.py (fixtures under tests/golden/, checked by tests/test_oracle_golden.py).

Each function cites the reference lines it restates (paths relative to the reference tree;
`mb` = transformers/pytorch_transformers/modeling_bert.py,
`vl` = oscar/modeling/modeling_vlbert.py).

Randomness: the reference draws random numbers inside forward (torch.randperm vl:556,
torch.randint vl:1548, random.choice vl:1573).  The oracle takes them from a `Draws` object so
that recorded reference draws (or the product's draws) can be injected.  Dropout is identity
(p = 0 / eval), as in the fixtures.
"""
import math

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------------------------
class Draws:
    """Source of the three in-forward random draws; replays recorded lists when given."""

    def __init__(self, randperm=None, randint3=None, choice=None, seed=0, multinomial=None):
        self._multinomial = list(multinomial) if multinomial is not None else None
        self._randperm = list(randperm) if randperm is not None else None
        self._randint3 = list(randint3) if randint3 is not None else None
        self._choice = list(choice) if choice is not None else None
        self._gen = torch.Generator().manual_seed(seed)

    def randperm(self, n):
        if self._randperm is not None:
            v = torch.as_tensor(self._randperm.pop(0), dtype=torch.long)
            assert v.numel() == n
            return v
        return torch.randperm(n, generator=self._gen)

    def randint3(self, n):
        if self._randint3 is not None:
            v = torch.as_tensor(self._randint3.pop(0), dtype=torch.long).reshape(-1)
            assert v.numel() == n
            return v
        return torch.randint(0, 3, (n,), generator=self._gen)

    def multinomial(self, probs):
        """one draw per row of probs (vl:538,540)."""
        if self._multinomial is not None:
            v = torch.as_tensor(self._multinomial.pop(0), dtype=torch.long).reshape(-1)
            assert v.numel() == probs.shape[0]
            return v
        return torch.multinomial(probs, num_samples=1, generator=self._gen).squeeze(1)

    def choice(self, options):
        if self._choice is not None:
            v = int(self._choice.pop(0))
            assert v in options
            return v
        return options[int(torch.randint(0, len(options), (1,), generator=self._gen))]


# -------------------------------------------------------------------------------- building blocks
def gelu(x):
    """mb:142-148 — erf GELU."""
    return x * 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0)))


def layer_norm(x, weight, bias, eps):
    """mb:242-246 — TF-style LayerNorm, epsilon inside the square root, biased variance."""
    u = x.mean(-1, keepdim=True)
    s = (x - u).pow(2).mean(-1, keepdim=True)
    return weight * ((x - u) / torch.sqrt(s + eps)) + bias


def linear(sd, prefix, x):
    w = sd[prefix + ".weight"]
    b = sd.get(prefix + ".bias")
    y = x @ w.t()
    return y if b is None else y + b


def extended_mask(attention_mask, dtype=torch.float32):
    """vl:278-292 / vl:430-444 — [B,L] 0/1 mask -> additive [B,1,1,L] with -10000 on masked."""
    if attention_mask.dim() != 2:
        raise NotImplementedError
    m = attention_mask[:, None, None, :].to(dtype)
    return (1.0 - m) * -10000.0


def embeddings(sd, prefix, input_ids, token_type_ids=None, position_ids=None, eps=1e-12):
    """mb:262-277 — word + position + token-type embeddings, LayerNorm (dropout = identity)."""
    L = input_ids.size(1)
    if position_ids is None:
        position_ids = torch.arange(L, dtype=torch.long).unsqueeze(0).expand_as(input_ids)
    if token_type_ids is None:
        token_type_ids = torch.zeros_like(input_ids)
    e = (sd[prefix + ".word_embeddings.weight"][input_ids]
         + sd[prefix + ".position_embeddings.weight"][position_ids]
         + sd[prefix + ".token_type_embeddings.weight"][token_type_ids])
    return layer_norm(e, sd[prefix + ".LayerNorm.weight"], sd[prefix + ".LayerNorm.bias"], eps)


def _apply_drop(t, drop):
    """nn.Dropout with a given keep mask: drop = (keep 0/1 tensor broadcastable to t, 1 / (1 - p)) or None (identity)."""
    if drop is None:
        return t
    keep, scale = drop
    return t * keep.to(t.dtype) * scale


def self_attention(sd, prefix, x, ext_mask, heads, drop=None):
    """vl:63-103 (+ transpose_for_scores mb:299-303).  drop: keep mask [B, heads, L, L] + scale of the dropout on the
    attention probabilities (vl:90), None = identity (the golden fixtures run with dropout 0)."""
    B, L, H = x.shape
    d = H // heads

    def split(t):
        return t.view(B, L, heads, d).permute(0, 2, 1, 3)

    q = split(linear(sd, prefix + ".query", x))
    k = split(linear(sd, prefix + ".key", x))
    v = split(linear(sd, prefix + ".value", x))
    scores = q @ k.transpose(-1, -2) / math.sqrt(d) + ext_mask
    probs = _apply_drop(torch.softmax(scores, dim=-1), drop)
    ctx = (probs @ v).permute(0, 2, 1, 3).contiguous().view(B, L, H)
    return ctx


def encoder_layer(sd, prefix, x, ext_mask, heads, eps, drops=None):
    """vl:191-199 = attention vl:115-120 (+ BertSelfOutput mb:348-352), BertIntermediate
    mb:394-397, BertOutput mb:407-411.  drops: optional {"attn": ..., "attn_out": ..., "ffn_out": ...} keep masks (+ scale)
    of the layer's three dropout sites (vl:90, mb:350, mb:409) for tests that replay a kernel's masks; None = identity."""
    drops = drops or {}
    ctx = self_attention(sd, prefix + ".attention.self", x, ext_mask, heads, drops.get("attn"))
    a = _apply_drop(linear(sd, prefix + ".attention.output.dense", ctx), drops.get("attn_out"))
    a = layer_norm(a + x, sd[prefix + ".attention.output.LayerNorm.weight"],
                   sd[prefix + ".attention.output.LayerNorm.bias"], eps)
    i = gelu(linear(sd, prefix + ".intermediate.dense", a))
    o = _apply_drop(linear(sd, prefix + ".output.dense", i), drops.get("ffn_out"))
    return layer_norm(o + a, sd[prefix + ".output.LayerNorm.weight"],
                      sd[prefix + ".output.LayerNorm.bias"], eps)


def encoder(sd, prefix, n_layers, x, ext_mask, heads, eps, return_at_layer=None, drops=None):
    """vl:134-178 (single mask, head_mask None, no history states).  drops: optional list of per-layer dropout masks
    (encoder_layer)."""
    mid = None
    for i in range(n_layers):
        x = encoder_layer(sd, "%s.layer.%d" % (prefix, i), x, ext_mask, heads, eps, None if drops is None else drops[i])
        if return_at_layer is not None and i == return_at_layer:
            mid = x
    return (x, mid) if return_at_layer is not None else x


def pooler(sd, prefix, x):
    """mb:468-474."""
    return torch.tanh(linear(sd, prefix + ".dense", x[:, 0]))


def img_embedding(sd, prefix, cfg, img_feats):
    """vl:328-333 / vl:498-503 — Linear(img_feature_dim -> hidden) (+ LayerNorm when
    use_img_layernorm; dropout = identity)."""
    e = linear(sd, prefix + ".img_embedding", img_feats)
    if cfg.get("use_img_layernorm"):
        e = layer_norm(e, sd[prefix + ".LayerNorm.weight"], sd[prefix + ".LayerNorm.bias"],
                       cfg["img_layer_norm_eps"])
    return e


def lm_head(sd, prefix, x, eps):
    """mb:477-516 — transform (dense, gelu, LayerNorm) + decoder (no bias) + bias."""
    h = layer_norm(gelu(linear(sd, prefix + ".transform.dense", x)),
                   sd[prefix + ".transform.LayerNorm.weight"], sd[prefix + ".transform.LayerNorm.bias"], eps)
    return h @ sd[prefix + ".decoder.weight"].t() + sd[prefix + ".bias"]


def ce(logits, labels):
    """CrossEntropyLoss(ignore_index=-1), mean over non-ignored rows (vl:1228)."""
    return F.cross_entropy(logits, labels, ignore_index=-1)


# --------------------------------------------------------------------------------------- backbones
def bert_img_model(sd, cfg, input_ids, token_type_ids=None, attention_mask=None, position_ids=None,
                   img_feats=None, prefix="bert"):
    """vl:251-348 BertImgModel.forward -> (sequence_output, pooled_output)."""
    if attention_mask is None:
        attention_mask = torch.ones_like(input_ids)
    if token_type_ids is None:
        token_type_ids = torch.zeros_like(input_ids)
    ext = extended_mask(attention_mask)
    x = embeddings(sd, prefix + ".embeddings", input_ids, token_type_ids, position_ids, cfg["layer_norm_eps"])
    if img_feats is not None:
        x = torch.cat((x, img_embedding(sd, prefix, cfg, img_feats)), 1)
    seq = encoder(sd, prefix + ".encoder", cfg["num_hidden_layers"], x, ext, cfg["num_attention_heads"], cfg["layer_norm_eps"])
    return seq, pooler(sd, prefix + ".pooler", seq)


def bi_uni_encoders(sd, cfg, input_ids_a, token_type_ids_a, attention_mask_a, input_ids_b,
                    token_type_ids_b, attention_mask_b, img_feats, position_ids_a=None,
                    position_ids_b=None, prefix="bert"):
    """vl:413-513 — masks, shared embeddings for text ids and tag ids, img embedding, cat,
    txt_encoder and vis_encoder."""
    if attention_mask_a is None:
        attention_mask_a = torch.ones_like(input_ids_a)
    if attention_mask_b is None:
        attention_mask_b = torch.ones_like(input_ids_b)
    if token_type_ids_a is None:
        token_type_ids_a = torch.zeros_like(input_ids_a)
    if token_type_ids_b is None:
        token_type_ids_b = torch.zeros_like(input_ids_b)
    ext_a = extended_mask(attention_mask_a)
    ext_b = extended_mask(attention_mask_b)
    eps, heads, nl = cfg["layer_norm_eps"], cfg["num_attention_heads"], cfg["num_hidden_layers"] // 2
    ea = embeddings(sd, prefix + ".embeddings", input_ids_a, token_type_ids_a, position_ids_a, eps)
    eb = embeddings(sd, prefix + ".embeddings", input_ids_b, token_type_ids_b, position_ids_b, eps)
    if img_feats is not None:
        eb = torch.cat((eb, img_embedding(sd, prefix, cfg, img_feats)), 1)
    txt = encoder(sd, prefix + ".txt_encoder", nl, ea, ext_a, heads, eps)
    vis = encoder(sd, prefix + ".vis_encoder", nl, eb, ext_b, heads, eps)
    return txt, vis, ext_a, ext_b


def clip_globals(sd, txt, vis, prefix="bert"):
    """vl:525-526."""
    gt = F.normalize(txt[:, 0, :] @ sd[prefix + ".txt_proj"], p=2, dim=-1)
    gi = F.normalize(vis[:, 0, :] @ sd[prefix + ".vis_proj"], p=2, dim=-1)
    return gt, gi


def bi_bert_img_model(sd, cfg, input_ids_a, token_type_ids_a=None, attention_mask_a=None,
                      max_tag_length=None, use_b=False, input_ids_b=None, token_type_ids_b=None,
                      attention_mask_b=None, img_feats=None, encode_hn=False, draws=None,
                      position_ids_a=None, position_ids_b=None, prefix="bert", hn_mod="hard", logit=None):
    """vl:410-609 BiBertImgModel.forward (phrase_layer None) ->
    ((seq, pooled, hard_seq, hard_pooled), (txt, vis, sim_mat), (hard_txt_idx_full, hard_img_idx_full))."""
    txt, vis, ext_a, ext_b = bi_uni_encoders(sd, cfg, input_ids_a, token_type_ids_a, attention_mask_a,
                                             input_ids_b, token_type_ids_b, attention_mask_b, img_feats,
                                             position_ids_a, position_ids_b, prefix)
    eps, heads, nl = cfg["layer_norm_eps"], cfg["num_attention_heads"], cfg["num_hidden_layers"] // 2
    cut = 1 if use_b else max_tag_length
    only_vis = vis[:, cut:, :]
    only_vis_mask = ext_b[:, :, :, cut:]
    gt, gi = clip_globals(sd, txt, vis, prefix)
    sim = gt @ gi.t()
    hard_seq_out = hard_pooled = hard_txt_full = hard_img_full = None
    if encode_hn:
        n = sim.shape[0]
        if hn_mod == "hard":      # vl:531-534
            masked = sim - 2 * torch.eye(n, dtype=sim.dtype)
            hard_img = torch.max(masked, dim=1)[1]
            hard_txt = torch.max(masked, dim=0)[1]
        elif hn_mod == "sample":  # vl:535-540
            masked = (logit * sim) - 10000 * torch.eye(n, dtype=sim.dtype)
            hard_img = (draws or Draws()).multinomial(F.softmax(masked, dim=1))
            hard_txt = (draws or Draws()).multinomial(F.softmax(masked.t(), dim=1))
        else:
            raise NotImplementedError
        hard_img_seq = torch.cat([txt, only_vis.index_select(0, hard_img)], dim=1)
        hard_img_mask = torch.cat([ext_a, only_vis_mask.index_select(0, hard_img)], dim=-1)
        hard_txt_seq = torch.cat([txt.index_select(0, hard_txt), only_vis], dim=1)
        hard_txt_mask = torch.cat([ext_a.index_select(0, hard_txt), only_vis_mask], dim=-1)
        dice = (draws or Draws()).randperm(n)
        first, second = dice[: n // 2], dice[n // 2:]
        hard_seqs = torch.cat([hard_img_seq.index_select(0, first), hard_txt_seq.index_select(0, second)], 0)
        hard_mask = torch.cat([hard_img_mask.index_select(0, first), hard_txt_mask.index_select(0, second)], 0)
        ar = torch.arange(n)
        hard_txt_full = torch.cat([ar.index_select(0, first), hard_txt.index_select(0, second)], 0)
        hard_img_full = torch.cat([hard_img.index_select(0, first), ar.index_select(0, second)], 0)
        hard_seq_out = encoder(sd, prefix + ".mul_encoder", nl, hard_seqs, hard_mask, heads, eps)
        hard_pooled = pooler(sd, prefix + ".pooler", hard_seq_out)
    joint = torch.cat([txt, only_vis], dim=1)
    joint_mask = torch.cat([ext_a, only_vis_mask], dim=-1)
    seq = encoder(sd, prefix + ".mul_encoder", nl, joint, joint_mask, heads, eps)
    pooled = pooler(sd, prefix + ".pooler", seq)
    return (seq, pooled, hard_seq_out, hard_pooled), (txt, vis, sim), (hard_txt_full, hard_img_full)


def bi_forward_single(sd, cfg, input_ids_a, token_type_ids_a=None, attention_mask_a=None,
                      input_ids_b=None, token_type_ids_b=None, attention_mask_b=None, img_feats=None,
                      prefix="bert"):
    """vl:611-723 forward_single -> (global_txt, global_img)."""
    txt, vis, _, _ = bi_uni_encoders(sd, cfg, input_ids_a, token_type_ids_a, attention_mask_a,
                                     input_ids_b, token_type_ids_b, attention_mask_b, img_feats,
                                     prefix=prefix)
    return clip_globals(sd, txt, vis, prefix)


# ------------------------------------------------------------------------------------------- WRA
def mask_slice_and_stack(features, valid_index):
    """vl:1502-1508."""
    out = []
    for i in range(features.shape[0]):
        out.append(features[i, int(valid_index[i, 0]):int(valid_index[i, 1])])
    return torch.cat(out, dim=0)


def t2i_sim(sim, draws):
    """vl:1543-1550 — per phrase a random one of the top-3 region similarities, then mean."""
    if sim.shape[0] == 0:
        return torch.zeros((), dtype=sim.dtype)
    top = sim.topk(3, dim=1)[0]
    pick = draws.randint3(top.shape[0])
    return top[torch.arange(top.shape[0]), pick].mean()


def get_pos_neg_sims(sims, text_index, img_index, draws):
    """vl:1553-1596."""
    tn = (text_index[:, 1] - text_index[:, 0]).tolist()
    im = (img_index[:, 1] - img_index[:, 0]).tolist()
    tb = [0]
    ib = [0]
    for v in tn:
        tb.append(tb[-1] + v)
    for v in im:
        ib.append(ib[-1] + v)
    n = text_index.shape[0]
    pos, neg = [], []
    for t in range(n):
        pos.append(t2i_sim(sims[tb[t]:tb[t + 1], ib[t]:ib[t + 1]], draws))
        options = list(range(0, t)) + list(range(t + 1, n))
        j = draws.choice(options)
        neg.append(t2i_sim(sims[tb[t]:tb[t + 1], ib[j]:ib[j + 1]], draws))
    return torch.stack(pos), torch.stack(neg)


def get_pos_sims(sequence_output, text_index, img_index, draws):
    """vl:1510-1527 — per sample: its phrases against its own regions."""
    out = []
    for i in range(text_index.shape[0]):
        t = F.normalize(sequence_output[i, int(text_index[i, 0]):int(text_index[i, 1])], p=2, dim=-1)
        v = F.normalize(sequence_output[i, int(img_index[i, 0]):int(img_index[i, 1])], p=2, dim=-1)
        out.append(t2i_sim(t @ v.t(), draws))
    return torch.stack(out)


def wra_loss_sample(sequence_output, phrase_index, img_index, draws):
    """vl:1285-1300 (phrase_mod='sample')."""
    vp = F.normalize(mask_slice_and_stack(sequence_output, phrase_index), p=2, dim=-1)
    vi = F.normalize(mask_slice_and_stack(sequence_output, img_index), p=2, dim=-1)
    full = vp @ vi.t()
    pos, neg = get_pos_neg_sims(full, phrase_index, img_index, draws)
    loss = torch.clamp(neg + 0.2 - pos, min=0)
    valid = (phrase_index[:, 1] - phrase_index[:, 0]) > 0
    return torch.mean(torch.masked_select(loss, valid))


# ------------------------------------------------------------------------------------ task models
def bert_img_for_pretraining(sd, cfg, input_ids, token_type_ids=None, attention_mask=None,
                             masked_lm_labels=None, next_sentence_label=None, img_feats=None):
    """vl:1102-1130 BertImgForPreTraining.forward ->
    (total, prediction_scores, seq_relationship_score, masked_lm_loss)."""
    seq, pooled = bert_img_model(sd, cfg, input_ids, token_type_ids, attention_mask, None, img_feats)
    T = cfg.get("max_text_seq_length")
    text = seq[:, :T, :] if T is not None else seq
    scores = lm_head(sd, "cls.predictions", text, cfg["layer_norm_eps"])
    rel = linear(sd, "cls.seq_relationship", pooled)
    if masked_lm_labels is None or next_sentence_label is None:
        return scores, rel
    labels = masked_lm_labels[:, :T] if T is not None else masked_lm_labels
    mlm = ce(scores.reshape(-1, scores.shape[-1]), labels.reshape(-1))
    nsp = ce(rel.view(-1, rel.shape[-1]), next_sentence_label.view(-1))
    return mlm + nsp, scores, rel, mlm


def bi_bert_img_for_pretraining(sd, cfg, input_ids_a, token_type_ids_a=None, attention_mask_a=None,
                                masked_lm_labels_a=None, input_ids_b=None, token_type_ids_b=None,
                                attention_mask_b=None, masked_lm_labels_b=None, max_tag_length=20,
                                img_feats=None, img_index=None, phrase_index=None, draws=None,
                                return_aux=False, qa_ans=None, phrase_mod="sample"):
    """vl:1218-1311 BiBertImgForPreTraining.forward ->
    (total, vis_mlm, retrieval, mlm, itm[, qa][, wra])."""
    draws = draws or Draws()
    outs, single, hard_idx = bi_bert_img_model(
        sd, cfg, input_ids_a, token_type_ids_a, attention_mask_a, max_tag_length, False, input_ids_b,
        token_type_ids_b, attention_mask_b, img_feats, True, draws)
    txt, vis, sim = single
    H = cfg["hidden_size"]
    eps = cfg["layer_norm_eps"]
    vmask = masked_lm_labels_b > -1
    vis_rows = torch.masked_select(vis, vmask.unsqueeze(-1)).reshape(-1, H)
    vis_mlm = ce(lm_head(sd, "half_mlm", vis_rows, eps), torch.masked_select(masked_lm_labels_b, vmask))
    logit = sim * sd["logit_scale"].exp()
    lab = torch.arange(sim.shape[0])
    retrieval = (ce(logit, lab) + ce(logit.t(), lab)) / 2
    seq, pooled, hard_seq, hard_pooled = outs
    La = input_ids_a.shape[1]
    tmask = masked_lm_labels_a > -1
    rows = torch.masked_select(seq[:, :La, :], tmask.unsqueeze(-1)).reshape(-1, H)
    scores = lm_head(sd, "cls.predictions", rows, eps)
    rel = linear(sd, "cls.seq_relationship", torch.cat([pooled, hard_pooled], 0))
    mlm = ce(scores, torch.masked_select(masked_lm_labels_a, tmask))
    n = pooled.shape[0]
    itm_labels = torch.cat([torch.zeros(n, dtype=torch.long), torch.ones(n, dtype=torch.long)])
    itm = ce(rel, itm_labels)
    total = vis_mlm + retrieval + mlm + itm
    out = (vis_mlm, retrieval, mlm, itm)
    if qa_ans is not None:        # vl:1264-1268
        qa = ce(linear(sd, "qa_head", pooled), qa_ans)
        total = total + qa
        out = out + (qa,)
    if phrase_index is not None:
        if phrase_mod == "sample":
            wra = wra_loss_sample(seq, phrase_index, img_index, draws)
        elif phrase_mod == "hard":   # vl:1271-1283
            hard_phrase = phrase_index.index_select(0, hard_idx[0])
            hard_object = img_index.index_select(0, hard_idx[1])
            pos = get_pos_sims(seq, phrase_index, img_index, draws)
            neg = get_pos_sims(hard_seq, hard_phrase, hard_object, draws)
            loss = torch.clamp(neg + 0.2 - pos, min=0)
            valid = ((phrase_index[:, 1] - phrase_index[:, 0]) > 0) & ((hard_phrase[:, 1] - hard_phrase[:, 0]) > 0)
            wra = torch.mean(torch.masked_select(loss, valid))
        else:
            raise NotImplementedError
        total = total + wra
        out = out + (wra,)
    res = (total,) + out
    if return_aux:
        return res, dict(sim_mat=sim, hard_txt_index=hard_idx[0], hard_img_index=hard_idx[1],
                         itm_labels=itm_labels, seq_relationship_score=rel, prediction_scores=scores,
                         sequence_output=seq, pooled_output=pooled, txt=txt, vis=vis)
    return res


def _classifier(sd, cfg, x):
    """vl:1615-1629 — 'linear' or 2-layer 'mlp' classifier."""
    if cfg.get("classifier", "linear") == "mlp":
        return linear(sd, "classifier.2", torch.relu(linear(sd, "classifier.0", x)))
    return linear(sd, "classifier", x)


def bi_retrieval(sd, cfg, mode, input_ids_a, token_type_ids_a=None, attention_mask_a=None,
                 input_ids_b=None, token_type_ids_b=None, attention_mask_b=None, max_tag_length=20,
                 img_feats=None, draws=None):
    """vl:1640-1712 BiImageBertForRetrieval (forward_mod = 'train' | 'coarse' | 'fine')."""
    if mode == "coarse":
        return bi_forward_single(sd, cfg, input_ids_a, token_type_ids_a, attention_mask_a, input_ids_b,
                                 token_type_ids_b, attention_mask_b, img_feats)
    outs, single, _ = bi_bert_img_model(sd, cfg, input_ids_a, token_type_ids_a, attention_mask_a,
                                        max_tag_length, False, input_ids_b, token_type_ids_b,
                                        attention_mask_b, img_feats, mode == "train", draws)
    seq, pooled, hard_seq, hard_pooled = outs
    if mode == "fine":
        return _classifier(sd, cfg, pooled)
    sim = single[2]
    logit = sim * sd["logit_scale"].exp()
    lab = torch.arange(sim.shape[0])
    retrieval = (ce(logit, lab) + ce(logit.t(), lab)) / 2
    rel = _classifier(sd, cfg, torch.cat([pooled, hard_pooled], 0))
    n = pooled.shape[0]
    labels = torch.cat([torch.ones(n, dtype=torch.long), torch.zeros(n, dtype=torch.long)])
    itm = ce(rel.view(-1, 2), labels)
    return retrieval + itm, rel, retrieval, itm, labels


def cls_loss(cfg, logits, labels, soft_label=False):
    """vl:1777-1797 / vl:1849-1869 — the loss switch shared by the VE / VQA wrappers."""
    n_labels = logits.shape[-1]
    if n_labels == 1:
        return F.mse_loss(logits.view(-1), labels.to(torch.float).view(-1))
    if soft_label:    # vl:27-40 soft_cross_entropy
        logp = F.log_softmax(logits, dim=1)
        t = torch.stack([1 - labels.float(), labels.float()], dim=1)
        return torch.mean(-torch.sum(t.view(t.shape[0], -1) * logp, dim=1))
    lt = cfg.get("loss_type", "ce")
    if lt == "kl":
        return F.kl_div(F.log_softmax(logits.contiguous().view(-1, 3129), dim=-1), labels.contiguous(), reduction="batchmean")
    if lt == "bce":   # vl:878-883 instance_bce_with_logits
        return F.binary_cross_entropy_with_logits(logits, labels, reduction="mean") * labels.size(1)
    return F.cross_entropy(logits.view(-1, n_labels), labels.view(-1))


def qa_head(sd, prefix, x, eps):
    """mb:518-533 BertQAPredictionHead."""
    return lm_head(sd, prefix, x, eps)


def bi_vqa(sd, cfg, input_ids_a, token_type_ids_a=None, attention_mask_a=None, labels=None,
           input_ids_b=None, token_type_ids_b=None, attention_mask_b=None, max_tag_length=20,
           img_feats=None, soft_label=False):
    """vl:1834-1870 BiImageBertForVQA.forward -> (loss, logits) or (logits,)."""
    outs, _, _ = bi_bert_img_model(sd, cfg, input_ids_a, token_type_ids_a, attention_mask_a,
                                   max_tag_length, False, input_ids_b, token_type_ids_b,
                                   attention_mask_b, img_feats, False)
    seq = outs[0]
    logits = qa_head(sd, "cls.predictions", seq[:, 0], cfg["layer_norm_eps"])
    if labels is None:
        return (logits,)
    return cls_loss(cfg, logits, labels, soft_label), logits


def bi_seq_cls(sd, cfg, input_ids_a, token_type_ids_a=None, attention_mask_a=None, labels=None,
               input_ids_b=None, token_type_ids_b=None, attention_mask_b=None, max_tag_length=20,
               use_b=False, img_feats=None, soft_label=False):
    """vl:1762-1798 BiImageBertForSequenceClassification.forward."""
    outs, _, _ = bi_bert_img_model(sd, cfg, input_ids_a, token_type_ids_a, attention_mask_a,
                                   max_tag_length, use_b, input_ids_b, token_type_ids_b,
                                   attention_mask_b, img_feats, False)
    logits = _classifier(sd, cfg, outs[1])
    if labels is None:
        return (logits,)
    return cls_loss(cfg, logits, labels, soft_label), logits


# ---------------------------------------------------------------------------------- optimisation
def adamw_step(params, grads, state, lr, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.0,
               correct_bias=True):
    """transformers/pytorch_transformers/optimization.py:131-187 — one AdamW step, in place."""
    b1, b2 = betas
    for name, p in params.items():
        g = grads.get(name)
        if g is None:
            continue
        st = state.setdefault(name, dict(step=0, m=torch.zeros_like(p), v=torch.zeros_like(p)))
        st["step"] += 1
        st["m"].mul_(b1).add_(g, alpha=1.0 - b1)
        st["v"].mul_(b2).addcmul_(g, g, value=1.0 - b2)
        denom = st["v"].sqrt().add_(eps)
        step_size = lr
        if correct_bias:
            step_size = step_size * math.sqrt(1.0 - b2 ** st["step"]) / (1.0 - b1 ** st["step"])
        p.addcdiv_(st["m"], denom, value=-step_size)
        wd = weight_decay(name) if callable(weight_decay) else weight_decay
        if wd > 0.0:
            p.add_(p, alpha=-lr * wd)


def clip_grad_norm(grads, max_norm):
    """torch.nn.utils.clip_grad_norm_(parameters, max_norm) as the reference's step applies it
    (oscar/run_pretrain_ml.py:639-640; norm_type 2): total = ||(||g_1||, ||g_2||, ...)||,
    coef = min(1, max_norm / (total + 1e-6)), every gradient scaled in place -> total norm."""
    gs = [g for g in grads.values() if g is not None]
    total = torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(g.float()) for g in gs]))
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    for g in gs:
        g.mul_(coef)
    return total


def warmup_linear(step, warmup_steps, t_total):
    """optimization.py:58-61 WarmupLinearSchedule.lr_lambda."""
    if step < warmup_steps:
        return float(step) / float(max(1, warmup_steps))
    return max(0.0, float(t_total - step) / float(max(1.0, t_total - warmup_steps)))


# ---------------------------------------------------------------------------------------------
# input pipeline
def decode_img_feature(b64_text, num_boxes, img_feature_dim, max_img_seq_length, dtype=torch.float32):
    """Region features of one TSV row -> [max_img_seq_length, img_feature_dim].
    oscar/oscar_datasets_ml/oscar_tsv4.py:716-724 (get_img_feature: base64 -> float32 [num_boxes, D]
    -> tensor of args.dtype) followed by __getitem__ :332-352 (keep the first max_img_seq_length rows,
    zero-pad the missing ones)."""
    import base64
    import numpy as np
    feat = np.frombuffer(base64.b64decode(b64_text), dtype=np.float32).reshape((num_boxes, img_feature_dim))
    feat = torch.tensor(np.copy(feat), dtype=dtype)
    if feat.shape[0] >= max_img_seq_length:
        feat = feat[0:max_img_seq_length, ]
    if feat.shape[0] < max_img_seq_length:
        pad = torch.zeros((max_img_seq_length - feat.shape[0], feat.shape[1]), dtype=dtype)
        feat = torch.cat((feat, pad), 0)
    return feat


# ---------------------------------------------------------------------------------------------
# retrieval evaluation, coarse stage
def compute_ranks_coarse(similarities, num_captions_per_img_train, num_captions_per_img_val, num_images_per_cap_val):
    """oscar/run_retrieval.py:481-522 on a numpy [n_img, n_cap] matrix.  Returns (i2t_ranks, t2i_ranks,
    i2t_index, t2i_index) with the candidate lists as plain indices: i2t_index[i] = caption indices
    (the reference stores (img_keys[ind // c], ind % c)), t2i_index[j] = image indices."""
    import numpy as np
    c = num_captions_per_img_train
    i2t_ranks, t2i_ranks, i2t_index, t2i_index = [], [], [], []
    for i in range(similarities.shape[0]):
        inds = np.argsort(similarities[i, :])[::-1]
        rank = similarities.shape[1]
        for r, ind in enumerate(inds):
            if i * c <= ind < (i + 1) * c:
                rank = r
                break
        i2t_ranks.append(rank)
        i2t_index.append([int(ind) for ind in inds[:num_captions_per_img_val]])
    for j in range(similarities.shape[1]):
        inds = np.argsort(similarities[:, j])[::-1]
        rank = similarities.shape[0]
        for r, ind in enumerate(inds):
            if ind == j // c:
                rank = r
                break
        t2i_ranks.append(rank)
        t2i_index.append([int(ind) for ind in inds[:num_images_per_cap_val]])
    return i2t_ranks, t2i_ranks, i2t_index, t2i_index
