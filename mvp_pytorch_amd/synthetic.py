"""Synthetic pre-training batches with the layout `OscarTSVDataset_C.__getitem__` produces
(oscar/oscar_datasets_ml/oscar_tsv4.py:363-377, convert_example_to_features :896-1092):
13 tensors per batch, masks prefix-contiguous, phrase ids above `only_word_size`, labels -1 =
ignore, >= 3 valid regions per image (the WRA top-3 needs them, modeling_vlbert.py:1547).
"""
import torch

CLS, SEP, MASK = 101, 102, 103


def synthetic_batch(dims, cfg, seed, single_stream=False, fixed_length=False, device=None):
    """dims: dict(B, T, P, G, R).  cfg: dict with vocab_size, only_word_size, img_feature_dim.
    fixed_length=True fills every slot (roofline run); otherwise lengths are random."""
    g = torch.Generator().manual_seed(int(seed))
    B, T, P, G, R = (dims[k] for k in ("B", "T", "P", "G", "R"))
    V, W, D = cfg["vocab_size"], cfg["only_word_size"], cfg["img_feature_dim"]
    lo = 110 if W < 2000 else 1000

    def ri(a, b, size=()):
        return torch.randint(a, b, size, generator=g)

    La = T if single_stream else T + P
    img = torch.zeros(B, R, D)
    ids_a = torch.zeros(B, La, dtype=torch.long)
    mask_a = torch.zeros(B, La, dtype=torch.long)
    lab_a = torch.full((B, La), -1, dtype=torch.long)
    ids_b = torch.zeros(B, G, dtype=torch.long)
    mask_b = torch.zeros(B, G + R, dtype=torch.long)
    lab_b = torch.full((B, G + R), -1, dtype=torch.long)
    phrase_index = torch.zeros(B, 2, dtype=torch.long)
    image_index = torch.zeros(B, 2, dtype=torch.long)
    for b in range(B):
        # lengths as SURVEY §8(d) specifies for the measurement: regions U{10..50}, tokens U{8..68},
        # phrases U{0..5}, tags U{3..18} (smaller floors for the tiny test shapes)
        n_r = R if fixed_length else int(ri(10 if R >= 20 else min(3, R), R + 1))
        feat = torch.randn(n_r, D, generator=g)
        feat[:, -6:] = torch.rand(n_r, 6, generator=g)
        img[b, :n_r] = feat
        max_t = T - 2
        n_t = max_t if fixed_length else int(ri(8 if max_t >= 16 else min(4, max_t), max_t + 1))
        n_p = 0 if single_stream else (P if fixed_length else int(ri(0, P + 1)))
        toks = ri(lo, W, (n_t,))
        seq = [CLS] + toks.tolist()
        if n_p:
            seq += ri(W, V, (n_p,)).tolist()
        seq.append(SEP)
        ids_a[b, :len(seq)] = torch.tensor(seq)
        mask_a[b, :len(seq)] = 1
        pick = torch.rand(n_t, generator=g) < 0.15
        if not pick.any():
            pick[int(ri(0, n_t))] = True
        for i in torch.nonzero(pick).flatten().tolist():
            lab_a[b, 1 + i] = toks[i]
            if torch.rand((), generator=g) < 0.8:
                ids_a[b, 1 + i] = MASK
        phrase_index[b] = torch.tensor([1 + n_t, 1 + n_t + n_p])
        image_index[b] = torch.tensor([La, La + n_r])
        max_g = G - 2
        n_g = max_g if fixed_length else int(ri(min(3, max_g), max_g + 1))
        tags = ri(lo, W, (n_g,))
        ids_b[b, :n_g + 2] = torch.tensor([CLS] + tags.tolist() + [SEP])
        mask_b[b, :n_g + 2] = 1
        mask_b[b, G:G + n_r] = 1
        pick = torch.rand(n_g, generator=g) < 0.15
        if not pick.any():
            pick[int(ri(0, n_g))] = True
        for i in torch.nonzero(pick).flatten().tolist():
            lab_b[b, 1 + i] = tags[i]
            if torch.rand((), generator=g) < 0.8:
                ids_b[b, 1 + i] = MASK
    if single_stream:
        batch = dict(img_feats=img, input_ids=ids_a, input_mask=torch.cat([mask_a, mask_b[:, G:]], 1),
                     segment_ids=torch.zeros_like(ids_a), lm_label_ids=lab_a, is_next=ri(0, 2, (B,)))
    else:
        batch = dict(img_feats=img, input_ids_a=ids_a, input_mask_a=mask_a, segment_ids_a=torch.zeros_like(ids_a),
                     lm_label_ids_a=lab_a, input_ids_b=ids_b, input_mask_b=mask_b,
                     segment_ids_b=torch.ones_like(ids_b), lm_label_ids_b=lab_b,
                     is_next=torch.zeros(B, dtype=torch.long), is_img_match=torch.zeros(B, dtype=torch.long),
                     phrase_index=phrase_index, image_index=image_index)
    if not single_stream:
        batch["host_counts"] = host_counts(batch)     # input-only row counts, taken while the batch is still on the host
        batch["word_rows"] = word_rows(batch)         # rows of the word table this batch looks up (data-parallel exchange)
    else:
        ln = batch["input_mask"].sum(1)
        batch["host_counts"] = HostCounts(rows=int(ln.sum()), lmax=int(ln.max()), scored=int((batch["lm_label_ids"][:, :T] > -1).sum()))
    if device is not None:
        batch = {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in batch.items()}
    return batch


class HostCounts(dict):
    """Plain host integers that travel inside a batch dict: `.to()` / `.cuda()` / `.cpu()` return the object itself, so
    the usual `{k: v.to(device) for k, v in batch.items()}` keeps working."""

    def to(self, *a, **k):
        return self

    cuda = cpu = pin_memory = to


class HostRows:
    """Host-side int64 ids that stay on the host inside a batch dict (`.to()` returns the object itself; train.GraphedStep leaves
    it out of a batch's signature): the rows of the word table a batch looks up, for dp.GradSync.exchange_rows_early."""
    host_only = True

    def __init__(self, ids):
        self.ids = ids

    def to(self, *a, **k):
        return self

    cuda = cpu = pin_memory = to


def word_rows(batch):
    """Unique word-table rows of a batch (text, phrase and tag ids; computed where the batch is built, on the host): the word
    table's gradient is zero outside them, so a data-parallel job exchanges only the ranks' union of these rows — and, knowing
    them before the step, forms that union host to host ahead of the backward pass (dp.GradSync.exchange_rows_early)."""
    ids = [batch[k].reshape(-1) for k in ("input_ids_a", "input_ids_b", "input_ids") if k in batch]
    return HostRows(torch.unique(torch.cat(ids).cpu()))


def host_counts(batch):
    """The data-dependent counts of a two-stage pre-training batch that only depend on its INPUTS, as a collate function
    or input_pipeline.PretrainBatchStager computes them on the host while the batch is still there: valid rows / longest
    sequence of the text and of the tag + region inputs, scored rows of the two MLM heads.  Handed to the model as
    `host_counts=` (train.model_inputs) so that a training step does not read them back from the device."""
    ma, mb = batch["input_mask_a"], batch["input_mask_b"]
    la, lb = ma.sum(1), mb.sum(1)
    return HostCounts(rows_a=int(la.sum()), lmax_a=int(la.max()), rows_b=int(lb.sum()), lmax_b=int(lb.max()),
                      scored_a=int((batch["lm_label_ids_a"] > -1).sum()), scored_b=int((batch["lm_label_ids_b"] > -1).sum()))


def finetune_host_counts(batch, max_tag_length=20):
    """The same for the fine-tune models (BiImageBertForVQA / ...ForSequenceClassification, `host_counts=`): valid rows / longest
    sequence of the two uni-modal passes and of the JOINT pass — text slots + the visual slots from `max_tag_length` on
    (modeling_vlbert.py:466-468; no hard negatives are mined on this path, so the joint count is input-only too)."""
    ma, mb = batch["input_mask_a"], batch["input_mask_b"]
    la, lb = ma.sum(1), mb.sum(1)
    lj = la + mb[:, max_tag_length:].sum(1)
    return dict(rows_a=int(la.sum()), lmax_a=int(la.max()), rows_b=int(lb.sum()), lmax_b=int(lb.max()), rows_j=int(lj.sum()), lmax_j=int(lj.max()))
