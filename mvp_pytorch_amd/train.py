"""Pre-training step harness: the counterpart of `forward_backward` + optimizer step in
oscar/run_pretrain_ml.py:519-562,632-644 (DeepSpeed branch call signature :528-531, without
DeepSpeed), plus the data-parallel gradient exchange."""
import inspect

import torch

from .optimization import AdamW, WarmupLinearSchedule


def build_optimizer(model, lr=5e-5, adam_epsilon=1e-8, weight_decay=0.01, warmup_steps=0, t_total=100000):
    """run_pretrain_ml.py:379-393 — no decay on biases and LayerNorm weights."""
    no_decay = ["bias", "LayerNorm.weight"]
    named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
    groups = [{"params": [p for n, p in named if not any(nd in n for nd in no_decay)], "weight_decay": weight_decay},
              {"params": [p for n, p in named if any(nd in n for nd in no_decay)], "weight_decay": 0.0}]
    opt = AdamW(groups, lr=lr, eps=adam_epsilon)
    sched = WarmupLinearSchedule(opt, warmup_steps=warmup_steps, t_total=t_total)
    return opt, sched


def model_inputs(batch, max_tag_length):
    """batch (13 tensors of OscarTSVDataset_C, oscar_tsv4.py:363-377) -> model kwargs
    (run_pretrain_ml.py:528-531); a single-stream batch (input_ids / input_mask / ...) maps to
    BertImgForPreTraining's arguments (run_pretrain_ml.py:533 positional order)."""
    feats = batch.get("img_feats")
    hc = batch.get("host_counts")
    if hc is not None and "input_mask_a" in batch:
        # cheap host-side sanity of counts a caller may have carried over from another batch (the device-side checks —
        # mvptr_check_counts on rows / longest sequence — traps on a mismatch; mvptr_compact_scored on the scored rows — raises at the next read-back)
        ma, mb = batch["input_mask_a"], batch["input_mask_b"]
        ok = (0 < int(hc["rows_a"]) <= ma.numel() and 0 < int(hc["rows_b"]) <= mb.numel() and 0 < int(hc["lmax_a"]) <= ma.shape[1]
              and 0 < int(hc["lmax_b"]) <= mb.shape[1] and 0 <= int(hc["scored_a"]) <= ma.numel() and 0 <= int(hc["scored_b"]) <= mb.numel())
        if not ok:
            raise ValueError("host_counts %r do not fit this batch (masks %s / %s)" % (dict(hc), tuple(ma.shape), tuple(mb.shape)))
    if "img_feats_bf16" in batch:
        # K-padded bf16 operand produced by the input pipeline (input_pipeline.PretrainBatchStager): the model skips its cast
        fb = batch["img_feats_bf16"]
        feats = fb.view(batch["input_ids_b" if "input_ids_b" in batch else "input_ids"].shape[0], -1, fb.shape[-1])
    if "input_ids" in batch:
        return dict(input_ids=batch["input_ids"], token_type_ids=batch["segment_ids"], attention_mask=batch["input_mask"],
                    masked_lm_labels=batch["lm_label_ids"], next_sentence_label=batch["is_next"], img_feats=feats)
    return dict(input_ids_a=batch["input_ids_a"], token_type_ids_a=batch["segment_ids_a"],
                attention_mask_a=batch["input_mask_a"], masked_lm_labels_a=batch["lm_label_ids_a"],
                input_ids_b=batch["input_ids_b"], img_feats=feats,
                token_type_ids_b=batch["segment_ids_b"], attention_mask_b=batch["input_mask_b"],
                masked_lm_labels_b=batch["lm_label_ids_b"], phrase_index=batch.get("phrase_index"),
                img_index=batch.get("image_index"), max_tag_length=max_tag_length, host_counts=batch.get("host_counts"))


def clip_coefficient(model, grad_sync, max_grad_norm):
    """run_pretrain_ml.py:639-640 (torch.nn.utils.clip_grad_norm_(model.parameters(), max_grad_norm)); the
    DeepSpeed recipe clips at 10.0 (oscar/tmp_config.json).  With a gradient arena on a HIP device: global norm
    over the flat buckets in two kernel passes, returned as a device scalar for the fused AdamW (no pass that
    rescales 1 GB of gradients, no host sync).  Otherwise torch's clip (rescales in place) -> None."""
    flats = grad_sync.flats() if (grad_sync is not None and hasattr(grad_sync, "flats")) else None
    if flats and flats[0].is_cuda:
        if hasattr(grad_sync, "clip_coef"):
            # per-chunk partial sums in fixed slots: the buckets a multi-rank exchange has already summed behind their
            # collectives (GradSync.__call__(want_norm=True)) are not read again
            return grad_sync.clip_coef(max_grad_norm)[1]
        from . import hip
        # the partial-sum scratch belongs to the arena it is sized for (one GradSync per model)
        _, coef, grad_sync._clip_scratch = hip.grad_clip_coef(flats, max_grad_norm, getattr(grad_sync, "_clip_scratch", None))
        return coef
    torch.nn.utils.clip_grad_norm_(model.parameters(), max_grad_norm)
    return None


def _takes_want_norm(fn):
    """dp.GradSync (and anything with its call signature) is told whether the clip norm is wanted; a plain callable is not"""
    try:
        params = inspect.signature(fn).parameters
    except (TypeError, ValueError):
        return False
    return "want_norm" in params or any(p.kind == inspect.Parameter.VAR_KEYWORD for p in params.values())


def pretrain_step(model, batch, optimizer, scheduler, max_tag_length=20, loss_weight=1.0, max_grad_norm=0.0,
                  grad_sync=None, return_losses=False):
    """One optimisation step.  grad_sync: optional callable run between backward and the
    optimizer (the data-parallel all-reduce, mvp_pytorch_amd.dp.GradSync)."""
    if grad_sync is not None and getattr(grad_sync, "sparse", None):
        # the word table's gradient is row-sparse: tell the exchange which rows this shard looks up
        # (before backward: a hot bucket goes out from the hook of its last gradient)
        emb = getattr(getattr(getattr(model, "bert", None), "embeddings", None), "word_embeddings", None)
        if emb is not None and emb.weight in grad_sync.sparse:
            grad_sync.note_rows(emb.weight, [batch.get("input_ids_a"), batch.get("input_ids_b"), batch.get("input_ids")])
    outputs = model(**model_inputs(batch, max_tag_length))
    loss = loss_weight * outputs[0]
    loss.backward()
    if grad_sync is not None:
        if _takes_want_norm(grad_sync):     # chosen up front: an exception inside the exchange must not re-run it
            grad_sync(want_norm=max_grad_norm > 0)
        else:                               # a plain callable
            grad_sync()
    grad_scale = None
    if max_grad_norm > 0:
        grad_scale = clip_coefficient(model, grad_sync, max_grad_norm)
    if grad_scale is not None:
        optimizer.step(grad_scale=grad_scale)     # the clip coefficient multiplies the gradients inside the update kernels
    else:
        optimizer.step()
    scheduler.step()
    if grad_sync is not None and hasattr(grad_sync, "zero_grad"):
        grad_sync.zero_grad()   # keeps p.grad attached to the communication buckets
    else:
        optimizer.zero_grad(set_to_none=True)
    if return_losses:
        return [o.detach() for o in outputs]
    return loss.detach()
