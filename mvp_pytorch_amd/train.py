"""Pre-training step harness: the counterpart of `forward_backward` + optimizer step in
oscar/run_pretrain_ml.py:519-562,632-644 (DeepSpeed branch call signature :528-531, without
DeepSpeed), plus the data-parallel gradient exchange."""
import inspect

import torch

from .optimization import AdamW, WarmupLinearSchedule


def build_optimizer(model, lr=5e-5, adam_epsilon=1e-8, weight_decay=0.01, warmup_steps=0, t_total=100000, grad_sync=None):
    """run_pretrain_ml.py:379-393 — no decay on biases and LayerNorm weights.  grad_sync: a dp.GradSync(shard_optimizer=True)
    selects the ZeRO-1 optimizer over its buckets (optimization.ShardedAdamW)."""
    no_decay = ["bias", "LayerNorm.weight"]
    named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
    groups = [{"params": [p for n, p in named if not any(nd in n for nd in no_decay)], "weight_decay": weight_decay},
              {"params": [p for n, p in named if any(nd in n for nd in no_decay)], "weight_decay": 0.0}]
    if grad_sync is not None and getattr(grad_sync, "shard_optimizer", False):
        from .optimization import ShardedAdamW
        opt = ShardedAdamW(groups, grad_sync, lr=lr, eps=adam_epsilon)
    else:
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
        kw = dict(input_ids=batch["input_ids"], token_type_ids=batch["segment_ids"], attention_mask=batch["input_mask"],
                  masked_lm_labels=batch["lm_label_ids"], next_sentence_label=batch["is_next"], img_feats=feats)
        if hc is not None:
            kw["host_counts"] = hc      # rows / lmax / scored of the single-stream batch: no read-back inside the step
        return kw
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
                  grad_sync=None, return_losses=False, forward=None):
    """One optimisation step.  grad_sync: optional callable run between backward and the
    optimizer (the data-parallel all-reduce, mvp_pytorch_amd.dp.GradSync).  forward: optional callable(model, batch) -> outputs
    (loss first) for models that do not take the pre-training batch layout (the fine-tune wrappers)."""
    if grad_sync is not None and getattr(grad_sync, "sparse", None):
        # the word table's gradient is row-sparse: tell the exchange which rows this shard looks up
        # (before backward: a hot bucket goes out from the hook of its last gradient)
        emb = getattr(getattr(getattr(model, "bert", None), "embeddings", None), "word_embeddings", None)
        if emb is not None and emb.weight in grad_sync.sparse:
            hr = batch.get("word_rows")
            if hr is not None and hasattr(grad_sync, "exchange_rows_early"):
                # the batch came with its looked-up rows (synthetic.word_rows / a collate function): the ranks' union is formed now,
                # host to host, and waits on the device when the table's bucket goes out (no collective or read-back in the launch)
                grad_sync.exchange_rows_early(emb.weight, hr.ids)
            else:
                grad_sync.note_rows(emb.weight, [batch.get("input_ids_a"), batch.get("input_ids_b"), batch.get("input_ids")])
    outputs = forward(model, batch) if forward is not None else model(**model_inputs(batch, max_tag_length))
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
        return [o.detach() if isinstance(o, torch.Tensor) else o for o in outputs]
    return loss.detach()


class GraphedStep:
    """pretrain_step whose DEVICE work is captured once per batch signature as a HIP graph and replayed (the host
    needs 7-9 ms to queue the ~460 launches of a step; a replay is one launch).  Same contract as pretrain_step: every call is one
    optimisation step.

        step = GraphedStep(model, optimizer, scheduler, max_tag_length=20, max_grad_norm=10.0, grad_sync=sync)
        loss = step(batch)

    What a captured step needs, and how it is met:
      * no host read-back inside the step: the batch carries `host_counts` (synthetic.host_counts / a collate function), the joint +
        hard-negative pass runs on its device-side row count (mvptr_layer_desc.rows_dev) — the sync-free step of round 4;
      * nothing the host changes per step may be a launch argument: the learning rate / bias correction / weight decay live in the
        AdamW kernels' descriptor tables, refreshed from the host before every replay (AdamW.advance); the dropout seeds are
        arguments, so the kernels mix in a device-side salt word the graph increments (mvptr_set_dropout_salt); torch's own
        draws (randperm of the hard-negative split, WRA picks) come from the graph-registered device generator;
      * static addresses: inputs are copied into the graph's own buffers, gradients live in the GradSync arena.
    A signature (tensor shapes + dtypes + the host counts + train mode) is stepped eagerly `warm_steps` times first (allocations,
    weight copies, optimizer state), then captured; a capture that fails — a host read-back (batches without host_counts), a
    collective, an unfused optimizer group — is remembered and that signature stays eager.  Variable-length data gives every
    batch its own counts, i.e. its own signature: such jobs run eagerly (bound-sized uni-modal passes would lift that; not built).
    Multi-rank jobs run eagerly (the bucketed exchange is driven from Python hooks)."""

    def __init__(self, model, optimizer, scheduler, max_tag_length=20, loss_weight=1.0, max_grad_norm=0.0, grad_sync=None,
                 warm_steps=2, max_graphs=4, enabled=True, forward=None):
        self.forward = forward       # callable(model, batch) -> outputs, loss first (default: the pre-training batch layout, model_inputs)
        self.model, self.optimizer, self.scheduler = model, optimizer, scheduler
        self.max_tag_length, self.loss_weight, self.max_grad_norm, self.grad_sync = max_tag_length, loss_weight, max_grad_norm, grad_sync
        self.warm_steps, self.max_graphs, self.enabled = int(warm_steps), int(max_graphs), bool(enabled)
        self._seen = {}        # signature -> eager steps so far
        self._graphs = {}      # signature -> dict(graph, static, plan, loss, outputs) | None (capture failed: stay eager)
        self.replays = self.eager_steps = self.captures = 0
        self.last_error = None
        self._stream = None

    # ------------------------------------------------------------------
    def _eager(self, batch, return_losses):
        self.eager_steps += 1
        return pretrain_step(self.model, batch, self.optimizer, self.scheduler, self.max_tag_length, self.loss_weight,
                             self.max_grad_norm, self.grad_sync, return_losses, self.forward)

    def _signature(self, batch):
        items = []
        for k in sorted(batch):
            v = batch[k]
            if isinstance(v, torch.Tensor):
                if not v.is_cuda:
                    return None
                items.append((k, tuple(v.shape), str(v.dtype)))
            elif isinstance(v, dict):
                items.append((k, tuple(sorted((kk, int(vv)) for kk, vv in v.items()))))
            elif getattr(v, "host_only", False):
                continue                # host-side companions of a batch (synthetic.HostRows): not an input of the device work
            elif v is not None:
                return None
        return tuple(items) + (("training", bool(self.model.training)),)

    def _capturable(self):
        if not self.enabled or self.grad_sync is None or not hasattr(self.grad_sync, "zero_grad"):
            return False
        if getattr(self.grad_sync, "exchange", False):
            return False
        if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
            return False
        p = next(self.model.parameters(), None)
        return p is not None and p.is_cuda and hasattr(self.optimizer, "advance")

    def __call__(self, batch, return_losses=False):
        sig = self._signature(batch) if self._capturable() else None
        if sig is None:
            return self._eager(batch, return_losses)
        # Every step of a capturable job — eager warm-up, capture, replay — runs on this object's own stream: autograd pins an
        # AccumulateGrad node to the stream that was current when the node was made, and a node made on the legacy default stream
        # makes the captured backward pass wait on a stream that is not capturing (hipStreamEndCapture then takes the process down)
        dev = next(self.model.parameters()).device
        if self._stream is None:
            self._stream = torch.cuda.Stream(dev)
        outer = torch.cuda.current_stream(dev)
        self._stream.wait_stream(outer)
        with torch.cuda.stream(self._stream):
            out = self._dispatch(sig, batch, return_losses)
        outer.wait_stream(self._stream)
        for t in (out if isinstance(out, (list, tuple)) else (out,)):
            if isinstance(t, torch.Tensor) and t.is_cuda:
                t.record_stream(outer)
        return out

    def _dispatch(self, sig, batch, return_losses):
        ent = self._graphs.get(sig, False)
        if ent is None:                                    # capture failed before
            return self._eager(batch, return_losses)
        if ent is False:
            n = self._seen.get(sig, 0)
            if n < self.warm_steps or len(self._graphs) >= self.max_graphs:
                self._seen[sig] = n + 1
                return self._eager(batch, return_losses)
            ent = self._capture(sig, batch)
            if ent is None:
                return self._eager(batch, return_losses)
        return self._replay(ent, batch, return_losses)

    # ------------------------------------------------------------------
    def _step_body(self, batch):
        """exactly pretrain_step's device work (the optimizer in launch-only mode, the scheduler left to the host)"""
        outputs = self.forward(self.model, batch) if self.forward is not None else self.model(**model_inputs(batch, self.max_tag_length))
        loss = self.loss_weight * outputs[0]
        loss.backward()
        self.grad_sync(want_norm=self.max_grad_norm > 0) if _takes_want_norm(self.grad_sync) else self.grad_sync()
        scale = clip_coefficient(self.model, self.grad_sync, self.max_grad_norm) if self.max_grad_norm > 0 else None
        if scale is not None:
            self.optimizer.step(grad_scale=scale)
        else:
            self.optimizer.step()
        self.grad_sync.zero_grad()
        return loss.detach(), [o.detach() if isinstance(o, torch.Tensor) else o for o in outputs]

    def _capture(self, sig, batch):
        from . import hip
        dev = next(self.model.parameters()).device
        static = {k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}
        salt = hip.dropout_salt(dev)
        plan = []
        graph = torch.cuda.CUDAGraph()
        torch.cuda.synchronize(dev)
        self.optimizer._graph_plan = plan
        try:
            with torch.cuda.graph(graph, stream=self._stream, capture_error_mode="thread_local"):
                salt.add_(1)                               # fresh dropout masks per replay (the seeds are captured arguments)
                loss, outputs = self._step_body(static)
        except Exception as e:                             # noqa: BLE001 — anything that cannot be captured: stay eager, say why once
            self.optimizer._graph_plan = None
            self._graphs[sig] = None
            import traceback
            import warnings
            where = [ln for ln in traceback.format_exc().splitlines() if ln.lstrip().startswith("File ")][-3:]
            self.last_error = "%s: %s  [%s]" % (type(e).__name__, str(e)[:300], " <- ".join(w.strip() for w in reversed(where)))
            warnings.warn("GraphedStep: this step cannot be captured (%s); running it eagerly" % self.last_error)
            torch.cuda.synchronize(dev)
            try:
                self.grad_sync.zero_grad()                 # the aborted pass may have marked gradients as delivered
            except Exception:                              # noqa: BLE001
                pass
            return None
        self.optimizer._graph_plan = None
        ent = dict(graph=graph, static=static, plan=plan, loss=loss, outputs=outputs)
        self._graphs[sig] = ent
        self.captures += 1
        return ent

    def _replay(self, ent, batch, return_losses):
        static, last = ent["static"], ent.setdefault("last", {})
        for k, v in batch.items():
            if isinstance(v, torch.Tensor):
                tag = (v.data_ptr(), v._version)
                if last.get(k) != tag:                     # the same tensor, unchanged since its last copy (a loop over one batch): no copy
                    static[k].copy_(v, non_blocking=True)
                    last[k] = tag
        self.optimizer.advance(ent["plan"])
        ent["graph"].replay()
        self.scheduler.step()
        self.replays += 1
        if return_losses:
            return [o.clone() if isinstance(o, torch.Tensor) else o for o in ent["outputs"]]
        return ent["loss"].clone()
