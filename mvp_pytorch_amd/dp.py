"""Data-parallel gradient exchange: one process per GPU, image-text pairs sharded across ranks,
gradients averaged with bucketed all-reduce on RCCL (backend 'nccl' on ROCm) over xGMI,
overlapped with the backward pass.

Replaces what DeepSpeed ZeRO-2 / DDP did implicitly for the reference
(oscar/run_pretrain_ml.py:406-418; oscar/tmp_config.json:11-20 — fp16 gradients, reduce buckets of
2e8 elements, overlap_comm).  The in-batch contrastive / hard-negative step stays rank-local exactly
as in the reference (no feature all-gather, modeling_vlbert.py:525-534), so the gradient all-reduce
is the only per-step collective.

Design
  * gradients live in flat f32 bucket buffers (>= `bucket_mb` MiB each, filled in reverse parameter
    order = the order backward finishes them); every p.grad is a view into its bucket, so autograd
    accumulates straight into the communication buffer and nothing is copied before or after;
  * HOT and COLD buckets.  A parameter is hot once ANY rank has produced a gradient for it (the
    used-parameter bitmap below is all-reduced, so every rank holds the same hot set).  Hot buckets
    hold hot parameters only: a post-accumulate-grad hook per parameter counts readiness and, when the
    last one of a bucket has landed, its all-reduce is launched asynchronously while the remaining
    backward kernels keep running.  Parameters that have never produced a gradient (qa_head when
    qa_ans is None, modeling_vlbert.py:1184; a data-conditional head before its first use) sit in
    cold buckets, which are only reduced in finish() — so a parameter that produces its FIRST
    gradient in any later step is still exchanged correctly (it is promoted to hot afterwards), and a
    never-used parameter cannot hold back the overlap of the others.  In the first step nothing is
    known yet and every bucket is reduced in finish();
  * buckets are launched strictly in index order on every rank (a ready bucket waits for its
    predecessors, as DDP does): which hot parameters receive a gradient in a given step can differ
    between ranks (a shard without a masked tag row skips half_mlm), and collectives on one
    communicator must be issued in the same order everywhere; finish() launches whatever is left;
  * `comm_dtype=torch.bfloat16` (default under RCCL): each bucket is rounded to bf16 for the wire
    and the averaged result converted back into the f32 bucket — half the xGMI bytes (0.49 instead of
    0.98 GB per step for BiBertImgForPreTraining).  The reference exchanged fp16 gradients under
    DeepSpeed; bf16 keeps f32's exponent range, so no loss scaling is involved;
  * ROW-SPARSE parameters (`sparse_rows=[...]`, the 86 051 x 768 word-embedding table whose f32
    gradient is 264 MB and is produced LAST, so it cannot overlap the backward pass): each gets a
    bucket of its own; when the step has told the exchange which rows were looked up
    (note_rows(param, ids): token, phrase and tag ids of the rank's shard), the ranks all-gather their
    unique row ids, every rank forms the same sorted union, and only those rows are all-reduced (a
    compact [U, H] buffer in the wire dtype) and scattered back; rows outside the union are zero on
    every rank already.  A rank without noted ids (or a caller that never calls note_rows) makes all
    ranks fall back to the dense all-reduce of that bucket for the step, so the result never depends
    on the optimisation;
  * one backward per zero_grad(), or gradient accumulation inside `with sync.no_sync():` for all but
    the last backward — a second backward outside no_sync() would add into buckets that are already
    being reduced and raises.
xGMI is point-to-point (7 links x ~153 GB/s per GPU): a ring all-reduce of S bytes moves 2*(7/8)*S
per GPU at the per-link rate, ~5.6 ms for 0.49 GB on one ring if it were not overlapped (RCCL runs
several rings/trees over the links in parallel; the multi-GPU curve of this design has not been
measured by the builder: no multi-GPU box was available).
"""
import contextlib

import torch
import torch.distributed as dist


class GradSync:
    def __init__(self, model, bucket_mb=64, process_group=None, overlap=True, comm_dtype="auto", sparse_rows=()):
        self.group = process_group
        self.sparse = {p for p in sparse_rows if p.requires_grad and p.dim() == 2}
        self._rows = {}           # row-sparse parameter -> list of id tensors noted for this step
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.overlap = overlap
        self.params = [p for p in model.parameters() if p.requires_grad]
        self.index = {p: i for i, p in enumerate(self.params)}
        self.cap = max(1, int(bucket_mb * (1 << 20) // 4))
        backend = dist.get_backend(process_group) if dist.is_initialized() else "none"
        self._avg = backend == "nccl"   # RCCL averages in the collective; gloo sums, we scale
        if comm_dtype == "auto":
            comm_dtype = torch.bfloat16 if backend == "nccl" else torch.float32
        self.comm_dtype = comm_dtype
        self._hot = None          # params some rank has produced a gradient for; None = unknown (step 0)
        self._accumulating = False
        self._rebuild = False
        self._build()
        if self.world > 1:
            for p in self.params:
                p.register_post_accumulate_grad_hook(self._hook)
        self.zero_grad()

    # ------------------------------------------------------------------ bucket layout
    def _build(self):
        """(Re)build the buckets: hot parameters in reverse order (early-launchable), then cold ones."""
        self.buckets = []        # dicts: flat, items [(param, offset, numel)], hot, pending, work, ...
        self.where = {}          # param -> bucket index
        self.span = {}           # param -> (offset, numel, address of its view)
        hot = [p for p in reversed(self.params) if self._hot is not None and p in self._hot]
        cold = [p for p in reversed(self.params) if self._hot is None or p not in self._hot]
        for group, is_hot in ((hot, True), (cold, False)):
            cur, cur_n = [], 0
            for p in group:
                if p in self.sparse:      # a bucket of its own, exchanged by rows
                    if cur:
                        self._close(cur, cur_n, is_hot)
                        cur, cur_n = [], 0
                    self._close([(p, 0, p.numel())], p.numel(), is_hot)
                    self.buckets[-1]["rows_of"] = p
                    continue
                if cur and cur_n + p.numel() > self.cap:
                    self._close(cur, cur_n, is_hot)
                    cur, cur_n = [], 0
                cur.append((p, cur_n, p.numel()))
                cur_n += p.numel()
            if cur:
                self._close(cur, cur_n, is_hot)
        self.n_hot = sum(1 for b in self.buckets if b["hot"])

    def _close(self, items, n, is_hot):
        p0 = items[0][0]
        idx = len(self.buckets)
        flat = torch.zeros(n, device=p0.device, dtype=torch.float32)
        self.buckets.append(dict(flat=flat, items=items, hot=is_hot, pending=0, work=None, wire=None, streams=set(),
                                 rows_of=None, union=None))
        for p, off, k in items:
            self.where[p] = idx
            self.span[p] = (off, k, flat.data_ptr() + 4 * off)   # O(1) lookups in the per-parameter hook

    # ------------------------------------------------------------------ per step
    def zero_grad(self):
        """Zero the bucket buffers and (re)attach every p.grad as a view into its bucket.  Use this
        instead of optimizer.zero_grad() when a GradSync is active.  Re-lays the buckets first when
        the hot set changed in the exchange that just finished."""
        if self._rebuild:
            self._rebuild = False
            for p in self.params:
                p.grad = None
            self._build()
        for b in self.buckets:
            b["flat"].zero_()
            b["work"] = b["wire"] = b["union"] = None
            b["streams"] = set()
            for p, off, n in b["items"]:
                view = b["flat"][off:off + n].view_as(p)
                if p.grad is None or p.grad.data_ptr() != view.data_ptr():
                    p.grad = view
            b["pending"] = len(b["items"])
        self._ready = set()
        self._rows = {}
        self._next = 0            # buckets [0, _next) have been launched this step

    def note_rows(self, param, ids):
        """Tell the exchange which rows of a row-sparse parameter this rank's step looks up (every id
        tensor that indexes the table: call once per tensor or pass a list), BEFORE the backward pass
        (a hot bucket is launched from the hook of its last gradient).  Without it the bucket is
        reduced densely."""
        if self.world == 1 or param not in self.sparse:
            return
        ids = ids if isinstance(ids, (list, tuple)) else [ids]
        self._rows.setdefault(param, []).extend(t.reshape(-1) for t in ids if t is not None)

    def _row_union(self, b):
        """All ranks' unique row ids of this step -> the sorted union (identical on every rank), or
        None when some rank has no ids noted (dense fallback, decided from the gathered data alone)."""
        p = b["rows_of"]
        dev = b["flat"].device
        noted = self._rows.get(p)
        mine = torch.unique(torch.cat(noted).to(dev)) if noted else None
        cnt = torch.tensor([-1 if mine is None else mine.numel()], dtype=torch.int64, device=dev)
        cnts = [torch.empty_like(cnt) for _ in range(self.world)]
        dist.all_gather(cnts, cnt, group=self.group)
        cnts = [int(c.item()) for c in cnts]
        if min(cnts) < 0:
            return None
        cap = max(1, max(cnts))
        pad = torch.full((cap,), -1, dtype=torch.int64, device=dev)
        if mine is not None and mine.numel():
            pad[:mine.numel()] = mine
        parts = [torch.empty_like(pad) for _ in range(self.world)]
        dist.all_gather(parts, pad, group=self.group)
        allids = torch.cat([q[:c] for q, c in zip(parts, cnts)])
        union = torch.unique(allids)
        rows = p.shape[0]
        if union.numel() and (int(union.min()) < 0 or int(union.max()) >= rows):
            raise RuntimeError("GradSync.note_rows: row id outside the table")
        if union.numel() == 0 or union.numel() * 2 > rows:
            return None               # nothing to gain: dense exchange (same decision on every rank)
        return union

    @contextlib.contextmanager
    def no_sync(self):
        """Gradient accumulation: backward passes inside this context only accumulate into the
        buckets (no readiness counting, no launches); the last backward goes outside it."""
        self._accumulating = True
        try:
            yield
        finally:
            self._accumulating = False

    def _hook(self, p):
        if self.world == 1:
            return
        idx = self.where[p]
        b = self.buckets[idx]
        off, n, ptr = self.span[p]
        if p.grad.data_ptr() != ptr:
            # autograd replaced the view (e.g. dtype change): copy into the bucket, re-attach
            b["flat"][off:off + n].copy_(p.grad.reshape(-1))
            p.grad = b["flat"][off:off + n].view_as(p)
        if p.grad.is_cuda:
            # the gradient was produced on the stream current in this hook (sub-networks may run on a
            # side stream, engine.side_stream): remember which streams fed this bucket
            b["streams"].add(torch.cuda.current_stream(p.grad.device))
        if self._accumulating:
            return
        if idx < self._next:
            raise RuntimeError("GradSync: a gradient arrived for a bucket that is already being reduced — run one "
                               "backward per zero_grad(), or wrap all but the last backward in `with sync.no_sync():`")
        if p in self._ready:
            return                # the same parameter used twice in one graph fires once; be tolerant
        self._ready.add(p)
        b["pending"] -= 1
        if self.overlap and b["hot"]:
            while self._next < self.n_hot and self.buckets[self._next]["pending"] == 0:
                self._launch(self._next)

    def _launch(self, idx):
        b = self.buckets[idx]
        op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
        assert idx == self._next, "buckets are launched in index order"
        if b["flat"].is_cuda:
            # gradients produced on another stream than the launching one may still be in flight
            # although their hooks have fired on the host
            cur = torch.cuda.current_stream(b["flat"].device)
            for st in b["streams"]:
                if st != cur:
                    cur.wait_stream(st)   # everything queued there so far includes the gradients
        src = b["flat"]
        if b["rows_of"] is not None:
            union = self._row_union(b)     # two small all-gathers; every rank takes the same branch
            if union is not None:
                b["union"] = union
                src = b["flat"].view_as(b["rows_of"]).index_select(0, union)
        wire = src if (self.comm_dtype == torch.float32 and src is b["flat"]) else src.to(self.comm_dtype)
        b["wire"] = wire
        b["work"] = dist.all_reduce(wire, op=op, group=self.group, async_op=True)
        self._next = idx + 1

    def __call__(self):
        """Finish the step's exchange: launch the buckets that are still waiting (in index order on
        every rank), wait for all of them, scale if the backend summed."""
        if self.world == 1:
            return
        for idx in range(self._next, len(self.buckets)):
            self._launch(idx)
        # which parameters produced a gradient on ANY rank (DDP's used-parameter bitmap): those keep
        # the averaged gradient on every rank, the others keep grad = None everywhere, so replicas
        # apply identical updates even when a shard skipped a head
        dev = self.buckets[0]["flat"].device
        used = torch.tensor([1 if p in self._ready else 0 for p in self.params], dtype=torch.int32).to(dev)
        used_work = dist.all_reduce(used, op=dist.ReduceOp.MAX, group=self.group, async_op=True)
        for b in self.buckets:
            b["work"].wait()
            if b["union"] is not None:
                # compact rows back into the table gradient (rows outside the union are zero everywhere)
                if b["wire"].is_cuda:
                    b["wire"].record_stream(torch.cuda.current_stream(b["wire"].device))
                red = b["wire"].to(torch.float32)
                if not self._avg:
                    red = red * (1.0 / self.world)
                b["flat"].view_as(b["rows_of"]).index_copy_(0, b["union"], red)
                b["wire"] = b["union"] = None
                continue
            if b["wire"] is not b["flat"]:
                if b["wire"].is_cuda:
                    b["wire"].record_stream(torch.cuda.current_stream(b["wire"].device))
                b["flat"].copy_(b["wire"])          # bf16 -> f32, ordered after the collective by wait()
            if not self._avg:
                b["flat"].mul_(1.0 / self.world)
            b["wire"] = None
        used_work.wait()
        used = used.tolist()
        # parameters no rank produced a gradient for keep grad = None, as under DDP with
        # find_unused_parameters=True (run_pretrain_ml.py:415-418): the optimizer skips them
        for p, u in zip(self.params, used):
            if not u:
                p.grad = None
        # hot set = every parameter that has EVER produced a gradient on any rank (identical on all
        # ranks: it is derived from the all-reduced bitmap only).  One that is missing in some step
        # only delays launches to finish(); a newly used one moves its bucket layout at the next step.
        now = {p for p, u in zip(self.params, used) if u}
        hot = now if self._hot is None else (self._hot | now)
        if hot != self._hot:
            self._hot = hot
            self._rebuild = True


def all_reduce_metrics(values, device):
    """The reference's only direct collective: a 3-float all_reduce of [loss, n_examples, n_steps]
    at checkpoint time (run_pretrain_ml.py:688-689)."""
    t = torch.tensor(values, dtype=torch.float32, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t)
    return t.tolist()
