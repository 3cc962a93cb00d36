"""Data-parallel gradient exchange: one process per GPU, image-text pairs sharded across ranks,
gradients averaged with bucketed all-reduce on RCCL (backend 'nccl' on ROCm) over xGMI,
overlapped with the backward pass.

Replaces what DeepSpeed ZeRO-2 / DDP did implicitly for the reference
(oscar/run_pretrain_ml.py:406-418; oscar/tmp_config.json:11-20 — reduce buckets of 2e8 elements,
overlap_comm).  The in-batch contrastive / hard-negative step stays rank-local exactly as in
the reference (no feature all-gather, modeling_vlbert.py:525-534), so the gradient all-reduce is
the only per-step collective.

Design
  * gradients live in flat f32 bucket buffers (>= `bucket_mb` MiB each, filled in reverse parameter
    order = the order backward finishes them); every p.grad is a view into its bucket, so autograd
    accumulates straight into the communication buffer and nothing is copied before or after;
  * a post-accumulate-grad hook per parameter counts readiness; when the last expected gradient of
    a bucket has landed, its all-reduce is launched asynchronously on RCCL's stream while the
    remaining backward kernels keep running (xGMI is point-to-point, 7 links x ~153 GB/s: a ring
    all-reduce of S bytes costs ~2*(7/8)*S / link rate, ~11 ms for the 0.98 GB of f32 gradients of
    BiBertImgForPreTraining if not overlapped);
  * parameters that produced no gradient in the previous step (qa_head when qa_ans is None,
    modeling_vlbert.py:1184) are not waited for;
  * buckets are launched strictly in index order (a ready bucket waits for its predecessors, as
    DDP does): which parameters receive a gradient can differ between ranks (a shard without a
    masked tag row skips half_mlm), and collectives on one communicator must be issued in the
    same order everywhere; finish() launches whatever is left, again in index order.
"""
import torch
import torch.distributed as dist


class GradSync:
    def __init__(self, model, bucket_mb=64, process_group=None, overlap=True):
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.overlap = overlap
        self.params = [p for p in model.parameters() if p.requires_grad]
        self.buckets = []        # dicts: flat, items [(param, offset, numel)], pending, work
        self.where = {}          # param -> bucket index
        self.span = {}           # param -> (offset, numel, address of its view)
        cap = max(1, int(bucket_mb * (1 << 20) // 4))
        cur, cur_n = [], 0
        for p in reversed(self.params):
            if cur and cur_n + p.numel() > cap:
                self._close(cur, cur_n)
                cur, cur_n = [], 0
            cur.append((p, cur_n, p.numel()))
            cur_n += p.numel()
        if cur:
            self._close(cur, cur_n)
        self._expected = None     # params that produced a gradient in the previous step
        self._ready = set()
        self._launched = []
        backend = dist.get_backend(process_group) if dist.is_initialized() else "none"
        self._avg = backend == "nccl"   # RCCL averages in the collective; gloo sums, we scale
        if self.world > 1:
            for p in self.params:
                p.register_post_accumulate_grad_hook(self._hook)
        self.zero_grad()

    def _close(self, items, n):
        p0 = items[0][0]
        idx = len(self.buckets)
        self.buckets.append(dict(flat=torch.zeros(n, device=p0.device, dtype=torch.float32), items=items,
                                 pending=0, work=None, streams=set()))
        flat = self.buckets[-1]["flat"]
        for p, off, n in items:
            self.where[p] = idx
            self.span[p] = (off, n, flat.data_ptr() + 4 * off)   # O(1) lookups in the per-parameter hook

    # ------------------------------------------------------------------ per step
    def zero_grad(self):
        """Zero the bucket buffers and (re)attach every p.grad as a view into its bucket.  Use this
        instead of optimizer.zero_grad() when a GradSync is active."""
        for b in self.buckets:
            b["flat"].zero_()
            b["work"] = None
            b["streams"] = set()
            n_exp = 0
            for p, off, n in b["items"]:
                view = b["flat"][off:off + n].view_as(p)
                if p.grad is None or p.grad.data_ptr() != view.data_ptr():
                    p.grad = view
                if self._expected is None or p in self._expected:
                    n_exp += 1
            b["pending"] = n_exp
        self._ready = set()
        self._launched = []
        self._next = 0            # buckets [0, _next) have been launched this step

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
        if p in self._ready:
            return
        self._ready.add(p)
        if p.grad.is_cuda:
            # the gradient was produced on the stream current in this hook (sub-networks may run on a
            # side stream, engine.side_stream): remember which streams fed this bucket
            b["streams"].add(torch.cuda.current_stream(p.grad.device))
        if idx < self._next:
            # its bucket is already being reduced in place: the parameter never produced a gradient
            # before, so nobody was waiting for it.  Failing loudly beats a silently unsynchronised
            # gradient; construct GradSync(..., overlap=False) for models whose set of used
            # parameters grows during training.
            raise RuntimeError("GradSync: a parameter produced its first gradient after its bucket was launched")
        if self._expected is None or p in self._expected:
            b["pending"] -= 1
            if self.overlap:
                while self._next < len(self.buckets) and self.buckets[self._next]["pending"] == 0:
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
        b["work"] = dist.all_reduce(b["flat"], op=op, group=self.group, async_op=True)
        self._launched.append(idx)
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
            if not self._avg:
                b["flat"].mul_(1.0 / self.world)
        # wait for every parameter that has EVER produced a gradient on this rank: one that is missing
        # in some step only delays launches to finish(), it cannot reorder them
        self._expected = set(self._ready) if self._expected is None else (self._expected | self._ready)
        # parameters no rank produced a gradient for keep grad = None, as under DDP with
        # find_unused_parameters=True (run_pretrain_ml.py:415-418): the optimizer skips them
        used_work.wait()
        for p, u in zip(self.params, used.tolist()):
            if not u:
                p.grad = None


def all_reduce_metrics(values, device):
    """The reference's only direct collective: a 3-float all_reduce of [loss, n_examples, n_steps]
    at checkpoint time (run_pretrain_ml.py:688-689)."""
    t = torch.tensor(values, dtype=torch.float32, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t)
    return t.tolist()
