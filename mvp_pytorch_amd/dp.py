"""Gradient arena + data-parallel gradient exchange: one process per GPU, image-text pairs sharded
across ranks, gradients averaged with bucketed all-reduce on RCCL (backend 'nccl' on ROCm) over xGMI,
overlapped with the backward pass.

Replaces what DeepSpeed ZeRO-2 / DDP did implicitly for the reference
(oscar/run_pretrain_ml.py:406-418; oscar/tmp_config.json:11-20 — fp16 gradients, reduce buckets of
2e8 elements, overlap_comm).  The in-batch contrastive / hard-negative step stays rank-local exactly
as in the reference (no feature all-gather, modeling_vlbert.py:525-534), so the gradient all-reduce
is the only per-step collective.

Design
  * GRADIENT ARENA (every world size, 1 included): gradients live in flat f32 bucket buffers
    (>= `bucket_mb` MiB each, filled in reverse parameter order = the order backward finishes them);
    every p.grad is a view into its bucket.  The HIP autograd functions (mvp_pytorch_amd.engine) ask
    for the arena of a parameter (`direct`) or of a whole encoder layer (`arena`: the layer's sixteen
    gradients are laid out back to back in the order of mvptr_layer_grads, include/mvptr.h) and let
    the kernels accumulate straight into it — no per-layer zero fill, no gradient copy, no
    AccumulateGrad add — then report `delivered(p)`.  Everything else reaches the arena through
    autograd's in-place accumulation into p.grad.  One zero fill per bucket per step;
  * HOT and COLD buckets.  A parameter is hot once ANY rank has produced a gradient for it (the
    used-parameter bitmap below is all-reduced, so every rank holds the same hot set).  Hot buckets
    hold hot parameters only: readiness is counted per parameter (post-accumulate-grad hook or
    `delivered`) and, when the last one of a bucket has landed, its all-reduce is launched
    asynchronously while the remaining backward kernels keep running.  Parameters that have never
    produced a gradient (qa_head when qa_ans is None, modeling_vlbert.py:1184; a data-conditional head
    before its first use) sit in cold buckets, which are only reduced in finish() — so a parameter that
    produces its FIRST gradient in any later step is still exchanged correctly (it is promoted to hot
    afterwards), and a never-used parameter cannot hold back the overlap of the others.  A hot
    parameter that no rank has used for `demote_after` consecutive steps goes back to cold (the
    decision comes from the all-reduced bitmap, so it is the same on every rank); a step whose hot
    launches stalled before finish() is counted in `stalled_steps`.  In the first step nothing is
    known yet and every bucket is reduced in finish();
  * buckets are launched strictly in index order on every rank (a ready bucket waits for its
    predecessors, as DDP does): which hot parameters receive a gradient in a given step can differ
    between ranks (a shard without a masked tag row skips half_mlm), and collectives on one
    communicator must be issued in the same order everywhere; finish() launches whatever is left;
  * wire format: `comm_dtype=torch.bfloat16` — each bucket is rounded to bf16 for the wire and the
    averaged result converted back into the f32 bucket: half the xGMI bytes (0.49 instead of 0.98 GB
    per step for BiBertImgForPreTraining) at 8 mantissa bits per summand (the reference exchanged fp16
    under DeepSpeed) — is the default of the two-stage models since round 5 (default_exchange); f32 for
    everything else and on request;
  * ROW-SPARSE parameters (`sparse_rows=[...]`; default for the two-stage models' untied word table
    since round 5; the 86 051 x 768 word-embedding table whose
    f32 gradient is 264 MB and is produced LAST, so it cannot overlap the backward pass): each gets a
    bucket of its own; when the step has told the exchange which rows were looked up
    (note_rows(param, ids): token, phrase and tag ids of the rank's shard), the ranks all-gather their
    unique row ids, every rank forms the same sorted union, and only those rows are all-reduced (a
    compact [U, H] buffer in the wire dtype) and scattered back; rows outside the union are zero on
    every rank already.  A rank without noted ids (or a caller that never calls note_rows) makes all
    ranks fall back to the dense all-reduce of that bucket for the step.  PRECONDITION: every gradient
    row of the table comes from the noted lookups — a table that is tied to another module's weight
    (BertImgForPreTraining ties the MLM decoder to it, modeling_vlbert.py:1095-1100) gets a dense
    gradient and is rejected here.  Formed from DEVICE ids the union needs two small blocking all-gathers and host reads
    inside the launch; a batch that brings its looked-up rows on the HOST (synthetic.word_rows, batch["word_rows"]) has the
    union formed ahead of the backward pass over the gloo control group (exchange_rows_early, round 6): nothing is gathered
    or read back inside the launch then;
  * one backward per zero_grad(), or gradient accumulation inside `with sync.no_sync():` for all but
    the last backward — a second backward outside no_sync() would add into buckets that are already
    being reduced and raises.
  * GLOBAL-NORM CLIP THAT SURVIVES THE OVERLAP (round 5): the recipe clips the global gradient norm
    (oscar/tmp_config.json, run_pretrain_ml.py:639-640), and the coefficient needs every bucket.  Buckets
    start on multiples of NORM_CHUNK elements, so the norm's per-chunk partial sums of squares of ONE
    bucket occupy a fixed run of slots: finish(want_norm=True) queues a bucket's partial sums right
    behind that bucket's collective (while the later buckets are still on the wire), and after the last
    one only the one-workgroup coefficient kernel and AdamW remain (`clip_coef`).  Same slots, same
    values, same summation order as the pass over the whole arena: bit-identical norm;
  * `collective="rs_ag"` (opt-in): every dense bucket goes out as reduce-scatter + all-gather over its chunk-padded span
     instead of one all-reduce; bench.py times both.  Any world size (spans are padded to NORM_CHUNK x world
    elements when the world is not a power of two); on gloo, which has no reduce_scatter_tensor, the same span goes out as
    one all-reduce (CPU tests of the layout);
`force_collectives=True` runs the whole exchange (hooks, bucket launches, wire conversion, row union,
used-parameter bitmap) on a process group of ONE rank: the RCCL code path on a single GPU
(tests/test_dp_gpu.py).
xGMI is point-to-point (7 links x ~153 GB/s per GPU): a ring all-reduce of S bytes moves 2*(7/8)*S
per GPU at the per-link rate, ~11 ms for 0.98 GB on one ring if it were not overlapped (RCCL runs
several rings/trees over the links in parallel).
"""
import contextlib

import sys

import torch
import torch.distributed as dist

from . import engine


NORM_CHUNK = 16384       # elements per partial sum of the gradient-norm pass (SUMSQ_BLOCK_ELEMS, csrc/optim.hip)

_CONTROL_GROUPS = {}     # data group -> host-side (gloo) group of the same ranks, made once per process


def _control_group(group):
    """A gloo group over the ranks of `group` for the few host-side integers a step exchanges (the used-parameter bitmap).
    Over RCCL that exchange used to ride the data communicator: a pageable host-to-device copy, an all-reduce queued behind
    every gradient bucket and a `.tolist()` — i.e. the host waited for the whole backward pass and all collectives before it
    could queue the clip and AdamW, every step (dp.py:373,395), and a multi-rank job lost the step of lead a
    single-rank one has.  The bitmap is host knowledge (which hooks fired), so it is exchanged host to host while the
    device is still busy with the backward pass.  Collective: every rank of the default group must construct its GradSync
    (same order), as with dist.new_group."""
    if dist.get_backend(group) == "gloo":
        return group
    if group not in _CONTROL_GROUPS:
        ranks = None if group is None else dist.get_process_group_ranks(group)
        _CONTROL_GROUPS[group] = dist.new_group(ranks=ranks, backend="gloo")
    return _CONTROL_GROUPS[group]


def default_exchange(model):
    """Exchange options a GradSync takes when none are given (round 5).  The two-stage pre-training / fine-tuning models
    (a `bert.embeddings.word_embeddings` table that no other module shares: their MLM decoders are CLONES of its first
    30 522 rows, modeling_utils.py:279-282) exchange bf16 on the wire (the reference exchanged fp16 under DeepSpeed,
    oscar/tmp_config.json:3-9) and the word table by looked-up rows (264 MB of f32 produced last otherwise: it cannot overlap
    the backward pass) — covered as the default by tests/test_dp_gpu.py::test_two_rank_training_step_keeps_replicas_identical.
    Anything else (tied tables, plain modules): f32 wire, dense exchange."""
    emb = getattr(getattr(getattr(model, "bert", None), "embeddings", None), "word_embeddings", None)
    w = getattr(emb, "weight", None)
    if w is not None and w.requires_grad and w.dim() == 2:
        owners = sum(1 for m in model.modules() for q in m._parameters.values() if q is w)
        if owners == 1:
            return dict(comm_dtype=torch.bfloat16, sparse_rows=[w])
    return dict(comm_dtype=torch.float32, sparse_rows=[])


class GradSync:
    _logged_exchange = False

    def __init__(self, model, bucket_mb=64, process_group=None, overlap=True, comm_dtype="default", sparse_rows="default",
                 force_collectives=False, demote_after=8, check_mixed_use=False, collective="all_reduce", shard_optimizer=False):
        self.check_mixed_use = check_mixed_use
        # ZeRO-1 over the buckets (round 6): every rank keeps the reduce-scattered 1 / world of each bucket's gradient, runs the clip
        # partial sums and AdamW on that shard only (optimization.ShardedAdamW: moments for 1 / world of the parameters) and the
        # UPDATED PARAMETERS are all-gathered instead of the reduced gradients.  Needs a static layout (every parameter "hot" from the
        # first step, no re-layout) and a flat parameter arena congruent with the gradient arena (p.data becomes a view into it).
        self.shard_optimizer = bool(shard_optimizer)
        dflt = default_exchange(model)
        if isinstance(comm_dtype, str) and comm_dtype == "default":
            comm_dtype = dflt["comm_dtype"]
        if isinstance(sparse_rows, str) and sparse_rows == "default":
            sparse_rows = [] if self.shard_optimizer else dflt["sparse_rows"]
        if self.shard_optimizer and sparse_rows:
            raise ValueError("GradSync(shard_optimizer=True) exchanges every parameter densely (sparse_rows must be empty)")
        if collective not in ("all_reduce", "rs_ag"):
            raise ValueError("GradSync: collective must be 'all_reduce' or 'rs_ag'")
        self.collective = collective
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        if force_collectives and not dist.is_initialized():
            raise RuntimeError("GradSync(force_collectives=True) needs an initialised process group")
        self.exchange = self.world > 1 or force_collectives   # collectives are issued
        self.overlap = overlap
        self.params = [p for p in model.parameters() if p.requires_grad]
        self.index = {p: i for i, p in enumerate(self.params)}
        self.sparse = {p for p in sparse_rows if p.requires_grad and p.dim() == 2}
        for p in self.sparse:
            owners = sum(1 for m in model.modules() for q in m._parameters.values() if q is p)
            if owners > 1:
                raise ValueError("GradSync: a row-sparse parameter is shared by %d modules (tied weights): its gradient is "
                                 "dense, exchange it densely" % owners)
        self._rows = {}           # row-sparse parameter -> list of id tensors noted for this step
        self._early_union = {}    # row-sparse parameter -> union formed ahead of the backward pass (exchange_rows_early), False = dense
        self.cap = max(1, int(bucket_mb * (1 << 20) // 4))
        backend = dist.get_backend(process_group) if dist.is_initialized() else "none"
        self._avg = backend == "nccl"   # RCCL averages in the collective; gloo sums, we scale
        # rs_ag on gloo (no reduce_scatter_tensor there) is EMULATED by an all-reduce of the same padded span: CPU tests of the layout
        # only.  World sizes that are not a power of two pad every bucket to NORM_CHUNK x world elements (below), so the span
        # always divides evenly.
        self._ctl = _control_group(process_group) if self.exchange else None      # host-side exchange of the used-parameter bitmap
        self.comm_dtype = torch.float32 if comm_dtype in (None, "auto") else comm_dtype
        self._hot = None          # params some rank has produced a gradient for; None = unknown (step 0)
        self._parena = None       # shard_optimizer: flat f32 parameters laid out like the gradient arena
        self._idle = {}           # hot parameter -> consecutive steps without a gradient on any rank
        self.demote_after = demote_after
        self.stalled_steps = 0    # steps whose hook launches stopped short of the hot buckets
        self._accumulating = False
        self._rebuild = False
        # parameter groups whose gradients the kernels want back to back in a fixed order (encoder layers)
        self._model_units = []
        claimed = set()
        for m in model.modules():
            fn = getattr(m, "grad_arena_units", None)
            if fn is None:
                continue
            for u in fn():
                u = list(u)
                if u and all((p in self.index) and (p not in claimed) and (p not in self.sparse) for p in u):
                    self._model_units.append(u)
                    claimed.update(u)
        if self.shard_optimizer:
            if any(p.dtype != torch.float32 for p in self.params):
                raise ValueError("GradSync(shard_optimizer=True) needs f32 parameters")
            self.exchange = True                       # the shard logic runs at every world size (a world of 1 owns everything)
            if self._ctl is None and dist.is_initialized() and not (self.world > 1 or force_collectives):
                self._ctl = _control_group(process_group)      # (a one-rank group without force_collectives skipped it above)
            self._hot = set(self.params)               # layout of step 0; fixed for good at the end of that step's backward (below)
            self._layout_final = False
        self._build()
        if self.shard_optimizer:
            self._flatten_parameters()
        self._hook_handles = [p.register_post_accumulate_grad_hook(self._hook) for p in self.params]
        engine.set_grad_sink(self)
        self._restore_reserve = None
        if self.world > 1 and backend == "nccl" and engine.WGRAD_RESERVE_CUS == 0:
            # the stack-wide weight-gradient launch (engine.EncoderFn) keeps every CU it gets for 0.6 - 3 ms: leave a few to
            # the collectives' kernels so that buckets already on the wire keep moving meanwhile (a sixteenth of the CUs;
            # close() puts the previous value back: the setting is process-wide)
            self._restore_reserve = engine.WGRAD_RESERVE_CUS
            ncu = torch.cuda.get_device_properties(self.params[0].device).multi_processor_count if self.params and self.params[0].is_cuda else 256
            engine.WGRAD_RESERVE_CUS = max(8, ncu // 16)
        if self.exchange and not GradSync._logged_exchange:
            GradSync._logged_exchange = True
            print("[mvptr] GradSync exchange: wire %s, %d row-sparse table(s), collective %s, world %d" %
                  (str(self.comm_dtype).replace("torch.", ""), len(self.sparse), self.collective, self.world), file=sys.stderr)
        self.zero_grad()

    def close(self):
        """Detach from the model: remove the hooks, drop the gradient views and the engine registration (a second
        GradSync with other options can then be built over the same parameters, bench.py's opt-in leg)."""
        for h in self._hook_handles:
            h.remove()
        self._hook_handles = []
        for p in self.params:
            p.grad = None
            if engine.grad_sink(p) is self:
                engine._SINKS.pop(id(p), None)
        if engine.grad_sink() is self:
            engine._last_sink_ref[0] = None
        if self._restore_reserve is not None:
            engine.WGRAD_RESERVE_CUS = self._restore_reserve
            self._restore_reserve = None

    # ------------------------------------------------------------------ bucket layout
    def _build(self):
        """(Re)build the buckets: hot units in reverse order (early-launchable), then cold ones."""
        self.buckets = []        # dicts: flat, items [(param, offset, numel)], hot, pending, work, ...
        self.where = {}          # param -> bucket index
        self.span = {}           # param -> (offset, numel, address of its view)
        self.unit_at = {}        # first parameter of a multi-parameter unit -> (unit, bucket index, offset, numel)
        claimed = set()
        units = []
        for u in self._model_units:
            units.append(u)
            claimed.update(u)
        units += [[p] for p in self.params if p not in claimed]
        units.sort(key=lambda u: -max(self.index[p] for p in u))
        is_hot = lambda u: self._hot is not None and any(p in self._hot for p in u)   # noqa: E731
        for group, hot in (([u for u in units if is_hot(u)], True), ([u for u in units if not is_hot(u)], False)):
            cur, cur_n = [], 0
            for u in group:
                n_u = sum(p.numel() for p in u)
                if len(u) == 1 and u[0] in self.sparse:      # a bucket of its own, exchanged by rows
                    if cur:
                        self._close(cur, cur_n, hot)
                        cur, cur_n = [], 0
                    self._close([(u[0], 0, n_u)], n_u, hot)
                    self.buckets[-1]["rows_of"] = u[0]
                    continue
                if cur and cur_n + n_u > self.cap:
                    self._close(cur, cur_n, hot)
                    cur, cur_n = [], 0
                if len(u) > 1:
                    self.unit_at[u[0]] = (u, len(self.buckets), cur_n, n_u)
                for p in u:
                    cur.append((p, cur_n, p.numel()))
                    cur_n += p.numel()
            if cur:
                self._close(cur, cur_n, hot)
        self.n_hot = sum(1 for b in self.buckets if b["hot"])
        # one allocation for all buckets, each starting on a multiple of NORM_CHUNK elements (64 KiB): zero_grad() is one
        # fill, and the chunks of the gradient-norm pass never straddle two buckets (per-bucket partial sums, clip_coef)
        total = 0
        # a rank's shard of a bucket = whole norm chunks (sharded optimizer); rs_ag needs spans that divide by the world size
        # (NORM_CHUNK = 2^14 already does for power-of-two worlds)
        odd_world = self.collective == "rs_ag" and (self.world & (self.world - 1)) != 0
        quantum = NORM_CHUNK * (self.world if (self.shard_optimizer or odd_world) else 1)
        for b in self.buckets:
            b["base"] = total
            b["padded"] = (b["n"] + quantum - 1) // quantum * quantum
            total += b["padded"]
        dev = self.params[0].device if self.params else torch.device("cpu")
        self._arena = torch.zeros(total, device=dev, dtype=torch.float32)
        for idx, b in enumerate(self.buckets):
            b["flat"] = self._arena[b["base"]:b["base"] + b["n"]]
            for p, off, k in b["items"]:
                self.where[p] = idx
                self.span[p] = (off, k, b["flat"].data_ptr() + 4 * off)   # O(1) lookups in the per-parameter hook

    def _close(self, items, n, is_hot):
        self.buckets.append(dict(flat=None, n=n, items=items, hot=is_hot, pending=0, work=None, wire=None, streams=set(),
                                 rows_of=None, union=None))

    # ------------------------------------------------------------------ sharded optimizer (ZeRO-1)
    def _flatten_parameters(self):
        """p.data of every parameter becomes a view into a flat arena laid out exactly like the gradient arena: the rank's
        shard of a bucket is then one contiguous range of parameters, gradients and moments, and the updated parameters are
        all-gathered in place."""
        self._parena = torch.zeros_like(self._arena)
        with torch.no_grad():
            for b in self.buckets:
                for p, off, n in b["items"]:
                    view = self._parena[b["base"] + off:b["base"] + off + n].view_as(p)
                    view.copy_(p.data)
                    p.data = view
        self.rank = dist.get_rank(self.group) if dist.is_initialized() else 0

    def _relayout(self, hot):
        """the one re-layout of the sharded mode (end of step 0's backward): new buckets for the hot set `hot`, this step's
        gradients and the parameters carried over"""
        saved = {}
        for b in self.buckets:
            for p, off, n in b["items"]:
                saved[p] = b["flat"][off:off + n].clone()
        streams = set()
        for b in self.buckets:
            streams |= b["streams"]
        self._hot = set(hot)
        self._build()
        self._flatten_parameters()
        for b in self.buckets:
            b["work"] = b["wire"] = b["union"] = None
            b["streams"] = set(streams)
            for p, off, n in b["items"]:
                b["flat"][off:off + n].copy_(saved[p])
                view = b["flat"][off:off + n].view_as(p)
                if p.grad is not None:
                    p.grad = view
            b["pending"] = 0
        self._next = 0
        self._norm_buf = None

    def shard_range(self, b):
        """[lo, hi) of this rank's shard inside bucket b's padded span"""
        s = b["padded"] // self.world
        return self.rank * s, (self.rank + 1) * s

    def gather_parameters(self):
        """All-gather the updated shards of the parameter arena, bucket by bucket, in place."""
        for b in self.buckets:
            span = self._parena[b["base"]:b["base"] + b["padded"]]
            lo, hi = self.shard_range(b)
            if not dist.is_initialized():
                continue
            if self._avg:      # RCCL
                dist.all_gather_into_tensor(span, span[lo:hi], group=self.group)
            else:
                s = hi - lo
                dist.all_gather([span[k * s:(k + 1) * s] for k in range(self.world)], span[lo:hi].clone(), group=self.group)

    def flats(self):
        """The flat f32 gradient buffers (global-norm clipping reads these instead of ~400 tensors): the whole arena
        as one buffer (the alignment gaps between buckets stay zero)."""
        return [self._arena]

    # ------------------------------------------------------------------ global gradient norm
    def _norm_scratch(self):
        n = self._arena.numel() // NORM_CHUNK
        sc = getattr(self, "_norm_buf", None)
        if sc is None or sc.numel() != n + 2 or sc.device != self._arena.device:
            sc = self._norm_buf = torch.zeros(n + 2, device=self._arena.device, dtype=torch.float32)
        return sc

    def _sumsq_span(self, first_chunk, n_chunks):
        """Partial sums of squares of arena chunks [first_chunk, +n_chunks) into their slots (written, not accumulated)."""
        if n_chunks <= 0:
            return
        sc = self._norm_scratch()
        x = self._arena[first_chunk * NORM_CHUNK:(first_chunk + n_chunks) * NORM_CHUNK]
        if x.is_cuda:
            from . import hip
            hip._check(hip.load().mvptr_sumsq_partial(hip._p(x), x.numel(), hip.c_void_p(sc.data_ptr() + 4 * first_chunk), hip._stream()))
        else:
            sc[first_chunk:first_chunk + n_chunks] = (x.view(n_chunks, NORM_CHUNK) ** 2).sum(1)

    def _sumsq_bucket(self, b):
        if self.shard_optimizer:
            # partial sums of THIS rank's reduced shard into the shard's own slots; the other ranks' slots stay zero until the
            # slot vector is summed over the ranks (clip_coef): the same chunk sums in the same slots as the replicated pass
            g = b["gshard"]
            lo, _ = self.shard_range(b)
            first, n_chunks = (b["base"] + lo) // NORM_CHUNK, g.numel() // NORM_CHUNK
            sc = self._norm_scratch()
            if g.is_cuda:
                from . import hip
                hip._check(hip.load().mvptr_sumsq_partial(hip._p(g), g.numel(), hip.c_void_p(sc.data_ptr() + 4 * first), hip._stream()))
            else:
                sc[first:first + n_chunks] = (g.view(n_chunks, NORM_CHUNK) ** 2).sum(1)
            self._norm_done.add(id(b))
            return
        self._sumsq_span(b["base"] // NORM_CHUNK, (b["n"] + NORM_CHUNK - 1) // NORM_CHUNK)
        self._norm_done.add(id(b))

    def clip_coef(self, max_norm):
        """Global 2-norm of all gradients and min(1, max_norm / (norm + 1e-6)) as device scalars -> (norm [1], coef [1])
        (torch.nn.utils.clip_grad_norm_, run_pretrain_ml.py:639-640; the coefficient multiplies the gradients inside the
        fused AdamW).  Buckets whose partial sums finish(want_norm=True) already queued behind their collectives are not
        read again; the rest (everything at world size 1) goes in one pass.  Either way the same chunk sums in the same
        slots, added in slot order: the result does not depend on which path produced a slot."""
        sc = self._norm_scratch()
        n = self._arena.numel() // NORM_CHUNK
        if self.shard_optimizer:
            for b in self.buckets:
                if id(b) not in self._norm_done:
                    self._sumsq_bucket(b)
            if self.world > 1 and not self._norm_summed:
                dist.all_reduce(sc[:n], op=dist.ReduceOp.SUM, group=self.group)      # every slot is non-zero on exactly one rank: exact
            self._norm_summed = True
        elif len(self._norm_done) < len(self.buckets):
            if not self._norm_done:
                self._sumsq_span(0, n)
            else:
                for b in self.buckets:
                    if id(b) not in self._norm_done:
                        self._sumsq_bucket(b)
        norm, coef = sc[n:n + 1], sc[n + 1:n + 2]
        if sc.is_cuda:
            from . import hip
            hip._check(hip.load().mvptr_clip_coef(hip._p(sc), n, float(max_norm), hip._p(norm), hip._p(coef), hip._stream()))
        else:
            nv = sc[:n].double().sum().sqrt().float()
            norm.copy_(nv.reshape(1))
            coef.copy_(torch.clamp(max_norm / (nv + 1e-6), max=1.0).reshape(1))
        return norm, coef

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
        self._arena.zero_()
        for b in self.buckets:
            b["work"] = b["wire"] = b["union"] = None
            b["streams"] = set()
            for p, off, n in b["items"]:
                view = b["flat"][off:off + n].view_as(p)
                if p.grad is None or p.grad.data_ptr() != view.data_ptr():
                    p.grad = view
            b["pending"] = len(b["items"])
        self._ready = set()       # parameters counted towards their bucket's readiness in the exchanging backward
        self._touched = set()     # parameters that received a gradient in ANY backward of this step
        self._uses, self._done = {}, {}   # direct delivery: uses noted in forward passes / deliveries so far
        self._hook_skip = {}              # parameter whose next post-accumulate hook repeats a completed direct delivery -> None (or, check_mixed_use, a clone of its .grad then)
        self._rows = {}
        self._early_union = {}
        self._next = 0            # buckets [0, _next) have been launched this step
        self._norm_done = set()   # buckets whose partial sums of squares are in their slots (finish(want_norm=True))
        self._norm_summed = False
        if self.shard_optimizer and getattr(self, "_norm_buf", None) is not None:
            self._norm_buf.zero_()  # the other ranks' slots must read zero before the slot vector is summed

    def note_rows(self, param, ids):
        """Tell the exchange which rows of a row-sparse parameter this rank's step looks up (every id
        tensor that indexes the table: call once per tensor or pass a list), BEFORE the backward pass
        (a hot bucket is launched from the hook of its last gradient).  Without it the bucket is
        reduced densely.  Every gradient row of the table must come from the noted lookups."""
        if not self.exchange or param not in self.sparse:
            return
        ids = ids if isinstance(ids, (list, tuple)) else [ids]
        self._rows.setdefault(param, []).extend(t.reshape(-1) for t in ids if t is not None)

    def exchange_rows_early(self, param, host_ids):
        """The row union of this step formed AHEAD of the backward pass, host to host (round 6).  `host_ids`: CPU int64 tensor(s) with
        the ids this rank's shard looks up in `param` — input data, known where the batch is built (synthetic.host_counts puts the
        unique ids into batch["word_rows"], synthetic.word_rows).  The ranks exchange them over the gloo control group while the device is
        still busy with earlier work, every rank forms the same sorted union, and the union goes to the device as one pinned,
        asynchronous copy: the launch of the sparse bucket then finds it ready — no all-gather on the data communicator, no device
        read-back and no host wait inside the launch (_row_union's late form has two of each).  COLLECTIVE: every rank calls it in
        the same steps (same program, same batch layout); a step without it takes the late form."""
        if not self.exchange or param not in self.sparse:
            return
        ids = host_ids if isinstance(host_ids, (list, tuple)) else [host_ids]
        ids = [t.reshape(-1) for t in ids if t is not None]
        if any(t.is_cuda for t in ids):
            raise ValueError("GradSync.exchange_rows_early takes HOST ids (the device ids go to note_rows)")
        mine = torch.unique(torch.cat(ids).to(torch.int64)) if ids else torch.empty(0, dtype=torch.int64)
        rows = param.shape[0]
        if mine.numel() and (int(mine.min()) < 0 or int(mine.max()) >= rows):
            raise RuntimeError("GradSync.exchange_rows_early: row id outside the table")
        if dist.is_initialized() and self.world > 1:
            cnt = torch.tensor([mine.numel()], dtype=torch.int64)
            cnts = [torch.empty_like(cnt) for _ in range(self.world)]
            dist.all_gather(cnts, cnt, group=self._ctl)
            cnts = [int(c) for c in cnts]
            pad = torch.full((max(1, max(cnts)),), -1, dtype=torch.int64)
            pad[:mine.numel()] = mine
            parts = [torch.empty_like(pad) for _ in range(self.world)]
            dist.all_gather(parts, pad, group=self._ctl)
            union = torch.unique(torch.cat([q[:c] for q, c in zip(parts, cnts)]))
        else:
            union = mine
        if union.numel() == 0 or union.numel() * 2 > rows:
            self._early_union[param] = False            # nothing to gain: dense exchange (same decision on every rank)
            return
        dev = param.device
        if dev.type == "cuda":
            union = union.pin_memory().to(dev, non_blocking=True)
        self._early_union[param] = union

    def _row_union(self, b):
        """All ranks' unique row ids of this step -> the sorted union (identical on every rank), or
        None when some rank has no ids noted (dense fallback, decided from the gathered data alone)."""
        p = b["rows_of"]
        early = self._early_union.pop(p, None)
        if early is not None:
            return None if early is False else early
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

    # ---- engine protocol: kernels that accumulate straight into the arena -----------------------------
    def direct(self, p):
        """The arena view to accumulate p's gradient into (p.grad itself), or None when p is not laid out here."""
        sp = self.span.get(p)
        g = p.grad
        if sp is None or g is None or g.data_ptr() != sp[2]:
            return None
        return g

    def arena(self, params):
        """Flat f32 view holding the gradients of `params` back to back in exactly that order, or None."""
        ent = self.unit_at.get(params[0])
        if ent is None:
            return None
        u, idx, off, n = ent
        if len(u) != len(params) or any(a is not b for a, b in zip(u, params)):
            return None
        if any(self.direct(p) is None for p in params):
            return None
        return self.buckets[idx]["flat"][off:off + n]

    def note_use(self, p):
        """Forward pass: an autograd function that may deliver p's gradient directly has used p.  A parameter used
        by several function calls of one graph (the word table: text and tag lookups; the embedding LayerNorm) is
        ready for the exchange only when ALL of them have delivered — autograd's AccumulateGrad node gives that for
        free (it runs once, after every contribution has arrived), direct delivery has to count."""
        if self.exchange:
            self._uses[p] = self._uses.get(p, 0) + 1

    def delivered(self, p):
        """A kernel has queued p's gradient into the arena on the current stream (the engine's replacement for
        the AccumulateGrad node + hook of a gradient that autograd never sees)."""
        if self.exchange:
            d = self._done.get(p, 0) + 1
            self._done[p] = d
            if d < self._uses.get(p, 0):
                self._note(p, final=False)
                return
            # this torch runs the parameter's AccumulateGrad node (and so its post-accumulate hook) even when every
            # function returned None for it: that one call is this delivery seen again, not a new gradient.  INVARIANT:
            # within one graph a parameter is used EITHER through engine functions (counted by note_use) OR through
            # plain torch ops, never both — a torch-side contribution would arrive after the bucket may have been
            # launched and its hook would be taken for the repeat.  Every model of this package keeps to it (tied
            # weights go through engine functions only); `check_mixed_use=True` verifies it by value for new heads
            # (a clone per delivery: debugging only — version counters cannot tell, all views of the arena share one).
            self._hook_skip[p] = p.grad.detach().clone() if self.check_mixed_use else None
        self._note(p)

    # ---------------------------------------------------------------------------------------------------
    def _hook(self, p):
        if p in self._hook_skip:
            snap = self._hook_skip.pop(p)
            if snap is not None and p.grad is not None and not torch.equal(snap, p.grad):
                raise RuntimeError("GradSync: a parameter received a gradient through a torch op AND through the HIP engine in "
                                   "one graph; its bucket may already be in flight.  Use one path per parameter and graph.")
            return
        off, n, ptr = self.span[p]
        if p.grad.data_ptr() != ptr:
            # autograd replaced the view (e.g. dtype change): copy into the bucket, re-attach
            b = self.buckets[self.where[p]]
            b["flat"][off:off + n].copy_(p.grad.reshape(-1))
            p.grad = b["flat"][off:off + n].view_as(p)
        self._note(p)

    def _note(self, p, final=True):
        self._touched.add(p)      # also under no_sync(): a head used only in the accumulation micro-batches is "used"
        if not self.exchange:
            return
        idx = self.where[p]
        b = self.buckets[idx]
        if p.grad.is_cuda:
            # the gradient was produced on the stream current here (sub-networks may run on a side
            # stream, engine.side_stream): remember which streams fed this bucket
            b["streams"].add(torch.cuda.current_stream(p.grad.device))
        if idx < self._next and not self._accumulating:
            raise RuntimeError("GradSync: a gradient arrived for a bucket that is already being reduced — run one "
                               "backward per zero_grad(), or wrap all but the last backward in `with sync.no_sync():`")
        if self._accumulating or not final:
            return
        if idx < self._next:
            raise RuntimeError("GradSync: a gradient arrived for a bucket that is already being reduced — run one "
                               "backward per zero_grad(), or wrap all but the last backward in `with sync.no_sync():`")
        if p in self._ready:
            return                # the same parameter used twice in one graph fires once; be tolerant
        self._ready.add(p)
        b["pending"] -= 1
        if self.overlap and b["hot"] and (not self.shard_optimizer or self._layout_final):
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
        if self.shard_optimizer:
            # reduce-scatter only: this rank keeps 1 / world of the bucket's (chunk-padded) span; no gradient all-gather follows —
            # the optimizer runs on the shard and the updated parameters are gathered instead (optimization.ShardedAdamW)
            span = self._arena[b["base"]:b["base"] + b["padded"]]
            wire = span if self.comm_dtype == torch.float32 else span.to(self.comm_dtype)
            b["wire"] = wire
            if not dist.is_initialized():          # no process group: the rank owns the whole span
                b["shard"], b["work"] = wire, None
            elif self._avg:                         # RCCL (also a one-rank group: the real collectives run)
                b["shard"] = torch.empty(wire.numel() // self.world, device=wire.device, dtype=wire.dtype)
                b["work"] = dist.reduce_scatter_tensor(b["shard"], wire, op=op, group=self.group, async_op=True)
            else:                     # gloo has no reduce_scatter_tensor: all-reduce, keep the own slice (tests on CPU)
                b["shard"] = None
                b["work"] = dist.all_reduce(wire, op=op, group=self.group, async_op=True)
            self._next = idx + 1
            return
        if self.collective == "rs_ag" and src is b["flat"]:
            # reduce-scatter + all-gather over the bucket's chunk-padded span (the padding is zero and stays zero): every
            # rank owns 1 / world of it between the two collectives
            span = self._arena[b["base"]:b["base"] + b["padded"]]
            assert span.numel() % self.world == 0
            wire = span if self.comm_dtype == torch.float32 else span.to(self.comm_dtype)
            b["wire"] = wire
            if not self._avg:        # gloo: the same span as ONE all-reduce (emulation, see __init__)
                b["work"] = dist.all_reduce(wire, op=op, group=self.group, async_op=True)
                self._next = idx + 1
                return
            shard = torch.empty(wire.numel() // self.world, device=wire.device, dtype=wire.dtype)
            dist.reduce_scatter_tensor(shard, wire, op=op, group=self.group, async_op=True)
            b["shard"] = shard      # kept alive until the all-gather has run
            b["work"] = dist.all_gather_into_tensor(wire, shard, group=self.group, async_op=True)   # same communicator: ordered behind the reduce-scatter
            self._next = idx + 1
            return
        wire = src if (self.comm_dtype == torch.float32 and src is b["flat"]) else src.to(self.comm_dtype)
        b["wire"] = wire
        b["work"] = dist.all_reduce(wire, op=op, group=self.group, async_op=True)
        self._next = idx + 1

    def __call__(self, want_norm=False):
        """Finish the step's exchange: launch the buckets that are still waiting (in index order on
        every rank), wait for all of them, scale if the backend summed.  Parameters that received no
        gradient (on any rank) end with grad = None, as under DDP find_unused_parameters
        (run_pretrain_ml.py:415-418) and as without an arena: the optimizer skips them.
        want_norm: the caller will clip by the global norm (clip_coef): queue every bucket's partial sums of squares
        right behind its collective, so that only the coefficient kernel is left after the last bucket."""
        if not self.exchange:
            for p in self.params:
                if p not in self._touched:
                    p.grad = None
            return
        used = None
        if self.shard_optimizer and not self._layout_final:
            # Step 0 of the sharded mode: nothing has gone out yet (the hooks hold back while the layout may still move).  The
            # parameters that produced a gradient on SOME rank become the hot buckets, the never-used ones (qa_head without
            # answers, ...) go behind them — the one re-layout of this mode: gradients and parameters move inside the rank, no
            # optimizer state exists yet.  From here on the layout — and with it every shard — is fixed.
            used = torch.tensor([1 if p in self._touched else 0 for p in self.params], dtype=torch.int32)
            if dist.is_initialized():
                dist.all_reduce(used, op=dist.ReduceOp.MAX, group=self._ctl)
            self._layout_final = True
            hot = {p for p, u in zip(self.params, used.tolist()) if u}
            if hot != self._hot:
                self._relayout(hot)
        if self._hot is not None and self._next < self.n_hot:
            self.stalled_steps += 1       # some hot bucket never became ready from the hooks: no overlap behind it
        for idx in range(self._next, len(self.buckets)):
            self._launch(idx)
        # which parameters produced a gradient on ANY rank (DDP's used-parameter bitmap): those keep
        # the averaged gradient on every rank, the others keep grad = None everywhere, so replicas
        # apply identical updates even when a shard skipped a head
        # (host to host over the control group: the device and its queue of collectives are not involved, nothing here waits
        # for the backward pass)
        if used is None:
            used = torch.tensor([1 if p in self._touched else 0 for p in self.params], dtype=torch.int32)
            if dist.is_initialized():      # (_ctl None = the default group, itself gloo)
                dist.all_reduce(used, op=dist.ReduceOp.MAX, group=self._ctl)
        if self.shard_optimizer:
            for b in self.buckets:
                if b["work"] is not None:
                    b["work"].wait()
                lo, hi = self.shard_range(b)
                sh = b["shard"] if b["shard"] is not None else b["wire"][lo:hi]
                if sh.is_cuda and sh is not b["wire"]:
                    sh.record_stream(torch.cuda.current_stream(sh.device))
                g = sh if sh.dtype == torch.float32 else sh.to(torch.float32)     # (an f32 slice of the arena itself is fine: it is read before zero_grad())
                if self.world > 1 and not self._avg:
                    g.mul_(1.0 / self.world)
                b["gshard"] = g           # the rank's reduced, averaged gradient shard (f32) until zero_grad()
                b["wire"] = b["shard"] = None
                if want_norm:
                    self._sumsq_bucket(b)
            used = used.tolist()
            for p, u in zip(self.params, used):
                if not u:
                    p.grad = None
            return
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
                if want_norm:
                    self._sumsq_bucket(b)
                continue
            if b["wire"] is not b["flat"] and b["wire"].data_ptr() != b["flat"].data_ptr():
                if b["wire"].is_cuda:
                    b["wire"].record_stream(torch.cuda.current_stream(b["wire"].device))
                b["flat"].copy_(b["wire"][:b["n"]])          # bf16 -> f32, ordered after the collective by wait()
            if not self._avg:
                b["flat"].mul_(1.0 / self.world)
            b["wire"] = b["shard"] = None
            if want_norm:
                self._sumsq_bucket(b)         # behind THIS bucket's collective; the later buckets are still on the wire
        used = used.tolist()
        for p, u in zip(self.params, used):
            if not u:
                p.grad = None
        # hot set = every parameter that has produced a gradient on any rank and has not been idle for
        # `demote_after` steps since (identical on all ranks: derived from the all-reduced bitmap only).
        # One that is missing in some step only delays launches to finish(); a change moves the bucket
        # layout at the next zero_grad().
        now = {p for p, u in zip(self.params, used) if u}
        hot = set(now) if self._hot is None else (self._hot | now)
        for p in list(hot):
            if p in now:
                self._idle[p] = 0
            else:
                self._idle[p] = self._idle.get(p, 0) + 1
                if self._idle[p] >= self.demote_after:
                    hot.discard(p)
                    self._idle.pop(p, None)
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
