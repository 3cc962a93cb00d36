"""Data-parallel gradient exchange: one process per GPU, image-text pairs sharded across ranks,
gradients averaged with bucketed all-reduce on RCCL (backend 'nccl' on ROCm) over xGMI.

Replaces what DeepSpeed ZeRO-2 / DDP did implicitly for the reference
(oscar/run_pretrain_ml.py:406-418; oscar/tmp_config.json:11-20).  The in-batch contrastive /
hard-negative step stays rank-local exactly as in the reference (no feature all-gather,
modeling_vlbert.py:525-534), so the gradient all-reduce is the only per-step collective.

Buckets are flat f32 buffers of >= `bucket_mb` MiB filled in reverse parameter order (the order
gradients become final in backward); each bucket is reduced asynchronously on RCCL's stream and
the averaged values are copied back into the .grad tensors.  Parameters that received no
gradient (e.g. qa_head when qa_ans is None, modeling_vlbert.py:1184) contribute zeros so every
rank issues identical collectives.
"""
import torch
import torch.distributed as dist


class GradSync:
    def __init__(self, model, bucket_mb=64, process_group=None):
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.params = [p for p in model.parameters() if p.requires_grad]
        self.buckets = []  # list of (flat buffer, [(param, offset, numel)])
        cap = int(bucket_mb * (1 << 20) // 4)
        cur, cur_n = [], 0
        for p in reversed(self.params):
            if cur and cur_n + p.numel() > cap:
                self._close(cur, cur_n)
                cur, cur_n = [], 0
            cur.append((p, cur_n, p.numel()))
            cur_n += p.numel()
        if cur:
            self._close(cur, cur_n)

    def _close(self, items, n):
        p0 = items[0][0]
        self.buckets.append((torch.zeros(n, device=p0.device, dtype=torch.float32), items))

    def __call__(self):
        if self.world == 1:
            return
        works = []
        for flat, items in self.buckets:
            for p, off, n in items:
                if p.grad is None:
                    flat[off:off + n].zero_()
                else:
                    flat[off:off + n].copy_(p.grad.reshape(-1))
            works.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        inv = 1.0 / self.world
        for (flat, items), w in zip(self.buckets, works):
            w.wait()
            flat.mul_(inv)
            for p, off, n in items:
                if p.grad is None:
                    p.grad = flat[off:off + n].view_as(p).clone()
                else:
                    p.grad.copy_(flat[off:off + n].view_as(p))


def all_reduce_metrics(values, device):
    """The reference's only direct collective: a 3-float all_reduce of [loss, n_examples, n_steps]
    at checkpoint time (run_pretrain_ml.py:688-689)."""
    t = torch.tensor(values, dtype=torch.float32, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t)
    return t.tolist()
