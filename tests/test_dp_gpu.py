"""Two data-parallel ranks sharing the one GPU of the test box (gloo transport, the same hook /
bucket-view code path the RCCL run uses): the HIP training step with overlapped gradient exchange
keeps the replicas bit-identical and matches a single-process step on the concatenated batch's
averaged gradients."""
import os

import pytest
import torch

import golden_util as gu

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, q, sparse):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mvp_pytorch_amd import dp, modeling, train
    from mvp_pytorch_amd.synthetic import synthetic_batch
    dev = torch.device("cuda:0")
    cfg = dict(gu.TINY_CFG, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    torch.manual_seed(0)
    model = modeling.BiBertImgForPreTraining(modeling.make_config(cfg)).to(dev)
    model.train()
    opt, sched = train.build_optimizer(model, lr=1e-3, t_total=10)
    sparse_rows = [model.bert.embeddings.word_embeddings.weight] if sparse else []
    sync = dp.GradSync(model, bucket_mb=0.25, sparse_rows=sparse_rows)
    dims = dict(B=4, T=12, P=3, G=6, R=5)
    losses = []
    for step in range(3):
        batch = synthetic_batch(dims, cfg, 100 + 10 * step + rank, device=dev)
        torch.manual_seed(step)  # same hard-negative permutation draw on both ranks
        out = train.pretrain_step(model, batch, opt, sched, max_tag_length=dims["G"], grad_sync=sync, return_losses=True)
        losses.append(float(out[0]))
    torch.cuda.synchronize()
    probe = {n: p.detach().float().cpu().numpy() for n, p in model.named_parameters()
             if n in ("bert.txt_encoder.layer.0.attention.self.query.weight", "bert.embeddings.word_embeddings.weight",
                      "cls.predictions.decoder.weight", "logit_scale", "qa_head.weight")}
    n_sparse = sum(1 for b in sync.buckets if b["rows_of"] is not None)
    q.put((rank, losses, probe, len(sync.buckets), n_sparse))
    dist.destroy_process_group()


def _run(sparse):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + (os.getpid() % 1000) + (11 if sparse else 0)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, sparse)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in range(2)], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return res


def test_two_rank_training_step_keeps_replicas_identical(dev):
    """Dense exchange, then the row-sparse exchange of the word-embedding gradient (GradSync.note_rows from
    train.pretrain_step): replicas identical in both, and both modes give the same parameters."""
    import numpy as np
    probes = []
    for sparse in (False, True):
        (_, l0, p0, nb, ns), (_, l1, p1, _, _) = _run(sparse)
        assert nb > 1 and ns == (1 if sparse else 0)
        assert all(np.isfinite(l0)) and all(np.isfinite(l1))
        for k in p0:
            assert np.array_equal(p0[k], p1[k]), k   # same averaged gradients -> identical replicas
        print("two-rank losses", "sparse" if sparse else "dense", l0, l1)
        probes.append((l0, p0))
    (ld, pd_), (ls, ps) = probes
    assert ld == ls
    for k in pd_:
        assert np.array_equal(pd_[k], ps[k]), k     # rows outside the union are zero on both ranks: same sums


def test_bench_launcher_two_ranks_on_one_gpu(dev):
    """`python bench.py --gpus 2` as the driver starts it without a launcher: two rank processes, one JSON line from rank 0
    with n_gpus = 2 and the whole-job rate.  The test box has one GPU, so both ranks are pinned to it and talk through gloo
    (MVPTR_BENCH_DEVICE / MVPTR_DIST_BACKEND: the hooks bench.py has for exactly this); on a multi-GPU node the same code path
    runs one rank per GPU over RCCL."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MVPTR_BENCH_DEVICE="0", MVPTR_DIST_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    o = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "16",
                        "--no-extras"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert o.returncode == 0, o.stderr.decode()[-2000:]
    line = json.loads(o.stdout.decode().strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["warmup"] == 1
    assert line["config"]["parallelism"] == "dp2" and line["config"]["global_batch"] == 32
    assert line["scaling"] == "weak" and line["value"] > 0 and line["ms_per_step"] > 0
    assert abs(line["value"] - 32 / (line["ms_per_step"] * 1e-3)) < 0.02 * line["value"]
