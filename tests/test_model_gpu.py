"""Model-level parity on MI355X: the HIP path (through the oscar.modeling-style classes and the
C ABI) against the CPU oracle and the reference's golden vectors, same inputs, dropout off.

Tolerances: the kernels compute in bf16 with f32 accumulation, the oracle in f32.
  losses            1e-3 relative (BASELINE.json north_star) for the total and the token-averaged
                    losses; the two losses computed from <= 2B pooled rows (4x4 contrastive logits,
                    8 ITM rows at B=4) get 3e-3 at B=4 and 1e-3 at B=64 (error ~ 1/sqrt(rows))
  ITM labels        bit-exact; hard-negative indices bit-exact wherever the f32 top-2 margin
                    exceeds the bf16 noise of sim_mat (rows closer than that are reported)
  sim_mat           5e-3 absolute
  gradients         3e-2 relative L2 per tensor (bf16 activations and gradients); the tensors of
                    the contrastive branch (txt_proj, vis_proj, logit_scale) get 1e-1: their
                    gradient multiplies the bf16 noise of sim_mat by exp(logit_scale) ~ 14.  In the
                    tiny fixture the 4 global embeddings are almost parallel (f32 top-2 margin
                    1.4e-3), so d(loss)/d(proj) = sum_j (p_j - y_j) g_j cancels down to the 1e-3
                    differences between them: ill-conditioned in ANY 8-bit-mantissa arithmetic, and
                    reported, not asserted, there (asserted on the BERT-base fixture)
"""
import random

import numpy as np
import pytest
import torch

import golden_util as gu
from oracle import mvptr_oracle as orc

pytestmark = pytest.mark.gpu

LOSS_RTOL = 1e-3
CLIP_BRANCH = ("bert.txt_proj", "bert.vis_proj", "logit_scale")
SMALL_ROWS_RTOL = 3e-3  # retrieval / ITM losses at B=4 (see module docstring)

# Per-element quantities (logits, gradient norms) cannot meet 1e-3 in bf16 (every rounding of a GEMM operand adds ~1.1e-3
# relative error, a 12-layer stack has ~60 of them): each is bounded by TWICE the value measured on MI355X in round 3
# (gpurun_out/r03f/parity_values.txt, r03g), not by a blanket tolerance.  relative L2 unless stated.
MEASURED = {
    "tiny_single_pretrain:prediction_scores": 0.0083, "tiny_single_pretrain:seq_relationship": 0.0241,
    "cfg1_single_pretrain:prediction_scores": 0.0118, "cfg1_single_pretrain:seq_relationship": 0.0079,
    "tiny_bi_pretrain:gnorm": 0.0272, "cfg1_bi_pretrain:gnorm": 0.0229, "tiny_bi_hn:gnorm": 0.0366,
    "tiny_bi_pretrain:grad": 0.0191, "cfg1_bi_pretrain:grad": 0.0191,
    "tiny_single_pretrain:gnorm": 0.0050, "tiny_single_pretrain:gnorm_loss_only": 0.0050,
    "cfg1_single_pretrain:gnorm": 0.0246, "cfg1_single_pretrain:gnorm_loss_only": 0.0246,
    "tiny_finetune:ret_fine_logits": 0.0041, "tiny_finetune:ve_logits": 0.0052, "tiny_finetune:vqa_logits": 0.0062,
    # round 4 (gpurun_out/r04k/parity_values.txt, profiles/r04_parity_values.txt): the row-packed TRAINING path against the
    # reference's 5-tuple fixtures, 8-bit gelu' stash and sync-free joint pass included
    "tiny_bi_pretrain_nophrase:train_gnorm": 0.0040, "tiny_bi_pretrain_nophrase:train_grad": 0.0178,
    "cfg1_bi_pretrain_nophrase:train_gnorm": 0.0228, "cfg1_bi_pretrain_nophrase:train_grad": 0.0273,
    # round 5 (profiles/r05_parity_values.txt): backward pass at the configs[4] shape against the oracle (6-question subset)
    "configs4:vqa_grad": 0.0194,
}


# Absolute ceilings per class of comparison ("twice what we measured" is self-referential — a slow drift of the
# measurements would move the bounds with it).  A BERT-base pass rounds ~60 GEMM operands to bf16 (2^-9 relative each, 1.1e-3 rms):
# added in quadrature that is 0.9e-2, a few times more where a loss sits on a few rows or a gradient is a sum of cancelling terms.
# No tagged comparison may exceed these, whatever MEASURED says.
CEILING = {"logits": 0.04, "grad": 0.06}


def check_measured(tag, value, fallback):
    """value < min(2 x the recorded measurement of `tag`, the absolute ceiling of its class) (fallback bound for a tag measured
    for the first time)."""
    bound = 2.0 * MEASURED[tag] if tag in MEASURED else fallback
    kind = "grad" if ("grad" in tag or "gnorm" in tag) else "logits"
    bound = min(bound, CEILING[kind])
    print("PARITY %s %.5f (bound %.5f)" % (tag, value, bound))
    assert value < bound, (tag, value, bound)


class Replay:
    """Feed recorded reference draws to the product's torch.randperm / torch.randint / random.choice."""

    def __init__(self, d, dev):
        self.perm = [torch.as_tensor(x) for x in d.get("draw_randperm", [])]
        sizes = d["draw_randint3_sizes"].tolist() if "draw_randint3_sizes" in d else []
        flat = d.get("draw_randint3", np.zeros(0, np.int64))
        self.ints, o = [], 0
        for s in sizes:
            self.ints.append(torch.as_tensor(flat[o:o + s]))
            o += s
        self.choice = d["draw_choice"].tolist() if "draw_choice" in d else []
        self.multi = [torch.as_tensor(x) for x in d.get("draw_multinomial", [])]
        self.dev = dev

    def __enter__(self):
        self._orig = (torch.randperm, torch.randint, random.choice)
        self._orig_multi = torch.multinomial
        o_int = self._orig[1]
        o_multi = self._orig_multi

        def multi(probs, num_samples=1, *a, **k):
            if self.multi:
                return self.multi.pop(0).to(self.dev).view(-1, 1)
            return o_multi(probs, num_samples, *a, **k)

        torch.multinomial = multi

        def perm(n, *a, **k):
            return self.perm.pop(0).to(self.dev)

        def rint(lo, hi, size, *a, **k):
            if hi == 3 and self.ints:
                return self.ints.pop(0).to(self.dev)
            return o_int(lo, hi, size, *a, **k)

        def choice(seq):
            return self.choice.pop(0)

        torch.randperm, torch.randint, random.choice = perm, rint, choice
        return self

    def __exit__(self, *exc):
        torch.randperm, torch.randint, random.choice = self._orig
        torch.multinomial = self._orig_multi


def _rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / (b.norm() + 1e-12)).item()


def _build(cls_name, cfg, seed, dev, train=False, gain=1.0):
    from mvp_pytorch_amd import modeling
    cfg = dict(cfg, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    model = getattr(modeling, cls_name)(modeling.make_config(cfg))
    sd = {k: torch.from_numpy(v) for k, v in gu.det_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed, gain).items()}
    model.load_state_dict(sd)
    model.to(dev)
    model.train(train)
    if hasattr(model, "wra_on_device"):
        model.wra_on_device = False  # parity runs replay the reference's host draws
    return model, sd


def _bi_inputs(d, dev):
    t = lambda k: torch.from_numpy(d["in:" + k]).to(dev)  # noqa: E731
    return dict(input_ids_a=t("input_ids_a"), token_type_ids_a=t("segment_ids_a"), attention_mask_a=t("input_mask_a"),
                input_ids_b=t("input_ids_b"), token_type_ids_b=t("segment_ids_b"), attention_mask_b=t("input_mask_b"),
                img_feats=t("img_feats"))


@pytest.mark.parametrize("name", ["tiny_bi_pretrain", "cfg1_bi_pretrain", "tiny_bi_hn"])
def test_bi_pretrain_parity(dev, name):
    d = gu.load(name)
    cfg, dims = d["config"], d["dims"]
    model, sd = _build("BiBertImgForPreTraining", cfg, int(d["seed"]), dev, gain=float(d["weight_gain"]))
    kw = _bi_inputs(d, dev)
    t = lambda k: torch.from_numpy(d["in:" + k]).to(dev)  # noqa: E731
    # 1) free-running hard-negative mining: indices vs golden where the margin allows
    with torch.no_grad(), Replay(d, dev):
        outs, single, hard = model.bert(max_tag_length=dims["G"], encode_hn=True, **kw)
    sim = single[2].float().cpu()
    sim_ref = torch.from_numpy(d["sim_mat"])
    sim_err = (sim - sim_ref).abs().max().item()
    print(name, "sim_mat max abs err", sim_err, "golden argmax margin", float(d["argmax_margin"]))
    assert sim_err < 5e-3
    same_t = np.array_equal(hard[0].cpu().numpy(), d["hard_txt_index"])
    same_i = np.array_equal(hard[1].cpu().numpy(), d["hard_img_index"])
    print(name, "hard indices equal:", same_t, same_i)
    free_running = name in ("tiny_bi_hn", "cfg1_bi_pretrain")
    if free_running:
        # the hard-negative fixture and the BERT-base fixture (input batch chosen among 950 seeds for its margin,
        # tools/gen_golden.py): every f32 top-2 margin is >= 10x the bf16 error of sim_mat, so the integer outputs
        # of vl:531-566 must be bit-exact, unconditionally
        assert float(d["argmax_margin"]) > 10 * sim_err, (float(d["argmax_margin"]), sim_err)
        assert same_t and same_i
    elif float(d["argmax_margin"]) > 4 * sim_err:
        assert same_t and same_i
    # 2) losses + gradients with the reference's captured draws; the hard batch is the reference's
    #    (free-running — nothing injected — on the hard-negative fixture and on the BERT-base fixture: the model mines
    #    its own hard negatives from its own bf16 sim_mat and must land on the reference's losses)
    n = sim_ref.shape[0]
    masked = sim_ref - 2 * torch.eye(n)
    import contextlib
    inject = contextlib.nullcontext() if free_running else gu.InjectHard(model.bert, masked.max(1)[1], masked.max(0)[1])
    with Replay(d, dev), inject:
        res = model(masked_lm_labels_a=t("lm_label_ids_a"), masked_lm_labels_b=t("lm_label_ids_b"),
                    max_tag_length=dims["G"], img_index=t("image_index"), phrase_index=t("phrase_index"), **kw)
    got = np.array([x.item() for x in res])
    ref = d["losses"]
    rel = np.abs(got - ref) / np.abs(ref)
    print(name, "losses", got, "ref", ref, "rel", rel)
    assert len(res) == 6
    # total, masked-concept, MLM: north_star's 1e-3.  The hard-negative fixture exists for the INTEGER
    # outputs asserted above; its 5x weight gain amplifies every bf16 rounding by the same factor
    # (measured 1.2e-3 .. 2.5e-3 on these losses, 1.2e-2 on the 4-row ITM loss), so its losses are only
    # sanity-checked at 5x the tolerance — the 1e-3 loss parity is asserted on the gain-1 fixtures.
    hn = name == "tiny_bi_hn"
    assert max(rel[0], rel[1], rel[3]) < (5 * LOSS_RTOL if hn else LOSS_RTOL), rel
    assert max(rel[2], rel[4]) < (3e-2 if hn else SMALL_ROWS_RTOL), rel   # contrastive, ITM (B=4 rows)
    assert abs(got[5] - ref[5]) < 2e-3 + LOSS_RTOL * abs(ref[5])  # WRA hinge (small value, clamp)
    res[0].backward()
    worst = ("", 0.0)
    for pname, p in model.named_parameters():
        key = "gnorm:" + pname
        if key not in d:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, pname
            continue
        gn = p.grad.double().norm().item()
        rn = float(d[key])
        if pname == "logit_scale":  # scalar, sum of cancelling terms: absolute tolerance
            assert abs(gn - rn) < 5e-3, (pname, gn, rn)
        elif rn > 1e-6:
            e = abs(gn - rn) / rn
            if e > worst[1]:
                worst = (pname, e)
            if pname in CLIP_BRANCH and name.startswith("tiny"):
                print("   (ill-conditioned, reported only) grad-norm", pname, e)
            else:
                assert e < (1e-1 if pname in CLIP_BRANCH else 2.0 * MEASURED.get(name + ":gnorm", 2.5e-2)), (pname, gn, rn)
        else:
            # analytically zero gradient (key bias: softmax is invariant to it); the reference holds
            # f32 rounding noise there, the bf16 path bf16 rounding noise: absolute bound only
            # (the bound scales with the fixture's weight gain, which amplifies every rounding)
            print("   (zero in exact arithmetic) grad-norm", pname, gn, "ref", rn)
            assert gn < 2e-3 * max(1.0, float(d["weight_gain"])), (pname, gn, rn)
            continue
        full = "grad:" + pname
        if full in d:
            e = _rel(p.grad, torch.from_numpy(d[full]))
            print("   grad", pname, "rel L2", e)
            if not (pname in CLIP_BRANCH and name.startswith("tiny")):
                assert e < (1e-1 if pname in CLIP_BRANCH else 2.0 * MEASURED.get(name + ":grad", 1.5e-2)) or pname == "logit_scale", (pname, e)
    print(name, "worst grad-norm error", worst)


@pytest.mark.parametrize("name", ["tiny_single_pretrain", "cfg1_single_pretrain"])
def test_single_pretrain_parity(dev, name):
    d = gu.load(name)
    cfg = d["config"]
    model, sd = _build("BertImgForPreTraining", cfg, int(d["seed"]), dev)
    with torch.no_grad():  # reference ties decoder to the embeddings: value of the decoder key wins
        model.bert.embeddings.word_embeddings.weight.copy_(sd["cls.predictions.decoder.weight"])
        model.tie_weights()
    t = lambda k: torch.from_numpy(d["in:" + k]).to(dev)  # noqa: E731
    out = model(t("input_ids"), t("segment_ids"), t("input_mask"), t("lm_label_ids"), t("is_next"), img_feats=t("img_feats"))
    got = np.array([out[0].item(), out[3].item()])
    rel = np.abs(got - d["losses"]) / np.abs(d["losses"])
    print(name, "losses", got, d["losses"], rel)
    assert rel.max() < LOSS_RTOL
    e = _rel(out[1][..., :64], torch.from_numpy(d["prediction_scores_head"]))
    e2 = _rel(out[2], torch.from_numpy(d["seq_relationship_score"]))
    print(name, "prediction_scores rel L2", e, "seq_relationship rel L2", e2)
    check_measured(name + ":prediction_scores", e, 2e-2)
    check_measured(name + ":seq_relationship", e2, 5e-2)      # 4x2 ITM logits of small magnitude
    out[0].backward()
    worst = 0.0
    for pname, p in model.named_parameters():
        key = "gnorm:" + pname
        if key in d and float(d[key]) > 1e-6 and p.grad is not None:
            worst = max(worst, abs(p.grad.double().norm().item() - float(d[key])) / float(d[key]))
    check_measured(name + ":gnorm", worst, 5e-2)


@pytest.mark.parametrize("name", ["tiny_single_pretrain", "cfg1_single_pretrain"])
def test_single_pretrain_loss_only_path(dev, name):
    """return_prediction_scores = False: scored rows only, fused decoder + cross entropy (no logits
    tensor); same losses and gradient norms as the reference fixture."""
    d = gu.load(name)
    cfg = d["config"]
    model, sd = _build("BertImgForPreTraining", cfg, int(d["seed"]), dev)
    with torch.no_grad():
        model.bert.embeddings.word_embeddings.weight.copy_(sd["cls.predictions.decoder.weight"])
        model.tie_weights()
    model.return_prediction_scores = False
    t = lambda k: torch.from_numpy(d["in:" + k]).to(dev)  # noqa: E731
    out = model(t("input_ids"), t("segment_ids"), t("input_mask"), t("lm_label_ids"), t("is_next"), img_feats=t("img_feats"))
    got = np.array([out[0].item(), out[3].item()])
    rel = np.abs(got - d["losses"]) / np.abs(d["losses"])
    print(name, "losses (loss-only path)", got, d["losses"], rel)
    assert rel.max() < LOSS_RTOL
    assert out[1].shape[0] == 0
    out[0].backward()
    worst = 0.0
    for pname, p in model.named_parameters():
        key = "gnorm:" + pname
        if key in d and float(d[key]) > 1e-6 and p.grad is not None:
            worst = max(worst, abs(p.grad.double().norm().item() - float(d[key])) / float(d[key]))
    check_measured(name + ":gnorm_loss_only", worst, 5e-2)


def test_finetune_parity(dev):
    d = gu.load("tiny_finetune")
    cfg, dims, seed = d["config"], d["dims"], int(d["seed"])
    kw = _bi_inputs(d, dev)
    cr = dict(cfg, loss_type="ce", num_labels=2)
    model, _ = _build("BiImageBertForRetrieval", cr, seed + 1, dev)
    model.forward_mod = "coarse"
    with torch.no_grad():
        gt, gi = model(max_tag_length=dims["G"], **kw)
    assert (gt.cpu() - torch.from_numpy(d["ret_global_txt"])).abs().max() < 5e-3
    assert (gi.cpu() - torch.from_numpy(d["ret_global_img"])).abs().max() < 5e-3
    model.forward_mod = "fine"
    with torch.no_grad():
        fine = model(max_tag_length=dims["G"], **kw)
    check_measured("tiny_finetune:ret_fine_logits", _rel(fine, torch.from_numpy(d["ret_fine_logits"])), 2e-2)
    sim_ref = torch.from_numpy(d["ret_global_txt"]) @ torch.from_numpy(d["ret_global_img"]).t()
    masked = sim_ref - 2 * torch.eye(sim_ref.shape[0])
    model.forward_mod = "train"
    with Replay(dict(draw_randperm=[d["ret_randperm"]]), dev), gu.InjectHard(model.bert, masked.max(1)[1], masked.max(0)[1]):
        o = model(max_tag_length=dims["G"], **kw)
    got = np.array([o[0].item(), o[2].item(), o[3].item()])
    rel = np.abs(got - d["ret_train_losses"]) / np.abs(d["ret_train_losses"])
    print("retrieval train losses", got, d["ret_train_losses"], rel)
    assert rel.max() < SMALL_ROWS_RTOL
    assert np.array_equal(o[4].cpu().numpy(), d["ret_train_labels"])  # ITM labels bit-exact
    # VQA
    cv = dict(cfg, loss_type="bce", num_labels=37)
    model, _ = _build("BiImageBertForVQA", cv, seed + 2, dev)
    o = model(labels=torch.from_numpy(d["vqa_labels"]).to(dev), **kw)
    assert abs(o[0].item() - float(d["vqa_loss"])) / float(d["vqa_loss"]) < LOSS_RTOL
    check_measured("tiny_finetune:vqa_logits", _rel(o[1], torch.from_numpy(d["vqa_logits"])), 2e-2)
    o[0].backward()
    # VE
    ce_ = dict(cfg, loss_type="ce", num_labels=3, classifier="linear")
    model, _ = _build("BiImageBertForSequenceClassification", ce_, seed + 3, dev)
    o = model(labels=torch.from_numpy(d["ve_labels"]).to(dev), **kw)
    assert abs(o[0].item() - float(d["ve_loss"])) / float(d["ve_loss"]) < LOSS_RTOL
    check_measured("tiny_finetune:ve_logits", _rel(o[1], torch.from_numpy(d["ve_logits"])), 2e-2)


def test_data_parallel_replicas_match_single_module(dev):
    """oscar/run_retrieval.py:577-578,1125 wraps BiImageBertForRetrieval in nn.DataParallel (its only
    multi-GPU mode): replicas made by torch.nn.parallel.replicate share the module's non-tensor
    attributes by reference and run from one thread per device.  A replica (and nn.DataParallel on the
    visible devices) must give the single module's outputs bit for bit, from worker threads too."""
    import threading
    d = gu.load("tiny_finetune")
    cfg, dims, seed = d["config"], d["dims"], int(d["seed"])
    kw = _bi_inputs(d, dev)
    model, _ = _build("BiImageBertForRetrieval", dict(cfg, loss_type="ce", num_labels=2), seed + 1, dev)
    for mode in ("coarse", "fine"):
        model.forward_mod = mode
        with torch.no_grad():
            ref = model(max_tag_length=dims["G"], **kw)
            reps = torch.nn.parallel.replicate(model, [dev.index or 0, dev.index or 0])
            outs = [None, None]

            def run(i):
                with torch.cuda.device(dev):
                    outs[i] = reps[i](max_tag_length=dims["G"], **kw)

            for i in range(2):   # worker threads, as DataParallel's parallel_apply uses (one device here:
                t_ = threading.Thread(target=run, args=(i,))   # the replicas take turns)
                t_.start()
                t_.join()
            torch.cuda.synchronize()
            dp = torch.nn.DataParallel(model, device_ids=list(range(torch.cuda.device_count())))
            dpo = dp(max_tag_length=dims["G"], **kw)
        refs = ref if isinstance(ref, tuple) else (ref,)
        for o in outs + [dpo]:
            o = o if isinstance(o, tuple) else (o,)
            for a, b in zip(o, refs):
                assert torch.equal(a, b), mode


def test_unpadded_equals_padded_execution(dev):
    """Row-packed encoder execution (default) against the padded execution the reference performs:
    same losses, same hard-negative indices, same gradients; encoder outputs equal on the valid rows
    and zero on the padded ones."""
    from mvp_pytorch_amd.synthetic import synthetic_batch
    cfg = dict(gu.TINY_CFG, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    dims = dict(B=12, T=20, P=4, G=8, R=9)
    b = {k: v.to(dev) for k, v in synthetic_batch(dims, cfg, 31).items()}
    perm = torch.randperm(dims["B"], generator=torch.Generator().manual_seed(4))
    res, grads, outs = {}, {}, {}
    for unpad in (True, False):
        model, _ = _build("BiBertImgForPreTraining", cfg, 17, dev, train=True)
        model.wra_on_device = True
        model.bert.parallel_stacks = unpad   # fast path: packed + two streams; reference path: neither
        for enc in (model.bert.txt_encoder, model.bert.vis_encoder, model.bert.mul_encoder):
            enc.unpad = unpad
        torch.manual_seed(123)   # same device draws (WRA picks) in both runs
        with Replay(dict(draw_randperm=[perm.numpy()]), dev):
            o = model(input_ids_a=b["input_ids_a"], token_type_ids_a=b["segment_ids_a"], attention_mask_a=b["input_mask_a"],
                      masked_lm_labels_a=b["lm_label_ids_a"], input_ids_b=b["input_ids_b"], img_feats=b["img_feats"],
                      token_type_ids_b=b["segment_ids_b"], attention_mask_b=b["input_mask_b"],
                      masked_lm_labels_b=b["lm_label_ids_b"], max_tag_length=dims["G"], phrase_index=b["phrase_index"],
                      img_index=b["image_index"])
        o[0].backward()
        res[unpad] = torch.stack([x.detach() for x in o])
        grads[unpad] = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
        with torch.no_grad():
            (seq, pooled, _, _), (txt, vis, sim), _ = model.bert(
                input_ids_a=b["input_ids_a"], token_type_ids_a=b["segment_ids_a"], attention_mask_a=b["input_mask_a"],
                input_ids_b=b["input_ids_b"], token_type_ids_b=b["segment_ids_b"], attention_mask_b=b["input_mask_b"],
                img_feats=b["img_feats"], max_tag_length=dims["G"])
        outs[unpad] = (seq, txt, vis, sim)
    print("losses unpadded", res[True].tolist(), "padded", res[False].tolist())
    assert torch.allclose(res[True], res[False], rtol=2e-4, atol=1e-5)
    assert grads[True].keys() == grads[False].keys()
    # key.bias has a mathematically zero gradient (softmax is invariant to a per-query shift): pure rounding noise
    worst = max((_rel(grads[True][n], grads[False][n]), n) for n in grads[True]
                if grads[False][n].norm() > 1e-6 and not n.endswith("attention.self.key.bias"))
    print("worst gradient difference", worst)
    assert worst[0] < 6e-3      # one bf16 rounding (2^-8) of differently ordered sums: padded rows are added as zeros on one side only
    va = b["input_mask_a"].bool()
    vb = b["input_mask_b"].bool()
    (seq_u, txt_u, vis_u, sim_u), (seq_p, txt_p, vis_p, sim_p) = outs[True], outs[False]
    assert torch.equal(txt_u[va], txt_p[va]) and torch.equal(vis_u[vb], vis_p[vb]) and torch.equal(sim_u, sim_p)
    assert float(txt_u[~va].abs().max()) == 0.0 and float(vis_u[~vb].abs().max()) == 0.0
    vj = torch.cat([va, vb[:, dims["G"]:]], 1)
    assert torch.equal(seq_u[vj], seq_p[vj])


def test_stream_placement_does_not_change_results(dev):
    """Where a kernel is queued is not allowed to change what it computes: heads on the second stream (default) against
    everything on one stream, in the packed pipeline and in the general path: same losses and the same gradients up to the
    order of bf16 / f32 sums (checked at 1e-5 of each tensor's norm, far below the 2e-3 parity tolerance)."""
    from mvp_pytorch_amd.synthetic import synthetic_batch
    cfg = dict(gu.TINY_CFG, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    dims = dict(B=8, T=16, P=3, G=6, R=7)
    b = {k: v.to(dev) for k, v in synthetic_batch(dims, cfg, 77).items()}

    def run(heads, packed):
        model, _ = _build("BiBertImgForPreTraining", cfg, 5, dev, train=True)
        model.wra_on_device = True
        model.heads_beside = heads
        model.packed_pipeline = packed
        torch.manual_seed(321)
        o = model(input_ids_a=b["input_ids_a"], token_type_ids_a=b["segment_ids_a"], attention_mask_a=b["input_mask_a"],
                  masked_lm_labels_a=b["lm_label_ids_a"], input_ids_b=b["input_ids_b"], img_feats=b["img_feats"],
                  token_type_ids_b=b["segment_ids_b"], attention_mask_b=b["input_mask_b"],
                  masked_lm_labels_b=b["lm_label_ids_b"], max_tag_length=dims["G"], phrase_index=b["phrase_index"],
                  img_index=b["image_index"])
        o[0].backward()
        torch.cuda.synchronize()
        return torch.stack([x.detach() for x in o]), {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}

    for packed in (False, True):
        ref_l, ref_g = run(0, packed)
        for heads in (2, 1):
            l, g = run(heads, packed)
            assert torch.allclose(l, ref_l, rtol=1e-6, atol=0), (heads, packed, l.tolist(), ref_l.tolist())
            assert g.keys() == ref_g.keys()
            # bf16 atomics of the row scatter (packed pipeline) sum in arrival order: rounding differs at 2^-9 of single rows
            tol = 2e-3 if packed else 1e-5
            worst = max((_rel(g[n], ref_g[n]), n) for n in g if ref_g[n].norm() > 1e-6 and not n.endswith("attention.self.key.bias"))
            assert worst[0] < tol, (heads, packed, worst)


def test_finetune_models_unpadded_two_streams_equal_padded_one_stream(dev):
    """VQA (encode_hn=False path) and retrieval-train wrappers in training mode: row-packed stacks +
    text/visual stacks on two streams (defaults) against padded, single-stream execution."""
    from mvp_pytorch_amd.synthetic import synthetic_batch
    cfg = dict(gu.TINY_CFG, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    dims = dict(B=10, T=18, P=3, G=7, R=8)
    b = {k: v.to(dev) for k, v in synthetic_batch(dims, cfg, 41).items()}
    kw = dict(input_ids_a=b["input_ids_a"], token_type_ids_a=b["segment_ids_a"], attention_mask_a=b["input_mask_a"],
              input_ids_b=b["input_ids_b"], token_type_ids_b=b["segment_ids_b"], attention_mask_b=b["input_mask_b"],
              img_feats=b["img_feats"])
    labels = torch.rand(dims["B"], 37, generator=torch.Generator().manual_seed(1)).to(dev)
    perm = torch.randperm(dims["B"], generator=torch.Generator().manual_seed(2))
    for cls_name, extra in (("BiImageBertForVQA", dict(loss_type="bce", num_labels=37)),
                            ("BiImageBertForRetrieval", dict(loss_type="ce", num_labels=2))):
        got = {}
        for fast in (True, False):
            model, _ = _build(cls_name, dict(cfg, **extra), 23, dev, train=True)
            model.bert.parallel_stacks = fast
            for enc in (model.bert.txt_encoder, model.bert.vis_encoder, model.bert.mul_encoder):
                enc.unpad = "train" if fast else False
            if cls_name.endswith("VQA"):
                o = model(labels=labels, **kw)
            else:
                model.forward_mod = "train"
                with Replay(dict(draw_randperm=[perm.numpy()]), dev):
                    o = model(max_tag_length=dims["G"], **kw)
            o[0].backward()
            torch.cuda.synchronize()
            got[fast] = (o[0].detach(), {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None})
        assert torch.allclose(got[True][0], got[False][0], rtol=2e-4, atol=1e-5), (cls_name, got[True][0], got[False][0])
        worst = max((_rel(got[True][1][n], got[False][1][n]), n) for n in got[True][1] if got[False][1][n].norm() > 1e-6)
        print(cls_name, "loss", got[True][0].item(), "worst gradient difference", worst)
        assert worst[0] < 2e-3


def test_single_stream_unpadded_equals_padded(dev):
    """BertImgForPreTraining (single-stream backbone, interior padding between text and regions) in
    training mode: row-packed execution against the padded one."""
    from mvp_pytorch_amd.synthetic import synthetic_batch
    cfg = dict(gu.TINY_CFG, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, max_text_seq_length=16)
    dims = dict(B=9, T=16, P=0, G=4, R=7)
    b = {k: v.to(dev) for k, v in synthetic_batch(dims, cfg, 51, single_stream=True).items()}
    got = {}
    for unpad in (True, False):
        model, _ = _build("BertImgForPreTraining", cfg, 29, dev, train=True)
        model.bert.encoder.unpad = "train" if unpad else False
        o = model(input_ids=b["input_ids"], token_type_ids=b["segment_ids"], attention_mask=b["input_mask"],
                  masked_lm_labels=b["lm_label_ids"], next_sentence_label=b["is_next"], img_feats=b["img_feats"])
        o[0].backward()
        got[unpad] = (o[0].detach(), o[3].detach(), {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None})
    assert torch.allclose(got[True][0], got[False][0], rtol=2e-4) and torch.allclose(got[True][1], got[False][1], rtol=2e-4)
    worst = max((_rel(got[True][2][n], got[False][2][n]), n) for n in got[True][2] if got[False][2][n].norm() > 1e-6)
    print("single-stream unpadded vs padded: losses", got[True][0].item(), got[False][0].item(), "worst gradient difference", worst)
    assert worst[0] < 2e-3


def test_wra_device_path_equals_host_path(dev):
    """wra_sample_on_device (fixed shapes, no host round trip) against the host-index version of
    vl:1553-1596 with the same draws, forward and gradient; includes samples without phrases."""
    from mvp_pytorch_amd.modeling import modeling_vlbert as mv
    g = torch.Generator().manual_seed(21)
    B, La, R, H = 9, 14, 11, 64
    Lj = La + R
    seq = torch.randn(B, Lj, H, generator=g)
    n_t = torch.randint(2, La - 6, (B,), generator=g)
    n_p = torch.randint(0, 5, (B,), generator=g)
    n_p[0] = 0                                      # a sample without phrases
    phrase_index = torch.stack([1 + n_t, 1 + n_t + n_p], 1)
    n_r = torch.randint(3, R + 1, (B,), generator=g)
    img_index = torch.stack([torch.full((B,), La), La + n_r], 1)
    neg_img = (torch.arange(B) + 1 + torch.randint(0, B - 1, (B,), generator=g)) % B
    assert (neg_img != torch.arange(B)).all()
    for Pw in (La, 4):   # phrase grid as wide as the text, and config.max_phrases = 4
        pos_grid = torch.randint(0, 3, (B, Pw), generator=g)
        neg_grid = torch.randint(0, 3, (B, Pw), generator=g)
        flat = lambda grid: torch.cat([grid[t, :int(n_p[t])] for t in range(B)])  # noqa: E731
        a = seq.to(dev).requires_grad_(True)
        pos_d, neg_d = mv.wra_sample_on_device(a, phrase_index.to(dev), img_index.to(dev), La, draws=(pos_grid, neg_grid, neg_img),
                                               max_phrases=None if Pw == La else Pw)
        _check_wra_against_host(mv, seq, phrase_index, img_index, flat, pos_grid, neg_grid, neg_img, a, pos_d, neg_d, g, dev, B)


def _check_wra_against_host(mv, seq, phrase_index, img_index, flat, pos_grid, neg_grid, neg_img, a, pos_d, neg_d, g, dev, B):
    b = seq.to(dev).requires_grad_(True)
    vp = torch.nn.functional.normalize(mv.mask_slice_and_stack(b, phrase_index.to(dev)), p=2, dim=-1)
    vi = torch.nn.functional.normalize(mv.mask_slice_and_stack(b, img_index.to(dev)), p=2, dim=-1)
    pos_h, neg_h = mv.get_pos_neg_sims(vp @ vi.t(), phrase_index.to(dev), img_index.to(dev),
                                       draws=(flat(pos_grid), flat(neg_grid), neg_img.tolist()))
    assert torch.allclose(pos_d, pos_h, atol=1e-5) and torch.allclose(neg_d, neg_h, atol=1e-5)
    assert pos_d[0] == 0 and neg_d[0] == 0
    w = torch.randn(B, generator=g).to(dev)
    ((pos_d - neg_d) * w).sum().backward()
    ((pos_h - neg_h) * w).sum().backward()
    assert _rel(a.grad, b.grad) < 1e-5


def test_retrieval_cached_rerank_equals_fine(dev):
    """Two-stage retrieval with cached uni-modal outputs: every (caption, image) pair
    of a small cross product scores exactly as forward_mod='fine' on the materialised pair batch, the
    golden 'fine' logits are reproduced on the matched pairs, and the coarse similarity matrix equals
    forward_mod='coarse'."""
    d = gu.load("tiny_finetune")
    cfg, dims, seed = d["config"], d["dims"], int(d["seed"])
    kw = _bi_inputs(d, dev)
    model, _ = _build("BiImageBertForRetrieval", dict(cfg, loss_type="ce", num_labels=2), seed + 1, dev)
    n = kw["input_ids_a"].shape[0]
    enc = lambda packed: (  # noqa: E731
        model.encode_text(input_ids_a=kw["input_ids_a"], token_type_ids_a=kw["token_type_ids_a"],
                          attention_mask_a=kw["attention_mask_a"], packed=packed),
        model.encode_image(input_ids_b=kw["input_ids_b"], img_feats=kw["img_feats"], token_type_ids_b=kw["token_type_ids_b"],
                           attention_mask_b=kw["attention_mask_b"], max_tag_length=dims["G"], packed=packed))
    text, image = enc(False)          # padded execution: bit-identical to forward_mod='fine'
    text_p, image_p = enc(True)       # default: padded slots skipped
    ti, ii = torch.meshgrid(torch.arange(n, device=dev), torch.arange(n, device=dev), indexing="ij")
    ti, ii = ti.reshape(-1), ii.reshape(-1)
    got = model.rerank(text, image, ti, ii, chunk=5, packed=False)   # ragged chunks on purpose
    got_packed = model.rerank(text_p, image_p, ti, ii, chunk=7)      # default: padded slots skipped
    model.forward_mod = "fine"
    pair_kw = {k: v.index_select(0, ti if k.endswith("_a") else ii) for k, v in kw.items()}
    with torch.no_grad():
        ref = model(max_tag_length=dims["G"], **pair_kw)
    print("cached rerank vs fine: max abs diff", (got - ref).abs().max().item())
    assert torch.equal(got, ref)
    assert _rel(got_packed, ref) < 5e-3
    diag = torch.arange(n, device=dev) * (n + 1)
    assert _rel(got.index_select(0, diag), torch.from_numpy(d["ret_fine_logits"])) < 2e-2
    model.forward_mod = "coarse"
    with torch.no_grad():
        gt, gi = model(max_tag_length=dims["G"], **kw)
    # same f32 kernel (mvptr_sgemm_small) on the same rows: bit-identical; and equal to the library product to f32 round-off
    assert torch.equal(model.coarse_scores(text, image), model.bert._sim(gt, gi))
    assert (model.coarse_scores(text, image) - gt @ gi.t()).abs().max().item() < 2e-6
    # ranking of the images per caption from the cached scores == ranking from the pair-wise scores
    p_match = torch.softmax(got.float(), -1)[:, 1].view(n, n)
    assert torch.equal(p_match.argsort(1, descending=True), torch.softmax(ref.float(), -1)[:, 1].view(n, n).argsort(1, descending=True))


@pytest.mark.parametrize("gain", [1.0, 3.0])
def test_bi_pretrain_parity_b64_vs_oracle(dev, gain):
    """BERT-base, BASELINE configs[0] lengths but 64 pairs: every loss within 1e-3 of the oracle
    (the oracle itself is pinned to the reference by tests/test_oracle_golden.py).  gain 3: weights
    that spread the [CLS] embeddings out, for the integer outputs (hard-negative indices)."""
    from mvp_pytorch_amd.synthetic import synthetic_batch
    cfg = dict(gu.BASE_CFG, vocab_size=31000)  # phrase ids just above only_word_size; keeps the test light
    dims = dict(B=64, T=35, P=5, G=20, R=10)
    # 3x weight gain: the 64 [CLS] embeddings are spread out, so most top-2 margins of sim_mat sit
    # well above the bf16 noise and the integer outputs can be asserted
    model, sd = _build("BiBertImgForPreTraining", cfg, 99, dev, gain=gain)
    b = synthetic_batch(dims, cfg, 5)
    perm = torch.randperm(dims["B"], generator=torch.Generator().manual_seed(1))
    torch.set_num_threads(min(16, torch.get_num_threads()))
    with torch.no_grad():
        res_o, aux = orc.bi_bert_img_for_pretraining(
            sd, cfg, b["input_ids_a"], b["segment_ids_a"], b["input_mask_a"], b["lm_label_ids_a"], b["input_ids_b"],
            b["segment_ids_b"], b["input_mask_b"], b["lm_label_ids_b"], dims["G"], b["img_feats"],
            draws=orc.Draws(randperm=[perm.numpy()]), return_aux=True)
    n = dims["B"]
    masked = aux["sim_mat"] - 2 * torch.eye(n)
    bd = {k: v.to(dev) for k, v in b.items()}
    with torch.no_grad(), Replay(dict(draw_randperm=[perm.numpy()]), dev), gu.InjectHard(model.bert, masked.max(1)[1], masked.max(0)[1]):
        res = model(input_ids_a=bd["input_ids_a"], token_type_ids_a=bd["segment_ids_a"], attention_mask_a=bd["input_mask_a"],
                    masked_lm_labels_a=bd["lm_label_ids_a"], input_ids_b=bd["input_ids_b"], img_feats=bd["img_feats"],
                    token_type_ids_b=bd["segment_ids_b"], attention_mask_b=bd["input_mask_b"],
                    masked_lm_labels_b=bd["lm_label_ids_b"], max_tag_length=dims["G"])
    got = np.array([x.item() for x in res])
    ref = np.array([x.item() for x in res_o])
    rel = np.abs(got - ref) / np.abs(ref)
    print("B=64 gain", gain, "losses", got, "oracle", ref, "rel", rel)
    # gain 3 spreads the embeddings for the integer outputs below and amplifies the bf16 rounding with
    # it: 3e-3 on the token-row losses, 1.5e-2 on the 128-row ITM loss (measured 7.7e-3)
    if gain == 1.0:
        assert rel.max() < LOSS_RTOL, rel
    else:
        assert rel[:4].max() < 3e-3 and rel[4] < 1.5e-2, rel
    if gain == 1.0:
        # once more in TRAINING mode (dropout 0): the row-packed pipeline bench.py times, against the same oracle losses
        # 
        model.train()
        model.wra_on_device = True
        with torch.no_grad(), Replay(dict(draw_randperm=[perm.numpy()]), dev), gu.InjectHard(model.bert, masked.max(1)[1], masked.max(0)[1]):
            res_t = model(input_ids_a=bd["input_ids_a"], token_type_ids_a=bd["segment_ids_a"], attention_mask_a=bd["input_mask_a"],
                          masked_lm_labels_a=bd["lm_label_ids_a"], input_ids_b=bd["input_ids_b"], img_feats=bd["img_feats"],
                          token_type_ids_b=bd["segment_ids_b"], attention_mask_b=bd["input_mask_b"],
                          masked_lm_labels_b=bd["lm_label_ids_b"], max_tag_length=dims["G"])
        rel_t = np.abs(np.array([x.item() for x in res_t]) - ref) / np.abs(ref)
        print("B=64 training mode (packed pipeline) rel", rel_t)
        assert rel_t[:4].max() < LOSS_RTOL and rel_t[4] < SMALL_ROWS_RTOL, rel_t      # [4]: the 128-row ITM loss (measured 1.06e-3)
        model.eval()
        model.wra_on_device = False
    # free-running argmax: hard-negative indices against the f32 oracle's
    with torch.no_grad(), Replay(dict(draw_randperm=[perm.numpy()]), dev):
        _, single, hard = model.bert(input_ids_a=bd["input_ids_a"], token_type_ids_a=bd["segment_ids_a"],
                                     attention_mask_a=bd["input_mask_a"], input_ids_b=bd["input_ids_b"],
                                     token_type_ids_b=bd["segment_ids_b"], attention_mask_b=bd["input_mask_b"],
                                     img_feats=bd["img_feats"], max_tag_length=dims["G"], encode_hn=True)
    agree = ((hard[0].cpu() == aux["hard_txt_index"]).float().mean().item(), (hard[1].cpu() == aux["hard_img_index"]).float().mean().item())
    sim_err = (single[2].cpu() - aux["sim_mat"]).abs().max().item()
    print("B=64 hard-negative index agreement with the f32 oracle:", agree, "sim_mat max abs err", sim_err)
    assert sim_err < (5e-3 if gain == 1.0 else 1e-2)   # cosines of 12-layer bf16 outputs; gain 3 measured 5.2e-3
    # every row / column whose f32 top-2 margin exceeds 2x the measured max error CANNOT flip (two entries
    # move by at most sim_err each) and must agree exactly, unconditionally; the
    # (text, image) rows of the hard batch follow from them through the injected permutation
    m2 = aux["sim_mat"] - 2 * torch.eye(n)
    top_r, top_c = m2.topk(2, dim=1)[0], m2.t().topk(2, dim=1)[0]
    safe_r, safe_c = (top_r[:, 0] - top_r[:, 1]) > 2 * sim_err, (top_c[:, 0] - top_c[:, 1]) > 2 * sim_err
    first, second = perm[: n // 2], perm[n // 2:]
    img_rows_safe = torch.cat([safe_r[first], torch.ones(n - n // 2, dtype=torch.bool)])
    txt_rows_safe = torch.cat([torch.ones(n // 2, dtype=torch.bool), safe_c[second]])
    assert torch.equal(hard[1].cpu()[img_rows_safe], aux["hard_img_index"][img_rows_safe])
    assert torch.equal(hard[0].cpu()[txt_rows_safe], aux["hard_txt_index"][txt_rows_safe])
    print("B=64 rows with a safe margin:", int(safe_r.sum()), int(safe_c.sum()), "of", n)
    if gain > 1.0:
        assert min(agree) >= 0.9


def test_train_step_dropout_runs(dev):
    """A full training step with the reference's default dropout (0.1): finite losses, every
    parameter that should learn gets a finite gradient, AdamW moves the weights."""
    from mvp_pytorch_amd import modeling, train
    from mvp_pytorch_amd.synthetic import synthetic_batch
    cfg = dict(gu.TINY_CFG, hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1)
    torch.manual_seed(0)
    model = modeling.BiBertImgForPreTraining(modeling.make_config(cfg)).to(dev)
    model.train()
    opt, sched = train.build_optimizer(model, lr=1e-3, t_total=10)
    dims = dict(B=8, T=12, P=3, G=6, R=5)
    batch = synthetic_batch(dims, cfg, 7, device=dev)
    before = model.bert.txt_encoder.layer[0].attention.self.query.weight.detach().clone()
    losses = train.pretrain_step(model, batch, opt, sched, max_tag_length=dims["G"], return_losses=True)
    assert all(torch.isfinite(x).item() for x in losses)
    after = model.bert.txt_encoder.layer[0].attention.self.query.weight.detach()
    assert not torch.equal(before, after)
    l2 = train.pretrain_step(model, batch, opt, sched, max_tag_length=dims["G"], return_losses=True)
    assert all(torch.isfinite(x).item() for x in l2)


# ------------------------------------------------------------------------------ round-2 additions
def test_pretrain_step_matches_reference_adamw_probes(dev):
    """SURVEY §8 a-H: ONE full train.pretrain_step (forward, backward, fused AdamW, scheduler, zero_grad)
    on the device against the parameters the REFERENCE holds after its own backward + AdamW step
    (`adamw:*` probes of tiny_bi_pretrain.npz: lr 5e-3, eps 1e-8, wd 0.01 / 0 on bias + LayerNorm,
    run_pretrain_ml.py:379-393,632-644).  Adam's first step moves every element by lr * g / (|g| + eps),
    i.e. by +-lr wherever |g| >> eps: the comparison is on the UPDATE, element by element."""
    from mvp_pytorch_amd import train
    from mvp_pytorch_amd.optimization import AdamW, ConstantLRSchedule
    d = gu.load("tiny_bi_pretrain")
    cfg, dims = d["config"], d["dims"]
    model, sd = _build("BiBertImgForPreTraining", cfg, int(d["seed"]), dev, train=True)
    no_decay = ["bias", "LayerNorm.weight"]
    named = list(model.named_parameters())
    groups = [{"params": [p for n, p in named if not any(nd in n for nd in no_decay)], "weight_decay": 0.01},
              {"params": [p for n, p in named if any(nd in n for nd in no_decay)], "weight_decay": 0.0}]
    opt = AdamW(groups, lr=5e-3, eps=1e-8)
    sched = ConstantLRSchedule(opt)
    batch = {k[3:]: torch.from_numpy(v).to(dev) for k, v in d.items() if k.startswith("in:")}
    sim_ref = torch.from_numpy(d["sim_mat"])
    masked = sim_ref - 2 * torch.eye(sim_ref.shape[0])
    with Replay(d, dev), gu.InjectHard(model.bert, masked.max(1)[1], masked.max(0)[1]):
        losses = train.pretrain_step(model, batch, opt, sched, max_tag_length=dims["G"], return_losses=True)
    got = np.array([x.item() for x in losses])
    assert np.abs(got - d["losses"]).max() / np.abs(d["losses"]).max() < 3e-3
    params = dict(model.named_parameters())
    for n in gu.ADAMW_PROBES:
        before = sd[n].numpy().astype(np.float64)
        ref_upd = d["adamw:" + n].astype(np.float64) - before
        upd = params[n].detach().cpu().numpy().astype(np.float64) - before
        close = np.abs(upd - ref_upd) < 2e-4          # lr = 5e-3: a sign flip would be 1e-2
        frac = float(close.mean())
        rel = np.linalg.norm(upd - ref_upd) / (np.linalg.norm(ref_upd) + 1e-30)
        print("   adamw probe", n, "elements within 2e-4 of the reference update: %.4f" % frac, "rel L2 %.4f" % rel)
        if n == "logit_scale":
            assert abs(upd - ref_upd).max() < 2e-3, (upd, ref_upd)   # scalar whose gradient is a sum of cancelling terms
        else:
            assert frac > 0.97 and rel < 0.2, (n, frac, rel)
    assert all(p.grad is None for p in model.parameters())   # optimizer.zero_grad(set_to_none) ran


@pytest.mark.parametrize("dropout", [0.0, 0.1])
def test_graphed_step_matches_eager_steps(dev, dropout, request):
    """train.GraphedStep — the device work of a pre-training step captured as ONE HIP graph per batch signature
    and replayed — against train.pretrain_step on the same model, batch and draws.  Two eager warm-up steps, then captured steps
    under a warm-up schedule (the learning rate changes every step: it must reach the captured AdamW kernels through their
    descriptor tables, AdamW.advance).  Gradients are accumulated with f32 atomics, so two EAGER runs already differ in the last
    bits; the captured run must sit at that noise level (losses to 2e-5 relative over 6 steps, total parameter update to 1e-2
    relative L2 against 2x the eager / eager figure), every optimizer step counter must read 6, and with dropout on the salt word
    must have advanced once per replay (fresh masks: the seeds themselves are captured launch arguments)."""
    from mvp_pytorch_amd import dp, hip, train
    from mvp_pytorch_amd.optimization import AdamW, WarmupLinearSchedule
    from mvp_pytorch_amd.synthetic import synthetic_batch
    cfg = dict(gu.TINY_CFG, hidden_dropout_prob=dropout, attention_probs_dropout_prob=dropout, max_phrases=3)
    dims = dict(B=16, T=12, P=3, G=6, R=5)
    batch = synthetic_batch(dims, cfg, 33, device=dev)
    assert "host_counts" in batch
    steps = 6
    if dropout == 0.0:
        # no random draw left in the step: the WRA picks go (no phrase / image index) and the hard-negative split is pinned, so
        # the eager and the captured runs compute the SAME function (a captured draw replays at other generator offsets than an
        # eager one); the dropout variant keeps every device-side draw inside the capture
        batch.pop("phrase_index"), batch.pop("image_index")
        perm_dev = torch.randperm(dims["B"], generator=torch.Generator().manual_seed(9)).to(dev)
        orig_randperm = torch.randperm
        torch.randperm = lambda n, *a, **k: perm_dev
        request.addfinalizer(lambda: setattr(torch, "randperm", orig_randperm))

    def run(graphed):
        torch.manual_seed(0)
        from mvp_pytorch_amd import engine
        engine._seed_counter[0] = 0x5DEECE66D
        from mvp_pytorch_amd import modeling
        model = modeling.BiBertImgForPreTraining(modeling.make_config(cfg))      # (not _build: it switches dropout and the device-side WRA draws off)
        sd = {k: torch.from_numpy(v) for k, v in gu.det_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, 17, 1.0).items()}
        model.load_state_dict(sd)
        model.to(dev).train()
        opt = AdamW([{"params": list(model.parameters()), "weight_decay": 0.01}], lr=1e-3, eps=1e-8)
        sched = WarmupLinearSchedule(opt, warmup_steps=10, t_total=100)
        sync = dp.GradSync(model)
        stepper = train.GraphedStep(model, opt, sched, max_tag_length=dims["G"], max_grad_norm=1.0, grad_sync=sync, enabled=graphed)
        losses = []
        for _ in range(steps):
            losses.append(float(stepper(batch)))
        torch.cuda.synchronize()
        hip.check_device_errors(dev)
        upd = {n: (p.detach().float().cpu() - sd[n].float()) for n, p in model.named_parameters()}
        steps_seen = {int(opt.state[p]["step"]) for p in model.parameters() if p in opt.state and len(opt.state[p])}
        sync.close()
        return losses, upd, steps_seen, stepper

    salt0 = int(hip.dropout_salt(dev).item())
    l_a, u_a, s_a, _ = run(False)
    l_b, u_b, s_b, _ = run(False)
    l_g, u_g, s_g, st = run(True)
    assert st.last_error is None, st.last_error
    assert st.captures == 1 and st.replays == steps - 2 and st.eager_steps == 2, (st.captures, st.replays, st.eager_steps)
    assert s_a == s_g == {steps}, (s_a, s_g)
    assert int(hip.dropout_salt(dev).item()) == salt0 + steps - 2
    hip.dropout_salt(dev).zero_()

    def upd_rel(x, y):
        num = sum(float((x[n] - y[n]).pow(2).sum()) for n in x)
        den = sum(float(y[n].pow(2).sum()) for n in x)
        return (num / den) ** 0.5

    noise = upd_rel(u_b, u_a)
    got = upd_rel(u_g, u_a)
    print("dropout", dropout, "losses eager", l_a, "graphed", l_g, "update rel L2: eager/eager %.3e graphed/eager %.3e" % (noise, got))
    if dropout == 0.0:
        assert max(abs(a - g) / abs(a) for a, g in zip(l_a, l_g)) < 2e-5 + 4 * max(abs(a - b) / abs(a) for a, b in zip(l_a, l_b))
        assert got < max(1e-2, 2 * noise), (got, noise)
    else:
        # the first two (eager) steps draw the same masks; the replays draw others than the eager run's (salted seeds): same
        # distribution, finite, and still descending like the eager run
        assert max(abs(a - g) / abs(a) for a, g in zip(l_a[:2], l_g[:2])) < 2e-5 + 4 * max(abs(a - b) / abs(a) for a, b in zip(l_a[:2], l_b[:2]))
        assert all(np.isfinite(l_g)) and abs(l_g[-1] - l_a[-1]) / abs(l_a[-1]) < 0.1


def test_finetune_host_counts_and_captured_vqa_step(dev):
    """The fine-tune path without read-backs: BiImageBertForVQA(host_counts=synthetic.finetune_host_counts(...)) gives the loss and
    gradients of the step that reads its row counts back (same kernels on the same rows: bit-identical loss), and that step —
    forward, backward, fused clip, AdamW — runs as a captured HIP graph (train.GraphedStep with a custom forward)."""
    from mvp_pytorch_amd import dp, train
    from mvp_pytorch_amd.optimization import AdamW, ConstantLRSchedule
    from mvp_pytorch_amd.synthetic import finetune_host_counts, synthetic_batch
    cfg = dict(gu.TINY_CFG, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, loss_type="bce", num_labels=37)
    dims = dict(B=10, T=18, P=3, G=20, R=8)
    b = {k: v.to(dev) for k, v in synthetic_batch(dims, cfg, 41).items() if isinstance(v, torch.Tensor)}
    kw = dict(input_ids_a=b["input_ids_a"], token_type_ids_a=b["segment_ids_a"], attention_mask_a=b["input_mask_a"],
              input_ids_b=b["input_ids_b"], token_type_ids_b=b["segment_ids_b"], attention_mask_b=b["input_mask_b"], img_feats=b["img_feats"])
    labels = torch.rand(dims["B"], 37, generator=torch.Generator().manual_seed(1)).to(dev)
    hc = finetune_host_counts(b, 20)
    got = {}
    for use_hc in (False, True):
        model, _ = _build("BiImageBertForVQA", cfg, 23, dev, train=True)
        o = model(labels=labels, host_counts=hc if use_hc else None, **kw)
        o[0].backward()
        torch.cuda.synchronize()
        got[use_hc] = (o[0].detach().clone(), {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None})
    assert torch.equal(got[True][0], got[False][0])
    worst = max((_rel(got[True][1][n], got[False][1][n]), n) for n in got[False][1] if got[False][1][n].norm() > 1e-6)
    assert worst[0] < 1e-5, worst
    model, _ = _build("BiImageBertForVQA", cfg, 23, dev, train=True)
    opt = AdamW([{"params": list(model.parameters()), "weight_decay": 0.01}], lr=1e-3, eps=1e-8)
    sync = dp.GradSync(model)
    step = train.GraphedStep(model, opt, ConstantLRSchedule(opt), max_grad_norm=1.0, grad_sync=sync, forward=lambda m, bb: m(**bb))
    vb = dict(kw, labels=labels, host_counts=hc)
    losses = [float(step(vb)) for _ in range(5)]
    torch.cuda.synchronize()
    assert step.last_error is None and step.captures == 1 and step.replays == 3, (step.last_error, step.captures, step.replays)
    assert abs(losses[0] - float(got[True][0])) < 1e-6 * abs(losses[0]) and losses[-1] < losses[0]
    from mvp_pytorch_amd import hip
    hip.dropout_salt(dev).zero_()
    sync.close()


def test_single_stream_host_counts_and_captured_step(dev):
    """BertImgForPreTraining (loss-only training, a17) with the batch's own host counts — valid rows / longest sequence / scored rows —
    gives the loss of the step that reads them back, a wrong scored count raises at the next host look (device error word), and
    the whole step runs as a captured HIP graph."""
    from mvp_pytorch_amd import dp, hip, modeling, train
    from mvp_pytorch_amd.optimization import AdamW, ConstantLRSchedule
    from mvp_pytorch_amd.synthetic import synthetic_batch
    cfg = dict(gu.TINY_CFG, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, max_text_seq_length=12, vocab_size=1000)
    dims = dict(B=12, T=12, P=0, G=6, R=5)
    batch = synthetic_batch(dims, cfg, 5, single_stream=True, device=dev)
    hc = batch["host_counts"]

    def build():
        torch.manual_seed(0)
        m = modeling.BertImgForPreTraining(modeling.make_config(cfg)).to(dev).train()
        m.return_prediction_scores = False
        return m

    kw = train.model_inputs(batch, dims["G"])
    plain = {k: v for k, v in kw.items() if k != "host_counts"}
    l0 = build()(**plain)[0]
    l1 = build()(**kw)[0]
    assert torch.equal(l0, l1), (l0, l1)
    hip.check_device_errors(dev)
    bad = dict(kw, host_counts=dict(hc, scored=int(hc["scored"]) - 1))
    build()(**bad)
    with pytest.raises(RuntimeError, match="more scored"):
        hip.check_device_errors(dev)
    model = build()
    opt = AdamW([{"params": list(model.parameters()), "weight_decay": 0.01}], lr=1e-3, eps=1e-8)
    sync = dp.GradSync(model)
    step = train.GraphedStep(model, opt, ConstantLRSchedule(opt), max_tag_length=dims["G"], max_grad_norm=1.0, grad_sync=sync)
    losses = [float(step(batch)) for _ in range(5)]
    torch.cuda.synchronize()
    assert step.last_error is None and step.captures == 1 and step.replays == 3, (step.last_error, step.captures, step.replays)
    assert abs(losses[0] - float(l1)) < 1e-6 * abs(losses[0]) and losses[-1] < losses[0]
    hip.dropout_salt(dev).zero_()
    sync.close()


def test_branches_parity(dev):
    """qa_ans + phrase_mod='hard' (vl:1264-1283), hn_mod='sample' (vl:535-540), use_b, classifier='mlp',
    soft-label / MSE / BCE / KL losses (vl:1777-1797) against tiny_branches.npz (reference outputs)."""
    d = gu.load("tiny_branches")
    cfg, dims, seed = d["config"], d["dims"], int(d["seed"])
    kw = _bi_inputs(d, dev)
    t = lambda k: torch.from_numpy(d["in:" + k]).to(dev)  # noqa: E731
    model, _ = _build("BiBertImgForPreTraining", cfg, seed, dev)
    rp = {k[3:]: v for k, v in d.items() if k.startswith("qa_draw_")}
    sim_ref = torch.from_numpy(d["qa_sim_mat"])
    masked = sim_ref - 2 * torch.eye(sim_ref.shape[0])
    with Replay(rp, dev), gu.InjectHard(model.bert, masked.max(1)[1], masked.max(0)[1]):
        res = model(masked_lm_labels_a=t("lm_label_ids_a"), masked_lm_labels_b=t("lm_label_ids_b"), max_tag_length=dims["G"],
                    img_index=t("image_index"), phrase_index=t("phrase_index"), qa_ans=torch.from_numpy(d["qa_ans"]).to(dev),
                    phrase_mod="hard", **kw)
    assert len(res) == 7
    got, ref = np.array([x.item() for x in res]), d["qa_losses"]
    rel = np.abs(got - ref) / np.abs(ref)
    print("qa_ans + phrase_mod='hard' losses", got, ref, rel)
    assert max(rel[0], rel[1], rel[3]) < LOSS_RTOL and max(rel[2], rel[4], rel[5]) < SMALL_ROWS_RTOL
    assert abs(got[6] - ref[6]) < 2e-3 + LOSS_RTOL * abs(ref[6])
    res[0].backward()
    gq = model.qa_head.weight.grad.double().norm().item()
    assert abs(gq - float(d["qa_gnorm:qa_head.weight"])) / float(d["qa_gnorm:qa_head.weight"]) < 5e-2
    # sampled hard negatives: the reference's multinomial draws replayed -> same integer outputs
    with torch.no_grad(), Replay(dict(draw_randperm=[d["hs_randperm"]], draw_multinomial=list(d["hs_multinomial"])), dev):
        o, _, hard = model.bert(max_tag_length=dims["G"], encode_hn=True, hn_mod="sample", logit=model.logit_scale.exp(), **kw)
    assert np.array_equal(hard[0].cpu().numpy(), d["hs_hard_txt_index"]) and np.array_equal(hard[1].cpu().numpy(), d["hs_hard_img_index"])
    assert _rel(o[3], torch.from_numpy(d["hs_hard_pooled_output"])) < 2e-2
    cases = [("mlp", dict(loss_type="ce", num_labels=3, classifier="mlp", cls_hidden_scale=3), 1, dict(use_b=True)),
             ("soft", dict(loss_type="ce", num_labels=2, classifier="linear"), 2, dict(soft_label=True)),
             ("mse", dict(loss_type="ce", num_labels=1, classifier="linear"), 3, {}),
             ("bce", dict(loss_type="bce", num_labels=37, classifier="linear"), 4, {})]
    for tag, extra, off, fkw in cases:
        m, _ = _build("BiImageBertForSequenceClassification", dict(cfg, **extra), seed + off, dev)
        o = m(labels=torch.from_numpy(d[tag + "_labels"]).to(dev), **fkw, **kw)
        e_loss = abs(o[0].item() - float(d[tag + "_loss"])) / abs(float(d[tag + "_loss"]))
        e_log = _rel(o[1], torch.from_numpy(d[tag + "_logits"]))
        e_abs = (o[1].float().cpu() - torch.from_numpy(d[tag + "_logits"])).abs().max().item()
        print("   ", tag, "loss rel", e_loss, "logits rel L2", e_log, "max abs", e_abs)
        # the one-logit regression head ("mse") outputs 4 values near zero: their error is the absolute
        # bf16 noise of the pooled row (a few 1e-3), not a fraction of their own size
        assert e_loss < (3e-3 if tag == "mse" else LOSS_RTOL) and (e_log < 2e-2 or (tag == "mse" and e_abs < 5e-3)), tag
        o[0].backward()
    m, _ = _build("BiImageBertForVQA", dict(cfg, loss_type="kl", num_labels=3129), seed + 5, dev)
    o = m(labels=torch.from_numpy(d["kl_labels"]).to(dev), **kw)
    assert abs(o[0].item() - float(d["kl_loss"])) / float(d["kl_loss"]) < LOSS_RTOL
    assert o[1].shape == (dims["B"], 3129) and _rel(o[1][:, :128], torch.from_numpy(d["kl_logits_head"])) < 2e-2
    o[0].backward()
    gk = m.cls.predictions.decoder.weight.grad.double().norm().item()
    ref = float(d["kl_gnorm:cls.predictions.decoder.weight"])
    assert abs(gk - ref) / ref < 5e-2


def test_half_model_inference_matches_float(dev):
    """`model.half()` (run_retrieval.py / SURVEY §8b: the scripts may call it): f16 parameters and f16
    region features go through the same bf16 kernels; scores equal the f32-parameter run to rounding."""
    d = gu.load("tiny_finetune")
    cfg, dims, seed = d["config"], d["dims"], int(d["seed"])
    kw = _bi_inputs(d, dev)
    model, _ = _build("BiImageBertForRetrieval", dict(cfg, loss_type="ce", num_labels=2), seed + 1, dev)
    model.forward_mod = "fine"
    with torch.no_grad():
        ref = model(max_tag_length=dims["G"], **kw).float()
        model.half()
        kw16 = dict(kw, img_feats=kw["img_feats"].half())
        got = model(max_tag_length=dims["G"], **kw16).float()
    print("half() vs float fine logits rel L2", _rel(got, ref))
    assert _rel(got, ref) < 2e-2


def test_shard_without_masked_rows(dev):
    """A batch / data-parallel shard with no masked tag row and no masked text row (15 % masking gives
    no guarantee, oscar_tsv4.py:782-893): the step must not raise (M = 0 GEMMs), the two MLM losses are
    exact zeros and every head parameter still receives a (zero) gradient, so ranks stay in lockstep."""
    from mvp_pytorch_amd.synthetic import synthetic_batch
    cfg = dict(gu.TINY_CFG, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    dims = dict(B=4, T=12, P=3, G=6, R=5)
    model, _ = _build("BiBertImgForPreTraining", cfg, 5, dev, train=True)
    b = synthetic_batch(dims, cfg, 9, device=dev)
    b["lm_label_ids_a"].fill_(-1)
    b["lm_label_ids_b"].fill_(-1)
    from mvp_pytorch_amd import train
    out = model(**train.model_inputs(b, dims["G"]))
    assert out[1].item() == 0.0 and out[3].item() == 0.0 and torch.isfinite(out[0])
    out[0].backward()
    for n, p in model.named_parameters():
        if n.startswith("half_mlm.") or n.startswith("cls.predictions."):
            assert p.grad is not None and float(p.grad.abs().max()) == 0.0, n
    assert model.bert.txt_encoder.layer[0].attention.self.query.weight.grad.abs().max() > 0


def test_weight_cache_sees_data_updates(dev):
    """an optimizer that updates through `.data` (the reference's own AdamW,
    optimization.py:176,187) does not bump Tensor._version; in training the bf16 working copies are
    rebuilt at every forward, in inference invalidate_weight_caches() drops them."""
    from mvp_pytorch_amd import engine
    cfg = dict(gu.TINY_CFG, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    d = gu.load("tiny_finetune")
    kw = _bi_inputs(d, dev)
    model, _ = _build("BiImageBertForRetrieval", dict(cfg, loss_type="ce", num_labels=2), 3, dev, train=True)
    model.forward_mod = "fine"
    a = model(max_tag_length=20, **kw).detach().clone()
    # only parameters that reach the output through bf16 working copies alone (joint stack): if the copies were not
    # rebuilt the output could not move
    for n, p in model.named_parameters():
        if n.startswith("bert.mul_encoder.") and n.endswith("dense.weight"):
            p.data.mul_(0.5)                   # invisible to the version counters
    a2 = model(max_tag_length=20, **kw).detach().clone()
    assert not torch.allclose(a, a2)           # training mode: copies refreshed unconditionally
    for n, p in model.named_parameters():
        if n.startswith("bert.mul_encoder.") and n.endswith("dense.weight"):
            p.data.mul_(2.0)
    for p in model.parameters():
        p.data.mul_(0.5)
    b = model(max_tag_length=20, **kw).detach().clone()
    assert not torch.allclose(a, b)
    model.eval()
    with torch.no_grad():
        c = model(max_tag_length=20, **kw).clone()
        for p in model.parameters():
            p.data.mul_(2.0)
        assert engine.invalidate_weight_caches(model) > 0
        fresh = model(max_tag_length=20, **kw).clone()
    assert not torch.allclose(fresh, c)
    assert _rel(fresh, a) < 2e-2               # back at the original weights


def test_configs1_shape_full_batch(dev):
    """BASELINE configs[1] at FULL size (B=256, 70+5 / 20 / 50, BERT-base, all heads): size-independent
    properties — every loss finite and in its a-priori range at random init, row-packed execution equal
    to the padded execution on losses and gradients, ITM labels / hard indices well-formed, every
    trained parameter receives a finite gradient."""
    from mvp_pytorch_amd import train
    from mvp_pytorch_amd.synthetic import synthetic_batch
    cfg = dict(gu.BASE_CFG, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, max_phrases=5)
    dims = gu.CFG2_DIMS
    b = synthetic_batch(dims, cfg, 77, device=dev)
    perm = torch.randperm(dims["B"], generator=torch.Generator().manual_seed(5))
    res, grads = {}, {}
    for unpad in (True, False):
        model, _ = _build("BiBertImgForPreTraining", cfg, 23, dev, train=True)   # deterministic weights
        model.wra_on_device = True
        for enc in (model.bert.txt_encoder, model.bert.vis_encoder, model.bert.mul_encoder):
            enc.unpad = unpad
        torch.manual_seed(11)   # same device draws (WRA picks)
        with Replay(dict(draw_randperm=[perm.numpy()]), dev):
            outputs, single, hard = model.bert(
                input_ids_a=b["input_ids_a"], token_type_ids_a=b["segment_ids_a"], attention_mask_a=b["input_mask_a"],
                input_ids_b=b["input_ids_b"], token_type_ids_b=b["segment_ids_b"], attention_mask_b=b["input_mask_b"],
                img_feats=b["img_feats"], max_tag_length=dims["G"], encode_hn=True) if unpad else (None, None, None)
        torch.manual_seed(11)
        # the padded run trains on the hard batch the packed run mined: at random init the top-2 margins
        # of sim_mat are ~1e-5, so the two executions' roundings would otherwise pick different negatives
        # for a few rows and the comparison would measure that choice, not the arithmetic
        import contextlib
        inject = contextlib.nullcontext() if unpad else gu.InjectHard(model.bert, mined[0], mined[1])
        with Replay(dict(draw_randperm=[perm.numpy()]), dev), inject:
            o = model(**train.model_inputs(b, dims["G"]))
        o[0].backward()
        torch.cuda.synchronize()
        res[unpad] = torch.stack([x.detach() for x in o]).cpu()
        names = ["bert.mul_encoder.layer.5.output.dense.weight", "bert.txt_encoder.layer.0.attention.self.query.weight",
                 "bert.vis_encoder.layer.2.intermediate.dense.weight", "bert.img_embedding.weight", "cls.predictions.transform.dense.weight"]
        pd = dict(model.named_parameters())
        grads[unpad] = {n: pd[n].grad.clone() for n in names}
        if unpad:
            n = dims["B"]
            ht, hi = hard[0].cpu(), hard[1].cpu()
            masked = single[2].float() - 2 * torch.eye(n, device=dev)
            mined = (masked.max(1)[1], masked.max(0)[1])     # (hard_img_index, hard_txt_index) of vl:531-534
            assert ht.dtype == torch.int64 and hi.shape == (n,) and int(ht.min()) >= 0 and int(hi.max()) < n
            ar = torch.arange(n)
            first, second = perm[: n // 2], perm[n // 2:]
            # rows drawn first keep their own text and take another image; the others keep their image
            assert torch.equal(ht[: n // 2], ar[first]) and torch.equal(hi[n // 2:], ar[second])
            assert (hi[: n // 2] != ar[first]).all() and (ht[n // 2:] != ar[second]).all()
            for nme, p in model.named_parameters():
                if nme.startswith("qa_head"):
                    continue
                assert p.grad is not None and torch.isfinite(p.grad).all(), nme
        del model
    print("configs[1] losses packed", res[True].tolist(), "padded", res[False].tolist())
    l = res[True]
    assert torch.isfinite(l).all()
    assert 9.5 < l[1] < 12.5 and 9.5 < l[3] < 12.5          # ln(30522) = 10.3 for an untrained decoder
    assert 4.5 < l[2] < 6.5 and 0.4 < l[4] < 1.2            # ln(256) = 5.55 ; ln 2 = 0.69
    assert torch.allclose(res[True], res[False], rtol=5e-4, atol=1e-4)
    worst = max((_rel(grads[True][n], grads[False][n]), n) for n in grads[True])
    print("configs[1] worst gradient difference packed vs padded", worst)
    # the two executions sum attention in different orders; at this size either one sits 1.5e-2 (median
    # over parameters, 6e-2 worst) from the f32 oracle's gradients and ~0.7e-2 (2e-2 worst) from the other
    # (tests/diag_cfg1_grads.py 256): bf16 rounding through 12 layers and a 256 x 256 contrastive softmax
    assert worst[0] < 3e-2


def test_configs3_retrieval_scale(dev):
    """BASELINE configs[3] at its size: 1 000 images x 5 captions (COCO-1k shape, README retrieval lengths
    50 tok + 5 phrases / 30 tags / 50 regions), BERT-base, random init.  Two-stage engine: encode every caption
    and image once, coarse ranks on the device, re-rank the top-8 images of every caption (40 000 pairs,
    row-packed).  Size-independent properties: coarse ranks == an independent numpy argsort walk of the same
    similarity matrix; on a sample of pairs the padded cached re-rank equals forward_mod='fine' bit for bit and the
    row-packed one to bf16 rounding; second-stage ranks are well-formed; everything finite."""
    from mvp_pytorch_amd import modeling, retrieval_eval
    from mvp_pytorch_amd.synthetic import synthetic_batch
    n_img, c, topk = 1000, 5, 8
    dims = dict(B=n_img * c, T=50, P=5, G=30, R=50)
    cfg = dict(gu.BASE_CFG, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, loss_type="ce", num_labels=2)
    torch.manual_seed(3)
    model = modeling.BiImageBertForRetrieval(modeling.make_config(cfg)).to(dev).eval()
    b = synthetic_batch(dims, cfg, 31, device=dev)
    n_cap = dims["B"]
    img_rows = torch.arange(0, n_cap, c, device=dev)       # caption j describes image j // c; one copy of every image

    def encode(packed):
        text = {k: [] for k in ("seq", "mask", "glob")}
        for s0 in range(0, n_cap, 1000):
            sl = slice(s0, s0 + 1000)
            t = model.encode_text(input_ids_a=b["input_ids_a"][sl], token_type_ids_a=b["segment_ids_a"][sl],
                                  attention_mask_a=b["input_mask_a"][sl], packed=packed)
            for k in text:
                text[k].append(t[k])
        text = {k: torch.cat(v) for k, v in text.items()}
        image = model.encode_image(input_ids_b=b["input_ids_b"][img_rows], img_feats=b["img_feats"][img_rows],
                                   token_type_ids_b=b["segment_ids_b"][img_rows], attention_mask_b=b["input_mask_b"][img_rows],
                                   max_tag_length=dims["G"], packed=packed)
        return text, image

    text, image = encode(True)
    sim_ti = model.coarse_scores(text, image)                # [n_cap, n_img]
    assert sim_ti.shape == (n_cap, n_img) and bool(torch.isfinite(sim_ti).all())
    sim = sim_ti.t().contiguous()                            # run_retrieval.py layout: [n_img, n_cap]
    out = retrieval_eval.coarse_ranks(sim, c, 20, topk)
    s_np = sim.cpu().numpy()
    order_t2i = np.argsort(-s_np, axis=0, kind="stable")     # independent check: full argsort walk
    pos = np.empty_like(order_t2i)
    np.put_along_axis(pos, order_t2i, np.arange(n_img)[:, None], axis=0)
    gt_img = np.arange(n_cap) // c
    ties = (np.sort(s_np, axis=0)[1:] == np.sort(s_np, axis=0)[:-1]).any()
    if not ties:
        assert np.array_equal(out["t2i_ranks"].cpu().numpy(), pos[gt_img, np.arange(n_cap)])
        assert np.array_equal(out["t2i_topk"].cpu().numpy(), order_t2i[:topk].T)
    cand = out["t2i_topk"]                                   # [n_cap, topk] images to re-rank per caption
    ti = torch.arange(n_cap, device=dev).repeat_interleave(topk)
    ii = cand.reshape(-1)
    scores = model.rerank(text, image, ti, ii, chunk=4096)   # row-packed (default)
    assert scores.shape == (n_cap * topk, 2) and bool(torch.isfinite(scores.float()).all())
    p_match = torch.softmax(scores.float(), -1)[:, 1].view(n_cap, topk)
    gt_mask = cand == torch.from_numpy(gt_img).to(dev)[:, None]
    rr = retrieval_eval.rerank_ranks(p_match, cand, gt_mask)
    assert int(rr.min()) >= 0 and int(rr.max()) <= topk
    assert bool((rr[~gt_mask.any(1)] == topk).all())
    # the README's candidate counts (run_retrieval.py evaluation: 64 images re-ranked per caption, 128 captions per image) on
    # the first 100 images / their 500 captions: 32 000 + 12 800 pairs through the cached engine
    sub_i, sub_c = 100, 100 * c
    c64 = sim_ti[:sub_c].topk(64, dim=1).indices                                  # [500, 64] images per caption
    s64 = model.rerank(text, image, torch.arange(sub_c, device=dev).repeat_interleave(64), c64.reshape(-1), chunk=4096)
    c128 = sim[:sub_i].topk(128, dim=1).indices                                   # [100, 128] captions per image
    s128 = model.rerank(text, image, c128.reshape(-1), torch.arange(sub_i, device=dev).repeat_interleave(128), chunk=4096)
    assert s64.shape == (sub_c * 64, 2) and s128.shape == (sub_i * 128, 2)
    assert bool(torch.isfinite(s64.float()).all()) and bool(torch.isfinite(s128.float()).all())
    p64 = torch.softmax(s64.float(), -1)[:, 1].view(sub_c, 64)
    rr64 = retrieval_eval.rerank_ranks(p64, c64, c64 == torch.from_numpy(gt_img[:sub_c]).to(dev)[:, None])
    assert int(rr64.min()) >= 0 and int(rr64.max()) <= 64
    # the same pair scored inside either candidate list gets the same logits (row-packed, batch-independent kernels)
    hit = (c64[:, :, None] == torch.arange(sub_i, device=dev)[None, None, :]).any(1)    # caption x image (< 100) in c64
    both_ways = 0
    for img in range(0, sub_i, 17):
        for k, cap in enumerate(c128[img].tolist()):
            if cap < sub_c and bool(hit[cap, img]):
                j = int((c64[cap] == img).nonzero()[0])
                assert torch.equal(s64.view(sub_c, 64, 2)[cap, j], s128.view(sub_i, 128, 2)[img, k]) or \
                    float((s64.view(sub_c, 64, 2)[cap, j].float() - s128.view(sub_i, 128, 2)[img, k].float()).abs().max()) < 2e-2
                both_ways += 1
    print("README candidate counts: 32000 + 12800 pairs re-ranked; pairs present in both lists checked:", both_ways)
    # a sample of pairs against the reference evaluation's per-pair forward pass
    g = torch.Generator().manual_seed(9)
    pick = torch.randperm(n_cap * topk, generator=g)[:384].to(dev)
    t_s, i_s = ti[pick], ii[pick]
    text_pad, image_pad = encode(False)
    got_pad = model.rerank(text_pad, image_pad, t_s, i_s, chunk=384, packed=False)      # same batch as the per-pair pass below
    got_ragged = model.rerank(text_pad, image_pad, t_s, i_s, chunk=100, packed=False)   # other batch sizes: torch's f32 heads may
                                                                                         # pick other GEMM kernels (last-bit differences)
    model.forward_mod = "fine"
    r = img_rows[i_s]
    with torch.no_grad():
        ref = model(input_ids_a=b["input_ids_a"][t_s], token_type_ids_a=b["segment_ids_a"][t_s], attention_mask_a=b["input_mask_a"][t_s],
                    input_ids_b=b["input_ids_b"][r], token_type_ids_b=b["segment_ids_b"][r], attention_mask_b=b["input_mask_b"][r],
                    img_feats=b["img_feats"][r], max_tag_length=dims["G"])
    assert torch.equal(got_pad, ref)
    assert (got_ragged.float() - ref.float()).abs().max().item() < 1e-5
    dp = (torch.softmax(scores[pick].float(), -1)[:, 1] - torch.softmax(ref.float(), -1)[:, 1]).abs().max().item()
    print("configs[3]: %d pairs re-ranked; row-packed vs per-pair 'fine' max |delta p(match)| = %.2e" % (n_cap * topk, dp))
    assert dp < 2e-2


def test_configs4_vqa_shape_vs_oracle(dev):
    """BASELINE configs[4] shapes (run_vqa.py README: max_seq_length 128 + 5 phrases -> La = 133, 30 tag
    slots + 50 regions -> Lb = 80, no max_tag_length forwarded -> joint length 133 + 60 = 193; 3129-way
    answer head, BCE loss): the HIP model on 64 questions, the f32 oracle on the first 6 of them
    (logits rows are per-sample: parity on a B-subset), and the batch-level loss on those 6."""
    from mvp_pytorch_amd.synthetic import synthetic_batch
    cfg = dict(gu.BASE_CFG, vocab_size=31000, loss_type="bce", num_labels=3129, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    dims = dict(B=64, T=128, P=5, G=30, R=50)
    model, sd = _build("BiImageBertForVQA", cfg, 41, dev)
    b = synthetic_batch(dims, cfg, 8)
    g = torch.Generator().manual_seed(2)
    labels = (torch.rand(dims["B"], 3129, generator=g) < 0.002).float() * torch.rand(dims["B"], 3129, generator=g)
    kw = lambda bb, n: dict(input_ids_a=bb["input_ids_a"][:n], token_type_ids_a=bb["segment_ids_a"][:n],  # noqa: E731
                            attention_mask_a=bb["input_mask_a"][:n], input_ids_b=bb["input_ids_b"][:n],
                            token_type_ids_b=bb["segment_ids_b"][:n], attention_mask_b=bb["input_mask_b"][:n],
                            img_feats=bb["img_feats"][:n])
    bd = {k: v.to(dev) for k, v in b.items()}
    with torch.no_grad():
        loss64, logits64 = model(labels=labels.to(dev), **kw(bd, 64))[:2]
        loss6, logits6 = model(labels=labels[:6].to(dev), **kw(bd, 6))[:2]
        torch.set_num_threads(min(16, torch.get_num_threads()))
        ref_loss, ref_logits = orc.bi_vqa(sd, cfg, labels=labels[:6], **kw(b, 6))
    assert logits64.shape == (64, 3129) and torch.isfinite(loss64)
    e = _rel(logits6, ref_logits)
    e_rows = _rel(logits64[:6], ref_logits)
    print("configs[4] shape: logits rel L2 vs oracle", e, "(rows of the 64-batch:", e_rows, ") loss", loss6.item(), ref_loss.item())
    assert e < 2e-2 and e_rows < 2e-2
    assert abs(loss6.item() - ref_loss.item()) / ref_loss.item() < LOSS_RTOL
    # training step at this shape runs (row-packed joint length 193 <= 256 rows of LDS)
    model.train()
    out = model(labels=labels.to(dev), **kw(bd, 64))
    out[0].backward()
    assert torch.isfinite(model.cls.predictions.decoder.weight.grad).all()
    # the BACKWARD pass at this shape against the oracle on the 6-question subset (dropout 0): the
    # 3129-way decoder weight and bias, the head transform, one encoder weight of each stack and the region embedding
    model.zero_grad(set_to_none=True)
    out6 = model(labels=labels[:6].to(dev), **kw(bd, 6))
    out6[0].backward()
    torch.cuda.synchronize()
    sdg = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items()}
    torch.set_num_threads(min(16, torch.get_num_threads()))
    rl, _ = orc.bi_vqa(sdg, cfg, labels=labels[:6], **kw(b, 6))
    rl.backward()
    assert abs(out6[0].item() - rl.item()) / rl.item() < LOSS_RTOL
    names = ["cls.predictions.decoder.weight", "cls.predictions.bias", "cls.predictions.transform.dense.weight",
             "bert.txt_encoder.layer.5.output.dense.weight", "bert.vis_encoder.layer.0.attention.self.query.weight",
             "bert.mul_encoder.layer.3.intermediate.dense.weight", "bert.mul_encoder.layer.0.attention.output.dense.weight",
             "bert.img_embedding.weight"]
    pd = dict(model.named_parameters())
    worst = 0.0
    for n in names:
        assert pd[n].grad is not None and sdg[n].grad is not None, n
        e = _rel(pd[n].grad, sdg[n].grad)
        print("configs[4] gradient", n, "rel L2 vs oracle", e)
        worst = max(worst, e)
    check_measured("configs4:vqa_grad", max(worst, 1e-9), 6e-2)


# ------------------------------------------------------------------------------ round-3 additions
@pytest.mark.parametrize("streams", [True, False])
def test_packed_pipeline_equals_general_path(dev, streams):
    """The training fast path (BiBertImgModel.forward_packed: index maps from hip.pack_maps, rows tapped from packed
    buffers, f32 HIP heads) against the general path through padded tensors on the same weights, inputs and draws:
    same six losses and same gradients up to the summation order of bf16 / f32 atomics."""
    from mvp_pytorch_amd import modeling
    from mvp_pytorch_amd.synthetic import synthetic_batch
    cfg = dict(gu.TINY_CFG, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, parallel_stacks=streams, max_phrases=3)
    dims = dict(B=16, T=12, P=3, G=6, R=5)
    batch = synthetic_batch(dims, cfg, 21, device=dev)
    res = {}
    for packed in (False, True):
        torch.manual_seed(0)
        model = modeling.BiBertImgForPreTraining(modeling.make_config(cfg)).to(dev)
        model.train()
        model.packed_pipeline = packed
        torch.manual_seed(5)
        out = model(input_ids_a=batch["input_ids_a"], token_type_ids_a=batch["segment_ids_a"], attention_mask_a=batch["input_mask_a"],
                    masked_lm_labels_a=batch["lm_label_ids_a"], input_ids_b=batch["input_ids_b"], img_feats=batch["img_feats"],
                    token_type_ids_b=batch["segment_ids_b"], attention_mask_b=batch["input_mask_b"],
                    masked_lm_labels_b=batch["lm_label_ids_b"], phrase_index=batch["phrase_index"], img_index=batch["image_index"],
                    max_tag_length=dims["G"])
        out[0].backward()
        torch.cuda.synchronize()
        res[packed] = ([float(x) for x in out], {n: p.grad.detach().float().clone() for n, p in model.named_parameters() if p.grad is not None})
    la, lb = np.array(res[False][0]), np.array(res[True][0])
    print("general", la, "packed", lb)
    assert len(la) == len(lb) == 6
    assert np.abs(la - lb).max() / np.abs(la).max() < 5e-4
    ga, gb = res[False][1], res[True][1]
    assert set(ga) == set(gb)
    # key.bias has a mathematically zero gradient (softmax is invariant to a per-query shift): pure rounding noise
    worst = max((_rel(gb[n], ga[n]), n) for n in ga if ga[n].norm() > 1e-6 and not n.endswith("attention.self.key.bias"))
    print("worst gradient rel L2", worst)
    assert worst[0] < 1e-2, worst


@pytest.mark.parametrize("k", [0, 1])
def test_return_at_layer_and_phrase_layer(dev, k):
    """vl:162-163,176-177 (CaptionBertEncoder.forward(return_at_layer=k) -> ((final,), hidden states after layer k)) and
    vl:570-572,589-592,605-608 (BiBertImgModel.forward(phrase_layer=k) -> a fourth output (mid_joint, mid_hard)): against the
    oracle's layer loop, values and gradients through both segments."""
    from mvp_pytorch_amd import modeling
    from oracle import mvptr_oracle as orc
    cfg = dict(gu.TINY_CFG, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    torch.manual_seed(3)
    model = modeling.BiBertImgModel(modeling.make_config(cfg)).to(dev).train()
    sd = {n: p.detach().float().cpu() for n, p in model.state_dict().items()}
    enc = model.mul_encoder
    n_layers, heads = len(enc.layer), cfg["num_attention_heads"]
    g = torch.Generator().manual_seed(8)
    B, L, H = 5, 17, cfg["hidden_size"]
    x = (torch.randn(B, L, H, generator=g) * 0.5).to(torch.bfloat16)
    lens = torch.tensor([17, 9, 12, 3, 17])
    mask01 = (torch.arange(L)[None, :] < lens[:, None]).long()
    add = ((1.0 - mask01.float()) * -10000.0)
    xd = x.to(dev).requires_grad_(True)
    (final,), mid = enc(xd, add.to(dev), return_at_layer=k)
    ref_final, ref_mid = orc.encoder(sd, "mul_encoder", n_layers, x.float(), add[:, None, None, :], heads, cfg["layer_norm_eps"], return_at_layer=k)
    valid = mask01.bool()
    assert _rel(final.float().cpu()[valid], ref_final[valid]) < 2e-2
    assert _rel(mid.float().cpu()[valid], ref_mid[valid]) < 2e-2
    # same values as the plain call; gradients flow through both segments
    plain = enc(x.to(dev), add.to(dev))[0]
    assert _rel(final.float().cpu()[valid], plain.float().cpu()[valid]) < 4e-3
    (final.float()[valid.to(dev)].sum() + mid.float()[valid.to(dev)].sum()).backward()
    assert xd.grad is not None and torch.isfinite(xd.grad.float()).all()
    first = enc.layer[0].attention.self.query.weight.grad
    last = enc.layer[-1].output.dense.weight.grad
    assert first is not None and float(first.abs().sum()) > 0 and last is not None and float(last.abs().sum()) > 0
    # the backbone's fourth output
    dims = gu.TINY_DIMS
    from mvp_pytorch_amd.synthetic import synthetic_batch
    b = synthetic_batch(dict(dims, B=6), cfg, 4, device=dev)
    kw = dict(input_ids_a=b["input_ids_a"], token_type_ids_a=b["segment_ids_a"], attention_mask_a=b["input_mask_a"],
              input_ids_b=b["input_ids_b"], token_type_ids_b=b["segment_ids_b"], attention_mask_b=b["input_mask_b"],
              img_feats=b["img_feats"], max_tag_length=dims["G"], encode_hn=True)
    model.eval()
    with torch.no_grad():
        torch.manual_seed(1)
        out4 = model(phrase_layer=k, **kw)
        torch.manual_seed(1)
        out3 = model(**kw)
    assert len(out4) == 4 and len(out3) == 3
    mid_joint, mid_hard = out4[3]
    assert mid_joint.shape == out4[0][0].shape and mid_hard.shape == out4[0][2].shape
    assert torch.equal(out4[2][0], out3[2][0]) and _rel(out4[0][0], out3[0][0]) < 4e-3
    if k == n_layers - 1:
        assert torch.equal(mid_joint, out4[0][0])


def test_encoder_hidden_states_attentions_and_phase_masks(dev):
    """vl:131-178: the outputs only an inspecting caller asks for — config.output_hidden_states (input of every layer + the last
    output), config.output_attentions (softmax(QK^T/8 + mask) per layer) and a LIST of masks (one per phase of the stack, the
    first phase's output handed back as `stage_output`) — against the oracle's layer loop; tuple order as the reference's;
    the plain call's values; BertImgModel / BertImgForPreTraining pass the extras on (vl:347, vl:1116)."""
    from mvp_pytorch_amd import modeling
    from oracle import mvptr_oracle as orc
    cfg = dict(gu.TINY_CFG, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    torch.manual_seed(5)
    plain_model = modeling.BertImgModel(modeling.make_config(cfg)).to(dev).eval()
    both = modeling.BertImgModel(modeling.make_config(dict(cfg, output_hidden_states=True, output_attentions=True))).to(dev).eval()
    both.load_state_dict(plain_model.state_dict())
    sd = {n: p.detach().float().cpu() for n, p in plain_model.state_dict().items()}
    enc, enc_p = both.encoder, plain_model.encoder
    n, heads, eps = len(enc.layer), cfg["num_attention_heads"], cfg["layer_norm_eps"]
    g = torch.Generator().manual_seed(9)
    B, L, H = 4, 21, cfg["hidden_size"]
    x = (torch.randn(B, L, H, generator=g) * 0.5).to(torch.bfloat16)
    lens = torch.tensor([21, 9, 14, 3])
    mask01 = (torch.arange(L)[None, :] < lens[:, None]).long()
    add = (1.0 - mask01.float()) * -10000.0
    valid = mask01.bool()

    def ref_probs(prefix, h):
        d = H // heads
        q = orc.linear(sd, prefix + ".query", h).view(B, L, heads, d).permute(0, 2, 1, 3)
        k = orc.linear(sd, prefix + ".key", h).view(B, L, heads, d).permute(0, 2, 1, 3)
        return torch.softmax(q @ k.transpose(-1, -2) / d ** 0.5 + add[:, None, None, :], dim=-1)

    with torch.no_grad():
        out = enc(x.to(dev), add.to(dev))
        plain = enc_p(x.to(dev), add.to(dev))[0]
    assert len(out) == 3 and len(out[1]) == n + 1 and len(out[2]) == n
    assert torch.equal(out[0], out[1][-1]) and torch.equal(out[1][0].cpu(), x)
    assert _rel(out[0].float().cpu()[valid], plain.float().cpu()[valid]) < 4e-3
    h = x.float()
    for i in range(n):
        # each layer against the oracle's layer on the SAME (bf16) input, so errors do not accumulate across the comparison
        h_in = out[1][i].float().cpu()
        pr = ref_probs("encoder.layer.%d.attention.self" % i, h_in)
        got = out[2][i].cpu()
        assert got.shape == (B, heads, L, L) and got.dtype == torch.float32
        assert float((got - pr).abs().max()) < 2e-2 and float((got.sum(-1) - 1).abs().max()) < 1e-5
        assert float((got * (1 - mask01)[:, None, None, :].float()).max()) == 0.0       # padded keys weigh exp(-10000) = 0
        ref_out = orc.encoder_layer(sd, "encoder.layer.%d" % i, h_in, add[:, None, None, :], heads, eps)
        assert _rel(out[1][i + 1].float().cpu()[valid], ref_out[valid]) < 2e-2, i
        h = orc.encoder_layer(sd, "encoder.layer.%d" % i, h, add[:, None, None, :], heads, eps)
    assert _rel(out[0].float().cpu()[valid], h[valid]) < 3e-2
    # only one of the two flags: tuple positions as vl:170-175
    enc.output_attentions = False
    with torch.no_grad():
        o = enc(x.to(dev), add.to(dev))
    assert len(o) == 2 and len(o[1]) == n + 1 and torch.equal(o[0], out[0])
    enc.output_attentions, enc.output_hidden_states = True, False
    with torch.no_grad():
        o = enc(x.to(dev), add.to(dev))
    assert len(o) == 2 and len(o[1]) == n and torch.equal(o[1][0], out[2][0])
    enc.output_hidden_states = True
    # phase masks: first ceil(n / 2) layers see every key, the rest the padded mask; + return_at_layer
    open_add = torch.zeros_like(add)
    per = -(-n // 2)
    with torch.no_grad():
        (final, stage), mid = enc_p(x.to(dev), [open_add.to(dev), add.to(dev)], return_at_layer=0)
    h = x.float()
    for i in range(n):
        h = orc.encoder_layer(sd, "encoder.layer.%d" % i, h, (open_add if i < per else add)[:, None, None, :], heads, eps)
        if i == 0:
            assert _rel(mid.float().cpu(), h) < 2e-2
        if i == per - 1:
            assert _rel(stage.float().cpu(), h) < 3e-2
    assert _rel(final.float().cpu()[valid], h[valid]) < 3e-2
    # gradients reach the first layer through the collected states
    both.train()
    xd = x.to(dev).requires_grad_(True)
    o = enc(xd, add.to(dev))
    sum(t.float()[valid.to(dev)].sum() for t in o[1][1:]).backward()
    w0 = enc.layer[0].attention.self.query.weight.grad
    assert xd.grad is not None and w0 is not None and float(w0.abs().sum()) > 0
    both.eval()
    # the models hand the extras on
    from mvp_pytorch_amd.synthetic import synthetic_batch
    b = synthetic_batch(dict(gu.TINY_DIMS, B=3), cfg, 4, single_stream=True, device=dev)
    kw = dict(input_ids=b["input_ids"], token_type_ids=b["segment_ids"], attention_mask=b["input_mask"], img_feats=b["img_feats"])
    with torch.no_grad():
        o = both(**kw)
        o_plain = plain_model(**kw)
    Lt = b["input_mask"].shape[1]
    assert len(o) == 4 and len(o_plain) == 2 and len(o[2]) == n + 1 and o[3][0].shape == (3, heads, Lt, Lt)
    v = b["input_mask"].bool()
    assert _rel(o[0].float()[v], o_plain[0].float()[v]) < 4e-3
    pre = modeling.BertImgForPreTraining(modeling.make_config(dict(cfg, output_hidden_states=True, max_text_seq_length=gu.TINY_DIMS["T"]))).to(dev).eval()
    with torch.no_grad():
        o = pre(masked_lm_labels=b["lm_label_ids"], next_sentence_label=b["is_next"], **kw)
        o2 = pre(**kw)
    assert len(o) == 5 and len(o[3]) == n + 1 and o[4].dim() == 0 and len(o2) == 3 and len(o2[2]) == n + 1
    # what stays outside: dropout on the probabilities leaves no L x L mask to hand back
    both.train()
    enc.layer[0].attention.self.dropout.p = 0.1
    with pytest.raises(NotImplementedError):
        enc(x.to(dev), add.to(dev))


@pytest.mark.parametrize("name", ["tiny_bi_pretrain_nophrase", "cfg1_bi_pretrain_nophrase"])
def test_packed_training_path_matches_reference(dev, name):
    """the path bench.py times — model.train(), row-packed pipeline (_forward_packed + forward_packed +
    MultiTapFn), HIP ITM / contrastive heads, heads on the second stream — against the REFERENCE directly: the fixture is
    the reference's 5-tuple with phrase_index=None (vl:1309; nothing but the hard-negative permutation is drawn), dropout
    0 on both sides.  Losses to north_star's 1e-3, gradient norms / full gradients within twice their measured error,
    hard-negative indices bit-exact (free-running on the BERT-base fixture, whose margin allows it)."""
    d = gu.load(name)
    cfg, dims = d["config"], d["dims"]
    model, sd = _build("BiBertImgForPreTraining", cfg, int(d["seed"]), dev, train=True, gain=float(d["weight_gain"]))
    model.wra_on_device = True                   # the product default; no phrase_index here anyway
    assert model.training and model.packed_pipeline and model.heads_beside == 2
    kw = _bi_inputs(d, dev)
    t = lambda k: torch.from_numpy(d["in:" + k]).to(dev)  # noqa: E731
    calls = {"packed": 0, "hard": None}
    orig_fp, orig_bp = model._forward_packed, model.bert.forward_packed

    def spy_fp(*a, **k):
        calls["packed"] += 1
        return orig_fp(*a, **k)

    def spy_bp(*a, **k):
        out = orig_bp(*a, **k)
        calls["hard"] = (out["hard_txt_full"].cpu().numpy(), out["hard_img_full"].cpu().numpy())
        return out

    model._forward_packed, model.bert.forward_packed = spy_fp, spy_bp
    sim_ref = torch.from_numpy(d["sim_mat"])
    masked = sim_ref - 2 * torch.eye(sim_ref.shape[0])
    free_running = name.startswith("cfg1")
    import contextlib
    inject = contextlib.nullcontext() if free_running else gu.InjectHard(model.bert, masked.max(1)[1], masked.max(0)[1])
    with Replay(d, dev), inject:
        res = model(masked_lm_labels_a=t("lm_label_ids_a"), masked_lm_labels_b=t("lm_label_ids_b"), max_tag_length=dims["G"], **kw)
    assert calls["packed"] == 1, "the training step did not take the row-packed pipeline"
    assert len(res) == 5
    assert np.array_equal(calls["hard"][0], d["hard_txt_index"]) and np.array_equal(calls["hard"][1], d["hard_img_index"])
    got = np.array([x.item() for x in res])
    ref = d["losses"]
    rel = np.abs(got - ref) / np.abs(ref)
    print(name, "losses", got, "ref", ref, "rel", rel)
    assert max(rel[0], rel[1], rel[3]) < LOSS_RTOL, rel
    assert max(rel[2], rel[4]) < SMALL_ROWS_RTOL, rel            # contrastive, ITM (B = 4 rows)
    res[0].backward()
    torch.cuda.synchronize()
    worst_n = worst_g = 0.0
    for pname, p in model.named_parameters():
        key = "gnorm:" + pname
        if key not in d:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, pname
            continue
        gn, rn = p.grad.double().norm().item(), float(d[key])
        if pname == "logit_scale":
            assert abs(gn - rn) < 5e-3, (pname, gn, rn)
        elif rn > 1e-6:
            e = abs(gn - rn) / rn
            if pname in CLIP_BRANCH:
                if not name.startswith("tiny"):
                    assert e < 1e-1, (pname, gn, rn)
            else:
                worst_n = max(worst_n, e)
        else:
            assert gn < 2e-3, (pname, gn, rn)
            continue
        full = "grad:" + pname
        if full in d and pname != "logit_scale" and not (pname in CLIP_BRANCH and name.startswith("tiny")):
            e = _rel(p.grad, torch.from_numpy(d[full]))
            if pname in CLIP_BRANCH:
                assert e < 1e-1, (pname, e)
            else:
                worst_g = max(worst_g, e)
    check_measured(name + ":train_gnorm", max(worst_n, 1e-9), 6e-2)
    check_measured(name + ":train_grad", max(worst_g, 1e-9), 5e-2)


def test_sync_free_joint_pass_and_host_counts(dev):
    """the training step without count read-backs — the input-only counts come with the batch
    (synthetic.host_counts), the joint + hard-negative pass is sized for its bound and clamps to a DEVICE-side row
    count (mvptr_layer_desc.rows_dev; launches planned for the previous step's count).  Same losses and gradients as
    the pass that waits for its counts, over three steps (the third one runs with the planning hint), and
    engine.AsyncCounts is never awaited inside the step."""
    from mvp_pytorch_amd import engine, modeling
    from mvp_pytorch_amd.synthetic import synthetic_batch
    cfg = dict(gu.TINY_CFG, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, max_phrases=3)
    dims = dict(B=16, T=12, P=3, G=6, R=5)
    batch = synthetic_batch(dims, cfg, 33, device=dev)
    perm = torch.randperm(dims["B"], generator=torch.Generator().manual_seed(9))
    res = {}
    waits = {}
    orig_get = engine.AsyncCounts.get
    for mode in ("sized", "free"):
        torch.manual_seed(0)
        model = modeling.BiBertImgForPreTraining(modeling.make_config(dict(cfg, sync_free_joint=(mode == "free")))).to(dev)
        model.train()
        assert model.bert.sync_free_joint == (mode == "free")
        kw = dict(input_ids_a=batch["input_ids_a"], token_type_ids_a=batch["segment_ids_a"], attention_mask_a=batch["input_mask_a"],
                  masked_lm_labels_a=batch["lm_label_ids_a"], input_ids_b=batch["input_ids_b"], img_feats=batch["img_feats"],
                  token_type_ids_b=batch["segment_ids_b"], attention_mask_b=batch["input_mask_b"], masked_lm_labels_b=batch["lm_label_ids_b"],
                  max_tag_length=dims["G"], host_counts=batch["host_counts"] if mode == "free" else None)
        n_wait = [0]

        def counting_get(self, _n=n_wait):
            if not self.ready():
                _n[0] += 1
            return orig_get(self)

        engine.AsyncCounts.get = counting_get
        try:
            for step in range(3):
                model.zero_grad(set_to_none=True)
                with Replay(dict(draw_randperm=[perm.numpy()]), dev):
                    out = model(**kw)
                out[0].backward()
                torch.cuda.synchronize()
        finally:
            engine.AsyncCounts.get = orig_get
        waits[mode] = n_wait[0]
        res[mode] = ([float(x) for x in out], {n: p.grad.detach().float().clone() for n, p in model.named_parameters() if p.grad is not None})
    print("count read-backs awaited inside the three steps:", waits)
    assert waits["free"] == 0
    la, lb = np.array(res["sized"][0]), np.array(res["free"][0])
    print("sized", la, "sync-free", lb)
    assert np.abs(la - lb).max() / np.abs(la).max() < 5e-4
    ga, gb = res["sized"][1], res["free"][1]
    assert set(ga) == set(gb)
    worst = max((_rel(gb[n], ga[n]), n) for n in ga if ga[n].norm() > 1e-6 and not n.endswith("attention.self.key.bias"))
    print("worst gradient rel L2", worst)
    assert worst[0] < 1e-2, worst
    # (host_counts whose row / longest-sequence numbers do not describe the batch trap on the device — every buffer behind them
    #  is sized from them — and are not exercised here; bert.verify_host_counts checks on the host instead, below)
    model.bert.verify_host_counts = True
    hc = batch["host_counts"]
    bad = dict(hc, rows_a=int(hc["rows_a"]) - 1)
    with pytest.raises(ValueError, match="host_counts"):
        model(**dict(kw, host_counts=bad))
    model.bert.verify_host_counts = False
    # too few slots for the scored rows: the step runs (the surplus rows drop out of the loss), the device error word is set and
    # the host raises at its next look — catchably, the process stays alive (ABI 5-6: trap)
    from mvp_pytorch_amd import hip
    hip.check_device_errors(dev)
    short = dict(hc, scored_a=max(int(hc["scored_a"]) - 2, 1))
    with Replay(dict(draw_randperm=[perm.numpy()]), dev):
        out = model(**dict(kw, host_counts=short))
    assert torch.isfinite(out[0]).all()
    with pytest.raises(RuntimeError, match="more scored"):
        hip.check_device_errors(dev)


@pytest.mark.parametrize("bound,rows,plan", [(20000, 9000, 4500), (9000, 9000, 4500), (3000, 1100, 550),
                                             (44000, 26000, 26000), (44000, 20000, 26000), (44000, 30000, 23000)])
def test_encoder_stack_with_device_side_row_count(dev, bound, rows, plan):
    """mvptr_layer_desc.rows_dev at BERT-base width: one encoder layer over `rows` packed rows inside buffers sized for
    `bound`, the count living on the device, against the same layer run on exactly `rows` rows — output rows, input
    gradient and every weight gradient (the 256 x 256 GEMM tiles, the weight-gradient M-splits and the LayerNorm
    kernels all clamp to the device count).  The tile height of the GEMMs (gemm_nt.hip launch(): 256 or 192 rows) follows
    the PLANNED rows; the 44 000-row cases run with as many, fewer and more rows on the device than planned."""
    from mvp_pytorch_amd import modeling
    cfg = dict(gu.BASE_CFG, num_hidden_layers=1, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    torch.manual_seed(0)
    enc = modeling.modeling_vlbert.CaptionBertEncoder(modeling.make_config(cfg)).to(dev).train()
    g = torch.Generator().manual_seed(rows)
    L = 100
    lens = torch.randint(20, L + 1, (bound // 20,), generator=g)
    cum = torch.cumsum(lens, 0)
    nseq = int((cum <= rows).sum())
    lens = lens[:nseq].clone()
    lens[-1] += rows - int(lens.sum())            # exactly `rows` rows
    assert int(lens.sum()) == rows and int(lens.max()) <= 2 * L
    starts = (torch.cumsum(lens, 0) - lens).to(torch.int32).to(dev)
    lens_d = lens.to(torch.int32).to(dev)
    x = (torch.randn(rows, 768, generator=g) * 0.5).to(torch.bfloat16).to(dev)
    dy = (torch.randn(rows, 768, generator=g) * 0.1).to(torch.bfloat16).to(dev)
    lmax = int(lens.max())

    def run(use_dev):
        enc.zero_grad(set_to_none=True)
        if use_dev:
            xin = torch.zeros(bound, 768, dtype=torch.bfloat16, device=dev)
            xin[:rows] = x
            xin.requires_grad_(True)
            cnt = torch.tensor([rows, lmax], dtype=torch.int64, device=dev)
            lb = max(lmax, -(-bound // nseq))       # the bound must fit n_seq x L (the length is an upper bound here too)
            assert lb <= 256
            y = enc.forward_rows(xin, starts, lens_d, nseq, lb, rows_dev=cnt, rows_plan=plan)
            d = torch.zeros(bound, 768, dtype=torch.bfloat16, device=dev)
            d[:rows] = dy
            y.backward(d)
            return y[:rows].float(), xin.grad[:rows].float(), {n: p.grad.float().clone() for n, p in enc.named_parameters()}
        xin = x.clone().requires_grad_(True)
        y = enc.forward_rows(xin, starts, lens_d, nseq, lmax)
        y.backward(dy)
        return y.float(), xin.grad.float(), {n: p.grad.float().clone() for n, p in enc.named_parameters()}

    y0, dx0, g0 = run(False)
    y1, dx1, g1 = run(True)
    torch.cuda.synchronize()
    assert _rel(y1, y0) < 2e-3 and _rel(dx1, dx0) < 4e-3, (_rel(y1, y0), _rel(dx1, dx0))
    for n in g0:
        if g0[n].norm() > 1e-6 and not n.endswith("attention.self.key.bias"):
            assert _rel(g1[n], g0[n]) < 4e-3, (n, _rel(g1[n], g0[n]))


@pytest.mark.parametrize("layers,rows,dropout,use_dev", [(3, 5000, 0.0, False), (2, 9000, 0.1, True), (1, 700, 0.0, False)])
def test_deferred_stack_weight_gradients_equal_per_layer(dev, layers, rows, dropout, use_dev):
    """Round 5: EncoderFn keeps every layer's dY operands alive and issues the weight gradients of the whole stack as ONE
    balanced launch (mvptr_encoder_layer_bwd_defer + mvptr_gemm_tn_stack; the bias gradients of intermediate.dense and of
    Q/K/V ride on it) — same input gradient bit for bit, same weight / bias gradients to f32 summation order as the per-layer
    grouped launches (engine.DEFER_WGRAD = False), with dropout (same seeds: same masks) and with a device-side row count."""
    from mvp_pytorch_amd import engine, modeling
    cfg = dict(gu.BASE_CFG, num_hidden_layers=layers, hidden_dropout_prob=dropout, attention_probs_dropout_prob=dropout)
    torch.manual_seed(0)
    enc = modeling.modeling_vlbert.CaptionBertEncoder(modeling.make_config(cfg)).to(dev).train()
    g = torch.Generator().manual_seed(rows)
    L = 90
    lens = torch.randint(10, L + 1, (rows // 10,), generator=g)
    nseq = int((torch.cumsum(lens, 0) <= rows).sum())
    lens = lens[:nseq].clone()
    lens[-1] += rows - int(lens.sum())
    starts = (torch.cumsum(lens, 0) - lens).to(torch.int32).to(dev)
    lens_d = lens.to(torch.int32).to(dev)
    lmax = int(lens.max())
    bound = rows + 1500 if use_dev else rows
    x = torch.zeros(bound, 768, dtype=torch.bfloat16, device=dev)
    x[:rows] = (torch.randn(rows, 768, generator=g) * 0.5).to(torch.bfloat16).to(dev)
    dy = torch.zeros(bound, 768, dtype=torch.bfloat16, device=dev)
    dy[:rows] = (torch.randn(rows, 768, generator=g) * 0.1).to(torch.bfloat16).to(dev)

    def run(defer):
        prev = engine.DEFER_WGRAD
        engine.DEFER_WGRAD = defer
        try:
            enc.zero_grad(set_to_none=True)
            torch.manual_seed(1234)            # dropout seeds come from torch's generator
            engine._seed_counter[0] = 0x5DEECE66D
            xin = x.clone().requires_grad_(True)
            if use_dev:
                cnt = torch.tensor([rows, lmax], dtype=torch.int64, device=dev)
                lb = max(lmax, -(-bound // nseq))
                y = enc.forward_rows(xin, starts, lens_d, nseq, lb, rows_dev=cnt, rows_plan=rows // 2)
            else:
                y = enc.forward_rows(xin, starts, lens_d, nseq, lmax)
            y.backward(dy)
            torch.cuda.synchronize()
            return xin.grad[:rows].clone(), {n: p.grad.float().clone() for n, p in enc.named_parameters()}
        finally:
            engine.DEFER_WGRAD = prev

    dx0, g0 = run(False)
    dx1, g1 = run(True)
    assert torch.equal(dx0, dx1)
    assert set(g0) == set(g1) and len(g0) == 16 * layers
    for n in g0:
        if n.endswith("attention.self.key.bias"):       # mathematically zero: rounding noise on both sides
            continue
        assert g0[n].norm() > 0, n
        # intermediate.dense.bias: the per-layer path sums dU in the GELU-backward epilogue BEFORE its bf16 rounding, the
        # deferred path sums the bf16 dU the weight gradient reads (column sums on the weight-gradient kernel): one bf16
        # rounding per summand apart (measured 1.6e-3)
        tol = 5e-3 if n.endswith("intermediate.dense.bias") else 2e-5
        assert _rel(g1[n], g0[n]) < tol, (n, _rel(g1[n], g0[n]))


def test_deferred_weight_gradients_fall_back_when_memory_is_short(dev):
    """the deferred mode keeps one backward workspace per layer alive; a stack that may not (byte cap,
    engine.DEFER_WGRAD_MAX_BYTES) or cannot (allocation failure) have that takes the per-layer launches and gets their gradients."""
    from mvp_pytorch_amd import engine, modeling
    cfg = dict(gu.BASE_CFG, num_hidden_layers=2, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    torch.manual_seed(0)
    enc = modeling.modeling_vlbert.CaptionBertEncoder(modeling.make_config(cfg)).to(dev).train()
    g = torch.Generator().manual_seed(6)
    B, L = 24, 64
    x = (torch.randn(B * L, 768, generator=g) * 0.5).to(torch.bfloat16).to(dev)
    dy = (torch.randn(B * L, 768, generator=g) * 0.1).to(torch.bfloat16).to(dev)
    starts = (torch.arange(B, dtype=torch.int32) * L).to(dev)
    lens = torch.full((B,), L, dtype=torch.int32, device=dev)

    def run(defer, cap=None, fail_alloc=False):
        prev = (engine.DEFER_WGRAD, engine.DEFER_WGRAD_MAX_BYTES)
        engine.DEFER_WGRAD, engine.DEFER_WGRAD_MAX_BYTES = defer, cap
        real_empty = torch.empty
        state = {"failed": 0}

        def empty(*a, **k):      # the first large byte buffer (the per-layer workspaces) does not fit
            if fail_alloc and not state["failed"] and k.get("dtype") == torch.uint8 and a and isinstance(a[0], int) and a[0] > (1 << 20):
                state["failed"] = 1
                raise torch.OutOfMemoryError("simulated")
            return real_empty(*a, **k)
        try:
            enc.zero_grad(set_to_none=True)
            xin = x.clone().requires_grad_(True)
            y = enc.forward_rows(xin, starts, lens, B, L)
            torch.empty = empty
            try:
                y.backward(dy)
            finally:
                torch.empty = real_empty
            torch.cuda.synchronize()
            assert (not fail_alloc) or state["failed"] == 1
            return engine.EncoderFn.last_backward_deferred, xin.grad.clone(), {n: p.grad.clone() for n, p in enc.named_parameters()}
        finally:
            engine.DEFER_WGRAD, engine.DEFER_WGRAD_MAX_BYTES = prev

    was0, dx0, g0 = run(False)
    was1, dx1, g1 = run(True)
    was2, dx2, g2 = run(True, cap=1 << 20)
    was3, dx3, g3 = run(True, fail_alloc=True)
    assert (was0, was1, was2, was3) == (False, True, False, False)
    for dx, gg in ((dx2, g2), (dx3, g3)):
        assert torch.equal(dx, dx0)
        for n in g0:
            # the same launches as DEFER_WGRAD = False (their split-K partial sums arrive in any order: f32 summation order apart)
            assert g0[n].norm() == 0 or _rel(gg[n], g0[n]) < 2e-5, n
    assert torch.equal(dx1, dx0)


def test_deferred_weight_gradients_reach_non_f32_parameters(dev):
    """with deferred weight gradients the scratch arena of a layer whose parameters are NOT f32 is only
    complete after the stack-wide launch; the conversion to the parameter's dtype used to run inside the layer loop and handed
    autograd zeros for Wqkv / bqkv / Wo / Wi / bi / Wout.  A bf16-parameter stack must get the gradients the per-layer path gives."""
    from mvp_pytorch_amd import engine, modeling
    cfg = dict(gu.BASE_CFG, num_hidden_layers=2, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    torch.manual_seed(0)
    enc = modeling.modeling_vlbert.CaptionBertEncoder(modeling.make_config(cfg)).to(dev).to(torch.bfloat16).train()
    g = torch.Generator().manual_seed(5)
    B, L = 40, 64
    x = (torch.randn(B * L, 768, generator=g) * 0.5).to(torch.bfloat16).to(dev)
    dy = (torch.randn(B * L, 768, generator=g) * 0.1).to(torch.bfloat16).to(dev)
    starts = (torch.arange(B, dtype=torch.int32) * L).to(dev)
    lens = torch.full((B,), L, dtype=torch.int32, device=dev)

    def run(defer):
        prev = engine.DEFER_WGRAD
        engine.DEFER_WGRAD = defer
        try:
            enc.zero_grad(set_to_none=True)
            xin = x.clone().requires_grad_(True)
            enc.forward_rows(xin, starts, lens, B, L).backward(dy)
            torch.cuda.synchronize()
            return {n: p.grad.float().clone() for n, p in enc.named_parameters()}
        finally:
            engine.DEFER_WGRAD = prev

    g0, g1 = run(False), run(True)
    assert set(g0) == set(g1) and len(g0) == 32
    for n in g0:
        if n.endswith("attention.self.key.bias"):
            continue
        assert g0[n].norm() > 0 and g1[n].norm() > 0, n
        assert all(p.grad.dtype == torch.bfloat16 for p in enc.parameters())
        assert _rel(g1[n], g0[n]) < 8e-3, (n, _rel(g1[n], g0[n]))


def test_gelu_stash_formats_agree(dev):
    """config.gelu_stash = "bf16" (the stash format of rounds 1-3, selectable) against the default 8-bit fixed
    point on the tiny two-stage training fixture, same weights and inputs, dropout off: identical losses (the forward pass only
    differs in what it SAVES), gradients equal to what the 8-bit grid allows (|error of gelu'| < 0.005, zero mean, on values of
    order 0.1 - 1)."""
    d = gu.load("tiny_bi_pretrain_nophrase")
    cfg, dims = d["config"], d["dims"]
    t = lambda k: torch.from_numpy(d["in:" + k]).to(dev)  # noqa: E731
    outs = {}
    for fmt in ("u8", "bf16"):
        model, _ = _build("BiBertImgForPreTraining", dict(cfg, gelu_stash=fmt), int(d["seed"]), dev, train=True, gain=float(d["weight_gain"]))
        model.wra_on_device = True
        assert model.bert.txt_encoder.gelu_stash_bf16 == (fmt == "bf16") and model.cls.predictions.transform._act == ("gelu16" if fmt == "bf16" else "gelu")
        with Replay(d, dev):
            res = model(masked_lm_labels_a=t("lm_label_ids_a"), masked_lm_labels_b=t("lm_label_ids_b"), max_tag_length=dims["G"], **_bi_inputs(d, dev))
        res[0].backward()
        torch.cuda.synchronize()
        outs[fmt] = ([float(x) for x in res], {n: p.grad.float().clone() for n, p in model.named_parameters() if p.grad is not None})
    (l8, g8), (l16, g16) = outs["u8"], outs["bf16"]
    assert l8 == l16
    assert set(g8) == set(g16)
    worst = max((_rel(g8[n], g16[n]), n) for n in g16 if g16[n].norm() > 1e-6 and n not in CLIP_BRANCH and n != "logit_scale" and
                not n.endswith("attention.self.key.bias"))
    print("gelu' stash u8 vs bf16: worst gradient rel L2", worst)
    assert worst[0] < 1.5e-2, worst


def test_encoder_stack_with_dropout_matches_oracle_with_replayed_masks(dev):
    """the timed step runs with dropout 0.1, the reference fixtures with dropout 0.  Here a row-packed
    BERT-base encoder stack (2 layers, sequences of 20 - 110 rows) runs in training mode with dropout 0.1 at its three sites
    per layer (attention probabilities vl:90, attention-output dense mb:350, FFN-output dense mb:409); the per-layer seeds are
    pinned, the kernels' own keep masks are materialised with mvptr_dropout_mask (element indices as include/mvptr.h documents
    them: m * H + n on the packed rows, ((seq * heads + h) * Lmax + q) * Lp + key for the probabilities) and fed to the oracle's
    encoder: same outputs on the valid rows, same input / weight gradients to bf16 accuracy — i.e. dropout is applied to the
    right elements with the right scale, in the forward AND the backward pass (regenerated, not stored)."""
    from mvp_pytorch_amd import engine, hip, modeling
    layers, heads, H = 2, 12, 768
    p_drop = 0.1
    cfg = dict(gu.BASE_CFG, num_hidden_layers=layers, hidden_dropout_prob=p_drop, attention_probs_dropout_prob=p_drop)
    torch.manual_seed(0)
    enc = modeling.modeling_vlbert.CaptionBertEncoder(modeling.make_config(cfg)).to(dev).train()
    g = torch.Generator().manual_seed(3)
    lens = torch.randint(20, 111, (9,), generator=g)
    B, lmax, rows = lens.numel(), int(lens.max()), int(lens.sum())
    starts = (torch.cumsum(lens, 0) - lens).to(torch.int32)
    x = (torch.randn(rows, H, generator=g) * 0.5).to(torch.bfloat16)
    dy = (torch.randn(rows, H, generator=g) * 0.1).to(torch.bfloat16)
    seeds = [0x1234567812345678 + 977 * i for i in range(layers)]
    it = iter(seeds)
    orig = engine.next_seed
    engine.next_seed = lambda: next(it)
    try:
        xin = x.to(dev).requires_grad_(True)
        y = enc.forward_rows(xin, starts.to(dev), lens.to(torch.int32).to(dev), B, lmax)
        y.backward(dy.to(dev))
        torch.cuda.synchronize()
    finally:
        engine.next_seed = orig

    # the kernels' masks: layer.hip site_drop(seed, site): site 0 attention probabilities, 1 attention output, 2 FFN output
    def site(seed, k):
        d = hip.Dropout()
        d.seed_lo = ((seed & 0xffffffff) ^ ((0x9E3779B9 * (k + 1)) & 0xffffffff)) & 0xffffffff
        d.seed_hi = ((seed >> 32) + 0x85EBCA6B * (k + 1)) & 0xffffffff
        d.thresh16 = int(round(p_drop * 65536.0))
        d.pad_ = 0
        return d

    scale = 65536.0 / (65536.0 - int(round(p_drop * 65536.0)))
    Lp = (lmax + 31) // 32 * 32
    pad_idx = torch.cat([torch.arange(int(n)) + b * lmax for b, n in enumerate(lens.tolist())])       # packed row -> slot of [B, lmax]
    drops = []
    for s in seeds:
        attn = hip.dropout_mask(site(s, 0), B * heads * lmax * Lp, dev).reshape(B, heads, lmax, Lp)[..., :lmax].float().cpu()
        dense = []
        for k in (1, 2):
            keep_rows = hip.dropout_mask(site(s, k), rows * H, dev).reshape(rows, H).float().cpu()
            keep = torch.ones(B * lmax, H)
            keep[pad_idx] = keep_rows
            dense.append(keep.view(B, lmax, H))
        drops.append({"attn": (attn, scale), "attn_out": (dense[0], scale), "ffn_out": (dense[1], scale)})
        assert abs(float(attn.mean()) - 0.9) < 0.01 and abs(float(dense[0].view(-1, H)[pad_idx].mean()) - 0.9) < 0.01
    # oracle on the padded layout, f32, bf16-rounded weights (what the kernels multiply with)
    sd = {"enc." + k: v.detach().float().cpu().to(torch.bfloat16).float().requires_grad_(True) if v.dim() == 2 else v.detach().float().cpu().requires_grad_(True)
          for k, v in enc.state_dict().items()}
    xp = torch.zeros(B * lmax, H)
    xp[pad_idx] = x.float()
    xp = xp.view(B, lmax, H).requires_grad_(True)
    mask = torch.zeros(B, lmax, dtype=torch.long)
    for b, n in enumerate(lens.tolist()):
        mask[b, :n] = 1
    torch.set_num_threads(min(16, torch.get_num_threads()))
    yo = orc.encoder(sd, "enc", layers, xp, orc.extended_mask(mask), heads, cfg["layer_norm_eps"], drops=drops)
    dyp = torch.zeros(B * lmax, H)
    dyp[pad_idx] = dy.float()
    yo.backward(dyp.view(B, lmax, H))
    e_y = _rel(y.detach().float().cpu(), yo.detach().view(-1, H)[pad_idx])
    e_dx = _rel(xin.grad.float().cpu(), xp.grad.view(-1, H)[pad_idx])
    print("dropout replay: y rel L2 %.4f, dx rel L2 %.4f" % (e_y, e_dx))
    assert e_y < 8e-3 and e_dx < 2.5e-2, (e_y, e_dx)
    worst = 0.0
    for n, p in enc.named_parameters():
        if n.endswith("attention.self.key.bias"):
            continue
        e = _rel(p.grad.float().cpu(), sd["enc." + n].grad)
        worst = max(worst, e)
        assert e < 4e-2, (n, e)
    print("dropout replay: worst weight-gradient rel L2 %.4f" % worst)
    # and the masks matter: without them the oracle is far away
    y0 = orc.encoder({k: v.detach() for k, v in sd.items()}, "enc", layers, xp.detach(), orc.extended_mask(mask), heads, cfg["layer_norm_eps"])
    assert _rel(y.detach().float().cpu(), y0.view(-1, H)[pad_idx]) > 5 * e_y


@pytest.mark.parametrize("packed", [True, False])
def test_layernorm_folded_inference_stack_matches_oracle(dev, packed):
    """NS-1 (north_star "fused LayerNorm + QKV projection"): in a no-grad eval forward with config.fold_layernorm the stack runs
    engine.encoder_infer_folded — every LayerNorm (mb:348-352, 407-411) folded into the GEMMs around it, one LayerNorm launch per
    stack — and agrees with the oracle's encoder (BertLayerNorm in f32, mb:242-246) as closely as the unfused product path does."""
    from mvp_pytorch_amd import engine, modeling
    layers, heads, H = 3, 12, 768
    cfg = dict(gu.BASE_CFG, num_hidden_layers=layers, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    torch.manual_seed(1)
    conf = modeling.make_config(cfg)
    enc = modeling.modeling_vlbert.CaptionBertEncoder(conf).to(dev).eval()
    with torch.no_grad():      # LayerNorm weights / biases away from (1, 0): the fold has to carry them
        for n, p in enc.named_parameters():
            if "LayerNorm.weight" in n:
                p.add_(torch.randn_like(p) * 0.2)
            elif "LayerNorm.bias" in n:
                p.add_(torch.randn_like(p) * 0.2)
    g = torch.Generator().manual_seed(4)
    lens = torch.randint(15, 120, (11,), generator=g)
    B, lmax = lens.numel(), int(lens.max())
    x = (torch.randn(B, lmax, H, generator=g) * 0.6).to(torch.bfloat16)
    mask = torch.zeros(B, lmax, dtype=torch.long)
    for b, n in enumerate(lens.tolist()):
        mask[b, :n] = 1
    add = orc.extended_mask(mask).view(B, lmax).to(dev)
    valid = mask.bool()

    calls = []
    orig = engine.encoder_infer_folded
    engine.encoder_infer_folded = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    try:
        with torch.no_grad():
            enc.unpad = packed
            enc.fold_layernorm = False
            y_unf = enc(x.to(dev), add)[0].float().cpu()
            enc.fold_layernorm = True
            y_fold = enc(x.to(dev), add)[0].float().cpu()
            assert len(calls) == 1, "the folded path did not run"
            enc.train()                     # training never folds: the LayerNorm output is an operand of the weight gradients
            enc(x.to(dev), add)
            enc.eval()
        assert len(calls) == 1
        enc(x.to(dev), add)                 # autograd on: not folded either
        assert len(calls) == 1
    finally:
        engine.encoder_infer_folded = orig
    sd = {"enc." + k: (v.detach().float().cpu().to(torch.bfloat16).float() if v.dim() == 2 else v.detach().float().cpu()) for k, v in enc.state_dict().items()}
    torch.set_num_threads(min(16, torch.get_num_threads()))
    with torch.no_grad():
        yo = orc.encoder(sd, "enc", layers, x.float(), orc.extended_mask(mask), heads, cfg["layer_norm_eps"])
    e_fold, e_unf = _rel(y_fold[valid], yo[valid]), _rel(y_unf[valid], yo[valid])
    print("LayerNorm-folded inference stack (packed=%s): rel L2 vs oracle folded %.2e, unfused %.2e, folded vs unfused %.2e"
          % (packed, e_fold, e_unf, _rel(y_fold[valid], y_unf[valid])))
    assert e_fold < 8e-3 and e_fold < 1.5 * e_unf + 1e-3, (e_fold, e_unf)
