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
        self.dev = dev

    def __enter__(self):
        self._orig = (torch.randperm, torch.randint, random.choice)
        o_int = self._orig[1]

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


def _rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / (b.norm() + 1e-12)).item()


def _build(cls_name, cfg, seed, dev, train=False):
    from mvp_pytorch_amd import modeling
    cfg = dict(cfg, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    model = getattr(modeling, cls_name)(modeling.make_config(cfg))
    sd = {k: torch.from_numpy(v) for k, v in gu.det_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed).items()}
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


@pytest.mark.parametrize("name", ["tiny_bi_pretrain", "cfg1_bi_pretrain"])
def test_bi_pretrain_parity(dev, name):
    d = gu.load(name)
    cfg, dims = d["config"], d["dims"]
    model, sd = _build("BiBertImgForPreTraining", cfg, int(d["seed"]), dev)
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
    if float(d["argmax_margin"]) > 4 * sim_err:
        assert same_t and same_i
    # 2) losses + gradients with the reference's captured draws (and indices) injected
    n = sim_ref.shape[0]
    masked = sim_ref - 2 * torch.eye(n)
    model.bert.hard_override = (masked.max(1)[1], masked.max(0)[1])
    with Replay(d, dev):
        res = model(masked_lm_labels_a=t("lm_label_ids_a"), masked_lm_labels_b=t("lm_label_ids_b"),
                    max_tag_length=dims["G"], img_index=t("image_index"), phrase_index=t("phrase_index"), **kw)
    got = np.array([x.item() for x in res])
    ref = d["losses"]
    rel = np.abs(got - ref) / np.abs(ref)
    print(name, "losses", got, "ref", ref, "rel", rel)
    assert len(res) == 6
    assert max(rel[0], rel[1], rel[3]) < LOSS_RTOL, rel          # total, masked-concept, MLM
    assert max(rel[2], rel[4]) < SMALL_ROWS_RTOL, rel            # contrastive, ITM (B=4)
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
                assert e < (1e-1 if pname in CLIP_BRANCH else 5e-2), (pname, gn, rn)
        else:
            # analytically zero gradient (key bias: softmax is invariant to it); the reference holds
            # f32 rounding noise there, the bf16 path bf16 rounding noise: absolute bound only
            print("   (zero in exact arithmetic) grad-norm", pname, gn, "ref", rn)
            assert gn < 2e-3, (pname, gn, rn)
            continue
        full = "grad:" + pname
        if full in d:
            e = _rel(p.grad, torch.from_numpy(d[full]))
            print("   grad", pname, "rel L2", e)
            if not (pname in CLIP_BRANCH and name.startswith("tiny")):
                assert e < (1e-1 if pname in CLIP_BRANCH else 3e-2) or pname == "logit_scale", (pname, e)
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
    assert e < 2e-2 and e2 < 5e-2  # 4x2 ITM logits of small magnitude
    out[0].backward()
    for pname, p in model.named_parameters():
        key = "gnorm:" + pname
        if key in d and float(d[key]) > 1e-6 and p.grad is not None:
            err = abs(p.grad.double().norm().item() - float(d[key])) / float(d[key])
            assert err < 5e-2, (pname, err)


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
    assert _rel(fine, torch.from_numpy(d["ret_fine_logits"])) < 2e-2
    sim_ref = torch.from_numpy(d["ret_global_txt"]) @ torch.from_numpy(d["ret_global_img"]).t()
    masked = sim_ref - 2 * torch.eye(sim_ref.shape[0])
    model.bert.hard_override = (masked.max(1)[1], masked.max(0)[1])
    model.forward_mod = "train"
    with Replay(dict(draw_randperm=[d["ret_randperm"]]), dev):
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
    assert _rel(o[1], torch.from_numpy(d["vqa_logits"])) < 2e-2
    o[0].backward()
    # VE
    ce_ = dict(cfg, loss_type="ce", num_labels=3, classifier="linear")
    model, _ = _build("BiImageBertForSequenceClassification", ce_, seed + 3, dev)
    o = model(labels=torch.from_numpy(d["ve_labels"]).to(dev), **kw)
    assert abs(o[0].item() - float(d["ve_loss"])) / float(d["ve_loss"]) < LOSS_RTOL
    assert _rel(o[1], torch.from_numpy(d["ve_logits"])) < 2e-2


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
    worst = max((_rel(grads[True][n], grads[False][n]), n) for n in grads[True] if grads[False][n].norm() > 1e-6)
    print("worst gradient difference", worst)
    assert worst[0] < 2e-3
    va = b["input_mask_a"].bool()
    vb = b["input_mask_b"].bool()
    (seq_u, txt_u, vis_u, sim_u), (seq_p, txt_p, vis_p, sim_p) = outs[True], outs[False]
    assert torch.equal(txt_u[va], txt_p[va]) and torch.equal(vis_u[vb], vis_p[vb]) and torch.equal(sim_u, sim_p)
    assert float(txt_u[~va].abs().max()) == 0.0 and float(vis_u[~vb].abs().max()) == 0.0
    vj = torch.cat([va, vb[:, dims["G"]:]], 1)
    assert torch.equal(seq_u[vj], seq_p[vj])


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
    assert torch.equal(model.coarse_scores(text, image), gt @ gi.t())
    # ranking of the images per caption from the cached scores == ranking from the pair-wise scores
    p_match = torch.softmax(got.float(), -1)[:, 1].view(n, n)
    assert torch.equal(p_match.argsort(1, descending=True), torch.softmax(ref.float(), -1)[:, 1].view(n, n).argsort(1, descending=True))


def test_bi_pretrain_parity_b64_vs_oracle(dev):
    """BERT-base, BASELINE configs[0] lengths but 64 pairs: every loss within 1e-3 of the oracle
    (the oracle itself is pinned to the reference by tests/test_oracle_golden.py)."""
    from mvp_pytorch_amd.synthetic import synthetic_batch
    cfg = dict(gu.BASE_CFG, vocab_size=31000)  # phrase ids just above only_word_size; keeps the test light
    dims = dict(B=64, T=35, P=5, G=20, R=10)
    model, sd = _build("BiBertImgForPreTraining", cfg, 99, dev)
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
    model.bert.hard_override = (masked.max(1)[1], masked.max(0)[1])
    bd = {k: v.to(dev) for k, v in b.items()}
    with torch.no_grad(), Replay(dict(draw_randperm=[perm.numpy()]), dev):
        res = model(input_ids_a=bd["input_ids_a"], token_type_ids_a=bd["segment_ids_a"], attention_mask_a=bd["input_mask_a"],
                    masked_lm_labels_a=bd["lm_label_ids_a"], input_ids_b=bd["input_ids_b"], img_feats=bd["img_feats"],
                    token_type_ids_b=bd["segment_ids_b"], attention_mask_b=bd["input_mask_b"],
                    masked_lm_labels_b=bd["lm_label_ids_b"], max_tag_length=dims["G"])
    got = np.array([x.item() for x in res])
    ref = np.array([x.item() for x in res_o])
    rel = np.abs(got - ref) / np.abs(ref)
    print("B=64 losses", got, "oracle", ref, "rel", rel)
    assert rel.max() < LOSS_RTOL, rel
    # free-running argmax: fraction of hard-negative indices equal to the f32 oracle's
    model.bert.hard_override = None
    with torch.no_grad(), Replay(dict(draw_randperm=[perm.numpy()]), dev):
        _, single, hard = model.bert(input_ids_a=bd["input_ids_a"], token_type_ids_a=bd["segment_ids_a"],
                                     attention_mask_a=bd["input_mask_a"], input_ids_b=bd["input_ids_b"],
                                     token_type_ids_b=bd["segment_ids_b"], attention_mask_b=bd["input_mask_b"],
                                     img_feats=bd["img_feats"], max_tag_length=dims["G"], encode_hn=True)
    agree = ((hard[0].cpu() == aux["hard_txt_index"]).float().mean().item(), (hard[1].cpu() == aux["hard_img_index"]).float().mean().item())
    print("B=64 hard-negative index agreement with the f32 oracle:", agree,
          "sim_mat max abs err", (single[2].cpu() - aux["sim_mat"]).abs().max().item())


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
