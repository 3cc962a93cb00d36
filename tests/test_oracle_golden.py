"""The oracle (oracle/mvptr_oracle.py) against the golden vectors produced by the imported
reference (tools/gen_golden.py).  CPU only.  fp32 vs fp32: tolerances are round-off level."""
import numpy as np
import pytest
import torch

import golden_util as gu
from oracle import mvptr_oracle as orc


def _sd(cfg_shapes_from, seed, gain=1.0):
    return {k: torch.from_numpy(v) for k, v in gu.det_state_dict(cfg_shapes_from, seed, gain).items()}


def _t(d, k):
    return torch.from_numpy(d["in:" + k])


def _draws(d, pre=""):
    sizes = d[pre + "draw_randint3_sizes"].tolist()
    flat = d[pre + "draw_randint3"]
    chunks, o = [], 0
    for s in sizes:
        chunks.append(flat[o:o + s])
        o += s
    return orc.Draws(randperm=list(d[pre + "draw_randperm"]), randint3=chunks, choice=d[pre + "draw_choice"].tolist())


def _bi_kwargs(d):
    return dict(input_ids_a=_t(d, "input_ids_a"), token_type_ids_a=_t(d, "segment_ids_a"),
                attention_mask_a=_t(d, "input_mask_a"), input_ids_b=_t(d, "input_ids_b"),
                token_type_ids_b=_t(d, "segment_ids_b"), attention_mask_b=_t(d, "input_mask_b"),
                img_feats=_t(d, "img_feats"))


@pytest.mark.parametrize("name", ["tiny_bi_pretrain", "cfg1_bi_pretrain", "tiny_bi_hn", "tiny_bi_pretrain_nophrase", "cfg1_bi_pretrain_nophrase"])
def test_bi_pretrain(name):
    from mvp_pytorch_amd.modeling import param_shapes
    d = gu.load(name)
    cfg = d["config"]
    sd = _sd(param_shapes("BiBertImgForPreTraining", cfg), int(d["seed"]), float(d["weight_gain"]))
    for v in sd.values():
        v.requires_grad_(True)
    nophrase = name.endswith("_nophrase")
    res, aux = orc.bi_bert_img_for_pretraining(
        sd, cfg, masked_lm_labels_a=_t(d, "lm_label_ids_a"), masked_lm_labels_b=_t(d, "lm_label_ids_b"),
        max_tag_length=d["dims"]["G"], img_index=None if nophrase else _t(d, "image_index"),
        phrase_index=None if nophrase else _t(d, "phrase_index"),
        draws=orc.Draws(randperm=list(d["draw_randperm"])) if nophrase else _draws(d), return_aux=True, **_bi_kwargs(d))
    assert len(res) == (5 if nophrase else 6)      # vl:1309: the 5-tuple without word-region alignment
    got = np.array([x.item() for x in res])
    np.testing.assert_allclose(got, d["losses"], rtol=2e-5, atol=1e-6)
    assert np.array_equal(aux["hard_txt_index"].numpy(), d["hard_txt_index"])
    assert np.array_equal(aux["hard_img_index"].numpy(), d["hard_img_index"])
    np.testing.assert_allclose(aux["sim_mat"].detach().numpy(), d["sim_mat"], atol=2e-6)
    if "sequence_output" in d:
        np.testing.assert_allclose(aux["sequence_output"].detach().numpy(), d["sequence_output"], atol=2e-4)
    if name == "tiny_bi_hn":   # the hard-negative fixture: margins far above any bf16 noise
        assert float(d["argmax_margin"]) > 1.2e-2
    res[0].backward()
    for k in d:
        if k.startswith("grad:"):
            g = sd[k[5:]].grad
            ref = d[k]
            if g is None:
                assert np.abs(ref).max() == 0, k
                continue
            np.testing.assert_allclose(g.numpy(), ref, atol=1e-5 + 1e-4 * np.abs(ref).max(), err_msg=k)
        if k.startswith("gnorm:"):
            g = sd[k[6:]].grad
            ref = float(d[k])
            got_n = 0.0 if g is None else g.double().norm().item()
            assert abs(got_n - ref) <= 1e-4 * max(ref, 1e-6) + 1e-7, (k, got_n, ref)


def test_adamw_step_tiny():
    from mvp_pytorch_amd.modeling import param_shapes
    d = gu.load("tiny_bi_pretrain")
    cfg = d["config"]
    sd = _sd(param_shapes("BiBertImgForPreTraining", cfg), int(d["seed"]))
    grads = {k[5:]: torch.from_numpy(v) for k, v in d.items() if k.startswith("grad:")}
    no_decay = ["bias", "LayerNorm.weight"]
    orc.adamw_step(sd, grads, {}, lr=5e-3, eps=1e-8,
                   weight_decay=lambda n: 0.0 if any(nd in n for nd in no_decay) else 0.01)
    for n in gu.ADAMW_PROBES:
        np.testing.assert_allclose(sd[n].numpy(), d["adamw:" + n], rtol=1e-5, atol=1e-7, err_msg=n)


def test_warmup_linear_kat():
    """transformers/pytorch_transformers/tests/optimization_test.py:105-110 known-answer schedule."""
    lrs = [10.0 * orc.warmup_linear(s, 2, 10) for s in range(1, 11)]
    np.testing.assert_allclose(lrs, [5.0, 10.0, 8.75, 7.5, 6.25, 5.0, 3.75, 2.5, 1.25, 0.0], atol=1e-9)


@pytest.mark.parametrize("name", ["tiny_single_pretrain", "cfg1_single_pretrain"])
def test_single_pretrain(name):
    from mvp_pytorch_amd.modeling import param_shapes
    d = gu.load(name)
    cfg = d["config"]
    sd = _sd(param_shapes("BertImgForPreTraining", cfg), int(d["seed"]))
    # reference ties decoder.weight to word_embeddings (torchscript=False, vl:1095-1100): one
    # tensor, whose loaded value is the one stored under the decoder key (loaded last)
    sd["bert.embeddings.word_embeddings.weight"] = sd["cls.predictions.decoder.weight"]
    for v in sd.values():
        v.requires_grad_(True)
    total, scores, rel, mlm = orc.bert_img_for_pretraining(
        sd, cfg, _t(d, "input_ids"), _t(d, "segment_ids"), _t(d, "input_mask"), _t(d, "lm_label_ids"),
        _t(d, "is_next"), _t(d, "img_feats"))
    np.testing.assert_allclose([total.item(), mlm.item()], d["losses"], rtol=2e-5)
    np.testing.assert_allclose(rel.detach().numpy(), d["seq_relationship_score"], atol=2e-5)
    np.testing.assert_allclose(scores.detach().numpy()[..., :64], d["prediction_scores_head"], atol=1e-4)
    assert abs(scores.detach().double().sum().item() - float(d["prediction_scores_sum"])) < 1e-3 * max(1.0, abs(float(d["prediction_scores_sum"])))
    total.backward()
    for k in d:
        if k.startswith("gnorm:"):
            n = k[6:]
            g = sd[n].grad
            ref = float(d[k])
            got_n = 0.0 if g is None else g.double().norm().item()
            assert abs(got_n - ref) <= 2e-4 * max(ref, 1e-6) + 1e-7, (k, got_n, ref)


def test_finetune_heads():
    from mvp_pytorch_amd.modeling import param_shapes
    d = gu.load("tiny_finetune")
    cfg = d["config"]
    seed = int(d["seed"])
    kw = _bi_kwargs(d)
    G = d["dims"]["G"]
    # retrieval
    cr = dict(cfg, loss_type="ce", num_labels=2)
    sd = _sd(param_shapes("BiImageBertForRetrieval", cr), seed + 1)
    o = orc.bi_retrieval(sd, cr, "train", max_tag_length=G, draws=orc.Draws(randperm=[d["ret_randperm"]]), **kw)
    np.testing.assert_allclose([o[0].item(), o[2].item(), o[3].item()], d["ret_train_losses"], rtol=2e-5)
    np.testing.assert_allclose(o[1].detach().numpy(), d["ret_train_logits"], atol=2e-5)
    assert np.array_equal(o[4].numpy(), d["ret_train_labels"])
    gt, gi = orc.bi_retrieval(sd, cr, "coarse", max_tag_length=G, **kw)
    np.testing.assert_allclose(gt.numpy(), d["ret_global_txt"], atol=2e-6)
    np.testing.assert_allclose(gi.numpy(), d["ret_global_img"], atol=2e-6)
    fine = orc.bi_retrieval(sd, cr, "fine", max_tag_length=G, **kw)
    np.testing.assert_allclose(fine.numpy(), d["ret_fine_logits"], atol=2e-5)
    # VQA (run_vqa.py never forwards max_tag_length -> model default 20)
    cv = dict(cfg, loss_type="bce", num_labels=37)
    sd = _sd(param_shapes("BiImageBertForVQA", cv), seed + 2)
    loss, logits = orc.bi_vqa(sd, cv, labels=torch.from_numpy(d["vqa_labels"]), **kw)
    np.testing.assert_allclose(loss.item(), float(d["vqa_loss"]), rtol=2e-5)
    np.testing.assert_allclose(logits.numpy(), d["vqa_logits"], atol=2e-5)
    # VE
    ce_ = dict(cfg, loss_type="ce", num_labels=3, classifier="linear")
    sd = _sd(param_shapes("BiImageBertForSequenceClassification", ce_), seed + 3)
    loss, logits = orc.bi_seq_cls(sd, ce_, labels=torch.from_numpy(d["ve_labels"]), **kw)
    np.testing.assert_allclose(loss.item(), float(d["ve_loss"]), rtol=2e-5)
    np.testing.assert_allclose(logits.numpy(), d["ve_logits"], atol=2e-5)


def test_branches():
    """qa_ans + phrase_mod='hard', hn_mod='sample', use_b, mlp classifier, soft / mse / bce / kl losses
    (tests/golden/tiny_branches.npz, produced by the reference)."""
    from mvp_pytorch_amd.modeling import param_shapes
    d = gu.load("tiny_branches")
    cfg, seed, G = d["config"], int(d["seed"]), d["dims"]["G"]
    kw = _bi_kwargs(d)
    sd = _sd(param_shapes("BiBertImgForPreTraining", cfg), seed)
    for v in sd.values():
        v.requires_grad_(True)
    res, aux = orc.bi_bert_img_for_pretraining(
        sd, cfg, masked_lm_labels_a=_t(d, "lm_label_ids_a"), masked_lm_labels_b=_t(d, "lm_label_ids_b"), max_tag_length=G,
        img_index=_t(d, "image_index"), phrase_index=_t(d, "phrase_index"), draws=_draws(d, "qa_"), return_aux=True,
        qa_ans=torch.from_numpy(d["qa_ans"]), phrase_mod="hard", **kw)
    assert len(res) == 7
    np.testing.assert_allclose([x.item() for x in res], d["qa_losses"], rtol=2e-5, atol=1e-6)
    assert np.array_equal(aux["hard_txt_index"].numpy(), d["qa_hard_txt_index"])
    assert np.array_equal(aux["hard_img_index"].numpy(), d["qa_hard_img_index"])
    res[0].backward()
    for k in d:
        if k.startswith("qa_gnorm:"):
            g = sd[k[9:]].grad
            ref = float(d[k])
            got_n = 0.0 if g is None else g.double().norm().item()
            assert abs(got_n - ref) <= 1e-4 * max(ref, 1e-6) + 1e-7, (k, got_n, ref)
    assert float(d["qa_gnorm:qa_head.weight"]) > 0
    with torch.no_grad():
        dr = orc.Draws(randperm=[d["hs_randperm"]], multinomial=list(d["hs_multinomial"]))
        o, _, hard = orc.bi_bert_img_model(sd, cfg, max_tag_length=G, encode_hn=True, draws=dr, hn_mod="sample",
                                           logit=sd["logit_scale"].exp(), **kw)
    assert np.array_equal(hard[0].numpy(), d["hs_hard_txt_index"]) and np.array_equal(hard[1].numpy(), d["hs_hard_img_index"])
    np.testing.assert_allclose(o[3].numpy(), d["hs_hard_pooled_output"], atol=2e-5)
    cases = [("mlp", dict(loss_type="ce", num_labels=3, classifier="mlp", cls_hidden_scale=3), 1, dict(use_b=True)),
             ("soft", dict(loss_type="ce", num_labels=2, classifier="linear"), 2, dict(soft_label=True)),
             ("mse", dict(loss_type="ce", num_labels=1, classifier="linear"), 3, {}),
             ("bce", dict(loss_type="bce", num_labels=37, classifier="linear"), 4, {})]
    for tag, extra, off, fkw in cases:
        c = dict(cfg, **extra)
        sdc = _sd(param_shapes("BiImageBertForSequenceClassification", c), seed + off)
        with torch.no_grad():
            loss, logits = orc.bi_seq_cls(sdc, c, labels=torch.from_numpy(d[tag + "_labels"]), **fkw, **kw)
        np.testing.assert_allclose(loss.item(), float(d[tag + "_loss"]), rtol=2e-5, err_msg=tag)
        np.testing.assert_allclose(logits.numpy(), d[tag + "_logits"], atol=2e-5, err_msg=tag)
    ck = dict(cfg, loss_type="kl", num_labels=3129)
    sdk = _sd(param_shapes("BiImageBertForVQA", ck), seed + 5)
    with torch.no_grad():
        loss, logits = orc.bi_vqa(sdk, ck, labels=torch.from_numpy(d["kl_labels"]), **kw)
    np.testing.assert_allclose(loss.item(), float(d["kl_loss"]), rtol=2e-5)
    np.testing.assert_allclose(logits.numpy()[:, :128], d["kl_logits_head"], atol=2e-5)


def _feature_rows(tag):
    z = np.load(gu.GOLDEN_DIR + "/tiny_features.npz")
    D, n = int(z[tag + ":D"]), int(z[tag + ":n"])
    return D, [(z["%s:%d:text" % (tag, i)].tobytes(), int(z["%s:%d:num_boxes" % (tag, i)]), z["%s:%d:feat" % (tag, i)])
               for i in range(n)]


@pytest.mark.parametrize("tag,R", [("small", 5), ("small", 50), ("wide", 4), ("wide", 50)])
def test_decode_img_feature_matches_reference_rows(tag, R):
    """oracle.decode_img_feature == the reference's get_img_feature (oscar_tsv4.py:696-724) on the
    fixture's TSV rows, bit for bit, with the truncation / zero padding of __getitem__ :332-352."""
    D, rows = _feature_rows(tag)
    for text, nb, feat in rows:
        got = orc.decode_img_feature(text, nb, D, R).numpy()
        assert got.shape == (R, D)
        keep = min(nb, R)
        assert np.array_equal(got[:keep].view(np.uint32), feat[:keep].view(np.uint32))
        assert not got[keep:].any()


def test_compute_ranks_coarse_oracle_and_device_routine_match_reference():
    """oracle.compute_ranks_coarse and the product's tensor routine (retrieval_eval.coarse_ranks, plain
    torch: runs on the CPU here, on the GPU in tests/test_pipeline_gpu.py) == the reference's
    compute_ranks_coarse (oscar/run_retrieval.py:481-522) on the fixture matrix: integer outputs,
    bit for bit (random f32 similarities: no ties)."""
    from mvp_pytorch_amd import retrieval_eval
    z = np.load(gu.GOLDEN_DIR + "/tiny_ranks.npz")
    sim, c, k_c, k_i = z["sim"], int(z["c"]), int(z["k_c"]), int(z["k_i"])
    i2t, t2i, i2t_idx, t2i_idx = orc.compute_ranks_coarse(sim, c, k_c, k_i)
    assert np.array_equal(np.array(i2t), z["i2t_ranks"]) and np.array_equal(np.array(t2i), z["t2i_ranks"])
    assert np.array_equal(np.array(i2t_idx), z["i2t_top"]) and np.array_equal(np.array(t2i_idx), z["t2i_top"])
    out = retrieval_eval.coarse_ranks(torch.from_numpy(sim), c, k_c, k_i)
    assert out["i2t_ranks"].dtype == torch.int64
    assert np.array_equal(out["i2t_ranks"].numpy(), z["i2t_ranks"]) and np.array_equal(out["t2i_ranks"].numpy(), z["t2i_ranks"])
    assert np.array_equal(out["i2t_topk"].numpy(), z["i2t_top"]) and np.array_equal(out["t2i_topk"].numpy(), z["t2i_top"])
    # second stage: rank of the first ground-truth candidate after sorting by score
    scores = torch.tensor([[0.1, 0.9, 0.5], [0.3, 0.2, 0.1], [0.3, 0.2, 0.1]])
    gt = torch.tensor([[False, False, True], [True, False, False], [False, False, False]])
    assert retrieval_eval.rerank_ranks(scores, None, gt).tolist() == [1, 0, 3]
