#!/usr/bin/env python
"""Generate tests/golden/*.npz by running the REFERENCE (imported from /root/reference, build
container only) on seeded synthetic inputs.  The reference source never travels: fixtures hold
inputs, the reference's outputs / gradients and the random draws it made inside forward.
Weights are not stored: both sides regenerate them with tests/golden_util.det_state_dict().

Run:  PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden.py
"""
import os
import random
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
REF = "/root/reference"


def install_shims():
    """the reference's vendored `transformers` is a namespace dir shadowed by the
    installed HF package; boto3/botocore/anytree are import-time-only dependencies."""
    t = types.ModuleType("transformers")
    t.__path__ = [os.path.join(REF, "transformers")]
    sys.modules["transformers"] = t
    for name in ("boto3", "botocore", "botocore.exceptions", "anytree"):
        sys.modules[name] = types.ModuleType(name)
    sys.modules["botocore.exceptions"].ClientError = Exception
    sys.modules["anytree"].AnyNode = object
    sys.path.insert(0, REF)


install_shims()
from oscar.modeling import modeling_vlbert as ref_vl  # noqa: E402
from transformers.pytorch_transformers.modeling_bert import BertConfig  # noqa: E402
from transformers.pytorch_transformers.optimization import AdamW  # noqa: E402

import golden_util as gu  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


class Recorder:
    """Records the reference's in-forward random draws (vl:556, vl:1548, vl:1573)."""

    def __init__(self):
        self.randperm, self.randint3, self.choice = [], [], []
        self._orig = (torch.randperm, torch.randint, random.choice)

    def __enter__(self):
        o_perm, o_int, o_choice = self._orig

        def perm(n, *a, **k):
            v = o_perm(n, *a, **k)
            self.randperm.append(v.cpu().numpy().copy())
            return v

        def rint(lo, hi, size, *a, **k):
            v = o_int(lo, hi, size, *a, **k)
            self.randint3.append(v.cpu().numpy().copy())
            return v

        def choice(seq):
            v = o_choice(seq)
            self.choice.append(int(v))
            return v

        torch.randperm, torch.randint, random.choice = perm, rint, choice
        return self

    def __exit__(self, *exc):
        torch.randperm, torch.randint, random.choice = self._orig

    def pack(self):
        return dict(draw_randperm=np.array(self.randperm, dtype=np.int64).reshape(len(self.randperm), -1),
                    draw_randint3=np.concatenate([x.reshape(-1) for x in self.randint3]) if self.randint3 else np.zeros(0, np.int64),
                    draw_randint3_sizes=np.array([x.size for x in self.randint3], dtype=np.int64),
                    draw_choice=np.array(self.choice, dtype=np.int64))


def make_config(c):
    cfg = BertConfig(vocab_size_or_config_json_file=c["vocab_size"], hidden_size=c["hidden_size"],
                     num_hidden_layers=c["num_hidden_layers"], num_attention_heads=c["num_attention_heads"],
                     intermediate_size=c["intermediate_size"], hidden_dropout_prob=0.0,
                     attention_probs_dropout_prob=0.0, layer_norm_eps=c["layer_norm_eps"])
    for k, v in c.items():
        setattr(cfg, k, v)
    cfg.torchscript = True  # clone (not slice-tie) the decoders
    return cfg


def load_det(model, seed, gain=1.0):
    sd = model.state_dict()
    det = gu.det_state_dict({k: tuple(v.shape) for k, v in sd.items()}, seed, gain)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in det.items()})
    return model


def grads_of(model, names=None):
    out = {}
    for n, p in model.named_parameters():
        if p.grad is None:
            continue
        if names is not None and n not in names:
            continue
        out["grad:" + n] = p.grad.detach().numpy().copy()
    return out


def grad_summary(model):
    """norm + first 32 elements of every gradient (for large models)."""
    out = {}
    for n, p in model.named_parameters():
        if p.grad is None:
            continue
        g = p.grad.detach().double().reshape(-1)
        out["gnorm:" + n] = np.array(g.norm().item())
        out["ghead:" + n] = g[:32].float().numpy().copy()
    return out


def np_batch(batch):
    return {"in:" + k: (v.numpy() if torch.is_tensor(v) else np.array(v)) for k, v in batch.items() if k != "host_counts" and not getattr(v, "host_only", False)}


def argmax_margin(sim):
    m = sim - 2 * torch.eye(sim.shape[0])
    a = m.topk(2, dim=1)[0]
    b = m.t().topk(2, dim=1)[0]
    return min((a[:, 0] - a[:, 1]).min().item(), (b[:, 0] - b[:, 1]).min().item())


def pick_seed(c, dims, seeds, wseed=None, gain=1.0):
    """bf16 kernels must reproduce the hard-negative argmax (vl:531-534): choose, among a few
    candidate input seeds, the batch whose top-2 similarity margin is largest."""
    cfg = make_config(c)
    best = None
    model = load_det(ref_vl.BiBertImgForPreTraining(cfg), seeds[0] if wseed is None else wseed, gain)
    model.eval()
    for seed in seeds:
        batch = gu.synthetic_batch(dims, c, seed)
        with torch.no_grad():
            gt, gi = model.bert.forward_single(input_ids_a=batch["input_ids_a"], token_type_ids_a=batch["segment_ids_a"],
                                               attention_mask_a=batch["input_mask_a"], img_feats=batch["img_feats"],
                                               input_ids_b=batch["input_ids_b"], token_type_ids_b=batch["segment_ids_b"],
                                               attention_mask_b=batch["input_mask_b"])
        mg = argmax_margin(gt @ gi.t())
        print("  seed", seed, "margin", mg)
        if best is None or mg > best[1]:
            best = (seed, mg)
    return best[0]


FULL_GRADS = ["bert.embeddings.LayerNorm.weight", "bert.img_embedding.bias", "bert.LayerNorm.weight",
              "bert.txt_proj", "logit_scale", "cls.seq_relationship.weight", "cls.predictions.bias",
              "half_mlm.transform.dense.weight", "bert.pooler.dense.weight",
              "bert.txt_encoder.layer.0.attention.self.query.weight", "bert.txt_encoder.layer.0.attention.self.key.bias",
              "bert.vis_encoder.layer.1.attention.self.value.weight", "bert.vis_encoder.layer.0.intermediate.dense.weight",
              "bert.mul_encoder.layer.1.output.dense.weight", "bert.mul_encoder.layer.1.output.LayerNorm.weight",
              "bert.mul_encoder.layer.0.attention.output.dense.bias", "bert.embeddings.token_type_embeddings.weight",
              "bert.embeddings.position_embeddings.weight"]


SMALL_GRADS = ["bert.embeddings.LayerNorm.weight", "bert.img_embedding.bias", "bert.LayerNorm.weight", "logit_scale",
               "cls.seq_relationship.weight", "cls.predictions.bias", "bert.txt_encoder.layer.0.attention.self.key.bias",
               "bert.mul_encoder.layer.1.output.LayerNorm.weight", "bert.mul_encoder.layer.0.attention.output.dense.bias",
               "bert.embeddings.token_type_embeddings.weight", "bert.vis_encoder.layer.1.intermediate.dense.bias",
               "half_mlm.transform.LayerNorm.bias"]


def gen_bi_pretrain(name, c, dims, wseed, seed, full_grads, gain=1.0, lean=False, phrase=True, grad_names=None):
    """phrase=False: the reference's 5-tuple (phrase_index=None, vl:1309) — no word-region alignment, so nothing in the
    step draws random numbers except the hard-negative permutation: the fixture the row-packed TRAINING path of the HIP
    model is compared with directly.  grad_names: full gradients to store when full_grads is False."""
    cfg = make_config(c)
    torch.manual_seed(seed)
    random.seed(seed)
    model = load_det(ref_vl.BiBertImgForPreTraining(cfg), wseed, gain)
    model.eval()
    batch = gu.synthetic_batch(dims, c, seed)
    with Recorder() as rec:
        outs = model(input_ids_a=batch["input_ids_a"], token_type_ids_a=batch["segment_ids_a"],
                     attention_mask_a=batch["input_mask_a"], masked_lm_labels_a=batch["lm_label_ids_a"],
                     input_ids_b=batch["input_ids_b"], img_feats=batch["img_feats"],
                     token_type_ids_b=batch["segment_ids_b"], attention_mask_b=batch["input_mask_b"],
                     masked_lm_labels_b=batch["lm_label_ids_b"], phrase_index=batch["phrase_index"] if phrase else None,
                     img_index=batch["image_index"] if phrase else None, max_tag_length=dims["G"])
    assert len(outs) == (6 if phrase else 5)
    outs[0].backward()
    # aux outputs: rerun the backbone with the recorded permutation to dump sim_mat / indices
    perm = rec.randperm[0]
    o_perm = torch.randperm
    torch.randperm = lambda n, *a, **k: torch.as_tensor(perm)
    with torch.no_grad():
        o, single, hard = model.bert(input_ids_a=batch["input_ids_a"], token_type_ids_a=batch["segment_ids_a"],
                                     attention_mask_a=batch["input_mask_a"], img_feats=batch["img_feats"],
                                     input_ids_b=batch["input_ids_b"], token_type_ids_b=batch["segment_ids_b"],
                                     attention_mask_b=batch["input_mask_b"], max_tag_length=dims["G"], encode_hn=True)
    torch.randperm = o_perm
    sim = single[2]
    margin = argmax_margin(sim)
    data = dict(np_batch(batch))
    data.update(rec.pack())
    data.update(losses=np.array([x.item() for x in outs], dtype=np.float64), sim_mat=sim.numpy(),
                hard_txt_index=hard[0].numpy(), hard_img_index=hard[1].numpy(),
                pooled_output=o[1].numpy(), hard_pooled_output=o[3].numpy(), argmax_margin=np.array(margin),
                seed=np.array(wseed), weight_gain=np.array(gain))
    if not lean:
        data.update(sequence_output=o[0].numpy(), txt_out=single[0].numpy(), vis_out=single[1].numpy())
    data.update(grad_summary(model))
    if full_grads:
        data.update(grads_of(model, FULL_GRADS))
    elif grad_names:
        data.update(grads_of(model, grad_names))
    if full_grads:
        # one reference AdamW step (run_pretrain_ml.py:379-393: lr 5e-5 style groups)
        no_decay = ["bias", "LayerNorm.weight"]
        groups = [{"params": [p for n, p in model.named_parameters() if not any(nd in n for nd in no_decay)], "weight_decay": 0.01},
                  {"params": [p for n, p in model.named_parameters() if any(nd in n for nd in no_decay)], "weight_decay": 0.0}]
        opt = AdamW(groups, lr=5e-3, eps=1e-8)
        opt.step()
        for n in gu.ADAMW_PROBES:
            data["adamw:" + n] = dict(model.named_parameters())[n].detach().numpy().copy()
    data["config_json"] = np.array(gu.to_json(c))
    data["dims_json"] = np.array(gu.to_json(dims))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **data)
    print(name, "losses", data["losses"], "argmax margin", margin)


def gen_single_pretrain(name, c, dims, seed):
    c = dict(c)
    c["max_text_seq_length"] = dims["T"]
    cfg = make_config(c)
    cfg.torchscript = False
    model = load_det(ref_vl.BertImgForPreTraining(cfg), seed)
    model.eval()
    batch = gu.synthetic_batch(dims, c, seed, single_stream=True)
    outs = model(batch["input_ids"], batch["segment_ids"], batch["input_mask"], batch["lm_label_ids"],
                 batch["is_next"], img_feats=batch["img_feats"])
    outs[0].backward()
    data = dict(np_batch(batch))
    ps = outs[1].detach()
    data.update(losses=np.array([outs[0].item(), outs[-1].item()]), prediction_scores_head=ps[..., :64].numpy().copy(),
                prediction_scores_sum=np.array(ps.double().sum().item()), seq_relationship_score=outs[2].detach().numpy(), seed=np.array(seed))
    data.update(grad_summary(model))
    data["config_json"] = np.array(gu.to_json(c))
    data["dims_json"] = np.array(gu.to_json(dims))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **data)
    print(name, "losses", data["losses"])


def gen_finetune(name, c, dims, seed):
    torch.manual_seed(seed)   # the hard-batch permutation (vl:556) is drawn from the global generator
    random.seed(seed)
    data = {}
    batch = gu.synthetic_batch(dims, c, seed)
    data.update(np_batch(batch))
    kw = dict(input_ids_a=batch["input_ids_a"], token_type_ids_a=batch["segment_ids_a"],
              attention_mask_a=batch["input_mask_a"], input_ids_b=batch["input_ids_b"],
              token_type_ids_b=batch["segment_ids_b"], attention_mask_b=batch["input_mask_b"],
              img_feats=batch["img_feats"], max_tag_length=dims["G"])
    # retrieval (vl:1598)
    cr = dict(c, loss_type="ce", num_labels=2)
    model = load_det(ref_vl.BiImageBertForRetrieval(make_config(cr)), seed + 1)
    model.eval()
    with Recorder() as rec:
        model.forward_mod = "train"
        o = model(**kw)
    o[0].backward()
    data.update({"ret_train_losses": np.array([o[0].item(), o[2].item(), o[3].item()]),
                 "ret_train_logits": o[1].detach().numpy(), "ret_train_labels": o[4].numpy(),
                 "ret_randperm": np.array(rec.randperm[0])})
    data.update({"ret_" + k: v for k, v in grad_summary(model).items()})
    with torch.no_grad():
        model.forward_mod = "coarse"
        gt, gi = model(**kw)
        model.forward_mod = "fine"
        fine = model(**kw)
    data.update(ret_global_txt=gt.numpy(), ret_global_img=gi.numpy(), ret_fine_logits=fine.numpy())
    # VQA (vl:1801)
    cv = dict(c, loss_type="bce", num_labels=37)
    model = load_det(ref_vl.BiImageBertForVQA(make_config(cv)), seed + 2)
    model.eval()
    g = torch.Generator().manual_seed(seed)
    labels = (torch.rand(dims["B"], 37, generator=g) < 0.1).float() * torch.rand(dims["B"], 37, generator=g)
    kwv = {k: v for k, v in kw.items() if k != "max_tag_length"}  # run_vqa.py:641-648 omits it
    o = model(labels=labels, **kwv)
    o[0].backward()
    data.update(vqa_labels=labels.numpy(), vqa_loss=np.array(o[0].item()), vqa_logits=o[1].detach().numpy())
    data.update({"vqa_" + k: v for k, v in grad_summary(model).items()})
    # VE / sequence classification (vl:1715)
    ce_ = dict(c, loss_type="ce", num_labels=3, classifier="linear")
    model = load_det(ref_vl.BiImageBertForSequenceClassification(make_config(ce_)), seed + 3)
    model.eval()
    lab = torch.randint(0, 3, (dims["B"],), generator=g)
    o = model(labels=lab, **kwv)
    o[0].backward()
    data.update(ve_labels=lab.numpy(), ve_loss=np.array(o[0].item()), ve_logits=o[1].detach().numpy())
    data.update({"ve_" + k: v for k, v in grad_summary(model).items()})
    data["config_json"] = np.array(gu.to_json(c))
    data["dims_json"] = np.array(gu.to_json(dims))
    data["seed"] = np.array(seed)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **data)
    print(name, "ret", data["ret_train_losses"], "vqa", data["vqa_loss"], "ve", data["ve_loss"])


class MultinomialRecorder:
    """hn_mod='sample' draws its hard negatives with torch.multinomial (vl:538,540)."""

    def __init__(self):
        self.draws = []

    def __enter__(self):
        self._orig = torch.multinomial

        def multi(*a, **k):
            v = self._orig(*a, **k)
            self.draws.append(v.cpu().numpy().copy())
            return v

        torch.multinomial = multi
        return self

    def __exit__(self, *exc):
        torch.multinomial = self._orig


def gen_branches(name, c, dims, seed):
    """Branches of the path no other fixture exercises: qa_ans + phrase_mod='hard' (vl:1264-1283),
    hn_mod='sample' (vl:535-540), use_b (vl:516), classifier='mlp' (vl:1737-1742), the kl / bce / mse
    loss types (vl:1777-1797, vl:1849-1869)."""
    torch.manual_seed(seed)
    random.seed(seed)
    data = {}
    batch = gu.synthetic_batch(dims, c, seed)
    data.update(np_batch(batch))
    g = torch.Generator().manual_seed(seed + 50)
    B = dims["B"]
    # (1) pre-training with a QA answer and the 'hard' WRA mode
    model = load_det(ref_vl.BiBertImgForPreTraining(make_config(c)), seed)
    model.eval()
    qa_ans = torch.randint(0, c["qa_answer_size"], (B,), generator=g)
    with Recorder() as rec:
        outs = model(input_ids_a=batch["input_ids_a"], token_type_ids_a=batch["segment_ids_a"],
                     attention_mask_a=batch["input_mask_a"], masked_lm_labels_a=batch["lm_label_ids_a"], qa_ans=qa_ans,
                     input_ids_b=batch["input_ids_b"], img_feats=batch["img_feats"],
                     token_type_ids_b=batch["segment_ids_b"], attention_mask_b=batch["input_mask_b"],
                     masked_lm_labels_b=batch["lm_label_ids_b"], phrase_index=batch["phrase_index"],
                     img_index=batch["image_index"], max_tag_length=dims["G"], phrase_mod="hard")
    assert len(outs) == 7
    outs[0].backward()
    perm = rec.randperm[0]
    o_perm = torch.randperm
    torch.randperm = lambda n, *a, **k: torch.as_tensor(perm)
    with torch.no_grad():
        _, single, hard = model.bert(input_ids_a=batch["input_ids_a"], token_type_ids_a=batch["segment_ids_a"],
                                     attention_mask_a=batch["input_mask_a"], img_feats=batch["img_feats"],
                                     input_ids_b=batch["input_ids_b"], token_type_ids_b=batch["segment_ids_b"],
                                     attention_mask_b=batch["input_mask_b"], max_tag_length=dims["G"], encode_hn=True)
    torch.randperm = o_perm
    data.update({"qa_" + k: v for k, v in rec.pack().items()})
    data.update(qa_ans=qa_ans.numpy(), qa_losses=np.array([x.item() for x in outs], dtype=np.float64),
                qa_sim_mat=single[2].numpy(), qa_hard_txt_index=hard[0].numpy(), qa_hard_img_index=hard[1].numpy())
    data.update({"qa_" + k: v for k, v in grad_summary(model).items() if k.startswith("gnorm:")})
    # (2) backbone with sampled hard negatives
    kw = dict(input_ids_a=batch["input_ids_a"], token_type_ids_a=batch["segment_ids_a"],
              attention_mask_a=batch["input_mask_a"], input_ids_b=batch["input_ids_b"],
              token_type_ids_b=batch["segment_ids_b"], attention_mask_b=batch["input_mask_b"],
              img_feats=batch["img_feats"])
    with torch.no_grad(), Recorder() as rec, MultinomialRecorder() as mrec:
        o, single, hard = model.bert(max_tag_length=dims["G"], encode_hn=True, hn_mod="sample", logit=model.logit_scale.exp(), **kw)
    data.update(hs_randperm=np.array(rec.randperm[0]), hs_multinomial=np.stack([d.reshape(-1) for d in mrec.draws]),
                hs_hard_txt_index=hard[0].numpy(), hs_hard_img_index=hard[1].numpy(),
                hs_pooled_output=o[1].numpy(), hs_hard_pooled_output=o[3].numpy())
    # (3) sequence classification: mlp classifier + use_b, and the loss types
    kwv = dict(kw)   # run_ve.py:515-523 / run_vqa.py:641-648 omit max_tag_length
    cm = dict(c, loss_type="ce", num_labels=3, classifier="mlp", cls_hidden_scale=3)
    m = load_det(ref_vl.BiImageBertForSequenceClassification(make_config(cm)), seed + 1)
    m.eval()
    lab = torch.randint(0, 3, (B,), generator=g)
    o = m(labels=lab, use_b=True, **kwv)
    o[0].backward()
    data.update(mlp_labels=lab.numpy(), mlp_loss=np.array(o[0].item()), mlp_logits=o[1].detach().numpy())
    data.update({"mlp_" + k: v for k, v in grad_summary(m).items() if k.startswith("gnorm:")})
    soft = torch.rand(B, generator=g)
    cs = dict(c, loss_type="ce", num_labels=2, classifier="linear")
    m = load_det(ref_vl.BiImageBertForSequenceClassification(make_config(cs)), seed + 2)
    m.eval()
    o = m(labels=soft, soft_label=True, **kwv)
    data.update(soft_labels=soft.numpy(), soft_loss=np.array(o[0].item()), soft_logits=o[1].detach().numpy())
    cr = dict(c, loss_type="ce", num_labels=1, classifier="linear")
    m = load_det(ref_vl.BiImageBertForSequenceClassification(make_config(cr)), seed + 3)
    m.eval()
    reg = torch.rand(B, generator=g)
    o = m(labels=reg, **kwv)
    data.update(mse_labels=reg.numpy(), mse_loss=np.array(o[0].item()), mse_logits=o[1].detach().numpy())
    cb = dict(c, loss_type="bce", num_labels=37, classifier="linear")
    m = load_det(ref_vl.BiImageBertForSequenceClassification(make_config(cb)), seed + 4)
    m.eval()
    lb = (torch.rand(B, 37, generator=g) < 0.1).float() * torch.rand(B, 37, generator=g)
    o = m(labels=lb, **kwv)
    data.update(bce_labels=lb.numpy(), bce_loss=np.array(o[0].item()), bce_logits=o[1].detach().numpy())
    # (4) VQA head with the KL loss over the 3129 answers (vl:1856-1861 hard-codes the width)
    ck = dict(c, loss_type="kl", num_labels=3129)
    m = load_det(ref_vl.BiImageBertForVQA(make_config(ck)), seed + 5)
    m.eval()
    lk = (torch.rand(B, 3129, generator=g) < 0.002).float() * torch.rand(B, 3129, generator=g)
    lk[:, 0] += 0.5
    lk = lk / lk.sum(1, keepdim=True)
    o = m(labels=lk, **kwv)
    o[0].backward()
    data.update(kl_labels=lk.numpy(), kl_loss=np.array(o[0].item()), kl_logits_head=o[1].detach().numpy()[:, :128].copy(),
                kl_logits_sum=np.array(o[1].detach().double().sum().item()))
    data.update({"kl_" + k: v for k, v in grad_summary(m).items() if k.startswith("gnorm:")})
    data["config_json"] = np.array(gu.to_json(c))
    data["dims_json"] = np.array(gu.to_json(dims))
    data["seed"] = np.array(seed)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **data)
    print(name, "qa losses", data["qa_losses"], "mlp", data["mlp_loss"], "soft", data["soft_loss"], "mse", data["mse_loss"],
          "bce", data["bce_loss"], "kl", data["kl_loss"])



def gen_features(name):
    """Input-pipeline fixture: TSV rows (image id, num_boxes, base64 of the float32
    features) are written to a scratch TSV file and read back through the REFERENCE's own reader
    and decoder — oscar/utils/tsv_file.py TSVFile.seek and OscarTSVDataset_C.get_img_feature
    (oscar/oscar_datasets_ml/oscar_tsv4.py:696-724) — on a stand-in `self` that only carries the
    attributes that method reads.  Stored: the base64 text of every row and the decoded arrays."""
    import base64
    import tempfile
    from oscar.oscar_datasets_ml.oscar_tsv4 import OscarTSVDataset_C
    from oscar.utils.tsv_file import TSVFile
    rng = np.random.RandomState(20260101)
    cases = [("small", 38, [0, 1, 3, 5, 7, 12]), ("wide", 2054, [3, 11])]
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        for tag, D, boxes in cases:
            path = os.path.join(tmp, tag + ".tsv")
            feats = []
            with open(path, "w") as f:
                for i, nb in enumerate(boxes):
                    a = rng.randn(nb, D).astype(np.float32)
                    if nb:
                        a[:, -6:] = rng.rand(nb, 6).astype(np.float32)
                        a[0, 0] = np.float32(1e-38)      # subnormal-range, sign and extreme bit patterns survive
                        a[-1, -1] = np.float32(-3.4e38)
                    feats.append(a)
                    f.write("%d\t%d\t%s\n" % (i, nb, base64.b64encode(a.tobytes()).decode()))
            tsv = TSVFile(path, generate_lineidx=True)
            stub = types.SimpleNamespace(
                check_img_feature_file=lambda: None, check_img_feature_offset_map=lambda: None,
                datasets_with_splits=[], img_feat_offset_map={"coco": {str(i): i for i in range(len(boxes))}},
                img_feature_file={"coco": tsv}, args=types.SimpleNamespace(img_feature_dim=D, dtype=torch.float32))
            for i, nb in enumerate(boxes):
                got = OscarTSVDataset_C.get_img_feature(stub, "coco_%d" % i)
                assert got.dtype == torch.float32 and tuple(got.shape) == (nb, D)
                assert np.array_equal(got.numpy().view(np.uint32), feats[i].view(np.uint32))
                row = tsv.seek(i)
                out["%s:%d:text" % (tag, i)] = np.frombuffer(row[-1].encode(), dtype=np.uint8).copy()
                out["%s:%d:num_boxes" % (tag, i)] = np.array(int(row[1]))
                out["%s:%d:feat" % (tag, i)] = got.numpy()
            out[tag + ":D"] = np.array(D)
            out[tag + ":n"] = np.array(len(boxes))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(name, "rows", sum(len(c[2]) for c in cases))



def gen_ranks(name):
    """Coarse-stage ranking fixture: the reference's compute_ranks_coarse (oscar/run_retrieval.py:481-522)
    on a random similarity matrix, with a stand-in dataset object that carries the attributes it reads."""
    from oscar import run_retrieval as ref_rr
    rng = np.random.RandomState(7)
    n_img, c, k_c, k_i = 23, 5, 16, 8
    sim = rng.randn(n_img, n_img * c).astype(np.float32)
    sim[np.arange(n_img), np.arange(n_img) * c + 2] += 1.5     # matched pairs tend to score high, not always first
    ds = types.SimpleNamespace(img_keys=list(range(100, 100 + n_img)),
                               args=types.SimpleNamespace(num_captions_per_img_train=c, num_captions_per_img_val=k_c,
                                                          num_images_per_cap_val=k_i))
    i2t, t2i, i2t_index, t2i_index = ref_rr.compute_ranks_coarse(ds, sim)
    i2t_top = np.array([[(key - 100) * c + cc for key, cc in i2t_index[ds.img_keys[i]]] for i in range(n_img)])
    t2i_top = np.array([t2i_index[(ds.img_keys[j // c], j % c)] for j in range(n_img * c)])
    np.savez_compressed(os.path.join(OUT, name + ".npz"), sim=sim, c=np.array(c), k_c=np.array(k_c), k_i=np.array(k_i),
                        i2t_ranks=np.array(i2t), t2i_ranks=np.array(t2i), i2t_top=i2t_top, t2i_top=t2i_top)
    print(name, "i2t R@1", np.mean(np.array(i2t) < 1), "t2i R@1", np.mean(np.array(t2i) < 1))


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    only = set(sys.argv[1:])   # optional: names of the fixtures to (re)generate
    want = lambda n: not only or n in only  # noqa: E731
    if want("tiny_bi_pretrain"):
        s1 = pick_seed(gu.TINY_CFG, gu.TINY_DIMS, list(range(1234, 1234 + 24)))
        gen_bi_pretrain("tiny_bi_pretrain", gu.TINY_CFG, gu.TINY_DIMS, 1234, s1, full_grads=True)
    if want("tiny_single_pretrain"):
        gen_single_pretrain("tiny_single_pretrain", gu.TINY_CFG, gu.TINY_DIMS, 1235)
    if want("tiny_finetune"):
        gen_finetune("tiny_finetune", gu.TINY_CFG, gu.TINY_FT_DIMS, 1236)
    if want("cfg1_bi_pretrain"):
        # input seed 9810 = the largest f32 top-2 margin of sim_mat (0.0118) among 950 candidate batches (seeds 4321-4330 and
        # 9000-9819, searched once with pick_seed): >= 10x the bf16 error of the kernels' sim_mat, so the BERT-base
        # hard-negative indices (vl:531-566) are asserted bit-exact, unconditionally
        s2 = pick_seed(gu.BASE_CFG, gu.CFG1_DIMS, [9810])
        gen_bi_pretrain("cfg1_bi_pretrain", gu.BASE_CFG, gu.CFG1_DIMS, 4321, s2, full_grads=False)
    if want("tiny_bi_pretrain_nophrase"):
        s1 = pick_seed(gu.TINY_CFG, gu.TINY_DIMS, list(range(1234, 1234 + 24)))
        gen_bi_pretrain("tiny_bi_pretrain_nophrase", gu.TINY_CFG, gu.TINY_DIMS, 1234, s1, full_grads=True, phrase=False, lean=True)
    if want("cfg1_bi_pretrain_nophrase"):
        gen_bi_pretrain("cfg1_bi_pretrain_nophrase", gu.BASE_CFG, gu.CFG1_DIMS, 4321, 9810, full_grads=False, phrase=False, lean=True,
                        grad_names=SMALL_GRADS)
    if want("cfg1_single_pretrain"):
        gen_single_pretrain("cfg1_single_pretrain", dict(gu.BASE_CFG, vocab_size=30522), gu.CFG1_DIMS, 4322)
    if want("tiny_bi_hn"):
        # hard-negative fixture: weights with a gain > 1 separate the [CLS] embeddings, so the f32 top-2
        # margin of every row / column of sim_mat is >= 10x the bf16 error of the kernels' sim_mat
        # (~1e-3): the argmax indices (vl:531-534) must then come out bit-exact, unconditionally
        s3 = pick_seed(gu.TINY_CFG, gu.HN_DIMS, list(range(7000, 7000 + 600)), wseed=777, gain=gu.HN_GAIN)
        gen_bi_pretrain("tiny_bi_hn", gu.TINY_CFG, gu.HN_DIMS, 777, s3, full_grads=False, gain=gu.HN_GAIN, lean=True)
    if want("tiny_features"):
        gen_features("tiny_features")
    if want("tiny_ranks"):
        gen_ranks("tiny_ranks")
    if want("tiny_branches"):
        gen_branches("tiny_branches", gu.TINY_CFG, gu.TINY_FT_DIMS, 1237)
