"""Input pipeline on the GPU, through the C ABI: the base64 feature decode kernel
against the rows the reference's own reader / decoder produced (tests/golden/tiny_features.npz,
oscar_tsv4.py:696-724), and the double-buffered staging against plain `.to(device)` copies of the
same batches (run_pretrain_ml.py:474-513).  Byte work: everything is compared bit for bit."""
import base64

import numpy as np
import pytest
import torch

import golden_util as gu
from oracle import mvptr_oracle as orc

pytestmark = pytest.mark.gpu


def _rows(tag):
    z = np.load(gu.GOLDEN_DIR + "/tiny_features.npz")
    D, n = int(z[tag + ":D"]), int(z[tag + ":n"])
    return D, [(z["%s:%d:text" % (tag, i)].tobytes(), int(z["%s:%d:num_boxes" % (tag, i)]), z["%s:%d:feat" % (tag, i)])
               for i in range(n)]


def _pack(rows, dev):
    pos, offs, lens, nbs, buf = 0, [], [], [], bytearray()
    for text, nb, _ in rows:
        offs.append(pos)
        lens.append(len(text))
        nbs.append(nb)
        pad = (-len(text)) % 16
        buf += text + b"\0" * pad
        pos += len(text) + pad
    buf += b"\0" * 16
    return (torch.frombuffer(bytearray(buf), dtype=torch.uint8).to(dev), torch.tensor(offs, dtype=torch.int64, device=dev),
            torch.tensor(lens, dtype=torch.int64, device=dev), torch.tensor(nbs, dtype=torch.int32, device=dev))


@pytest.mark.parametrize("tag,R", [("small", 5), ("small", 50), ("small", 1), ("wide", 4), ("wide", 50)])
def test_b64_decode_matches_reference_rows(dev, tag, R):
    from mvp_pytorch_amd import hip
    D, rows = _rows(tag)
    text, offs, lens, nbs = _pack(rows, dev)
    n = len(rows)
    ld = (D + 7) & ~7
    out = torch.full((n, R, D), 7.0, device=dev)
    outb = torch.full((n * R, ld), 7.0, device=dev, dtype=torch.bfloat16)
    err = hip.b64_decode_features(text, offs, lens, nbs, R, D, out_f32=out, out_bf16=outb)
    assert int(err.item()) == 0
    got = out.cpu().numpy()
    for i, (t, nb, feat) in enumerate(rows):
        want = orc.decode_img_feature(t, nb, D, R).numpy()   # == the reference rows (CPU suite) + padding
        assert np.array_equal(got[i].view(np.uint32), want.view(np.uint32)), (tag, i, nb)
    wantb = torch.zeros(n * R, ld, dtype=torch.bfloat16)
    wantb[:, :D] = torch.from_numpy(got).reshape(n * R, D).to(torch.bfloat16)
    assert torch.equal(outb.cpu().view(torch.int16), wantb.view(torch.int16))
    # bf16 only / f32 only
    outb2 = torch.empty_like(outb)
    hip.b64_decode_features(text, offs, lens, nbs, R, D, out_bf16=outb2)
    assert torch.equal(outb2.view(torch.int16), outb.view(torch.int16))


def test_b64_decode_flags_bad_input(dev):
    from mvp_pytorch_amd import hip
    D, R = 38, 6
    a = np.arange(4 * D, dtype=np.float32).reshape(4, D)
    good = base64.b64encode(a.tobytes())
    for text, nb, flag in ((good[:-8], 4, 1), (good, 5, 1), (good[:40] + b"!" + good[41:], 4, 2), (good, 4, 0),
                           (good + b"AAAA", 4, 0)):
        t, o, l, n = _pack([(text, nb, None)], dev)
        out = torch.empty(1, R, D, device=dev)
        err = hip.b64_decode_features(t, o, l, n, R, D, out_f32=out)
        assert int(err.item()) & 3 == flag, (len(text), nb, flag, int(err.item()))
        if flag == 0:
            assert np.array_equal(out[0, :4].cpu().numpy(), a) and not out[0, 4:].any()
    with pytest.raises(RuntimeError):
        t, o, l, n = _pack([(good, 4, None)], dev)
        hip.b64_decode_features(t[1:], o, l, n, R, D, out_f32=torch.empty(1, R, D, device=dev))   # misaligned text


def _samples(dims, cfg, seed, B, encode):
    from mvp_pytorch_amd.input_pipeline import INT_FIELDS, encode_features_b64
    from mvp_pytorch_amd.synthetic import synthetic_batch
    b = synthetic_batch(dict(dims, B=B), cfg, seed)
    b["is_next"] = torch.zeros(B, dtype=torch.long)
    b["is_img_match"] = torch.zeros(B, dtype=torch.long)
    out = []
    for i in range(B):
        nb = int(b["input_mask_b"][i, dims["G"]:].sum())
        img = (encode_features_b64(b["img_feats"][i, :nb].numpy()), nb) if encode else b["img_feats"][i]
        out.append((img,) + tuple(b[k][i] for k in INT_FIELDS))
    return out, b


@pytest.mark.parametrize("decoded", [False, True])
def test_stager_batches_equal_plain_copies(dev, decoded):
    """Three batches through the double-buffered stager (two slots: the third reuses the first) equal
    the tensors data_process would have produced with .to(device)."""
    from mvp_pytorch_amd.input_pipeline import INT_FIELDS, PretrainBatchStager
    dims, cfg, B = dict(T=12, P=3, G=6, R=5), dict(gu.TINY_CFG), 6
    st = PretrainBatchStager(dev, B, dims, cfg["img_feature_dim"], depth=2, features="f32" if decoded else "both")
    made = [_samples(dims, cfg, 100 + i, B, encode=not decoded) for i in range(3)]
    seen = 0
    for batch, (_, ref) in zip(st.batches((m[0] for m in made), decoded=decoded), made):
        for k in INT_FIELDS:
            assert torch.equal(batch[k].cpu(), ref[k].reshape(batch[k].shape)), k
        assert torch.equal(batch["img_feats"].cpu().view(torch.int32), ref["img_feats"].view(torch.int32))
        if not decoded:
            D = cfg["img_feature_dim"]
            assert torch.equal(batch["img_feats_bf16"][:, :D].cpu().view(torch.int16),
                               ref["img_feats"].reshape(-1, D).to(torch.bfloat16).view(torch.int16))
        _ = batch["img_feats"].sum()    # "the step": work on the compute stream that reads the slot
        seen += 1
    assert seen == 3
    torch.cuda.synchronize()


def test_stager_feeds_pretrain_step(dev):
    """A staged batch drives train.pretrain_step exactly like the resident synthetic batch."""
    from mvp_pytorch_amd import modeling, train
    from mvp_pytorch_amd.input_pipeline import PretrainBatchStager
    dims, B = dict(gu.TINY_DIMS), gu.TINY_DIMS["B"]
    cfg = dict(gu.TINY_CFG, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    samples, ref = _samples(dims, cfg, 5, B, encode=True)
    losses = []
    for staged in (False, True):
        torch.manual_seed(0)
        model = modeling.BiBertImgForPreTraining(modeling.make_config(cfg)).to(dev).train()
        model.wra_on_device = False
        opt, sch = train.build_optimizer(model)
        if staged:
            st = PretrainBatchStager(dev, B, dims, cfg["img_feature_dim"])
            st.put(samples)
            batch = st.get(check=True)
        else:
            batch = {k: v.to(dev) for k, v in ref.items()}
        import random
        random.seed(1)
        torch.manual_seed(1)
        out = train.pretrain_step(model, batch, opt, sch, max_tag_length=dims["G"], return_losses=True)
        losses.append([float(x) for x in out])
    assert losses[0] == losses[1], losses


def test_coarse_ranks_on_device(dev):
    """retrieval_eval.coarse_ranks on the GPU == the reference fixture (run_retrieval.py:481-522), and at
    the COCO-1k size (1 000 x 5 000, BASELINE configs[3]) == the oracle's argsort walk."""
    from mvp_pytorch_amd import retrieval_eval
    z = np.load(gu.GOLDEN_DIR + "/tiny_ranks.npz")
    out = retrieval_eval.coarse_ranks(torch.from_numpy(z["sim"]).to(dev), int(z["c"]), int(z["k_c"]), int(z["k_i"]))
    for k, g in (("i2t_ranks", "i2t_ranks"), ("t2i_ranks", "t2i_ranks"), ("i2t_topk", "i2t_top"), ("t2i_topk", "t2i_top")):
        assert np.array_equal(out[k].cpu().numpy(), z[g]), k
    rng = np.random.RandomState(3)
    sim = rng.randn(1000, 5000).astype(np.float32)
    i2t, t2i, i2t_idx, t2i_idx = orc.compute_ranks_coarse(sim, 5, 20, 20)
    out = retrieval_eval.coarse_ranks(torch.from_numpy(sim).to(dev), 5, 20, 20)
    assert np.array_equal(out["i2t_ranks"].cpu().numpy(), np.array(i2t))
    assert np.array_equal(out["t2i_ranks"].cpu().numpy(), np.array(t2i))
    assert np.array_equal(out["i2t_topk"].cpu().numpy(), np.array(i2t_idx))
    assert np.array_equal(out["t2i_topk"].cpu().numpy(), np.array(t2i_idx))
