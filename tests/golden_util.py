"""Shared by tools/gen_golden.py (reference side) and the tests (oracle / HIP side):
deterministic weights, configs and shapes of the committed golden fixtures."""
import json
import os
import zlib

import numpy as np

from mvp_pytorch_amd.synthetic import synthetic_batch  # noqa: F401  (re-exported)

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def free_port():
    """A TCP port the kernel has just handed out on 127.0.0.1 (bind to port 0) — rendezvous ports derived from the pid
    collide between test processes."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]

TINY_CFG = dict(vocab_size=1200, only_word_size=1000, hidden_size=128, num_hidden_layers=4,
                num_attention_heads=2, intermediate_size=512, layer_norm_eps=1e-12,
                img_feature_dim=2054, img_feature_type="faster_r-cnn", use_img_layernorm=1,
                img_layer_norm_eps=1e-12, num_contrast_classes=2, qa_answer_size=10,
                max_position_embeddings=64, type_vocab_size=2, hidden_act="gelu",
                initializer_range=0.02, loss_type="ce", num_labels=2)
TINY_DIMS = dict(B=4, T=12, P=3, G=6, R=5)
TINY_FT_DIMS = dict(B=4, T=12, P=3, G=20, R=5)   # fine-tune scripts rely on the default max_tag_length=20
HN_DIMS = dict(B=4, T=12, P=3, G=6, R=5)         # hard-negative fixture (tiny_bi_hn)
HN_GAIN = 5.0

BASE_CFG = dict(vocab_size=86051, only_word_size=30522, hidden_size=768, num_hidden_layers=12,
                num_attention_heads=12, intermediate_size=3072, layer_norm_eps=1e-12,
                img_feature_dim=2054, img_feature_type="faster_r-cnn", use_img_layernorm=1,
                img_layer_norm_eps=1e-12, num_contrast_classes=2, qa_answer_size=10,
                max_position_embeddings=512, type_vocab_size=2, hidden_act="gelu",
                initializer_range=0.02, loss_type="ce", num_labels=2)
CFG1_DIMS = dict(B=4, T=35, P=5, G=20, R=10)      # BASELINE.json configs[0]
CFG2_DIMS = dict(B=256, T=70, P=5, G=20, R=50)    # BASELINE.json configs[1]

ADAMW_PROBES = ["bert.txt_encoder.layer.0.attention.self.query.weight",
                "bert.mul_encoder.layer.1.output.LayerNorm.weight",
                "bert.img_embedding.bias", "cls.seq_relationship.weight", "logit_scale"]


def to_json(d):
    return json.dumps(d, sort_keys=True)


def _lowbias32(x):
    x = x.astype(np.uint64)
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x7FEB352D)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(15)
    x = (x * np.uint64(0x846CA68B)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(16)
    return x


def det_uniform(name, shape, seed):
    """Platform-independent uniform(-1, 1) tensor keyed by (name, seed): integer hash only."""
    n = int(np.prod(shape)) if len(shape) else 1
    key = np.uint64((zlib.crc32(name.encode()) ^ (seed * 0x9E3779B1)) & 0xFFFFFFFF)
    h = _lowbias32(_lowbias32(np.arange(n, dtype=np.uint64) + key) ^ np.uint64(0x5BD1E995))
    u = (h.astype(np.float64) + 0.5) / 4294967296.0
    return (2.0 * u - 1.0).reshape(shape)


def det_state_dict(shapes, seed, gain=1.0):
    """name -> float32 array.  Linear/embedding weights ~ U(-s, s) with std 0.03 * gain, LayerNorm
    weight 1 +- 0.1, biases +- 0.05, logit_scale ln(1/0.07), projections scaled like the reference
    init (* gain).  gain > 1 makes the network's output depend on its input strongly enough that the
    [CLS] embeddings of different samples are well separated (hard-negative fixtures)."""
    out = {}
    for name in sorted(shapes):
        shape = tuple(shapes[name])
        u = det_uniform(name, shape, seed)
        if name.endswith("logit_scale"):
            v = np.full(shape, np.log(1 / 0.07))
        elif name.endswith("LayerNorm.weight"):
            v = 1.0 + 0.1 * u
        elif name.endswith("bias") or name.endswith("LayerNorm.bias"):
            v = 0.05 * u
        elif name.endswith("txt_proj") or name.endswith("vis_proj"):
            v = u * (3.0 ** 0.5) * shape[0] ** -0.5 * gain
        else:
            v = u * (3.0 ** 0.5) * 0.03 * gain
        out[name] = v.astype(np.float32)
    return out


def load(name):
    path = os.path.join(GOLDEN_DIR, name + ".npz")
    z = np.load(path, allow_pickle=False)
    d = {k: z[k] for k in z.files}
    d["config"] = json.loads(str(d["config_json"]))
    d["dims"] = json.loads(str(d["dims_json"]))
    return d


class InjectHard:
    """Test-side injection of captured hard-negative indices: replaces the
    backbone's mine_hard_negatives() on the INSTANCE for the duration of a parity run, so that losses
    and gradients are compared on the same hard batch the reference built.  Nothing in the product
    reads such a hook."""

    def __init__(self, bert, hard_img, hard_txt):
        self.bert, self.hi, self.ht = bert, hard_img, hard_txt

    def __enter__(self):
        hi, ht = self.hi, self.ht
        self.bert.mine_hard_negatives = lambda sim, hn_mod="hard", logit=None: (hi.to(sim.device), ht.to(sim.device))
        return self

    def __exit__(self, *exc):
        del self.bert.mine_hard_negatives
