"""Drop-in class surface of `oscar.modeling.modeling_vlbert` / the vendored `modeling_bert`."""
import torch

from .modeling_bert import BertConfig, BertLayerNorm  # noqa: F401
from .modeling_utils import ImgPreTrainedModel, PreTrainedModel, PretrainedConfig  # noqa: F401
from . import modeling_vlbert
from .modeling_vlbert import (BertImgForPreTraining, BertImgModel, BiBertImgForPreTraining,  # noqa: F401
                              BiBertImgModel, BiImageBertForRetrieval,
                              BiImageBertForSequenceClassification, BiImageBertForVQA)


def make_config(cfg_dict, **overrides):
    """BertConfig from a plain dict of attributes (the reference mutates its config the same
    way: oscar/run_pretrain_ml.py:294-312)."""
    d = dict(cfg_dict)
    d.update(overrides)
    c = BertConfig(vocab_size_or_config_json_file=d["vocab_size"], hidden_size=d["hidden_size"],
                   num_hidden_layers=d["num_hidden_layers"], num_attention_heads=d["num_attention_heads"],
                   intermediate_size=d["intermediate_size"], hidden_act=d.get("hidden_act", "gelu"),
                   hidden_dropout_prob=d.get("hidden_dropout_prob", 0.1),
                   attention_probs_dropout_prob=d.get("attention_probs_dropout_prob", 0.1),
                   max_position_embeddings=d.get("max_position_embeddings", 512),
                   type_vocab_size=d.get("type_vocab_size", 2), initializer_range=d.get("initializer_range", 0.02),
                   layer_norm_eps=d.get("layer_norm_eps", 1e-12))
    for k, v in d.items():
        setattr(c, k, v)
    return c


def param_shapes(class_name, cfg_dict):
    """state_dict name -> shape for a model class, without allocating its weights."""
    cls = getattr(modeling_vlbert, class_name)
    with torch.device("meta"):
        m = cls(make_config(cfg_dict))
    return {k: tuple(v.shape) for k, v in m.state_dict().items()}
