"""Config / checkpoint plumbing with the reference's surface.

Mirrors (behaviour, not code) transformers/pytorch_transformers/modeling_utils.py:
PretrainedConfig :70-216 (JSON <-> attribute bag, from_pretrained / save_pretrained),
PreTrainedModel :219-520 (save_pretrained = config.json + pytorch_model.bin state_dict,
from_pretrained = build, load with 'bert.' prefix auto add/strip and legacy gamma/beta
renaming, tie_weights(), eval()), and oscar/modeling/modeling_utils.py:680-874
(ImgPreTrainedModel: tolerate a size mismatch on cls.seq_relationship).
Only local paths are supported (no S3/HTTP cache).
"""
import copy
import json
import logging
import os

import torch
from torch import nn

logger = logging.getLogger(__name__)

CONFIG_NAME = "config.json"
WEIGHTS_NAME = "pytorch_model.bin"


class PretrainedConfig(object):
    pretrained_config_archive_map = {}

    def __init__(self, **kwargs):
        self.finetuning_task = kwargs.pop("finetuning_task", None)
        self.num_labels = kwargs.pop("num_labels", 2)
        self.output_attentions = kwargs.pop("output_attentions", False)
        self.output_hidden_states = kwargs.pop("output_hidden_states", False)
        self.torchscript = kwargs.pop("torchscript", False)

    # ---- persistence
    def save_pretrained(self, save_directory):
        if not os.path.isdir(save_directory):
            raise AssertionError("Saving path should be a directory where the model and configuration can be saved")
        self.to_json_file(os.path.join(save_directory, CONFIG_NAME))

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, **kwargs):
        kwargs.pop("cache_dir", None)
        return_unused = kwargs.pop("return_unused_kwargs", False)
        path = pretrained_model_name_or_path
        if os.path.isdir(path):
            path = os.path.join(path, CONFIG_NAME)
        if not os.path.isfile(path):
            logger.error("config file '%s' not found (only local paths are supported)", path)
            return None
        config = cls.from_json_file(path)
        used = [k for k in kwargs if hasattr(config, k)]
        for k in used:
            setattr(config, k, kwargs.pop(k))
        return (config, kwargs) if return_unused else config

    @classmethod
    def from_dict(cls, json_object):
        config = cls(vocab_size_or_config_json_file=-1)
        config.__dict__.update(json_object)
        return config

    @classmethod
    def from_json_file(cls, json_file):
        with open(json_file, "r", encoding="utf-8") as f:
            return cls.from_dict(json.loads(f.read()))

    def __eq__(self, other):
        return self.__dict__ == other.__dict__

    def __repr__(self):
        return str(self.to_json_string())

    def to_dict(self):
        return copy.deepcopy(self.__dict__)

    def to_json_string(self):
        return json.dumps(self.to_dict(), indent=2, sort_keys=True) + "\n"

    def to_json_file(self, json_file_path):
        with open(json_file_path, "w", encoding="utf-8") as f:
            f.write(self.to_json_string())


class PreTrainedModel(nn.Module):
    config_class = PretrainedConfig
    base_model_prefix = ""
    tolerate_seq_relationship_mismatch = False

    def __init__(self, config, *inputs, **kwargs):
        super().__init__()
        if not isinstance(config, PretrainedConfig):
            raise ValueError("Parameter config in `{0}(config)` should be an instance of class `PretrainedConfig`. "
                             "To create a model from a pretrained model use "
                             "`model = {0}.from_pretrained(PRETRAINED_MODEL_NAME)`".format(self.__class__.__name__))
        self.config = config

    # ---- weights
    def init_weights(self, module):
        """normal(0, initializer_range) for Linear/Embedding, LayerNorm = (1, 0), zero biases
        (modeling_bert.py:579-590)."""
        from .modeling_bert import BertLayerNorm
        if isinstance(module, (nn.Linear, nn.Embedding)):
            module.weight.data.normal_(mean=0.0, std=self.config.initializer_range)
        elif isinstance(module, BertLayerNorm):
            module.bias.data.zero_()
            module.weight.data.fill_(1.0)
        if isinstance(module, nn.Linear) and module.bias is not None:
            module.bias.data.zero_()

    def _tie_or_clone_weights(self, first_module, second_module, only_vocab=False, only_word_size=None):
        """modeling_utils.py:275-287.  The reference's only_vocab tie assigns a tensor slice to
        nn.Linear.weight, which modern torch rejects; like the reference
        under torchscript=True the decoder becomes an independent clone of the first
        `only_word_size` embedding rows."""
        src = second_module.weight
        if only_vocab:
            first_module.weight = nn.Parameter(src[:only_word_size, :].detach().clone())
        elif self.config.torchscript:
            first_module.weight = nn.Parameter(src.detach().clone())
        else:
            first_module.weight = src

    def _get_resized_embeddings(self, old_embeddings, new_num_tokens=None):
        if new_num_tokens is None:
            return old_embeddings
        old_n, dim = old_embeddings.weight.size()
        if old_n == new_num_tokens:
            return old_embeddings
        new = nn.Embedding(new_num_tokens, dim).to(old_embeddings.weight.device)
        self.init_weights(new)
        n = min(old_n, new_num_tokens)
        new.weight.data[:n, :] = old_embeddings.weight.data[:n, :]
        return new

    def resize_token_embeddings(self, new_num_tokens=None):
        base = getattr(self, self.base_model_prefix, self)
        emb = base._resize_token_embeddings(new_num_tokens)
        if new_num_tokens is None:
            return emb
        self.config.vocab_size = new_num_tokens
        base.vocab_size = new_num_tokens
        if hasattr(self, "tie_weights"):
            self.tie_weights()
        return emb

    def prune_heads(self, heads_to_prune):
        raise NotImplementedError("head pruning is not supported by the HIP encoder (head_mask must stay None)")

    # ---- persistence
    def save_pretrained(self, save_directory):
        if not os.path.isdir(save_directory):
            raise AssertionError("Saving path should be a directory where the model and configuration can be saved")
        model = self.module if hasattr(self, "module") else self
        model.config.save_pretrained(save_directory)
        torch.save(model.state_dict(), os.path.join(save_directory, WEIGHTS_NAME))

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, *model_args, **kwargs):
        config = kwargs.pop("config", None)
        state_dict = kwargs.pop("state_dict", None)
        kwargs.pop("cache_dir", None)
        if kwargs.pop("from_tf", False):
            raise NotImplementedError("TensorFlow checkpoints are not supported")
        output_loading_info = kwargs.pop("output_loading_info", False)
        if config is None:
            config, model_kwargs = cls.config_class.from_pretrained(
                pretrained_model_name_or_path, *model_args, return_unused_kwargs=True, **kwargs)
        else:
            model_kwargs = kwargs
        archive = pretrained_model_name_or_path
        if os.path.isdir(archive):
            archive = os.path.join(archive, WEIGHTS_NAME)
        model = cls(config, *model_args, **model_kwargs)
        if state_dict is None:
            if not os.path.isfile(archive):
                logger.error("weights file '%s' not found (only local paths are supported)", archive)
                return None
            state_dict = torch.load(archive, map_location="cpu")
        state_dict = {k.replace("gamma", "weight").replace("beta", "bias"): v for k, v in state_dict.items()}
        missing, unexpected, errors = [], [], []
        prefix = cls.base_model_prefix
        has_prefixed = any(k.startswith(prefix) for k in state_dict) if prefix else False
        target, start = model, ""
        if prefix and not hasattr(model, prefix) and has_prefixed:
            start = prefix + "."
        if prefix and hasattr(model, prefix) and not has_prefixed:
            target = getattr(model, prefix)

        def visit(module, pfx):
            module._load_from_state_dict(state_dict, pfx, {}, True, missing, unexpected, errors)
            for name, child in module._modules.items():
                if child is not None:
                    visit(child, pfx + name + ".")

        visit(target, start)
        if missing:
            logger.info("Weights of %s not initialized from pretrained model: %s", cls.__name__, missing)
        if errors:
            tolerated = (cls.tolerate_seq_relationship_mismatch and len(errors) == 2
                         and "size mismatch for cls.seq_relationship.weight" in errors[0])
            if not tolerated:
                raise RuntimeError("Error(s) in loading state_dict for {}:\n\t{}".format(cls.__name__, "\n\t".join(errors)))
            logger.info("tolerated: %s", errors)
        if hasattr(model, "tie_weights"):
            model.tie_weights()
        model.eval()
        if output_loading_info:
            return model, {"missing_keys": missing, "unexpected_keys": unexpected, "error_msgs": errors}
        return model


class ImgPreTrainedModel(PreTrainedModel):
    """oscar/modeling/modeling_utils.py:680-874 — same loader, tolerant of a resized ITM head."""
    tolerate_seq_relationship_mismatch = True
