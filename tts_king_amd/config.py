"""YAML config loader.

The reference reads config.yaml through OmegaConf (reference: tts_king.py:20, train.py:240) and then
uses BOTH attribute access (`cfg.hifi.upsample_rates`, `model_config.use_cwt`) and item access
(`model_config["transformer"]["encoder_hidden"]`) on it.  OmegaConf is not installed on the MI355X
image, so this is a small PyYAML-backed mapping that supports both access styles and nothing else.
"""
import copy
import yaml


class Config(dict):
    """dict with attribute access, applied recursively to nested mappings."""

    def __init__(self, mapping=None, **kw):
        super().__init__()
        for k, v in dict(mapping or {}, **kw).items():
            self[k] = v

    @staticmethod
    def _wrap(v):
        if isinstance(v, dict) and not isinstance(v, Config):
            return Config(v)
        return v

    def __setitem__(self, k, v):
        super().__setitem__(k, Config._wrap(v))

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k) from None

    def __setattr__(self, k, v):
        self[k] = v

    def get(self, k, default=None):
        return self[k] if k in self else default

    def __deepcopy__(self, memo):
        return Config({k: copy.deepcopy(v, memo) for k, v in self.items()})


def load_config(path="./config.yaml"):
    with open(path, "r") as f:
        return Config(yaml.safe_load(f))


def default_config():
    """The config.yaml shipped at the repo root."""
    import os
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = load_config(os.path.join(here, "config.yaml"))
    cfg.preprocess_config.path.preprocessed_path = os.path.join(here, "pretrained")
    return cfg
