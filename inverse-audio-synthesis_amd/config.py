"""Hydra-style config composition without hydra (hydra/omegaconf are not in this image).

Mirrors what ``@hydra.main(config_path="conf", config_name="config")`` does for the reference
(/root/reference/pretrain.py:51): load ``conf/config.yaml``, resolve its ``defaults:`` list
(``- vicreg: full`` -> ``conf/vicreg/full.yaml`` mounted at key ``vicreg``), then apply command-line
overrides ``a.b.c=value`` (values parsed as YAML) and group overrides ``vicreg=fast``.
"""
import os
import re

import yaml


class _Loader(yaml.SafeLoader):
    """SafeLoader that also reads ``1e-6`` as a float (YAML 1.1 wants ``1.0e-6``; omegaconf accepts both)."""


_Loader.add_implicit_resolver(
    "tag:yaml.org,2002:float",
    re.compile(r"^[-+]?(\d+\.?\d*|\.\d+)[eE][-+]?\d+$"),
    list("-+0123456789."),
)


def _parse(text):
    return yaml.load(text, Loader=_Loader)


class Cfg(dict):
    """dict with attribute access (cfg.vicreg.batch_size), like omegaconf's DictConfig."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    @staticmethod
    def wrap(obj):
        if isinstance(obj, dict):
            return Cfg({k: Cfg.wrap(v) for k, v in obj.items()})
        if isinstance(obj, list):
            return [Cfg.wrap(v) for v in obj]
        return obj

    def to_dict(self):
        def un(o):
            if isinstance(o, dict):
                return {k: un(v) for k, v in o.items()}
            if isinstance(o, list):
                return [un(v) for v in o]
            return o
        return un(self)


def _yaml(path):
    with open(path) as f:
        return _parse(f.read()) or {}


def _set(d, dotted, value):
    keys = dotted.split(".")
    for k in keys[:-1]:
        if k not in d or not isinstance(d[k], dict):
            d[k] = {}
        d = d[k]
    d[keys[-1]] = value


def load_config(config_path="conf", config_name="config", overrides=()):
    root = _yaml(os.path.join(config_path, config_name + ".yaml"))
    defaults = root.pop("defaults", [])
    groups = {}
    for entry in defaults:
        if isinstance(entry, dict):
            groups.update(entry)
    value_overrides = []
    for ov in overrides:
        assert "=" in ov, f"override '{ov}' is not key=value"
        k, v = ov.split("=", 1)
        k = k.lstrip("+")
        if "." not in k and k in groups and os.path.exists(os.path.join(config_path, k, v + ".yaml")):
            groups[k] = v
        else:
            value_overrides.append((k, _parse(v)))
    cfg = dict(root)
    for group, option in groups.items():
        if option is None:
            continue
        cfg[group] = _yaml(os.path.join(config_path, group, str(option) + ".yaml"))
    for k, v in value_overrides:
        _set(cfg, k, v)
    return Cfg.wrap(cfg)
