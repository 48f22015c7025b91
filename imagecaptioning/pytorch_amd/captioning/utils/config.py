"""Minimal ``_BASE_``-aware yaml config loader (reference captioning/utils/config.py:34-95 uses yacs, which is
not a dependency here): ``load(path)`` returns a flat dict with base files merged first."""
import os

import yaml


def load(path):
    with open(path) as f:
        cfg = yaml.safe_load(f) or {}
    base = cfg.pop('_BASE_', None)
    out = {}
    if base:
        if not os.path.isabs(base):
            base = os.path.join(os.path.dirname(path), base)
        out.update(load(base))
    out.update(cfg)
    return out
