"""Config helpers: an attribute dict compatible with the reference's use of
``easydict.EasyDict`` (easydict is not installed in this image) and the JSON
defaults of src/experiments/config/*.json."""
from __future__ import annotations

import json
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
TRAINING_CONFIG_PATH = os.path.join(_HERE, "training_config.json")
SIMCLR_CONFIG = os.path.join(_HERE, "simclr_config.json")
PECLR_CONFIG = os.path.join(_HERE, "peclr_config.json")
# The reference's constants.py:20 points at a non-existent simhand_config.json; the shipped
# file is handclr_config.json.  Both names resolve here.
SIMHAND_CONFIG = os.path.join(_HERE, "handclr_config.json")


class edict(dict):
    """Attribute-access dict, nested dicts converted recursively."""

    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in dict(d or {}, **kw).items():
            self[k] = v

    def __setitem__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, edict):
            v = edict(v)
        super().__setitem__(k, v)

    __setattr__ = __setitem__

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def update(self, d=None, **kw):
        for k, v in dict(d or {}, **kw).items():
            self[k] = v


def read_json(path: str) -> dict:
    with open(path, "r") as f:
        return json.load(f)
