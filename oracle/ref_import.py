"""Import the reference's own Python (read-only, /root/reference) on CPU.

TEST INFRASTRUCTURE ONLY, and only usable in the build container: the GPU box
has no /root/reference.  Used by ``oracle/make_golden.py`` to produce the
fixtures under ``tests/golden`` and by ``tests/test_reference_live.py`` (which
skips itself when the reference tree is absent).

The reference needs third-party modules this image lacks:
kornia, comet_ml, torchvision, cv2, yacs, pytorch_lightning, pl_bolts,
easydict.  They are replaced by inert stub modules -- none of their code is on
the arithmetic path we pin -- except:
  * ``easydict.EasyDict``: a 15-line attribute dict;
  * ``pytorch_lightning.core.lightning.LightningModule``: nn.Module + no-op
    ``log`` / ``save_hyperparameters``;
  * ``torchvision.models.resnetXX``: factory hooks that return the oracle's
    torchvision-style ResNet (``oracle.step.TorchvisionStyleResNet``), so the
    reference's ResNetModel / HandCLR_W run unmodified on top of it.
"""
from __future__ import annotations

import contextlib
import importlib
import importlib.abc
import importlib.machinery
import io
import os
import sys
import types

REFERENCE_ROOT = "/root/reference"
_STUB_ROOTS = {"kornia", "comet_ml", "torchvision", "cv2", "yacs", "pytorch_lightning", "pl_bolts",
               "matplotlib", "tensorboard", "pandas_stub_never"}


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "src", "models"))


class _Stub(types.ModuleType):
    """Package-like module whose unknown attributes are child stubs / dummy classes."""

    __path__: list = []

    def __getattr__(self, name: str):
        if name.startswith("__"):
            raise AttributeError(name)
        full = f"{self.__name__}.{name}"
        if name[:1].isupper():
            obj = type(name, (), {"__init__": lambda self, *a, **k: None, "__module__": self.__name__})
        else:
            obj = sys.modules.get(full) or _Stub(full)
            sys.modules[full] = obj
        setattr(self, name, obj)
        return obj


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        if fullname.split(".")[0] in _STUB_ROOTS:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        return sys.modules.get(spec.name) or _Stub(spec.name)

    def exec_module(self, module):
        pass


class EasyDict(dict):
    """Minimal stand-in for easydict.EasyDict (attribute access, recursive)."""

    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in dict(d or {}, **kw).items():
            self[k] = v

    def __setitem__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, EasyDict):
            v = EasyDict(v)
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


_installed = False


def install() -> None:
    """Idempotently prepare sys.path / sys.modules / env for importing ``src.*``."""
    global _installed
    if _installed:
        return
    if not reference_available():
        raise RuntimeError("reference tree not present; golden fixtures are the only pin on this machine")
    import torch
    from torch import nn

    from oracle.step import TorchvisionStyleResNet

    sys.dont_write_bytecode = True  # never write into the read-only reference tree
    os.environ.setdefault("BASE_PATH", REFERENCE_ROOT)
    for k in ("DATA_PATH", "SAVED_MODELS_BASE_PATH", "SAVED_META_INFO_PATH"):
        os.environ.setdefault(k, "/tmp/simhand_oracle_unused")
    sys.meta_path.insert(0, _StubFinder())
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)

    ed = types.ModuleType("easydict")
    ed.EasyDict = EasyDict
    sys.modules["easydict"] = ed

    class LightningModule(nn.Module):
        def __init__(self, *a, **k):
            super().__init__()
            self.logged = {}

        def log(self, name, value, *a, **k):
            self.logged[name] = value

        def save_hyperparameters(self, *a, **k):
            pass

    pl = importlib.import_module("pytorch_lightning")
    core = importlib.import_module("pytorch_lightning.core")
    lightning = importlib.import_module("pytorch_lightning.core.lightning")
    lightning.LightningModule = LightningModule
    core.LightningModule = LightningModule
    pl.LightningModule = LightningModule
    cb = importlib.import_module("pytorch_lightning.callbacks")
    for name in ("Callback", "ModelCheckpoint", "LearningRateMonitor"):
        setattr(cb, name, type(name, (), {"__init__": lambda self, *a, **k: None}))

    tv = importlib.import_module("torchvision")
    tvm = importlib.import_module("torchvision.models")
    tv.models = tvm

    def factory(size):
        def make(pretrained=False, norm_layer=None, **kw):
            # ImageNet weights cannot be downloaded here: seeded random init on both sides
            return TorchvisionStyleResNet(size)
        return make

    for size in ("18", "34", "50", "101", "152"):
        setattr(tvm, f"resnet{size}", factory(size))
    _installed = True


@contextlib.contextmanager
def quiet():
    """The reference prints the whole module on every encoder forward
    (src/models/resnet_model.py:49)."""
    with contextlib.redirect_stdout(io.StringIO()):
        yield


def models_utils():
    install()
    return importlib.import_module("src.models.utils")


def step_class(name: str):
    """HandCLR_W / PeCLR_W / SimCLR / PeCLR / SimCLR_W from the reference."""
    install()
    mod = {
        "HandCLR_W": "src.models.unsupervised.simhand_w_model",
        "PeCLR_W": "src.models.unsupervised.peclr_w_model",
        "SimCLR": "src.models.unsupervised.simclr_model",
        "PeCLR": "src.models.unsupervised.peclr_model",
        "SimCLR_W": "src.models.unsupervised.simclr_w_model",
    }[name]
    return getattr(importlib.import_module(mod), name)
