"""Checkpoint lookup and encoder export with the reference's names ("next" row 8f-3).

* ``get_latest_checkpoint`` / ``get_encoder_state_dict`` -- src/models/utils.py:504-540: checkpoints live under
  ``$SAVED_MODELS_BASE_PATH/<experiment>/checkpoints``; without an explicit name the newest ``epoch=N.ckpt`` is taken
  (the reference sorts by ``int(name[6:-5])``, i.e. it expects Lightning's default file names); the encoder's weights are
  the ``state_dict`` entries whose key contains "encoder", with the leading ``"encoder."`` (8 characters) cut off.
* ``peclr_to_torchvision`` -- src/models/port_model.py:7-48: every ``state_dict`` entry whose key contains "features" is
  copied BY POSITION onto the state-dict entries of a torchvision ResNet (conv1, bn1, layer1..4 in registration order); the
  copy stops at the first pair whose last key component differs.  torchvision is not installed in this image (and not
  vendored by the reference), so the destination here is any module / dict with torchvision's key order --
  ``torchvision_resnet(size)`` builds one (same registration order as torchvision's ResNet, checked in the tests
  against the published key list).
* ``export_torchvision_state_dict`` writes the ``resnet50_simhand.pth`` file hubconf.py:6-23 downloads: a plain
  torchvision-keyed state dict a downstream ``resnet50().load_state_dict`` accepts; ``resnet50_simhand`` is the hub entry.
"""
from __future__ import annotations

import os
from collections import OrderedDict
from typing import Dict, Union

import torch
from torch import nn

from .resnet_model import _Backbone


def saved_models_base_path() -> str:
    """``SAVED_MODELS_BASE_PATH`` of src/constants.py (an environment variable there as well)."""
    return os.environ.get("SAVED_MODELS_BASE_PATH", "./runs")


def get_latest_checkpoint(experiment_name: str, checkpoint: str = "") -> str:
    checkpoint_path = os.path.join(saved_models_base_path(), experiment_name, "checkpoints")
    if checkpoint == "":
        names = os.listdir(checkpoint_path)
        latest = sorted(names, key=lambda x: int(x[6:-5]))[-1]  # "epoch=<N>.ckpt"
    else:
        latest = checkpoint
    return os.path.join(checkpoint_path, latest)


def get_encoder_state_dict(saved_model_path: str, checkpoint: str) -> Dict[str, torch.Tensor]:
    saved = torch.load(get_latest_checkpoint(saved_model_path, checkpoint), map_location="cpu", weights_only=False)["state_dict"]
    return {key[8:]: value for key, value in saved.items() if "encoder" in key}


def torchvision_resnet(size: Union[int, str]) -> nn.Module:
    """A parameter container with torchvision.models.resnet<size>'s modules, names and registration order
    (conv1, bn1, layer1..layer4, fc) -- the destination type of ``peclr_to_torchvision`` when torchvision is absent."""
    return _Backbone(str(size))


def peclr_to_torchvision(resnet_model: Union[nn.Module, "OrderedDict[str, torch.Tensor]"], path_to_peclr_weights: str) -> Dict[str, torch.Tensor]:
    """Positional copy of the checkpoint's ``features`` tensors into ``resnet_model`` (in place); returns its state dict.
    Unlike the reference, a mismatch raises instead of printing and leaving the model half copied."""
    ckpt = torch.load(path_to_peclr_weights, map_location=torch.device("cpu"), weights_only=False)
    src = [(k, v) for k, v in ckpt["state_dict"].items() if "features" in k]
    own = resnet_model.state_dict() if isinstance(resnet_model, nn.Module) else resnet_model
    dst = list(own.items())
    if len(src) > len(dst):
        raise ValueError(f"checkpoint has {len(src)} feature tensors, the ResNet only {len(dst)}: different depth")
    with torch.no_grad():
        for (dk, dv), (sk, sv) in zip(dst, src):
            if dk.split(".")[-1] != sk.split(".")[-1] or tuple(dv.shape) != tuple(sv.shape):
                raise ValueError(f"layers do not match: {sk} {tuple(sv.shape)} -> {dk} {tuple(dv.shape)} (different ResNet size?)")
            dv.copy_(sv)
    return own


def export_torchvision_state_dict(path_to_ckpt: str, out_path: str, resnet_size: Union[int, str] = 50) -> Dict[str, torch.Tensor]:
    """Checkpoint -> torchvision-keyed ``resnet<size>`` state dict (encoder tensors copied, ``fc`` left at its init, as
    after the reference's port) saved at ``out_path``."""
    sd = peclr_to_torchvision(torchvision_resnet(resnet_size), path_to_ckpt)
    sd = OrderedDict((k, v.detach().clone()) for k, v in sd.items())
    torch.save(sd, out_path)
    return sd


def resnet50_simhand(pretrained: bool = False, path: str = "", **kwargs) -> nn.Module:
    """hubconf.py:6-23.  The reference downloads ``resnet50_simhand.pth`` from its GitHub release; there is no network
    here, so ``pretrained=True`` needs the local ``path`` of such a file (e.g. one written by
    ``export_torchvision_state_dict``)."""
    model = torchvision_resnet(50)
    if pretrained:
        if not path:
            raise ValueError("pretrained=True needs path=<resnet50_simhand.pth> (no network access to the release asset)")
        model.load_state_dict(torch.load(path, map_location=torch.device("cpu")))
    return model
