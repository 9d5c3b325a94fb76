"""CPU restatement (torch fp32/fp64, no torchvision / Lightning / easydict) of
the SiMHand contrastive pre-training step.  TEST INFRASTRUCTURE ONLY -- see
``oracle/__init__.py``.

Every function cites the reference file:line it follows (paths relative to
/root/reference).  The arithmetic below is written from the maths in
SURVEY.md Appendix A, not copied from the reference; ``oracle/make_golden.py``
pins it against the reference's own code executed on CPU.

Pieces marked "parity unpinned" restate third-party code that is not vendored
in the reference (torchvision 0.13.1 ``models.resnet``) from its published
definition (He et al. 2015 + the "v1.5" stride-on-3x3 variant).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
from torch import Tensor, nn
from torch.nn import functional as F

TEMPERATURE = 0.5  # default arg, never overridden: src/models/utils.py:157,391,430,468

# --------------------------------------------------------------------------
# a4: backbone.  torchvision 0.13.1 resnet{18,34,50,101,152} -- PARITY UNPINNED
# (module not under /root/reference; call site src/models/resnet_model.py:13-26).
# Structural pins: parameter counts and torchvision key names
# implied by hubconf.py:14-22 / src/models/port_model.py:24-46.
# --------------------------------------------------------------------------


def _conv3x3(cin: int, cout: int, stride: int = 1) -> nn.Conv2d:
    return nn.Conv2d(cin, cout, 3, stride=stride, padding=1, bias=False)


def _conv1x1(cin: int, cout: int, stride: int = 1) -> nn.Conv2d:
    return nn.Conv2d(cin, cout, 1, stride=stride, bias=False)


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes: int, planes: int, stride: int = 1, downsample: Optional[nn.Module] = None):
        super().__init__()
        self.conv1 = _conv3x3(inplanes, planes, stride)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=False)
        self.conv2 = _conv3x3(planes, planes)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x: Tensor) -> Tensor:
        identity = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        if self.downsample is not None:
            identity = self.downsample(x)
        return self.relu(out + identity)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes: int, planes: int, stride: int = 1, downsample: Optional[nn.Module] = None):
        super().__init__()
        width = planes
        self.conv1 = _conv1x1(inplanes, width)
        self.bn1 = nn.BatchNorm2d(width)
        self.conv2 = _conv3x3(width, width, stride)  # v1.5: stride on the 3x3
        self.bn2 = nn.BatchNorm2d(width)
        self.conv3 = _conv1x1(width, planes * self.expansion)
        self.bn3 = nn.BatchNorm2d(planes * self.expansion)
        self.relu = nn.ReLU(inplace=False)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x: Tensor) -> Tensor:
        identity = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        if self.downsample is not None:
            identity = self.downsample(x)
        return self.relu(out + identity)


RESNET_SPECS = {
    "18": (BasicBlock, [2, 2, 2, 2]),
    "34": (BasicBlock, [3, 4, 6, 3]),
    "50": (Bottleneck, [3, 4, 6, 3]),
    "101": (Bottleneck, [3, 4, 23, 3]),
    "152": (Bottleneck, [3, 8, 36, 3]),
}


class TorchvisionStyleResNet(nn.Module):
    """conv1/bn1/relu/maxpool/layer1..4/avgpool/fc with torchvision's names and
    init (kaiming-normal fan_out on convs, BN gamma=1 beta=0, no
    zero-init-residual, default Linear init)."""

    def __init__(self, size: str, num_classes: int = 1000):
        super().__init__()
        block, layers = RESNET_SPECS[str(size)]
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=False)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
        self.layer1 = self._make_layer(block, 64, layers[0], 1)
        self.layer2 = self._make_layer(block, 128, layers[1], 2)
        self.layer3 = self._make_layer(block, 256, layers[2], 2)
        self.layer4 = self._make_layer(block, 512, layers[3], 2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512 * block.expansion, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1.0)
                nn.init.constant_(m.bias, 0.0)

    def _make_layer(self, block, planes: int, blocks: int, stride: int) -> nn.Sequential:
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(
                _conv1x1(self.inplanes, planes * block.expansion, stride),
                nn.BatchNorm2d(planes * block.expansion),
            )
        layers = [block(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes))
        return nn.Sequential(*layers)


class ResNetModelOracle(nn.Module):
    """src/models/resnet_model.py:6-58 (mode="pretraining"): ``features`` =
    Sequential(conv1,bn1,relu,maxpool,layer1..4,AdaptiveAvgPool2d(1)) and an
    unused ``final_layer`` = Linear(C, 21*3+1); forward returns the flattened
    (N, C) embedding.  The reference's per-forward ``print`` is dropped."""

    def __init__(self, size: str):
        super().__init__()
        m = TorchvisionStyleResNet(size)
        self.features = nn.Sequential(
            m.conv1, m.bn1, m.relu, m.maxpool, m.layer1, m.layer2, m.layer3, m.layer4,
            nn.AdaptiveAvgPool2d((1, 1)),
        )
        self.final_layer = nn.Sequential(nn.Linear(m.fc.in_features, 21 * 3 + 1))
        self.out_features = m.fc.in_features

    def forward(self, x: Tensor) -> Tensor:
        return self.features(x).flatten(start_dim=1)


def projection_head(input_dim: int, hidden_dim: int = 512, output_dim: int = 128) -> nn.Sequential:
    """a5: src/models/unsupervised/simclr_model.py:22-39."""
    return nn.Sequential(
        nn.Linear(input_dim, hidden_dim, bias=True),
        nn.BatchNorm1d(hidden_dim),
        nn.ReLU(),
        nn.Linear(hidden_dim, output_dim, bias=False),
    )


# --------------------------------------------------------------------------
# a6-a8: projection post-process
# --------------------------------------------------------------------------


def rownorm(x: Tensor, eps: float = 1e-12) -> Tensor:
    """a6: F.normalize over the flat 128-vector, simhand_w_model.py:57-58,92-93."""
    return x / x.norm(dim=1, keepdim=True).clamp_min(eps)


def translate(points: Tensor, tx: Tensor, ty: Tensor) -> Tensor:
    """a7: src/models/utils.py:661-684.  points (N,64,2); the per-row ranges
    are constants in backward (``.detach()`` at :674-675)."""
    d = points.detach()
    rng = d.max(dim=1).values - d.min(dim=1).values  # (N,2)
    shift = torch.stack((tx.to(points.dtype) * rng[:, 0], ty.to(points.dtype) * rng[:, 1]), dim=1)
    return points + shift[:, None, :]


def rotate(points: Tensor, angle_deg: Tensor) -> Tensor:
    """a8: src/models/utils.py:606-658.  The 3x2 matrix is evaluated in the
    angle's dtype (float64 in the collated batch, App. B) and rounded to fp32
    when it is stored (``torch.zeros((n,3,2))`` at :625); centre = mean of the
    points, detached (:648)."""
    c = points.detach().mean(dim=1)  # (N,2)
    th = angle_deg * math.pi / 180
    alpha, beta = torch.cos(th), torch.sin(th)
    m20 = ((1 - alpha) * c[:, 0] - beta * c[:, 1]).to(points.dtype)
    m21 = ((1 - alpha) * c[:, 1] + beta * c[:, 0]).to(points.dtype)
    a, b = alpha.to(points.dtype)[:, None], beta.to(points.dtype)[:, None]
    x, y = points[..., 0], points[..., 1]
    xr = x * a + y * b + m20[:, None]
    yr = -x * b + y * a + m21[:, None]
    return torch.stack((xr, yr), dim=-1)


def transformed_projections(
    head_out: Tensor,
    jitter_x: Optional[Tensor],
    jitter_y: Optional[Tensor],
    angle: Optional[Tensor],
    image_hw: Tuple[int, int],
) -> Tensor:
    """a3 after the head: simhand_w_model.py:56-93 == peclr_w_model.py:53-90.
    head_out (N,128) with rows cat(view1, view2); jitter_* int64 (N,), angle
    float64 (N,) or None when the flag is off.  Returns unit-norm (N,128)."""
    n = head_out.shape[0]
    p = rownorm(head_out).view(n, -1, 2)
    if jitter_x is not None:
        # jitter_x / H and jitter_y / W (simhand_w_model.py:68-81), negated at :83
        tx = -(jitter_x / float(image_hw[0]))
        ty = -(jitter_y / float(image_hw[1]))
        p = translate(p, tx, ty)
    if angle is not None:
        p = rotate(p, -angle)  # simhand_w_model.py:89
    return rownorm(p.reshape(n, -1))


def projection_stats(points: Tensor, name: str) -> Dict[str, Tensor]:
    """a9: simhand_w_model.py:138-151 on raw head output viewed (B,64,2)."""
    pm = points.mean(dim=1)
    pmed = points.median(dim=1).values
    pmin = points.min(dim=1).values
    pmax = points.max(dim=1).values
    out = {}
    for axis, idx in (("x", 0), ("y", 1)):
        out[f"{name}{axis}_mean"] = pm.mean(dim=0)[idx]
        out[f"{name}{axis}_median"] = pmed.mean(dim=0)[idx]
        out[f"{name}{axis}_min"] = pmin.mean(dim=0)[idx]
        out[f"{name}{axis}_max"] = pmax.mean(dim=0)[idx]
    return out


# --------------------------------------------------------------------------
# a10: adaptive weights
# --------------------------------------------------------------------------


def pos_distance(j1: Tensor, j2: Tensor, diff_type: str) -> Tensor:
    """Per-pair distance d_k, src/models/utils.py:219-231 (== :305-317).
    j (B,21,2) -- or (B,F) for the PCA variants :265-274."""
    d = j1 - j2
    if j1.dim() == 2:  # *_with_pca: all three diff types are the plain L2 norm
        return d.norm(dim=-1)
    if diff_type == "w_o_abs":
        return d.mean(dim=1).norm(dim=1)
    if diff_type == "w_abs":
        return d.abs().mean(dim=1).norm(dim=1)
    if diff_type == "mpjpe":
        return d.norm(dim=-1).mean(dim=1)
    raise ValueError(diff_type)


def neg_distance(j1: Tensor, j2: Tensor, diff_type: str) -> Tensor:
    """All-pairs distance D_ab over cat(J1,J2), src/models/utils.py:237-253.
    NOTE the w_abs / w_o_abs branches average over the xy axis first and then
    take the L2 norm over the 21 joints (:243-244,:248-249) -- a different
    formula from the positive branch."""
    j = torch.cat((j1, j2), dim=0)
    d = j.unsqueeze(1) - j.unsqueeze(0)
    if j.dim() == 2:
        return d.norm(dim=-1)
    if diff_type == "w_o_abs":
        return d.mean(dim=-1).norm(dim=2)
    if diff_type == "w_abs":
        return d.abs().mean(dim=-1).norm(dim=2)
    if diff_type == "mpjpe":
        return d.norm(dim=-1).mean(dim=2)
    raise ValueError(diff_type)


def weights_linear(j1: Tensor, j2: Tensor, diff_type: str) -> Tuple[Tensor, Tensor]:
    """src/models/utils.py:218-261 (and :264-301 when j is (B,F))."""
    dp = pos_distance(j1, j2, diff_type)
    wp = (dp.max() - dp) / (dp.max() - dp.min())
    dn = neg_distance(j1, j2, diff_type)
    wn = (dn.max() - dn) / (dn.max() - dn.min())
    return wp, wn


def weights_nonlinear(j1: Tensor, j2: Tensor, lam_pos: float, lam_neg: float, diff_type: str) -> Tuple[Tensor, Tensor]:
    """src/models/utils.py:304-346 (and :349-388 when j is (B,F))."""
    dp = pos_distance(j1, j2, diff_type)
    wp = 1 / (1 + torch.exp(torch.tensor(lam_pos) * (dp - dp.mean())))
    dn = neg_distance(j1, j2, diff_type)
    wn = 1 / (1 + torch.exp(torch.tensor(lam_neg) * (dn - dn.mean())))
    return wp, wn


# --------------------------------------------------------------------------
# a11/a12: NT-Xent variants and the closed-form gradient
# --------------------------------------------------------------------------


def ntxent(z1: Tensor, z2: Tensor, w_pos: Optional[Tensor] = None, w_neg: Optional[Tensor] = None,
           temperature: float = TEMPERATURE) -> Tensor:
    """src/models/utils.py:157-189 (both None), :391-427 (both), :430-465
    (w_pos only), :468-501 (w_neg only).  The positive column stays inside the
    negative sum and is weighted by w_neg (App. D #9)."""
    z = torch.cat((z1, z2), dim=0)
    n = z.shape[0]
    s = z @ z.t()
    if w_neg is not None:
        s = s * w_neg
    e = torch.exp(s / temperature)
    neg = e.sum(dim=1) - e.diagonal()
    sp = (z1 * z2).sum(dim=-1)
    if w_pos is not None:
        sp = sp * w_pos
    sp = torch.cat((sp, sp), dim=0)
    return -(sp / temperature - torch.log(neg)).mean()


def ntxent_reference_order(z1: Tensor, z2: Tensor, w_pos: Optional[Tensor], w_neg: Optional[Tensor],
                           temperature: float = TEMPERATURE) -> Tensor:
    """Same value as :func:`ntxent`, evaluated in the reference's operation
    order (masked_select of the off-diagonal, log of the ratio) so fp32
    round-off matches to the last bits; used to pin the golden vectors."""
    z = torch.cat((z1, z2), dim=0)
    n = z.shape[0]
    cov = z @ z.t()
    if w_neg is not None:
        cov = cov * w_neg
    sim = torch.exp(cov / temperature)
    mask = ~torch.eye(n, dtype=torch.bool)
    neg = sim.masked_select(mask).view(n, -1).sum(dim=-1)
    pos = (z1 * z2).sum(dim=-1)
    if w_pos is not None:
        pos = pos * w_pos
    pos = torch.exp(pos / temperature)
    pos = torch.cat((pos, pos), dim=0)
    return -torch.log(pos / neg).mean()


def ntxent_closed_form(z: Tensor, w_pos: Optional[Tensor], w_neg: Optional[Tensor],
                       temperature: float = TEMPERATURE) -> Tuple[Tensor, Tensor, Tensor]:
    """Loss, d loss / d z (N,128) and the per-row negative sums from the
    closed form of SURVEY a12 / Appendix A:
      G_ij = (1/N) [w_ij e^{w_ij s_ij/t} / (t neg_i)]_{j!=i} - (1/N)(w+_i/t)[j=pair(i)]
      dL/dz = (G + G^T) z.
    Independent of autograd; used to check the HIP backward."""
    n = z.shape[0]
    b = n // 2
    s = z @ z.t()
    w = torch.ones_like(s) if w_neg is None else w_neg
    e = torch.exp(w * s / temperature)
    e = e - torch.diag(e.diagonal())
    neg = e.sum(dim=1)
    wp = torch.ones(b, dtype=z.dtype) if w_pos is None else w_pos
    wp2 = torch.cat((wp, wp))
    pair = torch.cat((torch.arange(b, n), torch.arange(0, b)))
    spos = (z * z[pair]).sum(dim=1)
    loss = -(wp2 * spos / temperature - torch.log(neg)).mean()
    g = (w * e) / (temperature * neg[:, None]) / n
    g[torch.arange(n), pair] -= wp2 / temperature / n
    dz = (g + g.t()) @ z
    return loss, dz, neg


# --------------------------------------------------------------------------
# a1-a3: the whole step (used by the end-to-end parity tests and as the timed
# CPU baseline in bench.py)
# --------------------------------------------------------------------------


class StepOracle(nn.Module):
    """HandCLR_W / PeCLR_W / SimCLR step on CPU.

    ``experiment`` in {"simclr", "peclr", "peclr_w", "simhand_w", "simclr_w"};
    config keys as produced by src/experiments/utils.py:725-755 + main.py:127-129.
    state_dict keys equal the reference's (``encoder.features.*``,
    ``encoder.final_layer.0.*``, ``projection_head.{0,1,3}.*``).
    """

    def __init__(self, experiment: str, resnet_size: str = "18", augmentation: Sequence[str] = (),
                 weight_type: str = "linear", diff_type: str = "mpjpe", pos_neg: str = "pos_neg",
                 lambda_pos: float = 5.0, lambda_neg: float = 0.05, hidden_dim: int = 512, output_dim: int = 128):
        super().__init__()
        self.experiment = experiment
        self.encoder = ResNetModelOracle(resnet_size)
        # quirk honoured: input dim follows the encoder, not the JSON's 2048
        self.projection_head = projection_head(self.encoder.out_features, hidden_dim, output_dim)
        self.augmentation = list(augmentation)
        self.weight_type, self.diff_type, self.pos_neg = weight_type, diff_type, pos_neg
        self.lambda_pos, self.lambda_neg = lambda_pos, lambda_neg
        self.last: Dict[str, Tensor] = {}

    @property
    def unwarps(self) -> bool:
        return self.experiment in ("peclr", "peclr_w", "simhand_w", "handclr_w")

    @property
    def weighted(self) -> bool:
        return self.experiment in ("peclr_w", "simhand_w", "handclr_w", "simclr_w")

    def embed(self, images: Tensor) -> Tuple[Tensor, Tensor]:
        enc = self.encoder(images)
        return enc, self.projection_head(enc)

    def contrastive_step(self, batch: Dict[str, Tensor]) -> Tensor:
        x = torch.cat((batch["transformed_image1"], batch["transformed_image2"]), dim=0)
        hw = tuple(batch["transformed_image1"].shape[-2:])
        b = x.shape[0] // 2
        enc, p = self.embed(x)
        self.last = {"encoding": enc, "head_out": p}
        if self.unwarps:
            jx = jy = ang = None
            if "crop" in self.augmentation:
                jx = torch.cat((batch["jitter_x_1"], batch["jitter_x_2"]))
                jy = torch.cat((batch["jitter_y_1"], batch["jitter_y_2"]))
            if "rotate" in self.augmentation:
                ang = torch.cat((batch["angle_1"], batch["angle_2"]))
            z = transformed_projections(p, jx, jy, ang, hw)
        else:
            z = rownorm(p)
        self.last["z"] = z
        wp = wn = None
        if self.weighted:
            j1 = batch["joints1_aug"][:, :, :2]
            j2 = batch["joints2_aug"][:, :, :2]
            if self.weight_type == "linear":
                wp, wn = weights_linear(j1, j2, self.diff_type)
            else:
                wp, wn = weights_nonlinear(j1, j2, self.lambda_pos, self.lambda_neg, self.diff_type)
            if self.pos_neg == "pos":
                wn = None
            elif self.pos_neg == "neg":
                wp = None
        return ntxent(z[:b], z[b:], wp, wn)


def sharded_step(model: "StepOracle", batch: Dict[str, Tensor], ranks: int) -> Tuple[Tensor, Tensor]:
    """what R data-parallel ranks must compute together.  Encoder + head + un-warp run shard by shard
    (contiguous pair ranges, both views of a pair in one shard; train-mode BatchNorm statistics per shard, as in the
    reference's DataParallel replicas, src/experiments/main.py:152-163), the projections are concatenated in the
    reference's row order cat(all view-1, all view-2) and the weights + NT-Xent run ONCE over the global batch
    (== the reference's single-device loss at the global batch).  Returns (loss, z (2B,128))."""
    b = batch["transformed_image1"].shape[0]
    if b % ranks:
        raise ValueError(f"global batch {b} is not divisible by {ranks} ranks")
    bl = b // ranks
    hw = tuple(batch["transformed_image1"].shape[-2:])
    z1s, z2s = [], []
    for r in range(ranks):
        sub = {k: v[r * bl:(r + 1) * bl] for k, v in batch.items()}
        x = torch.cat((sub["transformed_image1"], sub["transformed_image2"]), dim=0)
        _, p = model.embed(x)
        if model.unwarps:
            jx = jy = ang = None
            if "crop" in model.augmentation:
                jx = torch.cat((sub["jitter_x_1"], sub["jitter_x_2"]))
                jy = torch.cat((sub["jitter_y_1"], sub["jitter_y_2"]))
            if "rotate" in model.augmentation:
                ang = torch.cat((sub["angle_1"], sub["angle_2"]))
            z = transformed_projections(p, jx, jy, ang, hw)
        else:
            z = rownorm(p)
        z1s.append(z[:bl])
        z2s.append(z[bl:])
    z1, z2 = torch.cat(z1s), torch.cat(z2s)
    wp = wn = None
    if model.weighted:
        j1 = batch["joints1_aug"][:, :, :2]
        j2 = batch["joints2_aug"][:, :, :2]
        if model.weight_type == "linear":
            wp, wn = weights_linear(j1, j2, model.diff_type)
        else:
            wp, wn = weights_nonlinear(j1, j2, model.lambda_pos, model.lambda_neg, model.diff_type)
        if model.pos_neg == "pos":
            wn = None
        elif model.pos_neg == "neg":
            wp = None
    return ntxent(z1, z2, wp, wn), torch.cat((z1, z2))


# --------------------------------------------------------------------------
# synthetic batch -- shared by tests, smoke() and bench.py
# --------------------------------------------------------------------------


def synthetic_batch(b: int, size: int = 224, seed: int = 5, device: str = "cpu") -> Dict[str, Tensor]:
    """Deterministic synthetic batch with the schema of SURVEY Appendix B:
    images ~ N(0,1); joints xy ~ U(0,size), z = 1; view-2 joints = view-1 +
    N(0, 8^2); integer angles U{-45..45} as float64; integer jitters
    -U{0..14} as int64."""
    g = torch.Generator().manual_seed(seed)
    j1 = torch.rand(b, 21, 3, generator=g) * size
    j1[:, :, 2] = 1.0
    j2 = j1.clone()
    j2[:, :, :2] += torch.randn(b, 21, 2, generator=g) * 8.0
    batch = {
        "transformed_image1": torch.randn(b, 3, size, size, generator=g),
        "transformed_image2": torch.randn(b, 3, size, size, generator=g),
        "joints1_aug": j1,
        "joints2_aug": j2,
        "joints1_ori": j1 / size,
        "joints2_ori": j2 / size,
        "angle_1": torch.randint(-45, 46, (b,), generator=g).to(torch.float64),
        "angle_2": torch.randint(-45, 46, (b,), generator=g).to(torch.float64),
        "jitter_x_1": -torch.randint(0, 15, (b,), generator=g),
        "jitter_x_2": -torch.randint(0, 15, (b,), generator=g),
        "jitter_y_1": -torch.randint(0, 15, (b,), generator=g),
        "jitter_y_2": -torch.randint(0, 15, (b,), generator=g),
    }
    return {k: v.to(device) for k, v in batch.items()}
