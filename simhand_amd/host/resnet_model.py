"""Encoder: ``ResNetModel`` with the reference's interface, executed by HIP kernels.

Mirrors src/models/resnet_model.py:6-58 of the reference (``features`` =
Sequential(conv1, bn1, relu, maxpool, layer1..4, AdaptiveAvgPool2d(1)) indexed
0..8, unused ``final_layer`` = Linear(C, 64); ``forward`` returns the flattened
(N, C) embedding in mode "pretraining") on top of a torchvision-v1.5-style
ResNet-18/34/50/101/152 (torchvision 0.13.1 is not vendored by the reference;
structure restated from its published definition).  The nn.Conv2d /
nn.BatchNorm2d modules below are PARAMETER CONTAINERS ONLY (state_dict keys,
init, optimizer param groups): their ``forward`` is never called.  All
arithmetic runs in ``ResNetEngine`` through the C ABI:

  stem  : NCHW fp32 -> zero-padded NHWC4 -> direct 7x7/2 MFMA GEMM, K = 8 filter rows x (8 taps x 4 ch)
          (+fused BN partial sums; no im2col matrix) -> BN finalize -> BN-apply+ReLU -> MaxPool(3,2,1)
  block : conv (implicit GEMM, fused BN partial sums) -> BN finalize ->
          BN-apply(+ReLU)(+residual) ...
  tail  : global average pool -> fp32 (N, C)

and the hand-written backward (BN bwd -> wgrad + dgrad per conv, residual
gradients merged by the dgrad epilogue's accumulate flag).  Activations are
NHWC in the compute dtype (fp32 parity mode or bf16), weights are re-packed
OIHW fp32 -> KRSC / CRSK compute dtype when the parameter version changes.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch
from torch import Tensor, nn

from .. import _lib, ops

_H16 = (torch.bfloat16, torch.float16)  # the 16-bit storage formats (which one: the library build, _lib.use_half)



# --------------------------------------------------------------------------
# parameter containers with torchvision's module / key names
# --------------------------------------------------------------------------
def _conv(cin, cout, k, stride=1, pad=0) -> nn.Conv2d:
    return nn.Conv2d(cin, cout, k, stride=stride, padding=pad, bias=False)


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = _conv(inplanes, planes, 3, stride, 1)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU()
        self.conv2 = _conv(planes, planes, 3, 1, 1)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample

    def units(self):
        return [(self.conv1, self.bn1), (self.conv2, self.bn2)]


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = _conv(inplanes, planes, 1)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = _conv(planes, planes, 3, stride, 1)  # v1.5: stride on the 3x3
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = _conv(planes, planes * 4, 1)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU()
        self.downsample = downsample

    def units(self):
        return [(self.conv1, self.bn1), (self.conv2, self.bn2), (self.conv3, self.bn3)]


_SPECS = {"18": (BasicBlock, [2, 2, 2, 2]), "34": (BasicBlock, [3, 4, 6, 3]), "50": (Bottleneck, [3, 4, 6, 3]),
          "101": (Bottleneck, [3, 4, 23, 3]), "152": (Bottleneck, [3, 8, 36, 3])}


class _Backbone(nn.Module):
    """Same construction order as torchvision's ResNet so a given torch seed
    yields the same initial weights."""

    def __init__(self, size: str):
        super().__init__()
        if size not in _SPECS:
            raise NotImplementedError(size)
        block, layers = _SPECS[size]
        self.inplanes = 64
        self.conv1 = _conv(3, 64, 7, 2, 3)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU()
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = self._make(block, 64, layers[0], 1)
        self.layer2 = self._make(block, 128, layers[1], 2)
        self.layer3 = self._make(block, 256, layers[2], 2)
        self.layer4 = self._make(block, 512, layers[3], 2)
        self.fc = nn.Linear(512 * block.expansion, 1000)  # replaced by the wrapper; kept for init-stream parity
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1.0)
                nn.init.constant_(m.bias, 0.0)

    def _make(self, block, planes, blocks, stride):
        down = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            down = nn.Sequential(_conv(self.inplanes, planes * block.expansion, 1, stride), nn.BatchNorm2d(planes * block.expansion))
        mods = [block(self.inplanes, planes, stride, down)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            mods.append(block(self.inplanes, planes))
        return nn.Sequential(*mods)


# --------------------------------------------------------------------------
# engine
# --------------------------------------------------------------------------
class _Packed:
    __slots__ = ("krsc", "crsk", "version", "w", "stem")


class _Unit:
    """Saved tensors of one conv+BN unit for the backward pass."""
    __slots__ = ("conv", "bn", "desc", "x", "y", "a", "st", "relu", "stem", "has_res", "mask", "x_in", "s2", "t2", "ws2", "xq")


class _Pending:
    """A unit whose BatchNorm + ReLU has not been applied yet: the next 3x3 convolution applies it inside its LDS ring
    (ops.conv2d_fwd_bnin) and hands the activation back as a by-product."""
    __slots__ = ("y", "st", "unit")

    def __init__(self, y, st, unit):
        self.y, self.st, self.unit = y, st, unit

    @property
    def shape(self):
        return self.y.shape


class ResNetEngine:
    def __init__(self, features: nn.Sequential, dtype: torch.dtype = torch.float32, fp8: bool = False):
        self.features = features
        self.dtype = dtype
        # fp8 slice (BASELINE configs[4]): bf16 storage everywhere, e4m3 operands + the scaled K = 128 MFMA for the FORWARD of
        # the matrix-core-bound layers (3x3, and 1x1 with >= 512 input channels) whose BatchNorm is not folded; data / weight
        # gradients stay bf16.  Scales: ops.FP8Scaler (per-tensor; weights current, activations delayed).
        self.fp8 = fp8
        self.fp8_all = False  # A/B: the round-2 fp8 set (every 3x3 and every 1x1 with >= 512 input channels)
        self.fp8_wgrad = True  # fp8 configuration: e4m3 weight gradient of the 3x3 / stride-1 layers with >= 256 channels (A/B attribute)
        self._fp8_sites: Dict[int, tuple] = {}  # id(conv.weight) -> (activation scaler, weight scaler, packed weights, version)
        self._fp8_pre = None  # (activation tensor, its e4m3 codes) emitted by the BatchNorm-apply in front of an fp8 convolution
        self._packs: Dict[int, _Packed] = {}
        self._pack_plan = None  # (key, ops.PackPlan) of the one-launch re-pack
        # BN-backward partial sums of a unit fused into the epilogue of the dgrad that produces its incoming gradient
        self.fuse_bn_bwd = True
        # bottleneck conv3 + bn3: BatchNorm backward folded into the 1x1 conv's own gradients (_unit3_bwd_folded)
        self._fold_bn3 = True
        # both terms of the folded input gradient in one launch where the tile kernels take a second K segment
        self.concat_fold = True
        # the folds need every block's incoming gradient in masked form, which only the all-1x1 tails of Bottleneck nets give
        self._bottleneck = all(isinstance(b, Bottleneck) for li in (4, 5, 6, 7) for b in features[li])
        # BN-apply (+ReLU) of the unit in front of a folded conv runs inside the Gram launch the fold needs anyway
        self.fuse_apply_gram = True  # (plain attributes: A/B runs set them, e.g. bench.py --engine fuse_apply_gram=0)
        # BN-backward apply of a 1x1 / stride-1 unit without residual runs inside that unit's weight-gradient launch
        # (off: measured on MI355X at 2048 x 224^2 the BatchNorm class drops 4.3 ms but the weight-gradient class grows 8.5 ms --
        # every cin tile of the launch re-derives the dy operand from TWO tensors; kept for the experiment record)
        self._fuse_bwd_apply_wgrad = False
        # BN-backward apply of a Bottleneck's conv1 + bn1 runs in the A-operand load of conv1's DATA gradient (the activation-
        # stationary kernel loads each gradient row exactly once, straight into MFMA operand registers): the stand-alone pass
        # (2 reads + 1 write of the narrow tensor) becomes 1 extra read + 1 write inside that launch; the weight gradient then
        # reads the dy it wrote (attribute: A/B timing only)
        self._fuse_bwd_apply_dgrad = True
        self._gram = None  # (activation tensor, a^T a, sum a) of the unit just applied that way
        # the next block's conv1 chained onto this block's conv3 + bn3 + add + ReLU launch (ops.conv2d_fwd_bnact_chain): the block
        # output is written but not read back by that conv1 (attribute: A/B timing only)
        self.chain_conv1 = True
        self._chain = None  # (block output tensor, the chained conv module, its raw output, its BatchNorm partial sums)
        # folded stride-2 shortcut convolutions run as dense 1x1 / stride-1 launches over the subsampled input (attribute: A/B timing only)
        self.dense_shortcut = True
        # backward of a stage-entry block: the shortcut's dense data gradient is merged into the main branch's conv1 data gradient
        # (sh_dgrad_opts.sub_grad) instead of scatter-added onto it afterwards (attribute: A/B timing only)
        self.merge_shortcut = True
        # bn1 + ReLU of a Bottleneck applied inside conv2's launch where that 3x3 kernel keeps its activation rows in an LDS ring (the
        # 64- and 128-channel stride-1 layers): the bn_apply pass of those units disappears (attribute: A/B timing only)
        self.bn_on_load = True
        # multi-GPU: host.dist.OverlappedGradReducer -- finished parameter gradients go out block by block during backward
        self.grad_reducer = None

    # Synchronised BatchNorm (ops.set_bn_sync, SURVEY 8e "optional SyncBN"): the statistics of every BatchNorm are all-reduced in its
    # finalize step, so the forms that never materialise those sums as per-block partials -- the Gram-matrix fold of conv3 + bn3 / the
    # shortcut, the BatchNorm-backward apply fused into a consumer's operand load -- are off; the fused partial sums in the dgrad
    # epilogues stay (their finalize step is the synchronisation point).
    @property
    def fold_bn3(self) -> bool:
        return self._fold_bn3 and not ops.bn_sync_active()

    @fold_bn3.setter
    def fold_bn3(self, on: bool) -> None:
        self._fold_bn3 = bool(on)

    @property
    def fuse_bwd_apply_wgrad(self) -> bool:
        return self._fuse_bwd_apply_wgrad and not ops.bn_sync_active()

    @fuse_bwd_apply_wgrad.setter
    def fuse_bwd_apply_wgrad(self, on: bool) -> None:
        self._fuse_bwd_apply_wgrad = bool(on)

    @property
    def fuse_bwd_apply_dgrad(self) -> bool:
        return self._fuse_bwd_apply_dgrad and not ops.bn_sync_active()

    @fuse_bwd_apply_dgrad.setter
    def fuse_bwd_apply_dgrad(self, on: bool) -> None:
        self._fuse_bwd_apply_dgrad = bool(on)

    # -- weights -------------------------------------------------------------
    def _pack(self, conv: nn.Conv2d, need_t: bool, stem: bool = False) -> _Packed:
        w = conv.weight
        p = self._packs.get(id(w))
        ver = (w._version, w.data_ptr(), self.dtype)
        if p is None or p.version != ver:
            p = _Packed()
            if stem:  # [64][256]: column r*32 + tap*4 + c of the direct stem kernel
                p.krsc = ops.stem_pack_weights(w.detach(), self.dtype)
            else:
                p.krsc = ops.pack_krsc(w.detach(), self.dtype)
            p.crsk = None
            p.version = ver
            p.w, p.stem = w, stem
            self._packs[id(w)] = p
        if need_t and p.crsk is None:
            p.crsk = ops.pack_crsk(w.detach(), self.dtype)
        return p

    def _repack_stale(self) -> None:
        """After an optimizer step every packed copy is stale: re-pack all of them (KRSC, and CRSK where a data gradient asked for
        one before) in ONE launch instead of two per convolution (ops.pack_weights_multi); _pack() then finds them current."""
        stale = [p for p in self._packs.values()
                 if not p.stem and p.version != (p.w._version, p.w.data_ptr(), self.dtype) and p.version[1:] == (p.w.data_ptr(), self.dtype)]
        if len(stale) < 2:
            return
        key = tuple((p.w.data_ptr(), p.krsc.data_ptr(), 0 if p.crsk is None else p.crsk.data_ptr()) for p in stale)
        if self._pack_plan is None or self._pack_plan[0] != key:
            self._pack_plan = (key, ops.PackPlan([(p.w.detach(), p.krsc, p.crsk) for p in stale], self.dtype))
        ops.pack_weights_multi(self._pack_plan[1])
        for p in stale:
            p.version = (p.w._version, p.w.data_ptr(), self.dtype)

    # -- forward ---------------------------------------------------------------
    def _bn(self, bn: nn.BatchNorm2d, part, m, c, training):
        if training:
            return ops.bn_finalize(part, m, c, bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var,
                                   bn.num_batches_tracked, None, bn.eps, bn.momentum)
        return ops.bn_eval_state(c, bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var, bn.eps)

    def _fold_fwd_ok(self, conv, relu, residual) -> bool:
        """1x1 convolutions with a narrow input whose BatchNorm statistics follow from the input's Gram matrix: a
        Bottleneck's conv3 (+ identity + ReLU) and its shortcut conv (no ReLU)."""
        return (self.fold_bn3 and self._bottleneck and self.dtype in _H16 and conv.kernel_size == (1, 1) and conv.padding == (0, 0)
                and conv.in_channels % 64 == 0 and conv.out_channels % 64 == 0 and conv.out_channels >= 2 * conv.in_channels
                and (residual is not None or not relu))

    def _conv_bn_folded(self, conv, bn, x, relu, residual, training, save: Optional[list], chain_conv=None):
        """y = conv1x1(x), out = act(bn(y) (+ residual)) in ONE pass over the wide tensor: train-mode batch statistics of y
        come from the Gram matrix of the (narrow) input -- mean_c = W_c . sum(x) / M, E[y^2]_c = W_c^T (x^T x) W_c / M --
        so they are known before the convolution runs and BN + residual + ReLU live in its epilogue; y itself is never
        stored (the folded backward does not need it either)."""
        n, h, w, cin = x.shape
        s = conv.stride[0]
        cout = conv.out_channels
        d = ops.conv_desc(n, h, w, cin, cout, 1, 1, s, 0, self.dtype)
        pk = self._pack(conv, need_t=False)
        x_in = x if s == 1 else ops.subsample2(x)
        m = n * d.ho * d.wo
        s2 = t2 = ws2 = None
        if training:
            if self._gram is not None and self._gram[0] is x_in:            # came out of the producer's fused BN-apply + Gram launch
                s2, t2 = self._gram[1], self._gram[2]
                self._gram = None                                            # consumed (a stage-entry block's shortcut conv, which runs
            else:                                                            # between conv2 and conv3, must not throw conv3's Gram away)
                dww = ops.conv_desc(n, d.ho, d.wo, cin, cin, 1, 1, 1, 0, self.dtype)
                s2, t2 = ops.conv2d_wgrad_colsum(dww, x_in, x_in)            # x^T x (fp32) and sum x ride on one kernel
            # sum y = W . sum x, sum y^2 = rowdot(W S2, W) with the weights the MFMAs see -> statistics, running stats, W S2
            st, ws2 = ops.bn_fold_fwd(conv.weight.detach().view(cout, cin), True, s2, t2, m, bn.weight.detach(), bn.bias.detach(),
                                      bn.running_mean, bn.running_var, bn.num_batches_tracked, bn.eps, bn.momentum)
        else:
            st = ops.bn_eval_state(cout, bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var, bn.eps)
        want_mask = save is not None and relu and residual is not None
        chain = (self.chain_conv1 and chain_conv is not None and want_mask and training and s == 1 and chain_conv.kernel_size == (1, 1)
                 and chain_conv.stride == (1, 1) and chain_conv.padding == (0, 0) and chain_conv.in_channels == cout
                 and chain_conv.out_channels == cin and ops.conv2d_fwd_chain_ok(d))
        if chain:
            a, mask, cy, cpart = ops.conv2d_fwd_bnact_chain(d, x, pk.krsc, st, residual, self._pack(chain_conv, need_t=True).krsc)
            self._chain = (a, chain_conv, cy, cpart)
        else:
            if s == 2 and self.dense_shortcut:
                # the subsampled copy exists anyway (Gram launch, backward): a dense stride-1 1x1 over it instead of a strided walk over x
                dd = ops.conv_desc(n, d.ho, d.wo, cin, cout, 1, 1, 1, 0, self.dtype)
                res = ops.conv2d_fwd_bnact(dd, x_in, pk.krsc, st, relu, residual, want_mask=want_mask)
            else:
                res = ops.conv2d_fwd_bnact(d, x, pk.krsc, st, relu, residual, want_mask=want_mask)
            a, mask = res if want_mask else (res, None)
        if save is not None:
            u = _Unit()
            u.conv, u.bn, u.desc, u.x, u.y, u.a, u.st, u.relu, u.stem = conv, bn, d, x, None, a, st, relu, False
            u.has_res = residual is not None
            u.mask = mask
            u.x_in, u.s2, u.t2, u.ws2 = x_in, s2, t2, ws2
            save.append(u)
        return a

    def _fp8_ok(self, conv, d) -> bool:
        """fp8 forward where it measured FASTER than the bf16 kernel it replaces (ops.conv2d_fwd_fp8_pays: the 3x3 layers with >= 256
        channels, on the e4m3 variant of the 256 x 256 LDS-DMA kernel); engine.fp8_all = True restores the round-2 set (every 3x3 and
        every 1x1 with >= 512 input channels) for A/B runs."""
        if not (self.fp8 and self.dtype == torch.bfloat16):
            return False
        if self.fp8_all:
            return (conv.kernel_size == (3, 3) or conv.in_channels >= 512) and ops.conv2d_fwd_fp8_supported(d)
        return ops.conv2d_fwd_fp8_pays(d)

    def _fp8_next_ok(self, conv, y) -> bool:
        n, h, w, cin = y.shape
        k, s_, p_ = conv.kernel_size[0], conv.stride[0], conv.padding[0]
        return self._fp8_ok(conv, ops.conv_desc(n, h, w, cin, conv.out_channels, k, k, s_, p_, self.dtype))

    def _fp8_site(self, conv, device):
        w = conv.weight
        site = self._fp8_sites.get(id(w))
        if site is None:
            site = [ops.FP8Scaler(device, delayed=True), ops.FP8Scaler(device, delayed=False), None, None]
            self._fp8_sites[id(w)] = site
        return site

    # -- fp8 scaling state <-> checkpoints (delayed-scaling amax rings; the e4m3 weight copies are rebuilt from the masters) ------------
    def fp8_state_dict(self) -> dict:
        names = {id(p): k for k, p in self.features.named_parameters()}
        out = {}
        for key, site in self._fp8_sites.items():
            wid, tag = (key[1], "bwd") if isinstance(key, tuple) else (key, "fwd")
            if wid in names:
                out[f"{tag}:{names[wid]}"] = {"state": site[0].state.detach().cpu().clone(), "calls": site[0].calls}
        return out

    def load_fp8_state_dict(self, state: dict, device) -> None:
        params = dict(self.features.named_parameters())
        for key, rec in (state or {}).items():
            tag, name = key.split(":", 1)
            if name not in params:
                continue
            wid = id(params[name])
            site = [ops.FP8Scaler(device, delayed=True), ops.FP8Scaler(device, delayed=False), None, None]
            site[0].state.copy_(rec["state"].to(device))
            site[0].calls = int(rec["calls"])
            self._fp8_sites[("bwd", wid) if tag == "bwd" else wid] = site

    def _fp8_site_bwd(self, conv, device):
        """[dy scaler (delayed), weight scaler (current), e4m3 CRSK weights, version] of one fp8 data-gradient site."""
        w = conv.weight
        site = self._fp8_sites.get(("bwd", id(w)))
        if site is None:
            site = [ops.FP8Scaler(device, delayed=True), ops.FP8Scaler(device, delayed=False), None, None]
            self._fp8_sites[("bwd", id(w))] = site
        ver = (w._version, w.data_ptr())
        if site[3] != ver:
            site[2] = ops.fp8_pack_crsk(site[1], w)
            site[3] = ver
        return site

    def _conv_fwd_fp8(self, conv, d, x, training):
        w = conv.weight
        site = self._fp8_site(conv, x.device)
        ver = (w._version, w.data_ptr())
        if site[3] != ver:  # new parameter version: re-pack (and re-scale) the e4m3 weights
            site[2] = site[1].pack_weights(w)
            site[3] = ver
        pre, self._fp8_pre = self._fp8_pre, None
        if pre is not None and pre[0] is x:  # its e4m3 codes came out of the producing BatchNorm-apply pass
            xq = pre[1]
        else:
            xq = site[0].quantize(x)
        # kept with the unit: the e4m3 weight gradient reads the same codes.  Their de-scale is SNAPSHOT here (state[0:2], one stream-ordered
        # 8-byte copy): the live scaler's state[1] is overwritten by every later quantisation of this site -- a second forward, an eval
        # forward or a recompute between this forward and its backward would otherwise de-scale dW by the wrong factor
        self._fp8_xq = (xq, site[0].state[:2].clone())
        return ops.conv2d_fwd_fp8(d, xq, site[2], site[0], site[1], want_stats=training)

    def _conv_bn(self, conv, bn, x, relu, residual, training, save: Optional[list], need_dgrad=True, gram_next=False, chain_conv=None,
                 fp8_next=None, bnin_next=None):
        """gram_next: the activation feeds a folded 1x1 convolution (which needs a^T a and sum a): BatchNorm-apply + ReLU then
        run inside that Gram launch (ops.bn_apply_gram) instead of as a pass of their own."""
        if self._fold_fwd_ok(conv, relu, residual):
            return self._conv_bn_folded(conv, bn, x, relu, residual, training, save, chain_conv=chain_conv)
        pend = x if isinstance(x, _Pending) else None
        n, h, w, cin = x.shape
        k, s, p = conv.kernel_size[0], conv.stride[0], conv.padding[0]
        d = ops.conv_desc(n, h, w, cin, conv.out_channels, k, k, s, p, self.dtype)
        pk = self._pack(conv, need_t=save is not None and need_dgrad)
        chained, self._chain = self._chain, None
        self._fp8_xq = None
        if pend is not None:
            # the unit in front left its BatchNorm + ReLU to this launch; its activation comes back as a by-product
            x, y, part = ops.conv2d_fwd_bnin(d, pend.y, pend.st, pk.krsc, want_stats=training)
            if pend.unit is not None:
                pend.unit.a = x
        elif chained is not None and chained[0] is x and chained[1] is conv:
            y, part = chained[2], chained[3]  # computed by the previous block's conv3 launch from its output chunks
        elif self._fp8_ok(conv, d):
            y, part = self._conv_fwd_fp8(conv, d, x, training)
        else:
            y, part = ops.conv2d_fwd(d, x, pk.krsc, want_stats=training)
        m = n * d.ho * d.wo
        st = self._bn(bn, part, m, conv.out_channels, training)
        mask = None
        defer = False
        if bnin_next is not None and self.bn_on_load and relu and residual is None and not gram_next and self.dtype in _H16:
            k2, s2, p2 = bnin_next.kernel_size[0], bnin_next.stride[0], bnin_next.padding[0]
            d2 = ops.conv_desc(n, d.ho, d.wo, conv.out_channels, bnin_next.out_channels, k2, k2, s2, p2, self.dtype)
            defer = ops.conv2d_fwd_bnin_ok(d2) and not self._fp8_ok(bnin_next, d2)
        if defer:
            a = None  # filled in by the consumer (u.a, below)
        elif save is not None and residual is not None and relu:
            a, mask = ops.bn_apply(y, st, m, conv.out_channels, relu, residual, want_mask=True)
        elif gram_next and training and residual is None and self.fuse_apply_gram and self.dtype in _H16:
            a, s2, t2 = ops.bn_apply_gram(y, st, relu)
            self._gram = (a, s2, t2)
        elif fp8_next is not None and residual is None and self._fp8_next_ok(fp8_next, y):
            # the consumer is an fp8 convolution: its e4m3 operand leaves this BatchNorm-apply pass (no separate quantise pass)
            a, xq = self._fp8_site(fp8_next, y.device)[0].bn_apply_quantize(y, st, relu)
            self._fp8_pre = (a, xq)
        else:
            a = ops.bn_apply(y, st, m, conv.out_channels, relu, residual)
        if save is not None:
            u = _Unit()
            u.conv, u.bn, u.desc, u.x, u.y, u.a, u.st, u.relu, u.stem = conv, bn, d, x, y, a, st, relu, False
            u.has_res = residual is not None
            u.mask = mask
            u.x_in = u.s2 = u.t2 = u.ws2 = None
            u.xq = self._fp8_xq  # (codes of x, snapshot of their scale state) when the forward ran on e4m3 operands
            save.append(u)
        if defer:
            return _Pending(y, st, save[-1] if save is not None else None)
        return a

    def forward(self, images, training: bool, want_ctx: bool):
        """images: NCHW fp32 batch, or a tuple of such batches (the views) that is processed as their concatenation."""
        f = self.features
        conv1, bn1 = f[0], f[1]
        views = tuple(images) if isinstance(images, (tuple, list)) else (images,)
        n = sum(v.shape[0] for v in views)
        h, w = views[0].shape[-2:]
        ctx: Optional[dict] = {"units": [], "blocks": []} if want_ctx else None
        self._repack_stale()
        # stem: direct 7x7/2 conv from the zero-padded NHWC4 copy of the batch (0.9 GB at 2048 x 224^2 -- an im2col
        # matrix would be 9.9 GB); the same copy feeds the stem's weight gradient
        xp = ops.stem_pad_input(tuple(v.contiguous() for v in views), self.dtype)
        pk = self._pack(conv1, need_t=False, stem=True)
        d = ops.conv_desc(n, h, w, 3, 64, 7, 7, 2, 3, self.dtype)  # bookkeeping only (n, h, w, ho, wo, cout)
        ywin = None
        y, part = ops.stem_conv_fwd(xp, pk.krsc, h, w, want_stats=training)
        ho, wo = y.shape[1], y.shape[2]
        m = n * ho * wo
        st = self._bn(bn1, part, m, 64, training)
        # BN + ReLU + MaxPool in one pass: the 112x112x64 activation in between is never stored
        if want_ctx:  # + the winning taps' raw conv outputs: the backward's statistics pass then runs over pooled-size tensors
            x, idx, ywin = ops.bn_relu_maxpool_fwd(y, st, want_winner=True)
        else:
            x, idx = ops.bn_relu_maxpool_fwd(y, st)
        if want_ctx:
            u = _Unit()
            u.conv, u.bn, u.desc, u.x, u.y, u.a, u.st, u.relu, u.stem = conv1, bn1, d, xp, y, None, st, True, True
            u.has_res = False
            u.mask = None
            u.x_in = u.s2 = u.t2 = u.ws2 = None
            ctx["stem"] = u
            ctx["pool_idx"] = idx
            ctx["pool_ywin"] = ywin
        self._chain = None
        self._gram = None
        for li in (4, 5, 6, 7):
            stage = list(f[li])
            for bi, blk in enumerate(stage):
                nxt = stage[bi + 1] if bi + 1 < len(stage) else None
                saved: Optional[list] = [] if want_ctx else None
                inp = x
                units = blk.units()
                t = inp
                last_conv = units[-1][0]
                for ui, (conv, bn) in enumerate(units[:-1]):
                    # the unit in front of a folded stride-1 conv3: its BN-apply rides on the Gram launch
                    gram_next = (ui == len(units) - 2 and last_conv.stride == (1, 1) and self._fold_fwd_ok(last_conv, True, inp))
                    fp8_next = units[ui + 1][0] if (self.fp8 and ui + 1 < len(units) - 1) else None
                    bnin_next = units[ui + 1][0]  # the conv that reads this unit's activation
                    t = self._conv_bn(conv, bn, t, True, None, training, saved, gram_next=gram_next, fp8_next=fp8_next, bnin_next=bnin_next)
                idn = inp
                dsaved: Optional[list] = [] if want_ctx else None
                if blk.downsample is not None:
                    idn = self._conv_bn(blk.downsample[0], blk.downsample[1], inp, False, None, training, dsaved)
                conv, bn = units[-1]
                # (the chained conv1 is a bf16 launch: fine in fp8 mode too, whose default fp8 set holds 3x3 layers only)
                fp8_all = self.fp8 and self.fp8_all
                chain_conv = nxt.units()[0][0] if (nxt is not None and nxt.downsample is None and not fp8_all) else None
                x = self._conv_bn(conv, bn, t, True, idn, training, saved, chain_conv=chain_conv)
                if want_ctx:
                    ctx["blocks"].append((saved, dsaved[0] if dsaved else None))
        enc = ops.avgpool_fwd(x)
        if want_ctx:
            ctx["last_shape"] = tuple(x.shape)
        out = enc if enc.dtype == torch.float32 else ops.cast(enc, torch.float32)
        return out, ctx

    # -- backward --------------------------------------------------------------
    @staticmethod
    def _foldable(u: Optional[_Unit]) -> bool:
        """1x1 / stride-1 conv followed by BN + residual + ReLU (the last unit of a Bottleneck)."""
        return (u is not None and u.has_res and u.mask is not None and u.conv.kernel_size == (1, 1) and u.conv.stride == (1, 1)
                and u.conv.in_channels % 64 == 0)

    def _small_gemm(self, a: Tensor, bt: Tensor) -> Tensor:
        """a [m][k] @ bt[n][k]^T in fp32 through the exact-f32 MFMA tile kernel (parameter-sized operands)."""
        m, k = a.shape
        n = bt.shape[0]
        d = ops.conv_desc(m, 1, 1, k, n, 1, 1, 1, 0, torch.float32)
        y, _ = ops.conv2d_fwd(d, a.contiguous().view(m, 1, 1, k), bt.contiguous(), want_stats=False)
        return y.view(m, n)

    def _fold_1x1_bn(self, u: _Unit, g, a_in, grads: dict, s=None):
        """Core of the folded BatchNorm backward of y = conv1x1(a_in; W), z = bn(y): a_in [n][ho][wo][cw] dense (already
        subsampled for a strided shortcut), g [n][ho][wo][cc] the gradient w.r.t. z.  Because y = a_in W^T pixel by
        pixel, everything BN backward needs is parameter-sized:  G = g^T a_in (the un-normalised weight gradient),
        s = sum g, S2 = a_in^T a_in, t2 = sum a_in;
            sum g*y = rowdot(G, W);   dgamma = invstd (sum g*y - mean s);   dbeta = s
            dy = A g - B y + C  with  A = gamma invstd,  B = invstd A dgamma / M,  C = -A dbeta / M + mean B
            dW  = diag(A) G - diag(B) W S2 + C t2^T
            d a_in = g (diag(A) W) - a_in (W^T diag(B) W) + C W
        so y and dy (cc wide) are neither read nor written.  Sets the three parameter gradients; returns the packed
        dgrad weights of the two terms, the bias C W, and s.
        Replaces (reference): autograd's native_batch_norm_backward + Conv2d backward of the 1x1 conv + BN pairs of
        torchvision's Bottleneck (conv3 / bn3, downsample) -- src/models/resnet_model.py:13-58."""
        n, ho, wo, cw = a_in.shape
        cc = g.shape[-1]
        m = n * ho * wo
        st = u.st
        f32 = torch.float32
        d1 = ops.conv_desc(n, ho, wo, cw, cc, 1, 1, 1, 0, self.dtype)
        wmaster = u.conv.weight.detach().view(cc, cw)
        rnd = self.dtype in _H16                                  # the algebra uses the weights the MFMAs saw
        if s is None and self.dtype in _H16:
            gmat, s = ops.conv2d_wgrad_colsum(d1, a_in, g)                  # [cc][cw] fp32, [cc]: sum g rides along
        else:
            gmat = ops.conv2d_wgrad(d1, a_in, g)
            if s is None:
                s = ops.colsum(g.view(m, cc), m, cc)
        if u.s2 is not None:                                                # Gram products saved by the folded forward
            s2, t2, ws2 = u.s2, u.t2, u.ws2
        else:
            dww = ops.conv_desc(n, ho, wo, cw, cw, 1, 1, 1, 0, self.dtype)
            s2 = ops.conv2d_wgrad(dww, a_in, a_in)                          # [cw][cw] fp32 (symmetric)
            t2 = ops.colsum(a_in.view(m, cw), m, cw)
            w2 = wmaster.to(self.dtype).to(f32) if rnd else wmaster.to(f32)
            ws2 = self._small_gemm(w2, s2)                                  # W S2   (S2 symmetric)
        # the parameter-sized algebra: two launches (per-channel coefficients / dW / operands, then W^T diag(B) W and C W)
        dgamma, dbeta, dw, wa, wm, bias = ops.bn_fold_bwd(wmaster, rnd, gmat, s, ws2, t2, st, u.bn.weight.detach(), m, self.dtype)
        grads[u.bn.weight] = dgamma
        grads[u.bn.bias] = dbeta
        grads[u.conv.weight] = dw.view(cc, cw, 1, 1)
        return wa, wm, bias, s

    def _unit3_bwd_folded(self, u: _Unit, g, grads: dict, prev: _Unit):
        """conv3 + bn3 of a Bottleneck (out = relu(bn(conv1x1(a2)) + identity)) without the two BatchNorm-backward
        passes: `g` is the incoming gradient already gated by the output ReLU mask (stored that way by the dgrad that
        produced it).  12.5 instead of 30 passes over a2-sized data.  Returns (da2, raw partial sums for `prev` or None, s)."""
        d = u.desc
        a2 = u.x
        wa, wm, bias, s = self._fold_1x1_bn(u, g, a2, grads)
        dww = ops.conv_desc(d.n, d.h, d.w, d.cin, d.cin, 1, 1, 1, 0, self.dtype)
        if self.concat_fold and ops.conv2d_dgrad_concat_ok(d, d.cin):
            # both terms in one accumulation (K = cout + cin): da2 is written once, and the tile kernel carries prev's sums
            if self.fuse_bn_bwd:
                da2, part = ops.conv2d_dgrad_ex(d, g, wa, bias=bias, x2=a2, wt2=wm, fuse_mode=2 if prev.relu else 0, prev_y=prev.y,
                                                prev_st=prev.st)
                return da2, part, s
            da2, _ = ops.conv2d_dgrad_ex(d, g, wa, bias=bias, x2=a2, wt2=wm)
            return da2, None, s
        da2, _ = ops.conv2d_dgrad_ex(d, g, wa, bias=bias)
        if self.fuse_bn_bwd and ops.conv2d_dgrad_fuse_pays(dww):
            da2, part = ops.conv2d_dgrad_ex(dww, a2, wm, dx=da2, accumulate=True, fuse_mode=2 if prev.relu else 0, prev_y=prev.y,
                                            prev_st=prev.st)
            return da2, part, s
        da2, _ = ops.conv2d_dgrad_ex(dww, a2, wm, dx=da2, accumulate=True)
        return da2, None, s

    def _ds_bwd_folded(self, u: _Unit, g, s, grads: dict, dx, below: Optional[_Unit], masked_store: bool, scatter: bool = True):
        """Shortcut conv1x1(/stride) + BN of a stage's first block, folded like conv3 + bn3: its incoming gradient is the
        same masked g (and the same s) as the block's bn3.  Both terms of the input gradient accumulate into `dx` (the
        main branch's input gradient); for a stride-2 shortcut they only touch the even pixels -- so does the bias."""
        d = u.desc
        x_in = u.x_in if u.x_in is not None else (u.x if d.stride == 1 else ops.subsample2(u.x))
        wa, wm, bias, _ = self._fold_1x1_bn(u, g, x_in, grads, s=s)
        kw = dict(fuse_mode=4, prev_mask=below.mask, want_sums=False) if masked_store else {}
        if d.stride == 1:
            dterm = ops.conv_desc(d.n, d.h, d.w, d.cin, d.cin, 1, 1, 1, 0, self.dtype)
            if self.concat_fold and ops.conv2d_dgrad_concat_ok(d, d.cin):
                ops.conv2d_dgrad_ex(d, g, wa, dx=dx, accumulate=True, bias=bias, x2=x_in, wt2=wm, **kw)
                return dx
            ops.conv2d_dgrad_ex(d, g, wa, dx=dx, accumulate=True, bias=bias, **kw)
            ops.conv2d_dgrad_ex(dterm, x_in, wm, dx=dx, accumulate=True, **kw)
            return dx
        # stride 2: both terms densely at the output resolution (the stride-1 kernels run near their bounds, the
        # parity-class dgrad does not), then one scatter-add onto the even pixels of dx, through the consumer's mask
        dd = ops.conv_desc(d.n, d.ho, d.wo, d.cin, d.cout, 1, 1, 1, 0, self.dtype)
        dterm = ops.conv_desc(d.n, d.ho, d.wo, d.cin, d.cin, 1, 1, 1, 0, self.dtype)
        if self.concat_fold and ops.conv2d_dgrad_concat_ok(dd, dd.cin):
            dsub, _ = ops.conv2d_dgrad_ex(dd, g, wa, bias=bias, x2=x_in, wt2=wm)
        else:
            dsub, _ = ops.conv2d_dgrad_ex(dd, g, wa, bias=bias)
            ops.conv2d_dgrad_ex(dterm, x_in, wm, dx=dsub, accumulate=True)
        if not scatter:  # the caller merges it into the main branch's data gradient (sub_grad)
            return dsub
        ops.scatter2_add(dsub, dx, below.mask if masked_store else None)
        return dx

    def _unit_bwd(self, u: _Unit, da, grads: dict, need_dx: bool, dx_into=None, relu_mask=None, res_grad=None, res_mask=None,
                  raw_partial=None, prev: Optional[_Unit] = None, prev_masked_store: bool = False, sub_grad=None):
        """BN bwd -> wgrad (+ dgrad).  relu_mask: bit mask that gates `da` (residual units: their own output mask;
        downsample branch: the block output's mask).  res_grad/res_mask: merge the identity-branch gradient
        res_grad * bit(res_mask) into dx inside the dgrad epilogue.  raw_partial: this unit's BN-backward sums
        already produced by the dgrad that wrote `da`.  prev: the unit whose incoming gradient is the dx computed
        here -- its BN-backward partial sums are then fused into this dgrad's epilogue.
        Returns (dx or None, raw partial sums for `prev` or None)."""
        d = u.desc
        m = d.n * d.ho * d.wo
        c = d.cout
        if u.y is None:
            raise RuntimeError("this unit ran the folded forward (its raw conv output was never stored): its backward must be folded too")
        w = u.conv.weight
        fuse_apply = (self.fuse_bwd_apply_wgrad and self.dtype in _H16 and not u.stem and relu_mask is None and not u.has_res
                      and u.conv.kernel_size == (1, 1) and u.conv.stride == (1, 1) and u.conv.padding == (0, 0))
        fuse_dg = (self.fuse_bwd_apply_dgrad and not fuse_apply and self.dtype in _H16 and not u.stem and relu_mask is None
                   and not u.has_res and need_dx and prev is not None and prev_masked_store and u.conv.kernel_size == (1, 1)
                   and u.conv.stride == (1, 1) and u.conv.padding == (0, 0) and ops.conv2d_dgrad_dysrc_ok(d))
        # fp8 configuration: the data gradient of the 3x3 layers with >= 256 channels runs on e4m3 operands; dy's codes leave the
        # BatchNorm-backward apply pass (delayed scaling), the CRSK weights are re-quantised per parameter version
        f8 = None
        if (self.fp8 and need_dx and not u.stem and not fuse_apply and not fuse_dg and relu_mask is None and res_grad is None and dx_into is None
                and not (prev is not None and prev_masked_store) and ops.conv2d_dgrad_fp8_pays(d)):
            f8 = self._fp8_site_bwd(u.conv, da.device)
        bw = ops.bn_backward(da, u.a, u.y, u.st, u.bn.weight.detach(), m, c, u.relu, False,
                             mask_from_y=u.relu and not u.has_res, relu_mask=relu_mask, raw_partial=raw_partial,
                             apply=not (fuse_apply or fuse_dg), fp8_scaler=f8[0] if f8 is not None else None,
                             want_coefs=fuse_apply or fuse_dg)
        dy, _, dg, db = bw[:4]
        dyq = bw[4] if (len(bw) > 4 and not (fuse_apply or fuse_dg)) else None
        coefs = bw[4] if (fuse_apply or fuse_dg) else None  # (A, B, C) of dy = A g - B y + C, from the launch that folded the sums
        grads[u.bn.weight] = dg
        grads[u.bn.bias] = db
        if fuse_dg:
            # data gradient FIRST: it derives dy from (da, y) while loading its operand rows and writes it out for the weight gradient
            pk = self._pack(u.conv, need_t=True)
            dy = torch.empty_like(u.y)
            dxm, _ = ops.conv2d_dgrad_ex(d, None, pk.crsk, dx=dx_into, accumulate=dx_into is not None, res_grad=res_grad, res_mask=res_mask,
                                         fuse_mode=4, prev_mask=prev.mask, want_sums=False,
                                         dy_src=(da.contiguous(), u.y, u.st, coefs, u.relu, dy), sub_grad=sub_grad)
            grads[w] = ops.conv2d_wgrad_oihw(d, u.x, dy, tuple(w.shape))
            return dxm, None
        if fuse_apply:
            grads[w], dy = ops.conv2d_wgrad_bnbwd(d, u.x, da, u.y, u.st, coefs, u.relu, tuple(w.shape))
        elif u.stem:
            grads[w] = ops.stem_conv_wgrad(u.x, dy, d.h, d.w)
        elif dyq is not None and self.fp8_wgrad and getattr(u, "xq", None) is not None and ops.conv2d_wgrad_fp8_pays(d):
            # fp8 configuration: both operands already exist as e4m3 codes (x from the forward's BatchNorm-apply, dy from the pass above)
            grads[w] = ops.conv2d_wgrad_fp8(d, u.xq[0], dyq, u.xq[1], f8[0].state)
        else:
            grads[w] = ops.conv2d_wgrad_oihw(d, u.x, dy, tuple(w.shape))  # split-K reduce writes weight.grad's layout
        if not need_dx:
            return None, None
        pk = self._pack(u.conv, need_t=True)
        if prev is not None and prev_masked_store:
            # store dx gated by prev's output ReLU mask + its channel sums: the form _unit3_bwd_folded consumes
            dxm, _ = ops.conv2d_dgrad_ex(d, dy, pk.crsk, dx=dx_into, accumulate=dx_into is not None, res_grad=res_grad, res_mask=res_mask,
                                         fuse_mode=4, prev_mask=prev.mask, want_sums=False, sub_grad=sub_grad)
            return dxm, None
        if sub_grad is not None:
            raise RuntimeError("sub_grad: only for the masked-store data gradient of a stage-entry conv1")
        if dyq is not None:  # e4m3 reduction (same epilogue options: the previous unit's BatchNorm-backward sums where they pay)
            fuse = prev is not None and self.fuse_bn_bwd and ops.conv2d_dgrad_fuse_pays(d) and not prev.has_res
            return ops.conv2d_dgrad_ex(d, dy, pk.crsk, fuse_mode=(2 if prev.relu else 0) if fuse else None, prev_y=prev.y if fuse else None,
                                       prev_st=prev.st if fuse else None, fp8=(dyq, f8[2], f8[0], f8[1]))
        if prev is not None and self.fuse_bn_bwd and ops.conv2d_dgrad_fuse_pays(d):
            # prev's incoming gradient = this dx; residual units gate it with their output mask, the others with
            # the mask recomputed from their own y
            return ops.conv2d_dgrad_fused(d, dy, pk.crsk, prev.y, prev.st if prev.relu else None, prev.mask if prev.has_res else None,
                                          dx=dx_into, accumulate=dx_into is not None, res_grad=res_grad, res_mask=res_mask)
        if res_grad is not None:
            return ops.conv2d_dgrad_masked_residual(d, dy, pk.crsk, res_grad, res_mask), None
        if dx_into is not None:
            return ops.conv2d_dgrad(d, dy, pk.crsk, dx=dx_into, accumulate=True), None
        return ops.conv2d_dgrad(d, dy, pk.crsk), None

    def backward(self, ctx: dict, d_enc: Tensor) -> Dict[nn.Parameter, Tensor]:
        grads: Dict[nn.Parameter, Tensor] = {}
        red = self.grad_reducer if (self.grad_reducer is not None and self.grad_reducer.active()) else None
        sent = 0  # gradients of `grads` (insertion order) already handed to the reducer
        g = d_enc.contiguous() if d_enc.dtype == self.dtype else ops.cast(d_enc.contiguous(), self.dtype)
        blocks = ctx["blocks"]
        top = blocks[-1][0][-1] if blocks else None
        top_fold = self.fold_bn3 and self._foldable(top)
        # the last block's bn3 folds as well: its incoming gradient leaves the pooling backward already gated by the output ReLU mask
        dz = ops.avgpool_bwd(g, ctx["last_shape"], mask=top.mask if top_fold else None)
        dz_part = None  # per-tile sums for the current block's last unit, when the next block's dgrad produced them
        dz_masked = top_fold  # dz is stored already gated by this block's output ReLU mask
        for bi in range(len(blocks) - 1, -1, -1):
            saved, ds = blocks[bi]
            last = saved[-1]
            # out = relu(bn(conv(t)) + idn): the gradient of both branches is dz gated by the output's ReLU mask
            s_g = None
            if dz_masked:
                dt_, part, s_g = self._unit3_bwd_folded(last, dz, grads, prev=saved[-2])
            else:
                dt_, part = self._unit_bwd(last, dz, grads, True, relu_mask=last.mask, raw_partial=dz_part, prev=saved[-2])
            for ui in range(len(saved) - 2, 0, -1):
                dt_, part = self._unit_bwd(saved[ui], dt_, grads, True, raw_partial=part, prev=saved[ui - 1])
            first = saved[0]
            if ds is not None:
                # main branch first (plain store), then the shortcut accumulates: for a stride-2 1x1 shortcut the
                # dgrad kernel then only visits the one parity class its taps can reach (1/4 of dx).  When the block
                # below folds its bn3 backward every kernel stores through its output mask (mask(mask(a) + b) = mask(a + b)).
                below = blocks[bi - 1][0][-1] if bi > 0 else None
                fold = self.fold_bn3 and self._foldable(below)
                ds_folds = dz_masked and self.fold_bn3 and ds.conv.in_channels % 64 == 0
                # stride-2 shortcut whose consumer stores through a mask: its dense gradient first, merged by conv1's data gradient
                merge = (ds_folds and self.merge_shortcut and fold and ds.desc.stride == 2 and first.desc.h % 2 == 0 and first.desc.w % 2 == 0
                         and first.conv.kernel_size == (1, 1) and first.conv.stride == (1, 1) and self.dtype in _H16)
                sub = self._ds_bwd_folded(ds, dz, s_g, grads, None, below, fold, scatter=False) if merge else None
                dx, _ = self._unit_bwd(first, dt_, grads, True, raw_partial=part, prev=below if fold else None, prev_masked_store=fold,
                                       sub_grad=sub)
                if merge:
                    dz = dx
                elif ds_folds:
                    dz = self._ds_bwd_folded(ds, dz, s_g, grads, dx, below, fold)
                else:
                    dz, _ = self._unit_bwd(ds, dz, grads, True, relu_mask=last.mask, dx_into=dx, prev=below if fold else None,
                                           prev_masked_store=fold)
                dz_part, dz_masked = None, fold
            else:
                # identity block: dz of the block below = conv1's dgrad + masked dz; that block's last unit is `below`
                below = blocks[bi - 1][0][-1] if bi > 0 else None
                fold = self.fold_bn3 and self._foldable(below)
                dz, dz_part = self._unit_bwd(first, dt_, grads, True, res_grad=dz, res_mask=last.mask, raw_partial=part, prev=below,
                                             prev_masked_store=fold)
                dz_masked = fold
            saved.clear()
            if red is not None:  # this block's parameter gradients are final: their all-reduce overlaps the blocks below
                items = list(grads.items())
                red.submit(items[sent:])
                sent = len(items)
        # stem: the pooled gradient is gathered through the winner index inside the BatchNorm-backward passes
        u = ctx["stem"]
        dy, dg, db = ops.maxpool_bn_backward(dz, ctx["pool_idx"], u.y, u.st, u.bn.weight.detach(), ywin=ctx.get("pool_ywin"))
        grads[u.conv.weight] = ops.stem_conv_wgrad(u.x, dy, u.desc.h, u.desc.w)
        grads[u.bn.weight] = dg
        grads[u.bn.bias] = db
        if red is not None:
            red.submit(list(grads.items())[sent:])
            grads.update(red.finish())  # waits for the buckets in flight; the reduced buckets' views ARE the gradients autograd receives
        return grads


class _EncoderFn(torch.autograd.Function):
    """One autograd node for the whole backbone; params are passed so autograd
    routes their gradients, the engine does the work."""

    @staticmethod
    def forward(ctx, images, engine, training, grad_mode, *params):
        need = grad_mode and any(p.requires_grad for p in params)
        if need and not training:
            raise NotImplementedError("backward through eval-mode BatchNorm is not part of the training step")
        enc, ectx = engine.forward(images, training, need)
        ctx.engine, ctx.ectx, ctx.params = engine, ectx, params
        return enc

    @staticmethod
    def backward(ctx, d_enc):
        grads = ctx.engine.backward(ctx.ectx, d_enc)
        ctx.ectx = None
        return (None, None, None, None) + tuple(grads.get(p) if p.requires_grad else None for p in ctx.params)


class ResNetModel(nn.Module):
    """Reference-compatible wrapper (src/models/resnet_model.py:6-58).

    ``config.model.backend_model`` in {"resnet18",...,"resnet152"};
    ``config.model.pretrained`` is accepted but ImageNet weights are not
    fetched (no network): seeded random init, as in the parity oracle.
    The reference's per-forward ``print(self.features)`` (:49) is dropped."""

    def __init__(self, config, mode: str = "", compute_dtype: torch.dtype = torch.float32):
        super().__init__()
        self.mode = mode
        name = config.model.backend_model.lower()
        if not name.startswith("resnet") or name[6:] not in _SPECS:
            raise NotImplementedError(name)
        m = _Backbone(name[6:])
        self.features = nn.Sequential(m.conv1, m.bn1, m.relu, m.maxpool, m.layer1, m.layer2, m.layer3, m.layer4,
                                      nn.AdaptiveAvgPool2d((1, 1)))
        self.final_layer = nn.Sequential(nn.Linear(m.fc.in_features, 21 * 3 + 1))
        self.out_features = m.fc.in_features
        self.engine = ResNetEngine(self.features, compute_dtype)

    def set_compute_dtype(self, dtype: torch.dtype, fp8: bool = False) -> None:
        """torch.float32 (exact-fp32 parity mode), torch.bfloat16, or torch.float16 (the reference's precision=16 storage type: selects
        the fp16 build of the library; the caller scales the loss, host/amp.py)."""
        if dtype in _H16:
            _lib.use_half("f16" if dtype == torch.float16 else "bf16")
        self.engine = ResNetEngine(self.features, dtype, fp8=fp8)

    def forward(self, x) -> Tensor:
        """x: image batch, or a tuple of batches (the two views) encoded as their concatenation -- the stem reads each
        view in place, so the 1.2 GB torch.cat of the reference's training_step is not materialised."""
        if self.mode != "pretraining":
            raise NotImplementedError("only mode='pretraining' (the contrastive hot path) is built; "
                                      "the supervised 2.5D head is out of scope (SURVEY 2 #5)")
        params = [p for p in self.features.parameters()]
        return _EncoderFn.apply(x, self.engine, self.training, torch.is_grad_enabled(), *params)
