"""Test helper: make gradient parity immune to ReLU-kink flips.

An activation within an fp32 ulp of zero can land on either side of the ReLU kink under a different
(equally valid) summation order, which perturbs every upstream gradient by ~1e-3..1e-2.  To compare
gradients to round-off anyway, the HIP forward's ReLU masks are recorded and imposed on the oracle:
its ReLU modules return ``input * mask`` (same value wherever both sides agree, identical routing of
the gradient everywhere)."""
import contextlib

import torch


@contextlib.contextmanager
def record_hip_relu_masks(store: list):
    from simhand_amd import ops

    orig = ops.bn_apply

    def bn_apply(y, st, m, c, relu, residual=None, out=None, **kw):
        res = orig(y, st, m, c, relu, residual, out, **kw)
        a = res[0] if isinstance(res, tuple) else res
        if relu:
            store.append((a > 0).cpu())
        return res

    orig_stem = ops.bn_relu_maxpool_fwd

    def bn_relu_maxpool_fwd(y, st, want_winner=False):
        # the fused stem kernel never stores the activation: take its ReLU mask from the unfused kernel (same fp32 expression)
        n, h, w, c = y.shape
        store.append((orig(y.view(n * h * w, c), st, n * h * w, c, True, None) > 0).view(n, h, w, c).cpu())
        return orig_stem(y, st, want_winner)

    orig_fused = ops.conv2d_fwd_bnact

    def conv2d_fwd_bnact(d, x, w, st, relu, residual=None, want_mask=False):
        res = orig_fused(d, x, w, st, relu, residual, want_mask)
        if relu:
            store.append(((res[0] if want_mask else res) > 0).cpu())
        return res

    orig_chain = ops.conv2d_fwd_bnact_chain

    def conv2d_fwd_bnact_chain(d, x, w, st, residual, chain_w):
        res = orig_chain(d, x, w, st, residual, chain_w)  # (out, mask, chained raw conv output, its partial sums): ReLU is implied
        store.append((res[0] > 0).cpu())
        return res

    orig_gram = ops.bn_apply_gram

    def bn_apply_gram(y, st, relu=True):
        res = orig_gram(y, st, relu)
        if relu:
            store.append((res[0] > 0).cpu())
        return res

    orig_bnin = ops.conv2d_fwd_bnin

    def conv2d_fwd_bnin(d, y_in, st_in, w, want_stats=True):
        res = orig_bnin(d, y_in, st_in, w, want_stats)  # (activation of the unit in front, conv output, partial sums): ReLU is implied
        store.append((res[0] > 0).cpu())
        return res

    ops.conv2d_fwd_bnin = conv2d_fwd_bnin
    ops.bn_apply_gram = bn_apply_gram
    ops.bn_apply = bn_apply
    ops.bn_relu_maxpool_fwd = bn_relu_maxpool_fwd
    ops.conv2d_fwd_bnact = conv2d_fwd_bnact
    ops.conv2d_fwd_bnact_chain = conv2d_fwd_bnact_chain
    try:
        yield store
    finally:
        ops.bn_apply = orig
        ops.bn_relu_maxpool_fwd = orig_stem
        ops.conv2d_fwd_bnact = orig_fused
        ops.conv2d_fwd_bnact_chain = orig_chain
        ops.bn_apply_gram = orig_gram
        ops.conv2d_fwd_bnin = orig_bnin


@contextlib.contextmanager
def impose_relu_masks(oracle_model: torch.nn.Module, masks: list):
    """masks: NHWC (or (N,C)) boolean tensors in forward call order."""
    it = iter(masks)
    flips = []

    def hook(mod, inp, out):
        m = next(it)
        x = inp[0]
        if x.dim() == 4:
            m = m.reshape(x.shape[0], x.shape[2], x.shape[3], x.shape[1]).permute(0, 3, 1, 2)
        else:
            m = m.reshape(x.shape)
        flips.append(int(((x > 0) != m).sum()))
        return x * m.to(x.dtype)

    handles = [mod.register_forward_hook(hook) for mod in oracle_model.modules() if isinstance(mod, torch.nn.ReLU)]
    try:
        yield flips
    finally:
        for h in handles:
            h.remove()
