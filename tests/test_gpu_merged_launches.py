"""GPU: the merged parameter-sized launches of round 6 (VERDICT r5 "Next" 1a) against the launch chains they replace, BIT FOR BIT.

Every merge keeps each output element's additions in the order the separate launches used (the split-K reduction's slice tree, the
fp64 fold of the channel sums, the centring of the Gram matrix, the slice sum of the small products, the BatchNorm-backward
coefficients with pinned roundings), so `torch.equal` is the criterion -- SH_SW_FOLD_LEGACY selects the round-5 chains.  Route counters
and the launch count of a whole ResNet-50 step are asserted next to the numbers."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
DT = torch.bfloat16


def _both(fn):
    """fn() under the merged launches (default) and under the round-5 chains."""
    from simhand_amd import ops

    ops.hooks_reset()
    new = fn()
    ops.test_switch("FOLD_LEGACY", 1)
    try:
        old = fn()
    finally:
        ops.hooks_reset()
    torch.cuda.synchronize()
    return new, old


def _eq(a, b, tag):
    if isinstance(a, (tuple, list)):
        assert len(a) == len(b), tag
        for i, (x, y) in enumerate(zip(a, b)):
            _eq(x, y, f"{tag}[{i}]")
        return
    if a is None:
        assert b is None, tag
        return
    assert a.shape == b.shape and a.dtype == b.dtype, tag
    assert torch.equal(a, b), f"{tag}: {int((a != b).sum())} of {a.numel()} elements differ, max |d| {float((a.float() - b.float()).abs().max()):.3e}"


@pytest.mark.parametrize("n,h,cin,cout", [(8, 28, 64, 256), (4, 28, 128, 512), (8, 14, 256, 1024), (16, 7, 512, 2048), (6, 14, 256, 512),
                                          (3, 9, 1024, 2048), (40, 56, 64, 256)])
def test_weight_gradient_with_channel_sums_folded_in_the_reduction_launch(n, h, cin, cout):
    """G = g^T a + sum g (the folded BatchNorm backward's Gram launch): every split-K variant of the reduction (1 / 4 / 16 slices), the
    pointer-walking and the LDS-DMA weight-gradient kernels."""
    from simhand_amd import ops

    g = torch.Generator().manual_seed(n * h + cin)
    a = torch.randn(n, h, h, cin, generator=g).relu().to(DT).to(DEV)
    dy = torch.randn(n, h, h, cout, generator=g).to(DT).to(DEV)
    d = ops.conv_desc(n, h, h, cin, cout, 1, 1, 1, 0, DT)
    new, old = _both(lambda: ops.conv2d_wgrad_colsum(d, a, dy))
    _eq(new, old, "wgrad_colsum")
    m = n * h * h
    want = dy.float().view(m, cout).sum(0)
    assert (new[1] - want).abs().max() <= 1e-3 * want.abs().max() + 1e-3


@pytest.mark.parametrize("n,h,c", [(8, 28, 64), (40, 56, 64), (4, 28, 128), (8, 14, 256), (16, 7, 512)])
def test_gram_launch_with_channel_sums_folded_in_the_reduction_launch(n, h, c):
    from simhand_amd import ops

    g = torch.Generator().manual_seed(c + n)
    y = torch.randn(n, h, h, c, generator=g).to(DT).to(DEV)
    st = ops.BNState(c, DEV)
    st.scale.copy_(torch.rand(c, generator=g) + 0.5)
    st.shift.copy_(torch.randn(c, generator=g) * 0.3)
    new, old = _both(lambda: ops.bn_apply_gram(y, st, True))
    _eq(new, old, "bn_apply_gram")


@pytest.mark.parametrize("cc,cw", [(256, 64), (512, 128), (1024, 256), (2048, 512), (512, 256), (2048, 1024)])
def test_folded_batchnorm_algebra_merged_launches_equal_the_chains(cc, cw):
    """simhand_bn_fold_fwd (centre-on-load + slice sum in the per-channel kernel: 2 launches for 4) and simhand_bn_fold_bwd (slice sum + bias
    in one launch: 3 for 4) on conv3 / shortcut shapes of every stage."""
    from simhand_amd import ops

    g = torch.Generator().manual_seed(cc + cw)
    m = 4096
    a = (torch.randn(m, cw, generator=g).relu() + 0.1).to(DT).float()
    s2 = (a.t() @ a).to(DEV)
    t2 = a.sum(0).to(DEV)
    w = (torch.randn(cc, cw, generator=g) / math.sqrt(cw)).to(DEV)
    gamma = (torch.rand(cc, generator=g) + 0.5).to(DEV)
    beta = (torch.randn(cc, generator=g) * 0.1).to(DEV)

    def fwd():
        rm, rv, nbt = torch.zeros(cc, device=DEV), torch.ones(cc, device=DEV), torch.zeros(1, dtype=torch.int64, device=DEV)
        st, ws2 = ops.bn_fold_fwd(w, True, s2, t2, m, gamma, beta, rm, rv, nbt)
        return (st.mean.clone(), st.invstd.clone(), st.scale.clone(), st.shift.clone(), ws2, rm, rv, nbt), st

    (new, st), (old, _) = _both(fwd)
    _eq(new, old, "bn_fold_fwd")
    gm = torch.randn(cc, cw, generator=g).to(DEV)
    sg = torch.randn(cc, generator=g).to(DEV)
    newb, oldb = _both(lambda: ops.bn_fold_bwd(w, True, gm, sg, new[4], t2, st, gamma, m, DT))
    _eq(newb, oldb, "bn_fold_bwd")


@pytest.mark.parametrize("nblk,c", [(98, 64), (1568, 256), (6272, 128)])
def test_batchnorm_backward_coefficients_from_the_finalize_launch(nblk, c):
    from simhand_amd import ops

    g = torch.Generator().manual_seed(nblk + c)
    part = torch.randn(nblk, 2, c, generator=g).to(DEV)
    st = ops.BNState(c, DEV)
    st.mean.copy_(torch.randn(c, generator=g))
    st.invstd.copy_(torch.rand(c, generator=g) + 0.5)
    gamma = (torch.rand(c, generator=g) + 0.5).to(DEV)
    m = 12345
    y = torch.zeros(2, 2, c, dtype=DT, device=DEV)  # (shapes only: apply=False never reads them)

    def run():
        out = ops.bn_backward(y, None, y, st, gamma, m, c, True, False, mask_from_y=True, raw_partial=part, apply=False, want_coefs=True)
        return out[2], out[3], out[4]

    new, old = _both(run)
    _eq(new, old, "finalize + coefficients")


def test_rn50_step_is_bit_identical_with_fewer_launches():
    """A whole ResNet-50 HandCLR_W bf16 step: loss and EVERY parameter gradient equal bit for bit under both launch structures (the merged one
    issues ~120 kernel launches fewer per step -- 40 channel-sum folds, 20 centrings, ~36 slice sums, 12 coefficient launches, 20 bias launches:
    profiles/r06_step_timeline.txt; launches are counted there, by rocprofv3, not here)."""
    from oracle import step as orc
    from simhand_amd import ops
    from tests.test_gpu_configs import _oracle, _product

    wcfg = dict(weight_type="linear", diff_type="mpjpe", pos_neg="pos_neg")
    b, img = 8, 224
    om = _oracle("simhand_w", "50", wcfg, 41, 0.1)
    batch = {k: v.to(DEV) for k, v in orc.synthetic_batch(b, size=img, seed=41).items()}

    def run():
        model = _product("HandCLR_W", "50", wcfg, om, torch.bfloat16, b)
        ops.route_reset()
        out = model.training_step(batch, 0)
        out["loss"].backward()
        rc = ops.route_counts()
        assert rc["bn_fold_fwd"] == 20 and rc["bn_fold_bwd"] == 20 and rc["wgrad_colsum"] >= 36 and rc["dgrad_dysrc"] >= 12, rc
        return out["loss"].detach().clone(), [p.grad.detach().clone() for p in model.parameters() if p.grad is not None]

    new, old = _both(run)
    _eq(new, old, "step")


@pytest.mark.parametrize("nblk,c", [(3, 64), (392, 2048), (1568, 256), (25088, 64), (6272, 128), (257, 96)])
def test_batchnorm_finalize_in_one_launch_with_a_last_block_ticket(nblk, c):
    """simhand_bn_finalize_ticket: level-1 fold + finalize in one launch (the block that draws the last ticket finalizes) against the two
    launches, bit for bit -- statistics, running buffers, the batch counter; called repeatedly (the ticket must come back to zero)."""
    from simhand_amd import ops

    g = torch.Generator().manual_seed(nblk + c)
    m = nblk * 128
    part = torch.empty(nblk, 2, c)
    part[:, 0] = torch.randn(nblk, c, generator=g) * 128
    part[:, 1] = (torch.randn(nblk, c, generator=g) ** 2) * 128 + 64
    part = part.to(DEV)
    gamma, beta = (torch.rand(c, generator=g) + 0.5).to(DEV), torch.randn(c, generator=g).to(DEV)

    def run():
        rm, rv, nbt = torch.zeros(c, device=DEV), torch.ones(c, device=DEV), torch.zeros(1, dtype=torch.int64, device=DEV)
        outs = []
        for _ in range(3):
            st = ops.bn_finalize(part, m, c, gamma, beta, rm, rv, nbt)
            outs += [st.mean.clone(), st.invstd.clone(), st.scale.clone(), st.shift.clone()]
        return outs + [rm, rv, nbt]

    new, old = _both(run)
    _eq(new, old, "bn_finalize")
    assert int(new[-1]) == 3
    torch.cuda.synchronize()
    assert all(int(t.abs().sum()) == 0 for t in ops._TICKETS.values()), "a ticket word was left non-zero"
