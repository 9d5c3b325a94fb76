"""GPU parity: the fused weighted NT-Xent path (distances -> weights -> loss -> closed-form
backward) and the projection post-process, through the C ABI, against
  (1) the golden vectors produced by the reference's own Python (tests/golden), and
  (2) the CPU oracle on fresh seeded inputs, incl. ragged / sharded / large-N cases.
Tolerances (fp32): loss 1e-5 relative, gradients 1e-4 relative of the max |grad| (+1e-7 abs).
"""
import os

import numpy as np
import pytest
import torch

from oracle import step as orc

pytestmark = pytest.mark.gpu

DIFFS = ("mpjpe", "w_abs", "w_o_abs")
DEV = "cuda"


def _hip_loss(z1, z2, j1, j2, diff, weight_type, pos_neg, ranks=1, lam=(5.0, 0.05), backward=True, fused=False):
    """Run the HIP loss for all `ranks` row shards on one GPU; returns (loss, dz (N,128)).  fused: the distance tiles are
    computed inside the loss kernels (simhand_ntxent_*_fused), no D row block."""
    from simhand_amd import ops

    B = z1.shape[0]
    N = 2 * B
    Z = torch.cat((z1, z2)).to(DEV).contiguous()
    weighted = weight_type is not None
    J = None
    if weighted:
        J = torch.cat((j1, j2)).reshape(N, -1).to(DEV).contiguous()
    use_wpos = weighted and pos_neg in ("pos_neg", "pos")
    use_wneg = weighted and pos_neg in ("pos_neg", "neg")
    assert B % ranks == 0
    b_loc = B // ranks
    mode = "l2" if (weighted and j1.dim() == 2) else diff
    stats_r, D_r, plans = [], [], []
    d_pos = None
    for r in range(ranks):
        stats = torch.zeros(8, dtype=torch.float64, device=DEV)
        D = None
        if weighted:
            d_pos = ops.pos_dist(J, B, mode, stats)
            D = ops.neg_dist(J, B, mode, b_loc, r * b_loc, stats, stats_only=fused and use_wneg)
        stats_r.append(stats)
        D_r.append(D)
        plans.append(ops.NtxentPlan(B, b_loc, r * b_loc, weight_type, use_wpos, use_wneg, 0.5, lam[0], lam[1]))
    # the "all-reduce": max / min / sum of the row-block stats
    g = torch.stack(stats_r)
    glob = g[0].clone()
    glob[0], glob[1], glob[2] = g[:, 0].max(), g[:, 1].min(), g[:, 2].sum()
    if torch.isnan(g[:, :2]).any():
        glob[0] = glob[1] = float("nan")
    neg_all = torch.empty(N, dtype=torch.float32, device=DEV)
    loss = torch.zeros(1, dtype=torch.float32, device=DEV)
    for r in range(ranks):
        if fused and use_wneg:
            neg, lp = ops.ntxent_fwd_fused(plans[r], Z, J, mode, d_pos, glob)
        else:
            neg, lp = ops.ntxent_fwd(plans[r], Z, D_r[r], d_pos, glob)
        neg_all[r * b_loc:(r + 1) * b_loc] = neg[:b_loc]
        neg_all[B + r * b_loc:B + (r + 1) * b_loc] = neg[b_loc:]
        loss += lp
    dz = None
    if backward:
        dz = torch.empty(N, 128, dtype=torch.float32, device=DEV)
        for r in range(ranks):
            if fused and use_wneg:
                d = ops.ntxent_bwd_fused(plans[r], Z, J, mode, d_pos, glob, neg_all, None)
            else:
                d = ops.ntxent_bwd(plans[r], Z, D_r[r], d_pos, glob, neg_all, None)
            dz[r * b_loc:(r + 1) * b_loc] = d[:b_loc]
            dz[B + r * b_loc:B + (r + 1) * b_loc] = d[b_loc:]
        dz = dz.cpu()
    return loss.item(), dz


def _close_grad(got, want, tag):
    scale = np.abs(want).max()
    err = np.abs(got - want).max()
    assert err <= 1e-4 * scale + 1e-7, f"{tag}: grad err {err:.3e} vs scale {scale:.3e}"


@pytest.mark.parametrize("B", [2, 8, 32])
def test_loss_against_reference_golden(golden_dir, B):
    g = np.load(os.path.join(golden_dir, f"loss_B{B}.npz"))
    z1, z2 = torch.from_numpy(g["z1"]), torch.from_numpy(g["z2"])
    j1, j2 = torch.from_numpy(g["j1"]), torch.from_numpy(g["j2"])
    loss, dz = _hip_loss(z1, z2, j1, j2, "mpjpe", None, "pos_neg")
    assert abs(loss - float(g["loss.simclr"])) <= 1e-5 * abs(float(g["loss.simclr"]))
    _close_grad(dz[:B].numpy(), g["dz1.simclr"], "simclr dz1")
    _close_grad(dz[B:].numpy(), g["dz2.simclr"], "simclr dz2")
    for diff in DIFFS:
        for wt in ("linear", "non_linear"):
            for mode in ("pos_neg", "pos", "neg"):
                tag = f"{diff}.{wt}.{mode}"
                want = float(g[f"loss.{tag}"])
                loss, dz = _hip_loss(z1, z2, j1, j2, diff, wt, mode)
                if not np.isfinite(want):  # reference yields NaN/inf (flat distances): replicate, App. D #11
                    assert not np.isfinite(loss), tag
                    continue
                assert abs(loss - want) <= 1e-5 * abs(want), f"{tag}: {loss} vs {want}"
                _close_grad(dz[:B].numpy(), g[f"dz1.{tag}"], tag + " dz1")
                _close_grad(dz[B:].numpy(), g[f"dz2.{tag}"], tag + " dz2")


def test_pca_feature_weights_against_golden(golden_dir):
    """*_with_pca variants: plain L2 over 14 features (the PCA itself is host-side)."""
    g = np.load(os.path.join(golden_dir, "weights_pca.npz"))
    f1, f2 = torch.from_numpy(g["f1"]), torch.from_numpy(g["f2"])
    gen = torch.Generator().manual_seed(3)
    z1 = torch.nn.functional.normalize(torch.randn(8, 128, generator=gen))
    z2 = torch.nn.functional.normalize(torch.randn(8, 128, generator=gen))
    for wt in ("linear", "non_linear"):
        wp, wn = torch.from_numpy(g[f"wpos.mpjpe.{wt}"]), torch.from_numpy(g[f"wneg.mpjpe.{wt}"])
        want = orc.ntxent(z1, z2, wp, wn).item()
        got, _ = _hip_loss(z1, z2, f1, f2, "mpjpe", wt, "pos_neg", backward=False)
        assert abs(got - want) <= 1e-5 * abs(want), (wt, got, want)


@pytest.mark.parametrize("B,ranks", [(3, 1), (5, 1), (48, 2), (96, 4), (200, 8), (100, 1)])
@pytest.mark.parametrize("wt,diff,mode", [("linear", "mpjpe", "pos_neg"), ("non_linear", "w_abs", "neg"),
                                          ("linear", "w_o_abs", "pos"), (None, "mpjpe", "pos_neg")])
def test_loss_against_oracle_sharded(B, ranks, wt, diff, mode):
    """Row-block sharding: R shards + max/min/sum + gathered neg == single-process oracle."""
    gen = torch.Generator().manual_seed(1000 + B)
    z1 = torch.nn.functional.normalize(torch.randn(B, 128, generator=gen))
    z2 = torch.nn.functional.normalize(torch.randn(B, 128, generator=gen))
    j1 = torch.rand(B, 21, 2, generator=gen) * 224
    j2 = j1 + torch.randn(B, 21, 2, generator=gen) * 8
    wp = wn = None
    if wt == "linear":
        wp, wn = orc.weights_linear(j1, j2, diff)
    elif wt == "non_linear":
        wp, wn = orc.weights_nonlinear(j1, j2, 2.5, 0.01, diff)
    if mode == "pos":
        wn = None
    if mode == "neg":
        wp = None
    z = torch.cat((z1, z2))
    want, dz_want, _ = orc.ntxent_closed_form(z.double(), None if wp is None else wp.double(), None if wn is None else wn.double())
    got, dz = _hip_loss(z1, z2, j1, j2, diff, wt, mode, ranks=ranks, lam=(2.5, 0.01))
    assert abs(got - want.item()) <= 1e-5 * abs(want.item()), (got, want.item())
    _close_grad(dz.numpy(), dz_want.float().numpy(), f"B={B} R={ranks}")


def test_loss_full_size_properties():
    """BASELINE config-2 size (B=1024, N=2048): no oracle needed -- size-independent properties:
    sharded == unsharded, and a central finite difference along the gradient direction
    reproduces ||dL/dz||."""
    B = 1024
    gen = torch.Generator().manual_seed(7)
    z1 = torch.nn.functional.normalize(torch.randn(B, 128, generator=gen))
    z2 = torch.nn.functional.normalize(torch.randn(B, 128, generator=gen))
    j1 = torch.rand(B, 21, 2, generator=gen) * 224
    j2 = j1 + torch.randn(B, 21, 2, generator=gen) * 8
    l1, dz1 = _hip_loss(z1, z2, j1, j2, "mpjpe", "linear", "pos_neg", ranks=1)
    l8, dz8 = _hip_loss(z1, z2, j1, j2, "mpjpe", "linear", "pos_neg", ranks=8)
    assert abs(l1 - l8) <= 2e-6 * abs(l1)
    assert (dz1 - dz8).abs().max() <= 1e-5 * dz1.abs().max()
    # central finite difference along the (unit) gradient direction: dL ~= ||dz|| * eps
    v = dz1 / dz1.norm()
    eps = 0.25
    lp, _ = _hip_loss(z1 + eps * v[:B], z2 + eps * v[B:], j1, j2, "mpjpe", "linear", "pos_neg", backward=False)
    lm, _ = _hip_loss(z1 - eps * v[:B], z2 - eps * v[B:], j1, j2, "mpjpe", "linear", "pos_neg", backward=False)
    fd = (lp - lm) / (2 * eps)
    assert abs(fd - dz1.norm().item()) <= 2e-2 * dz1.norm().item(), (fd, dz1.norm().item())


@pytest.mark.parametrize("name", ["none", "crop", "rotate", "crop_rotate"])
def test_postprocess_against_reference_golden(golden_dir, name):
    from simhand_amd import ops

    g = np.load(os.path.join(golden_dir, "postprocess.npz"))
    P = torch.from_numpy(g["head_out"]).to(DEV)
    up = torch.from_numpy(g["upstream"]).to(DEV)
    hw = tuple(int(v) for v in g["image_hw"])
    jx = jy = ang = None
    if "crop" in name:
        jx = torch.from_numpy(np.concatenate((g["jitter_x_1"], g["jitter_x_2"]))).to(DEV)
        jy = torch.from_numpy(np.concatenate((g["jitter_y_1"], g["jitter_y_2"]))).to(DEV)
    if "rotate" in name:
        ang = torch.from_numpy(np.concatenate((g["angle_1"], g["angle_2"]))).to(DEV)
    Z = ops.proj_postprocess_fwd(P, jx, jy, ang, hw)
    dP = ops.proj_postprocess_bwd(P, jx, jy, ang, hw, up)
    np.testing.assert_allclose(Z.cpu().numpy(), g[f"z.{name}"], rtol=1e-5, atol=1e-6)
    _close_grad(dP.cpu().numpy(), g[f"dhead.{name}"], name)


def test_projection_stats_against_reference_golden(golden_dir):
    from simhand_amd import ops

    g = np.load(os.path.join(golden_dir, "postprocess.npz"))
    P = torch.from_numpy(g["head_out"]).to(DEV)
    b = P.shape[0] // 2
    order = ["x_mean", "x_median", "x_min", "x_max", "y_mean", "y_median", "y_min", "y_max"]
    for view, name in ((P[:b].contiguous(), "proj1"), (P[b:].contiguous(), "proj2")):
        out = ops.proj_stats(view).cpu().numpy()
        for i, k in enumerate(order):
            want = float(g[f"stat.{name}{k}"])
            assert abs(out[i] - want) <= 1e-6 + 1e-5 * abs(want), (name, k, out[i], want)


@pytest.mark.parametrize("B,ranks", [(5, 1), (96, 4), (200, 8), (1024, 2)])
@pytest.mark.parametrize("wt,diff,mode", [("linear", "mpjpe", "pos_neg"), ("non_linear", "w_abs", "neg"), ("linear", "w_o_abs", "pos_neg")])
def test_fused_distance_loss_is_bit_identical_to_the_row_block_form(B, ranks, wt, diff, mode):
    """North-star form: similarity tile + joint-distance tile + weighted softmax-cross-entropy in ONE LDS-tiled kernel, no
    [rows][N] distance block in HBM.  The in-tile distances follow simhand_neg_dist's operation order, so loss and gradient
    equal the row-block path bit for bit (which the golden / oracle tests above pin to the reference)."""
    from simhand_amd import ops

    gen = torch.Generator().manual_seed(77 + B)
    z1 = torch.nn.functional.normalize(torch.randn(B, 128, generator=gen))
    z2 = torch.nn.functional.normalize(torch.randn(B, 128, generator=gen))
    j1 = torch.rand(B, 21, 2, generator=gen) * 224
    j2 = j1 + torch.randn(B, 21, 2, generator=gen) * 8
    la, dza = _hip_loss(z1, z2, j1, j2, diff, wt, mode, ranks=ranks, lam=(2.5, 0.01))
    ops.route_reset()
    lb, dzb = _hip_loss(z1, z2, j1, j2, diff, wt, mode, ranks=ranks, lam=(2.5, 0.01), fused=True)
    assert ops.route_counts()["ntxent_fused_dist"] == 2 * ranks
    assert la == lb and torch.equal(dza, dzb)
    # PCA features (plain L2 over 14 columns)
    f1, f2 = torch.randn(B, 14, generator=gen), torch.randn(B, 14, generator=gen)
    lc, dzc = _hip_loss(z1, z2, f1, f2, "mpjpe", wt, mode, ranks=ranks, lam=(2.5, 0.01))
    ld, dzd = _hip_loss(z1, z2, f1, f2, "mpjpe", wt, mode, ranks=ranks, lam=(2.5, 0.01), fused=True)
    assert lc == ld and torch.equal(dzc, dzd)


def test_fused_distance_through_the_autograd_surface_and_timing():
    """LossConfig.fuse_dist routes ShardedNtxent through the fused kernels; prints both forms' time at the BASELINE sizes
    (N = 2048 single GPU; 2048 local rows x N = 16 384 = one rank of the 8-GPU config)."""
    import time

    from simhand_amd.host import dist_loss

    gen = torch.Generator().manual_seed(5)
    for B, b_loc in ((1024, 1024), (8192, 1024)):
        z = torch.nn.functional.normalize(torch.randn(2 * B, 128, generator=gen)).to(DEV)
        J = (torch.rand(2 * B, 42, generator=gen) * 224).to(DEV)
        res = {}
        for fused in (False, True):
            cfg = dist_loss.LossConfig(weight_type="linear", diff_type="mpjpe", use_wpos=True, use_wneg=True, fuse_dist=fused)
            if b_loc == B:
                zz = z.clone().requires_grad_(True)
                loss = dist_loss.ShardedNtxent.apply(zz, J, cfg, None, None, None)
                loss.backward()
                res[fused] = (loss.item(), zz.grad.clone())
            # timing of one rank's work (rows 0 .. 2 b_loc of N) through the raw ops
            from simhand_amd import ops
            stats = torch.zeros(8, dtype=torch.float64, device=DEV)
            plan = ops.NtxentPlan(B, b_loc, 0, "linear", True, True)
            dpos = ops.pos_dist(J, B, "mpjpe", stats)

            def run():
                D = ops.neg_dist(J, B, "mpjpe", b_loc, 0, stats, stats_only=fused)
                if fused:
                    neg, _ = ops.ntxent_fwd_fused(plan, z, J, "mpjpe", dpos, stats)
                else:
                    neg, _ = ops.ntxent_fwd(plan, z, D, dpos, stats)
                neg_all = neg if b_loc == B else torch.ones(2 * B, device=DEV)
                if fused:
                    ops.ntxent_bwd_fused(plan, z, J, "mpjpe", dpos, stats, neg_all, None)
                else:
                    ops.ntxent_bwd(plan, z, D, dpos, stats, neg_all, None)

            run()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                run()
            torch.cuda.synchronize()
            print(f"loss path N={2 * B} rows={2 * b_loc} fused={fused}: {(time.perf_counter() - t0) / 5 * 1e3:.3f} ms")
        if b_loc == B:
            assert res[False][0] == res[True][0] and torch.equal(res[False][1], res[True][1])


def test_explicit_asymmetric_negative_weights_gradient():
    """The functional surface takes ANY (N,N) neg_weights tensor, like the reference (src/models/utils.py:391-427, :468-501):
    for a non-symmetric one the column term of dL/dz must use w_ji, not w_ij.  Checked against autograd of the
    oracle in fp64."""
    from simhand_amd.host import model_utils as mu

    g = torch.Generator().manual_seed(12)
    B = 24
    z1 = torch.nn.functional.normalize(torch.randn(B, 128, generator=g))
    z2 = torch.nn.functional.normalize(torch.randn(B, 128, generator=g))
    wn = torch.rand(2 * B, 2 * B, generator=g)          # deliberately NOT symmetric
    wp = torch.rand(B, generator=g)
    assert (wn - wn.t()).abs().max() > 0.5
    a, b = z1.double().requires_grad_(True), z2.double().requires_grad_(True)
    want = orc.ntxent(a, b, wp.double(), wn.double())
    want.backward()
    for fn, args in ((mu.vanila_weights_contrastive_loss, (wp.to(DEV), wn.to(DEV))), (mu.vanila_neg_weights_contrastive_loss, (wn.to(DEV),))):
        x1, x2 = z1.to(DEV).requires_grad_(True), z2.to(DEV).requires_grad_(True)
        loss = fn(x1, x2, *args)
        loss.backward()
        if len(args) == 2:
            assert abs(loss.item() - want.item()) <= 1e-5 * abs(want.item())
            _close_grad(x1.grad.cpu().numpy(), a.grad.float().numpy(), "asymmetric dz1")
            _close_grad(x2.grad.cpu().numpy(), b.grad.float().numpy(), "asymmetric dz2")
        else:
            c, d = z1.double().requires_grad_(True), z2.double().requires_grad_(True)
            w2 = orc.ntxent(c, d, None, wn.double())
            w2.backward()
            assert abs(loss.item() - w2.item()) <= 1e-5 * abs(w2.item())
            _close_grad(x1.grad.cpu().numpy(), c.grad.float().numpy(), "asymmetric (neg only) dz1")


def test_projection_width_is_checked_at_the_c_abi():
    """A direct ABI caller (the INTEGRATION.md route) that hands over rows of another width gets an error code and a message,
    not an out-of-bounds access: `width` travels to simhand_proj_postprocess_{fwd,bwd} / simhand_proj_stats (SH_PROJ_DIM = 128)."""
    import ctypes as C

    from simhand_amd import _lib

    lib = _lib.load()
    P = torch.randn(8, 64, device=DEV)
    Z = torch.empty_like(P)
    ws = torch.empty(8, 8, device=DEV)
    out = torch.empty(8, device=DEV)
    null = C.c_void_p(0)
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    rc = lib.simhand_proj_postprocess_fwd(C.c_void_p(P.data_ptr()), 8, 64, null, null, null, null, null, 0, 0, 3, C.c_void_p(Z.data_ptr()), s)
    assert rc != 0 and b"128" in lib.simhand_last_error()
    rc = lib.simhand_proj_postprocess_bwd(C.c_void_p(P.data_ptr()), 8, 64, null, null, null, null, null, 0, 0, 3, C.c_void_p(Z.data_ptr()),
                                          C.c_void_p(Z.data_ptr()), s)
    assert rc != 0
    assert lib.simhand_proj_stats(C.c_void_p(P.data_ptr()), 8, 64, C.c_void_p(ws.data_ptr()), C.c_void_p(out.data_ptr()), s) != 0
    P2 = torch.randn(8, 128, device=DEV)
    Z2 = torch.empty_like(P2)
    assert lib.simhand_proj_postprocess_fwd(C.c_void_p(P2.data_ptr()), 8, 128, null, null, null, null, null, 0, 0, 3, C.c_void_p(Z2.data_ptr()), s) == 0
    torch.cuda.synchronize()
    assert torch.allclose(Z2, torch.nn.functional.normalize(torch.nn.functional.normalize(P2)), atol=1e-6)
