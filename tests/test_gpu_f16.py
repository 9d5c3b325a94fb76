"""GPU: the fp16-storage build (libsimhand_hip_f16.so) -- the reference's own precision policy: `Trainer(precision=16,
amp_backend="native")` (src/experiments/main.py:158-159, config/training_config.json:9) = fp16 storage under autocast + GradScaler.
Same sources as the bf16 build; only the 16-bit unpack / round-to-nearest-even pack / MFMA operand type differ (csrc/common.h), so
the checks are: the operators against ATen on fp16-rounded operands (tolerance = fp16's 2^-11, four times tighter than the bf16
suite's 2^-8 bands), the ResNet-50 step against the fp32 oracle and the oracle's fp16-storage twin, and the loss-scaling loop."""
import math

import pytest
import torch
import torch.nn.functional as F

from oracle import step as orc

pytestmark = pytest.mark.gpu
DEV = "cuda"
H = torch.float16


@pytest.fixture(autouse=True)
def _fp16_build():
    from simhand_amd import _lib

    _lib.use_half("f16")
    assert _lib.load().simhand_half_format() == 1
    yield
    _lib.use_half("bf16")


def _rnd(t):
    return t.to(H).float()


def _check(got, want, tol, tag):
    scale = want.abs().max().item()
    err = (got - want).abs().max().item()
    assert err <= tol * scale + 1e-6, f"{tag}: err {err:.3e} scale {scale:.3e}"


@pytest.mark.parametrize("shape", [(2, 14, 14, 64, 256, 1, 1, 0), (3, 9, 9, 128, 128, 3, 1, 1), (2, 16, 16, 64, 64, 3, 1, 1), (2, 16, 16, 128, 128, 3, 2, 1),
                                   (2, 16, 16, 256, 512, 1, 2, 0), (2, 20, 20, 1024, 256, 1, 1, 0), (3, 14, 14, 256, 256, 3, 1, 1), (5, 7, 7, 512, 64, 1, 1, 0),
                                   (24, 28, 28, 128, 128, 3, 1, 1), (6, 56, 56, 128, 128, 3, 2, 1), (90, 14, 14, 1024, 256, 1, 1, 0)])
def test_conv_fwd_dgrad_wgrad_fp16_storage(shape):
    """Every kernel family (activation-stationary 1x1, 128-row tile kernel, 256 x 256 LDS-DMA kernel [forced], register-resident 64-channel
    3x3, the 128-channel LDS-ring 3x3 [>= 16 384 padded positions], all-taps (stride 1 and 2) / generic / pointer-walking / LDS-DMA 1x1
    [>= 16 384 pixels] weight gradients) in the fp16 build."""
    from simhand_amd import _lib, ops

    _lib.load().simhand_test_igemm256_enable(2)
    n, h, w, cin, cout, k, stride, pad = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = _rnd(torch.randn(n, cin, h, w, generator=g)).requires_grad_(True)
    wt = _rnd(torch.randn(cout, cin, k, k, generator=g) / math.sqrt(cin * k * k)).requires_grad_(True)
    y = F.conv2d(x, wt, stride=stride, padding=pad)
    dy = _rnd(torch.randn(y.shape, generator=g))
    y.backward(dy)
    d = ops.conv_desc(n, h, w, cin, cout, k, k, stride, pad, H)
    xd = x.detach().permute(0, 2, 3, 1).contiguous().to(DEV).to(H)
    wd, wtd = ops.pack_krsc(wt.detach().to(DEV), H), ops.pack_crsk(wt.detach().to(DEV), H)
    ops.route_reset()
    yd, part = ops.conv2d_fwd(d, xd, wd, want_stats=True)
    assert yd.dtype == H
    if shape == (24, 28, 28, 128, 128, 3, 1, 1):
        assert ops.route_counts()["r128_fwd"] == 1
    _check(yd.float().cpu().permute(0, 3, 1, 2), y.detach(), 2e-3, "fwd")
    m = n * d.ho * d.wo
    yf = y.detach().permute(0, 2, 3, 1).reshape(m, cout)
    _check(part[:, 0].sum(0).cpu() / m, yf.mean(0), 1e-2, "stat mean")
    dyd = dy.permute(0, 2, 3, 1).contiguous().to(DEV).to(H)
    _check(ops.conv2d_dgrad(d, dyd, wtd).float().cpu().permute(0, 3, 1, 2), x.grad, 2e-3, "dgrad")
    _check(ops.conv2d_wgrad_oihw(d, xd, dyd, (cout, cin, k, k)).cpu(), wt.grad, 1e-3, "wgrad")


@pytest.mark.parametrize("n,h,c,route", [(3, 56, 64, "c64_fwd"), (24, 28, 128, "r128_fwd")])
def test_bn_on_load_forward_fp16_storage(n, h, c, route):
    """simhand_conv2d_fwd_bnin in the fp16 build: the pad page holds fp16 NaNs (0x7e00) there -- a bf16 NaN pattern read as fp16 would be a
    finite number and leak scale * x + shift into the padding.  Bit-identical to bn_apply + conv2d_fwd, zero scale / positive shift included."""
    from simhand_amd import ops

    g = torch.Generator().manual_seed(n + h + c)
    d = ops.conv_desc(n, h, h, c, c, 3, 3, 1, 1, H)
    assert ops.conv2d_fwd_bnin_ok(d)
    y_in = (torch.randn(n, h, h, c, generator=g) * 2.0 + 0.3).to(DEV).to(H)
    wt = _rnd(torch.randn(c, c, 3, 3, generator=g) / math.sqrt(9 * c)).to(DEV)
    wk = ops.pack_krsc(wt, H)
    st = ops.BNState(c, DEV)
    st.scale.copy_((torch.randn(c, generator=g) * 0.8).to(DEV))
    st.shift.copy_((torch.randn(c, generator=g) * 0.5).to(DEV))
    st.scale[3] = 0.0
    st.shift[3] = 0.7
    m = n * h * h
    a_ref = ops.bn_apply(y_in.view(m, c), st, m, c, True, None).view(n, h, h, c)
    y_ref, p_ref = ops.conv2d_fwd(d, a_ref, wk, want_stats=True)
    ops.route_reset()
    a, y, part = ops.conv2d_fwd_bnin(d, y_in, st, wk, want_stats=True)
    rc = ops.route_counts()
    assert rc["fwd_bnin"] == 1 and rc[route] == 1
    assert torch.equal(a, a_ref) and torch.equal(y, y_ref) and torch.equal(part, p_ref)
    assert bool(torch.isfinite(y.float()).all())


def test_a_bf16_tensor_is_refused_by_the_fp16_build():
    from simhand_amd import _lib, ops

    x = torch.zeros(2, 4, 4, 64, dtype=torch.bfloat16, device=DEV)
    d = ops.conv_desc(2, 4, 4, 64, 64, 1, 1, 1, 0, H)
    with pytest.raises(_lib.SimhandHipError):
        ops.conv_desc(2, 4, 4, 64, 64, 1, 1, 1, 0, torch.bfloat16)
    assert d.dtype == _lib.SH_BF16  # the enum value "16-bit storage"; its meaning is the build's


def test_batchnorm_fwd_bwd_fp16_storage():
    from simhand_amd import ops

    g = torch.Generator().manual_seed(4)
    m, c = 1000, 256
    y = _rnd(torch.randn(m, c, generator=g) * 1.5 + 0.3)
    gamma, beta = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.2
    da = _rnd(torch.randn(m, c, generator=g))
    yr = y.clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    a = F.relu(F.batch_norm(yr, None, None, gr, br, True, 0.1, 1e-5))
    a.backward(da)
    yd = y.to(DEV).to(H)
    part = ops.bn_partial_stats(yd, m, c)
    st = ops.bn_finalize(part, m, c, gamma.to(DEV), beta.to(DEV), torch.zeros(c, device=DEV), torch.ones(c, device=DEV),
                         torch.zeros(1, dtype=torch.int64, device=DEV))
    ad = ops.bn_apply(yd, st, m, c, True)
    _check(ad.float().cpu(), a.detach(), 2e-3, "bn fwd")
    dyd, _, dg, db = ops.bn_backward(da.to(DEV).to(H), ad, yd, st, gamma.to(DEV), m, c, True, False)
    _check(dyd.float().cpu(), yr.grad, 4e-3, "bn dy")
    _check(dg.cpu(), gr.grad, 2e-3, "dgamma")
    _check(db.cpu(), br.grad, 2e-3, "dbeta")


def test_rn50_handclr_w_step_fp16_storage_against_oracle_and_its_fp16_twin():
    """BASELINE configs[1]'s network in the REFERENCE's precision (fp16 storage): loss, embeddings and per-tensor gradients against the
    fp32 oracle, next to the oracle's own fp16-storage twin.  Same conditioning as tests/test_gpu_configs.py (gamma3 = 0.1, the HIP
    forward's ReLU masks imposed for the strict gradient comparison).  Loss scaling 2^12 on the product's backward, divided back out."""
    from simhand_amd import _lib, ops
    from tests._kink import impose_relu_masks, record_hip_relu_masks
    from tests.test_gpu_configs import _bf16_storage_twin, _cos, _grad_cosines, _oracle, _product, _summary

    wcfg = dict(weight_type="linear", diff_type="mpjpe", pos_neg="pos_neg")
    b = 8
    batch = orc.synthetic_batch(b, size=224, seed=11)
    dev_batch = {k: v.to(DEV) for k, v in batch.items()}
    om = _oracle("simhand_w", "50", wcfg, 11, 0.1)
    model = _product("HandCLR_W", "50", wcfg, om, H, b)
    assert _lib.half_format() == "f16"
    _lib.load().simhand_test_igemm256_enable(2)
    masks = []
    scale = 4096.0
    ops.route_reset()
    with record_hip_relu_masks(masks):
        out = model.training_step(dev_batch, 0)
    (out["loss"] * scale).backward()
    routes = ops.route_counts()
    for r in ("igemm256_fwd", "igemm256_dgrad", "c64_fwd", "gemm1x1_fwd_bnact", "dgrad_concat", "bn_fold_fwd", "bn_fold_bwd", "wgrad3x3", "fwd_chain", "dgrad_dysrc"):
        assert routes[r] > 0, r
    loss = float(out["loss"].detach())
    with torch.no_grad():
        z = torch.cat(model.get_transformed_projections(dev_batch)).float().cpu()
    grads = {k: (p.grad.detach().float() / scale if p.grad is not None else None) for k, p in model.named_parameters()}
    lo = om.contrastive_step(batch)
    lo.backward()
    z_o = om.last["z"].detach().clone()
    om.zero_grad()
    with impose_relu_masks(om, masks):
        lm = om.contrastive_step(batch)
    lm.backward()
    gm = _summary(_grad_cosines(grads, om))
    ref_grads = {k: p.grad.detach().clone() for k, p in om.named_parameters() if p.grad is not None}
    twin = _bf16_storage_twin(om, torch.float16).train()
    twin.zero_grad()
    with impose_relu_masks(twin, masks):
        lt = twin.contrastive_step(batch)
    lt.backward()
    gt = _summary({k: _cos(p.grad, ref_grads[k]) for k, p in twin.named_parameters()
                   if p.grad is not None and k in ref_grads and ref_grads[k].abs().max() >= 1e-7})
    rc = lambda a, bb: F.cosine_similarity(a.double(), bb.double(), dim=1)  # noqa: E731
    zh, zt = float(1 - rc(z, z_o).mean()), float(1 - rc(twin.last["z"].detach(), z_o).mean())
    dl_h, dl_t = abs(loss - float(lo)) / abs(float(lo)), abs(float(lt) - float(lo)) / abs(float(lo))
    print({"loss": (loss, float(lo), float(lt)), "z_err": (zh, zt), "grad_masked": gm, "grad_twin": gt})
    # fp16 keeps 3 more significand bits than bf16: bands 8x tighter than test_config1's (loss 1e-2 / z 1e-3 / gradients 5e-3)
    assert dl_h <= 1.5e-3 and dl_h <= 2 * dl_t + 1e-3, (dl_h, dl_t)
    assert zh <= 1.5e-4 and zh <= 2 * zt + 2e-5, (zh, zt)
    assert gm["median"] >= 0.9995 and gm["p10"] >= 0.999, (gm, gt)
    assert 1 - gm["median"] <= 2 * (1 - gt["median"]) + 3e-4, (gm, gt)


def test_trainer_precision_16_scales_the_loss_and_skips_overflowed_steps(tmp_path):
    """`--precision 16` (the reference's training_config.json default) = fp16 storage + GradScaler: a short run trains with finite losses,
    the scaler state is in the checkpoint under Lightning's key, and an absurd initial scale overflows -> the step is skipped, the scale
    halves, the weights do not move."""
    from simhand_amd import _lib
    from simhand_amd.host.main import main

    argv = ["--experiment_type", "handclr_w", "--color_jitter", "--random_crop", "--rotate", "--crop", "--resize", "-resnet_size", "18",
            "-sources", "ego4d", "--datasets_scale", "1m", "-epochs", "1", "-batch_size", "8", "-save_top_k", "1", "--weight_type", "linear",
            "--joints_type", "augmented", "--diff_type", "mpjpe", "--pos_neg", "pos_neg", "--synthetic", "--synthetic_samples", "32", "--image_size", "64",
            "--precision", "16", "--max_steps", "4", "--out_dir", str(tmp_path)]
    t = main(argv)
    assert _lib.half_format() == "f16" and t.scaler.enabled and t.scaler.get_scale() == 65536.0 and t.scaler.skipped_steps == 0
    losses = [float(x) for x in t.step_losses]
    assert len(losses) == 4 and all(l == l and 0 < l < 20 for l in losses), losses
    import glob
    ck = torch.load(glob.glob(str(tmp_path / "checkpoints" / "*.ckpt"))[0], map_location="cpu", weights_only=False)
    assert ck["native_amp_scaling_state"]["scale"] == 65536.0
    # overflow: scale 2^40 makes the fp16 gradient tensors inf -> skipped step, halved scale
    from simhand_amd.host import lightning

    orig = lightning.Trainer.__init__

    def patched(self, *a, **k):
        orig(self, *a, **k)
        self.scaler._scale = 2.0 ** 40

    lightning.Trainer.__init__ = patched
    try:
        t2 = main(argv[:-4] + ["--max_steps", "2", "--out_dir", str(tmp_path / "b")])
    finally:
        lightning.Trainer.__init__ = orig
    assert t2.scaler.skipped_steps == 2 and t2.scaler.get_scale() == 2.0 ** 38
