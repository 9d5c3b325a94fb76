"""GPU parity of the BENCHMARKED configurations against the CPU oracle ("Next" #1):

  * configs[1] -- ResNet-50 `handclr_w` in bf16 at 224 x 224: the composed step (every bf16-only kernel route: the 256 x 256
    LDS-DMA tile kernel, the activation-stationary 1x1 kernel, the register-resident 64-channel 3x3 kernel, the all-taps
    3x3 weight gradient, the Gram-matrix BatchNorm folds forward and backward, the two-segment data gradient) -- loss,
    embeddings and every parameter gradient against `oracle.StepOracle` (fp32);
  * configs[3] -- ResNet-152 `peclr_w` bf16;
  * configs[2] -- the loss at the 8-GPU size N = 16 384, 8 row shards.

Conditioning.  A BatchNorm-ResNet at plain random init is CHAOTIC (perturbations grow exponentially with depth: measured
here, the oracle's own bf16-storage twin ends 15-20 % away from the fp32 oracle at the encoder output of ResNet-50 with
16 images, gradient cosines ~0.6 -- any implementation's), which no trained network is.  The strict tests therefore
load seeded-init weights whose last BatchNorm of every residual block has gamma = 0.1 (the "zero-init-residual"
regime: each block a small correction to its shortcut), in BOTH oracle and product; there bf16 round-off stays
round-off and the bands below are tight enough to catch a wrong coefficient in any single layer.  The plain random init
is run as well and bounded relative to the twin only.

Tolerances for bf16.  bf16 has 8 mantissa bits; through 53 (155) convolution + BatchNorm layers at random init the
round-off of the STORED tensors is amplified by every BatchNorm (division by a batch standard deviation), so the
deviation of ANY bf16 implementation from the fp32 oracle is orders of magnitude above fp32 round-off and depends on
depth and batch.  The yardstick is therefore measured, not guessed: `_bf16_storage_twin` is the oracle itself with
bf16-rounded weights and every convolution input / output rounded to bf16 (fp32 accumulate) -- the numerics model of the
HIP path (bf16 operands and stored tensors, fp32 MFMA accumulators).  The HIP result has to be as close to the fp32
oracle as that twin is (x SLACK), AND inside the absolute bands stated in each test.  The fp32 mode of the same nets is
checked at <= 1e-3 (loss, embeddings) next to it.  Gradients are compared with the HIP forward's ReLU masks imposed on
the oracle (tests/_kink.py: a near-zero activation may take either side of the kink; the strict numbers are conditional
on the product's own masks -- the unconditioned comparison is printed and bounded more loosely).
"""
import copy

import pytest
import torch
from torch import nn

from oracle import step as orc
from tests._kink import impose_relu_masks, record_hip_relu_masks

pytestmark = pytest.mark.gpu
DEV = "cuda"
AUG = ["color_jitter", "crop", "random_crop", "resize", "rotate"]
SLACK = 2.0  # HIP deviation <= SLACK x the deviation of the oracle's own bf16-storage twin (+ the absolute floors below)


def _config(size, wcfg, b):
    from simhand_amd.host.config import edict

    return edict(resnet_size=size, projection_head_input_dim=2048, projection_head_hidden_dim=512, output_dim=128,
                 augmentation=list(AUG), joints_type="augmented", use_pca=False, non_linear_lambda_pos=5.0,
                 non_linear_lambda_neg=0.05, lr=1e-4, opt_weight_decay=1e-6, warmup_epochs=10, num_of_mini_batch=1,
                 optimizer="LARS", batch_size=b, num_samples=8 * b, **wcfg)


def _product(cname, size, wcfg, om, dtype, b):
    from simhand_amd.host import unsupervised

    model = getattr(unsupervised, cname)(_config(size, wcfg, b), None, "train")
    model.load_state_dict(om.state_dict(), strict=True)
    model.set_compute_dtype(dtype)
    return model.to(DEV).train()


def _bf16_storage_twin(om: nn.Module, storage: torch.dtype = torch.bfloat16) -> nn.Module:
    """The oracle with the HIP path's storage model: encoder weights rounded to the 16-bit storage type (bf16, or fp16 for the
    precision=16 mode), every encoder convolution's input and output rounded to it (the cast is differentiable, so gradients
    crossing it are rounded as well)."""
    keep, om.last = om.last, {}  # non-leaf tensors of the last step cannot be deep-copied
    twin = copy.deepcopy(om)
    om.last = keep
    rnd = lambda t: t.to(storage).to(torch.float32)  # noqa: E731
    with torch.no_grad():
        for m in twin.encoder.modules():
            if isinstance(m, nn.Conv2d):
                m.weight.copy_(rnd(m.weight))
    for m in twin.encoder.modules():
        if isinstance(m, nn.Conv2d):
            m.register_forward_pre_hook(lambda mod, inp: (rnd(inp[0]),))
            m.register_forward_hook(lambda mod, inp, out: rnd(out))
    return twin


def _cos(a, b):
    a, b = a.flatten().double(), b.flatten().double()
    return float(torch.dot(a, b) / (a.norm() * b.norm() + 1e-300))


def _grad_cosines(named_grads, om):
    og = dict(om.named_parameters())
    out = {}
    for k, g in named_grads.items():
        w = og[k].grad
        if w is None or g is None or w.abs().max() < 1e-7:
            continue  # BN-cancelled biases: analytically zero
        out[k] = _cos(g.cpu().float(), w)
    return out


def _summary(cos: dict):
    v = sorted(cos.values())
    return {"median": v[len(v) // 2], "p10": v[len(v) // 10], "min": v[0], "argmin": min(cos, key=cos.get)}


def _oracle(exp, size, wcfg, seed, gamma3):
    torch.manual_seed(seed)
    om = orc.StepOracle(exp, size, AUG, **wcfg).train()
    if gamma3 is not None:
        with torch.no_grad():
            for k, p in om.named_parameters():
                if k.endswith("bn3.weight") or (size in ("18", "34") and k.endswith("bn2.weight")):
                    p.fill_(gamma3)
    return om


def _run_case(cname, exp, size, wcfg, b, img, seed, force_256, gamma3=0.1):
    """HIP bf16 step vs fp32 oracle vs the oracle's bf16-storage twin.  Returns a dict of deviations."""
    from simhand_amd import _lib, ops

    batch = orc.synthetic_batch(b, size=img, seed=seed)
    dev_batch = {k: v.to(DEV) for k, v in batch.items()}
    om = _oracle(exp, size, wcfg, seed, gamma3)
    model = _product(cname, size, wcfg, om, torch.bfloat16, b)
    lib = _lib.load()
    masks = []
    try:
        if force_256:
            lib.simhand_test_igemm256_enable(2)  # "whenever legal": at 16 images the 14^2 / 7^2 layers are below the default size gate
        ops.route_reset()
        with record_hip_relu_masks(masks):
            out = model.training_step(dev_batch, 0)
        out["loss"].backward()
        routes = ops.route_counts()
    finally:
        ops.hooks_reset()
    loss = float(out["loss"].detach())
    with torch.no_grad():
        z1, z2 = model.get_transformed_projections(dev_batch)
    z = torch.cat((z1, z2)).float().cpu()
    grads = {k: (p.grad.detach().clone() if p.grad is not None else None) for k, p in model.named_parameters()}

    # fp32 oracle, plain (its own ReLU decisions)
    lo = om.contrastive_step(batch)
    lo.backward()
    z_o = om.last["z"].detach().clone()
    cos_plain = _grad_cosines(grads, om)
    # fp32 oracle with the HIP forward's ReLU masks imposed
    om.zero_grad()
    with impose_relu_masks(om, masks) as flips:
        lo_m = om.contrastive_step(batch)
    lo_m.backward()
    cos_masked = _grad_cosines(grads, om)
    ref_grads = {k: p.grad.detach().clone() for k, p in om.named_parameters() if p.grad is not None}
    # the oracle's bf16-storage twin (same masks: isolates storage round-off from kink flips)
    twin = _bf16_storage_twin(om).train()
    twin.zero_grad()
    with impose_relu_masks(twin, masks):
        lt = twin.contrastive_step(batch)
    lt.backward()
    z_t = twin.last["z"].detach()
    cos_twin = {}
    for k, p in twin.named_parameters():
        if p.grad is not None and k in ref_grads and ref_grads[k].abs().max() >= 1e-7:
            cos_twin[k] = _cos(p.grad, ref_grads[k])
    row_cos = lambda a, bb: torch.nn.functional.cosine_similarity(a.double(), bb.double(), dim=1)  # noqa: E731
    enc_o = om.last["encoding"].detach()
    res = {
        "routes": routes, "flips": sum(flips), "mask_elems": sum(m.numel() for m in masks),
        "loss_hip": loss, "loss_oracle": float(lo), "loss_twin": float(lt),
        "z_err_hip": float(1 - row_cos(z, z_o).mean()), "z_err_twin": float(1 - row_cos(z_t, z_o).mean()),
        "z_min_cos_hip": float(row_cos(z, z_o).min()),
        "grad_plain": _summary(cos_plain), "grad_masked": _summary(cos_masked), "grad_twin": _summary(cos_twin),
        "no_grad": sorted(k for k, g in grads.items() if g is None),
    }
    print("\nCASE", cname, size, b, img, "gamma3", gamma3, {k: v for k, v in res.items() if k != "routes"}, flush=True)
    print("ROUTES", {k: v for k, v in routes.items() if v}, flush=True)
    return res


def _check_bf16(res, loss_band, zcos_floor, gmed_floor, gp10_floor, flip_frac):
    """Absolute bands (None = not asserted: plain random init) AND the twin-relative criterion."""
    dl_h = abs(res["loss_hip"] - res["loss_oracle"]) / abs(res["loss_oracle"])
    dl_t = abs(res["loss_twin"] - res["loss_oracle"]) / abs(res["loss_oracle"])
    gm, gt = res["grad_masked"], res["grad_twin"]
    if loss_band is not None:
        assert dl_h <= loss_band, ("loss", dl_h, dl_t)
        assert res["z_err_hip"] <= 1 - zcos_floor, ("z", res["z_err_hip"], res["z_err_twin"])
        assert gm["median"] >= gmed_floor and gm["p10"] >= gp10_floor, ("grads", gm, gt)
    assert dl_h <= SLACK * dl_t + 5e-3, ("loss vs the bf16-storage twin", dl_h, dl_t)  # one scalar: the twin's own error can be luckily small
    assert res["z_err_hip"] <= SLACK * res["z_err_twin"] + 1e-4, ("z vs twin", res["z_err_hip"], res["z_err_twin"])
    assert 1 - gm["median"] <= SLACK * (1 - gt["median"]) + 2e-3, ("grad median vs twin", gm, gt)
    assert 1 - gm["p10"] <= SLACK * (1 - gt["p10"]) + 5e-3, ("grad p10 vs twin", gm, gt)
    assert res["no_grad"] == ["encoder.final_layer.0.bias", "encoder.final_layer.0.weight"]
    # bf16 values near zero take either side of the ReLU kink: a fraction of a percent of the activations; more = a bug
    assert res["flips"] <= flip_frac * res["mask_elems"], (res["flips"], res["mask_elems"])


BF16_ROUTES_RN50 = ("stem_fwd", "stem_bn_pool", "c64_fwd", "c64_dgrad", "gemm1x1_fwd", "gemm1x1_fwd_bnact", "gemm1x1_dgrad", "igemm128_fwd",
                    "igemm128_dgrad", "igemm256_fwd", "igemm256_dgrad", "fwd_bnact", "dgrad_concat", "dgrad_fused_sums", "dgrad_parity",
                    "wgrad3x3", "wgrad_plain", "wgrad_generic", "wgrad_stem", "wgrad_colsum", "bn_apply_gram", "bn_fold_fwd", "bn_fold_bwd", "bn_apply",
                    "bn_bwd_apply", "ntxent_fwd", "ntxent_bwd", "fwd_chain", "dgrad_dysrc", "fwd_bnin", "n128_fwd", "n128_dgrad")


def test_rn50_handclr_w_bf16_at_the_reference_128px_geometry():
    """The reference's own training geometry (README recipes pass --resize; src/experiments/config/training_config.json:38-41 resizes to 128 x 128):
    the ResNet-50 handclr_w bf16 step at 128^2 against the oracle and its bf16-storage twin, with the routes the production dispatch takes at
    that geometry asserted -- since round 6 the stem's LDS-ring kernels (forward and weight gradient) take 128 x 128 inputs too
    (VERDICT r5 "Next" 4); stage grids 32^2 / 16^2 / 8^2 / 4^2."""
    res = _run_case("HandCLR_W", "simhand_w", "50", dict(weight_type="linear", diff_type="mpjpe", pos_neg="pos_neg"), 16, 128, 13, True)
    for r in BF16_ROUTES_RN50 + ("stem_ring_fwd", "stem_ring_wgrad"):
        assert res["routes"][r] > 0, f"the 128 x 128 step never ran the {r} route"
    assert res["routes"]["stem_ring_fwd"] == 1 and res["routes"]["stem_ring_wgrad"] == 1
    _check_bf16(res, loss_band=1e-2, zcos_floor=0.999, gmed_floor=0.995, gp10_floor=0.99, flip_frac=0.02)
    assert res["z_min_cos_hip"] >= 0.995, res["z_min_cos_hip"]


def test_config1_rn50_handclr_w_bf16_every_route_against_oracle():
    """BASELINE configs[1] arithmetic (ResNet-50 handclr_w, bf16, 224^2, linear MPJPE weights, crop + rotate un-warp) at 8
    pairs, with the 256 x 256 kernel taking every layer it legally can (as it does at the benchmarked 2048 images)."""
    res = _run_case("HandCLR_W", "simhand_w", "50", dict(weight_type="linear", diff_type="mpjpe", pos_neg="pos_neg"), 8, 224, 11, True)
    for r in BF16_ROUTES_RN50:
        assert res["routes"][r] > 0, f"the step never ran the {r} route"
    # measured on MI355X: loss 2.9e-4, mean z cosine 0.99957 (twin 0.99960), gradient cosines median 0.99915 / p10 0.99885 (twin 0.99917 / 0.99887)
    _check_bf16(res, loss_band=1e-2, zcos_floor=0.999, gmed_floor=0.995, gp10_floor=0.99, flip_frac=0.02)
    # the UNCONDITIONED comparison (the oracle keeps its own ReLU decisions; nothing of the product is imposed on it): near-zero
    # activations take either side of the kink, which shifts upstream gradients by 1e-3 .. 1e-2 -- measured on MI355X: median per-tensor
    # cosine 0.966.  A floor, so that a regression cannot hide behind the mask-conditioned / twin-relative criteria above.
    gp = res["grad_plain"]
    assert gp["median"] >= 0.93 and gp["p10"] >= 0.80, ("unconditioned gradient cosines", gp)
    assert res["z_min_cos_hip"] >= 0.995, res["z_min_cos_hip"]


def test_bn_on_load_step_is_bit_identical_to_the_separate_bn_apply_pass():
    """ResNetEngine.bn_on_load (bn1 + ReLU of the 64- and 128-channel Bottlenecks applied inside conv2's LDS ring, the activation a
    by-product of that launch) against the stand-alone bn_apply pass, same weights and batch: the activation, the convolution output and
    its BatchNorm sums are bit-identical, so the whole step is -- loss and every gradient compared with torch.equal.  12 pairs = 24 images:
    both ring kernels run (3 stage-1 units on conv3x3_c64, 3 stage-2 units on conv3x3_r128)."""
    from simhand_amd import ops

    wcfg = dict(weight_type="linear", diff_type="mpjpe", pos_neg="pos_neg")
    b = 12
    batch = {k: v.to(DEV) for k, v in orc.synthetic_batch(b, size=224, seed=23).items()}
    om = _oracle("simhand_w", "50", wcfg, 23, 0.1)
    res = {}
    for on in (False, True):
        model = _product("HandCLR_W", "50", wcfg, om, torch.bfloat16, b)
        model.encoder.engine.bn_on_load = on
        ops.route_reset()
        out = model.training_step(batch, 0)
        out["loss"].backward()
        rc = ops.route_counts()
        res[on] = (out["loss"].detach().float().cpu().clone(), {k: p.grad.detach().float().cpu().clone() for k, p in model.named_parameters()
                                                                 if p.grad is not None}, rc)
        del model
    (l0, g0, r0), (l1, g1, r1) = res[False], res[True]
    assert r0["fwd_bnin"] == 0 and r1["fwd_bnin"] == 6, (r0["fwd_bnin"], r1["fwd_bnin"])
    assert r0["bn_apply"] - r1["bn_apply"] == 6, (r0["bn_apply"], r1["bn_apply"])
    assert torch.equal(l0, l1), (l0, l1)
    assert g0.keys() == g1.keys() and len(g0) > 100
    for k in g0:
        assert torch.equal(g0[k], g1[k]), k


def test_config1_rn50_bf16_plain_random_init_tracks_the_twin():
    """Same step at plain random init (the chaotic regime, see the module docstring): no absolute band is meaningful --
    the HIP path must stay as close to the fp32 oracle as the oracle's bf16-storage twin does."""
    res = _run_case("HandCLR_W", "simhand_w", "50", dict(weight_type="linear", diff_type="mpjpe", pos_neg="pos_neg"), 8, 224, 11, True,
                    gamma3=None)
    _check_bf16(res, None, None, None, None, flip_frac=0.2)
    # absolute sanity floors for the chaotic regime (measured: loss within 2 %, mean z cosine 0.80 (twin 0.68-0.80), masked gradient
    # cosine median 0.58 (twin 0.57)): far below the strict test's bands, far above what a wrong layer produces (cosines ~ 0)
    assert abs(res["loss_hip"] - res["loss_oracle"]) <= 0.1 * abs(res["loss_oracle"]), (res["loss_hip"], res["loss_oracle"])
    assert res["z_err_hip"] <= 0.45, res["z_err_hip"]
    assert res["grad_masked"]["median"] >= 0.35, res["grad_masked"]


def test_config1_rn50_default_routing_at_88_images():
    """Same step at 44 pairs (88 images): 14^2 x 88 = 17 248 rows pass the 256 x 256 kernel's DEFAULT size gate, so
    the production dispatch (no hook touched) is what runs.  Forward parity (loss, embeddings) against the oracle's
    train-mode forward; the backward runs for the route assertions and must be finite."""
    from simhand_amd import ops

    wcfg = dict(weight_type="linear", diff_type="mpjpe", pos_neg="pos_neg")
    b = 44
    batch = orc.synthetic_batch(b, size=224, seed=12)
    dev_batch = {k: v.to(DEV) for k, v in batch.items()}
    om = _oracle("simhand_w", "50", wcfg, 12, 0.1)
    model = _product("HandCLR_W", "50", wcfg, om, torch.bfloat16, b)
    ops.hooks_reset()
    ops.route_reset()
    out = model.training_step(dev_batch, 0)
    out["loss"].backward()
    routes = ops.route_counts()
    for r in ("igemm256_fwd", "igemm256_dgrad", "c64_fwd", "c64_dgrad", "r128_fwd", "r128_dgrad", "gemm1x1_fwd_bnact", "dgrad_concat", "bn_fold_fwd",
              "bn_fold_bwd", "wgrad3x3"):
        assert routes[r] > 0, f"default dispatch never took the {r} route at 88 images"
    for k, p in model.named_parameters():
        assert p.grad is None or bool(torch.isfinite(p.grad).all()), k
    with torch.no_grad():
        lo = om.contrastive_step(batch)
        twin = _bf16_storage_twin(om).train()
        lt = twin.contrastive_step(batch)
        z1, z2 = model.get_transformed_projections(dev_batch)
    z = torch.cat((z1, z2)).float().cpu()
    cosr = lambda a, bb: torch.nn.functional.cosine_similarity(a.double(), bb.double(), dim=1)  # noqa: E731
    e_h, e_t = float(1 - cosr(z, om.last["z"]).mean()), float(1 - cosr(twin.last["z"], om.last["z"]).mean())
    dl_h = abs(float(out["loss"]) - float(lo)) / float(lo)
    dl_t = abs(float(lt) - float(lo)) / float(lo)
    print({"loss": (float(out["loss"]), float(lo), float(lt)), "z_err": (e_h, e_t)})
    assert dl_h <= 1e-2 and dl_h <= SLACK * dl_t + 2e-3, (dl_h, dl_t)
    assert e_h <= 1e-3 and e_h <= SLACK * e_t + 1e-4, (e_h, e_t)


def test_config3_rn152_peclr_w_bf16_against_oracle():
    """BASELINE configs[3] arithmetic: ResNet-152 peclr_w (non-linear w_abs weights on the negatives), bf16, 4 pairs at 224^2."""
    res = _run_case("PeCLR_W", "peclr_w", "152", dict(weight_type="non_linear", diff_type="w_abs", pos_neg="neg"), 4, 224, 13, True)
    for r in ("c64_fwd", "igemm256_fwd", "igemm256_dgrad", "gemm1x1_fwd_bnact", "dgrad_concat", "bn_fold_fwd", "bn_fold_bwd", "wgrad3x3"):
        assert res["routes"][r] > 0, r
    # 155 convolutions at 8 images: three times the depth of ResNet-50, slightly wider bands
    # measured: loss 3.0e-3, mean z cosine 0.9986 (twin 0.9988), gradient cosines median 0.9958 / p10 0.9948 (twin 0.9968 / 0.9959)
    _check_bf16(res, loss_band=1e-2, zcos_floor=0.998, gmed_floor=0.99, gp10_floor=0.985, flip_frac=0.03)


@pytest.mark.parametrize("cname,exp,size,wcfg,b,img", [
    ("HandCLR_W", "simhand_w", "50", dict(weight_type="linear", diff_type="mpjpe", pos_neg="pos_neg"), 8, 224),
    ("PeCLR_W", "peclr_w", "152", dict(weight_type="non_linear", diff_type="w_abs", pos_neg="neg"), 2, 160),
])
def test_same_nets_fp32_mode_within_1e3(cname, exp, size, wcfg, b, img):
    """The north-star bar (1e-3 relative fp32) on the same networks in the exact-fp32 MFMA mode: loss and embeddings."""
    batch = orc.synthetic_batch(b, size=img, seed=17)
    dev_batch = {k: v.to(DEV) for k, v in batch.items()}
    om = _oracle(exp, size, wcfg, 17, 0.1)
    model = _product(cname, size, wcfg, om, torch.float32, b)
    with torch.no_grad():
        loss = model.training_step(dev_batch, 0)["loss"]
        z1, z2 = model.get_transformed_projections(dev_batch)
        lo = om.contrastive_step(batch)
    z = torch.cat((z1, z2)).cpu()
    assert abs(float(loss) - float(lo)) <= 1e-3 * abs(float(lo)), (float(loss), float(lo))
    assert (z - om.last["z"]).abs().max() <= 1e-3 * om.last["z"].abs().max()


def test_config2_loss_at_n16384_eight_row_shards():
    """BASELINE configs[2] loss size: B = 8192 pairs (N = 16 384 rows), linear MPJPE weights, computed as the 8 row shards
    an 8-GPU run uses (one GPU plays all ranks; the all-reduces are the max / min / sum folds of tests/test_gpu_loss.py).
      (1) sharded == unsharded;
      (2) central finite difference along the gradient direction reproduces |dL/dz|;
      (3) loss, the negative sums and dL/dz of a 512-row slice against the closed form evaluated in fp64 on the host
          (distances in fp32 as the kernel takes them; O(N^2) but chunked: ~1 GB of host memory)."""
    from tests.test_gpu_loss import _hip_loss

    B = 8192
    N = 2 * B
    gen = torch.Generator().manual_seed(23)
    z1 = torch.nn.functional.normalize(torch.randn(B, 128, generator=gen))
    z2 = torch.nn.functional.normalize(z1 + 0.5 * torch.randn(B, 128, generator=gen))
    j1 = torch.rand(B, 21, 2, generator=gen) * 224
    j2 = j1 + torch.randn(B, 21, 2, generator=gen) * 8
    l1, dz1 = _hip_loss(z1, z2, j1, j2, "mpjpe", "linear", "pos_neg", ranks=1)
    l8, dz8 = _hip_loss(z1, z2, j1, j2, "mpjpe", "linear", "pos_neg", ranks=8)
    assert abs(l1 - l8) <= 2e-6 * abs(l1), (l1, l8)
    assert (dz1 - dz8).abs().max() <= 1e-5 * dz1.abs().max()
    v = dz8 / dz8.norm()
    eps = 0.25
    lp, _ = _hip_loss(z1 + eps * v[:B], z2 + eps * v[B:], j1, j2, "mpjpe", "linear", "pos_neg", ranks=8, backward=False)
    lm, _ = _hip_loss(z1 - eps * v[:B], z2 - eps * v[B:], j1, j2, "mpjpe", "linear", "pos_neg", ranks=8, backward=False)
    fd = (lp - lm) / (2 * eps)
    assert abs(fd - dz8.norm().item()) <= 2e-2 * dz8.norm().item(), (fd, dz8.norm().item())

    # ---- fp64 closed form on the host -------------------------------------------------------------------------------
    J = torch.cat((j1, j2))                      # (N,21,2) fp32
    Z = torch.cat((z1, z2)).double()
    t = 0.5
    # pass 1: the distance matrix (fp32, 1 GB) and its global max (the min is the diagonal's 0), chunked
    CH = 256
    Dfull = torch.empty(N, N)
    for r0 in range(0, N, CH):
        Dfull[r0:r0 + CH] = (J[r0:r0 + CH, None] - J[None]).norm(dim=-1).mean(dim=-1)  # mpjpe, utils.py:251-253
    dmax = float(Dfull.max())
    dp = (j1 - j2).norm(dim=-1).mean(dim=-1)
    wpos = ((dp.max() - dp) / (dp.max() - dp.min())).double()
    # pass 2: negative sums of every row; keep the slice's rows
    rows = torch.arange(4000, 4512)  # view-1 rows of pairs 4000..4511 (their partners are rows B + k)
    neg = torch.empty(N, dtype=torch.float64)
    keep = {}
    for r0 in range(0, N, CH):
        w = ((dmax - Dfull[r0:r0 + CH]) / dmax).double()
        s = Z[r0:r0 + CH] @ Z.t()
        e = torch.exp(w * s / t)
        idx = torch.arange(r0, min(r0 + CH, N))
        e[idx - r0, idx] = 0.0
        neg[r0:r0 + CH] = e.sum(dim=1)
        if r0 >= 4000 - CH and r0 < 4512:
            keep[r0] = (w, e)
    want_loss_rows, want_dz = [], []
    for i in rows.tolist():
        r0 = (i // CH) * CH
        w, e = keep[r0]
        wi, ei = w[i - r0], e[i - r0]
        k = i  # i < B
        sp = float((Z[i] * Z[B + k]).sum())
        want_loss_rows.append(-(float(wpos[k]) * sp / t - float(torch.log(neg[i]))))
        coef = wi * ei * (1.0 / neg[i] + 1.0 / neg) / (t * N)
        g = coef @ Z
        g -= 2 * float(wpos[k]) / (t * N) * Z[B + k]   # G_ip + G_pi: the positive pair appears in both rows
        want_dz.append(g)
    want_dz = torch.stack(want_dz)
    got_dz = dz8[rows].double()
    err = (got_dz - want_dz).abs().max() / want_dz.abs().max()
    assert err <= 2e-4, err
    # the loss: mean over all N rows of the per-row terms; check it through the slice's rows via the negative sums
    # (neg_i of the slice is what the kernel's forward produced: recompute L from the host's neg for ALL rows)
    sp_all = (Z[:B] * Z[B:]).sum(dim=1)
    want_loss = float(-(torch.cat((wpos * sp_all, wpos * sp_all)) / t - torch.log(neg)).mean())
    assert abs(l8 - want_loss) <= 2e-5 * abs(want_loss), (l8, want_loss)


def test_config4_simclr_loss_at_n32768_eight_row_shards():
    """BASELINE configs[4] loss size: ResNet-50 `simclr` (no weighting), B = 16 384 pairs -> N = 32 768 rows, as the 8 row shards an 8-GPU
    run uses (one GPU plays all ranks).  The loss arithmetic of that config is fp32 whatever the convolutions' operand type is.
      (1) sharded == unsharded; (2) central finite difference along the gradient reproduces |dL/dz|;
      (3) the loss and dL/dz of a 256-row slice against the closed form in fp64 on the host (src/models/utils.py:157-189:
          S = z z^T, neg_i = sum_{j != i} exp(S_ij / tau), L = -mean_i [S_i,pair(i) / tau - log neg_i])."""
    from tests.test_gpu_loss import _hip_loss

    B = 16384
    N = 2 * B
    gen = torch.Generator().manual_seed(29)
    z1 = torch.nn.functional.normalize(torch.randn(B, 128, generator=gen))
    z2 = torch.nn.functional.normalize(z1 + 0.5 * torch.randn(B, 128, generator=gen))
    l1, dz1 = _hip_loss(z1, z2, None, None, "mpjpe", None, "pos_neg", ranks=1)
    l8, dz8 = _hip_loss(z1, z2, None, None, "mpjpe", None, "pos_neg", ranks=8)
    assert abs(l1 - l8) <= 2e-6 * abs(l1), (l1, l8)
    assert (dz1 - dz8).abs().max() <= 1e-5 * dz1.abs().max()
    v = dz8 / dz8.norm()
    eps = 0.25
    lp, _ = _hip_loss(z1 + eps * v[:B], z2 + eps * v[B:], None, None, "mpjpe", None, "pos_neg", ranks=8, backward=False)
    lm, _ = _hip_loss(z1 - eps * v[:B], z2 - eps * v[B:], None, None, "mpjpe", None, "pos_neg", ranks=8, backward=False)
    fd = (lp - lm) / (2 * eps)
    assert abs(fd - dz8.norm().item()) <= 2e-2 * dz8.norm().item(), (fd, dz8.norm().item())
    # fp64 closed form on the host, chunked (N x N never materialised)
    Z = torch.cat((z1, z2)).double()
    t = 0.5
    neg = torch.empty(N, dtype=torch.float64)
    CH = 1024
    for r0 in range(0, N, CH):
        e = torch.exp(Z[r0:r0 + CH] @ Z.t() / t)
        idx = torch.arange(r0, r0 + CH)
        e[idx - r0, idx] = 0.0
        neg[r0:r0 + CH] = e.sum(dim=1)
    sp = (Z[:B] * Z[B:]).sum(dim=1)
    want_loss = float(-(torch.cat((sp, sp)) / t - torch.log(neg)).mean())
    assert abs(l8 - want_loss) <= 2e-5 * abs(want_loss), (l8, want_loss)
    rows = torch.arange(9000, 9256)  # view-1 rows; partners are rows B + k
    e = torch.exp(Z[rows] @ Z.t() / t)
    e[torch.arange(256), rows] = 0.0
    coef = e * (1.0 / neg[rows, None] + 1.0 / neg[None, :]) / (t * N)
    want_dz = coef @ Z - 2.0 / (t * N) * Z[B + rows]
    err = (dz8[rows].double() - want_dz).abs().max() / want_dz.abs().max()
    assert err <= 2e-4, err
