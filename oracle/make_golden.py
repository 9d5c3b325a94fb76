"""Generate the golden fixtures under tests/golden by executing the
reference's own Python on CPU (build container only; see oracle/ref_import.py).

    python -m oracle.make_golden            # rewrites tests/golden/*.npz, *.json

Each fixture holds the seeded inputs and the reference's outputs; while it
runs, this script also asserts that ``oracle/step.py`` (the restatement that
travels to the GPU box) agrees with the reference on the same inputs.

TEST INFRASTRUCTURE ONLY.
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np
import torch
from torch.nn import functional as F

from oracle import ref_import, step

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
DIFFS = ("mpjpe", "w_abs", "w_o_abs")
NONLINEAR = (5.0, 0.05)


def _np(t):
    return t.detach().cpu().numpy()


def _inputs(b: int, seed: int):
    g = torch.Generator().manual_seed(seed)
    z1 = F.normalize(torch.randn(b, 128, generator=g))
    z2 = F.normalize(torch.randn(b, 128, generator=g))
    j1 = torch.rand(b, 21, 2, generator=g) * 128
    j2 = j1 + torch.randn(b, 21, 2, generator=g) * 4
    return z1, z2, j1, j2


def golden_loss(mu) -> None:
    """a10 + a11 + a12 for every diff_type x weight_type x pos_neg at B in {2,8,32}."""
    for b in (2, 8, 32):
        z1, z2, j1, j2 = _inputs(b, 100 + b)
        out = {"z1": _np(z1), "z2": _np(z2), "j1": _np(j1), "j2": _np(j2)}
        for diff in DIFFS:
            for wt in ("linear", "non_linear"):
                if wt == "linear":
                    wp, wn = mu.get_weights_linear(j1, j2, diff)
                    wp_o, wn_o = step.weights_linear(j1, j2, diff)
                else:
                    wp, wn = mu.get_weights_nonlinear(j1, j2, NONLINEAR[0], NONLINEAR[1], diff)
                    wp_o, wn_o = step.weights_nonlinear(j1, j2, NONLINEAR[0], NONLINEAR[1], diff)
                assert torch.allclose(wp, wp_o, rtol=1e-6, atol=1e-7, equal_nan=True), (b, diff, wt)
                assert torch.allclose(wn, wn_o, rtol=1e-6, atol=1e-7, equal_nan=True), (b, diff, wt)
                tag = f"{diff}.{wt}"
                out[f"wpos.{tag}"] = _np(wp)
                out[f"wneg.{tag}"] = _np(wn)
                for mode in ("pos_neg", "pos", "neg"):
                    a = z1.clone().requires_grad_(True)
                    c = z2.clone().requires_grad_(True)
                    if mode == "pos_neg":
                        loss = mu.vanila_weights_contrastive_loss(a, c, wp, wn)
                        lo = step.ntxent(z1, z2, wp, wn)
                    elif mode == "pos":
                        loss = mu.vanila_pos_weights_contrastive_loss(a, c, wp)
                        lo = step.ntxent(z1, z2, wp, None)
                    else:
                        loss = mu.vanila_neg_weights_contrastive_loss(a, c, wn)
                        lo = step.ntxent(z1, z2, None, wn)
                    loss.backward()
                    if torch.isfinite(loss):
                        assert abs(loss.item() - lo.item()) <= 2e-6 * max(1.0, abs(loss.item())), (b, tag, mode)
                    out[f"loss.{tag}.{mode}"] = _np(loss)
                    out[f"dz1.{tag}.{mode}"] = _np(a.grad)
                    out[f"dz2.{tag}.{mode}"] = _np(c.grad)
        a = z1.clone().requires_grad_(True)
        c = z2.clone().requires_grad_(True)
        loss = mu.vanila_contrastive_loss(a, c)
        loss.backward()
        assert abs(loss.item() - step.ntxent(z1, z2).item()) < 2e-6 * abs(loss.item())
        out["loss.simclr"], out["dz1.simclr"], out["dz2.simclr"] = _np(loss), _np(a.grad), _np(c.grad)
        np.savez_compressed(os.path.join(OUT, f"loss_B{b}.npz"), **out)


def golden_pca(mu) -> None:
    """a10 PCA variants (src/models/utils.py:264-301,349-388) on given 14-D
    features; torch.pca_lowrank itself (:212) is randomised and host-side, so
    only the distance/weight arithmetic after it is pinned."""
    g = torch.Generator().manual_seed(77)
    f1 = torch.randn(8, 14, generator=g) * 30
    f2 = f1 + torch.randn(8, 14, generator=g) * 3
    out = {"f1": _np(f1), "f2": _np(f2)}
    for diff in DIFFS:
        wp, wn = mu.get_weights_linear_with_pca(f1, f2, diff)
        wp_o, wn_o = step.weights_linear(f1, f2, diff)
        assert torch.allclose(wp, wp_o, rtol=1e-6, atol=1e-7) and torch.allclose(wn, wn_o, rtol=1e-6, atol=1e-7)
        out[f"wpos.{diff}.linear"], out[f"wneg.{diff}.linear"] = _np(wp), _np(wn)
        wp, wn = mu.get_weights_nonlinear_with_pca(f1, f2, NONLINEAR[0], NONLINEAR[1], diff)
        wp_o, wn_o = step.weights_nonlinear(f1, f2, NONLINEAR[0], NONLINEAR[1], diff)
        assert torch.allclose(wp, wp_o, rtol=1e-6, atol=1e-7) and torch.allclose(wn, wn_o, rtol=1e-6, atol=1e-7)
        out[f"wpos.{diff}.non_linear"], out[f"wneg.{diff}.non_linear"] = _np(wp), _np(wn)
    np.savez_compressed(os.path.join(OUT, "weights_pca.npz"), **out)


class _Preset(torch.nn.Module):
    """Stands in for ``self.encoder`` so the reference's own
    get_transformed_projections runs on a chosen head output."""

    def __init__(self, value):
        super().__init__()
        self.value = value

    def forward(self, x):
        return self.value


def golden_postprocess() -> None:
    """a3/a6/a7/a8/a9 through the reference's HandCLR_W.get_transformed_projections
    (simhand_w_model.py:35-94) for flags {none, crop, rotate, crop+rotate}."""
    cls = ref_import.step_class("HandCLR_W")
    b, hw = 8, (224, 224)
    batch = step.synthetic_batch(b, size=8, seed=11)  # images only provide .size()[-2:]
    batch["transformed_image1"] = torch.zeros(b, 3, *hw)
    batch["transformed_image2"] = torch.zeros(b, 3, *hw)
    g = torch.Generator().manual_seed(12)
    head_out = torch.randn(2 * b, 128, generator=g) * 3.0
    upstream = torch.randn(2 * b, 128, generator=g)
    out = {"head_out": _np(head_out), "upstream": _np(upstream), "image_hw": np.array(hw)}
    for k in ("angle_1", "angle_2", "jitter_x_1", "jitter_x_2", "jitter_y_1", "jitter_y_2"):
        out[k] = _np(batch[k])
    cfg = ref_import.EasyDict(resnet_size="18", projection_head_input_dim=512, projection_head_hidden_dim=512,
                              output_dim=128, augmentation=[])
    torch.manual_seed(0)
    with ref_import.quiet():
        model = cls(cfg, None, "train")
    model.projection_head = torch.nn.Identity()
    for name, aug in (("none", []), ("crop", ["crop"]), ("rotate", ["rotate"]), ("crop_rotate", ["crop", "rotate"])):
        model.config.augmentation = aug
        h = head_out.clone().requires_grad_(True)
        model.encoder = _Preset(h)
        model.train_metrics = {}
        p1, p2 = model.get_transformed_projections(batch)
        z = torch.cat((p1, p2), dim=0)
        (z * upstream).sum().backward()
        jx = torch.cat((batch["jitter_x_1"], batch["jitter_x_2"])) if "crop" in aug else None
        jy = torch.cat((batch["jitter_y_1"], batch["jitter_y_2"])) if "crop" in aug else None
        ang = torch.cat((batch["angle_1"], batch["angle_2"])) if "rotate" in aug else None
        h2 = head_out.clone().requires_grad_(True)
        zo = step.transformed_projections(h2, jx, jy, ang, hw)
        (zo * upstream).sum().backward()
        assert torch.allclose(z, zo, rtol=1e-5, atol=1e-6), name
        assert torch.allclose(h.grad, h2.grad, rtol=1e-4, atol=1e-6), name
        out[f"z.{name}"] = _np(z)
        out[f"dhead.{name}"] = _np(h.grad)
        if name == "none":
            stats_o = {**step.projection_stats(head_out.view(2 * b, -1, 2)[:b], "proj1"),
                       **step.projection_stats(head_out.view(2 * b, -1, 2)[b:], "proj2")}
            for k, v in model.train_metrics.items():
                assert torch.allclose(v, stats_o[k], rtol=1e-6, atol=1e-7), k
                out[f"stat.{k}"] = _np(v)
    np.savez_compressed(os.path.join(OUT, "postprocess.npz"), **out)


def golden_step() -> None:
    """a1/a2 end-to-end: the reference's HandCLR_W / PeCLR_W / SimCLR
    training_step on top of the restated RN18 (torchvision shim), seeded init,
    synthetic batch B=4 at 64x64.  Stored: loss, z, the returned metric keys,
    and gradient summaries; weights are reproduced on the other side by
    seeding torch identically (StepOracle builds its modules in the same
    order), checked by a parameter checksum."""
    res = {}
    cases = [
        ("HandCLR_W", "simhand_w", dict(weight_type="linear", diff_type="mpjpe", pos_neg="pos_neg")),
        ("PeCLR_W", "peclr_w", dict(weight_type="non_linear", diff_type="w_abs", pos_neg="neg")),
        ("SimCLR", "simclr", dict()),
    ]
    b, size, seed = 4, 64, 5
    batch = step.synthetic_batch(b, size=size, seed=seed)
    arrays = {}
    for cname, exp, wcfg in cases:
        aug = ["color_jitter", "crop", "random_crop", "resize", "rotate"]
        cfg = ref_import.EasyDict(resnet_size="18", projection_head_input_dim=512, projection_head_hidden_dim=512,
                                  output_dim=128, augmentation=aug, joints_type="augmented", use_pca=False,
                                  non_linear_lambda_pos=5.0, non_linear_lambda_neg=0.05, **wcfg)
        torch.manual_seed(seed)
        with ref_import.quiet():
            model = ref_import.step_class(cname)(cfg, None, "train")
            model.train()
            metrics = model.training_step({k: v.clone() for k, v in batch.items()}, 0)
            metrics["loss"].backward()
        torch.manual_seed(seed)
        orc = step.StepOracle(exp, "18", aug, lambda_pos=5.0, lambda_neg=0.05, **wcfg)
        orc.train()
        sd_ref = model.state_dict()
        sd_orc = orc.state_dict()
        assert list(sd_ref.keys()) == list(sd_orc.keys()), "state_dict key order differs"
        for k in sd_ref:
            assert torch.equal(sd_ref[k], sd_orc[k]) or "running" in k or "tracked" in k, k
        lo = orc.contrastive_step(batch)
        lo.backward()
        assert abs(lo.item() - metrics["loss"].item()) < 1e-5 * abs(lo.item()), (cname, lo.item(), metrics["loss"].item())
        gref = dict(model.named_parameters())
        for k, p in orc.named_parameters():
            if p.grad is None:
                assert gref[k].grad is None, k
                continue
            # biases feeding a train-mode BN have an analytically zero gradient
            # (round-off noise only): compare those absolutely
            err = (p.grad - gref[k].grad).abs().max()
            denom = gref[k].grad.abs().max()
            assert err < 2e-3 * denom or err < 1e-6, (cname, k, float(err), float(denom))
        res[cname] = {
            "experiment": exp, "config": dict(wcfg), "augmentation": aug, "B": b, "size": size, "seed": seed,
            "loss": float(metrics["loss"]),
            "logged": sorted(model.logged.keys()),
            "metric_keys": sorted(metrics.keys()),
            "plot_params_keys": sorted(model.plot_params.keys()),
            "state_dict_keys": list(sd_ref.keys()),
            "param_checksum": float(sum(p.double().abs().sum() for p in model.parameters())),
            "grad_norms": {k: float(p.grad.norm()) for k, p in model.named_parameters() if p.grad is not None},
            "no_grad": [k for k, p in model.named_parameters() if p.grad is None],
            "metrics": {k: float(v) for k, v in metrics.items()},
        }
        arrays[f"{cname}.dW_head3"] = _np(gref["projection_head.3.weight"].grad)
        arrays[f"{cname}.dW_stem"] = _np(gref["encoder.features.0.weight"].grad)
    np.savez_compressed(os.path.join(OUT, "step_rn18.npz"), **arrays)
    with open(os.path.join(OUT, "step_rn18.json"), "w") as f:
        json.dump(res, f, indent=1, sort_keys=True)


def golden_cli() -> None:
    """main.py CLI surface: parsed namespace + merged train/model params for the
    three README command lines (README.md:53-125), through the reference's
    get_general_args / update_train_params / update_model_params / prepare_name
    (src/experiments/utils.py:30,345,404,725).  The shipped imports of
    experiments/utils.py are broken, so the missing module / class
    names are aliased before import."""
    import importlib
    import types

    ref_import.install()
    sw = importlib.import_module("src.models.unsupervised.simhand_w_model")
    sw.SiMHand_W = sw.HandCLR_W
    sv = importlib.import_module("src.models.unsupervised.simhand_vis_model")
    sv.SiMHand_VIS = sv.HandCLR_VIS
    v0 = types.ModuleType("src.models.unsupervised.simhand_v0_model")
    v0.SiMHand = importlib.import_module("src.models.unsupervised.simhand_model").SiMHand
    sys.modules["src.models.unsupervised.simhand_v0_model"] = v0
    eu = importlib.import_module("src.experiments.utils")
    from src.utils import read_json
    from src.constants import TRAINING_CONFIG_PATH, PECLR_CONFIG, SIMCLR_CONFIG

    common = ("--gpus 0,1,2,3,4,5,6,7 --color_jitter --random_crop --rotate --crop -resnet_size 50 --resize "
              "-sources ego4d --datasets_scale 1m -epochs 100 -batch_size 8192 -accumulate_grad_batches 1 "
              "-save_top_k 100 -save_period 1 -num_workers 24 --weight_type linear --joints_type augmented "
              "--diff_type mpjpe --pos_neg pos_neg").split()
    simclr_w = ("--gpus 0,1,2,3,4,5,6,7 --color_jitter --crop -resnet_size 50 -sources ego4d --datasets_scale 1m "
                "--resize -epochs 100 -batch_size 1024 -accumulate_grad_batches 1 -save_top_k 100 -save_period 1 "
                "-num_workers 4 --weight_type linear --joints_type augmented --diff_type mpjpe --pos_neg pos_neg").split()
    cmds = {
        "handclr_w": ["--experiment_type", "handclr_w"] + common,
        "peclr_w": ["--experiment_type", "peclr_w"] + common,
        "simclr_w": ["--experiment_type", "simclr_w"] + simclr_w,
    }
    res = {}
    for name, argv in cmds.items():
        old = sys.argv
        sys.argv = ["main.py"] + argv
        try:
            args = eu.get_general_args("golden")
        finally:
            sys.argv = old
        train_param = ref_import.EasyDict(read_json(TRAINING_CONFIG_PATH))
        train_param = eu.update_train_params(args, train_param)
        model_param = ref_import.EasyDict(read_json(SIMCLR_CONFIG if "simclr" in name else PECLR_CONFIG))
        model_param = eu.update_model_params(model_param, args, 1000000, train_param)
        model_param.augmentation = [k for k, v in train_param.augmentation_flags.items() if v]
        res[name] = {
            "argv": argv,
            "args": {k: v for k, v in vars(args).items()},
            "train_param": json.loads(json.dumps(train_param)),
            "model_param": json.loads(json.dumps(model_param)),
            "experiment_name": eu.prepare_name(f"{args.experiment_type}_", train_param, hybrid_naming=False),
        }
    reg = {}
    for key in ("simclr", "peclr", "simhand-base", "simhand", "simhand_w", "simclr_w", "peclr_w", "simhand_vis", "handclr_w"):
        import contextlib, io
        with contextlib.redirect_stdout(io.StringIO()):
            cls = eu.get_model(key, False, False)
        reg[key] = None if cls is None else cls.__name__
    res["get_model"] = reg
    with open(os.path.join(OUT, "cli.json"), "w") as f:
        json.dump(res, f, indent=1, sort_keys=True)


def golden_sharded() -> None:
    """Row a13: what the RCCL path must equal.  The reference's HandCLR_W applied SHARD BY SHARD
    (same weights; train-mode BatchNorm statistics per shard, as in its DataParallel replicas), the shards' projections
    concatenated in the reference row order cat(all view-1, all view-2), then the reference's own
    get_weights_linear + vanila_weights_contrastive_loss over the GLOBAL batch -- for R in {1, 2, 4, 8} at
    B_glob = 32 (64 x 64 images, restated RN18 through the torchvision shim, seed 5)."""
    mu = ref_import.models_utils()
    aug = ["color_jitter", "crop", "random_crop", "resize", "rotate"]
    wcfg = dict(weight_type="linear", diff_type="mpjpe", pos_neg="pos_neg")
    b, size, seed = 32, 64, 5
    batch = step.synthetic_batch(b, size=size, seed=seed)
    cfg = ref_import.EasyDict(resnet_size="18", projection_head_input_dim=512, projection_head_hidden_dim=512, output_dim=128,
                              augmentation=aug, joints_type="augmented", use_pca=False, non_linear_lambda_pos=5.0,
                              non_linear_lambda_neg=0.05, **wcfg)
    res, arrays = {"B": b, "size": size, "seed": seed, "config": wcfg, "augmentation": aug, "ranks": {}}, {}
    j1, j2 = batch["joints1_aug"][:, :, :2], batch["joints2_aug"][:, :, :2]
    for ranks in (1, 2, 4, 8):
        torch.manual_seed(seed)
        with ref_import.quiet():
            model = ref_import.step_class("HandCLR_W")(cfg, None, "train")
            model.train()
            bl = b // ranks
            zs1, zs2 = [], []
            for r in range(ranks):
                sub = {k: v[r * bl:(r + 1) * bl].clone() for k, v in batch.items()}
                p1, p2 = model.get_transformed_projections(sub)
                zs1.append(p1)
                zs2.append(p2)
            z1, z2 = torch.cat(zs1), torch.cat(zs2)
            wp, wn = mu.get_weights_linear(j1, j2, "mpjpe")
            loss = mu.vanila_weights_contrastive_loss(z1, z2, wp, wn)
            loss.backward()
        # the restatement, sharded the same way, must agree before the numbers are committed
        torch.manual_seed(seed)
        orc = step.StepOracle("simhand_w", "18", aug, **wcfg).train()
        lo, _ = step.sharded_step(orc, batch, ranks)
        lo.backward()
        assert abs(lo.item() - loss.item()) < 1e-5 * abs(loss.item()), (ranks, lo.item(), loss.item())
        gref = dict(model.named_parameters())
        for k, p in orc.named_parameters():
            if p.grad is None:
                continue
            err, den = (p.grad - gref[k].grad).abs().max(), gref[k].grad.abs().max()
            assert err < 5e-3 * den or err < 1e-6, (ranks, k, float(err), float(den))
        res["ranks"][str(ranks)] = {
            "loss": float(loss.detach()),
            "param_checksum": float(sum(p.double().abs().sum() for p in model.parameters())),
            "grad_norms": {k: float(p.grad.norm()) for k, p in model.named_parameters() if p.grad is not None},
        }
        arrays[f"R{ranks}.z"] = _np(torch.cat((z1, z2)))
        arrays[f"R{ranks}.dW_head3"] = _np(gref["projection_head.3.weight"].grad)
    np.savez_compressed(os.path.join(OUT, "sharded_rn18.npz"), **arrays)
    with open(os.path.join(OUT, "sharded_rn18.json"), "w") as f:
        json.dump(res, f, indent=1, sort_keys=True)


def golden_augment() -> None:
    """Row 8f-2 (batch producer): the part of the reference's augmenter that does NOT go through OpenCV, executed as the
    reference's own code -- SampleAugmenter.get_crop_size (sample_augmenter.py:424-474) and crop_sample's joint / jitter
    bookkeeping (:173-195) for given jitters and crop margins, incl. boxes clamped at the image origin."""
    import importlib

    from oracle import augment

    ref_import.install()
    with ref_import.quiet():
        sa = importlib.import_module("src.data_loader.sample_augmenter")
    cfg = json.load(open(os.path.join(ref_import.REFERENCE_ROOT, "src", "experiments", "config", "training_config.json")))
    flags = ref_import.EasyDict(cfg["augmentation_flags"])
    params = ref_import.EasyDict(cfg["augmentation_params"])
    aug = sa.SampleAugmenter(flags, params)
    g = torch.Generator().manual_seed(77)
    cases = []
    for i in range(24):
        centre = torch.rand(2, generator=g) * 160 + (10 if i % 4 == 0 else 40)   # some hands near the top-left corner
        j = torch.cat((centre + torch.randn(21, 2, generator=g) * (8 + 3 * (i % 5)), torch.ones(21, 1)), dim=1)
        jitter = [int(torch.randint(0, 20, (1,), generator=g)), int(torch.randint(0, 20, (1,), generator=g))]
        margin = float(0.9 + 0.6 * torch.rand(1, generator=g))
        ox, oy, side = aug.get_crop_size(j.clone(), jitter, margin)
        want = {"origin_x": int(ox), "origin_y": int(oy), "side": int(side), "jitter_x": int(aug.jitter_x), "jitter_y": int(aug.jitter_y)}
        got = augment.crop_box(j[:, :2].numpy(), jitter, margin)
        assert all(got[k] == v for k, v in want.items()), (i, got, want)
        img = np.zeros((224, 224, 3), dtype=np.uint8)
        crop, jc, shift = aug.crop_sample(img, j.clone(), jitter)  # random_crop flag: margin drawn inside -> only shapes / shift are kept
        cases.append({"joints": j.numpy().tolist(), "jitter": jitter, "crop_margin": margin, **want})
        # rotation centre of rotate_sample (:256-259): get_crop_size(joints, [0, 0], 0.0)
        ox0, oy0, s0 = aug.get_crop_size(j.clone(), [0, 0], 0.0)
        cases[-1]["rot_center"] = [int(ox0 + s0 / 2), int(oy0 + s0 / 2)]
    # get_random_cut_out_box (:352-388) as the reference's own code: its first random.uniform draw is the box ratio, the other two are
    # degenerate (uniform(a, a)); the draw is patched to a recorded value so the bounds are a pure function of the inputs
    cut = []
    real_uniform = sa.random.uniform
    for i in range(16):
        dim0, dim1 = (224, 224) if i % 3 else (160, 200)
        c0 = float(torch.rand(1, generator=g) * (dim0 + 40) - 20)   # some centres outside the frame: the bounds are clipped
        c1 = float(torch.rand(1, generator=g) * (dim1 + 40) - 20)
        ratio = float(torch.rand(1, generator=g) * 0.16)
        box = {"n": 0}

        def fake(a, b, ratio=ratio, box=box):
            box["n"] += 1
            return ratio if box["n"] == 1 else real_uniform(a, b)

        sa.random.uniform = fake
        try:
            b0, b1 = aug.get_random_cut_out_box(dim0, dim1, c0, c1)
        finally:
            sa.random.uniform = real_uniform
        got = augment.cut_out_box(dim0, dim1, c0, c1, ratio)
        want = ([int(b0[0]), int(b0[1])], [int(b1[0]), int(b1[1])])
        assert [list(got[0]), list(got[1])] == [want[0], want[1]], (i, got, want)
        cut.append({"dim0": dim0, "dim1": dim1, "center0": c0, "center1": c1, "ratio": ratio, "bounds0": want[0], "bounds1": want[1]})
    with open(os.path.join(OUT, "augment_crop.json"), "w") as f:
        json.dump({"cases": cases, "cut_out": cut, "cut_out_fraction": list(params.cut_out_fraction), "noise_std": params.noise_std,
                   "sobel_kernel": params.sobel_kernel, "resize_shape": list(params.resize_shape), "crop_box_jitter": list(params.crop_box_jitter),
                   "crop_margin_range": list(params.crop_margin_range), "angle_range": [params.min_angle, params.max_angle],
                   "hue_factor_range": list(params.hue_factor_range), "sat_factor_range": list(params.sat_factor_range),
                   "value_factor_alpha_range": list(params.value_factor_alpha_range),
                   "value_factor_beta_range": list(params.value_factor_beta_range)}, f, indent=1, sort_keys=True)


def main() -> None:
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    if len(sys.argv) > 1 and sys.argv[1] == "sharded":  # only the a13 fixture
        golden_sharded()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "augment":  # only the 8f-2 fixture
        golden_augment()
        return
    mu = ref_import.models_utils()
    golden_loss(mu)
    golden_pca(mu)
    golden_postprocess()
    golden_step()
    golden_sharded()
    golden_augment()
    golden_cli()
    print("golden fixtures written to", OUT)
    for f in sorted(os.listdir(OUT)):
        print(f"  {f}: {os.path.getsize(os.path.join(OUT, f))} bytes")


if __name__ == "__main__":
    main()
