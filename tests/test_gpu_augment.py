"""GPU: the batch producer (row 8f-2) -- simhand_augment_batch through the C ABI against oracle/augment.py (numpy restatement
of the reference's chain; its crop-box logic is pinned to the reference's own code, its OpenCV pieces to the published
definitions), the reference golden for the crop box, the batch schema of SURVEY Appendix B, and the real entry point
running on produced batches.
Tolerances: joints / integer records exact (fp32 expressions mirrored operation by operation); images within one 8-bit level
on >= 99.5 % of the pixels and 3 levels everywhere (float summation order at the rounding steps of a 4-stage uint8 chain)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import augment as oaug

pytestmark = pytest.mark.gpu
DEV = "cuda"
LEVEL = 1.0 / 255.0 / 0.224


def _raw(n, size, seed):
    g = np.random.default_rng(seed)
    lo = g.integers(0, 256, (n, 9, 9, 3)).astype(np.float32)
    img = torch.nn.functional.interpolate(torch.from_numpy(lo).permute(0, 3, 1, 2), size=(size, size), mode="bilinear").permute(0, 2, 3, 1)
    img = (img + torch.from_numpy(g.normal(0, 10, img.shape).astype(np.float32))).clamp(0, 255).to(torch.uint8).contiguous()
    centre = g.uniform(0.3 * size, 0.7 * size, (n, 1, 2))
    j = np.concatenate([centre + g.normal(0, 0.08 * size, (n, 21, 2)), np.ones((n, 21, 1))], axis=2).astype(np.float32)
    return img, torch.from_numpy(j)


@pytest.mark.parametrize("size,out,rotate,color", [(224, 128, True, True), (160, 128, True, False), (96, 128, False, True), (224, 64, True, True),
                                                    (720, 64, True, True), (512, 32, False, False)])  # last two: area footprints of 7-16 taps per axis
def test_augment_batch_against_oracle(size, out, rotate, color):
    from simhand_amd import ops

    n = 6
    img, j = _raw(n, size, 5 + size)
    g = np.random.default_rng(size)
    angle = np.floor(g.uniform(-45, 45, n)).astype(np.float32)
    margin = g.uniform(0.9, 1.5, n).astype(np.float32)
    jitter = g.integers(0, 15, (n, 2)).astype(np.int32)
    hsab = np.stack([g.uniform(0.01, 1, n), g.uniform(0.01, 1, n), g.uniform(0.5, 1, n), g.uniform(5, 20, n)], axis=1).astype(np.float32)
    got_img, got_j, rec = ops.augment_batch(img.to(DEV), j.to(DEV), torch.from_numpy(angle).to(DEV) if rotate else None, torch.from_numpy(margin).to(DEV),
                                            torch.from_numpy(jitter).to(DEV), torch.from_numpy(hsab).to(DEV) if color else None, out_hw=(out, out))
    got_img, got_j, rec = got_img.cpu().numpy(), got_j.cpu().numpy(), rec.cpu().numpy()
    for i in range(n):
        params = {"angle": float(angle[i]), "crop_margin": float(margin[i]), "jitter": (int(jitter[i, 0]), int(jitter[i, 1])),
                  "h": float(hsab[i, 0]), "s": float(hsab[i, 1]), "a": float(hsab[i, 2]), "b": float(hsab[i, 3])}
        want_img, want_j, r = oaug.transform_sample(img[i].numpy(), j[i].numpy(), params, (out, out), rotate=rotate, do_color=color)
        assert rec[i, 0] == r["jitter_x"] and rec[i, 1] == r["jitter_y"] and rec[i, 2] == r["origin_x"] and rec[i, 3] == r["origin_y"], (i, rec[i], r)
        np.testing.assert_allclose(got_j[i], want_j, rtol=1e-6, atol=1e-4)
        diff = np.abs(got_img[i] - want_img) / LEVEL
        assert (diff <= 1.01).mean() >= 0.995 and diff.max() <= 3.01, (i, float((diff > 1.01).mean()), float(diff.max()))
        assert (diff <= 0.01).mean() >= 0.97, float((diff <= 0.01).mean())  # and the vast majority bit-identical


def test_augment_empty_crop_is_loud():
    """A crop box entirely off the canvas: the reference's cv2.resize raises; the kernel emits a NaN image + a zero-size record and
    GpuAugmenter(check=True) raises."""
    from simhand_amd import ops
    from simhand_amd.host import config as C
    from simhand_amd.host.config import edict, read_json
    from simhand_amd.host.data import GpuAugmenter

    img, j = _raw(2, 96, 3)
    j[1, :, :2] += 500.0  # joints (and so the crop box) far outside the 96 x 96 canvas
    margin = torch.full((2,), 1.2)
    jitter = torch.zeros(2, 2, dtype=torch.int32)
    out, _, rec = ops.augment_batch(img.to(DEV), j.to(DEV), None, margin.to(DEV), jitter.to(DEV), None, out_hw=(32, 32))
    assert torch.isfinite(out[0]).all() and torch.isnan(out[1]).all()
    assert rec[1, 4].item() == 0 and rec[1, 5].item() == 0
    tp = edict(read_json(C.TRAINING_CONFIG_PATH))
    tp.augmentation_flags["resize"] = True
    aug = GpuAugmenter(tp.augmentation_flags, tp.augmentation_params, check=True)
    with pytest.raises(ValueError):
        aug.transform(img.to(DEV), j.to(DEV), aug.draw(2, torch.device(DEV)))


def test_augment_crop_box_against_reference_golden(golden_dir):
    """The kernel's crop origin / jitter bookkeeping against SampleAugmenter.get_crop_size run as the reference's own code."""
    from simhand_amd import ops

    meta = json.load(open(os.path.join(golden_dir, "augment_crop.json")))
    cases = meta["cases"]
    n = len(cases)
    j = torch.tensor([c["joints"] for c in cases], dtype=torch.float32, device=DEV)
    margin = torch.tensor([c["crop_margin"] for c in cases], dtype=torch.float32, device=DEV)
    jitter = torch.tensor([c["jitter"] for c in cases], dtype=torch.int32, device=DEV)
    img = torch.zeros(n, 224, 224, 3, dtype=torch.uint8, device=DEV)
    _, ja, rec = ops.augment_batch(img, j, None, margin, jitter, None)
    rec = rec.cpu().numpy()
    for i, c in enumerate(cases):
        assert [int(v) for v in rec[i, :4]] == [c["jitter_x"], c["jitter_y"], c["origin_x"], c["origin_y"]], (i, rec[i], c)
        assert rec[i, 4] == min(c["origin_x"] + c["side"], 224) - c["origin_x"]


def test_batch_schema_and_training_on_produced_batches(tmp_path):
    """GpuAugmenter.make_batch emits the collated schema of SURVEY Appendix B (keys, dtypes, value ranges); the entry point
    trains on batches produced that way (--synthetic_raw)."""
    from simhand_amd.host import config as C
    from simhand_amd.host.config import edict, read_json
    from simhand_amd.host.data import GpuAugmenter, SyntheticRawPairs
    from simhand_amd.host.main import main

    tp = edict(read_json(C.TRAINING_CONFIG_PATH))
    for k in ("rotate", "crop", "random_crop", "resize", "color_jitter"):
        tp.augmentation_flags[k] = True
    aug = GpuAugmenter(tp.augmentation_flags, tp.augmentation_params)
    src = SyntheticRawPairs(aug, 64, 16, 0, 1, 5, torch.device(DEV))
    batch = next(iter(src))
    b = 16
    want = {"transformed_image1": (torch.float32, (b, 3, 128, 128)), "joints1_aug": (torch.float32, (b, 21, 3)), "joints1_ori": (torch.float32, (b, 21, 3)),
            "angle_1": (torch.float64, (b,)), "jitter_x_1": (torch.int64, (b,)), "jitter_y_1": (torch.int64, (b,)), "h_1": (torch.float64, (b,)),
            "s_1": (torch.float64, (b,)), "a_1": (torch.float64, (b,)), "b_1": (torch.float64, (b,)), "blur_flag_1": (torch.bool, (b,)),
            "crop_margin_scale_1": (torch.float64, (b,))}
    for k, (dt, shape) in want.items():
        for kk in (k, k.replace("1", "2") if k[-1] != "1" or "_1" not in k else k[:-1] + "2"):
            assert kk in batch, kk
            assert batch[kk].dtype == dt and tuple(batch[kk].shape) == shape, (kk, batch[kk].dtype, batch[kk].shape)
    assert (batch["jitter_x_1"] <= 0).all() and (batch["angle_1"] == batch["angle_1"].floor()).all() and batch["angle_1"].abs().max() <= 45
    assert float(batch["joints1_aug"][:, :, :2].mean()) > 20 and float(batch["joints1_aug"][:, :, :2].mean()) < 110  # hands inside the 128-px crop
    assert torch.isfinite(batch["transformed_image1"]).all() and float(batch["transformed_image1"].std()) > 0.3
    t = main(["--experiment_type", "handclr_w", "--color_jitter", "--random_crop", "--rotate", "--crop", "--resize", "-resnet_size", "18",
              "-sources", "ego4d", "--datasets_scale", "1m", "-epochs", "1", "-batch_size", "8", "-save_top_k", "1", "--weight_type", "linear",
              "--joints_type", "augmented", "--diff_type", "mpjpe", "--pos_neg", "pos_neg", "--synthetic_raw", "--synthetic_samples", "24",
              "--precision", "bf16", "--max_steps", "3", "--out_dir", str(tmp_path)])
    losses = [float(x) for x in t.step_losses]
    assert len(losses) == 3 and all(l == l and l > 0 for l in losses), losses
