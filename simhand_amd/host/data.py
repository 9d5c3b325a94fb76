"""GPU batch producer ("next" row 8f-2): raw frames + 2.5-D joints -> the collated batch dict of SURVEY Appendix B, with
the reference's augmentation chain run on the device for the whole batch (``ops.augment_batch``) instead of in 24 CPU
dataloader workers.

Mirrors, for the contrastive recipes (README :53-125: --rotate --crop --random_crop --resize --color_jitter):
  * ``SampleAugmenter`` draws -- src/data_loader/sample_augmenter.py:388-423 (crop margin, colour factors, angle `// 1`),
    :455-461 (crop-box jitter), with the parameter ranges of ``training_config.json`` (``augmentation_params``);
  * ``Data_Set.prepare_simhand_w_sample`` / ``prepare_experiment4_pretraining`` -- src/data_loader/data_set.py:646-691: which
    entries a sample carries (`joints{1,2}_aug`, `joints{1,2}_ori` only with --resize, `angle_*` only when rotation ran,
    `jitter_*` always -- with --crop off the crop still runs with jitter [0, 0]);
  * the default collate: floats -> float64 tensors, ints -> int64, bools -> bool.
The random number STREAM is the device generator's, not Python's ``random`` (the reference's draws are unseeded per worker
anyway); the kernels are deterministic functions of the draws.  The dataset readers (frames from disk) stay out of scope:
``SyntheticRawPairs`` stands in for them with raw uint8 frames and plausible hand joints."""
from __future__ import annotations

from typing import Dict, Iterator, Optional

import torch

from .. import ops
from . import dist as shdist


class GpuAugmenter:
    def __init__(self, augmentation_flags, augmentation_params, check: bool = False):
        """check: read the crop records back after every batch (one host sync) and raise ValueError for an empty crop -- the
        reference's cv2.resize raises there (sample_augmenter.py:197-224); without it such a sample's image is NaN."""
        f, p = augmentation_flags, augmentation_params
        self.check = check
        self.rotate, self.crop, self.random_crop = bool(f["rotate"]), bool(f["crop"]), bool(f["random_crop"])
        self.resize, self.color_jitter = bool(f["resize"]), bool(f["color_jitter"])
        # the coin-flip operations (sample_augmenter.py:138-171, :254-272, :302-388): each enabled flag fires with probability 1/2 per
        # sample (random.getrandbits(1)).  --flip is on the reference's CLI, but SampleAugmenter neither reads nor implements it: accepted, no-op.
        self.sobel, self.cut_out, self.blur = bool(f.get("sobel_filter", False)), bool(f.get("cut_out", False)), bool(f.get("gaussian_blur", False))
        self.noise, self.color_drop = bool(f.get("gaussian_noise", False)), bool(f.get("color_drop", False))
        self.cut_frac = tuple(p.get("cut_out_fraction", (0.0, 0.16)))
        self.noise_std = float(p.get("noise_std", 25))
        if self.sobel and int(p.get("sobel_kernel", 3)) != 3:
            raise NotImplementedError("sobel_kernel != 3 (training_config.json ships 3)")
        if not self.resize:
            raise NotImplementedError("a batch needs one image size: the recipes always pass --resize (the reference cannot collate otherwise)")
        # set_augmenation_params swaps min / max angle (sample_augmenter.py:484-485, harmless for uniform draws: App. D #7)
        self.angle_lo, self.angle_hi = float(min(p["min_angle"], p["max_angle"])), float(max(p["min_angle"], p["max_angle"]))
        self.margin_range, self.margin = tuple(p["crop_margin_range"]), float(p["crop_margin"])
        self.jitter_hi = float(p["crop_box_jitter"][1])
        self.hue, self.sat = tuple(p["hue_factor_range"]), tuple(p["sat_factor_range"])
        self.alpha, self.beta = tuple(p["value_factor_alpha_range"]), tuple(p["value_factor_beta_range"])
        self.resize_shape = tuple(int(v) for v in p["resize_shape"])  # (width, height)

    def draw(self, n: int, device, generator: Optional[torch.Generator] = None) -> Dict[str, torch.Tensor]:
        u = lambda lo, hi: lo + (hi - lo) * torch.rand(n, generator=generator, device=device)  # noqa: E731
        d = {}
        if self.rotate:
            d["angle"] = torch.floor(u(self.angle_lo, self.angle_hi))  # random.uniform(a, b) // 1
        d["crop_margin"] = u(*self.margin_range) if self.random_crop else torch.full((n,), self.margin, device=device)
        if self.crop:
            d["jitter"] = torch.stack((u(0.0, self.jitter_hi), u(0.0, self.jitter_hi)), dim=1).to(torch.int32)  # int() truncation
        else:
            d["jitter"] = torch.zeros(n, 2, dtype=torch.int32, device=device)  # override_jitter = [0, 0] (data_set.py:651-656)
        if self.color_jitter:
            d["hsab"] = torch.stack((u(*self.hue), u(*self.sat), u(*self.alpha), u(*self.beta)), dim=1)
        if self.sobel or self.cut_out or self.blur or self.noise or self.color_drop:
            coin = lambda on: (torch.rand(n, generator=generator, device=device) < 0.5) if on else torch.zeros(n, dtype=torch.bool, device=device)  # noqa: E731
            bits = [coin(self.sobel), coin(self.cut_out), coin(self.blur), coin(self.noise), coin(self.color_drop)]
            d["flags"] = sum(b.to(torch.int32) << i for i, b in enumerate(bits))
            if self.cut_out:  # np.random.randint(0, 20) joint, uniform ratio, np.uint8(np.random.randint(0, 255)) fill
                d["cut_joint"] = torch.randint(0, 20, (n,), generator=generator, device=device)
                d["cut_ratio"] = u(*self.cut_frac)
                d["cut_fill"] = torch.randint(0, 255, (n,), generator=generator, device=device).to(torch.uint8)
            if self.blur:
                d["blur_sigma"] = u(0.1, 2.0)
            if self.noise:
                d["noise"] = torch.randn(n, self.resize_shape[1], self.resize_shape[0], 3, generator=generator, device=device)
        return d

    @staticmethod
    def cut_out_boxes(joints: torch.Tensor, joint_idx: torch.Tensor, ratio: torch.Tensor, h: int, w: int) -> torch.Tensor:
        """get_random_cut_out_box (sample_augmenter.py:352-388) for a batch: [n][4] = rows [r0, r1), columns [c0, c1).  Quirk kept: the
        chosen joint's X positions the box along the rows, its Y along the columns (cut_out_sample :339-345)."""
        k = joint_idx.view(-1, 1, 1).expand(-1, 1, 2)
        c = torch.gather(joints[:, :, :2].double(), 1, k).squeeze(1)          # (x, y) of the chosen joint
        cut0, cut1 = torch.trunc(h * ratio.double()), torch.trunc(w * ratio.double())
        t0, t1 = torch.trunc(c[:, 0] - cut0 / 2), torch.trunc(c[:, 1] - cut1 / 2)
        box = torch.stack((t0.clamp(0, h), (t0 + cut0).clamp(0, h), t1.clamp(0, w), (t1 + cut1).clamp(0, w)), dim=1)
        return box.to(torch.int32).contiguous()

    def transform(self, images_u8: torch.Tensor, joints: torch.Tensor, draws: Dict[str, torch.Tensor]):
        """One view of a batch: (images fp32 (n,3,H,W) normalised, joints_aug (n,21,3), per-sample entries as collated)."""
        extra = None
        if "flags" in draws:
            h, w = images_u8.shape[1:3]
            k = [int(v * 0.1) for v in (h, w)]  # gaussian_blur_sample :316-321 (ksize = odd(0.1 * rows), odd(0.1 * cols), in that order)
            extra = {"flags": draws["flags"].contiguous(), "blur_k": tuple(v + 1 if v % 2 == 0 else v for v in k), "noise_std": self.noise_std,
                     "any_sobel": self.sobel, "any_cut_out": self.cut_out, "any_blur": self.blur, "any_noise": self.noise}
            if self.cut_out:
                extra["cut_box"] = self.cut_out_boxes(joints.float(), draws["cut_joint"], draws["cut_ratio"], h, w)
                extra["cut_fill"] = draws["cut_fill"].contiguous()
            if self.blur:
                extra["blur_sigma"] = draws["blur_sigma"].float().contiguous()
            if self.noise:
                extra["noise"] = draws["noise"].contiguous()
        img, ja, rec = ops.augment_batch(images_u8.contiguous(), joints.contiguous().float(), draws.get("angle"), draws["crop_margin"].float().contiguous(),
                                         draws["jitter"].contiguous(), draws.get("hsab"), out_hw=(self.resize_shape[1], self.resize_shape[0]), extra=extra)
        if self.check and bool((rec[:, 4:6] <= 0).any()):
            raise ValueError("augment_batch: empty crop (the crop box of a sample lies outside its frame)")
        ent = {"jitter_x": rec[:, 0].to(torch.int64), "jitter_y": rec[:, 1].to(torch.int64),
               "crop_margin_scale": draws["crop_margin"].to(torch.float64),
               "blur_flag": ((draws["flags"] >> 2) & 1).bool() if "flags" in draws else torch.zeros(img.shape[0], dtype=torch.bool, device=img.device)}
        if "angle" in draws:
            ent["angle"] = draws["angle"].to(torch.float64)
        if "hsab" in draws:
            for i, k in enumerate("hsab"):
                ent[k] = draws["hsab"][:, i].to(torch.float64)
        return img, ja, ent

    def make_batch(self, images1_u8, joints1, images2_u8, joints2, joints1_raw=None, joints2_raw=None, generator=None) -> Dict[str, torch.Tensor]:
        """Anchor / positive raw frames -> the batch dict of Appendix B (prepare_simhand_w_sample, data_set.py:646-691)."""
        n, dev = images1_u8.shape[0], images1_u8.device
        d1, d2 = self.draw(n, dev, generator), self.draw(n, dev, generator)
        img1, ja1, e1 = self.transform(images1_u8, joints1, d1)
        img2, ja2, e2 = self.transform(images2_u8, joints2, d2)
        batch = {"transformed_image1": img1, "transformed_image2": img2, "joints1_aug": ja1, "joints2_aug": ja2}
        if joints1_raw is not None:  # normalised raw joints scaled by resize_shape[1] (x) and [0] (y), data_set.py:658-665
            scale = torch.tensor([self.resize_shape[1], self.resize_shape[0], 1.0], device=dev)
            batch["joints1_ori"], batch["joints2_ori"] = joints1_raw * scale, joints2_raw * scale
        batch.update({f"{k}_1": v for k, v in e1.items()})
        batch.update({f"{k}_2": v for k, v in e2.items()})
        return batch


class SyntheticRawPairs:
    """Stand-in for the dataset readers (src/data_loader/*_loader.py, out of scope): raw 224 x 224 uint8 frames (the Ego4D
    crop size, ego4d_loader.py:21) with a textured blob around a plausible 21-joint hand, the positive a perturbed copy of the
    anchor's joints (similar hands) -- fed through ``GpuAugmenter`` every step."""

    def __init__(self, augmenter: GpuAugmenter, samples: int, global_batch: int, rank: int, world: int, seed: int, device, raw_size: int = 224):
        _, self.b_loc = shdist.shard_pairs(global_batch, rank, world)
        self.steps = max(1, samples // global_batch)
        self.aug, self.seed, self.rank, self.device, self.raw = augmenter, seed, rank, device, raw_size

    def __len__(self) -> int:
        return self.steps

    def raw_batch(self, g: torch.Generator):
        b, s, dev = self.b_loc, self.raw, self.device
        centre = 0.3 * s + 0.4 * s * torch.rand(b, 1, 2, generator=g, device=dev)
        j1 = torch.cat((centre + 0.08 * s * torch.randn(b, 21, 2, generator=g, device=dev), torch.ones(b, 21, 1, device=dev)), dim=2)
        j2 = j1.clone()
        j2[:, :, :2] += 0.03 * s * torch.randn(b, 21, 2, generator=g, device=dev)
        lo = torch.randint(0, 256, (2, b, 7, 7, 3), generator=g, device=dev, dtype=torch.int32).float()
        frames = torch.nn.functional.interpolate(lo.view(2 * b, 7, 7, 3).permute(0, 3, 1, 2), size=(s, s), mode="bilinear", align_corners=False)
        frames = (frames + 12.0 * torch.randn(frames.shape, generator=g, device=dev)).clamp(0, 255).to(torch.uint8).permute(0, 2, 3, 1).contiguous()
        return frames[:b], j1, frames[b:], j2

    def __iter__(self) -> Iterator[Dict[str, torch.Tensor]]:
        g = torch.Generator(device=self.device).manual_seed(self.seed * 1000 + self.rank)
        for _ in range(self.steps):
            f1, j1, f2, j2 = self.raw_batch(g)
            raw1, raw2 = j1 / float(self.raw), j2 / float(self.raw)
            yield self.aug.make_batch(f1, j1, f2, j2, raw1, raw2, generator=g)
