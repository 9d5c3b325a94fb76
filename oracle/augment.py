"""CPU restatement (numpy) of the reference's per-sample augmentation chain for the contrastive pre-training recipes --
TEST INFRASTRUCTURE ONLY (checker for simhand_amd's GPU batch producer, "next" row 8f-2).

Follows, in the reference's order (src/data_loader/sample_augmenter.py:50-136 `transform_sample`, flags of the README
recipes: rotate, crop + random_crop, resize, color_jitter):

  rotate_sample   :241-269  rotation about the integer joint centroid (get_crop_size with jitter [0,0], margin 0.0), cv2.warpAffine
                            (bilinear, constant-0 border, same canvas), joints @ rot_mat.T
  get_crop_size   :424-474  integer crop box from the joints (int() truncations, max(., 0) clamps, jitter_x / jitter_y)
  crop_sample     :173-195  numpy slicing (clipped at the canvas), joints - origin
  resize_sample   :197-224  cv2.resize(INTER_AREA) to resize_shape, joints * factor
  color_jitter    :292-318  BGR -> HSV (8-bit), hue*h, sat*s, value*a+b, clip, astype(uint8), HSV -> BGR
  sobel_filter    :138-156  (first, coin flip) BGR2GRAY, Sobel dx + Sobel dy (CV_64F, ksize 3), assigned into the uint8 image
  cut_out         :326-388  (coin flip) rectangle around a random joint filled with one random value (box: get_random_cut_out_box)
  gaussian_blur   :302-324  (coin flip) kernel = odd(0.1 * image dims), sigma ~ U(0.1, 2), cv2.GaussianBlur
  gaussian_noise  :158-171  (after colour jitter, coin flip) image += cv2.randn(uint8 zeros, 0, noise_std)
  color_drop      :254-272  (last, coin flip) all channels = BGR2GRAY
  flip                  --  the CLI has --flip (experiments/utils.py) but SampleAugmenter neither reads nor implements it: no-op
  transform             --  ToTensor + Normalize((0.485,0.456,0.406),(0.229,0.224,0.225))  src/data_loader/utils.py:279-285
  batch entries         --  src/data_loader/data_set.py:646-691, :804-838 (angle, jitter_x, jitter_y, h, s, a, b, crop_margin_scale)

PINNING.  OpenCV (cv2 4.x, un-vendored, not installed here) performs the resampling and colour conversions; its sources are
not under /root/reference.  What is pinned against the reference's OWN code (executed through oracle/ref_import.py, cv2
stubbed): `get_crop_size` and the crop / jitter bookkeeping (tests/golden/augment_crop.json).  What is restated from
OpenCV's published definitions and therefore PARITY UNPINNED: getRotationMatrix2D (closed form), warpAffine (ideal bilinear
in float instead of OpenCV's 5-bit fixed-point coordinates), INTER_AREA (exact area weights; for up-scaling the reference
falls into OpenCV's linear branch, restated here as plain bilinear), the 8-bit HSV conversions (float formulas with
round-half-up instead of OpenCV's LUT fixed point), BGR2GRAY (15-bit fixed point, coefficients 3735 / 19235 / 9798), Sobel
(3x3, BORDER_REFLECT_101; the float64 result assigned into a uint8 array wraps modulo 256 -- numpy's C cast), GaussianBlur (float
separable kernel exp(-x^2 / 2 sigma^2) normalised, REFLECT_101, one rounding at the end instead of OpenCV's 8.8 fixed-point
kernel), cv2.randn on uint8 (saturate_cast: negative draws become 0) added with uint8 wrap-around.
What of the new operations IS pinned to the reference's own code: get_random_cut_out_box (tests/golden/augment_crop.json,
"cut_out" cases -- executed with random.uniform patched to the recorded ratio).
Hand-derived known answers for every OpenCV-dependent piece (small integer images): tests/golden/augment_hand.json."""
from __future__ import annotations

import math
from typing import Dict, Tuple

import numpy as np

MEAN = np.array([0.485, 0.456, 0.406], dtype=np.float32)
STD = np.array([0.229, 0.224, 0.225], dtype=np.float32)


def crop_box(joints_xy: np.ndarray, jitter: Tuple[int, int], crop_margin: float) -> Dict[str, int]:
    """sample_augmenter.py:424-474.  joints_xy (21,2) float32 (x, y)."""
    x, y = joints_xy[:, 0].astype(np.float32), joints_xy[:, 1].astype(np.float32)
    center_y, center_x = int(np.float32(y.mean())), int(np.float32(x.mean()))
    r2 = ((y - np.float32(center_y)) ** 2 + (x - np.float32(center_x)) ** 2).max()
    # torch: (float32 scalar tensor) ** 0.5 * python float -> float32 arithmetic, then int() truncation
    side = int(np.float32(np.sqrt(np.float32(r2))) * np.float32(crop_margin))
    origin_x = max(center_x - side + int(jitter[0]), 0)
    origin_y = max(center_y - side + int(jitter[1]), 0)
    return {"origin_x": origin_x, "origin_y": origin_y, "side": int(2 * side), "jitter_x": center_x - side - origin_x,
            "jitter_y": center_y - side - origin_y, "center_x": center_x, "center_y": center_y}


def rotation_matrix(center: Tuple[int, int], angle_deg: float) -> np.ndarray:
    """cv2.getRotationMatrix2D(center, angle, 1.0) (published closed form; positive angle = counter-clockwise in image
    coordinates with the origin at the top-left)."""
    a = math.radians(angle_deg)
    al, be = math.cos(a), math.sin(a)
    cx, cy = center
    return np.array([[al, be, (1 - al) * cx - be * cy], [-be, al, be * cx + (1 - al) * cy]], dtype=np.float64)


def _round_u8(v: np.ndarray) -> np.ndarray:
    return np.clip(np.floor(v + 0.5), 0, 255)


def warp_affine(img: np.ndarray, m: np.ndarray) -> np.ndarray:
    """dst(x, y) = bilinear(src, M^-1 (x, y, 1)), constant-0 border, same size; uint8 in / out."""
    h, w = img.shape[:2]
    a = np.vstack([m, [0, 0, 1]])
    inv = np.linalg.inv(a)
    ys, xs = np.mgrid[0:h, 0:w].astype(np.float64)
    sx = inv[0, 0] * xs + inv[0, 1] * ys + inv[0, 2]
    sy = inv[1, 0] * xs + inv[1, 1] * ys + inv[1, 2]
    x0, y0 = np.floor(sx).astype(np.int64), np.floor(sy).astype(np.int64)
    fx, fy = (sx - x0).astype(np.float32), (sy - y0).astype(np.float32)
    src = img.astype(np.float32)

    def tap(yy, xx):
        ok = (yy >= 0) & (yy < h) & (xx >= 0) & (xx < w)
        v = src[np.clip(yy, 0, h - 1), np.clip(xx, 0, w - 1)]
        return v * ok[..., None]

    out = (tap(y0, x0) * ((1 - fx) * (1 - fy))[..., None] + tap(y0, x0 + 1) * (fx * (1 - fy))[..., None]
           + tap(y0 + 1, x0) * ((1 - fx) * fy)[..., None] + tap(y0 + 1, x0 + 1) * (fx * fy)[..., None])
    return _round_u8(out).astype(np.uint8)


def resize_area(img: np.ndarray, out_w: int, out_h: int) -> np.ndarray:
    """INTER_AREA: every destination pixel averages the source area it covers (fractional border weights); up-scaling axes
    use bilinear with half-pixel centres."""
    h, w = img.shape[:2]

    def weights(n_in, n_out):
        scale = n_in / n_out
        wm = np.zeros((n_out, n_in), dtype=np.float64)
        for o in range(n_out):
            if scale >= 1.0:
                lo, hi = o * scale, (o + 1) * scale
                for i in range(int(math.floor(lo)), min(int(math.ceil(hi)), n_in)):
                    wm[o, i] = max(0.0, min(hi, i + 1) - max(lo, i)) / scale
            else:
                c = (o + 0.5) * scale - 0.5
                i0 = int(math.floor(c))
                f = c - i0
                wm[o, min(max(i0, 0), n_in - 1)] += 1 - f
                wm[o, min(max(i0 + 1, 0), n_in - 1)] += f
        return wm

    wy, wx = weights(h, out_h), weights(w, out_w)
    out = np.einsum("oh,hwc->owc", wy, img.astype(np.float64))
    out = np.einsum("pw,owc->opc", wx, out)
    return _round_u8(out).astype(np.uint8)


def bgr_to_hsv_u8(img: np.ndarray):
    b, g, r = [img[..., i].astype(np.float32) for i in range(3)]
    v = np.maximum(np.maximum(b, g), r)
    mn = np.minimum(np.minimum(b, g), r)
    diff = v - mn
    s = np.where(v > 0, diff * 255.0 / np.maximum(v, 1), 0.0)
    d = np.maximum(diff, 1e-12)
    hdeg = np.where(v == r, 60.0 * (g - b) / d, np.where(v == g, 120.0 + 60.0 * (b - r) / d, 240.0 + 60.0 * (r - g) / d))
    hdeg = np.where(diff == 0, 0.0, hdeg)
    hdeg = np.where(hdeg < 0, hdeg + 360.0, hdeg)
    return _round_u8(hdeg / 2.0), _round_u8(s), v


def hsv_to_bgr_u8(hh, ss, vv) -> np.ndarray:
    h = hh.astype(np.float32) * 2.0 / 60.0  # sector coordinate
    s = ss.astype(np.float32) / 255.0
    v = vv.astype(np.float32) / 255.0
    h = np.where(h >= 6.0, h - 6.0, h)
    i = np.floor(h)
    f = h - i
    p, q, t = v * (1 - s), v * (1 - s * f), v * (1 - s * (1 - f))
    i = i.astype(np.int64) % 6
    r = np.choose(i, [v, q, p, p, t, v])
    g = np.choose(i, [t, v, v, q, p, p])
    b = np.choose(i, [p, p, t, v, v, q])
    return np.stack([_round_u8(b * 255.0), _round_u8(g * 255.0), _round_u8(r * 255.0)], axis=-1).astype(np.uint8)


def color_jitter(img: np.ndarray, h: float, s: float, a: float, b: float) -> np.ndarray:
    hue, sat, val = bgr_to_hsv_u8(img)
    hue = np.floor(np.clip(hue * np.float32(h), 0, 255))          # .astype(np.uint8) truncates
    sat = np.floor(np.clip(sat * np.float32(s), 0, 255))
    val = np.floor(np.clip(val * np.float32(a) + np.float32(b), 0, 255))
    return hsv_to_bgr_u8(hue, sat, val)


def gray_u8(img: np.ndarray) -> np.ndarray:
    """cv2.cvtColor(BGR2GRAY) on uint8: 15-bit fixed point (B 0.114, G 0.587, R 0.299)."""
    b, g, r = [img[..., i].astype(np.int64) for i in range(3)]
    return ((b * 3735 + g * 19235 + r * 9798 + (1 << 14)) >> 15).astype(np.uint8)


def _reflect101(i: np.ndarray, n: int) -> np.ndarray:
    if n == 1:
        return np.zeros_like(i)
    p = 2 * (n - 1)
    i = np.mod(i, p)
    return np.where(i >= n, p - i, i)


def sobel_sample(img: np.ndarray) -> np.ndarray:
    """sobel_filter_sample :138-156: gray -> Sobel x + Sobel y (3x3, float64) -> written into the uint8 image (wraps mod 256)."""
    g = gray_u8(img).astype(np.float64)
    h, w = g.shape
    ys, xs = np.arange(h), np.arange(w)
    gp = lambda dy, dx: g[_reflect101(ys + dy, h)][:, _reflect101(xs + dx, w)]  # noqa: E731
    sx = (gp(-1, 1) + 2 * gp(0, 1) + gp(1, 1)) - (gp(-1, -1) + 2 * gp(0, -1) + gp(1, -1))
    sy = (gp(1, -1) + 2 * gp(1, 0) + gp(1, 1)) - (gp(-1, -1) + 2 * gp(-1, 0) + gp(-1, 1))
    v = np.trunc(sx + sy).astype(np.int64) & 255
    return np.repeat(v.astype(np.uint8)[..., None], 3, axis=2)


def cut_out_box(dim0: int, dim1: int, center0: float, center1: float, ratio: float):
    """get_random_cut_out_box :352-388 (its random.uniform(a, a) draws are degenerate: the box is centred)."""
    c0, c1 = int(dim0 * ratio), int(dim1 * ratio)
    t0, t1 = int(center0 - c0 / 2), int(center1 - c1 / 2)
    cl = lambda v, hi: int(min(max(v, 0), hi))  # noqa: E731
    return (cl(t0, dim0), cl(t0 + c0, dim0)), (cl(t1, dim1), cl(t1 + c1, dim1))


def cut_out_sample(img: np.ndarray, joints: np.ndarray, joint_idx: int, ratio: float, fill: int) -> np.ndarray:
    """cut_out_sample :326-350.  Quirk kept: the joint's X coordinate positions the box along image dim 0 (rows) and its Y along
    dim 1 (the call passes joints[k, 0], joints[k, 1] as hand_center_dim0 / dim1)."""
    (r0, r1), (c0, c1) = cut_out_box(img.shape[0], img.shape[1], float(joints[joint_idx, 0]), float(joints[joint_idx, 1]), ratio)
    out = img.copy()
    out[r0:r1, c0:c1] = np.uint8(fill)
    return out


def blur_kernel_sizes(shape) -> Tuple[int, int]:
    """gaussian_blur_sample :316-321: (ksize.width, ksize.height) = odd(0.1 * rows), odd(0.1 * cols) -- the reference hands the
    tuple built from image.shape[:2] to cv2 as (width, height): swapped for non-square frames, kept."""
    k = [int(v * 0.1) for v in shape[:2]]
    k = [v + 1 if v % 2 == 0 else v for v in k]
    return k[0], k[1]


def gaussian_blur(img: np.ndarray, kx: int, ky: int, sigma: float) -> np.ndarray:
    def kern(n):
        x = np.arange(n, dtype=np.float64) - (n - 1) / 2.0
        k = np.exp(-(x * x) / (2.0 * sigma * sigma))
        return (k / k.sum()).astype(np.float32)

    h, w = img.shape[:2]
    a = img.astype(np.float32)
    kxv, kyv = kern(kx), kern(ky)
    tmp = np.zeros_like(a)
    xs, ys = np.arange(w), np.arange(h)
    for t in range(kx):
        tmp += kxv[t] * a[:, _reflect101(xs + t - kx // 2, w)]
    out = np.zeros_like(a)
    for t in range(ky):
        out += kyv[t] * tmp[_reflect101(ys + t - ky // 2, h)]
    return _round_u8(out).astype(np.uint8)


def gaussian_noise(img: np.ndarray, z: np.ndarray, std: float) -> np.ndarray:
    """image += cv2.randn(uint8 zeros, 0, std): the draw is saturate-cast to uint8 (negative -> 0), the sum wraps (numpy uint8)."""
    n8 = np.clip(np.rint(z.astype(np.float32) * np.float32(std)), 0, 255).astype(np.int64)
    return ((img.astype(np.int64) + n8) & 255).astype(np.uint8)


def color_drop(img: np.ndarray) -> np.ndarray:
    return np.repeat(gray_u8(img)[..., None], 3, axis=2)


def transform_sample(image: np.ndarray, joints: np.ndarray, params: Dict[str, float], resize_shape=(128, 128), rotate=True,
                     do_color=True):
    """image uint8 (H,W,3); joints (21,3) float32 [x, y, depth]; params: angle, crop_margin, jitter (jx, jy), h, s, a, b.
    Returns (normalised CHW float32 image, joints_aug (21,3), record of the batch entries)."""
    img = image.copy()
    j = joints.astype(np.float32).copy()
    rec = {}
    # the three coin-flip operations that run on the raw frame (params carry the outcome of the flips and their draws)
    if params.get("sobel"):
        img = sobel_sample(img)
    if params.get("cut_out") is not None:
        k, ratio, fill = params["cut_out"]
        img = cut_out_sample(img, j, int(k), float(ratio), int(fill))
    if params.get("blur_sigma") is not None:
        kx, ky = blur_kernel_sizes(img.shape)
        img = gaussian_blur(img, kx, ky, float(params["blur_sigma"]))
    rec["blur_flag"] = params.get("blur_sigma") is not None
    if rotate:
        cb = crop_box(j[:, :2], (0, 0), 0.0)
        center = (int(cb["origin_x"] + cb["side"] / 2), int(cb["origin_y"] + cb["side"] / 2))
        m = rotation_matrix(center, params["angle"])
        img = warp_affine(img, m)
        hom = np.concatenate([j[:, :2], np.ones((21, 1), np.float32)], axis=1).astype(np.float64)
        j[:, :2] = (hom @ m.T).astype(np.float32)
        rec["angle"] = float(params["angle"])
    cb = crop_box(j[:, :2], params["jitter"], params["crop_margin"])
    j[:, 0] -= cb["origin_x"]
    j[:, 1] -= cb["origin_y"]
    img = img[cb["origin_y"]:cb["origin_y"] + cb["side"], cb["origin_x"]:cb["origin_x"] + cb["side"], :]
    rec.update(jitter_x=cb["jitter_x"], jitter_y=cb["jitter_y"], crop_margin_scale=float(params["crop_margin"]),
               origin_x=cb["origin_x"], origin_y=cb["origin_y"], side=cb["side"])
    hc, wc = img.shape[:2]
    img = resize_area(img, resize_shape[0], resize_shape[1])
    j[:, 0] *= np.float32(resize_shape[0] / wc)
    j[:, 1] *= np.float32(resize_shape[1] / hc)
    if do_color:
        img = color_jitter(img, params["h"], params["s"], params["a"], params["b"])
        rec.update(h=params["h"], s=params["s"], a=params["a"], b=params["b"])
    if params.get("noise") is not None:
        img = gaussian_noise(img, np.asarray(params["noise"]), float(params.get("noise_std", 25)))
    if params.get("color_drop"):
        img = color_drop(img)
    t = (img.astype(np.float32) / 255.0 - MEAN) / STD
    return np.ascontiguousarray(t.transpose(2, 0, 1)), j, rec
