// GPU batch producer ("next" row 8f-2): the reference's per-sample augmentation chain of the contrastive pre-training
// recipes as two kernels over a whole batch -- replaces the 24 CPU dataloader workers (README :71) that would otherwise
// cap the step at a few thousand pairs/s.
//
// Replaces (reference): SampleAugmenter.transform_sample (src/data_loader/sample_augmenter.py:50-136) with the flags of
// the README recipes -- rotate_sample (:241-269, cv2.getRotationMatrix2D + cv2.warpAffine about the integer joint
// centroid), get_crop_size / crop_sample (:424-474, :173-195), resize_sample (:197-224, cv2.INTER_AREA), color_jitter_sample
// (:292-318, 8-bit HSV) -- followed by ToTensor + Normalize (src/data_loader/utils.py:279-285) and the bookkeeping of
// Data_Set.get_random_augment_param (src/data_loader/data_set.py:804-838).  The random DRAWS (angle, crop margin, jitter,
// h/s/a/b) are inputs: the host draws them (device generator), the kernels are deterministic.
// OpenCV is not vendored by the reference: its resampling / colour arithmetic is restated from the published definitions
// (see oracle/augment.py for what is pinned and what is not); every stage rounds to uint8 where OpenCV does.
//
// The coin-flip operations of the same chain (simhand_augment_batch_ex; the flips and their draws are inputs as well):
// sobel_filter_sample (:138-156), cut_out_sample (:326-388) and gaussian_blur_sample (:302-324) run on the RAW frame before the
// rotation -- three small pre-pass kernels write a processed uint8 copy of the frames that the chain above then reads;
// gaussian_noise_sample (:158-171) and color_drop_sample (:254-272) are pointwise tails of kernel 2.  --flip exists on the CLI but
// SampleAugmenter neither reads nor implements it: nothing to build.
//
// Kernel 1 (one thread per sample): rotation centre and matrix, rotated joints, crop box, final joints, the record.
// Kernel 2 (one thread per output pixel): area-average over the crop footprint of bilinear samples of the rotated source,
// HSV jitter, normalisation, written as CHW fp32.
#include "common.h"

// every float expression below mirrors oracle/augment.py operation by operation: no fused multiply-adds
#pragma STDC FP_CONTRACT OFF

namespace sh {

struct AugGeo {       // per sample, written by kernel 1
  double inv[6];      // source = inv * (x, y, 1) of the rotated canvas (identity when rotation is off)
  int ox, oy, wc, hc; // crop origin and (canvas-clipped) size
  int rotate;
  int pad;
};

__global__ void augment_geometry_kernel(const float* __restrict__ joints, const float* __restrict__ angle, const float* __restrict__ margin,
                                        const int* __restrict__ jitter, int n, int H, int W, int out_w, int out_h,
                                        float* __restrict__ joints_aug, int* __restrict__ rec, AugGeo* __restrict__ geo) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float jx[21], jy[21];
#pragma unroll
  for (int k = 0; k < 21; ++k) {
    jx[k] = joints[((long long)i * 21 + k) * 3 + 0];
    jy[k] = joints[((long long)i * 21 + k) * 3 + 1];
  }
  // get_crop_size (sample_augmenter.py:424-474): int() truncations, float32 distance, max(., 0) clamps
  auto box = [&](int jit_x, int jit_y, float mg, int& ox, int& oy, int& side2, int& rjx, int& rjy) {
    double sx = 0.0, sy = 0.0;
    for (int k = 0; k < 21; ++k) {
      sx += jx[k];
      sy += jy[k];
    }
    const int cx = (int)(float)(sx / 21.0), cy = (int)(float)(sy / 21.0);
    float r2 = 0.f;
    for (int k = 0; k < 21; ++k) {
      const float dy = jy[k] - (float)cy, dx = jx[k] - (float)cx;
      r2 = fmaxf(r2, __fadd_rn(__fmul_rn(dy, dy), __fmul_rn(dx, dx)));
    }
    const int side = (int)__fmul_rn(__fsqrt_rn(r2), mg);
    ox = max(cx - side + jit_x, 0);
    oy = max(cy - side + jit_y, 0);
    side2 = 2 * side;
    rjx = cx - side - ox;
    rjy = cy - side - oy;
  };
  AugGeo g;
  g.inv[0] = 1; g.inv[1] = 0; g.inv[2] = 0; g.inv[3] = 0; g.inv[4] = 1; g.inv[5] = 0;
  g.rotate = angle != nullptr ? 1 : 0;
  g.pad = 0;
  if (angle != nullptr) {
    int ox, oy, s2, a, b;
    box(0, 0, 0.0f, ox, oy, s2, a, b);  // rotate_sample :256-259: jitter [0, 0], crop_margin 0.0 -> the integer centroid
    const double cx = (double)(int)(ox + s2 / 2.0), cy = (double)(int)(oy + s2 / 2.0);
    const double th = (double)angle[i] * 3.14159265358979323846 / 180.0;
    const double al = cos(th), be = sin(th);
    // cv2.getRotationMatrix2D: M = [[al, be, (1-al) cx - be cy], [-be, al, be cx + (1-al) cy]]
    const double tx = (1.0 - al) * cx - be * cy, ty = be * cx + (1.0 - al) * cy;
    for (int k = 0; k < 21; ++k) {
      const double x = jx[k], y = jy[k];
      jx[k] = (float)(al * x + be * y + tx);
      jy[k] = (float)(-be * x + al * y + ty);
    }
    // inverse of [R | t]: R^T, -R^T t
    g.inv[0] = al; g.inv[1] = -be; g.inv[2] = -(al * tx - be * ty);
    g.inv[3] = be; g.inv[4] = al;  g.inv[5] = -(be * tx + al * ty);
  }
  int ox, oy, s2, rjx, rjy;
  box(jitter[2 * i], jitter[2 * i + 1], margin[i], ox, oy, s2, rjx, rjy);
  // numpy slicing image[oy : oy + side, ox : ox + side] clips at the canvas
  const int wc = max(min(ox + s2, W) - ox, 0), hc = max(min(oy + s2, H) - oy, 0);
  g.ox = ox; g.oy = oy; g.wc = wc; g.hc = hc;
  geo[i] = g;
  const float fx = wc > 0 ? (float)((double)out_w / (double)wc) : 1.f, fy = hc > 0 ? (float)((double)out_h / (double)hc) : 1.f;
  for (int k = 0; k < 21; ++k) {
    float* o = joints_aug + ((long long)i * 21 + k) * 3;
    o[0] = __fmul_rn(jx[k] - (float)ox, fx);
    o[1] = __fmul_rn(jy[k] - (float)oy, fy);
    o[2] = joints[((long long)i * 21 + k) * 3 + 2];
  }
  int* r = rec + (long long)i * 6;
  r[0] = rjx; r[1] = rjy; r[2] = ox; r[3] = oy; r[4] = wc; r[5] = hc;
}

__device__ __forceinline__ float round_u8(float v) { return fminf(fmaxf(floorf(v + 0.5f), 0.f), 255.f); }

// rotated canvas pixel (X, Y), channel triple, as uint8 values in float
__device__ __forceinline__ void canvas_pixel(const unsigned char* __restrict__ img, int H, int W, const AugGeo& g, int X, int Y, float (&o)[3]) {
  if (!g.rotate) {
    const unsigned char* p = img + ((long long)Y * W + X) * 3;
    o[0] = p[0]; o[1] = p[1]; o[2] = p[2];
    return;
  }
  const double sx = g.inv[0] * X + g.inv[1] * Y + g.inv[2], sy = g.inv[3] * X + g.inv[4] * Y + g.inv[5];
  const double fx0 = floor(sx), fy0 = floor(sy);
  const int x0 = (int)fx0, y0 = (int)fy0;
  const float fx = (float)(sx - fx0), fy = (float)(sy - fy0);
  const float w00 = __fmul_rn(1.f - fx, 1.f - fy), w01 = __fmul_rn(fx, 1.f - fy), w10 = __fmul_rn(1.f - fx, fy), w11 = __fmul_rn(fx, fy);
  auto tap = [&](int yy, int xx, int c) -> float {
    return ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) ? (float)img[((long long)yy * W + xx) * 3 + c] : 0.f;
  };
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float v = __fmul_rn(tap(y0, x0, c), w00);
    v = __fadd_rn(v, __fmul_rn(tap(y0, x0 + 1, c), w01));
    v = __fadd_rn(v, __fmul_rn(tap(y0 + 1, x0, c), w10));
    v = __fadd_rn(v, __fmul_rn(tap(y0 + 1, x0 + 1, c), w11));
    o[c] = round_u8(v);
  }
}

// 1-D resampling footprint of destination index o: taps i0 .. i0 + cnt - 1 (INTER_AREA for scale >= 1, else bilinear).  The
// area footprint is as long as the scale asks for (ceil(scale) + 1 taps at most): weights are evaluated on the fly by
// foot_weight, nothing is capped (a 6-tap cap used to drop taps -- and darken the image -- for crops > 5x the output side).
struct Foot {
  int i0, cnt;
  double lo, hi, scale;  // area branch: covered interval [lo, hi) in source pixels; bilinear branch: scale < 0, lo = weight of tap 0
};
__device__ __forceinline__ Foot footprint(int o, int n_in, int n_out) {
  Foot f;
  const double scale = (double)n_in / (double)n_out;
  if (scale >= 1.0) {
    f.lo = o * scale;
    f.hi = (o + 1) * scale;
    f.scale = scale;
    f.i0 = (int)floor(f.lo);
    f.cnt = min((int)ceil(f.hi), n_in) - f.i0;
  } else {
    const double c = (o + 0.5) * scale - 0.5;
    const int i0 = (int)floor(c);
    const double fr = c - i0;
    const int a = min(max(i0, 0), n_in - 1), b = min(max(i0 + 1, 0), n_in - 1);
    f.i0 = a;
    f.scale = -1.0;
    f.hi = 0.0;
    if (a == b) {
      f.cnt = 1;
      f.lo = 1.0;
    } else {
      f.cnt = 2;
      f.lo = 1.0 - fr;
    }
  }
  return f;
}
__device__ __forceinline__ double foot_weight(const Foot& f, int k) {
  if (f.scale < 0.0) return k == 0 ? f.lo : 1.0 - f.lo;
  const int i = f.i0 + k;
  return fmax(0.0, fmin(f.hi, (double)(i + 1)) - fmax(f.lo, (double)i)) / f.scale;
}

__device__ __forceinline__ int reflect101(int i, int n) {
  if (n == 1) return 0;
  const int p = 2 * (n - 1);
  i %= p;
  if (i < 0) i += p;
  return i >= n ? p - i : i;
}
// cv2.cvtColor(BGR2GRAY) on uint8: 15-bit fixed point
__device__ __forceinline__ int gray15(int b, int g, int r) { return (b * 3735 + g * 19235 + r * 9798 + (1 << 14)) >> 15; }

// pre-pass 1: Sobel (flag bit 0) then cut-out (bit 1) on the raw frame -> dst (uint8).  One thread per pixel.
__global__ __launch_bounds__(256) void augment_pre_sobel_cut_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst,
                                                                    const int* __restrict__ flags, const int* __restrict__ cut_box,
                                                                    const unsigned char* __restrict__ cut_fill, int H, int W) {
  const int n = blockIdx.y;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= H * W) return;
  const int y = idx / W, x = idx - y * W;
  const unsigned char* img = src + (long long)n * H * W * 3;
  unsigned char* out = dst + ((long long)n * H * W + idx) * 3;
  const int f = flags[n];
  int v0, v1, v2;
  if (f & 1) {
    auto gp = [&](int dy, int dx) -> double {
      const unsigned char* p = img + ((long long)reflect101(y + dy, H) * W + reflect101(x + dx, W)) * 3;
      return (double)gray15(p[0], p[1], p[2]);
    };
    const double sx = (gp(-1, 1) + 2.0 * gp(0, 1) + gp(1, 1)) - (gp(-1, -1) + 2.0 * gp(0, -1) + gp(1, -1));
    const double sy = (gp(1, -1) + 2.0 * gp(1, 0) + gp(1, 1)) - (gp(-1, -1) + 2.0 * gp(-1, 0) + gp(-1, 1));
    v0 = v1 = v2 = (int)((long long)(sx + sy) & 255);  // float64 -> uint8 array assignment: truncation, modulo 256
  } else {
    const unsigned char* p = img + (long long)idx * 3;
    v0 = p[0]; v1 = p[1]; v2 = p[2];
  }
  if (f & 2) {
    const int* b = cut_box + 4 * n;  // rows [b0, b1), columns [b2, b3)
    if (y >= b[0] && y < b[1] && x >= b[2] && x < b[3]) v0 = v1 = v2 = cut_fill[n];
  }
  out[0] = (unsigned char)v0; out[1] = (unsigned char)v1; out[2] = (unsigned char)v2;
}

// pre-pass 2 / 3: separable Gaussian blur (flag bit 2) of dst in place through a float scratch image: horizontal taps -> tmp,
// vertical taps + the one rounding -> dst.  Samples without the flag are left alone.  Weights exp(-x^2 / 2 sigma^2) / sum in fp32.
__global__ __launch_bounds__(256) void augment_blur_h_kernel(const unsigned char* __restrict__ img8, float* __restrict__ tmp,
                                                             const int* __restrict__ flags, const float* __restrict__ sigma, int H, int W, int kx) {
  const int n = blockIdx.y;
  if (!(flags[n] & 4)) return;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= H * W) return;
  const int y = idx / W, x = idx - y * W;
  const unsigned char* img = img8 + (long long)n * H * W * 3;
  // the oracle normalises in float64 and rounds the weights to fp32: k / k.sum() -> astype(float32)
  double sum = 0.0;
  const double sg = (double)sigma[n];
  for (int t = 0; t < kx; ++t) {
    const double xx = (double)t - (double)(kx - 1) / 2.0;
    sum += exp(-(xx * xx) / (2.0 * sg * sg));
  }
  float a0 = 0.f, a1 = 0.f, a2 = 0.f;
  for (int t = 0; t < kx; ++t) {
    const double xx = (double)t - (double)(kx - 1) / 2.0;
    const float wgt = (float)(exp(-(xx * xx) / (2.0 * sg * sg)) / sum);
    const unsigned char* p = img + ((long long)y * W + reflect101(x + t - kx / 2, W)) * 3;
    a0 = __fadd_rn(a0, __fmul_rn(wgt, (float)p[0]));
    a1 = __fadd_rn(a1, __fmul_rn(wgt, (float)p[1]));
    a2 = __fadd_rn(a2, __fmul_rn(wgt, (float)p[2]));
  }
  float* o = tmp + ((long long)n * H * W + idx) * 3;
  o[0] = a0; o[1] = a1; o[2] = a2;
}
__global__ __launch_bounds__(256) void augment_blur_v_kernel(const float* __restrict__ tmp, unsigned char* __restrict__ img8,
                                                             const int* __restrict__ flags, const float* __restrict__ sigma, int H, int W, int ky) {
  const int n = blockIdx.y;
  if (!(flags[n] & 4)) return;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= H * W) return;
  const int y = idx / W, x = idx - y * W;
  const float* src = tmp + (long long)n * H * W * 3;
  double sum = 0.0;
  const double sg = (double)sigma[n];
  for (int t = 0; t < ky; ++t) {
    const double xx = (double)t - (double)(ky - 1) / 2.0;
    sum += exp(-(xx * xx) / (2.0 * sg * sg));
  }
  float a0 = 0.f, a1 = 0.f, a2 = 0.f;
  for (int t = 0; t < ky; ++t) {
    const double xx = (double)t - (double)(ky - 1) / 2.0;
    const float wgt = (float)(exp(-(xx * xx) / (2.0 * sg * sg)) / sum);
    const float* p = src + ((long long)reflect101(y + t - ky / 2, H) * W + x) * 3;
    a0 = __fadd_rn(a0, __fmul_rn(wgt, p[0]));
    a1 = __fadd_rn(a1, __fmul_rn(wgt, p[1]));
    a2 = __fadd_rn(a2, __fmul_rn(wgt, p[2]));
  }
  unsigned char* o = img8 + ((long long)n * H * W + idx) * 3;
  o[0] = (unsigned char)round_u8(a0); o[1] = (unsigned char)round_u8(a1); o[2] = (unsigned char)round_u8(a2);
}

__global__ __launch_bounds__(128) void augment_image_kernel(const unsigned char* __restrict__ images, const AugGeo* __restrict__ geo,
                                                            const float* __restrict__ hsab, int H, int W, int out_w, int out_h,
                                                            float* __restrict__ out, const int* __restrict__ flags,
                                                            const float* __restrict__ noise, float noise_std) {
  const int n = blockIdx.y, v = blockIdx.x;
  const AugGeo g = geo[n];
  const unsigned char* img = images + (long long)n * H * W * 3;
  const long long plane = (long long)out_h * out_w;
  for (int u = threadIdx.x; u < out_w; u += blockDim.x) {
    float px[3] = {0.f, 0.f, 0.f};
    // an empty crop (box entirely off the canvas): cv2.resize raises in the reference; here the sample's image is NaN (loud
    // downstream) and rec[4..5] = (0, 0) lets the host raise (GpuAugmenter(check=True))
    const bool empty = g.wc <= 0 || g.hc <= 0;
    if (!empty) {
      const Foot fy = footprint(v, g.hc, out_h), fx = footprint(u, g.wc, out_w);
      double acc[3] = {0.0, 0.0, 0.0};
      for (int a = 0; a < fy.cnt; ++a) {
        const double wy = foot_weight(fy, a);
        for (int b = 0; b < fx.cnt; ++b) {
          float c3[3];
          canvas_pixel(img, H, W, g, g.ox + fx.i0 + b, g.oy + fy.i0 + a, c3);
          const double w = wy * foot_weight(fx, b);
          acc[0] += w * c3[0];
          acc[1] += w * c3[1];
          acc[2] += w * c3[2];
        }
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) px[c] = round_u8((float)acc[c]);
    }
    if (hsab != nullptr) {
      // color_jitter_sample :292-318 on 8-bit HSV; channel 0 plays "B" whatever the loader's channel order is (the readers
      // deliver RGB, the reference converts with COLOR_BGR2HSV all the same)
      const float hf = hsab[4 * n], sf = hsab[4 * n + 1], af = hsab[4 * n + 2], bf = hsab[4 * n + 3];
      const float b = px[0], gch = px[1], r = px[2];
      const float vmax = fmaxf(fmaxf(b, gch), r), vmin = fminf(fminf(b, gch), r), diff = vmax - vmin;
      float s8 = vmax > 0.f ? diff * 255.0f / fmaxf(vmax, 1.f) : 0.f;
      const float d = fmaxf(diff, 1e-12f);
      float hdeg = vmax == r ? 60.0f * (gch - b) / d : (vmax == gch ? 120.0f + 60.0f * (b - r) / d : 240.0f + 60.0f * (r - gch) / d);
      if (diff == 0.f) hdeg = 0.f;
      if (hdeg < 0.f) hdeg += 360.0f;
      float h8 = round_u8(hdeg / 2.0f);
      s8 = round_u8(s8);
      h8 = floorf(fminf(fmaxf(__fmul_rn(h8, hf), 0.f), 255.f));
      s8 = floorf(fminf(fmaxf(__fmul_rn(s8, sf), 0.f), 255.f));
      const float v8 = floorf(fminf(fmaxf(__fadd_rn(__fmul_rn(vmax, af), bf), 0.f), 255.f));
      float hh = h8 * 2.0f / 60.0f;
      const float ss = s8 / 255.0f, vv = v8 / 255.0f;
      if (hh >= 6.0f) hh -= 6.0f;
      const float fi = floorf(hh), ff = hh - fi;
      const float p = vv * (1.f - ss), q = vv * (1.f - ss * ff), t = vv * (1.f - ss * (1.f - ff));
      const int sec = ((int)fi) % 6;
      float rr, gg, bb;
      switch (sec) {
        case 0: rr = vv; gg = t; bb = p; break;
        case 1: rr = q; gg = vv; bb = p; break;
        case 2: rr = p; gg = vv; bb = t; break;
        case 3: rr = p; gg = q; bb = vv; break;
        case 4: rr = t; gg = p; bb = vv; break;
        default: rr = vv; gg = p; bb = q; break;
      }
      px[0] = round_u8(bb * 255.0f);
      px[1] = round_u8(gg * 255.0f);
      px[2] = round_u8(rr * 255.0f);
    }
    const int fl = flags != nullptr ? flags[n] : 0;
    if ((fl & 8) && noise != nullptr) {
      // image += cv2.randn(uint8 zeros, 0, std): the draw saturates to [0, 255] (negative -> 0, round half to even), the uint8 sum wraps
      const float* z = noise + (((long long)n * out_h + v) * out_w + u) * 3;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float n8 = fminf(fmaxf(rintf(__fmul_rn(z[c], noise_std)), 0.f), 255.f);
        px[c] = (float)(((int)px[c] + (int)n8) & 255);
      }
    }
    if (fl & 16) {  // color_drop_sample: all channels = BGR2GRAY
      const float gy = (float)gray15((int)px[0], (int)px[1], (int)px[2]);
      px[0] = px[1] = px[2] = gy;
    }
    const float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};
#pragma unroll
    for (int c = 0; c < 3; ++c)
      out[((long long)n * 3 + c) * plane + (long long)v * out_w + u] = empty ? __builtin_nanf("") : (px[c] / 255.0f - mean[c]) / stdv[c];
  }
}

}  // namespace sh

using namespace sh;

extern "C" {

size_t simhand_augment_workspace_bytes(int n) { return (size_t)(n > 0 ? n : 0) * sizeof(AugGeo); }

// geometry table + (with pre-pass operations) a processed uint8 copy of the frames + (with blur) its float scratch image
size_t simhand_augment_workspace_bytes_ex(int n, int h, int w, int pre_ops, int blur) {
  if (n <= 0 || h <= 0 || w <= 0) return 0;
  size_t b = ((size_t)n * sizeof(AugGeo) + 255) & ~(size_t)255;
  if (pre_ops || blur) b += (((size_t)n * h * w * 3) + 255) & ~(size_t)255;
  if (blur) b += (size_t)n * h * w * 3 * sizeof(float);
  return b;
}

static int augment_impl(const uint8_t* images, const float* joints, const float* angle, const float* crop_margin, const int32_t* jitter,
                        const float* hsab, const sh_augment_extra* ex, int n, int h, int w, int out_w, int out_h, float* out_images,
                        float* joints_aug, int32_t* rec, void* workspace, size_t workspace_bytes, sh_stream_t stream) {
  SH_REQUIRE(images && joints && crop_margin && jitter && out_images && joints_aug && rec && workspace, "augment_batch: NULL pointer");
  SH_REQUIRE(n >= 1 && h >= 1 && w >= 1 && out_w >= 1 && out_h >= 1, "augment_batch: bad shape");
  SH_REQUIRE((long long)h * w * 3 < (1ll << 31), "augment_batch: image too large");
  const bool pre = ex != nullptr && ex->flags != nullptr && (ex->any_sobel || ex->any_cut_out || ex->any_blur);
  const bool blur = pre && ex->any_blur;
  const size_t need = ex == nullptr ? simhand_augment_workspace_bytes(n) : simhand_augment_workspace_bytes_ex(n, h, w, pre ? 1 : 0, blur ? 1 : 0);
  SH_REQUIRE(workspace_bytes >= need, "augment_batch: workspace too small");
  if (ex != nullptr && ex->flags != nullptr) {
    SH_REQUIRE(!ex->any_cut_out || (ex->cut_box && ex->cut_fill), "augment_batch: cut-out needs its boxes and fill values");
    SH_REQUIRE(!ex->any_blur || (ex->blur_sigma && ex->blur_kx >= 1 && ex->blur_ky >= 1 && (ex->blur_kx & 1) && (ex->blur_ky & 1)),
               "augment_batch: blur needs sigma and odd kernel sizes");
    SH_REQUIRE(!ex->any_noise || ex->noise, "augment_batch: noise needs its standard-normal draws");
  }
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SH_PROF_MISC, s, 0, (double)n * ((double)h * w * 3 * (pre ? 3 : 1) + (double)out_w * out_h * 12));
  char* wsb = (char*)workspace;
  AugGeo* geo = (AugGeo*)wsb;
  const unsigned char* frames = images;
  if (pre) {
    unsigned char* img8 = (unsigned char*)(wsb + (((size_t)n * sizeof(AugGeo) + 255) & ~(size_t)255));
    const dim3 grid(ceil_div(h * w, 256), n);
    augment_pre_sobel_cut_kernel<<<grid, 256, 0, s>>>(images, img8, ex->flags, ex->cut_box, ex->cut_fill, h, w);
    if (check_launch("augment_batch pre")) return 1;
    if (blur) {
      float* tmp = (float*)((char*)img8 + ((((size_t)n * h * w * 3) + 255) & ~(size_t)255));
      augment_blur_h_kernel<<<grid, 256, 0, s>>>(img8, tmp, ex->flags, ex->blur_sigma, h, w, ex->blur_kx);
      if (check_launch("augment_batch blur h")) return 1;
      augment_blur_v_kernel<<<grid, 256, 0, s>>>(tmp, img8, ex->flags, ex->blur_sigma, h, w, ex->blur_ky);
      if (check_launch("augment_batch blur v")) return 1;
    }
    frames = img8;
  }
  augment_geometry_kernel<<<ceil_div(n, 64), 64, 0, s>>>(joints, angle, crop_margin, jitter, n, h, w, out_w, out_h, joints_aug, rec, geo);
  if (check_launch("augment_batch geometry")) return 1;
  const bool tails = ex != nullptr && ex->flags != nullptr;
  augment_image_kernel<<<dim3(out_h, n), 128, 0, s>>>(frames, geo, hsab, h, w, out_w, out_h, out_images, tails ? ex->flags : nullptr,
                                                      tails && ex->any_noise ? ex->noise : nullptr, tails ? ex->noise_std : 0.f);
  return check_launch("augment_batch image");
}

int simhand_augment_batch(const uint8_t* images, const float* joints, const float* angle, const float* crop_margin, const int32_t* jitter,
                          const float* hsab, int n, int h, int w, int out_w, int out_h, float* out_images, float* joints_aug, int32_t* rec,
                          void* workspace, size_t workspace_bytes, sh_stream_t stream) {
  return augment_impl(images, joints, angle, crop_margin, jitter, hsab, nullptr, n, h, w, out_w, out_h, out_images, joints_aug, rec, workspace,
                      workspace_bytes, stream);
}

int simhand_augment_batch_ex(const uint8_t* images, const float* joints, const float* angle, const float* crop_margin, const int32_t* jitter,
                             const float* hsab, const sh_augment_extra* extra, int n, int h, int w, int out_w, int out_h, float* out_images,
                             float* joints_aug, int32_t* rec, void* workspace, size_t workspace_bytes, sh_stream_t stream) {
  return augment_impl(images, joints, angle, crop_margin, jitter, hsab, extra, n, h, w, out_w, out_h, out_images, joints_aug, rec, workspace,
                      workspace_bytes, stream);
}

}  // extern "C"
