// Weight-gradient convolution for gfx950: dW[k][r][s][c] = sum over output pixels of
// dy[pix][k] * x[pix -> (ho*stride - pad + r, wo*stride - pad + s)][c], fp32 result.
//
// Replaces (reference): the weight-gradient of torch.nn.Conv2d / Linear produced by
// loss.backward() through torchvision's ResNet and the projection head
// (src/models/resnet_model.py:13-58, src/models/unsupervised/simclr_model.py:22-39).
//
// GEMM view per filter tap: M = cout, N = cin, K = output pixels (up to 25.7 M at batch
// 2048 x 112^2), so the reduction runs over the PIXEL axis while both operands are stored
// channel-contiguous (NHWC).  Tiles of [KP pixels][channels] are staged global -> VGPR -> LDS
// exactly as they lie in memory (coalesced 16-B chunks); the transposition the MFMA needs
// (k = pixel must be the per-lane contiguous axis) is done by the LDS read:
//   bf16: ds_read_b64_tr_b16 (gfx950 transpose read) - two reads give a lane 8 pixels of one
//         channel; row stride = row bytes + 32 B so the 8 rows a 32-lane group touches fall on
//         8 distinct 32-B bank slots (conflict-free);
//   fp32: plain ds_read_b32 (element = bank width), stride = row bytes + 64 B.
// The k permutation is free as long as both operands use the same one.
// Split-K over pixel ranges fills the chip; partial tiles go to a workspace and are combined
// by a second deterministic pass (no float atomics).  Blocks of one pixel range are adjacent
// in the logical block order and mapped to one XCD so dy / x tiles are shared in its L2.
#include "common.h"
#include "conv_1x1.h"

#include <stdlib.h>

namespace sh {

struct WgradArgs {
  const void* x;
  const void* dy;
  float* part;          // [splitk][Cout][taps*Cin]
  long long Mo;         // output pixels
  int Cout, Cin;
  int R, S, stride, pad;
  int Ho, Wo, H, W;
  int splitk;
  int pix_per_split;    // multiple of KP
  int mt, nt;           // cout tiles, cin tiles
  FastDiv div_hw, div_w;
  int use_tr;           // bf16: 1 = ds_read_b64_tr_b16, 0 = scalar fallback (self-test)
  float* dy_colsum;     // PLAIN kernel only: [splitk][2][Cout] -- row 0 = per-channel sums of dy over the split's pixels
                        // (emitted by the cin-tile-0 blocks from the tiles they stage anyway), row 1 = 0; null = off
  int stem_hp, stem_wp; // > 0: x is the zero-padded NHWC4 stem input [N][hp][wp][4]; Cin = 256 virtual channels
                        // = 8 filter rows x (8 taps x 4 channels), row r of output pixel (ho, wo) at (2ho + r, 2wo)
  // PLAIN kernel, XFORM != 0: the operand chunks are TRANSFORMED in registers between the global load and the LDS store,
  // and the transformed dy operand is also written out (by the cin-tile-0 blocks: every element exactly once).
  //   XFORM 1 (Gram launch, x == dy == raw conv output y of a conv + BN + ReLU unit): v = max(y * xs + xh, 0) on BOTH operands
  //           -> dw = a^T a, dy_colsum = sum a, xout = a.  The stand-alone BatchNorm-apply pass over y disappears.
  //   XFORM 2 (weight gradient of the conv whose OUTPUT feeds the BN): dy operand = BatchNorm backward of (da, y):
  //           v = xa * (da * [y * xs + xh > 0]) - xb * y + xc  -> dw = dy^T x, xout = dy.  The stand-alone
  //           BatchNorm-backward-apply pass disappears (dy2 = the raw conv output y laid out like dy).
  const float* xs = nullptr;   // [Cout] scale  (gamma * invstd)
  const float* xh = nullptr;   // [Cout] shift  (beta - mean * scale)
  const float* xa = nullptr;   // XFORM 2: [Cout] A = gamma * invstd
  const float* xb = nullptr;   // XFORM 2: [Cout] B = A * invstd * dgamma / M
  const float* xc = nullptr;   // XFORM 2: [Cout] C = -A * dbeta / M + mean * B
  const void* dy2 = nullptr;   // XFORM 2: y
  void* xout = nullptr;        // transformed dy operand [Mo][Cout]
  int xrelu = 1;               // XFORM 1 / 2: the unit has a ReLU
};

__device__ __forceinline__ int xcd_remap_w(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7;
  const int xcd = bid & 7, j = bid >> 3;
  const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + j;
}

template <typename T> struct WgCfg;
template <> struct WgCfg<bf16_t> {
  static constexpr int KP = 32;   // pixels per k-step
  static constexpr int PAD = 32;  // row padding in bytes
};
template <> struct WgCfg<float> {
  static constexpr int KP = 16;
  static constexpr int PAD = 64;
};

typedef __attribute__((ext_vector_type(4))) short s16x4;

// bf16 fragment: 8 pixels {4g..4g+3, 16+4g..16+4g+3} of channel (cb + lane&15)
__device__ __forceinline__ uint4 frag_bf16_tr(const char* tile, int stride, int cb, int lane) {
  const int p = lane & 15, g = lane >> 4;
  const char* a0 = tile + (4 * g + (p >> 2)) * stride + (cb + (p & 3) * 4) * 2;
  typedef s16x4 __attribute__((address_space(3))) * lds_ptr;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(a0));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(a0 + 16 * stride));
  uint4 r;
  r.x = (unsigned)(unsigned short)lo[0] | ((unsigned)(unsigned short)lo[1] << 16);
  r.y = (unsigned)(unsigned short)lo[2] | ((unsigned)(unsigned short)lo[3] << 16);
  r.z = (unsigned)(unsigned short)hi[0] | ((unsigned)(unsigned short)hi[1] << 16);
  r.w = (unsigned)(unsigned short)hi[2] | ((unsigned)(unsigned short)hi[3] << 16);
  return r;
}
__device__ __forceinline__ uint4 frag_bf16_scalar(const char* tile, int stride, int cb, int lane) {
  const int i = lane & 15, g = lane >> 4;
  unsigned short v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int pix = (e < 4 ? 4 * g + e : 16 + 4 * g + (e - 4));
    v[e] = *reinterpret_cast<const unsigned short*>(tile + pix * stride + (cb + i) * 2);
  }
  uint4 r;
  r.x = v[0] | ((unsigned)v[1] << 16);
  r.y = v[2] | ((unsigned)v[3] << 16);
  r.z = v[4] | ((unsigned)v[5] << 16);
  r.w = v[6] | ((unsigned)v[7] << 16);
  return r;
}

// PLAIN = 1x1 / stride 1 / pad 0 (x rows are the output pixels themselves): the loader walks two pointers instead of
// decoding a pixel and multiplying out 64-bit addresses every k-step -- the generic loop spends ~180 VALU instructions
// per 16 MFMAs on addressing and is VALU-issue-bound.
// KPM: k-step = KPM x the base 32 (bf16) / 16 (fp32) pixels.  PLAIN uses 2: one barrier (~250 cycles) per 32 MFMAs of a
// wave instead of per 16.
// SH_WGRAD_PF2 = 1: two k-steps of operand loads in flight in the 1x1 pointer-walking kernel.  Measured on MI355X (round 2,
// scripts/wg_ab.py): the 128 x 128 tile then needs > 256 VGPRs (scratch spills: 2-2.5x slower), the tiles that still fit gain
// nothing -- in isolation these launches already run at 5.5-6.1 TB/s on the 56^2 / 28^2 layers.  Kept off.
#ifndef SH_WGRAD_PF2
#define SH_WGRAD_PF2 0
#endif
template <typename T, int BM, int BN, bool STEM = false, bool PLAIN = false, int KPM = 1, int XFORM = 0>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(WgradArgs p) {
  static_assert(XFORM == 0 || (PLAIN && sizeof(T) == 2), "operand transforms exist for the bf16 1x1 pointer-walking kernel only");
  constexpr int KP = WgCfg<T>::KP * KPM;
  constexpr int SA = BM * (int)sizeof(T) + WgCfg<T>::PAD;  // dy tile row stride (bytes)
  constexpr int SB = BN * (int)sizeof(T) + WgCfg<T>::PAD;  // x tile row stride
  constexpr int VE = 16 / (int)sizeof(T);
  constexpr int CPR_A = BM * (int)sizeof(T) / 16, CPR_B = BN * (int)sizeof(T) / 16;
  constexpr int NA = KP * CPR_A / 256, NBL = KP * CPR_B / 256;  // chunks per thread
  constexpr int MI = BM / 32, NI = BN / 32;
  __shared__ __attribute__((aligned(16))) char smem[2 * KP * SA + 2 * KP * SB];
  char* sA = smem;
  char* sB = smem + 2 * KP * SA;

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int li = lane & 15, g = lane >> 4;
  const int wm = wave >> 1, wn = wave & 1;
  const int taps = p.R * p.S;
  // logical order: [split][tap][mt][nt] -> blocks of one pixel range are contiguous
  int logical = xcd_remap_w(blockIdx.x, gridDim.x);
  const int nt_i = logical % p.nt; logical /= p.nt;
  const int mt_i = logical % p.mt; logical /= p.mt;
  const int tap = logical % taps;
  const int split = logical / taps;
  const int fr = tap / p.S, fs = tap - fr * p.S;
  const int k0 = mt_i * BM, c0 = nt_i * BN;
  const long long pix_begin = (long long)split * p.pix_per_split;
  long long pix_end = pix_begin + p.pix_per_split;
  if (pix_end > p.Mo) pix_end = p.Mo;
  const int nk = pix_begin < pix_end ? (int)((pix_end - pix_begin + KP - 1) / KP) : 0;

  const T* __restrict__ xs = reinterpret_cast<const T*>(p.x);
  const T* __restrict__ dys = reinterpret_cast<const T*>(p.dy);
  const unsigned hw = (unsigned)(p.Ho * p.Wo);

  // staging registers of one k-step (a second set only with SH_WGRAD_PF2, see above)
  struct Regs {
    uint4 a[NA], b[NBL];
    uint4 y[XFORM == 2 ? NA : 1];  // XFORM 2: the raw conv output chunks next to the incoming-gradient chunks in a
    int live;                      // rows of the staged k-step that lie inside the block's pixel range
  };
  Regs r0, r1;
  r0.live = r1.live = 0;
  // PLAIN: per-thread row pointers of k-step 0 (chunk i = tile row (tid + 256 i) / CPR, 16-B chunk (tid + 256 i) % CPR)
  const char* pdy[NA];
  const char* px[NBL];
  int rowa[NA], rowb[NBL];
  // STEM walks its rows the same way: the padded-input address of output pixel (img, ho, wo) advances by 2 * KP pixels per
  // k-step, plus a constant at every row / image wrap (swo / sho track the position; no division in the loop)
  int swo[NBL], sho[NBL];
  if constexpr (PLAIN || STEM) {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int id = tid + 256 * i;
      rowa[i] = id / CPR_A;
      pdy[i] = reinterpret_cast<const char*>(dys + (pix_begin + rowa[i]) * p.Cout + k0 + (id - rowa[i] * CPR_A) * VE);
    }
#pragma unroll
    for (int i = 0; i < NBL; ++i) {
      const int id = tid + 256 * i;
      rowb[i] = id / CPR_B;
      if constexpr (STEM) {
        const long long m = pix_begin + rowb[i];
        const unsigned mu = m < p.Mo ? (unsigned)m : 0u;
        const unsigned img = fdiv(mu, p.div_hw);
        const unsigned rem = mu - img * hw;
        const unsigned ho = fdiv(rem, p.div_w);
        const unsigned wo = rem - ho * (unsigned)p.Wo;
        const int vc = c0 + (id - rowb[i] * CPR_B) * VE;  // virtual channel: filter row vc / 32, element vc % 32 of its 8 x 4 run
        const unsigned prow = (img * (unsigned)p.stem_hp + 2 * ho + (unsigned)(vc >> 5)) * (unsigned)p.stem_wp + 2 * wo;  // < 2^31
        px[i] = reinterpret_cast<const char*>(xs + (unsigned long long)prow * 4u + (vc & 31));
        swo[i] = (int)wo;
        sho[i] = (int)ho;
      } else {
        px[i] = reinterpret_cast<const char*>(xs + (pix_begin + rowb[i]) * p.Cin + c0 + (id - rowb[i] * CPR_B) * VE);
      }
    }
  }
  // PLAIN + dy_colsum: every chunk of this thread is the same 8 channels k0 + (tid % CPR_A) * 8 .. (CPR_A divides 256)
  const bool want_colsum = PLAIN && sizeof(T) == 2 && p.dy_colsum != nullptr && nt_i == 0;
  float csum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  // XFORM: this thread's chunks always cover the same 8 channels (CPR_A / CPR_B divide 256): coefficients in registers
  float ca_s[8], ca_h[8], ca_a[8], ca_b[8], ca_c[8], cb_s[8], cb_h[8];
  const char* pdy2[NA];
  char* pout[NA];
  const bool x_writes = XFORM != 0 && nt_i == 0 && p.xout != nullptr;
  const bool x_diag = XFORM == 1 && BM == BN && mt_i == nt_i;  // Gram diagonal tile: both operands are the same columns
  if constexpr (XFORM != 0) {
    const int cha = k0 + (tid % CPR_A) * VE;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      ca_s[e] = p.xs[cha + e];
      ca_h[e] = p.xh[cha + e];
      ca_a[e] = XFORM == 2 ? p.xa[cha + e] : 0.f;
      ca_b[e] = XFORM == 2 ? p.xb[cha + e] : 0.f;
      ca_c[e] = XFORM == 2 ? p.xc[cha + e] : 0.f;
    }
    if constexpr (XFORM == 1) {
      const int chb = c0 + (tid % CPR_B) * VE;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        cb_s[e] = p.xs[chb + e];
        cb_h[e] = p.xh[chb + e];
      }
    }
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const long long off = (pix_begin + rowa[i]) * (long long)p.Cout + k0 + (tid + 256 * i - rowa[i] * CPR_A) * VE;
      pdy2[i] = reinterpret_cast<const char*>(reinterpret_cast<const T*>(p.dy2) + off);
      pout[i] = reinterpret_cast<char*>(reinterpret_cast<T*>(p.xout) + off);
    }
  }
  auto bn_relu8 = [&](uint4 v, const float (&sc)[8], const float (&sh)[8]) __attribute__((always_inline)) -> uint4 {
    const unsigned w4[4] = {v.x, v.y, v.z, v.w};
    unsigned o[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float lo = h16_lo(w4[q]) * sc[2 * q] + sh[2 * q];
      float hi = h16_hi(w4[q]) * sc[2 * q + 1] + sh[2 * q + 1];
      if (p.xrelu) {
        lo = fmaxf(lo, 0.f);
        hi = fmaxf(hi, 0.f);
      }
      o[q] = pack_bf16x2(lo, hi);
    }
    return make_uint4(o[0], o[1], o[2], o[3]);
  };
  const unsigned step_dy = (unsigned)KP * (unsigned)p.Cout * (unsigned)sizeof(T);
  const unsigned step_x = (unsigned)KP * (unsigned)p.Cin * (unsigned)sizeof(T);
  const int rows_total = (int)(pix_end - pix_begin);  // <= pix_per_split
  const long long stem_row_skip = ((long long)2 * p.stem_wp - 2 * p.Wo) * 4 * (long long)sizeof(T);             // bytes, at a row wrap
  const long long stem_img_skip = ((long long)p.stem_hp - 2 * p.Ho) * p.stem_wp * 4 * (long long)sizeof(T);  // bytes, at an image wrap
  auto load_step = [&](int ks, Regs& R) __attribute__((always_inline)) {
    if constexpr (PLAIN || STEM) {
      const int left = rows_total - ks * KP;  // rows of this k-step inside the block's pixel range
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        const bool live = rowa[i] < left;
        R.a[i] = live ? *reinterpret_cast<const uint4*>(pdy[i]) : make_uint4(0, 0, 0, 0);
        pdy[i] += step_dy;
        if constexpr (XFORM == 2) {
          R.y[i] = live ? *reinterpret_cast<const uint4*>(pdy2[i]) : make_uint4(0, 0, 0, 0);
          pdy2[i] += step_dy;
        }
      }
      R.live = left;  // the transforms / column sums run in store_step, AFTER the MFMAs that cover these loads' latency
#pragma unroll
      for (int i = 0; i < NBL; ++i) {
        if constexpr (XFORM == 1 && BM == BN) {
          if (x_diag) {  // block-uniform: the Gram tile on the diagonal reuses the transformed dy chunks (store_step)
            px[i] += step_x;
            continue;
          }
        }
        R.b[i] = rowb[i] < left ? *reinterpret_cast<const uint4*>(px[i]) : make_uint4(0, 0, 0, 0);
        if constexpr (STEM) {
          px[i] += 2 * KP * 4 * (int)sizeof(T);
          swo[i] += KP;
          while (swo[i] >= p.Wo) {  // at most once for Wo >= KP (the 224^2 stem: Wo = 112)
            swo[i] -= p.Wo;
            px[i] += stem_row_skip;
            if (++sho[i] == p.Ho) {
              sho[i] = 0;
              px[i] += stem_img_skip;
            }
          }
        } else {
          px[i] += step_x;
        }
      }
      return;
    }
    // 32-bit pixel arithmetic (Mo < 2^31 is checked on the host); one 64-bit multiply-add per address
    const unsigned base = (unsigned)pix_begin + (unsigned)ks * KP, endu = (unsigned)pix_end;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int id = tid + 256 * i;
      const int row = id / CPR_A, ch = id - row * CPR_A;
      const unsigned m = base + row;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (m < endu) v = *reinterpret_cast<const uint4*>(dys + (unsigned long long)m * (unsigned)p.Cout + (k0 + ch * VE));
      R.a[i] = v;
    }
#pragma unroll
    for (int i = 0; i < NBL; ++i) {
      const int id = tid + 256 * i;
      const int row = id / CPR_B, ch = id - row * CPR_B;
      const unsigned mu = base + row;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (mu < endu) {
        const unsigned img = fdiv(mu, p.div_hw);
        const unsigned rem = mu - img * hw;
        const unsigned ho = fdiv(rem, p.div_w);
        const unsigned wo = rem - ho * (unsigned)p.Wo;
        if constexpr (STEM) {
          const int vc = c0 + ch * VE;  // virtual channel: filter row vc / 32, element vc % 32 of its 8 x 4 run
          const unsigned prow = (img * (unsigned)p.stem_hp + 2 * ho + (unsigned)(vc >> 5)) * (unsigned)p.stem_wp + 2 * wo;  // < 2^31 padded pixels
          v = *reinterpret_cast<const uint4*>(xs + (unsigned long long)prow * 4u + (vc & 31));
        } else {
          const int hs = (int)ho * p.stride - p.pad + fr;
          const int ws = (int)wo * p.stride - p.pad + fs;
          if ((unsigned)hs < (unsigned)p.H && (unsigned)ws < (unsigned)p.W) {
            const unsigned pix = (img * (unsigned)p.H + (unsigned)hs) * (unsigned)p.W + (unsigned)ws;  // < 2^31 input pixels
            v = *reinterpret_cast<const uint4*>(xs + (unsigned long long)pix * (unsigned)p.Cin + (c0 + ch * VE));
          }
        }
      }
      R.b[i] = v;
    }
  };
  auto store_step = [&](int buf, Regs& R) __attribute__((always_inline)) {
    char* dA = sA + buf * (KP * SA);
    char* dB = sB + buf * (KP * SB);
    if constexpr (XFORM != 0) {
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        const bool live = rowa[i] < R.live;
        if constexpr (XFORM == 1) {
          R.a[i] = live ? bn_relu8(R.a[i], ca_s, ca_h) : make_uint4(0, 0, 0, 0);  // rows past the range must stay zero
        } else {
          const unsigned g4[4] = {R.a[i].x, R.a[i].y, R.a[i].z, R.a[i].w}, y4[4] = {R.y[i].x, R.y[i].y, R.y[i].z, R.y[i].w};
          unsigned o[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            float r2[2];
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
              const int e = 2 * q + hh;
              const float g = hh == 0 ? h16_lo(g4[q]) : h16_hi(g4[q]);
              const float y = hh == 0 ? h16_lo(y4[q]) : h16_hi(y4[q]);
              const bool on = !p.xrelu || (y * ca_s[e] + ca_h[e] > 0.f);
              r2[hh] = ca_a[e] * (on ? g : 0.f) - ca_b[e] * y + ca_c[e];
            }
            o[q] = pack_bf16x2(r2[0], r2[1]);
          }
          R.a[i] = live ? make_uint4(o[0], o[1], o[2], o[3]) : make_uint4(0, 0, 0, 0);
        }
        if (x_writes && live) *reinterpret_cast<uint4*>(pout[i]) = R.a[i];
        pout[i] += step_dy;
      }
      if constexpr (XFORM == 1) {
#pragma unroll
        for (int i = 0; i < NBL; ++i) {
          if constexpr (BM == BN) {
            if (x_diag) {
              R.b[i] = R.a[i];
              continue;
            }
          }
          R.b[i] = rowb[i] < R.live ? bn_relu8(R.b[i], cb_s, cb_h) : make_uint4(0, 0, 0, 0);
        }
      }
    }
    if (want_colsum) {  // block-uniform; after the MFMAs so that the loads' latency is covered
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        const unsigned w4[4] = {R.a[i].x, R.a[i].y, R.a[i].z, R.a[i].w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          csum[2 * q] += h16_lo(w4[q]);
          csum[2 * q + 1] += h16_hi(w4[q]);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int id = tid + 256 * i;
      const int row = id / CPR_A, ch = id - row * CPR_A;
      *reinterpret_cast<uint4*>(dA + row * SA + ch * 16) = R.a[i];
    }
#pragma unroll
    for (int i = 0; i < NBL; ++i) {
      const int id = tid + 256 * i;
      const int row = id / CPR_B, ch = id - row * CPR_B;
      *reinterpret_cast<uint4*>(dB + row * SB + ch * 16) = R.b[i];
    }
  };

  f32x4 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};

  auto compute = [&](int buf) __attribute__((always_inline)) {
    const char* tA = sA + buf * (KP * SA);
    const char* tB = sB + buf * (KP * SB);
    if constexpr (sizeof(T) == 2) {
#pragma unroll
      for (int sub = 0; sub < KPM; ++sub) {  // 32-pixel sub-tiles of the k-step
        const char* uA = tA + sub * 32 * SA;
        const char* uB = tB + sub * 32 * SB;
        uint4 fa[MI], fb[NI];
        if (PLAIN || p.use_tr) {
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) fa[mi] = frag_bf16_tr(uA, SA, wm * (BM / 2) + mi * 16, lane);
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) fb[ni] = frag_bf16_tr(uB, SB, wn * (BN / 2) + ni * 16, lane);
        } else {
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) fa[mi] = frag_bf16_scalar(uA, SA, wm * (BM / 2) + mi * 16, lane);
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) fb[ni] = frag_bf16_scalar(uB, SB, wn * (BN / 2) + ni * 16, lane);
        }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            acc[mi][ni] = sh_mfma16(fa[mi], fb[ni], acc[mi][ni]);
      }
    } else {
#pragma unroll
      for (int j = 0; j < KP / 4; ++j) {
        float fa[MI], fb[NI];
        const int pix = 4 * j + g;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
          fa[mi] = *reinterpret_cast<const float*>(tA + pix * SA + (wm * (BM / 2) + mi * 16 + li) * 4);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
          fb[ni] = *reinterpret_cast<const float*>(tB + pix * SB + (wn * (BN / 2) + ni * 16 + li) * 4);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[mi], fb[ni], acc[mi][ni], 0, 0, 0);
      }
    }
  };
  constexpr bool PF2 = PLAIN && sizeof(T) == 2 && SH_WGRAD_PF2;
  if (nk > 0) {
    load_step(0, r0);
    store_step(0, r0);
  }
  if (PF2 && nk > 1) load_step(1, r1);
  __syncthreads();
  if constexpr (!PF2) {
    for (int ks = 0; ks < nk; ++ks) {
      const int buf = ks & 1;
      if (ks + 1 < nk) load_step(ks + 1, r0);
      compute(buf);
      if (ks + 1 < nk) store_step(buf ^ 1, r0);
      __syncthreads();
    }
  } else {
    // LDS buffer 0 / register set r0 carry the even k-steps, buffer 1 / r1 the odd ones
    for (int ks = 0; ks < nk; ks += 2) {
      if (ks + 2 < nk) load_step(ks + 2, r0);
      compute(0);
      if (ks + 1 < nk) store_step(1, r1);
      __syncthreads();
      if (ks + 1 >= nk) break;
      if (ks + 3 < nk) load_step(ks + 3, r1);
      compute(1);
      if (ks + 2 < nk) store_step(0, r0);
      __syncthreads();
    }
  }

  if (want_colsum) {
    // threads tid, tid + CPR_A, ... hold the same channel chunk: fold them through LDS in a fixed order
    float* red = reinterpret_cast<float*>(smem);  // [256 / CPR_A][CPR_A * 8]; the tiles are dead (loop ended with a barrier)
    const int chc = tid % CPR_A, rr = tid / CPR_A;
#pragma unroll
    for (int e = 0; e < 8; ++e) red[rr * (CPR_A * 8) + chc * 8 + e] = csum[e];
    __syncthreads();
    if (tid < CPR_A * 8) {
      float t = 0.f;
      for (int r2 = 0; r2 < 256 / CPR_A; ++r2) t += red[r2 * (CPR_A * 8) + tid];
      p.dy_colsum[((long long)split * 2 + 0) * p.Cout + k0 + tid] = t;
      p.dy_colsum[((long long)split * 2 + 1) * p.Cout + k0 + tid] = 0.f;
    }
  }

  // C[m = cout][n = cin]: lane holds cin = c0 + wn*BN/2 + ni*16 + li, couts k0 + wm*BM/2 + mi*16 + 4g + r
  const long long row_len = (long long)taps * p.Cin;
  float* dst = p.part + (long long)split * p.Cout * row_len;
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int k = k0 + wm * (BM / 2) + mi * 16 + 4 * g + r;
        const int c = c0 + wn * (BN / 2) + ni * 16 + li;
        dst[(long long)k * row_len + (long long)tap * p.Cin + c] = acc[mi][ni][r];
      }
}

// Second pass of the deterministic split-K: sums the `splitk` partial [Cout][R*S][Cin] matrices in a fixed order.
// SL threads share one float4 column group (split k = lane, lane+SL, ...; LDS tree in fixed order), so that short
// outputs with many splits (64x64 1x1: 1536 splits of 16 KB) still fill the chip.  `c_real > 0` writes the result in
// OIHW order [Cout][c_real][R][S] (the layout of the reference's nn.Conv2d.weight.grad), dropping padded channels.
template <int SL>
__device__ __forceinline__ void wgrad_reduce_body(const int vblock, float4* red, const float* __restrict__ part, long long count, int splitk,
                                                  float* __restrict__ dw, int cin, int rs, int c_real) {
  constexpr int COLS = 256 / SL;
  const int col = threadIdx.x % COLS, sl = threadIdx.x / COLS;
  const long long i = ((long long)vblock * COLS + col) * 4;
  float4 s = make_float4(0, 0, 0, 0);
  if (i < count)
    for (int k = sl; k < splitk; k += SL) {
      const float4 v = *reinterpret_cast<const float4*>(part + (long long)k * count + i);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  if (SL > 1) {
    red[threadIdx.x] = s;
    __syncthreads();
#pragma unroll
    for (int h = SL / 2; h >= 1; h >>= 1) {
      if (sl < h) {
        const float4 o = red[threadIdx.x + h * COLS];
        s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w;
        red[threadIdx.x] = s;
      }
      __syncthreads();
    }
    if (sl != 0) return;
  }
  if (i >= count) return;
  if (c_real == 0) {
    *reinterpret_cast<float4*>(dw + i) = s;
    return;
  }
  if (c_real < 0) {  // direct stem: column r*32 + tap*4 + c of row k -> OIHW [64][3][7][7]; padded columns dropped
    const long long k = i >> 8;
    const int col = (int)(i & 255), r = col >> 5, t = (col & 31) >> 2;
    if (r < 7 && t < 7) {
      float* o = dw + (k * 3 * 7 + r) * 7 + t;
      o[0] = s.x;
      o[49] = s.y;
      o[98] = s.z;
    }
    return;
  }
  const long long row = i / cin;  // = k * rs + tap
  const int c = (int)(i - row * cin);
  const long long k = row / rs;
  const int tap = (int)(row - k * rs);
  const float v[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (c + j < c_real) dw[(k * c_real + c + j) * rs + tap] = v[j];
}

template <int SL>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part, long long count, int splitk,
                                                           float* __restrict__ dw, int cin, int rs, int c_real) {
  __shared__ float4 red[SL > 1 ? 256 : 1];
  wgrad_reduce_body<SL>(blockIdx.x, red, part, count, splitk, dw, cin, rs, c_real);
}

// Round 6: the split-K reduction AND the fold of the per-split channel sums (dy_colsum rows of a Gram / colsum launch) in ONE launch --
// blocks [0, nred) are wgrad_reduce_kernel<SL>'s, blocks [nred, nred + ceil(c / 4)) fold column 0 of cs_part [splits][2][c] to cs_out[c]
// exactly as bn_bwd_finalize_kernel<float> does with mean = 0, invstd = 1 (64 row lanes x 4 independent double chains, the 64 lane sums added
// in lane order), four channels per 256-thread block instead of sixteen per 1024: the same additions in the same order per channel, so
// both outputs are bit-identical to the two launches this replaces (40 launches per ResNet-50 step).
template <int SL>
__global__ __launch_bounds__(256) void wgrad_finish_kernel(const float* __restrict__ part, long long count, int splitk, float* __restrict__ dw,
                                                           int cin, int rs, int c_real, int nred, const float* __restrict__ cs_part, int splits,
                                                           int c, float* __restrict__ cs_out) {
  __shared__ float4 red[SL > 1 ? 256 : 1];
  __shared__ double redc[64][4];
  if ((int)blockIdx.x < nred) {
    wgrad_reduce_body<SL>(blockIdx.x, red, part, count, splitk, dw, cin, rs, c_real);
    return;
  }
  const int cl = threadIdx.x & 3, rl = threadIdx.x >> 2;
  const int ch = ((int)blockIdx.x - nred) * 4 + cl;
  double a1[4] = {0.0, 0.0, 0.0, 0.0};
  if (ch < c) {
    int b = rl;
    for (; b + 3 * 64 < splits; b += 4 * 64) {
#pragma unroll
      for (int j = 0; j < 4; ++j) a1[j] += (double)cs_part[((long long)(b + j * 64) * 2 + 0) * c + ch];
    }
    for (; b < splits; b += 64) a1[0] += (double)cs_part[((long long)b * 2 + 0) * c + ch];
  }
  redc[rl][cl] = (a1[0] + a1[1]) + (a1[2] + a1[3]);
  __syncthreads();
  if (rl != 0 || ch >= c) return;
  double t1 = 0.0;
  for (int j = 0; j < 64; ++j) t1 += redc[j][cl];  // fixed order: deterministic
  cs_out[ch] = (float)t1;
}

static void launch_reduce(const float* part, long long count, int splitk, float* dw, int cin, int rs, int c_real, hipStream_t s) {
  const long long groups = count / 4;
  if (splitk <= 4)
    wgrad_reduce_kernel<1><<<ceil_div(groups, 256), 256, 0, s>>>(part, count, splitk, dw, cin, rs, c_real);
  else if (splitk <= 32 || groups >= 16384)
    wgrad_reduce_kernel<4><<<ceil_div(groups, 64), 256, 0, s>>>(part, count, splitk, dw, cin, rs, c_real);
  else
    wgrad_reduce_kernel<16><<<ceil_div(groups, 16), 256, 0, s>>>(part, count, splitk, dw, cin, rs, c_real);
}


// split-K reduction + fold of the channel sums in one launch (wgrad_finish_kernel); the SL choice is launch_reduce's
static void launch_finish(const float* part, long long count, int splitk, float* dw, int cin, int rs, int c_real, const float* cs_part, int splits,
                          int c, float* cs_out, hipStream_t s) {
  const long long groups = count / 4;
  const int nfin = ceil_div(c, 4);
  if (splitk <= 4) {
    const int nred = (int)ceil_div(groups, 256);
    wgrad_finish_kernel<1><<<nred + nfin, 256, 0, s>>>(part, count, splitk, dw, cin, rs, c_real, nred, cs_part, splits, c, cs_out);
  } else if (splitk <= 32 || groups >= 16384) {
    const int nred = (int)ceil_div(groups, 64);
    wgrad_finish_kernel<4><<<nred + nfin, 256, 0, s>>>(part, count, splitk, dw, cin, rs, c_real, nred, cs_part, splits, c, cs_out);
  } else {
    const int nred = (int)ceil_div(groups, 16);
    wgrad_finish_kernel<16><<<nred + nfin, 256, 0, s>>>(part, count, splitk, dw, cin, rs, c_real, nred, cs_part, splits, c, cs_out);
  }
}

// ======================================================================================================================
// 3x3 / stride 1 / pad 1 weight gradient, all nine taps in one block (bf16).
// The tap-by-tap kernel above re-stages the dy tile and a shifted x tile for every tap (16 KB of operands per 64 MFMAs of
// a 128x128 tile; 8 KB per 16 for the 64-channel layers).  Here the reduction runs over the ZERO-PADDED pixel grid
// with SHARED padding ((H+1) x (W+1) positions per image, images back to back: one zero column serves as the right pad of a row and
// the left pad of the next, one zero row as the bottom pad of an image and the top pad of the next): in that index space tap (r, s)
// is the constant row shift (r-1)(W+1) + (s-1), so one block keeps a sliding window of x rows in an LDS ring, reads the nine tap
// operands from it at nine row offsets, and reuses each dy fragment nine times: 8 KB staged per 144 MFMAs, no halo masks (pad
// positions hold zeros in both operands; the price is the (H+1)(W+1)/(HW) longer reduction -- 1.31 at 7^2, 1.15 at 14^2).
//   block = 64 cout x 64 cin x 9 taps, 4 waves side by side along cin (64 x 16 x 9 taps = 144 accumulator registers:
//   one x fragment address feeds four MFMAs -- the loop is VALU-issue-bound);
//   k-step = 32 padded pixels: one 16-B chunk of dy and one of x per thread, global -> VGPR -> LDS;
//   rows are 128 B + 32 B pad (the 8 rows a 32-lane group of ds_read_b64_tr_b16 touches fall on distinct bank slots);
//   x ring = 256 rows: window [32j - 64, 32j + 96) for step j, chunk j + 3 is staged during step j;
//   one barrier per k-step.  Split-K over padded-pixel ranges; partials in the layout of wgrad_kernel.
//
// STRIDE 2 (S2; H, W even): the reduction runs over the padded OUTPUT grid ((Ho+1) x (Wo+1) per image, pixel (ho, wo) at (ho+1, wo+1)),
// where dy lives; the input is seen as its four PARITY PLANES P_pq[h'][w'] = x[2h'+p][2w'+q], each on that same padded grid.  Filter
// row r reads input row 2ho + r - 1: r = 1 -> plane p = 0 at h' = ho, r = 0 -> plane 1 at ho - 1, r = 2 -> plane 1 at ho (columns
// alike), so every tap is again ONE plane at a constant row shift, now 0, -1, -(Wo+1) or -(Wo+2) -- never forward.  Four plane rings
// of 128 rows (the window is the current chunk and the one before it; two chunks in flight), filled by LDS-DMA lanes that gather
// every second pixel; the MFMA loop, fragment addressing and rotation keys are the stride-1 kernel's.  4x the x bytes per MFMA of
// stride 1 (the input is 4x the output), against the tap-by-tap kernel's 9 re-staged x tiles.
struct Wgrad3Args {
  const bf16_t* x;
  const bf16_t* dy;
  float* part;            // [splitk][Cout][9*Cin]
  int Cout, Cin, H, W;    // H, W: the grid the reduction runs over (= output size; = input size for stride 1)
  int nt;                 // cin tiles
  long long q_total;      // N * (H+1) * (W+1): padded grid with SHARED pad rows / columns (see wgrad3x3_kernel)
  int per_split;          // padded pixels per split (multiple of 32)
  FastDiv div_pp, div_wp; // (H+1)*(W+1), W+1
};

__device__ uint4 g_wg_zero_page[8];  // 128 B of zeros: source of the LDS-DMA lanes that fall on pad positions

template <bool S2>
__global__ __launch_bounds__(256, 2) void wgrad3x3_kernel(Wgrad3Args p) {
  constexpr int NPL = S2 ? 4 : 1;          // x rings (parity planes)
  constexpr int RING = S2 ? 128 : 256;     // rows per ring (chunks of 32)
  constexpr int KP = 32;
  constexpr int D = S2 ? 2 : 3;            // DMA distance in k-steps
  constexpr int LOOK = S2 ? 0 : 2;         // chunks of look-ahead the tap shifts need (stride 2 only looks back)
  constexpr int BACK = S2 ? 1 : 2;         // chunks of look-back: chunk c sits in ring slot (c + BACK) % (RING / 32)
  constexpr int AHEAD = LOOK + D;          // chunk issued at step j (dy rides along: one decode)
  constexpr int NDY = AHEAD + 1;           // dy buffers
  constexpr int NDMA = NPL + 1;            // DMA instructions per wave and step
  __shared__ __attribute__((aligned(16))) char smem[NPL * RING * 128 + NDY * KP * 128];
  char* ring = smem;
  char* sdy = smem + NPL * RING * 128;

  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int li = lane & 15, g = lane >> 4;
  const int wn = wave;  // waves side by side along cin: each 64 cout x 16 cin x 9 taps (a B fragment feeds 4 MFMAs)
  int logical = xcd_remap_w(blockIdx.x, gridDim.x);
  const int nt_i = logical % p.nt; logical /= p.nt;
  const int mt = p.Cout >> 6;
  const int mt_i = logical % mt;
  const int split = logical / mt;
  const int k0 = mt_i * 64, c0 = nt_i * 64;
  const long long q0 = (long long)split * p.per_split;
  long long q1 = q0 + p.per_split;
  if (q1 > p.q_total) q1 = p.q_total;
  const int nk = q0 < q1 ? (int)((q1 - q0 + KP - 1) / KP) : 0;
  // Padded grid with SHARED padding: row pitch W + 1 (ONE zero column: the right pad of a row is the left pad of the next), H + 1 rows
  // per image (ONE zero row: the bottom pad of an image is the top pad of the next); pixel (h, w) sits at (h + 1, w + 1).  A tap shift
  // that leaves the image lands on a pad position of this or the next row / image; past the last image positions read zeros.
  // (H+1)(W+1) instead of (H+2)(W+2) positions per image: 21 % fewer MFMAs at 7^2, 12 % at 14^2.
  const int WP = p.W + 1;

  // Operands go global -> LDS by LDS-DMA (inline asm: see igemm256_kernel), D k-steps ahead of their use.  A DMA
  // instruction of wave w fills rows 8w .. 8w+7 of a 32-row chunk: lane l = row 8w + (l >> 3), 16-B slot l & 7.  Rows are
  // 128 B, unpadded; the four 32-B channel groups of row R are rotated by R >> 1 (applied on the source side of the DMA
  // and in the fragment addresses) so the 8 consecutive rows a 32-lane group of ds_read_b64_tr_b16 touches fall on 8
  // distinct bank slots (SQ_LDS_BANK_CONFLICT = 0).  The loop is VALU-issue-bound, not latency-bound: everything that
  // does not depend on the step (rotations, tap offsets, channel offsets) is hoisted, and dy / x share one index decode.
  const int srow = tid >> 3, slot = tid & 7;
  auto dma16 = [](const void* src, unsigned lds_addr) __attribute__((always_inline)) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_addr), "v"(src));
  };
  const unsigned smem_addr = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  const char* zsrc = reinterpret_cast<const char*>(g_wg_zero_page) + slot * 16;
  // channel offset (elements) of the 16-B slot this lane fills in a row with rotation key (R >> 1) & 3: a chunk's rows
  // R = 32 * chunk + srow (+ 64) all have the key of srow
  const int ch_slot = ((((slot >> 1) - (srow >> 1)) & 3) * 16) + (slot & 1) * 8;
  const char* xb = reinterpret_cast<const char*>(p.x + c0 + ch_slot);
  const char* dyb = reinterpret_cast<const char*>(p.dy + k0 + ch_slot);
  const unsigned xstride = (unsigned)p.Cin * 2u, dystride = (unsigned)p.Cout * 2u;
  // chunk c (>= -BACK): x rows -> ring slot (c + BACK) % (RING / 32); dy rows (c >= 0, inside this block's range) -> buffer c % NDY
  auto dma_chunk = [&](int c, bool with_dy) __attribute__((always_inline)) {
    const long long q = q0 + (long long)c * KP + srow;
    const bool in = q >= 0 && q < p.q_total;
    const unsigned qu = in ? (unsigned)q : 0u;  // q_total < 2^31 (checked on the host)
    const unsigned img = fdiv(qu, p.div_pp);
    const unsigned rem = qu - img * p.div_pp.d;
    const unsigned hp = fdiv(rem, p.div_wp);
    const unsigned wp = rem - hp * p.div_wp.d;
    const bool ok = in && hp - 1u < (unsigned)p.H && wp - 1u < (unsigned)p.W;
    const unsigned pix = (img * (unsigned)p.H + (hp - 1u)) * (unsigned)p.W + (wp - 1u);  // < 2^31 real pixels (of the reduction grid)
    const unsigned dst = smem_addr + ((((c + BACK) * KP) & (RING - 1)) + wave * 8) * 128;
    if constexpr (S2) {
      // plane (pp, pq) of this position: input pixel (2 (hp-1) + pp, 2 (wp-1) + pq) of the 2H x 2W input
      const unsigned pix00 = (img * (unsigned)(2 * p.H) + 2u * (hp - 1u)) * (unsigned)(2 * p.W) + 2u * (wp - 1u);
      const unsigned long long rowb = (unsigned long long)(2 * p.W) * xstride;
      const char* s00 = xb + (unsigned long long)pix00 * xstride;
      dma16(ok ? s00 : zsrc, dst);
      dma16(ok ? s00 + xstride : zsrc, dst + RING * 128);
      dma16(ok ? s00 + rowb : zsrc, dst + 2 * RING * 128);
      dma16(ok ? s00 + rowb + xstride : zsrc, dst + 3 * RING * 128);
    } else {
      dma16(ok ? xb + (unsigned long long)pix * xstride : zsrc, dst);
    }
    if (with_dy)
      dma16(ok && q < q1 ? dyb + (unsigned long long)pix * dystride : zsrc, smem_addr + NPL * RING * 128 + ((c % NDY) * KP + wave * 8) * 128);
  };

  f32x4 acc[9][4];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) acc[t][mi] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // prologue, in the order the loop's counted vmcnt waits assume (NDMA instructions per step from chunk 0 on)
#pragma unroll
  for (int c = -BACK; c < 0; ++c) dma_chunk(c, false);
#pragma unroll
  for (int c = 0; c < AHEAD; ++c) dma_chunk(c, true);

  typedef s16x4 __attribute__((address_space(3))) * lds_ptr;
  auto tr2 = [](const char* a_lo, const char* a_hi) __attribute__((always_inline)) -> uint4 {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(a_lo));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(a_hi));
    uint4 f;
    f.x = (unsigned)(unsigned short)lo[0] | ((unsigned)(unsigned short)lo[1] << 16);
    f.y = (unsigned)(unsigned short)lo[2] | ((unsigned)(unsigned short)lo[3] << 16);
    f.z = (unsigned)(unsigned short)hi[0] | ((unsigned)(unsigned short)hi[1] << 16);
    f.w = (unsigned)(unsigned short)hi[2] | ((unsigned)(unsigned short)hi[3] << 16);
    return f;
  };
  // lane's rows inside a 32-row k-step: 4g + (li >> 2) and + 16; its 8 bytes inside the 32-B channel group: (li & 3) * 8
  const int frow = 4 * g + (li >> 2);
  const int fcol = (li & 3) * 8;
  // dy fragments: rows frow / frow + 16 of the buffer (same rotation key: 16 >> 1 = 0 mod 4), channel groups mi = 0..3
  const int key_a = (frow >> 1) & 3;
  int offa[4];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) offa[mi] = frow * 128 + (((mi + key_a) & 3) * 32) + fcol;
  // x fragment of tap t: ring row (32 (j + BACK) + frow + off_t) & (RING - 1) of the tap's plane; its key does not depend on j
  // (32 j = 0 mod 8).  Stride 2: filter row r -> plane row parity (r != 1), shift -1 for r = 0 (columns alike)
  int trow[9], tcol[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int r = t / 3, sx = t % 3;
    const int off = S2 ? ((r == 0 ? -1 : 0) * WP + (sx == 0 ? -1 : 0)) : ((r - 1) * WP + (sx - 1));
    const int plane = S2 ? ((r != 1 ? 2 : 0) + (sx != 1 ? 1 : 0)) : 0;
    trow[t] = BACK * 32 + frow + off;  // >= 0: stride 1 off >= -(W + 3) >= -61; stride 2 off >= -(W + 2) >= -32
    tcol[t] = plane * (RING * 128) + (((wn + (trow[t] >> 1)) & 3) * 32) + fcol;
  }
  for (int j = 0; j < nk; ++j) {
    // everything but the DMAs of the last D - 1 steps has landed (NDMA per step and wave); after the barrier every wave's
    // part of dy chunk j / x chunk j + LOOK is visible and every wave is done with step j - 1's operands
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(NDMA * (D - 1)) : "memory");
    dma_chunk(j + AHEAD, true);  // a ring slot outside the window (chunks j - BACK .. j + LOOK) and the chunks in flight; dy buffer
                                 // (j + AHEAD) % NDY = the one step j - 1 read
    const char* tA = sdy + (j % NDY) * (KP * 128);
    uint4 fa[4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) fa[mi] = tr2(tA + offa[mi], tA + offa[mi] + 16 * 128);
    const int jb = (j * KP) & (RING - 1);
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int r_lo = (jb + trow[t]) & (RING - 1), r_hi = (r_lo + 16) & (RING - 1);
      const uint4 fb = tr2(ring + r_lo * 128 + tcol[t], ring + r_hi * 128 + tcol[t]);
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
        acc[t][mi] = sh_mfma16(fa[mi], fb, acc[t][mi]);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this step's fragment reads are done before the next barrier
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // C[m = cout][n = cin] per tap: lane holds cin = c0 + wn*16 + li, couts k0 + mi*16 + 4g + r
  const long long row_len = 9ll * p.Cin;
  float* dst = p.part + (long long)split * p.Cout * row_len;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int k = k0 + mi * 16 + 4 * g + r;
        const int c = c0 + wn * 16 + li;
        dst[(long long)k * row_len + (long long)t * p.Cin + c] = acc[t][mi][r];
      }
}


// ======================================================================================================================
// e4m3 form of wgrad3x3_kernel (3x3 / stride 1 / pad 1; BASELINE configs[4]): both operands are e4m3 CODES -- x_q [N][H][W][Cin] (what the
// BatchNorm-apply in front of the fp8 forward already emits) and dy_q [N][H][W][Cout] (what the BatchNorm-backward apply emits for the fp8
// data gradient) -- and the reduction over the padded pixel grid runs on v_mfma_scale_f32_16x16x128_f8f6f4: ONE matrix instruction per
// (cout tile, tap) and 128 pixels where the bf16 kernel issues four.  Same decomposition: block = 64 cout x 64 cin x 9 taps, 4 waves side
// by side along cin, x rows in an LDS ring with the taps as constant row shifts, dy chunk reused by all nine taps.
//   k-step = 128 padded pixels; rows are 64 B (64 channels x 1 B); a DMA instruction fills 16 rows (lane l = row l >> 2, 16-B slot l & 3);
//   the reduction index of an MFMA operand is the PIXEL, so both operands are read transposed: ds_read_b64_tr_b8 hands lane l of a
//   16-lane group channel c0 + l of 8 consecutive pixels when lanes 2j, 2j + 1 address channels c0 .. c0 + 7 / c0 + 8 .. c0 + 15 of pixel
//   p0 + j (layout probed in round 4: profiles/r04_tr8_layout.txt); four such reads = the 32 bytes (32 pixels of k block lane >> 4) of one
//   operand, the same pixel order on both sides;
//   the four 16-B channel slots of row R are rotated by (R >> 2) + 2 ((R >> 5) & 1) on the DMA source side: the 8 rows of one read and
//   the rows of the neighbouring k block (+ 32) then fall on distinct bank quartets whatever the tap shift is;
//   x ring = 4 chunks of 128 rows (window: chunks j - 1 .. j + 1 for |shift| <= W + 2 <= 128, chunk j + 2 in flight), 3 dy buffers;
//   56 KB of LDS: two blocks per CU.  Partials = acc / (scale_x scale_dy) in wgrad_kernel's layout; the same reduce kernel.
struct Wgrad3F8Args {
  const unsigned char* x;   // e4m3 codes [N][H][W][Cin]
  const unsigned char* dy;  // e4m3 codes [N][H][W][Cout]
  float* part;              // [splitk][Cout][9*Cin]
  const float* x_state;     // [1] = 1 / scale of the x codes
  const float* dy_state;
  int Cout, Cin, H, W;
  int nt;
  long long q_total;
  int per_split;            // padded pixels per split (multiple of 128)
  FastDiv div_pp, div_wp;
};

__global__ __launch_bounds__(256, 2) void wgrad3x3_f8_kernel(Wgrad3F8Args p) {
  constexpr int KP = 128, NCH = 4, RING = NCH * KP, ROWB = 64;
  constexpr int BACK = 1, AHEAD = 2, NDY = AHEAD + 1;   // chunk c + AHEAD is issued at step c: x one step before its first use, dy two
  __shared__ __attribute__((aligned(16))) char smem[RING * ROWB + NDY * KP * ROWB];
  char* ring = smem;
  char* sdy = smem + RING * ROWB;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int l16 = lane & 15, g = lane >> 4;
  const int wn = wave;
  int logical = xcd_remap_w(blockIdx.x, gridDim.x);
  const int nt_i = logical % p.nt; logical /= p.nt;
  const int mt = p.Cout >> 6;
  const int mt_i = logical % mt;
  const int split = logical / mt;
  const int k0 = mt_i * 64, c0 = nt_i * 64;
  const long long q0 = (long long)split * p.per_split;
  long long q1 = q0 + p.per_split;
  if (q1 > p.q_total) q1 = p.q_total;
  const int nk = q0 < q1 ? (int)((q1 - q0 + KP - 1) / KP) : 0;
  const int WP = p.W + 1;

  auto dma16 = [](const void* src, unsigned lds_addr) __attribute__((always_inline)) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_addr), "v"(src));
  };
  const unsigned smem_addr = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  const int slot = lane & 3;
  const char* zsrc = reinterpret_cast<const char*>(g_wg_zero_page) + slot * 16;
  auto rot = [](int row) __attribute__((always_inline)) -> int { return ((row >> 2) + 2 * ((row >> 5) & 1)) & 3; };
  // chunk c: 128 rows = 8 DMA instructions of 16 rows, instruction 2 w + i by wave w; lane l = row + (l >> 2), physical slot l & 3
  auto dma_chunk = [&](int c, bool with_dy) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = (2 * wave + i) * 16 + (lane >> 2);       // row inside the chunk
      const long long q = q0 + (long long)c * KP + r;
      const bool in = q >= 0 && q < p.q_total;
      const unsigned qu = in ? (unsigned)q : 0u;              // q_total < 2^31 (checked on the host)
      const unsigned img = fdiv(qu, p.div_pp);
      const unsigned rem = qu - img * p.div_pp.d;
      const unsigned hp = fdiv(rem, p.div_wp);
      const unsigned wp = rem - hp * p.div_wp.d;
      const bool ok = in && hp - 1u < (unsigned)p.H && wp - 1u < (unsigned)p.W;
      const unsigned pix = (img * (unsigned)p.H + (hp - 1u)) * (unsigned)p.W + (wp - 1u);
      const int ch = ((slot - rot(r)) & 3) * 16;              // logical channel slot this lane's physical slot holds (128 | chunk base: the key is r's)
      const unsigned dst = ((((c + BACK) & (NCH - 1)) * KP) + (2 * wave + i) * 16) * ROWB;
      dma16(ok ? reinterpret_cast<const char*>(p.x + (unsigned long long)pix * p.Cin + c0 + ch) : zsrc, smem_addr + dst);
      if (with_dy)
        dma16(ok && q < q1 ? reinterpret_cast<const char*>(p.dy + (unsigned long long)pix * p.Cout + k0 + ch) : zsrc,
              smem_addr + RING * ROWB + ((c % NDY) * KP + (2 * wave + i) * 16) * ROWB);
    }
  };

  f32x4 acc[9][4];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) acc[t][mi] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // prologue in the order the loop's waits assume: x chunk -1 (look-back), then chunks 0 and 1 with their dy
  dma_chunk(-1, false);
  dma_chunk(0, true);
  dma_chunk(1, true);

  typedef __attribute__((ext_vector_type(2))) int v2i;
  typedef __attribute__((ext_vector_type(8))) int i32x8;
  typedef v2i __attribute__((address_space(3))) * lds2;
  // one 32-byte operand: channel (16 sl + l16) of the 32 pixels of k block g starting at ring / buffer row `row0` (row0 includes 32 g)
  auto frag = [&](const char* base, int row0, int mask, int sl) __attribute__((always_inline)) -> i32x8 {
    i32x8 f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = (row0 + 8 * r + (l16 >> 1)) & mask;
      const v2i v = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds2)(base + row * ROWB + (((sl + rot(row)) & 3) * 16) + (l16 & 1) * 8));
      f[2 * r] = v[0];
      f[2 * r + 1] = v[1];
    }
    return f;
  };
  int toff[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) toff[t] = BACK * KP + 32 * g + (t / 3 - 1) * WP + (t % 3 - 1);   // >= 128 - (W + 2) >= 0

  for (int j = 0; j < nk; ++j) {
    // everything issued so far has landed (x chunk j + 1 and dy chunk j + 1 went out a whole step ago); after the barrier every wave's
    // part is visible and every wave is done with step j - 1's operands
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    dma_chunk(j + AHEAD, true);  // ring slot (j + 3) & 3 = the one of chunk j - 2 (outside the window j - 1 .. j + 1); dy buffer (j + 2) % 3 = step j - 1's
    const char* tA = sdy + (j % NDY) * (KP * ROWB);
    i32x8 fa[4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) fa[mi] = frag(tA, 32 * g, KP - 1, mi);
    const int jb = (j * KP) & (RING - 1);
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const i32x8 fb = frag(ring, jb + toff[t], RING - 1, wn);
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
        acc[t][mi] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fa[mi], fb, acc[t][mi], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this step's fragment reads are done before the next barrier
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // C[m = cout][n = cin] per tap: lane holds cin = c0 + wn*16 + l16, couts k0 + mi*16 + 4g + r (the bf16 kernel's layout)
  const float descale = p.x_state[1] * p.dy_state[1];
  const long long row_len = 9ll * p.Cin;
  float* dst = p.part + (long long)split * p.Cout * row_len;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int k = k0 + mi * 16 + 4 * g + r;
        const int c = c0 + wn * 16 + l16;
        dst[(long long)k * row_len + (long long)t * p.Cin + c] = acc[t][mi][r] * descale;
      }
}


// ======================================================================================================================
// 1x1 / stride-1 weight gradient for the layers with >= 256 channels on both sides (bf16): 256 x 256 tile, operands by LDS-DMA.
// The pointer-walking kernel above stages a 256 x 128 tile through registers (192 B of operands per MFMA, one k-step of prefetch);
// here both operand tiles of a 32-pixel k-step go global -> LDS by DMA three steps ahead (128 B per MFMA, nothing staged in VGPRs):
//   8 waves as 2 (cout halves) x 4 (cin quarters), each 128 x 64 = 8 x 4 MFMA tiles (128 accumulator registers), one block per CU;
//   a stage = eight 4-KB sub-tiles of [32 pixels][64 channels] (four of dy, four of x) in exactly the layout of wgrad3x3_kernel
//   (128-B rows, 32-B channel groups rotated by row / 2: conflict-free ds_read_b64_tr_b16), wave w fills sub-tile w with four DMA
//   instructions of 8 rows; four stages of 32 KB; one barrier and a counted vmcnt per k-step.
// Split-K over pixel ranges; partials in the layout of wgrad_kernel.  dy's column sums (the by-product the folded BatchNorm backward
// wants) come from the matrix pipes too: one extra MFMA of a dy fragment against an all-ones operand gives the pixel sums of its 16
// channels; wave (wm, wn) does that for two of its eight channel groups (+2 MFMAs per 32), the cin-tile-0 blocks only.
// Not for the launches with an operand transform (the BatchNorm-fold Gram launches stay on the pointer-walking kernel).
struct Wgrad1Args {
  const bf16_t* x;
  const bf16_t* dy;
  float* part;        // [splitk][Cout][Cin]
  long long Mo;       // pixels
  int Cout, Cin;
  int mt, nt;         // 256-wide cout / cin tiles
  int per_split;      // pixels per split (multiple of 32)
  float* dy_colsum;   // [splitk][2][Cout] as in WgradArgs (row 0 = per-channel sums of dy over the split's pixels, row 1 = 0); null = off
};

__global__ __launch_bounds__(512, 1) void wgrad1x1_dma_kernel(Wgrad1Args p) {
  constexpr int KP = 32, NS = 4, D = 3;  // D: DMA distance in k-steps (4 instructions per wave and step)
  constexpr int SUB = KP * 128, STAGE = 8 * SUB;
  __shared__ __attribute__((aligned(16))) char smem[NS * STAGE];
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int li = lane & 15, g = lane >> 4;
  const int wm = wave >> 2, wn = wave & 3;
  int logical = xcd_remap_w(blockIdx.x, gridDim.x);
  const int nt_i = logical % p.nt; logical /= p.nt;
  const int mt_i = logical % p.mt;
  const int split = logical / p.mt;
  const int k0 = mt_i * 256, c0 = nt_i * 256;
  const long long q0 = (long long)split * p.per_split;
  long long q1 = q0 + p.per_split;
  if (q1 > p.Mo) q1 = p.Mo;
  const int rows_total = q0 < q1 ? (int)(q1 - q0) : 0;
  const int nk = (rows_total + KP - 1) / KP;

  auto dma16 = [](const void* src, unsigned lds_addr) __attribute__((always_inline)) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_addr), "v"(src));
  };
  const unsigned smem_addr = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  // wave w fills sub-tile w (0-3: dy channel groups k0 + 64 w; 4-7: x channel groups c0 + 64 (w - 4)): instruction i covers tile rows
  // 8 i .. 8 i + 7, lane l = row 8 i + (l >> 3), 16-B slot l & 7; the slot's channel offset carries the row's rotation (see wgrad3x3_kernel)
  const int slot = lane & 7, r8 = lane >> 3;
  const bool is_dy = wave < 4;
  const unsigned stride = (unsigned)(is_dy ? p.Cout : p.Cin) * 2u;
  const char* base = is_dy ? reinterpret_cast<const char*>(p.dy + k0 + wave * 64) : reinterpret_cast<const char*>(p.x + c0 + (wave - 4) * 64);
  const char* zsrc = reinterpret_cast<const char*>(g_wg_zero_page) + slot * 16;
  const char* ptr[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = 8 * i + r8;
    const int ch_slot = ((((slot >> 1) - (row >> 1)) & 3) * 16) + (slot & 1) * 8;
    ptr[i] = base + (unsigned long long)(q0 + row) * stride + ch_slot * 2;
  }
  const unsigned step_bytes = (unsigned)KP * stride;
  auto dma_chunk = [&](int c) __attribute__((always_inline)) {
    const int left = rows_total - c * KP;  // rows of chunk c inside the block's pixel range (<= 0: none -- the zero page)
    const unsigned dst = smem_addr + (unsigned)(c & (NS - 1)) * STAGE + wave * SUB;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      dma16(8 * i + r8 < left ? ptr[i] : zsrc, dst + i * 8 * 128);
      ptr[i] += step_bytes;
    }
  };

  f32x4 acc[8][4];
#pragma unroll
  for (int mi = 0; mi < 8; ++mi)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};

#pragma unroll
  for (int c = 0; c < D; ++c) dma_chunk(c);
  const bool want_cs = p.dy_colsum != nullptr && nt_i == 0;  // block-uniform
  f32x4 acs[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
  const unsigned one2 = pack_bf16x2(1.0f, 1.0f);
  const uint4 ones = make_uint4(one2, one2, one2, one2);

  typedef s16x4 __attribute__((address_space(3))) * lds_ptr;
  auto tr2 = [](const char* a_lo, const char* a_hi) __attribute__((always_inline)) -> uint4 {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(a_lo));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(a_hi));
    uint4 f;
    f.x = (unsigned)(unsigned short)lo[0] | ((unsigned)(unsigned short)lo[1] << 16);
    f.y = (unsigned)(unsigned short)lo[2] | ((unsigned)(unsigned short)lo[3] << 16);
    f.z = (unsigned)(unsigned short)hi[0] | ((unsigned)(unsigned short)hi[1] << 16);
    f.w = (unsigned)(unsigned short)hi[2] | ((unsigned)(unsigned short)hi[3] << 16);
    return f;
  };
  // lane's rows inside a 32-row k-step: 4g + (li >> 2) and + 16; its 8 bytes inside the 32-B channel group: (li & 3) * 8
  const int frow = 4 * g + (li >> 2);
  const int fcol = (li & 3) * 8;
  const int key = (frow >> 1) & 3;
  int offs[4];  // channel group q (16 channels) of a sub-tile, rows frow / frow + 16 (same rotation key)
#pragma unroll
  for (int q = 0; q < 4; ++q) offs[q] = frow * 128 + (((q + key) & 3) * 32) + fcol;

  for (int j = 0; j < nk; ++j) {
    // all but the DMAs of the last D - 1 steps have landed; after the barrier chunk j is visible and every wave is past step j - 1
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(4 * (D - 1)) : "memory");
    dma_chunk(j + D);  // stage (j + D) % NS = the one step j - 1 read
    const char* st = smem + (j & (NS - 1)) * STAGE;
    const char* tA = st + (wm * 2) * SUB;
    const char* tB = st + (4 + wn) * SUB;
    uint4 fb[4];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) fb[ni] = tr2(tB + offs[ni], tB + offs[ni] + 16 * 128);
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) {
      const char* sub = tA + (mi >> 2) * SUB;
      const uint4 fa = tr2(sub + offs[mi & 3], sub + offs[mi & 3] + 16 * 128);
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = sh_mfma16(fa, fb[ni], acc[mi][ni]);
      if (want_cs && wn == (mi >> 1)) acs[mi & 1] = sh_mfma16(fa, ones, acs[mi & 1]);  // wave-uniform
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this step's fragment reads are done before the next barrier
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // C[m = cout][n = cin]: lane holds cin = c0 + wn*64 + ni*16 + li, couts k0 + wm*128 + mi*16 + 4g + r
  float* dst = p.part + (long long)split * p.Cout * p.Cin;
#pragma unroll
  for (int mi = 0; mi < 8; ++mi)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int k = k0 + wm * 128 + mi * 16 + 4 * g + r;
        const int c = c0 + wn * 64 + ni * 16 + li;
        dst[(long long)k * p.Cin + c] = acc[mi][ni][r];
      }
  if (want_cs && li == 0) {  // every column of the ones-product holds the same sums: column 0 stores them
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int k = k0 + wm * 128 + (2 * wn + j) * 16 + 4 * g + r;
        p.dy_colsum[((long long)split * 2 + 0) * p.Cout + k] = acs[j][r];
        p.dy_colsum[((long long)split * 2 + 1) * p.Cout + k] = 0.f;
      }
  }
}

static hook_t g_wg_dma{-1};  // -1 = simhand_test_switch(SH_SW_WG_DMA) (default on), 0 / 1 forced
static bool use_wg_dma(const sh_conv_desc* d) {
  const int env = sw(SH_SW_WG_DMA);
  const int h = g_wg_dma;
  // cout != cin: never the Gram launches of the BatchNorm fold (x = dy = a, operand transforms), which share the splits query
  return (h >= 0 ? h : env) && d->dtype == SH_BF16 && d->r == 1 && d->s == 1 && d->stride == 1 && d->pad == 0 && d->cout % 256 == 0 &&
         d->cin % 256 == 0 && d->cout != d->cin && (long long)d->n * d->ho * d->wo >= 256 * 64;
}
// one block per CU, one round: split-K = CUs / tiles, at least 16 k-steps per block (the 3-chunk prologue is per block)
static void plan_dma(const sh_conv_desc* d, int* splitk, int* per) {
  const long long mo = (long long)d->n * d->ho * d->wo;
  const long long tiles = (long long)(d->cout / 256) * (d->cin / 256);
  const long long ksteps = (mo + 31) / 32;
  long long sk = 256 / tiles;
  const long long max_sk = (ksteps + 15) / 16;
  if (sk > max_sk) sk = max_sk;
  if (sk < 1) sk = 1;
  const long long pr = (ksteps + sk - 1) / sk;
  sk = (ksteps + pr - 1) / pr;
  *splitk = (int)sk;
  *per = (int)(pr * 32);
}

static hook_t g_use_wgrad3{1};
static hook_t g_wg3_blocks{512};  // all-taps 3x3 kernel: two blocks per CU, one round (512 beats 768 by 4-10 %)
static hook_t g_wgrad3_s2{-1};  // stride-2 form: -1 = simhand_test_switch(SH_SW_WG3_S2) (default on), 0 / 1 forced
static bool use_wgrad3(const sh_conv_desc* d) {
  if (!g_use_wgrad3 || d->dtype != SH_BF16 || d->r != 3 || d->s != 3 || d->pad != 1) return false;
  if (d->stride == 2) {  // parity-plane rings: even input, the largest tap shift -(wo + 2) inside one 32-row chunk
    const int env = sw(SH_SW_WG3_S2);
    const int h = g_wgrad3_s2;
    return (h >= 0 ? h : env) && d->h % 2 == 0 && d->w % 2 == 0 && d->wo + 2 <= 32 && (long long)d->n * (d->ho + 1) * (d->wo + 1) < (1ll << 31);
  }
  return d->stride == 1 && d->w + 3 <= 64 && (long long)d->n * (d->h + 1) * (d->w + 1) < (1ll << 31);
}
static void plan3(const sh_conv_desc* d, int* splitk, int* per) {
  const long long q_total = (long long)d->n * (d->ho + 1) * (d->wo + 1);  // (ho, wo) == (h, w) for stride 1
  const long long tiles = (long long)(d->cout / 64) * (d->cin / 64);
  const long long ksteps = (q_total + 31) / 32;
  long long sk = g_wg3_blocks / tiles;            // one full round of resident blocks (see plan)
  const long long max_sk = (ksteps + 31) / 32;    // at least 32 k-steps per block (the 5-chunk prologue is per block)
  if (sk > max_sk) sk = max_sk;
  if (sk < 1) sk = 1;
  long long pr = (ksteps + sk - 1) / sk;
  sk = (ksteps + pr - 1) / pr;
  *splitk = (int)sk;
  *per = (int)(pr * 32);
}

static hook_t g_use_tr{1};
// blocks a launch aims for (tiles x splits): ONE round of the two blocks a CU holds.  Measured per shape (scripts/wgrad_kpm.py):
// 512 beats 1024 by 1-7 % and 768 / 1536 (1.5 / 3 rounds: the last one half empty, more split-K partials) by 10-25 %
static hook_t g_wg_blocks{512};
static hook_t g_plain_kpm{2};  // k-step multiplier of the 1x1 pointer-walking kernel (2 = 64 pixels per barrier)
static bool is_plain(const sh_conv_desc* d) {
  return d->dtype == SH_BF16 && d->r == 1 && d->s == 1 && d->stride == 1 && d->pad == 0;
}
// 256 x 128 tiles (32-pixel k-steps, 8 x 4 MFMA tiles per wave) for the plain 1x1 kernel where cout % 256 == 0: a quarter less operand
// traffic per MFMA than 128 x 128 -- (1024, 256) @ 14^2 293 -> 264 us, (2048, 512) @ 7^2 268 -> 251 (SH_SW_WG_BIG = 0: A/B timing)
static hook_t g_wg_big{-1};
static bool wg_big(const sh_conv_desc* d) {
  const int env = sw(SH_SW_WG_BIG);
  const int h = g_wg_big;
  // cout != cin: never the Gram launches of the BatchNorm fold (x = dy = a: their operand transforms live in the 128-row tiles)
  return (h >= 0 ? h : env) && is_plain(d) && g_use_tr && d->cout % 256 == 0 && d->cin % 128 == 0 && d->cout != d->cin;
}
// one tile across the WIDE side of the stage-1 layers (64 <-> 256 channels): the narrow operand is then staged once per pixel range
// instead of once per 128-channel tile of the wide side (4.7 -> 4.1 GB per launch at 2048 x 56^2); SH_SW_WG_WIDE = 0: A/B timing
static bool wg_wide(const sh_conv_desc* d) {
  const int env = sw(SH_SW_WG_WIDE);
  return env && is_plain(d) && g_use_tr && ((d->cin == 64 && d->cout == 256) || (d->cout == 64 && d->cin == 256));
}
static void plan(const sh_conv_desc* d, int* bm, int* bn, int* splitk, int* pps, bool allow_big = true) {
  int kp = d->dtype == SH_F32 ? 16 : (is_plain(d) && g_use_tr ? 32 * g_plain_kpm : 32);
  *bm = d->cout % 128 == 0 ? 128 : 64;
  *bn = d->cin % 128 == 0 ? 128 : 64;
  if (allow_big && wg_big(d)) {  // (launches with an operand transform keep the 128-row tiles)
    *bm = 256;
    kp = 32;
  } else if (allow_big && wg_wide(d)) {
    *bm = d->cout;
    *bn = d->cin;
    kp = 32;
  }
  const long long mo = (long long)d->n * d->ho * d->wo;
  const long long tiles = (long long)(d->cout / *bm) * (d->cin / *bn) * d->r * d->s;
  const long long ksteps = (mo + kp - 1) / kp;
  // the pointer-walking 1x1 kernel (64-pixel k-steps, 70 KB of LDS) holds two blocks per CU, the generic one three
  const int wgb = g_wg_blocks;
  const int target = (is_plain(d) && g_use_tr && d->dtype == SH_BF16) ? wgb : (wgb * 3) / 2;
  long long sk = target / tiles;               // never more blocks than one round
  const long long max_sk = (ksteps + 7) / 8;   // at least 8 k-steps per block
  if (sk > max_sk) sk = max_sk;
  if (sk < 1) sk = 1;
  long long per = (ksteps + sk - 1) / sk;      // k-steps per split
  sk = (ksteps + per - 1) / per;
  *splitk = (int)sk;
  *pps = (int)(per * kp);
}

void hooks_reset_wgrad() {
  g_wg_dma = -1;
  g_use_wgrad3 = 1;
  g_wgrad3_s2 = -1;
  g_wg3_blocks = 512;
  g_use_tr = 1;
  g_wg_blocks = 512;
  g_plain_kpm = 2;
}

}  // namespace sh

using namespace sh;

extern "C" {

// tuning hook: all-taps 3x3 weight-gradient kernel for the bf16 stride-1 layers (1 = default)
int simhand_test_wgrad_dma_enable(int on) {
  g_wg_dma = on ? 1 : 0;
  return 0;
}

int simhand_test_wgrad3x3_enable(int on) {
  g_use_wgrad3 = on ? 1 : 0;
  return 0;
}

// test hook: choose the bf16 LDS transpose path (1 = ds_read_b64_tr_b16, 0 = scalar reads)
// tuning hook: pixels per k-step of the 1x1 pointer-walking kernel, 32 * kpm (kpm = 1 or 2; 2 = default)
int simhand_test_wgrad_target_blocks(int n, int n3x3) {
  g_wg_blocks = n >= 64 ? n : 512;
  g_wg3_blocks = n3x3 >= 64 ? n3x3 : 512;
  return 0;
}

int simhand_test_wgrad_plain_kpm(int kpm) {
  g_plain_kpm = kpm == 1 ? 1 : 2;
  return 0;
}

int simhand_test_wgrad_set_tr(int on) {
  g_use_tr = on ? 1 : 0;
  return 0;
}

size_t simhand_conv2d_wgrad_workspace_bytes(const sh_conv_desc* d) {
  if (!d) return 0;
  int bm, bn, sk, pps;
  if (use_wgrad3(d)) plan3(d, &sk, &pps);
  else {
    int sk2;
    plan(d, &bm, &bn, &sk, &pps);
    plan(d, &bm, &bn, &sk2, &pps, false);  // either tile plan may run (operand-transform launches take the 128-row one)
    if (sk2 > sk) sk = sk2;
    if (use_wg_dma(d)) {
      plan_dma(d, &sk2, &pps);
      if (sk2 > sk) sk = sk2;
    }
  }
  return (size_t)sk * d->cout * d->cin * d->r * d->s * sizeof(float);
}

struct WgXform {  // operand transform of the 1x1 pointer-walking kernel (see WgradArgs)
  int mode;        // 1 = BN-apply (+ReLU) Gram, 2 = BN-backward apply on the dy operand
  const float *scale, *shift, *ca, *cb, *cc;
  const void* y;   // mode 2
  void* out;
  int relu;
};

static int wgrad_impl(const sh_conv_desc* d, const void* x, const void* dy, float* dw, int c_real, void* workspace,
                      size_t workspace_bytes, sh_stream_t stream, int stem_hp = 0, int stem_wp = 0, float* dy_colsum = nullptr,
                      const WgXform* xf = nullptr, float* colsum_sum = nullptr) {
  SH_REQUIRE(d != nullptr, "conv2d_wgrad: desc is NULL");
  SH_REQUIRE(x && dy && dw && workspace, "conv2d_wgrad: NULL pointer");
  SH_REQUIRE(d->dtype == SH_F32 || d->dtype == SH_BF16, "conv2d_wgrad: bad dtype %d", d->dtype);
  SH_REQUIRE(d->stride == 1 || d->stride == 2, "conv2d_wgrad: stride %d unsupported", d->stride);
  SH_REQUIRE(d->cin % 64 == 0 && d->cout % 64 == 0, "conv2d_wgrad: cin=%d / cout=%d must be multiples of 64", d->cin, d->cout);
  SH_REQUIRE(d->ho == (d->h + 2 * d->pad - d->r) / d->stride + 1 && d->wo == (d->w + 2 * d->pad - d->s) / d->stride + 1,
             "conv2d_wgrad: ho/wo inconsistent");
  const long long mo = (long long)d->n * d->ho * d->wo;
  SH_REQUIRE(mo < (1ll << 31), "conv2d_wgrad: %lld output pixels exceed the 2^31 index range", mo);
  SH_REQUIRE((long long)d->n * d->h * d->w < (1ll << 31), "conv2d_wgrad: input pixels exceed the 2^31 index range");
  SH_REQUIRE(stem_wp == 0 || (long long)d->n * stem_hp * stem_wp < (1ll << 31), "conv2d_wgrad: padded stem pixels exceed the 2^31 index range");
  SH_REQUIRE(workspace_bytes >= simhand_conv2d_wgrad_workspace_bytes(d), "conv2d_wgrad: workspace too small");
  if (stem_wp == 0 && use_wgrad3(d)) {
    Wgrad3Args b;
    int sk, per;
    plan3(d, &sk, &per);
    b.x = (const bf16_t*)x; b.dy = (const bf16_t*)dy; b.part = (float*)workspace;
    b.Cout = d->cout; b.Cin = d->cin; b.H = d->ho; b.W = d->wo;  // the reduction grid (== the input grid for stride 1)
    b.nt = d->cin / 64;
    b.q_total = (long long)d->n * (d->ho + 1) * (d->wo + 1);
    b.per_split = per;
    b.div_pp = make_fastdiv((unsigned)((d->ho + 1) * (d->wo + 1)));
    b.div_wp = make_fastdiv((unsigned)(d->wo + 1));
    hipStream_t s3 = (hipStream_t)stream;
    const double flops3 = 2.0 * (double)mo * d->cout * d->cin * 9;
    const double bytes3 = 2.0 * ((double)d->n * d->h * d->w * d->cin + (double)mo * d->cout) + 4.0 * d->cout * d->cin * 9;
    ProfScope ps3(SH_PROF_CONV_WGRAD, s3, flops3, bytes3);
    route_hit(SH_ROUTE_WGRAD3X3);
    if (d->stride == 2) wgrad3x3_kernel<true><<<sk * (d->cout / 64) * (d->cin / 64), 256, 0, s3>>>(b);
    else wgrad3x3_kernel<false><<<sk * (d->cout / 64) * (d->cin / 64), 256, 0, s3>>>(b);
    if (check_launch("conv2d_wgrad (3x3)")) return 1;
    launch_reduce(b.part, (long long)d->cout * d->cin * 9, sk, dw, d->cin, 9, c_real, s3);
    return check_launch("conv2d_wgrad (3x3) reduce");
  }
  if (stem_wp == 0 && xf == nullptr && x != dy && d->cout != d->cin && g_use_tr && use_wg_dma(d)) {
    Wgrad1Args b;
    int sk, per;
    plan_dma(d, &sk, &per);
    b.x = (const bf16_t*)x; b.dy = (const bf16_t*)dy; b.part = (float*)workspace;
    b.Mo = mo; b.Cout = d->cout; b.Cin = d->cin;
    b.mt = d->cout / 256; b.nt = d->cin / 256;
    b.per_split = per;
    b.dy_colsum = dy_colsum;
    if (dy_colsum != nullptr) route_hit(SH_ROUTE_WGRAD_COLSUM);
    hipStream_t s1 = (hipStream_t)stream;
    ProfScope ps1(SH_PROF_CONV_WGRAD, s1, 2.0 * (double)mo * d->cout * d->cin,
                  2.0 * ((double)mo * d->cin + (double)mo * d->cout) + 4.0 * d->cout * d->cin);
    route_hit(SH_ROUTE_WGRAD_PLAIN);
    wgrad1x1_dma_kernel<<<sk * b.mt * b.nt, 512, 0, s1>>>(b);
    if (check_launch("conv2d_wgrad (1x1, DMA tiles)")) return 1;
    if (colsum_sum != nullptr) launch_finish(b.part, (long long)d->cout * d->cin, sk, dw, d->cin, 1, c_real, dy_colsum, sk, d->cout, colsum_sum, s1);
    else launch_reduce(b.part, (long long)d->cout * d->cin, sk, dw, d->cin, 1, c_real, s1);
    return check_launch("conv2d_wgrad (1x1, DMA tiles) reduce");
  }
  WgradArgs a;
  int bm, bn;
  plan(d, &bm, &bn, &a.splitk, &a.pix_per_split, xf == nullptr);
  a.x = x; a.dy = dy; a.part = (float*)workspace;
  a.Mo = mo; a.Cout = d->cout; a.Cin = d->cin;
  a.R = d->r; a.S = d->s; a.stride = d->stride; a.pad = d->pad;
  a.Ho = d->ho; a.Wo = d->wo; a.H = d->h; a.W = d->w;
  a.mt = d->cout / bm; a.nt = d->cin / bn;
  a.div_hw = make_fastdiv((unsigned)(d->ho * d->wo));
  a.div_w = make_fastdiv((unsigned)d->wo);
  a.use_tr = g_use_tr;
  a.stem_hp = stem_hp; a.stem_wp = stem_wp;
  a.dy_colsum = dy_colsum;
  const bool plain = d->dtype == SH_BF16 && d->r == 1 && d->s == 1 && d->stride == 1 && d->pad == 0 && a.use_tr && stem_wp == 0;
  SH_REQUIRE(dy_colsum == nullptr || plain, "conv2d_wgrad_colsum: only for bf16 1x1 / stride-1 convolutions");
  SH_REQUIRE(xf == nullptr || (plain && g_plain_kpm == 2), "fused BatchNorm operand transforms: only for bf16 1x1 / stride-1 convolutions");
  if (xf != nullptr) {
    a.xs = xf->scale; a.xh = xf->shift; a.xa = xf->ca; a.xb = xf->cb; a.xc = xf->cc; a.dy2 = xf->y; a.xout = xf->out; a.xrelu = xf->relu;
  }
  hipStream_t s = (hipStream_t)stream;
  const int nblk = a.splitk * d->r * d->s * a.mt * a.nt;
  const double es = d->dtype == SH_F32 ? 4 : 2;
  double flops = 2.0 * (double)mo * d->cout * d->cin * d->r * d->s;
  double bytes = es * ((double)d->n * d->h * d->w * d->cin + (double)mo * d->cout) + 4.0 * d->cout * d->cin * d->r * d->s;
  if (x == dy) flops = 0.0;  // Gram launch of the BatchNorm fold (a^T a): its time stays in the class, its FLOPs are not
                             // algorithmic work of the layer graph and are not counted towards any roofline figure
  if (stem_wp > 0) {  // algorithmic figures of the 7x7x3 stem, not of its padded K = 256 lowering
    flops = 2.0 * (double)mo * 64 * 147;
    bytes = es * ((double)d->n * stem_hp * stem_wp * 4 + (double)mo * 64) + 4.0 * 64 * 147;
  }
  ProfScope ps(SH_PROF_CONV_WGRAD, s, flops, bytes);
  route_hit(stem_wp > 0 ? SH_ROUTE_WGRAD_STEM : (plain ? SH_ROUTE_WGRAD_PLAIN : SH_ROUTE_WGRAD_GENERIC));
  if (dy_colsum != nullptr) route_hit(SH_ROUTE_WGRAD_COLSUM);
#define SH_WG(T, BM, BN) wgrad_kernel<T, BM, BN><<<nblk, 256, 0, s>>>(a)
  const int stem256 = sw(SH_SW_STEM_WG256);  // (A/B timing; 1.68 -> 1.57 ms at 2048 images)
  if (stem_wp > 0 && stem256 && d->dtype == SH_BF16) {
    // all 256 virtual channels in one tile: dy is staged once per pixel range instead of once per 128-channel tile; 32-pixel k-steps
    // (the 64 x 256 tile's LDS at 64 pixels would leave one block per CU); twice the splits keep the block count
    const long long ksteps = (mo + 31) / 32;
    long long sk = 2ll * a.splitk;
    long long per = (ksteps + sk - 1) / sk;
    sk = (ksteps + per - 1) / per;
    a.splitk = (int)sk;
    a.pix_per_split = (int)(per * 32);
    a.nt = 1;
    SH_REQUIRE(workspace_bytes >= (size_t)sk * d->cout * d->cin * sizeof(float), "stem_conv_wgrad: workspace too small");
    wgrad_kernel<bf16_t, 64, 256, true, false, 1><<<a.splitk, 256, 0, s>>>(a);
  } else if (stem_wp > 0) {  // cout 64 x 256 virtual channels -> the 64 x 128 tile with the NHWC4 address map
    if (d->dtype == SH_F32) wgrad_kernel<float, 64, 128, true><<<nblk, 256, 0, s>>>(a);
    else if (g_plain_kpm == 2) wgrad_kernel<bf16_t, 64, 128, true, false, 2><<<nblk, 256, 0, s>>>(a);  // 64-pixel k-steps: 16 MFMAs per barrier
    else wgrad_kernel<bf16_t, 64, 128, true><<<nblk, 256, 0, s>>>(a);
  } else if (d->dtype == SH_F32) {
    if (bm == 128 && bn == 128) SH_WG(float, 128, 128);
    else if (bm == 128) SH_WG(float, 128, 64);
    else if (bn == 128) SH_WG(float, 64, 128);
    else SH_WG(float, 64, 64);
  } else if (d->r == 1 && d->s == 1 && d->stride == 1 && d->pad == 0 && a.use_tr) {
#define SH_WGP(BM, BN)                                                                     \
  do {                                                                                     \
    if (g_plain_kpm == 2) wgrad_kernel<bf16_t, BM, BN, false, true, 2><<<nblk, 256, 0, s>>>(a); \
    else wgrad_kernel<bf16_t, BM, BN, false, true><<<nblk, 256, 0, s>>>(a);                  \
  } while (0)
#define SH_WGX(BM, BN, XF) wgrad_kernel<bf16_t, BM, BN, false, true, 2, XF><<<nblk, 256, 0, s>>>(a)
    if (xf != nullptr && xf->mode == 1) {
      route_hit(SH_ROUTE_BN_APPLY_GRAM);
      if (bm == 128 && bn == 128) SH_WGX(128, 128, 1);
      else if (bm == 128) SH_WGX(128, 64, 1);
      else if (bn == 128) SH_WGX(64, 128, 1);
      else SH_WGX(64, 64, 1);
    } else if (xf != nullptr) {
      route_hit(SH_ROUTE_WGRAD_BNBWD);
      if (bm == 128 && bn == 128) SH_WGX(128, 128, 2);
      else if (bm == 128) SH_WGX(128, 64, 2);
      else if (bn == 128) SH_WGX(64, 128, 2);
      else SH_WGX(64, 64, 2);
    } else if (bm == 256 && bn == 64) wgrad_kernel<bf16_t, 256, 64, false, true, 1><<<nblk, 256, 0, s>>>(a);
    else if (bm == 64 && bn == 256) wgrad_kernel<bf16_t, 64, 256, false, true, 1><<<nblk, 256, 0, s>>>(a);
    else if (bm == 256) wgrad_kernel<bf16_t, 256, 128, false, true, 1><<<nblk, 256, 0, s>>>(a);
    else if (bm == 128 && bn == 128) SH_WGP(128, 128);
    else if (bm == 128) SH_WGP(128, 64);
    else if (bn == 128) SH_WGP(64, 128);
    else SH_WGP(64, 64);
#undef SH_WGX
#undef SH_WGP
  } else {
    if (bm == 128 && bn == 128) SH_WG(bf16_t, 128, 128);
    else if (bm == 128) SH_WG(bf16_t, 128, 64);
    else if (bn == 128) SH_WG(bf16_t, 64, 128);
    else SH_WG(bf16_t, 64, 64);
  }
#undef SH_WG
  if (check_launch("conv2d_wgrad")) return 1;
  const long long count = (long long)d->cout * d->cin * d->r * d->s;
  if (colsum_sum != nullptr) launch_finish(a.part, count, a.splitk, dw, d->cin, d->r * d->s, c_real, dy_colsum, a.splitk, d->cout, colsum_sum, s);
  else launch_reduce(a.part, count, a.splitk, dw, d->cin, d->r * d->s, c_real, s);
  return check_launch("conv2d_wgrad reduce");
}

int simhand_conv2d_wgrad(const sh_conv_desc* d, const void* x, const void* dy, float* dw, void* workspace, size_t workspace_bytes,
                         sh_stream_t stream) {
  return wgrad_impl(d, x, dy, dw, 0, workspace, workspace_bytes, stream);
}

int simhand_conv2d_wgrad_splits(const sh_conv_desc* d) {
  if (!d) return 0;
  int bm, bn, sk, pps;
  if (use_wgrad3(d)) plan3(d, &sk, &pps);
  else if (g_use_tr && use_wg_dma(d)) plan_dma(d, &sk, &pps);
  else plan(d, &bm, &bn, &sk, &pps);
  return sk;
}

int simhand_conv2d_wgrad_colsum(const sh_conv_desc* d, const void* x, const void* dy, float* dw, float* dy_colsum, void* workspace,
                                size_t workspace_bytes, sh_stream_t stream) {
  SH_REQUIRE(dy_colsum != nullptr, "conv2d_wgrad_colsum: NULL dy_colsum");
  return wgrad_impl(d, x, dy, dw, 0, workspace, workspace_bytes, stream, 0, 0, dy_colsum);
}

// the same, with the channel sums folded to dy_sum[cout] inside the reduction launch (dy_colsum stays the scratch it was)
int simhand_conv2d_wgrad_colsum_sums(const sh_conv_desc* d, const void* x, const void* dy, float* dw, float* dy_colsum, float* dy_sum,
                                     void* workspace, size_t workspace_bytes, sh_stream_t stream) {
  SH_REQUIRE(dy_colsum != nullptr && dy_sum != nullptr, "conv2d_wgrad_colsum_sums: NULL dy_colsum / dy_sum");
  SH_REQUIRE(d != nullptr && simhand_conv2d_wgrad_splits(d) < 4096, "conv2d_wgrad_colsum_sums: >= 4096 splits (use simhand_conv2d_wgrad_colsum + simhand_bn_bwd_finalize_raw)");
  return wgrad_impl(d, x, dy, dw, 0, workspace, workspace_bytes, stream, 0, 0, dy_colsum, nullptr, dy_sum);
}

int simhand_bn_apply_gram_sums(const sh_conv_desc* d, const void* y, const float* scale, const float* shift, int relu, void* a, float* s2,
                               float* colsum_partial, float* colsum, void* workspace, size_t workspace_bytes, sh_stream_t stream) {
  SH_REQUIRE(d != nullptr && d->cin == d->cout, "bn_apply_gram_sums: the descriptor must be the c -> c 1x1 Gram descriptor");
  SH_REQUIRE(y && scale && shift && a && s2 && colsum_partial && colsum, "bn_apply_gram_sums: NULL pointer");
  SH_REQUIRE(simhand_conv2d_wgrad_splits(d) < 4096, "bn_apply_gram_sums: >= 4096 splits");
  WgXform xf = {1, scale, shift, nullptr, nullptr, nullptr, nullptr, a, relu};
  return wgrad_impl(d, y, y, s2, 0, workspace, workspace_bytes, stream, 0, 0, colsum_partial, &xf, colsum);
}

int simhand_bn_apply_gram(const sh_conv_desc* d, const void* y, const float* scale, const float* shift, int relu, void* a, float* s2,
                          float* colsum_partial, void* workspace, size_t workspace_bytes, sh_stream_t stream) {
  SH_REQUIRE(d != nullptr && d->cin == d->cout, "bn_apply_gram: the descriptor must be the c -> c 1x1 Gram descriptor");
  SH_REQUIRE(y && scale && shift && a && s2 && colsum_partial, "bn_apply_gram: NULL pointer");
  WgXform xf = {1, scale, shift, nullptr, nullptr, nullptr, nullptr, a, relu};
  return wgrad_impl(d, y, y, s2, 0, workspace, workspace_bytes, stream, 0, 0, colsum_partial, &xf);
}

int simhand_conv2d_wgrad_bnbwd(const sh_conv_desc* d, const void* x, const void* da, const void* y, const float* scale, const float* shift,
                               const float* coef_a, const float* coef_b, const float* coef_c, int relu, void* dy_out, float* dw_oihw,
                               int c_real, void* workspace, size_t workspace_bytes, sh_stream_t stream) {
  SH_REQUIRE(d != nullptr, "conv2d_wgrad_bnbwd: desc is NULL");
  SH_REQUIRE(x && da && y && scale && shift && coef_a && coef_b && coef_c && dy_out && dw_oihw, "conv2d_wgrad_bnbwd: NULL pointer");
  SH_REQUIRE(c_real >= 0 && c_real <= d->cin, "conv2d_wgrad_bnbwd: c_real=%d outside [0, cin=%d]", c_real, d->cin);
  WgXform xf = {2, scale, shift, coef_a, coef_b, coef_c, y, dy_out, relu};
  return wgrad_impl(d, x, da, dw_oihw, c_real, workspace, workspace_bytes, stream, 0, 0, nullptr, &xf);
}

// ---- e4m3 weight gradient (3x3 / stride 1 with >= 256 channels on both sides: the layers whose forward and data gradient run on e4m3) ----
static bool wgrad_f8_ok(const sh_conv_desc* d) {
  return d != nullptr && d->dtype == SH_BF16 && d->r == 3 && d->s == 3 && d->pad == 1 && d->stride == 1 && d->cin % 64 == 0 && d->cout % 64 == 0 &&
         d->cin >= 256 && d->cout >= 256 && d->w + 2 <= 128 && d->ho == d->h && d->wo == d->w &&
         (long long)d->n * (d->h + 1) * (d->w + 1) < (1ll << 31) && (long long)d->n * d->h * d->w * (d->cin > d->cout ? d->cin : d->cout) < (1ll << 32);
}
static void plan3_f8(const sh_conv_desc* d, int* splitk, int* per) {
  const long long q_total = (long long)d->n * (d->h + 1) * (d->w + 1);
  const long long tiles = (long long)(d->cout / 64) * (d->cin / 64);
  const long long ksteps = (q_total + 127) / 128;
  long long sk = g_wg3_blocks / tiles;            // one full round of the two blocks a CU holds
  const long long max_sk = (ksteps + 7) / 8;      // at least 8 k-steps (1024 padded pixels) per block
  if (sk > max_sk) sk = max_sk;
  if (sk < 1) sk = 1;
  long long pr = (ksteps + sk - 1) / sk;
  sk = (ksteps + pr - 1) / pr;
  *splitk = (int)sk;
  *per = (int)(pr * 128);
}

int simhand_conv2d_wgrad_fp8_pays(const sh_conv_desc* d) { return wgrad_f8_ok(d) ? 1 : 0; }

size_t simhand_conv2d_wgrad_fp8_workspace_bytes(const sh_conv_desc* d) {
  if (!wgrad_f8_ok(d)) return 0;
  int sk, per;
  plan3_f8(d, &sk, &per);
  return (size_t)sk * d->cout * d->cin * 9 * sizeof(float);
}

int simhand_conv2d_wgrad_fp8(const sh_conv_desc* d, const void* x_q, const void* dy_q, const float* x_state, const float* dy_state, float* dw_oihw,
                             void* workspace, size_t workspace_bytes, sh_stream_t stream) {
  SH_REQUIRE(wgrad_f8_ok(d), "conv2d_wgrad_fp8: only where simhand_conv2d_wgrad_fp8_pays(d) (3x3 / stride 1 / pad 1, >= 256 channels)");
  SH_REQUIRE(x_q && dy_q && x_state && dy_state && dw_oihw && workspace, "conv2d_wgrad_fp8: NULL pointer");
  SH_REQUIRE(workspace_bytes >= simhand_conv2d_wgrad_fp8_workspace_bytes(d), "conv2d_wgrad_fp8: workspace too small");
  Wgrad3F8Args b;
  int sk, per;
  plan3_f8(d, &sk, &per);
  b.x = (const unsigned char*)x_q; b.dy = (const unsigned char*)dy_q; b.part = (float*)workspace;
  b.x_state = x_state; b.dy_state = dy_state;
  b.Cout = d->cout; b.Cin = d->cin; b.H = d->h; b.W = d->w;
  b.nt = d->cin / 64;
  b.q_total = (long long)d->n * (d->h + 1) * (d->w + 1);
  b.per_split = per;
  b.div_pp = make_fastdiv((unsigned)((d->h + 1) * (d->w + 1)));
  b.div_wp = make_fastdiv((unsigned)(d->w + 1));
  hipStream_t s = (hipStream_t)stream;
  const double mo = (double)d->n * d->h * d->w;
  ProfScope ps(SH_PROF_CONV_WGRAD, s, 2.0 * mo * d->cout * d->cin * 9, mo * (d->cin + d->cout) + 4.0 * d->cout * d->cin * 9);
  route_hit(SH_ROUTE_WGRAD3X3);
  route_hit(SH_ROUTE_FP8_WGRAD);
  wgrad3x3_f8_kernel<<<sk * (d->cout / 64) * (d->cin / 64), 256, 0, s>>>(b);
  if (check_launch("conv2d_wgrad_fp8 (3x3)")) return 1;
  launch_reduce(b.part, (long long)d->cout * d->cin * 9, sk, dw_oihw, d->cin, 9, d->cin, s);
  return check_launch("conv2d_wgrad_fp8 (3x3) reduce");
}

int simhand_conv2d_wgrad_oihw(const sh_conv_desc* d, const void* x, const void* dy, float* dw_oihw, int c_real, void* workspace,
                              size_t workspace_bytes, sh_stream_t stream) {
  SH_REQUIRE(d != nullptr, "conv2d_wgrad_oihw: desc is NULL");
  SH_REQUIRE(c_real >= 1 && c_real <= d->cin, "conv2d_wgrad_oihw: c_real=%d outside [1, cin=%d]", c_real, d->cin);
  return wgrad_impl(d, x, dy, dw_oihw, c_real, workspace, workspace_bytes, stream);
}

// weight gradient of the direct stem (see simhand_stem_conv_fwd): dw is the OIHW fp32 [64][3][7][7] gradient
static void stem_wgrad_desc(sh_conv_desc* d, int n, int h, int w, int dtype) {
  memset(d, 0, sizeof(*d));
  d->n = n;
  d->ho = d->h = (h + 6 - 7) / 2 + 1;
  d->wo = d->w = (w + 6 - 7) / 2 + 1;
  d->cin = 256; d->cout = 64; d->r = d->s = 1; d->stride = 1; d->pad = 0; d->dtype = dtype;
}

// 224 x 224, 16-bit storage: both operands in LDS rings, one pass over dy and the padded input (stem_bwd.hip stem_wgrad_ring_kernel)
static bool stem_wgrad_ring_ok(int n, int h, int w, int dtype) {
  int hp, wp, ho, wo;
  if (n < 1 || dtype != SH_BF16 || !sw(SH_SW_STEM_WG_RING) || simhand_stem_geometry(h, w, &hp, &wp, &ho, &wo)) return false;
  return stem_ring_geometry_ok(hp, wp, ho, wo);  // 224^2 and, since round 6, 128^2 inputs
}

size_t simhand_stem_conv_wgrad_workspace_bytes(int n, int h, int w, int dtype) {
  sh_conv_desc d;
  stem_wgrad_desc(&d, n, h, w, dtype);
  const size_t tile = 2 * simhand_conv2d_wgrad_workspace_bytes(&d) + 4096;  // (the single-tile form runs twice the splits)
  const size_t ring = n >= 1 ? (size_t)stem_wgrad_ring_blocks(n) * 64 * 224 * sizeof(float) : 0;  // per-block partials of the ring kernel
  return tile > ring ? tile : ring;   // (independent of the test switch: either kernel may take the call)
}

int simhand_stem_conv_wgrad(const void* xp, const void* dy, float* dw_oihw, void* workspace, size_t workspace_bytes, int n, int h,
                            int w, int dtype, sh_stream_t stream) {
  SH_REQUIRE(n >= 1 && h >= 1 && w >= 1, "stem_conv_wgrad: bad shape");
  sh_conv_desc d;
  stem_wgrad_desc(&d, n, h, w, dtype);
  const int hp = h + 8, wp = (w + 8 + 7) / 8 * 8;
  if (stem_wgrad_ring_ok(n, h, w, dtype)) {
    SH_REQUIRE(xp && dy && dw_oihw && workspace, "stem_conv_wgrad: NULL pointer");
    SH_REQUIRE(workspace_bytes >= (size_t)stem_wgrad_ring_blocks(n) * 64 * 224 * sizeof(float), "stem_conv_wgrad: workspace too small");
    const double mo = (double)n * 112 * 112;
    ProfScope ps(SH_PROF_CONV_WGRAD, (hipStream_t)stream, 2.0 * mo * 64 * 147, 2.0 * (mo * 64 + (double)n * hp * wp * 4));
    route_hit(SH_ROUTE_WGRAD_STEM);
    launch_stem_wgrad_ring(xp, dy, dw_oihw, (float*)workspace, n, hp, wp, d.wo, (hipStream_t)stream);
    return check_launch("stem_conv_wgrad (LDS rings)");
  }
  return wgrad_impl(&d, xp, dy, dw_oihw, -1, workspace, workspace_bytes, stream, hp, wp);
}

}  // extern "C"
