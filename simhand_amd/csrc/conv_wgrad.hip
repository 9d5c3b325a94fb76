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

namespace sh {

struct FastDiv {
  unsigned d, mul, shr;
};
static FastDiv make_fastdiv(unsigned d) {
  FastDiv f;
  f.d = d;
  if (d == 1) {
    f.mul = 0;
    f.shr = 0;
    return f;
  }
  unsigned lg = 31 - __builtin_clz(d);
  if (d & (d - 1)) lg += 1;
  const unsigned p = 31 + lg;
  f.mul = (unsigned)(((1ull << p) + d - 1) / d);
  f.shr = p - 32;
  return f;
}
__device__ __forceinline__ unsigned fdiv(unsigned n, const FastDiv& f) {  // n < 2^31
  return f.d == 1 ? n : (__umulhi(n, f.mul) >> f.shr);
}

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
  int stem_hp, stem_wp; // > 0: x is the zero-padded NHWC4 stem input [N][hp][wp][4]; Cin = 256 virtual channels
                        // = 8 filter rows x (8 taps x 4 channels), row r of output pixel (ho, wo) at (2ho + r, 2wo)
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

template <typename T, int BM, int BN, bool STEM = false>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(WgradArgs p) {
  constexpr int KP = WgCfg<T>::KP;
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

  uint4 ra[NA], rb[NBL];
  auto load_step = [&](int ks) __attribute__((always_inline)) {
    const long long base = pix_begin + (long long)ks * KP;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int id = tid + 256 * i;
      const int row = id / CPR_A, ch = id - row * CPR_A;
      const long long m = base + row;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (m < pix_end) v = *reinterpret_cast<const uint4*>(dys + m * p.Cout + k0 + ch * VE);
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < NBL; ++i) {
      const int id = tid + 256 * i;
      const int row = id / CPR_B, ch = id - row * CPR_B;
      const long long m = base + row;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (m < pix_end) {
        const unsigned mu = (unsigned)m;
        const unsigned img = fdiv(mu, p.div_hw);
        const unsigned rem = mu - img * hw;
        const unsigned ho = fdiv(rem, p.div_w);
        const unsigned wo = rem - ho * (unsigned)p.Wo;
        if constexpr (STEM) {
          const int vc = c0 + ch * VE;  // virtual channel: filter row vc / 32, element vc % 32 of its 8 x 4 run
          v = *reinterpret_cast<const uint4*>(xs + (((long long)img * p.stem_hp + 2 * ho + (vc >> 5)) * p.stem_wp + 2 * wo) * 4 +
                                              (vc & 31));
        } else {
          const int hs = (int)ho * p.stride - p.pad + fr;
          const int ws = (int)wo * p.stride - p.pad + fs;
          if ((unsigned)hs < (unsigned)p.H && (unsigned)ws < (unsigned)p.W)
            v = *reinterpret_cast<const uint4*>(xs + (((long long)img * p.H + hs) * p.W + ws) * p.Cin + c0 + ch * VE);
        }
      }
      rb[i] = v;
    }
  };
  auto store_step = [&](int buf) __attribute__((always_inline)) {
    char* dA = sA + buf * (KP * SA);
    char* dB = sB + buf * (KP * SB);
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int id = tid + 256 * i;
      const int row = id / CPR_A, ch = id - row * CPR_A;
      *reinterpret_cast<uint4*>(dA + row * SA + ch * 16) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < NBL; ++i) {
      const int id = tid + 256 * i;
      const int row = id / CPR_B, ch = id - row * CPR_B;
      *reinterpret_cast<uint4*>(dB + row * SB + ch * 16) = rb[i];
    }
  };

  f32x4 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};

  if (nk > 0) {
    load_step(0);
    store_step(0);
  }
  __syncthreads();
  for (int ks = 0; ks < nk; ++ks) {
    const int buf = ks & 1;
    if (ks + 1 < nk) load_step(ks + 1);
    const char* tA = sA + buf * (KP * SA);
    const char* tB = sB + buf * (KP * SB);
    if constexpr (sizeof(T) == 2) {
      uint4 fa[MI], fb[NI];
      if (p.use_tr) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) fa[mi] = frag_bf16_tr(tA, SA, wm * (BM / 2) + mi * 16, lane);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) fb[ni] = frag_bf16_tr(tB, SB, wn * (BN / 2) + ni * 16, lane);
      } else {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) fa[mi] = frag_bf16_scalar(tA, SA, wm * (BM / 2) + mi * 16, lane);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) fb[ni] = frag_bf16_scalar(tB, SB, wn * (BN / 2) + ni * 16, lane);
      }
      typedef __attribute__((ext_vector_type(8))) __bf16 frag_t;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(frag_t, fa[mi]),
                                                                __builtin_bit_cast(frag_t, fb[ni]), acc[mi][ni], 0, 0, 0);
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
    if (ks + 1 < nk) store_step(buf ^ 1);
    __syncthreads();
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
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part, long long count, int splitk,
                                                           float* __restrict__ dw, int cin, int rs, int c_real) {
  constexpr int COLS = 256 / SL;
  __shared__ float4 red[SL > 1 ? 256 : 1];
  const int col = threadIdx.x % COLS, sl = threadIdx.x / COLS;
  const long long i = ((long long)blockIdx.x * COLS + col) * 4;
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

static void launch_reduce(const float* part, long long count, int splitk, float* dw, int cin, int rs, int c_real, hipStream_t s) {
  const long long groups = count / 4;
  if (splitk <= 4)
    wgrad_reduce_kernel<1><<<ceil_div(groups, 256), 256, 0, s>>>(part, count, splitk, dw, cin, rs, c_real);
  else if (splitk <= 32 || groups >= 16384)
    wgrad_reduce_kernel<4><<<ceil_div(groups, 64), 256, 0, s>>>(part, count, splitk, dw, cin, rs, c_real);
  else
    wgrad_reduce_kernel<16><<<ceil_div(groups, 16), 256, 0, s>>>(part, count, splitk, dw, cin, rs, c_real);
}

static int g_use_tr = 1;

static void plan(const sh_conv_desc* d, int* bm, int* bn, int* splitk, int* pps) {
  const int kp = d->dtype == SH_F32 ? 16 : 32;
  *bm = d->cout % 128 == 0 ? 128 : 64;
  *bn = d->cin % 128 == 0 ? 128 : 64;
  const long long mo = (long long)d->n * d->ho * d->wo;
  const long long tiles = (long long)(d->cout / *bm) * (d->cin / *bn) * d->r * d->s;
  const long long ksteps = (mo + kp - 1) / kp;
  long long sk = (1536 + tiles - 1) / tiles;   // aim for ~6 blocks per CU
  const long long max_sk = (ksteps + 7) / 8;   // at least 8 k-steps per block
  if (sk > max_sk) sk = max_sk;
  if (sk < 1) sk = 1;
  long long per = (ksteps + sk - 1) / sk;      // k-steps per split
  sk = (ksteps + per - 1) / per;
  *splitk = (int)sk;
  *pps = (int)(per * kp);
}

}  // namespace sh

using namespace sh;

extern "C" {

// test hook: choose the bf16 LDS transpose path (1 = ds_read_b64_tr_b16, 0 = scalar reads)
int simhand_wgrad_set_tr(int on) {
  g_use_tr = on ? 1 : 0;
  return 0;
}

size_t simhand_conv2d_wgrad_workspace_bytes(const sh_conv_desc* d) {
  if (!d) return 0;
  int bm, bn, sk, pps;
  plan(d, &bm, &bn, &sk, &pps);
  return (size_t)sk * d->cout * d->cin * d->r * d->s * sizeof(float);
}

static int wgrad_impl(const sh_conv_desc* d, const void* x, const void* dy, float* dw, int c_real, void* workspace,
                      size_t workspace_bytes, sh_stream_t stream, int stem_hp = 0, int stem_wp = 0) {
  SH_REQUIRE(d != nullptr, "conv2d_wgrad: desc is NULL");
  SH_REQUIRE(x && dy && dw && workspace, "conv2d_wgrad: NULL pointer");
  SH_REQUIRE(d->dtype == SH_F32 || d->dtype == SH_BF16, "conv2d_wgrad: bad dtype %d", d->dtype);
  SH_REQUIRE(d->stride == 1 || d->stride == 2, "conv2d_wgrad: stride %d unsupported", d->stride);
  SH_REQUIRE(d->cin % 64 == 0 && d->cout % 64 == 0, "conv2d_wgrad: cin=%d / cout=%d must be multiples of 64", d->cin, d->cout);
  SH_REQUIRE(d->ho == (d->h + 2 * d->pad - d->r) / d->stride + 1 && d->wo == (d->w + 2 * d->pad - d->s) / d->stride + 1,
             "conv2d_wgrad: ho/wo inconsistent");
  const long long mo = (long long)d->n * d->ho * d->wo;
  SH_REQUIRE(mo < (1ll << 31), "conv2d_wgrad: %lld output pixels exceed the 2^31 index range", mo);
  SH_REQUIRE(workspace_bytes >= simhand_conv2d_wgrad_workspace_bytes(d), "conv2d_wgrad: workspace too small");
  WgradArgs a;
  int bm, bn;
  plan(d, &bm, &bn, &a.splitk, &a.pix_per_split);
  a.x = x; a.dy = dy; a.part = (float*)workspace;
  a.Mo = mo; a.Cout = d->cout; a.Cin = d->cin;
  a.R = d->r; a.S = d->s; a.stride = d->stride; a.pad = d->pad;
  a.Ho = d->ho; a.Wo = d->wo; a.H = d->h; a.W = d->w;
  a.mt = d->cout / bm; a.nt = d->cin / bn;
  a.div_hw = make_fastdiv((unsigned)(d->ho * d->wo));
  a.div_w = make_fastdiv((unsigned)d->wo);
  a.use_tr = g_use_tr;
  a.stem_hp = stem_hp; a.stem_wp = stem_wp;
  hipStream_t s = (hipStream_t)stream;
  const int nblk = a.splitk * d->r * d->s * a.mt * a.nt;
  const double es = d->dtype == SH_F32 ? 4 : 2;
  double flops = 2.0 * (double)mo * d->cout * d->cin * d->r * d->s;
  double bytes = es * ((double)d->n * d->h * d->w * d->cin + (double)mo * d->cout) + 4.0 * d->cout * d->cin * d->r * d->s;
  if (stem_wp > 0) {  // algorithmic figures of the 7x7x3 stem, not of its padded K = 256 lowering
    flops = 2.0 * (double)mo * 64 * 147;
    bytes = es * ((double)d->n * stem_hp * stem_wp * 4 + (double)mo * 64) + 4.0 * 64 * 147;
  }
  ProfScope ps(SH_PROF_CONV_WGRAD, s, flops, bytes);
#define SH_WG(T, BM, BN) wgrad_kernel<T, BM, BN><<<nblk, 256, 0, s>>>(a)
  if (stem_wp > 0) {  // cout 64 x 256 virtual channels -> the 64 x 128 tile with the NHWC4 address map
    if (d->dtype == SH_F32) wgrad_kernel<float, 64, 128, true><<<nblk, 256, 0, s>>>(a);
    else wgrad_kernel<bf16_t, 64, 128, true><<<nblk, 256, 0, s>>>(a);
  } else if (d->dtype == SH_F32) {
    if (bm == 128 && bn == 128) SH_WG(float, 128, 128);
    else if (bm == 128) SH_WG(float, 128, 64);
    else if (bn == 128) SH_WG(float, 64, 128);
    else SH_WG(float, 64, 64);
  } else {
    if (bm == 128 && bn == 128) SH_WG(bf16_t, 128, 128);
    else if (bm == 128) SH_WG(bf16_t, 128, 64);
    else if (bn == 128) SH_WG(bf16_t, 64, 128);
    else SH_WG(bf16_t, 64, 64);
  }
#undef SH_WG
  if (check_launch("conv2d_wgrad")) return 1;
  const long long count = (long long)d->cout * d->cin * d->r * d->s;
  launch_reduce(a.part, count, a.splitk, dw, d->cin, d->r * d->s, c_real, s);
  return check_launch("conv2d_wgrad reduce");
}

int simhand_conv2d_wgrad(const sh_conv_desc* d, const void* x, const void* dy, float* dw, void* workspace, size_t workspace_bytes,
                         sh_stream_t stream) {
  return wgrad_impl(d, x, dy, dw, 0, workspace, workspace_bytes, stream);
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

size_t simhand_stem_conv_wgrad_workspace_bytes(int n, int h, int w, int dtype) {
  sh_conv_desc d;
  stem_wgrad_desc(&d, n, h, w, dtype);
  return simhand_conv2d_wgrad_workspace_bytes(&d);
}

int simhand_stem_conv_wgrad(const void* xp, const void* dy, float* dw_oihw, void* workspace, size_t workspace_bytes, int n, int h,
                            int w, int dtype, sh_stream_t stream) {
  SH_REQUIRE(n >= 1 && h >= 1 && w >= 1, "stem_conv_wgrad: bad shape");
  sh_conv_desc d;
  stem_wgrad_desc(&d, n, h, w, dtype);
  const int hp = h + 8, wp = (w + 8 + 7) / 8 * 8;
  return wgrad_impl(&d, xp, dy, dw_oihw, -1, workspace, workspace_bytes, stream, hp, wp);
}

}  // extern "C"
