// FP8 (OCP e4m3) first slice for BASELINE configs[4] ("ResNet-50 simclr fp8"): per-tensor scaled quantisation and the
// forward implicit-GEMM convolution on gfx950's block-scaled matrix instruction.
//
// The reference has NO fp8 path (its mixed precision is fp16 autocast + GradScaler, src/experiments/main.py:158-159,
// config/training_config.json:9), so there are no reference semantics to match: "parity: n/a", the gate is agreement
// with the bf16 path of this library (tests/test_gpu_fp8.py).
//
// gfx950 has two fp8 MFMA families: the carried-forward v_mfma_f32_16x16x32_fp8_fp8 (runs at the bf16 rate) and the
// block-scaled v_mfma_scale_f32_16x16x128_f8f6f4 (K = 128 per instruction, twice the bf16 rate, ~5 PF dense).  This kernel
// uses the second with all E8M0 block scales = 127 (2^0): the per-TENSOR scales are applied once in the epilogue,
//     y = (sum_k q(x) q(w)) / (scale_x * scale_w),   q(v) = e4m3(clamp(v * scale, +-448)).
// Tile 128 (pixels) x 128 (channels) x 128 B of k per step = ONE scaled MFMA per 16x16 tile pair and k-step (two
// 16-B LDS chunks per operand and lane; the k permutation inside the step is free because both operands use the same
// one).  Operand staging / swizzle / swapped-operand trick as conv_igemm.hip's 128-row kernel: global -> VGPR -> LDS,
// one LDS tile, three blocks per CU.  Output bf16 (+ fused BatchNorm partial sums of the descaled fp32 accumulators).
#include "common.h"

namespace sh {

typedef __attribute__((ext_vector_type(8))) int i32x8;
constexpr float kFp8Max = 448.f;  // largest finite e4m3fn

__device__ __forceinline__ float clamp_fp8(float v) { return fminf(fmaxf(v, -kFp8Max), kFp8Max); }
// 4 floats -> 4 e4m3 bytes (round-to-nearest-even, saturated by the clamp: the hardware convert alone would give NaN
// past 448 in the OCP "fn" encoding)
__device__ __forceinline__ unsigned pack_fp8x4(float a, float b, float c, float d) {
  unsigned r = 0;
  r = __builtin_amdgcn_cvt_pk_fp8_f32(clamp_fp8(a), clamp_fp8(b), r, false);
  r = __builtin_amdgcn_cvt_pk_fp8_f32(clamp_fp8(c), clamp_fp8(d), r, true);
  return r;
}

// ---- amax / scale bookkeeping ------------------------------------------------------------------------------------------
// max |x| as the uint bit pattern of a non-negative float (order-preserving): atomicMax is exact and order-independent
template <typename T>
__global__ __launch_bounds__(256) void fp8_amax_kernel(const T* __restrict__ x, long long nvec, unsigned* __restrict__ amax_bits) {
  float m = 0.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long long)gridDim.x * blockDim.x) {
    float v[Vec16<T>::N];
    Vec16<T>::load(x + i * Vec16<T>::N, v);
#pragma unroll
    for (int e = 0; e < Vec16<T>::N; ++e) m = fmaxf(m, fabsf(v[e]));
  }
  m = wave_max(m);
  __shared__ float red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    atomicMax(amax_bits, __float_as_uint(m));
  }
}

// state = float[ST_LEN]: [0] scale, [1] 1/scale, [2] ring position, [3] calls, [4 .. 4+HIST) amax history.
// mode 0 (current scaling): scale from amax_new alone; mode 1 / 2 (delayed): amax_new goes into the ring, the scale for
// the NEXT quantisation is 448 / max(ring) / 2^margin.  amax_new is consumed (reset to 0).
// state[0] = the scale the NEXT quantisation uses; state[1] = 1 / (the scale of the codes that exist NOW): after mode 1 -- the update that
// FOLLOWS a delayed site's quantisation -- that is the reciprocal of the scale the pass just used (until round 5 it was overwritten with the
// next scale's, so every consumer launched behind the update -- the convolutions -- de-scaled by the wrong factor whenever the ring's
// maximum had moved; BatchNorm and LARS hid it); modes 0 and 2 run BEFORE the quantisation they serve: 1 / new scale.
__global__ void fp8_scale_update_kernel(float* __restrict__ state, unsigned* __restrict__ amax_new_bits, int hist, float margin_pow2, int mode) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const float a = __uint_as_float(*amax_new_bits);
  *amax_new_bits = 0u;
  float m = a;
  if (mode != 0) {
    const int pos = (int)state[2];
    state[4 + pos] = a;
    state[2] = (float)((pos + 1) % hist);
    m = 0.f;
    for (int i = 0; i < hist; ++i) m = fmaxf(m, state[4 + i]);
  }
  state[3] += 1.f;
  const float used = state[0];
  const float s = (m > 0.f && isfinite(m)) ? kFp8Max / (m * margin_pow2) : 1.f;
  state[0] = s;
  state[1] = mode == 1 ? 1.f / used : 1.f / s;
}

// q = e4m3(clamp(x * scale)); also folds max|x| of THIS tensor into amax_bits (input of the next scale update)
template <typename T>
__global__ __launch_bounds__(256) void fp8_quantize_kernel(const T* __restrict__ x, unsigned char* __restrict__ q, long long nvec16,
                                                           const float* __restrict__ state, unsigned* __restrict__ amax_bits) {
  // one thread = 16 output bytes = 16 source elements (two 16-B bf16 vectors / four fp32 vectors)
  constexpr int PER = 16 / Vec16<T>::N;
  const float s = state[0];
  float m = 0.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec16; i += (long long)gridDim.x * blockDim.x) {
    float v[16];
#pragma unroll
    for (int p = 0; p < PER; ++p) {
      float t[Vec16<T>::N];
      Vec16<T>::template load<true>(x + (i * PER + p) * Vec16<T>::N, t);
#pragma unroll
      for (int e = 0; e < Vec16<T>::N; ++e) v[p * Vec16<T>::N + e] = t[e];
    }
    uint4 o;
    unsigned w[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int e = 0; e < 4; ++e) m = fmaxf(m, fabsf(v[4 * j + e]));
      w[j] = pack_fp8x4(v[4 * j] * s, v[4 * j + 1] * s, v[4 * j + 2] * s, v[4 * j + 3] * s);
    }
    o = make_uint4(w[0], w[1], w[2], w[3]);
    st16<true>(q + i * 16, o);
  }
  if (amax_bits != nullptr) {
    m = wave_max(m);
    __shared__ float red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(amax_bits, __float_as_uint(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]))));
  }
}

// OIHW fp32 master weights -> KRSC e4m3 rows [k][r][s][c] * scale
__global__ __launch_bounds__(256) void fp8_pack_krsc_kernel(const float* __restrict__ w, unsigned char* __restrict__ q, int k, int c, int r,
                                                            int s, const float* __restrict__ state) {
  const long long total4 = (long long)k * r * s * c / 4;
  const float sc = state[0];
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long long)gridDim.x * blockDim.x) {
    const long long e0 = i * 4;
    const int ci = (int)(e0 % c);
    long long t = e0 / c;
    const int si = (int)(t % s);
    t /= s;
    const int ri = (int)(t % r);
    const int ki = (int)(t / r);
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = w[(((long long)ki * c + ci + e) * r + ri) * s + si] * sc;
    reinterpret_cast<unsigned*>(q)[i] = pack_fp8x4(v[0], v[1], v[2], v[3]);
  }
}

// ---- forward implicit GEMM -----------------------------------------------------------------------------------------------
struct Fp8Args {
  const unsigned char* x;   // [N][H][W][Cin] e4m3
  const unsigned char* w;   // [Cout][R][S][Cin] e4m3
  unsigned short* y;        // [Mo][Cout] bf16
  float* bn_partial;        // [m_tiles][2][Cout] or null
  const float* x_state;     // [1] = 1 / scale_x
  const float* w_state;     // [1] = 1 / scale_w
  long long Mo;
  int Cin, Cout, R, S, stride, pad, H, W, Ho, Wo;
  int m_tiles, n_tiles;
  FastDiv div_hw, div_w;
};

__device__ __forceinline__ int xcd_remap8(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7;
  const int xcd = bid & 7, j = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
}

__global__ __launch_bounds__(256, 3) void igemm_fp8_fwd_kernel(Fp8Args p) {
  constexpr int BN = 128, NI = 4, MI = 4;
  __shared__ __attribute__((aligned(16))) char smem[128 * 128 + BN * 128];
  char* sA = smem;
  char* sB = smem + 128 * 128;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int li = lane & 15, g = lane >> 4;
  const int wm = wave >> 1, wn = wave & 1;
  int logical = xcd_remap8(blockIdx.x, gridDim.x);
  const int n_tile = logical % p.n_tiles;
  const int m_tile = logical / p.n_tiles;
  const unsigned m0 = (unsigned)m_tile * 128u;
  const int n0 = n_tile * BN;

  // loader: thread = 16-B chunk (tid & 7) of rows (tid >> 3) + 32 i
  const int chunk = tid & 7, lrow = tid >> 3;
  const unsigned char* pa[4];
  int h0[4], w0[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const unsigned m = m0 + lrow + 32 * i;
    const bool ok = m < (unsigned)p.Mo;
    const unsigned mm = ok ? m : 0u;
    const unsigned img = fdiv(mm, p.div_hw);
    const unsigned rem = mm - img * p.div_hw.d;
    const unsigned ho = fdiv(rem, p.div_w);
    const unsigned wo = rem - ho * p.div_w.d;
    h0[i] = ok ? (int)ho * p.stride - p.pad : -(1 << 20);
    w0[i] = (int)wo * p.stride - p.pad;
    pa[i] = p.x + (((long long)img * p.H + h0[i] * (ok ? 1 : 0)) * p.W + w0[i]) * p.Cin + chunk * 16;
    if (!ok) pa[i] = p.x;
  }
  const long long wrow = (long long)p.R * p.S * p.Cin;
  const unsigned char* pb = p.w + (long long)(n0 + lrow) * wrow + chunk * 16;
  const int csteps = p.Cin / 128;
  const int nk = p.R * p.S * csteps;
  auto swz = [](int row, int ch) __attribute__((always_inline)) -> int { return ch ^ ((row >> 1) & 7); };
  const int st0 = lrow * 128 + swz(lrow, chunk) * 16;  // rows lrow + 32 i share the key

  int l_cs = 0, l_tr = 0, l_ts = 0;
  uint4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
  auto load_step = [&]() __attribute__((always_inline)) {
    const int aoff = (l_tr * p.W + l_ts) * p.Cin + l_cs * 128;
    const int boff = (l_tr * p.S + l_ts) * p.Cin + l_cs * 128;
    auto load_a = [&](int i) __attribute__((always_inline)) -> uint4 {
      const bool ok = (unsigned)(h0[i] + l_tr) < (unsigned)p.H && (unsigned)(w0[i] + l_ts) < (unsigned)p.W;
      return ok ? *reinterpret_cast<const uint4*>(pa[i] + aoff) : make_uint4(0, 0, 0, 0);
    };
    ra0 = load_a(0); ra1 = load_a(1); ra2 = load_a(2); ra3 = load_a(3);
    rb0 = *reinterpret_cast<const uint4*>(pb + boff);
    rb1 = *reinterpret_cast<const uint4*>(pb + 32 * wrow + boff);
    rb2 = *reinterpret_cast<const uint4*>(pb + 64 * wrow + boff);
    rb3 = *reinterpret_cast<const uint4*>(pb + 96 * wrow + boff);
    if (++l_cs == csteps) {
      l_cs = 0;
      if (++l_ts == p.S) {
        l_ts = 0;
        ++l_tr;
      }
    }
  };
  auto store_step = [&]() __attribute__((always_inline)) {
    *reinterpret_cast<uint4*>(sA + st0) = ra0;
    *reinterpret_cast<uint4*>(sA + st0 + 32 * 128) = ra1;
    *reinterpret_cast<uint4*>(sA + st0 + 64 * 128) = ra2;
    *reinterpret_cast<uint4*>(sA + st0 + 96 * 128) = ra3;
    *reinterpret_cast<uint4*>(sB + st0) = rb0;
    *reinterpret_cast<uint4*>(sB + st0 + 32 * 128) = rb1;
    *reinterpret_cast<uint4*>(sB + st0 + 64 * 128) = rb2;
    *reinterpret_cast<uint4*>(sB + st0 + 96 * 128) = rb3;
  };

  f32x4 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};

  load_step();
  store_step();
  __syncthreads();
  // fragment of row (base + li): lane group g takes 16-B chunks g and g + 4 (32 B = its share of the 128-element k-step)
  const int fkey = (li >> 1) & 7;
  const int fo0 = (g ^ fkey) * 16;
  const char* fa_base = sA + (wm * 64 + li) * 128;
  const char* fb_base = sB + (wn * 64 + li) * 128;
  auto frag = [&](const char* row) __attribute__((always_inline)) -> i32x8 {
    const uint4 lo = *reinterpret_cast<const uint4*>(row + fo0);
    const uint4 hi = *reinterpret_cast<const uint4*>(row + (fo0 ^ 64));
    i32x8 r = {(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y, (int)hi.z, (int)hi.w};
    return r;
  };
  for (int ks = 0; ks < nk; ++ks) {
    if (ks + 1 < nk) load_step();
    i32x8 fx[MI], fw[NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) fx[mi] = frag(fa_base + mi * 16 * 128);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) fw[ni] = frag(fb_base + ni * 16 * 128);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
        // weights as the A operand: lane (li, g) ends up with channels ni*16 + 4g .. +3 of pixel mi*16 + li.
        // cbsz = blgp = 0: both operands e4m3; every E8M0 block scale = 127 (2^0)
        acc[mi][ni] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fw[ni], fx[mi], acc[mi][ni], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    __syncthreads();
    if (ks + 1 < nk) store_step();
    __syncthreads();
  }

  const float descale = p.x_state[1] * p.w_state[1];
  if (p.bn_partial != nullptr) {
    float* red = reinterpret_cast<float*>(smem);  // [2 (wm)][2][BN]
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          const float v = acc[mi][ni][r] * descale;  // rows past Mo are exact zeros
          s1 += v;
          s2 += v * v;
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) {  // over the 16 pixel lanes of a lane group
          s1 += __shfl_xor(s1, o);
          s2 += __shfl_xor(s2, o);
        }
        if (li == 0) {
          const int c = wn * 64 + ni * 16 + 4 * g + r;
          red[(wm * 2 + 0) * BN + c] = s1;
          red[(wm * 2 + 1) * BN + c] = s2;
        }
      }
    }
    __syncthreads();
    {
      const int which = tid / BN, c = tid - which * BN;  // 256 threads = 2 x BN
      p.bn_partial[((long long)m_tile * 2 + which) * p.Cout + n0 + c] = red[(0 * 2 + which) * BN + c] + red[(1 * 2 + which) * BN + c];
    }
  }
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const long long pix = (long long)m0 + wm * 64 + mi * 16 + li;
    if (pix >= p.Mo) continue;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const f32x4 a = acc[mi][ni];
      uint2 v;
      v.x = pack_bf16x2(a[0] * descale, a[1] * descale);
      v.y = pack_bf16x2(a[2] * descale, a[3] * descale);
      *reinterpret_cast<uint2*>(p.y + pix * p.Cout + n0 + wn * 64 + ni * 16 + 4 * g) = v;
    }
  }
}

static inline int grid_for(long long n) {
  long long g = (n + 255) / 256;
  if (g > 4096) g = 4096;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace sh

namespace sh {  // conv_igemm.hip: the e4m3 variant of the 256 x 256 LDS-DMA kernel
bool igemm256_fp8_ok(const sh_conv_desc* d);
int igemm256_fp8_stat_rows(const sh_conv_desc* d);
int igemm256_fp8_fwd(const sh_conv_desc* d, const void* xq, const void* wq, const float* x_state, const float* w_state, void* y, float* bn_partial,
                     hipStream_t s);
}  // namespace sh

using namespace sh;

extern "C" {

int simhand_fp8_state_floats(int history) { return 4 + (history < 1 ? 1 : history); }

int simhand_fp8_amax(const void* x, int64_t count, int dtype, uint32_t* amax_bits, sh_stream_t stream) {
  SH_REQUIRE(x && amax_bits && count >= 1, "fp8_amax: bad arguments");
  SH_REQUIRE(dtype == SH_F32 || dtype == SH_BF16, "fp8_amax: source dtype %d", dtype);
  const int ve = dtype == SH_F32 ? 4 : 8;
  SH_REQUIRE(count % ve == 0, "fp8_amax: count must be a multiple of %d", ve);
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SH_PROF_MISC, s, 0, (double)count * (dtype == SH_F32 ? 4 : 2));
  if (dtype == SH_F32) fp8_amax_kernel<float><<<grid_for(count / ve), 256, 0, s>>>((const float*)x, count / ve, amax_bits);
  else fp8_amax_kernel<bf16_t><<<grid_for(count / ve), 256, 0, s>>>((const bf16_t*)x, count / ve, amax_bits);
  return check_launch("fp8_amax");
}

int simhand_fp8_scale_update(float* state, uint32_t* amax_new_bits, int history, float margin_pow2, int mode, sh_stream_t stream) {
  SH_REQUIRE(state && amax_new_bits && history >= 1 && margin_pow2 > 0.f, "fp8_scale_update: bad arguments");
  fp8_scale_update_kernel<<<1, 64, 0, (hipStream_t)stream>>>(state, amax_new_bits, history, margin_pow2, mode < 0 || mode > 2 ? 1 : mode);
  return check_launch("fp8_scale_update");
}

int simhand_fp8_quantize(const void* x, void* q, int64_t count, int dtype, const float* state, uint32_t* amax_bits, sh_stream_t stream) {
  SH_REQUIRE(x && q && state && count >= 1, "fp8_quantize: bad arguments");
  SH_REQUIRE(dtype == SH_F32 || dtype == SH_BF16, "fp8_quantize: source dtype %d", dtype);
  SH_REQUIRE(count % 16 == 0, "fp8_quantize: count must be a multiple of 16");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SH_PROF_MISC, s, 0, (double)count * ((dtype == SH_F32 ? 4 : 2) + 1));
  if (dtype == SH_F32) fp8_quantize_kernel<float><<<grid_for(count / 16), 256, 0, s>>>((const float*)x, (unsigned char*)q, count / 16, state, amax_bits);
  else fp8_quantize_kernel<bf16_t><<<grid_for(count / 16), 256, 0, s>>>((const bf16_t*)x, (unsigned char*)q, count / 16, state, amax_bits);
  return check_launch("fp8_quantize");
}

int simhand_fp8_pack_krsc(const float* w_oihw, void* q, int k, int c, int r, int s, const float* state, sh_stream_t stream) {
  SH_REQUIRE(w_oihw && q && state && k >= 1 && c >= 4 && c % 4 == 0 && r >= 1 && s >= 1, "fp8_pack_krsc: bad arguments");
  fp8_pack_krsc_kernel<<<grid_for((long long)k * c * r * s / 4), 256, 0, (hipStream_t)stream>>>(w_oihw, (unsigned char*)q, k, c, r, s, state);
  return check_launch("fp8_pack_krsc");
}

int simhand_conv2d_fwd_fp8_supported(const sh_conv_desc* d) {
  return d && d->cin % 128 == 0 && d->cout % 128 == 0 && (d->stride == 1 || d->stride == 2) && (long long)d->n * d->ho * d->wo < (1ll << 31) - 256 &&
         (long long)d->n * d->h * d->w * d->cin < (1ll << 31);
}

/* 1 where the fp8 forward is FASTER than the bf16 one in the engine (round-3 measurements at 2048 images, scripts/fp8_bench.py): the 3x3
 * layers that run on the e4m3 variant of the 256 x 256 kernel (500 -> 320 us at 256 ch @ 14^2).  The 128-channel 3x3 layers (128-row fp8
 * kernel: 957 vs 586 us) and the long-K 1x1 layers (the e4m3 copy of their 4x wider input costs more than the matrix time saved) do not. */
int simhand_conv2d_fwd_fp8_pays(const sh_conv_desc* d) {
  return d && simhand_conv2d_fwd_fp8_supported(d) && d->r == 3 && d->s == 3 && igemm256_fp8_ok(d) ? 1 : 0;
}

int simhand_conv2d_fwd_fp8(const sh_conv_desc* d, const void* x_q, const void* w_q, const float* x_state, const float* w_state, void* y,
                           float* bn_partial, sh_stream_t stream) {
  SH_REQUIRE(d && x_q && w_q && x_state && w_state && y, "conv2d_fwd_fp8: NULL pointer");
  SH_REQUIRE(simhand_conv2d_fwd_fp8_supported(d), "conv2d_fwd_fp8: needs cin, cout multiples of 128, stride 1 / 2 and < 2^31 input elements");
  SH_REQUIRE(d->ho == (d->h + 2 * d->pad - d->r) / d->stride + 1 && d->wo == (d->w + 2 * d->pad - d->s) / d->stride + 1,
             "conv2d_fwd_fp8: ho/wo inconsistent");
  Fp8Args a;
  a.x = (const unsigned char*)x_q; a.w = (const unsigned char*)w_q; a.y = (unsigned short*)y; a.bn_partial = bn_partial;
  a.x_state = x_state; a.w_state = w_state;
  a.Mo = (long long)d->n * d->ho * d->wo;
  a.Cin = d->cin; a.Cout = d->cout; a.R = d->r; a.S = d->s; a.stride = d->stride; a.pad = d->pad;
  a.H = d->h; a.W = d->w; a.Ho = d->ho; a.Wo = d->wo;
  a.m_tiles = ceil_div(a.Mo, 128); a.n_tiles = d->cout / 128;
  a.div_hw = make_fastdiv((unsigned)(d->ho * d->wo));
  a.div_w = make_fastdiv((unsigned)d->wo);
  hipStream_t s = (hipStream_t)stream;
  const double flops = 2.0 * (double)a.Mo * d->cout * d->cin * d->r * d->s;
  const double bytes = (double)d->n * d->h * d->w * d->cin + 2.0 * (double)a.Mo * d->cout + (double)d->cout * d->cin * d->r * d->s;
  ProfScope ps(SH_PROF_CONV_FWD, s, flops, bytes);
  route_hit(SH_ROUTE_FP8_FWD);
  if (igemm256_fp8_ok(d)) {  // the matrix-bound layers: e4m3 variant of the 256 x 256 LDS-DMA kernel
    route_hit(SH_ROUTE_IGEMM256_FWD);
    return igemm256_fp8_fwd(d, x_q, w_q, x_state, w_state, y, bn_partial, s);
  }
  igemm_fp8_fwd_kernel<<<a.m_tiles * a.n_tiles, 256, 0, s>>>(a);
  return check_launch("conv2d_fwd_fp8");
}

/* rows of the bn_partial buffer simhand_conv2d_fwd_fp8 fills: [rows][2][cout] */
int simhand_conv2d_fwd_fp8_stat_blocks(const sh_conv_desc* d) {
  if (!d) return 0;
  if (igemm256_fp8_ok(d)) return igemm256_fp8_stat_rows(d);
  return (int)(((long long)d->n * d->ho * d->wo + 127) / 128);
}

}  // extern "C"
