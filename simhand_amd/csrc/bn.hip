// Train-mode BatchNorm (2d over [N*H*W][C] NHWC, 1d over [N][C]) + ReLU + residual add,
// forward and backward.  HBM-bound streaming kernels: 16-B vector accesses, fp32 math,
// deterministic two-stage channel reductions (no float atomics).
//
// Replaces (reference): torch.nn.BatchNorm2d / BatchNorm1d (training statistics, eps 1e-5,
// momentum 0.1), ReLU and the residual add inside torchvision's BasicBlock / Bottleneck
// (src/models/resnet_model.py:13-58) and the projection head's BatchNorm1d + ReLU
// (src/models/unsupervised/simclr_model.py:30-31), plus their autograd backward.
#include "common.h"

#include <stdlib.h>

namespace sh {

constexpr int kChunk = 256;  // level-1 partial blocks folded per level-2 block

// rows-per-block plan shared by the column-reduction kernels
static inline void col_plan(int64_t m, int* rows_per_blk, int* nblk) {
  int64_t rpb = (m + 2047) / 2048;
  if (rpb < 64) rpb = 64;
  rpb = (rpb + 15) / 16 * 16;
  *rows_per_blk = (int)rpb;
  *nblk = (int)((m + rpb - 1) / rpb);
}

// ---- generic column reduction skeleton --------------------------------------------------
// F(row, cvec, out s1[VE], s2[VE]) accumulates VE channels of one row.
template <typename T, typename F>
__device__ __forceinline__ void column_reduce(int64_t m, int c, int rows_per_blk, float* partial, F body) {
  constexpr int VE = Vec16<T>::N;
  __shared__ float red[2][256][VE];
  const int cvecs = c / VE;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_blk;
  int64_t r1 = r0 + rows_per_blk;
  if (r1 > m) r1 = m;
  // thread layout: lanes along channel vectors first (coalesced), remaining threads along rows
  const int span = cvecs < 256 ? cvecs : 256;   // channel vectors handled concurrently
  const int rowlanes = 256 / span;              // >= 1
  const int cv_l = threadIdx.x % span, rl = threadIdx.x / span;
  for (int cv0 = 0; cv0 < cvecs; cv0 += span) {
    const int cv = cv0 + cv_l;
    float s1[VE], s2[VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) s1[e] = s2[e] = 0.f;
    if (cv < cvecs && rl < rowlanes)
      for (int64_t r = r0 + rl; r < r1; r += rowlanes) body(r, cv, s1, s2);
    __syncthreads();
#pragma unroll
    for (int e = 0; e < VE; ++e) {
      red[0][threadIdx.x][e] = s1[e];
      red[1][threadIdx.x][e] = s2[e];
    }
    __syncthreads();
    if (rl == 0 && cv < cvecs) {
      for (int j = 1; j < rowlanes; ++j)
#pragma unroll
        for (int e = 0; e < VE; ++e) {
          s1[e] += red[0][j * span + cv_l][e];
          s2[e] += red[1][j * span + cv_l][e];
        }
      float* o1 = partial + ((int64_t)blockIdx.x * 2 + 0) * c + cv * VE;
      float* o2 = partial + ((int64_t)blockIdx.x * 2 + 1) * c + cv * VE;
#pragma unroll
      for (int e = 0; e < VE; ++e) {
        o1[e] = s1[e];
        o2[e] = s2[e];
      }
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void bn_partial_stats_kernel(const T* __restrict__ y, int64_t m, int c, int rows_per_blk,
                                                               float* __restrict__ partial) {
  constexpr int VE = Vec16<T>::N;
  column_reduce<T>(m, c, rows_per_blk, partial, [&](int64_t r, int cv, float(&s1)[VE], float(&s2)[VE]) {
    float v[VE];
    Vec16<T>::load(y + r * c + cv * VE, v);
#pragma unroll
    for (int e = 0; e < VE; ++e) {
      s1[e] += v[e];
      s2[e] += v[e] * v[e];
    }
  });
}

// level-1 partial [nblk][2][c] float -> level-2 [nchunk][2][c] double
__device__ __forceinline__ void fold_partials_body(double (*red)[2][64], const float* __restrict__ partial, int nblk, int c, double* lvl2) {
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int ch = blockIdx.y * 64 + cl;
  const int b0 = blockIdx.x * kChunk;
  int b1 = b0 + kChunk;
  if (b1 > nblk) b1 = nblk;
  double s1 = 0.0, s2 = 0.0;
  if (ch < c) {
    double a1[4] = {0.0, 0.0, 0.0, 0.0}, a2[4] = {0.0, 0.0, 0.0, 0.0};  // independent chains: loads overlap
    int b = b0 + rl;
    for (; b + 12 < b1; b += 16) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        a1[j] += (double)partial[((int64_t)(b + 4 * j) * 2 + 0) * c + ch];
        a2[j] += (double)partial[((int64_t)(b + 4 * j) * 2 + 1) * c + ch];
      }
    }
    for (; b < b1; b += 4) {
      a1[0] += (double)partial[((int64_t)b * 2 + 0) * c + ch];
      a2[0] += (double)partial[((int64_t)b * 2 + 1) * c + ch];
    }
    s1 = (a1[0] + a1[1]) + (a1[2] + a1[3]);
    s2 = (a2[0] + a2[1]) + (a2[2] + a2[3]);
  }
  red[rl][0][cl] = s1;
  red[rl][1][cl] = s2;
  __syncthreads();
  if (rl == 0 && ch < c) {
    for (int j = 1; j < 4; ++j) {
      s1 += red[j][0][cl];
      s2 += red[j][1][cl];
    }
    lvl2[((int64_t)blockIdx.x * 2 + 0) * c + ch] = s1;
    lvl2[((int64_t)blockIdx.x * 2 + 1) * c + ch] = s2;
  }
}

__global__ __launch_bounds__(256) void fold_partials_kernel(const float* __restrict__ partial, int nblk, int c,
                                                            double* __restrict__ lvl2) {
  __shared__ double red[4][2][64];
  fold_partials_body(red, partial, nblk, c, lvl2);
}

// one channel's statistics from its level-2 sums
__device__ __forceinline__ void bn_finalize_channel(const int ch, const double* lvl2, int nchunk, int64_t m, int c,
                                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                                    const float* __restrict__ pre_bias, float eps, float momentum,
                                                    float* __restrict__ running_mean, float* __restrict__ running_var,
                                                    float* __restrict__ mean_o, float* __restrict__ invstd_o, float* __restrict__ scale_o,
                                                    float* __restrict__ shift_o) {
  double s1 = 0.0, s2 = 0.0;
  {
    double a1[4] = {0.0, 0.0, 0.0, 0.0}, a2[4] = {0.0, 0.0, 0.0, 0.0};
    int b = 0;
    for (; b + 3 < nchunk; b += 4) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        a1[j] += lvl2[((int64_t)(b + j) * 2 + 0) * c + ch];
        a2[j] += lvl2[((int64_t)(b + j) * 2 + 1) * c + ch];
      }
    }
    for (; b < nchunk; ++b) {
      a1[0] += lvl2[((int64_t)b * 2 + 0) * c + ch];
      a2[0] += lvl2[((int64_t)b * 2 + 1) * c + ch];
    }
    s1 = (a1[0] + a1[1]) + (a1[2] + a1[3]);
    s2 = (a2[0] + a2[1]) + (a2[2] + a2[3]);
  }
  const double mean = s1 / (double)m;
  double var = s2 / (double)m - mean * mean;  // biased (normalisation) variance
  if (var < 0.0) var = 0.0;
  const float invstd = (float)(1.0 / sqrt(var + (double)eps));
  const float g = gamma ? gamma[ch] : 1.0f, b = beta ? beta[ch] : 0.0f;
  const float sc = g * invstd;
  mean_o[ch] = (float)mean;
  invstd_o[ch] = invstd;
  scale_o[ch] = sc;
  shift_o[ch] = b - (float)mean * sc;
  if (running_mean) {
    const double mfull = mean + (pre_bias ? (double)pre_bias[ch] : 0.0);  // Linear bias shifts only the mean
    const double unbiased = m > 1 ? var * (double)m / (double)(m - 1) : var;
    running_mean[ch] = (1.0f - momentum) * running_mean[ch] + momentum * (float)mfull;
    running_var[ch] = (1.0f - momentum) * running_var[ch] + momentum * (float)unbiased;
  }
}

__global__ __launch_bounds__(256) void bn_finalize_kernel(const double* __restrict__ lvl2, int nchunk, int64_t m, int c,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          const float* __restrict__ pre_bias, float eps, float momentum,
                                                          float* __restrict__ running_mean, float* __restrict__ running_var,
                                                          int64_t* __restrict__ nbt, float* __restrict__ mean_o,
                                                          float* __restrict__ invstd_o, float* __restrict__ scale_o,
                                                          float* __restrict__ shift_o) {
  const int ch = blockIdx.x * 256 + threadIdx.x;
  if (ch == 0 && nbt) nbt[0] += 1;
  if (ch >= c) return;
  bn_finalize_channel(ch, lvl2, nchunk, m, c, gamma, beta, pre_bias, eps, momentum, running_mean, running_var, mean_o, invstd_o, scale_o, shift_o);
}

// Round 6: both levels in ONE launch.  Every block folds its chunk exactly as fold_partials_kernel does, publishes its level-2 row
// (agent-scope release: __threadfence + a relaxed atomic ticket) and leaves; the block that draws the LAST ticket acquires, resets the
// ticket for the next launch on this stream and runs bn_finalize_kernel's per-channel code for all channels -- the same additions in the
// same order, bit-identical statistics, one kernel boundary less on the critical path of every conv + BatchNorm unit.  `ticket` is a
// caller-owned uint32 that is ZERO before the first call and that no concurrent launch shares (one per stream); every launch leaves it zero.
__global__ __launch_bounds__(256) void bn_finalize_ticket_kernel(const float* __restrict__ partial, int nblk, int nchunk, int64_t m, int c,
                                                                 double* lvl2, unsigned* ticket, const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, const float* __restrict__ pre_bias,
                                                                 float eps, float momentum, float* __restrict__ running_mean,
                                                                 float* __restrict__ running_var, int64_t* __restrict__ nbt,
                                                                 float* __restrict__ mean_o, float* __restrict__ invstd_o,
                                                                 float* __restrict__ scale_o, float* __restrict__ shift_o) {
  __shared__ double red[4][2][64];
  __shared__ int s_last;
  fold_partials_body(red, partial, nblk, c, lvl2);
  __threadfence();  // this block's level-2 row is visible device-wide (across the XCDs' L2s) before its ticket is
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned total = gridDim.x * gridDim.y;
    const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = (t == total - 1u) ? 1 : 0;
  }
  __syncthreads();
  if (!s_last) return;
  __threadfence();  // acquire: the other blocks' rows
  if (threadIdx.x == 0) {
    __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (nbt) nbt[0] += 1;
  }
  for (int ch = threadIdx.x; ch < c; ch += 256)
    bn_finalize_channel(ch, lvl2, nchunk, m, c, gamma, beta, pre_bias, eps, momentum, running_mean, running_var, mean_o, invstd_o, scale_o, shift_o);
}

__global__ __launch_bounds__(256) void bn_eval_params_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             const float* __restrict__ rm, const float* __restrict__ rv, float eps, int c,
                                                             float* __restrict__ scale, float* __restrict__ shift) {
  const int ch = blockIdx.x * 256 + threadIdx.x;
  if (ch >= c) return;
  const float sc = (gamma ? gamma[ch] : 1.0f) / sqrtf(rv[ch] + eps);
  scale[ch] = sc;
  shift[ch] = (beta ? beta[ch] : 0.0f) - rm[ch] * sc;
}

// Streaming kernels: a thread owns ONE 16-B channel vector (its per-channel coefficients live in
// registers for the whole launch) and walks rows; lanes run along the channel axis first so a wave
// covers whole contiguous rows (coalesced), blocks split the row range.
// VE consecutive fp32 per-channel coefficients by 16-B loads (VE = 4 or 8; the arrays are 16-B aligned: channel counts are multiples of VE)
template <int VE>
__device__ __forceinline__ void ld_coef(const float* __restrict__ p, float (&o)[VE]) {
#pragma unroll
  for (int e = 0; e < VE; e += 4) {
    const float4 v = *reinterpret_cast<const float4*>(p + e);
    o[e] = v.x; o[e + 1] = v.y; o[e + 2] = v.z; o[e + 3] = v.w;
  }
}

struct RowWalk {
  int cv, rl, span, rowlanes;
  int64_t r0, r1;
};
template <int VE>
__device__ __forceinline__ RowWalk row_walk(int64_t m, int c) {
  RowWalk w;
  const int cvecs = c / VE;
  w.span = cvecs < 256 ? cvecs : 256;
  w.rowlanes = 256 / w.span;
  w.cv = threadIdx.x % w.span;
  w.rl = threadIdx.x / w.span;
  const int64_t per = (m + gridDim.x - 1) / gridDim.x;
  w.r0 = (int64_t)blockIdx.x * per;
  w.r1 = w.r0 + per < m ? w.r0 + per : m;
  return w;
}

template <typename T, bool NT>
__global__ __launch_bounds__(256) void bn_apply_kernel(const T* __restrict__ y, const float* __restrict__ scale,
                                                       const float* __restrict__ shift, const T* __restrict__ res, int relu,
                                                       T* __restrict__ a, uint8_t* __restrict__ mask, int64_t m, int c) {
  constexpr int VE = Vec16<T>::N;
  const RowWalk w = row_walk<VE>(m, c);
  if (w.rl >= w.rowlanes) return;
  const int cvecs = c / VE;
  for (int cv = w.cv; cv < c / VE; cv += w.span) {
    float sc[VE], sh[VE];
    ld_coef<VE>(scale + cv * VE, sc);
    ld_coef<VE>(shift + cv * VE, sh);
    auto finish = [&](int64_t r, const float(&v)[VE], const float(&q)[VE]) __attribute__((always_inline)) {
      float o[VE];
#pragma unroll
      for (int e = 0; e < VE; ++e) o[e] = v[e] * sc[e] + sh[e];
      if (res) {
#pragma unroll
        for (int e = 0; e < VE; ++e) o[e] += q[e];
      }
      if (relu) {
        if (mask) {  // 1 bit per element (bit e = channel cv*VE + e passed the ReLU): the backward reads this, not `a`
          unsigned bits = 0;
#pragma unroll
          for (int e = 0; e < VE; ++e) bits |= (o[e] > 0.f ? 1u : 0u) << e;
          mask[r * cvecs + cv] = (uint8_t)bits;
        }
#pragma unroll
        for (int e = 0; e < VE; ++e) o[e] = o[e] > 0.f ? o[e] : 0.f;
      }
      Vec16<T>::template store<NT>(a + r * c + cv * VE, o);
    };
    // U rows per trip: all loads are issued before the first dependent use (more bytes in flight per wave)
    constexpr int U = 4;
    int64_t r = w.r0 + w.rl;
    for (; r + (U - 1) * w.rowlanes < w.r1; r += U * w.rowlanes) {
      float v[U][VE], q[U][VE];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t off = (r + u * w.rowlanes) * c + cv * VE;
        Vec16<T>::template load<NT>(y + off, v[u]);
        if (res) Vec16<T>::template load<NT>(res + off, q[u]);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) finish(r + u * w.rowlanes, v[u], q[u]);
    }
    for (; r < w.r1; r += w.rowlanes) {
      float v[VE], q[VE];
      const int64_t off = r * c + cv * VE;
      Vec16<T>::template load<NT>(y + off, v);
      if (res) Vec16<T>::template load<NT>(res + off, q);
      finish(r, v, q);
    }
  }
}

// bn_apply for the unit in front of an fp8 convolution (BASELINE configs[4]): a = act(y * scale + shift) as bf16 AND its e4m3
// quantisation q = e4m3(clamp(a * q_state[0], +-448)) in the same pass, with max |a| of this tensor folded into amax_bits (the input of the
// delayed-scaling ring update) -- the stand-alone simhand_fp8_quantize pass (one read of a, one write of q) disappears.  q is taken from the
// bf16-ROUNDED value, so q and amax are bit-identical to quantising the stored a afterwards.
template <bool NT>
__global__ __launch_bounds__(256) void bn_apply_fp8_kernel(const bf16_t* __restrict__ y, const float* __restrict__ scale,
                                                           const float* __restrict__ shift, int relu, bf16_t* __restrict__ a,
                                                           unsigned char* __restrict__ q, const float* __restrict__ q_state,
                                                           unsigned* __restrict__ amax_bits, int64_t m, int c) {
  constexpr int VE = 8;
  const RowWalk w = row_walk<VE>(m, c);
  const float qs = q_state[0];
  float amax = 0.f;
  if (w.rl < w.rowlanes) {
    for (int cv = w.cv; cv < c / VE; cv += w.span) {
      float sc[VE], sh[VE];
#pragma unroll
      for (int e = 0; e < VE; ++e) {
        sc[e] = scale[cv * VE + e];
        sh[e] = shift[cv * VE + e];
      }
      auto finish = [&](int64_t r, const float(&v)[VE]) __attribute__((always_inline)) {
        unsigned pk[4];
        float o[VE];
#pragma unroll
        for (int e = 0; e < VE; ++e) {
          o[e] = v[e] * sc[e] + sh[e];
          if (relu) o[e] = o[e] > 0.f ? o[e] : 0.f;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) pk[e] = pack_bf16x2(o[2 * e], o[2 * e + 1]);
        st16<NT>(a + r * c + cv * VE, make_uint4(pk[0], pk[1], pk[2], pk[3]));
        float rr[VE];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          rr[2 * e] = h16_lo(pk[e]);
          rr[2 * e + 1] = h16_hi(pk[e]);
        }
#pragma unroll
        for (int e = 0; e < VE; ++e) amax = fmaxf(amax, fabsf(rr[e]));
        uint2 qq;
        qq.x = fp8_pack4(rr[0] * qs, rr[1] * qs, rr[2] * qs, rr[3] * qs);
        qq.y = fp8_pack4(rr[4] * qs, rr[5] * qs, rr[6] * qs, rr[7] * qs);
        *reinterpret_cast<uint2*>(q + r * c + cv * VE) = qq;
      };
      constexpr int U = 4;
      int64_t r = w.r0 + w.rl;
      for (; r + (U - 1) * w.rowlanes < w.r1; r += U * w.rowlanes) {
        float v[U][VE];
#pragma unroll
        for (int u = 0; u < U; ++u) Vec16<bf16_t>::template load<NT>(y + (r + u * w.rowlanes) * c + cv * VE, v[u]);
#pragma unroll
        for (int u = 0; u < U; ++u) finish(r + u * w.rowlanes, v[u]);
      }
      for (; r < w.r1; r += w.rowlanes) {
        float v[VE];
        Vec16<bf16_t>::template load<NT>(y + r * c + cv * VE, v);
        finish(r, v);
      }
    }
  }
  // one atomic per BLOCK at most, and only when it would raise the running maximum; the bit pattern of a non-negative float is
  // order-preserving
  amax = wave_max(amax);
  __shared__ float red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = amax;
  __syncthreads();
  if (threadIdx.x == 0 && amax_bits != nullptr) {
    const unsigned bits = __float_as_uint(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])));
    if (bits > __atomic_load_n(amax_bits, __ATOMIC_RELAXED)) atomicMax(amax_bits, bits);
  }
}

template <typename T, bool NT>
__global__ __launch_bounds__(256) void bn_bwd_partial_kernel(const T* __restrict__ da, const T* __restrict__ a,
                                                             const T* __restrict__ y, const float* __restrict__ mean,
                                                             const float* __restrict__ invstd,
                                                             const float* __restrict__ scale, const float* __restrict__ shift,
                                                             int relu, int64_t m, int c, int rows_per_blk,
                                                             float* __restrict__ partial) {
  constexpr int VE = Vec16<T>::N;
  // per-channel coefficients of the channel vector this thread walks, in registers (16-B loads, refreshed when cv changes: c / VE > 256 only)
  float sc[VE], sh[VE], mu[VE], is[VE];
  int cv_held = -1;
  column_reduce<T>(m, c, rows_per_blk, partial, [&](int64_t r, int cv, float(&s1)[VE], float(&s2)[VE]) {
    if (cv != cv_held) {
      cv_held = cv;
      ld_coef<VE>(mean + cv * VE, mu);
      ld_coef<VE>(invstd + cv * VE, is);
      if (relu == 2) {
        ld_coef<VE>(scale + cv * VE, sc);
        ld_coef<VE>(shift + cv * VE, sh);
      }
    }
    float g[VE], yy[VE];
    Vec16<T>::template load<NT>(da + r * c + cv * VE, g);
    Vec16<T>::template load<NT>(y + r * c + cv * VE, yy);
    if (relu == 1) {
      float aa[VE];
      Vec16<T>::load(a + r * c + cv * VE, aa);
#pragma unroll
      for (int e = 0; e < VE; ++e) g[e] = aa[e] > 0.f ? g[e] : 0.f;
    } else if (relu == 2) {  // no residual: the ReLU input is recomputed from y (same fp32 expression as bn_apply)
#pragma unroll
      for (int e = 0; e < VE; ++e) g[e] = yy[e] * sc[e] + sh[e] > 0.f ? g[e] : 0.f;
    } else if (relu == 3) {  // residual unit: 1-bit mask written by bn_apply (`a` points to it)
      const unsigned bits = reinterpret_cast<const uint8_t*>(a)[r * (c / VE) + cv];
#pragma unroll
      for (int e = 0; e < VE; ++e) g[e] = (bits >> e) & 1u ? g[e] : 0.f;
    }
#pragma unroll
    for (int e = 0; e < VE; ++e) {
      const float xh = (yy[e] - mu[e]) * is[e];
      s1[e] += g[e];
      s2[e] += g[e] * xh;
    }
  });
}

// dbeta = sum g, dgamma = sum g*xhat over <= ~2048 partial blocks: 16 channels x 64 row lanes per block, four
// independent accumulator pairs per thread so the (L2-resident) partial loads overlap instead of forming one
// dependent chain; fixed summation order (deterministic)
// TIn = float (level-1 partials) or double (level-2 slices).  gridDim.y > 1: slice blockIdx.y of the partial rows is folded
// to lvl2[blockIdx.y][2][c] (doubles) for a second launch -- the fused dgrad epilogues emit one partial row per 128-row
// tile (50 176 rows at 2048 x 56 x 56), far too many for c / 16 blocks.
template <typename TIn>
__global__ __launch_bounds__(1024) void bn_bwd_finalize_kernel(const TIn* __restrict__ partial, int nblk, int c,
                                                               float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                               const float* __restrict__ mean = nullptr,
                                                               const float* __restrict__ invstd = nullptr,
                                                               double* __restrict__ lvl2 = nullptr,
                                                               const float* __restrict__ gamma = nullptr, float inv_m = 0.f,
                                                               float* __restrict__ coefs = nullptr) {
  __shared__ double red[64][2][16];
  const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int ch = blockIdx.x * 16 + cl;
  const int per = (nblk + gridDim.y - 1) / gridDim.y;
  const int b_begin = blockIdx.y * per;
  const int b_end = b_begin + per < nblk ? b_begin + per : nblk;
  double a1[4] = {0.0, 0.0, 0.0, 0.0}, a2[4] = {0.0, 0.0, 0.0, 0.0};
  if (ch < c) {
    int b = b_begin + rl;
    for (; b + 3 * 64 < b_end; b += 4 * 64) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        a1[j] += (double)partial[((int64_t)(b + j * 64) * 2 + 0) * c + ch];
        a2[j] += (double)partial[((int64_t)(b + j * 64) * 2 + 1) * c + ch];
      }
    }
    for (; b < b_end; b += 64) {
      a1[0] += (double)partial[((int64_t)b * 2 + 0) * c + ch];
      a2[0] += (double)partial[((int64_t)b * 2 + 1) * c + ch];
    }
  }
  double s1 = (a1[0] + a1[1]) + (a1[2] + a1[3]), s2 = (a2[0] + a2[1]) + (a2[2] + a2[3]);
  red[rl][0][cl] = s1;
  red[rl][1][cl] = s2;
  __syncthreads();
  if (rl != 0 || ch >= c) return;
  double t1 = 0.0, t2 = 0.0;
  for (int j = 0; j < 64; ++j) {  // fixed order: deterministic
    t1 += red[j][0][cl];
    t2 += red[j][1][cl];
  }
  if (lvl2 != nullptr) {
    lvl2[((int64_t)blockIdx.y * 2 + 0) * c + ch] = t1;
    lvl2[((int64_t)blockIdx.y * 2 + 1) * c + ch] = t2;
    return;
  }
  const float db = (float)t1;
  // raw sums (sum g, sum g*y) from the fused dgrad epilogue: sum g*xhat = invstd * (sum g*y - mean * sum g)
  const float dg = mean != nullptr ? (float)((double)invstd[ch] * (t2 - (double)mean[ch] * t1)) : (float)t2;
  dbeta[ch] = db;
  dgamma[ch] = dg;
  if (coefs != nullptr) {  // round 6: bn_bwd_coefs_kernel's arithmetic on the sums just rounded (coefs = [3][c]: A, B, C of dy = A g - B y + C)
    const float a = __fmul_rn(gamma[ch], invstd[ch]);
    const float b = __fmul_rn(__fmul_rn(__fmul_rn(a, invstd[ch]), dg), inv_m);
    coefs[ch] = a;
    coefs[c + ch] = b;
    coefs[2 * c + ch] = __fadd_rn(__fmul_rn(__fmul_rn(-a, db), inv_m), __fmul_rn(mean[ch], b));
  }
}

template <typename T, bool NT>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ da, const T* __restrict__ a,
                                                           const T* __restrict__ y, const float* __restrict__ mean,
                                                           const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                           const float* __restrict__ dgamma, const float* __restrict__ dbeta,
                                                           const float* __restrict__ scale, const float* __restrict__ shift,
                                                           int relu, T* __restrict__ dy, T* __restrict__ dres, int64_t m, int c,
                                                           float inv_m) {
  constexpr int VE = Vec16<T>::N;
  const RowWalk w = row_walk<VE>(m, c);
  if (w.rl >= w.rowlanes) return;
  for (int cv = w.cv; cv < c / VE; cv += w.span) {
    // dy = A*(g - k2) - xhat*k3 with xhat = (y - mu)*is ; A = gamma*is, k2 = dbeta/M, k3 = A*dgamma/M
    float mu[VE], is[VE], A[VE], k2[VE], k3[VE], sc[VE], sh[VE];
    {  // per-channel coefficients by 16-B loads (a block's prologue: with many short blocks it is a visible share of its life)
      float ga[VE], dg_[VE], db_[VE];
      ld_coef<VE>(mean + cv * VE, mu);
      ld_coef<VE>(invstd + cv * VE, is);
      ld_coef<VE>(dgamma + cv * VE, dg_);
      ld_coef<VE>(dbeta + cv * VE, db_);
      if (gamma) ld_coef<VE>(gamma + cv * VE, ga);
      if (relu == 2) {
        ld_coef<VE>(scale + cv * VE, sc);
        ld_coef<VE>(shift + cv * VE, sh);
      }
#pragma unroll
      for (int e = 0; e < VE; ++e) {
        A[e] = (gamma ? ga[e] : 1.0f) * is[e];
        k2[e] = db_[e] * inv_m;
        k3[e] = A[e] * dg_[e] * inv_m;
        if (relu != 2) sc[e] = sh[e] = 0.f;
      }
    }
    for (int64_t r = w.r0 + w.rl; r < w.r1; r += w.rowlanes) {
      const int64_t off = r * c + cv * VE;
      float g[VE], yy[VE], o[VE];
      Vec16<T>::template load<NT>(da + off, g);
      Vec16<T>::template load<NT>(y + off, yy);
      if (relu == 1) {
        float aa[VE];
        Vec16<T>::load(a + off, aa);
#pragma unroll
        for (int e = 0; e < VE; ++e) g[e] = aa[e] > 0.f ? g[e] : 0.f;
      } else if (relu == 2) {
#pragma unroll
        for (int e = 0; e < VE; ++e) g[e] = yy[e] * sc[e] + sh[e] > 0.f ? g[e] : 0.f;
      } else if (relu == 3) {
        const unsigned bits = reinterpret_cast<const uint8_t*>(a)[r * (c / VE) + cv];
#pragma unroll
        for (int e = 0; e < VE; ++e) g[e] = (bits >> e) & 1u ? g[e] : 0.f;
      }
#pragma unroll
      for (int e = 0; e < VE; ++e) o[e] = A[e] * (g[e] - k2[e]) - (yy[e] - mu[e]) * is[e] * k3[e];
      Vec16<T>::template store<NT>(dy + off, o);
      if (dres) Vec16<T>::template store<NT>(dres + off, g);
    }
  }
}


// bn_bwd_apply (bf16) that also emits the e4m3 codes of dy (from its bf16-rounded value) and folds max |dy| into amax_bits: the operand
// of the fp8 data gradient leaves the pass that produces dy (see bn_apply_fp8_kernel).
template <bool NT>
__global__ __launch_bounds__(256) void bn_bwd_apply_fp8_kernel(const bf16_t* __restrict__ da, const bf16_t* __restrict__ a,
                                                               const bf16_t* __restrict__ y, const float* __restrict__ mean,
                                                               const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                               const float* __restrict__ dgamma, const float* __restrict__ dbeta,
                                                               const float* __restrict__ scale, const float* __restrict__ shift, int relu,
                                                               bf16_t* __restrict__ dy, unsigned char* __restrict__ q,
                                                               const float* __restrict__ q_state, unsigned* __restrict__ amax_bits, int64_t m,
                                                               int c, float inv_m) {
  constexpr int VE = 8;
  const RowWalk w = row_walk<VE>(m, c);
  const float qs = q_state[0];
  float amax = 0.f;
  if (w.rl < w.rowlanes) {
    for (int cv = w.cv; cv < c / VE; cv += w.span) {
      float mu[VE], is[VE], A[VE], k2[VE], k3[VE], sc[VE], sh[VE];
#pragma unroll
      for (int e = 0; e < VE; ++e) {
        const int ch = cv * VE + e;
        mu[e] = mean[ch];
        is[e] = invstd[ch];
        A[e] = (gamma ? gamma[ch] : 1.0f) * is[e];
        k2[e] = dbeta[ch] * inv_m;
        k3[e] = A[e] * dgamma[ch] * inv_m;
        sc[e] = relu == 2 ? scale[ch] : 0.f;
        sh[e] = relu == 2 ? shift[ch] : 0.f;
      }
      for (int64_t r = w.r0 + w.rl; r < w.r1; r += w.rowlanes) {
        const int64_t off = r * c + cv * VE;
        float g[VE], yy[VE];
        Vec16<bf16_t>::template load<NT>(da + off, g);
        Vec16<bf16_t>::template load<NT>(y + off, yy);
        if (relu == 1) {
          float aa[VE];
          Vec16<bf16_t>::load(a + off, aa);
#pragma unroll
          for (int e = 0; e < VE; ++e) g[e] = aa[e] > 0.f ? g[e] : 0.f;
        } else if (relu == 2) {
#pragma unroll
          for (int e = 0; e < VE; ++e) g[e] = yy[e] * sc[e] + sh[e] > 0.f ? g[e] : 0.f;
        } else if (relu == 3) {
          const unsigned bits = reinterpret_cast<const uint8_t*>(a)[r * (c / VE) + cv];
#pragma unroll
          for (int e = 0; e < VE; ++e) g[e] = (bits >> e) & 1u ? g[e] : 0.f;
        }
        unsigned pk[4];
        float rr[VE];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float o0 = A[2 * e] * (g[2 * e] - k2[2 * e]) - (yy[2 * e] - mu[2 * e]) * is[2 * e] * k3[2 * e];
          const float o1 = A[2 * e + 1] * (g[2 * e + 1] - k2[2 * e + 1]) - (yy[2 * e + 1] - mu[2 * e + 1]) * is[2 * e + 1] * k3[2 * e + 1];
          pk[e] = pack_bf16x2(o0, o1);
          rr[2 * e] = h16_lo(pk[e]);
          rr[2 * e + 1] = h16_hi(pk[e]);
        }
        st16<NT>(dy + off, make_uint4(pk[0], pk[1], pk[2], pk[3]));
#pragma unroll
        for (int e = 0; e < VE; ++e) amax = fmaxf(amax, fabsf(rr[e]));
        uint2 qq;
        qq.x = fp8_pack4(rr[0] * qs, rr[1] * qs, rr[2] * qs, rr[3] * qs);
        qq.y = fp8_pack4(rr[4] * qs, rr[5] * qs, rr[6] * qs, rr[7] * qs);
        *reinterpret_cast<uint2*>(q + off) = qq;
      }
    }
  }
  amax = wave_max(amax);
  __shared__ float red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = amax;
  __syncthreads();
  if (threadIdx.x == 0 && amax_bits != nullptr) {
    const unsigned bits = __float_as_uint(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])));
    if (bits > __atomic_load_n(amax_bits, __ATOMIC_RELAXED)) atomicMax(amax_bits, bits);
  }
}


// ---- stem block: BatchNorm + ReLU + MaxPool(3, 2, 1) fused, forward and backward ------------------------------------------
// Replaces (reference): bn1 -> relu -> maxpool of torchvision's ResNet stem (src/models/resnet_model.py:13-26) and
// their autograd backward.  The 112x112x64 activation between ReLU and the pool (3.3 GB at 2048 images, bf16) is never
// written: forward reads the raw conv output y through the pool windows (L2 absorbs the 2.25x window overlap);
// backward gathers the pooled gradient through the stored winner index inside the BatchNorm-backward passes instead
// of materialising the un-pooled gradient.  Values are rounded exactly where the unfused kernels store them.
template <typename T> __device__ __forceinline__ float round_as(float v);
template <> __device__ __forceinline__ float round_as<float>(float v) { return v; }
template <> __device__ __forceinline__ float round_as<bf16_t>(float v) { return bf16_to_f32(f32_to_bf16(v)); }

template <typename T>
__global__ __launch_bounds__(256) void bn_relu_maxpool_fwd_kernel(const T* __restrict__ y, const float* __restrict__ scale,
                                                                  const float* __restrict__ shift, T* __restrict__ out,
                                                                  uint8_t* __restrict__ idx, T* __restrict__ ywin, int n, int h, int w, int c,
                                                                  int ho, int wo, FastDiv div_cv, FastDiv div_wo, FastDiv div_ho) {
  constexpr int VE = Vec16<T>::N;
  const int cvecs = c / VE;
  const unsigned total = (unsigned)n * ho * wo * cvecs;  // < 2^31 (checked on the host)
  // consecutive output rows share an input row: keep them on ONE XCD's L2 (blocks go to XCDs round-robin, so block b
  // takes the b / 8-th slot of XCD b % 8's contiguous eighth of each grid-stride window)
  const unsigned lb = (gridDim.x & 7u) == 0 ? (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3) : blockIdx.x;
  for (unsigned i = lb * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
    unsigned t = fdiv(i, div_cv);
    const int cv = (int)(i - t * (unsigned)cvecs);
    unsigned q = fdiv(t, div_wo);
    const int ow = (int)(t - q * (unsigned)wo);
    const unsigned im = fdiv(q, div_ho);
    const int oh = (int)(q - im * (unsigned)ho);
    const int img = (int)im;
    float sc[VE], sh[VE], best[VE], braw[VE];
    int bi[VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) {
      sc[e] = scale[cv * VE + e];
      sh[e] = shift[cv * VE + e];
      best[e] = -INFINITY;
      braw[e] = 0.f;
      bi[e] = 0;
    }
    // scan order kh then kw, strict '>' (first maximum wins) like ATen's max_pool2d.  All nine taps are requested before the
    // first one is used, branch-free (taps outside the image read a clamped address and are skipped by predicate): conditional
    // loads would make the compiler wait for each tap in turn (nine exposed round trips per thread)
    uint4 tv[9];
    bool tok[9];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int ih = oh * 2 - 1 + kh;
      const int ihc = min(max(ih, 0), h - 1);
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int iw = ow * 2 - 1 + kw;
        const int iwc = min(max(iw, 0), w - 1);
        tok[kh * 3 + kw] = (unsigned)ih < (unsigned)h && (unsigned)iw < (unsigned)w;
        tv[kh * 3 + kw] = ld16<false>(y + (((int64_t)img * h + ihc) * w + iwc) * c + cv * VE);
      }
    }
#pragma unroll
    for (int t9 = 0; t9 < 9; ++t9) {
      float v[VE];
      if constexpr (VE == 8) {
        const unsigned w4[4] = {tv[t9].x, tv[t9].y, tv[t9].z, tv[t9].w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          v[2 * q] = h16_lo(w4[q]);
          v[2 * q + 1] = h16_hi(w4[q]);
        }
      } else {
        v[0] = __uint_as_float(tv[t9].x); v[1] = __uint_as_float(tv[t9].y); v[2] = __uint_as_float(tv[t9].z); v[3] = __uint_as_float(tv[t9].w);
      }
#pragma unroll
      for (int e = 0; e < VE; ++e) {
        float a = v[e] * sc[e] + sh[e];
        a = round_as<T>(a > 0.f ? a : 0.f);  // the activation as bn_apply would have stored it
        if (tok[t9] && (a > best[e] || a != a)) {
          best[e] = a;
          braw[e] = v[e];
          bi[e] = t9;
        }
      }
    }
    Vec16<T>::template store<true>(out + (size_t)i * VE, best);
    if (ywin != nullptr) Vec16<T>::template store<true>(ywin + (size_t)i * VE, braw);  // raw conv output of the winning tap
    // winner taps: one 4- / 8-byte store
    unsigned lo = 0, hi = 0;
#pragma unroll
    for (int e = 0; e < VE; ++e) {
      if (e < 4) lo |= (unsigned)bi[e] << (8 * e);
      else hi |= (unsigned)bi[e] << (8 * (e - 4));
    }
    if (VE == 8) *reinterpret_cast<uint2*>(idx + (size_t)i * VE) = make_uint2(lo, hi);
    else *reinterpret_cast<unsigned*>(idx + (size_t)i * VE) = lo;
  }
}

// gradient w.r.t. the (never stored) post-ReLU activation at input pixel (img, ih, iw): sum of dz over the <= 4 windows
// whose winner is this pixel, rounded as maxpool_bwd would have stored it
template <typename T, int VE>
__device__ __forceinline__ void pool_gather(const T* __restrict__ dz, const uint8_t* __restrict__ idx, int img, int ih, int iw,
                                            int cv, int cvecs, int ho, int wo, float (&g)[VE]) {
#pragma unroll
  for (int e = 0; e < VE; ++e) g[e] = 0.f;
  for (int oh = ih / 2; oh <= (ih + 1) / 2; ++oh) {
    const int kh = ih + 1 - 2 * oh;
    if (oh >= ho) continue;
    for (int ow = iw / 2; ow <= (iw + 1) / 2; ++ow) {
      const int kw = iw + 1 - 2 * ow;
      if (ow >= wo) continue;
      const int64_t o = ((((int64_t)img * ho + oh) * wo + ow) * cvecs + cv) * VE;
      float d[VE];
      Vec16<T>::load(dz + o, d);
      const uint8_t* ip = idx + o;
      const int me = kh * 3 + kw;
      uint8_t wi[VE];
      if (VE == 8) {
        const uint2 q = *reinterpret_cast<const uint2*>(ip);
#pragma unroll
        for (int e = 0; e < VE; ++e) wi[e] = (uint8_t)(((e < 4 ? q.x : q.y) >> (8 * (e & 3))) & 0xffu);
      } else {
        const unsigned q = *reinterpret_cast<const unsigned*>(ip);
#pragma unroll
        for (int e = 0; e < VE; ++e) wi[e] = (uint8_t)((q >> (8 * e)) & 0xffu);
      }
#pragma unroll
      for (int e = 0; e < VE; ++e) g[e] += wi[e] == me ? d[e] : 0.f;
    }
  }
#pragma unroll
  for (int e = 0; e < VE; ++e) g[e] = round_as<T>(g[e]);
}

template <typename T>
__global__ __launch_bounds__(256) void pool_bn_bwd_partial_kernel(const T* __restrict__ dz, const uint8_t* __restrict__ idx,
                                                                  const T* __restrict__ y, const float* __restrict__ mean,
                                                                  const float* __restrict__ invstd, const float* __restrict__ scale,
                                                                  const float* __restrict__ shift, int n, int h, int w, int c,
                                                                  int ho, int wo, int rows_per_blk, float* __restrict__ partial,
                                                                  FastDiv div_w, FastDiv div_h) {
  constexpr int VE = Vec16<T>::N;
  const int cvecs = c / VE;
  const int64_t m = (int64_t)n * h * w;
  column_reduce<T>(m, c, rows_per_blk, partial, [&](int64_t r, int cv, float(&s1)[VE], float(&s2)[VE]) {
    const unsigned t = fdiv((unsigned)r, div_w);  // m < 2^31 (checked on the host)
    const int iw = (int)((unsigned)r - t * (unsigned)w);
    const unsigned im = fdiv(t, div_h);
    const int ih = (int)(t - im * (unsigned)h), img = (int)im;
    float g[VE], yy[VE];
    pool_gather<T, VE>(dz, idx, img, ih, iw, cv, cvecs, ho, wo, g);
    Vec16<T>::template load<true>(y + r * c + cv * VE, yy);
#pragma unroll
    for (int e = 0; e < VE; ++e) {
      const int ch = cv * VE + e;
      const float gv = yy[e] * scale[ch] + shift[ch] > 0.f ? g[e] : 0.f;
      s1[e] += gv;
      s2[e] += gv * ((yy[e] - mean[ch]) * invstd[ch]);
    }
  });
}

template <typename T>
__global__ __launch_bounds__(256) void pool_bn_bwd_apply_kernel(const T* __restrict__ dz, const uint8_t* __restrict__ idx,
                                                                const T* __restrict__ y, const float* __restrict__ mean,
                                                                const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                                const float* __restrict__ dgamma, const float* __restrict__ dbeta,
                                                                const float* __restrict__ scale, const float* __restrict__ shift,
                                                                T* __restrict__ dy, int n, int h, int w, int c, int ho, int wo,
                                                                float inv_m, FastDiv div_w, FastDiv div_h) {
  constexpr int VE = Vec16<T>::N;
  const int cvecs = c / VE;
  const int64_t m = (int64_t)n * h * w;
  const RowWalk rw = row_walk<VE>(m, c);
  if (rw.rl >= rw.rowlanes) return;
  for (int cv = rw.cv; cv < cvecs; cv += rw.span) {
    float mu[VE], is[VE], A[VE], k2[VE], k3[VE], sc[VE], sh[VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) {
      const int ch = cv * VE + e;
      mu[e] = mean[ch];
      is[e] = invstd[ch];
      A[e] = (gamma ? gamma[ch] : 1.0f) * is[e];
      k2[e] = dbeta[ch] * inv_m;
      k3[e] = A[e] * dgamma[ch] * inv_m;
      sc[e] = scale[ch];
      sh[e] = shift[ch];
    }
    for (int64_t r = rw.r0 + rw.rl; r < rw.r1; r += rw.rowlanes) {
      const unsigned t = fdiv((unsigned)r, div_w);
      const int iw = (int)((unsigned)r - t * (unsigned)w);
      const unsigned im = fdiv(t, div_h);
      const int ih = (int)(t - im * (unsigned)h), img = (int)im;
      float g[VE], yy[VE], o[VE];
      pool_gather<T, VE>(dz, idx, img, ih, iw, cv, cvecs, ho, wo, g);
      Vec16<T>::template load<true>(y + r * c + cv * VE, yy);
#pragma unroll
      for (int e = 0; e < VE; ++e) {
        const float gv = yy[e] * sc[e] + sh[e] > 0.f ? g[e] : 0.f;
        o[e] = A[e] * (gv - k2[e]) - (yy[e] - mu[e]) * is[e] * k3[e];
      }
      Vec16<T>::template store<true>(dy + r * c + cv * VE, o);
    }
  }
}

// out = x gated by a ReLU bit mask [row][c / VE] (bit e of byte (row, cv) = channel cv*VE + e)
template <typename T>
__global__ __launch_bounds__(256) void apply_bitmask_kernel(const T* __restrict__ x, const uint8_t* __restrict__ mask, T* __restrict__ out,
                                                            int64_t nvec) {
  constexpr int VE = Vec16<T>::N;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
    float v[VE];
    Vec16<T>::load(x + i * VE, v);
    const unsigned bits = mask[i];
#pragma unroll
    for (int e = 0; e < VE; ++e) v[e] = (bits >> e) & 1u ? v[e] : 0.f;
    Vec16<T>::store(out + i * VE, v);
  }
}



// 2x2-block form of the stem backward (even h, w; c / VE a power of two <= 256): a thread owns the input pixels
// (2a..2a+1, 2b..2b+1) of one channel vector.  They are covered by exactly the four windows (a..a+1, b..b+1), so dz and the
// winner index are loaded 4 times per 4 pixels instead of 9 (pool_gather).  APPLY = false: BatchNorm-backward partial
// sums (one [2][c] row per block, same meaning as pool_bn_bwd_partial_kernel); APPLY = true: dy.
template <typename T, bool APPLY>
__global__ __launch_bounds__(256) void pool_bn_bwd_blk_kernel(const T* __restrict__ dz, const uint8_t* __restrict__ idx,
                                                              const T* __restrict__ y, const float* __restrict__ mean,
                                                              const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                              const float* __restrict__ dgamma, const float* __restrict__ dbeta,
                                                              const float* __restrict__ scale, const float* __restrict__ shift,
                                                              T* __restrict__ dy, float* __restrict__ partial, int n, int h, int w,
                                                              int c, int ho, int wo, float inv_m, FastDiv div_wb, FastDiv div_hb,
                                                              int blocks_per_cta) {
  constexpr int VE = Vec16<T>::N;
  __shared__ float red[2][256][VE];
  const int cvecs = c / VE;
  const int cv = threadIdx.x & (cvecs - 1);
  const int lanes = 256 / cvecs;             // 2x2 blocks handled concurrently by a CTA
  const int bl = threadIdx.x / cvecs;
  const int hb = h / 2, wb = w / 2;
  const unsigned nblocks = (unsigned)n * hb * wb;  // < 2^31 (checked on the host)
  // APPLY: dy = A (g - dbeta/M) - xhat A dgamma/M  =  cA g + y cP + cQ  (three coefficient vectors instead of five);
  // statistics pass: mean / invstd for xhat.  Winner taps stay packed (one byte per channel) until they are compared.
  float mu[VE], is[VE], cA[VE], cP[VE], cQ[VE], sc[VE], sh[VE], s1[VE], s2[VE];
#pragma unroll
  for (int e = 0; e < VE; ++e) {
    const int ch = cv * VE + e;
    sc[e] = scale[ch];
    sh[e] = shift[ch];
    s1[e] = s2[e] = 0.f;
    if (APPLY) {
      const float m_ = mean[ch], i_ = invstd[ch];
      const float a_ = (gamma ? gamma[ch] : 1.0f) * i_;
      const float k2 = dbeta[ch] * inv_m, k3 = a_ * dgamma[ch] * inv_m;
      cA[e] = a_;
      cP[e] = -i_ * k3;
      cQ[e] = m_ * i_ * k3 - a_ * k2;
    } else {
      mu[e] = mean[ch];
      is[e] = invstd[ch];
    }
  }
  const unsigned b_begin = blockIdx.x * (unsigned)blocks_per_cta;
  unsigned b_end = b_begin + blocks_per_cta;
  if (b_end > nblocks) b_end = nblocks;
  for (unsigned q = b_begin + bl; q < b_end; q += lanes) {
    const unsigned t = fdiv(q, div_wb);
    const int b = (int)(q - t * (unsigned)wb);
    const unsigned img = fdiv(t, div_hb);
    const int a = (int)(t - img * (unsigned)hb);
    // the four windows (a + i, b + j): gradient and (packed) winner taps; 0xff = no such window
    float d[2][2][VE];
    unsigned wlo[2][2], whi[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        // branch-free: a window past the edge reads the clamped one and is voided by predicate (conditional loads would make the
        // compiler's merged vmcnt state wait for each of them in turn)
        const bool ok = a + i < ho && b + j < wo;
        const int ai = a + i < ho ? a + i : ho - 1, bj = b + j < wo ? b + j : wo - 1;
        const int64_t o = ((((int64_t)img * ho + ai) * wo + bj) * cvecs + cv) * VE;
        Vec16<T>::load(dz + o, d[i][j]);
        if (VE == 8) {
          const uint2 u = *reinterpret_cast<const uint2*>(idx + o);
          wlo[i][j] = ok ? u.x : 0xffffffffu;
          whi[i][j] = ok ? u.y : 0xffffffffu;
        } else {
          const unsigned u = *reinterpret_cast<const unsigned*>(idx + o);
          wlo[i][j] = ok ? u : 0xffffffffu;
          whi[i][j] = 0xffffffffu;
        }
      }
    // all four y rows are requested before any of the gather arithmetic (12 loads in flight per thread)
    float yq[2][2][VE];
#pragma unroll
    for (int pi = 0; pi < 2; ++pi)
#pragma unroll
      for (int pj = 0; pj < 2; ++pj)
        Vec16<T>::template load<true>(y + (((int64_t)img * h + 2 * a + pi) * w + 2 * b + pj) * c + cv * VE, yq[pi][pj]);
    // input pixel (2a + pi, 2b + pj): window (a + i, b + j) reaches it with tap kh = 2 pi + 1 - 2 i, kw = 2 pj + 1 - 2 j
#pragma unroll
    for (int pi = 0; pi < 2; ++pi)
#pragma unroll
      for (int pj = 0; pj < 2; ++pj) {
        float gq[VE];
#pragma unroll
        for (int e = 0; e < VE; ++e) gq[e] = 0.f;
#pragma unroll
        for (int i = 0; i <= pi; ++i)
#pragma unroll
          for (int j = 0; j <= pj; ++j) {
            const unsigned me = (unsigned)((pi + 1 - 2 * i) * 3 + (pj + 1 - 2 * j));
#pragma unroll
            for (int e = 0; e < VE; ++e) {
              const unsigned tap = ((e < 4 ? wlo[i][j] : whi[i][j]) >> (8 * (e & 3))) & 0xffu;
              gq[e] += tap == me ? d[i][j][e] : 0.f;
            }
          }
        const int64_t pix = ((int64_t)img * h + 2 * a + pi) * w + 2 * b + pj;
        const float(&yy)[VE] = yq[pi][pj];
        float o[VE];
#pragma unroll
        for (int e = 0; e < VE; ++e) {
          const float gv = yy[e] * sc[e] + sh[e] > 0.f ? round_as<T>(gq[e]) : 0.f;
          if (APPLY) {
            o[e] = cA[e] * gv + (yy[e] * cP[e] + cQ[e]);
          } else {
            const float xh = (yy[e] - mu[e]) * is[e];
            s1[e] += gv;
            s2[e] += gv * xh;
          }
        }
        if (APPLY) Vec16<T>::template store<true>(dy + pix * c + cv * VE, o);
      }
  }
  if (!APPLY) {
#pragma unroll
    for (int e = 0; e < VE; ++e) {
      red[0][threadIdx.x][e] = s1[e];
      red[1][threadIdx.x][e] = s2[e];
    }
    __syncthreads();
    if (bl == 0) {
      for (int r = 1; r < lanes; ++r)
#pragma unroll
        for (int e = 0; e < VE; ++e) {
          s1[e] += red[0][r * cvecs + cv][e];
          s2[e] += red[1][r * cvecs + cv][e];
        }
#pragma unroll
      for (int e = 0; e < VE; ++e) {
        partial[((int64_t)blockIdx.x * 2 + 0) * c + cv * VE + e] = s1[e];
        partial[((int64_t)blockIdx.x * 2 + 1) * c + cv * VE + e] = s2[e];
      }
    }
  }
}

// ---- folded BatchNorm of a 1x1 convolution: the parameter-sized algebra -----------------------------------
// y = a W^T pixel by pixel, so with S2 = a^T a [cw][cw], t2 = sum a [cw] (both from the 1x1 weight-gradient kernel):
//   sum y_c = W_c . t2,   sum y_c^2 = W_c^T S2 W_c.
// One block per output channel c.  W is the fp32 master weight, rounded to bf16 first when the MFMAs see bf16.
__device__ __forceinline__ float fold_w(const float* w, int idx, int round_bf16) {
  const float v = w[idx];
  return round_bf16 ? bf16_to_f32(f32_to_bf16(v)) : v;
}
__device__ __forceinline__ double block_sum256(double v, double* red) {  // 256 threads, fixed order
  red[threadIdx.x] = v;
  __syncthreads();
  for (int h = 128; h >= 1; h >>= 1) {
    if ((int)threadIdx.x < h) red[threadIdx.x] += red[threadIdx.x + h];
    __syncthreads();
  }
  const double r = red[0];
  __syncthreads();
  return r;
}


// C[m][n] (+ epilogue) = sum_k A(m, k) B(k, n) for parameter-sized fp32 operands with arbitrary strides; 64 x 64 tile,
// 16-deep slabs, 4 x 4 outputs per thread.  MODE 0: store fp32 (ldc); MODE 1: store -C in T (the CRSK operand of the
// second folded dgrad term).  A may be rounded to bf16 on load (the weights the MFMAs saw).
template <typename T, int MODE>
__global__ __launch_bounds__(256) void fold_sgemm_kernel(const float* __restrict__ a, int sam, int sak, int round_a,
                                                         const float* __restrict__ b, int sbk, int sbn, int m, int n, int k,
                                                         float* __restrict__ cf, T* __restrict__ ct, int ldc, int kper,
                                                         const float* __restrict__ ct2 = nullptr, double inv_m = 0.0) {
  // ct2 != nullptr (round 6): B is the raw Gram matrix S2 [k][n] and is CENTRED on load, B(k, n) - ct2[k] ct2[n] inv_m in fp64 rounded to fp32 --
  // fold_center_kernel's expression element by element (bit-identical operand), without the launch and the cw x cw round trip
  __shared__ float sa[16][68], sb[16][68];
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;  // outputs rows ty*4.., cols tx*4..
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
  // loader: 1024 elements per operand slab, 4 per thread; lanes run along the operand's unit-stride axis
  const bool a_k_fast = sak == 1, b_n_fast = sbn == 1;
  // split-K: slice blockIdx.z reduces k in [z * kper, (z + 1) * kper) and stores its fp32 partial at cf + z * m * ldc
  // (fold_sum_kernel adds the slices in a fixed order); gridDim.z == 1: final result with the MODE epilogue
  const int kbeg = blockIdx.z * kper;
  k = kbeg + kper < k ? kbeg + kper : k;
  const bool partial = gridDim.z > 1;
  if (partial) cf += (int64_t)blockIdx.z * m * ldc;
  for (int k0 = kbeg; k0 < k; k0 += 16) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int id = threadIdx.x + 256 * e;
      const int am = a_k_fast ? id >> 4 : id & 63, ak = a_k_fast ? id & 15 : id >> 6;
      float av = 0.f;
      if (m0 + am < m && k0 + ak < k) {
        av = a[(int64_t)(m0 + am) * sam + (int64_t)(k0 + ak) * sak];
        if (round_a) av = bf16_to_f32(f32_to_bf16(av));
      }
      sa[ak][am] = av;
      const int bn_ = b_n_fast ? id & 63 : id >> 4, bk = b_n_fast ? id >> 6 : id & 15;
      float bv = 0.f;
      if (n0 + bn_ < n && k0 + bk < k) {
        bv = b[(int64_t)(k0 + bk) * sbk + (int64_t)(n0 + bn_) * sbn];
        if (ct2 != nullptr) bv = (float)__dsub_rn((double)bv, __dmul_rn(__dmul_rn((double)ct2[k0 + bk], (double)ct2[n0 + bn_]), inv_m));
      }
      sb[bk][bn_] = bv;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      const float4 av = *reinterpret_cast<const float4*>(&sa[kk][ty * 4]);
      const float4 bv = *reinterpret_cast<const float4*>(&sb[kk][tx * 4]);
      const float ar[4] = {av.x, av.y, av.z, av.w}, br[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] += ar[i] * br[j];
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = m0 + ty * 4 + i;
    if (r >= m) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int cidx = n0 + tx * 4 + j;
      if (cidx >= n) continue;
      if (MODE == 0 || partial) cf[(int64_t)r * ldc + cidx] = acc[i][j];
      else Elem<T>::store(ct + (int64_t)r * ldc + cidx, -acc[i][j]);
    }
  }
}

template <typename T, int MODE>
__global__ __launch_bounds__(256) void fold_sum_kernel(const float* __restrict__ part, int ks, int64_t count, float* __restrict__ cf,
                                                       T* __restrict__ ct) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= count) return;
  float t = 0.f;
  for (int z = 0; z < ks; ++z) t += part[(int64_t)z * count + i];
  if (MODE == 0) cf[i] = t;
  else Elem<T>::store(ct + i, -t);
}

// split factor: ~512 blocks in flight, slices of at least 64 k, multiples of 16
static inline void fold_split(int tiles, int k, int* ks, int* kper) {
  int s_ = (2048 + tiles - 1) / tiles;  // ~8 blocks of 256 threads per CU: the slab loads are latency-bound, occupancy hides them
  const int maxs = k / 64 > 0 ? k / 64 : 1;
  if (s_ > maxs) s_ = maxs;
  if (s_ < 1) s_ = 1;
  int per = ((k + s_ - 1) / s_ + 15) / 16 * 16;
  *ks = (k + per - 1) / per;
  *kper = per;
}

// Centred second moments of the (narrow) input: S2c = a^T a - (sum a)(sum a)^T / M  (fp64 arithmetic on the fp32 sums).  The
// batch variance of y_c = W_c . a is then W_c S2c W_c^T / M directly -- the E[y^2] - mean^2 cancellation of a channel whose
// mean dwarfs its spread happens here, on the INPUT's statistics (post-ReLU activations: mean ~ std), not on the outputs'.
__global__ __launch_bounds__(256) void fold_center_kernel(const float* __restrict__ s2, const float* __restrict__ t2, int cw, double inv_m,
                                                          float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)cw * cw) return;
  const int j = (int)(i / cw), k = (int)(i - (int64_t)j * cw);
  out[i] = (float)__dsub_rn((double)s2[i], __dmul_rn(__dmul_rn((double)t2[j], (double)t2[k]), inv_m));  // (pinned roundings: the product's loader repeats it)
}

__global__ __launch_bounds__(256) void bn_fold_fwd_kernel(const float* __restrict__ w, int round_bf16, const float* __restrict__ s2,
                                                          const float* __restrict__ t2, int cc, int cw, int64_t m,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                          float momentum, float* __restrict__ running_mean,
                                                          float* __restrict__ running_var, int64_t* __restrict__ nbt,
                                                          float* __restrict__ mean_o, float* __restrict__ invstd_o,
                                                          float* __restrict__ scale_o, float* __restrict__ shift_o,
                                                          float* __restrict__ ws2, const float* __restrict__ part = nullptr, int ks = 1) {
  __shared__ double red[256];
  __shared__ double s_mean;
  const int c = blockIdx.x;
  double sy = 0.0, sy2 = 0.0;
  // part != nullptr (round 6): the product arrives as `ks` split-K slices [ks][cc][cw]; they are added here in fold_sum_kernel's order
  // (z = 0 .. ks - 1 into a float), once per loop below -- the same float both times, and the fold_sum launch is gone
  const int64_t count = (int64_t)gridDim.x * cw;
  auto ws2c = [&](int j) -> float {
    const int64_t o = (int64_t)c * cw + j;
    if (part == nullptr) return ws2[o];
    float t = 0.f;
    for (int z = 0; z < ks; ++z) t += part[(int64_t)z * count + o];
    return t;
  };
  for (int j = threadIdx.x; j < cw; j += 256) {  // W S2c (CENTRED Gram) from fold_sgemm_kernel
    const float wv = fold_w(w, c * cw + j, round_bf16);
    sy += (double)wv * (double)t2[j];
    sy2 += (double)ws2c(j) * (double)wv;
  }
  sy = block_sum256(sy, red);
  sy2 = block_sum256(sy2, red);
  if (threadIdx.x == 0) s_mean = sy / (double)m;
  __syncthreads();
  // the backward's algebra wants the UN-centred product: W S2 = W S2c + (W t2) t2^T / M = ws2c + mean_c t2
  for (int j = threadIdx.x; j < cw; j += 256) ws2[(int64_t)c * cw + j] = (float)((double)ws2c(j) + s_mean * (double)t2[j]);
  if (threadIdx.x == 0) {
    if (c == 0 && nbt) nbt[0] += 1;
    const double mean = s_mean;
    double var = sy2 / (double)m;  // no subtraction of mean^2: the second moments are centred already
    if (var < 0.0) var = 0.0;
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    const float sc = (gamma ? gamma[c] : 1.0f) * invstd;
    mean_o[c] = (float)mean;
    invstd_o[c] = invstd;
    scale_o[c] = sc;
    shift_o[c] = (beta ? beta[c] : 0.0f) - (float)mean * sc;
    if (running_mean) {
      const double unbiased = m > 1 ? var * (double)m / (double)(m - 1) : var;
      running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * (float)mean;
      running_var[c] = (1.0f - momentum) * running_var[c] + momentum * (float)unbiased;
    }
  }
}

// backward coefficients and everything that is per output channel:
//   sum g*y = G_c . W_c;  dgamma = invstd (sum g*y - mean s);  dbeta = s;  A = gamma invstd;  B = invstd A dgamma / M;
//   C = -A dbeta / M + mean B;   dW_c = A G_c - B (W S2)_c + C t2;   wa[j][c] = A W_c[j] (CRSK operand of the first dgrad
//   term);  bw_c = B W_c (operand of W^T diag(B) W);  ccoef[c] = C
template <typename T>
__global__ __launch_bounds__(256) void bn_fold_bwd_kernel(const float* __restrict__ w, int round_bf16, const float* __restrict__ gmat,
                                                          const float* __restrict__ s, const float* __restrict__ ws2,
                                                          const float* __restrict__ t2, const float* __restrict__ mean,
                                                          const float* __restrict__ invstd, const float* __restrict__ gamma, int cc,
                                                          int cw, int64_t m, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                          float* __restrict__ dw, T* __restrict__ wa, float* __restrict__ bw,
                                                          float* __restrict__ ccoef) {
  __shared__ double red[256];
  const int c = blockIdx.x;
  double sgy = 0.0;
  for (int j = threadIdx.x; j < cw; j += 256) sgy += (double)gmat[(int64_t)c * cw + j] * (double)fold_w(w, c * cw + j, round_bf16);
  sgy = block_sum256(sgy, red);
  const float is = invstd[c], mu = mean[c], sc_ = s[c];
  const float dg = is * (float)(sgy - (double)mu * (double)sc_);
  const float ca = (gamma ? gamma[c] : 1.0f) * is;
  const float inv_m = (float)(1.0 / (double)m);
  const float cb = is * (ca * dg * inv_m);
  const float cconst = -ca * sc_ * inv_m + mu * cb;
  if (threadIdx.x == 0) {
    dgamma[c] = dg;
    dbeta[c] = sc_;
    ccoef[c] = cconst;
  }
  for (int j = threadIdx.x; j < cw; j += 256) {
    const int64_t o = (int64_t)c * cw + j;
    const float wv = fold_w(w, (int)o, round_bf16);
    dw[o] = ca * gmat[o] - cb * ws2[o] + cconst * t2[j];
    Elem<T>::store(wa + (int64_t)j * cc + c, ca * wv);
    bw[o] = cb * wv;
  }
}

// bias[j] = sum_c C_c W[c][j]: 32 columns per block (lanes along j: coalesced rows of W), 32 row lanes over c, fixed-order fold
__global__ __launch_bounds__(1024) void bn_fold_bias_kernel(const float* __restrict__ w, int round_bf16, const float* __restrict__ ccoef,
                                                            int cc, int cw, float* __restrict__ bias) {
  __shared__ float red[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int j = blockIdx.x * 32 + tx;
  float acc = 0.f;
  if (j < cw)
    for (int c = ty; c < cc; c += 32) acc += ccoef[c] * fold_w(w, c * cw + j, round_bf16);
  red[ty][tx] = acc;
  __syncthreads();
  if (ty == 0 && j < cw) {
    float t = 0.f;
    for (int r = 0; r < 32; ++r) t += red[r][tx];
    bias[j] = t;
  }
}

// Round 6: the split-K sum of W^T diag(B) W (fold_sum_kernel<T, 1>'s arithmetic, 1024 elements per block) and bias = C W (bn_fold_bias_kernel,
// unchanged) in ONE launch: blocks [0, nsum) add the slices, blocks [nsum, nsum + ceil(cw / 32)) reduce the bias columns.
template <typename T>
__global__ __launch_bounds__(1024) void fold_sum_bias_kernel(const float* __restrict__ part, int ks, int64_t count, T* __restrict__ ct, int nsum,
                                                             const float* __restrict__ w, int round_bf16, const float* __restrict__ ccoef,
                                                             int cc, int cw, float* __restrict__ bias) {
  __shared__ float red[32][33];
  if ((int)blockIdx.x < nsum) {
    const int64_t i = (int64_t)blockIdx.x * 1024 + threadIdx.x;
    if (i >= count) return;
    float t = 0.f;
    for (int z = 0; z < ks; ++z) t += part[(int64_t)z * count + i];
    Elem<T>::store(ct + i, -t);
    return;
  }
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int j = ((int)blockIdx.x - nsum) * 32 + tx;
  float acc = 0.f;
  if (j < cw)
    for (int c = ty; c < cc; c += 32) acc += ccoef[c] * fold_w(w, c * cw + j, round_bf16);
  red[ty][tx] = acc;
  __syncthreads();
  if (ty == 0 && j < cw) {
    float t = 0.f;
    for (int r = 0; r < 32; ++r) t += red[r][tx];
    bias[j] = t;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ x, int64_t m, int c, int rows_per_blk,
                                                     float* __restrict__ partial) {
  constexpr int VE = Vec16<T>::N;
  column_reduce<T>(m, c, rows_per_blk, partial, [&](int64_t r, int cv, float(&s1)[VE], float(&s2)[VE]) {
    float v[VE];
    Vec16<T>::load(x + r * c + cv * VE, v);
#pragma unroll
    for (int e = 0; e < VE; ++e) s1[e] += v[e];
  });
}

// blocks for the row-walking kernels: >= ~8 rows per row lane, at most `cap` blocks.  The two big streaming passes (bn_apply,
// bn_bwd_apply) run with cap = 131072 (env SIMHAND_BN_GRID_APPLY / SIMHAND_BN_GRID_BWD: A/B timing): many short blocks keep more
// loads in flight than 2048 long-lived ones -- 5.1 -> 6.0-6.2 TB/s (bn_apply), 5.1 -> 5.9 (bn_bwd_apply, whose per-block coefficient
// prologue had to become 16-B loads first: with scalar loads more blocks made it slower), round 3, scripts/bn_bench.py
static inline int row_grid(int64_t m, int cvecs, int cap = 2048) {
  const int span = cvecs < 256 ? cvecs : 256;
  const int rowlanes = 256 / span;
  int64_t g = (m + (int64_t)rowlanes * 8 - 1) / ((int64_t)rowlanes * 8);
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

static inline int stream_grid(int64_t nvec) {
  int64_t g = (nvec + 255) / 256;
  if (g > 8192) g = 8192;
  if (g < 1) g = 1;
  return (int)g;
}

static hook_t g_bn_nt{1};  // non-temporal loads / stores in the streaming kernels (test hook: simhand_test_bn_set_nt)

void hooks_reset_bn() { g_bn_nt = 1; }

__global__ void bn_bwd_coefs_kernel(const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ gamma,
                                    const float* __restrict__ dgamma, const float* __restrict__ dbeta, float inv_m, int c,
                                    float* __restrict__ ca, float* __restrict__ cb, float* __restrict__ cc) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= c) return;
  // dy = gamma invstd (g - mean(g) - xhat mean(g xhat)),  xhat = (y - mean) invstd   ->   A g - B y + C
  const float a = __fmul_rn(gamma[i], invstd[i]);                       // (pinned roundings: bn_bwd_finalize_kernel repeats them)
  const float b = __fmul_rn(__fmul_rn(__fmul_rn(a, invstd[i]), dgamma[i]), inv_m);
  ca[i] = a;
  cb[i] = b;
  cc[i] = __fadd_rn(__fmul_rn(__fmul_rn(-a, dbeta[i]), inv_m), __fmul_rn(mean[i], b));
}

}  // namespace sh

using namespace sh;

extern "C" {

int simhand_bn_bwd_coefs(const float* mean, const float* invstd, const float* gamma, const float* dgamma, const float* dbeta, int64_t m, int c,
                         float* coef_a, float* coef_b, float* coef_c, sh_stream_t stream) {
  SH_REQUIRE(mean && invstd && gamma && dgamma && dbeta && coef_a && coef_b && coef_c && m >= 1 && c >= 1, "bn_bwd_coefs: bad arguments");
  bn_bwd_coefs_kernel<<<ceil_div(c, 256), 256, 0, (hipStream_t)stream>>>(mean, invstd, gamma, dgamma, dbeta, (float)(1.0 / (double)m), c,
                                                                         coef_a, coef_b, coef_c);
  return check_launch("bn_bwd_coefs");
}

int simhand_test_bn_set_nt(int on) {
  g_bn_nt = on ? 1 : 0;
  return 0;
}

int simhand_bn_stat_blocks(int64_t m, int c) {
  (void)c;
  int rpb, nblk;
  col_plan(m, &rpb, &nblk);
  return nblk;
}

int simhand_bn_partial_stats(const void* y, int64_t m, int c, int dtype, float* partial, sh_stream_t stream) {
  SH_REQUIRE(y && partial, "bn_partial_stats: NULL pointer");
  SH_REQUIRE(m >= 1 && c >= 1, "bn_partial_stats: bad shape");
  SH_REQUIRE(c % (dtype == SH_F32 ? 4 : 8) == 0, "bn_partial_stats: c=%d not a multiple of the 16-B vector", c);
  int rpb, nblk;
  col_plan(m, &rpb, &nblk);
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SH_PROF_BN, s, 0, (double)m * c * (dtype == SH_F32 ? 4 : 2));
  if (dtype == SH_F32) bn_partial_stats_kernel<float><<<nblk, 256, 0, s>>>((const float*)y, m, c, rpb, partial);
  else bn_partial_stats_kernel<bf16_t><<<nblk, 256, 0, s>>>((const bf16_t*)y, m, c, rpb, partial);
  return check_launch("bn_partial_stats");
}

size_t simhand_bn_finalize_workspace_bytes(int nblk, int c) {
  return (size_t)ceil_div(nblk, kChunk) * 2 * c * sizeof(double);
}

int simhand_bn_finalize(const float* partial, int nblk, int64_t m, int c, const float* gamma, const float* beta,
                        const float* pre_bias, float eps, float momentum, float* running_mean, float* running_var,
                        int64_t* num_batches_tracked, float* mean, float* invstd, float* scale, float* shift,
                        void* workspace, size_t workspace_bytes, sh_stream_t stream) {
  SH_REQUIRE(partial && mean && invstd && scale && shift && workspace, "bn_finalize: NULL pointer");
  SH_REQUIRE(nblk >= 1 && m >= 1 && c >= 1, "bn_finalize: bad shape");
  SH_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "bn_finalize: running stats must be given together");
  SH_REQUIRE(workspace_bytes >= simhand_bn_finalize_workspace_bytes(nblk, c), "bn_finalize: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const int nchunk = ceil_div(nblk, kChunk);
  ProfScope ps(SH_PROF_BN, s, 0, (double)nblk * 2 * c * 4);
  fold_partials_kernel<<<dim3(nchunk, ceil_div(c, 64)), 256, 0, s>>>(partial, nblk, c, (double*)workspace);
  if (check_launch("bn fold_partials")) return 1;
  bn_finalize_kernel<<<ceil_div(c, 256), 256, 0, s>>>((const double*)workspace, nchunk, m, c, gamma, beta, pre_bias, eps, momentum,
                                                       running_mean, running_var, num_batches_tracked, mean, invstd, scale, shift);
  return check_launch("bn_finalize");
}

int simhand_bn_finalize_ticket(const float* partial, int nblk, int64_t m, int c, const float* gamma, const float* beta,
                               const float* pre_bias, float eps, float momentum, float* running_mean, float* running_var,
                               int64_t* num_batches_tracked, float* mean, float* invstd, float* scale, float* shift,
                               void* workspace, size_t workspace_bytes, uint32_t* ticket, sh_stream_t stream) {
  SH_REQUIRE(ticket != nullptr, "bn_finalize_ticket: NULL ticket");
  if (sw(SH_SW_FOLD_LEGACY))
    return simhand_bn_finalize(partial, nblk, m, c, gamma, beta, pre_bias, eps, momentum, running_mean, running_var, num_batches_tracked, mean,
                               invstd, scale, shift, workspace, workspace_bytes, stream);
  SH_REQUIRE(partial && mean && invstd && scale && shift && workspace, "bn_finalize_ticket: NULL pointer");
  SH_REQUIRE(nblk >= 1 && m >= 1 && c >= 1, "bn_finalize_ticket: bad shape");
  SH_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "bn_finalize_ticket: running stats must be given together");
  SH_REQUIRE(workspace_bytes >= simhand_bn_finalize_workspace_bytes(nblk, c), "bn_finalize_ticket: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const int nchunk = ceil_div(nblk, kChunk);
  ProfScope ps(SH_PROF_BN, s, 0, (double)nblk * 2 * c * 4);
  bn_finalize_ticket_kernel<<<dim3(nchunk, ceil_div(c, 64)), 256, 0, s>>>(partial, nblk, nchunk, m, c, (double*)workspace, ticket, gamma, beta,
                                                                          pre_bias, eps, momentum, running_mean, running_var,
                                                                          num_batches_tracked, mean, invstd, scale, shift);
  return check_launch("bn_finalize_ticket");
}

int simhand_bn_eval_params(const float* gamma, const float* beta, const float* running_mean, const float* running_var, float eps,
                            int c, float* scale, float* shift, sh_stream_t stream) {
  SH_REQUIRE(running_mean && running_var && scale && shift, "bn_eval_params: NULL pointer");
  bn_eval_params_kernel<<<ceil_div(c, 256), 256, 0, (hipStream_t)stream>>>(gamma, beta, running_mean, running_var, eps, c, scale, shift);
  return check_launch("bn_eval_params");
}

int simhand_bn_apply(const void* y, const float* scale, const float* shift, const void* residual, int relu, void* a,
                     uint8_t* relu_mask, int64_t m, int c, int dtype, sh_stream_t stream) {
  SH_REQUIRE(y && scale && shift && a, "bn_apply: NULL pointer");
  SH_REQUIRE(!relu_mask || relu, "bn_apply: a ReLU mask is only produced with relu != 0");
  const int ve = dtype == SH_F32 ? 4 : 8;
  SH_REQUIRE(c % ve == 0, "bn_apply: c=%d not a multiple of %d", c, ve);
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SH_PROF_BN, s, 0, (double)m * c * (dtype == SH_F32 ? 4 : 2) * (residual ? 3 : 2));
  route_hit(SH_ROUTE_BN_APPLY);
  const int grid = row_grid(m, c / ve, sw(SH_SW_BN_GRID_APPLY));
#define SH_BN_APPLY(T, NT) bn_apply_kernel<T, NT><<<grid, 256, 0, s>>>((const T*)y, scale, shift, (const T*)residual, relu, (T*)a, relu_mask, m, c)
  if (dtype == SH_F32) { if (g_bn_nt) SH_BN_APPLY(float, true); else SH_BN_APPLY(float, false); }
  else { if (g_bn_nt) SH_BN_APPLY(bf16_t, true); else SH_BN_APPLY(bf16_t, false); }
#undef SH_BN_APPLY
  return check_launch("bn_apply");
}

int simhand_bn_apply_fp8(const void* y, const float* scale, const float* shift, int relu, void* a, void* q, const float* q_state,
                         uint32_t* amax_bits, int64_t m, int c, sh_stream_t stream) {
  SH_REQUIRE(y && scale && shift && a && q && q_state, "bn_apply_fp8: NULL pointer");
  SH_REQUIRE(c % 8 == 0 && m >= 1, "bn_apply_fp8: c=%d must be a multiple of 8", c);
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SH_PROF_BN, s, 0, (double)m * c * 5);
  route_hit(SH_ROUTE_BN_APPLY);
  const int grid = row_grid(m, c / 8);
  if (g_bn_nt) bn_apply_fp8_kernel<true><<<grid, 256, 0, s>>>((const bf16_t*)y, scale, shift, relu, (bf16_t*)a, (unsigned char*)q, q_state, amax_bits, m, c);
  else bn_apply_fp8_kernel<false><<<grid, 256, 0, s>>>((const bf16_t*)y, scale, shift, relu, (bf16_t*)a, (unsigned char*)q, q_state, amax_bits, m, c);
  return check_launch("bn_apply_fp8");
}

int simhand_bn_bwd_partial(const void* da, const void* a, const void* y, const float* mean, const float* invstd,
                           const float* scale, const float* shift, int relu, int64_t m, int c, int dtype, float* partial,
                           sh_stream_t stream) {
  SH_REQUIRE(da && y && mean && invstd && partial, "bn_bwd_partial: NULL pointer");
  SH_REQUIRE(relu >= 0 && relu <= 3, "bn_bwd_partial: relu mode %d", relu);
  SH_REQUIRE((relu != 1 && relu != 3) || a, "bn_bwd_partial: relu modes 1 / 3 need the activation output / its bit mask");
  SH_REQUIRE(relu != 2 || (scale && shift), "bn_bwd_partial: relu mode 2 needs scale/shift");
  SH_REQUIRE(c % (dtype == SH_F32 ? 4 : 8) == 0, "bn_bwd_partial: c=%d not a multiple of the 16-B vector", c);
  int rpb, nblk;
  col_plan(m, &rpb, &nblk);
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SH_PROF_BN, s, 0, (double)m * c * (dtype == SH_F32 ? 4 : 2) * (relu == 1 ? 3 : 2));
#define SH_BN_BP(T, NT) bn_bwd_partial_kernel<T, NT><<<nblk, 256, 0, s>>>((const T*)da, (const T*)a, (const T*)y, mean, invstd, scale, shift, relu, m, c, rpb, partial)
  if (dtype == SH_F32) { if (g_bn_nt) SH_BN_BP(float, true); else SH_BN_BP(float, false); }
  else { if (g_bn_nt) SH_BN_BP(bf16_t, true); else SH_BN_BP(bf16_t, false); }
#undef SH_BN_BP
  return check_launch("bn_bwd_partial");
}

int simhand_bn_bwd_finalize(const float* partial, int nblk, int c, float* dgamma, float* dbeta, sh_stream_t stream) {
  SH_REQUIRE(partial && dgamma && dbeta, "bn_bwd_finalize: NULL pointer");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SH_PROF_BN, s, 0, (double)nblk * 2 * c * 4);
  bn_bwd_finalize_kernel<float><<<ceil_div(c, 16), 1024, 0, s>>>(partial, nblk, c, dgamma, dbeta);
  return check_launch("bn_bwd_finalize");
}

static int raw_slices(int nblk) { return nblk >= 4096 ? 64 : 1; }

size_t simhand_bn_bwd_finalize_raw_workspace_bytes(int nblk, int c) {
  const int sl = raw_slices(nblk);
  return sl > 1 ? (size_t)sl * 2 * c * sizeof(double) : 16;
}

int simhand_bn_bwd_finalize_raw(const float* partial, int nblk, int c, const float* mean, const float* invstd, float* dgamma,
                                float* dbeta, void* workspace, size_t workspace_bytes, sh_stream_t stream) {
  SH_REQUIRE(partial && mean && invstd && dgamma && dbeta, "bn_bwd_finalize_raw: NULL pointer");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SH_PROF_BN, s, 0, (double)nblk * 2 * c * 4);
  const int sl = raw_slices(nblk);
  if (sl > 1) {
    SH_REQUIRE(workspace && workspace_bytes >= simhand_bn_bwd_finalize_raw_workspace_bytes(nblk, c), "bn_bwd_finalize_raw: workspace too small");
    double* lvl2 = (double*)workspace;
    bn_bwd_finalize_kernel<float><<<dim3(ceil_div(c, 16), sl), 1024, 0, s>>>(partial, nblk, c, nullptr, nullptr, nullptr, nullptr, lvl2);
    if (check_launch("bn_bwd_finalize_raw fold")) return 1;
    bn_bwd_finalize_kernel<double><<<ceil_div(c, 16), 1024, 0, s>>>(lvl2, sl, c, dgamma, dbeta, mean, invstd);
  } else {
    bn_bwd_finalize_kernel<float><<<ceil_div(c, 16), 1024, 0, s>>>(partial, nblk, c, dgamma, dbeta, mean, invstd);
  }
  return check_launch("bn_bwd_finalize_raw");
}

// the same + the coefficients (A, B, C) of dy = A g - B y + C (simhand_bn_bwd_coefs) out of the SAME launch: coefs [3][c]
int simhand_bn_bwd_finalize_raw_coefs(const float* partial, int nblk, int c, const float* mean, const float* invstd, const float* gamma, int64_t m,
                                      float* dgamma, float* dbeta, float* coefs, void* workspace, size_t workspace_bytes, sh_stream_t stream) {
  SH_REQUIRE(partial && mean && invstd && gamma && dgamma && dbeta && coefs && m >= 1, "bn_bwd_finalize_raw_coefs: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  if (sw(SH_SW_FOLD_LEGACY)) {  // the round-5 pair of launches
    if (simhand_bn_bwd_finalize_raw(partial, nblk, c, mean, invstd, dgamma, dbeta, workspace, workspace_bytes, stream)) return 1;
    return simhand_bn_bwd_coefs(mean, invstd, gamma, dgamma, dbeta, m, c, coefs, coefs + c, coefs + 2 * c, stream);
  }
  ProfScope ps(SH_PROF_BN, s, 0, (double)nblk * 2 * c * 4);
  const float inv_m = (float)(1.0 / (double)m);
  const int sl = raw_slices(nblk);
  if (sl > 1) {
    SH_REQUIRE(workspace && workspace_bytes >= simhand_bn_bwd_finalize_raw_workspace_bytes(nblk, c), "bn_bwd_finalize_raw_coefs: workspace too small");
    double* lvl2 = (double*)workspace;
    bn_bwd_finalize_kernel<float><<<dim3(ceil_div(c, 16), sl), 1024, 0, s>>>(partial, nblk, c, nullptr, nullptr, nullptr, nullptr, lvl2);
    if (check_launch("bn_bwd_finalize_raw_coefs fold")) return 1;
    bn_bwd_finalize_kernel<double><<<ceil_div(c, 16), 1024, 0, s>>>(lvl2, sl, c, dgamma, dbeta, mean, invstd, nullptr, gamma, inv_m, coefs);
  } else {
    bn_bwd_finalize_kernel<float><<<ceil_div(c, 16), 1024, 0, s>>>(partial, nblk, c, dgamma, dbeta, mean, invstd, nullptr, gamma, inv_m, coefs);
  }
  return check_launch("bn_bwd_finalize_raw_coefs");
}

int simhand_bn_bwd_apply(const void* da, const void* a, const void* y, const float* mean, const float* invstd, const float* gamma,
                         const float* dgamma, const float* dbeta, const float* scale, const float* shift, int relu, void* dy,
                         void* dres, int64_t m, int c, int dtype, sh_stream_t stream) {
  SH_REQUIRE(da && y && mean && invstd && dgamma && dbeta && dy, "bn_bwd_apply: NULL pointer");
  SH_REQUIRE(relu >= 0 && relu <= 3, "bn_bwd_apply: relu mode %d", relu);
  SH_REQUIRE((relu != 1 && relu != 3) || a, "bn_bwd_apply: relu modes 1 / 3 need the activation output / its bit mask");
  SH_REQUIRE(relu != 2 || (scale && shift), "bn_bwd_apply: relu mode 2 needs scale/shift");
  const int ve = dtype == SH_F32 ? 4 : 8;
  SH_REQUIRE(c % ve == 0, "bn_bwd_apply: c=%d not a multiple of %d", c, ve);
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SH_PROF_BN, s, 0, (double)m * c * (dtype == SH_F32 ? 4 : 2) * (3 + (relu == 1 ? 1 : 0) + (dres ? 1 : 0)));
  route_hit(SH_ROUTE_BN_BWD_APPLY);
  const float inv_m = (float)(1.0 / (double)m);
  const int grid = row_grid(m, c / ve, sw(SH_SW_BN_GRID_BWD));
#define SH_BN_BA(T, NT) bn_bwd_apply_kernel<T, NT><<<grid, 256, 0, s>>>((const T*)da, (const T*)a, (const T*)y, mean, invstd, gamma, dgamma, dbeta, scale, shift, relu, (T*)dy, (T*)dres, m, c, inv_m)
  if (dtype == SH_F32) { if (g_bn_nt) SH_BN_BA(float, true); else SH_BN_BA(float, false); }
  else { if (g_bn_nt) SH_BN_BA(bf16_t, true); else SH_BN_BA(bf16_t, false); }
#undef SH_BN_BA
  return check_launch("bn_bwd_apply");
}

int simhand_bn_bwd_apply_fp8(const void* da, const void* a, const void* y, const float* mean, const float* invstd, const float* gamma,
                             const float* dgamma, const float* dbeta, const float* scale, const float* shift, int relu, void* dy, void* q,
                             const float* q_state, uint32_t* amax_bits, int64_t m, int c, sh_stream_t stream) {
  SH_REQUIRE(da && y && mean && invstd && dgamma && dbeta && dy && q && q_state, "bn_bwd_apply_fp8: NULL pointer");
  SH_REQUIRE(relu >= 0 && relu <= 3, "bn_bwd_apply_fp8: relu mode %d", relu);
  SH_REQUIRE((relu != 1 && relu != 3) || a, "bn_bwd_apply_fp8: relu modes 1 / 3 need the activation output / its bit mask");
  SH_REQUIRE(relu != 2 || (scale && shift), "bn_bwd_apply_fp8: relu mode 2 needs scale/shift");
  SH_REQUIRE(c % 8 == 0, "bn_bwd_apply_fp8: c=%d not a multiple of 8", c);
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SH_PROF_BN, s, 0, (double)m * c * 7);
  route_hit(SH_ROUTE_BN_BWD_APPLY);
  const float inv_m = (float)(1.0 / (double)m);
  const int grid = row_grid(m, c / 8);
#define SH_BN_BAQ(NT) bn_bwd_apply_fp8_kernel<NT><<<grid, 256, 0, s>>>((const bf16_t*)da, (const bf16_t*)a, (const bf16_t*)y, mean, invstd, gamma, dgamma, dbeta, scale, shift, relu, (bf16_t*)dy, (unsigned char*)q, q_state, amax_bits, m, c, inv_m)
  if (g_bn_nt) SH_BN_BAQ(true); else SH_BN_BAQ(false);
#undef SH_BN_BAQ
  return check_launch("bn_bwd_apply_fp8");
}

int simhand_bn_relu_maxpool_fwd(const void* y, const float* scale, const float* shift, void* out, uint8_t* idx, void* ywin, int n, int h,
                                int w, int c, int dtype, sh_stream_t stream) {
  SH_REQUIRE(y && scale && shift && out && idx, "bn_relu_maxpool_fwd: NULL pointer");
  const int ve = dtype == SH_F32 ? 4 : 8;
  SH_REQUIRE(c % ve == 0, "bn_relu_maxpool_fwd: c=%d not a multiple of %d", c, ve);
  const int ho = (h + 2 - 3) / 2 + 1, wo = (w + 2 - 3) / 2 + 1;
  const int64_t total = (int64_t)n * ho * wo * (c / ve);
  hipStream_t s = (hipStream_t)stream;
  const double es = dtype == SH_F32 ? 4 : 2;
  ProfScope ps(SH_PROF_BN, s, 0, es * ((double)n * h * w * c + (double)n * ho * wo * c) + (double)n * ho * wo * c);
  route_hit(SH_ROUTE_STEM_BN_POOL);
  SH_REQUIRE(total < (1ll << 31), "bn_relu_maxpool_fwd: %lld output vectors exceed the 2^31 index range", (long long)total);
  const FastDiv d1 = make_fastdiv((unsigned)(c / ve)), d2 = make_fastdiv((unsigned)wo), d3 = make_fastdiv((unsigned)ho);
  if (dtype == SH_F32)
    bn_relu_maxpool_fwd_kernel<float><<<stream_grid(total), 256, 0, s>>>((const float*)y, scale, shift, (float*)out, idx, (float*)ywin, n, h, w, c, ho, wo, d1, d2, d3);
  else
    bn_relu_maxpool_fwd_kernel<bf16_t><<<stream_grid(total), 256, 0, s>>>((const bf16_t*)y, scale, shift, (bf16_t*)out, idx, (bf16_t*)ywin, n, h, w, c, ho, wo, d1, d2, d3);
  return check_launch("bn_relu_maxpool_fwd");
}

int simhand_maxpool_bn_bwd_partial(const void* dz, const uint8_t* idx, const void* y, const float* mean, const float* invstd,
                                   const float* scale, const float* shift, int n, int h, int w, int c, int dtype, float* partial,
                                   sh_stream_t stream) {
  SH_REQUIRE(dz && idx && y && mean && invstd && scale && shift && partial, "maxpool_bn_bwd_partial: NULL pointer");
  const int ve = dtype == SH_F32 ? 4 : 8;
  SH_REQUIRE(c % ve == 0, "maxpool_bn_bwd_partial: c=%d not a multiple of %d", c, ve);
  const int ho = (h + 2 - 3) / 2 + 1, wo = (w + 2 - 3) / 2 + 1;
  const int64_t m = (int64_t)n * h * w;
  SH_REQUIRE(m < (1ll << 31), "maxpool_bn_bwd_partial: %lld pixels exceed the 2^31 index range", (long long)m);
  const FastDiv dw_ = make_fastdiv((unsigned)w), dh_ = make_fastdiv((unsigned)h);
  int rpb, nblk;
  col_plan(m, &rpb, &nblk);
  hipStream_t s = (hipStream_t)stream;
  const double es = dtype == SH_F32 ? 4 : 2;
  ProfScope ps(SH_PROF_BN, s, 0, es * ((double)m * c + (double)n * ho * wo * c) + (double)n * ho * wo * c);
  const int cvecs_ = c / ve;
  if (h % 2 == 0 && w % 2 == 0 && cvecs_ <= 256 && (cvecs_ & (cvecs_ - 1)) == 0) {
    const int64_t nb2 = (int64_t)n * (h / 2) * (w / 2);
    const int bpc = (int)((nb2 + nblk - 1) / nblk);
    const FastDiv dwb = make_fastdiv((unsigned)(w / 2)), dhb = make_fastdiv((unsigned)(h / 2));
    // every one of the nblk partial rows is written (rows past the last 2x2 block get zeros)
    if (dtype == SH_F32)
      pool_bn_bwd_blk_kernel<float, false><<<nblk, 256, 0, s>>>((const float*)dz, idx, (const float*)y, mean, invstd, nullptr, nullptr, nullptr,
                                                                scale, shift, nullptr, partial, n, h, w, c, ho, wo, 0.f, dwb, dhb, bpc);
    else
      pool_bn_bwd_blk_kernel<bf16_t, false><<<nblk, 256, 0, s>>>((const bf16_t*)dz, idx, (const bf16_t*)y, mean, invstd, nullptr, nullptr,
                                                                 nullptr, scale, shift, nullptr, partial, n, h, w, c, ho, wo, 0.f, dwb, dhb, bpc);
    return check_launch("maxpool_bn_bwd_partial (2x2)");
  }
  if (dtype == SH_F32)
    pool_bn_bwd_partial_kernel<float><<<nblk, 256, 0, s>>>((const float*)dz, idx, (const float*)y, mean, invstd, scale, shift, n, h, w, c, ho, wo, rpb, partial, dw_, dh_);
  else
    pool_bn_bwd_partial_kernel<bf16_t><<<nblk, 256, 0, s>>>((const bf16_t*)dz, idx, (const bf16_t*)y, mean, invstd, scale, shift, n, h, w, c, ho, wo, rpb, partial, dw_, dh_);
  return check_launch("maxpool_bn_bwd_partial");
}

int simhand_maxpool_bn_bwd_apply(const void* dz, const uint8_t* idx, const void* y, const float* mean, const float* invstd,
                                 const float* gamma, const float* dgamma, const float* dbeta, const float* scale, const float* shift,
                                 void* dy, int n, int h, int w, int c, int dtype, sh_stream_t stream) {
  SH_REQUIRE(dz && idx && y && mean && invstd && dgamma && dbeta && scale && shift && dy, "maxpool_bn_bwd_apply: NULL pointer");
  const int ve = dtype == SH_F32 ? 4 : 8;
  SH_REQUIRE(c % ve == 0, "maxpool_bn_bwd_apply: c=%d not a multiple of %d", c, ve);
  const int ho = (h + 2 - 3) / 2 + 1, wo = (w + 2 - 3) / 2 + 1;
  const int64_t m = (int64_t)n * h * w;
  hipStream_t s = (hipStream_t)stream;
  const double es = dtype == SH_F32 ? 4 : 2;
  ProfScope ps(SH_PROF_BN, s, 0, es * (2.0 * (double)m * c + (double)n * ho * wo * c) + (double)n * ho * wo * c);
  route_hit(SH_ROUTE_STEM_BN_POOL);
  SH_REQUIRE(m < (1ll << 31), "maxpool_bn_bwd_apply: %lld pixels exceed the 2^31 index range", (long long)m);
  const FastDiv dw_ = make_fastdiv((unsigned)w), dh_ = make_fastdiv((unsigned)h);
  const float inv_m = (float)(1.0 / (double)m);
  const int grid = row_grid(m, c / ve);
  const int cvecs_ = c / ve;
  if (h % 2 == 0 && w % 2 == 0 && cvecs_ <= 256 && (cvecs_ & (cvecs_ - 1)) == 0) {
    const int64_t nb2 = (int64_t)n * (h / 2) * (w / 2);
    const int lanes = 256 / cvecs_;
    int64_t g2 = (nb2 + (int64_t)lanes * 4 - 1) / ((int64_t)lanes * 4);  // ~4 blocks of 2x2 per thread
    if (g2 > 16384) g2 = 16384;
    const int bpc = (int)((nb2 + g2 - 1) / g2);
    const FastDiv dwb = make_fastdiv((unsigned)(w / 2)), dhb = make_fastdiv((unsigned)(h / 2));
    if (dtype == SH_F32)
      pool_bn_bwd_blk_kernel<float, true><<<(int)g2, 256, 0, s>>>((const float*)dz, idx, (const float*)y, mean, invstd, gamma, dgamma, dbeta,
                                                                 scale, shift, (float*)dy, nullptr, n, h, w, c, ho, wo, inv_m, dwb, dhb, bpc);
    else
      pool_bn_bwd_blk_kernel<bf16_t, true><<<(int)g2, 256, 0, s>>>((const bf16_t*)dz, idx, (const bf16_t*)y, mean, invstd, gamma, dgamma, dbeta,
                                                                  scale, shift, (bf16_t*)dy, nullptr, n, h, w, c, ho, wo, inv_m, dwb, dhb, bpc);
    return check_launch("maxpool_bn_bwd_apply (2x2)");
  }
  if (dtype == SH_F32)
    pool_bn_bwd_apply_kernel<float><<<grid, 256, 0, s>>>((const float*)dz, idx, (const float*)y, mean, invstd, gamma, dgamma, dbeta, scale, shift, (float*)dy, n, h, w, c, ho, wo, inv_m, dw_, dh_);
  else
    pool_bn_bwd_apply_kernel<bf16_t><<<grid, 256, 0, s>>>((const bf16_t*)dz, idx, (const bf16_t*)y, mean, invstd, gamma, dgamma, dbeta, scale, shift, (bf16_t*)dy, n, h, w, c, ho, wo, inv_m, dw_, dh_);
  return check_launch("maxpool_bn_bwd_apply");
}

int simhand_apply_relu_bitmask(const void* x, const uint8_t* mask, void* out, int64_t m, int c, int dtype, sh_stream_t stream) {
  SH_REQUIRE(x && mask && out, "apply_relu_bitmask: NULL pointer");
  const int ve = dtype == SH_F32 ? 4 : 8;
  SH_REQUIRE(c % ve == 0, "apply_relu_bitmask: c=%d not a multiple of %d", c, ve);
  const int64_t nvec = m * (c / ve);
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SH_PROF_BN, s, 0, 2.0 * (double)m * c * (dtype == SH_F32 ? 4 : 2));
  if (dtype == SH_F32) apply_bitmask_kernel<float><<<stream_grid(nvec), 256, 0, s>>>((const float*)x, mask, (float*)out, nvec);
  else apply_bitmask_kernel<bf16_t><<<stream_grid(nvec), 256, 0, s>>>((const bf16_t*)x, mask, (bf16_t*)out, nvec);
  return check_launch("apply_relu_bitmask");
}

size_t simhand_bn_fold_workspace_bytes(int cc, int cw) {
  int ks1, kp1, ks2, kp2;
  fold_split(ceil_div(cw, 64) * ceil_div(cc, 64), cw, &ks1, &kp1);
  fold_split(ceil_div(cw, 64) * ceil_div(cw, 64), cc, &ks2, &kp2);
  const size_t a = (size_t)ks1 * cc * cw, b = (size_t)ks2 * cw * cw;
  return ((a > b ? a : b) + (size_t)cw * cw) * sizeof(float);  // + the centred Gram matrix of the forward fold
}

int simhand_bn_fold_fwd(const float* w, int round_bf16, const float* s2, const float* t2, int cc, int cw, int64_t m, const float* gamma,
                        const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                        int64_t* num_batches_tracked, float* mean, float* invstd, float* scale, float* shift, float* ws2,
                        void* workspace, size_t workspace_bytes, sh_stream_t stream) {
  SH_REQUIRE(w && s2 && t2 && mean && invstd && scale && shift && ws2, "bn_fold_fwd: NULL pointer");
  SH_REQUIRE(cc >= 1 && cw >= 1 && m >= 1, "bn_fold_fwd: bad shape");
  SH_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "bn_fold_fwd: running stats must be given together");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SH_PROF_BN, s, 2.0 * cc * (double)cw * cw, 4.0 * ((double)cc * cw * 2 + (double)cw * cw));
  route_hit(SH_ROUTE_BN_FOLD_FWD);
  if (sw(SH_SW_FOLD_LEGACY)) {  // the round-5 chain (centre, product, slice sum, per-channel): four launches, same numbers bit for bit (tests)
    // ws2[c][j] = sum_i W[c][i] S2[i][j]
    {
      int ks, kper;
      const int tiles = ceil_div(cw, 64) * ceil_div(cc, 64);
      fold_split(tiles, cw, &ks, &kper);
      SH_REQUIRE(workspace && workspace_bytes >= simhand_bn_fold_workspace_bytes(cc, cw), "bn_fold_fwd: workspace too small");
      float* s2c = (float*)workspace + (workspace_bytes / sizeof(float) - (size_t)cw * cw);  // the tail of the workspace
      fold_center_kernel<<<ceil_div((int64_t)cw * cw, 256), 256, 0, s>>>(s2, t2, cw, 1.0 / (double)m, s2c);
      if (check_launch("bn_fold_fwd centre")) return 1;
      float* dst = ks > 1 ? (float*)workspace : ws2;
      fold_sgemm_kernel<float, 0><<<dim3(ceil_div(cw, 64), ceil_div(cc, 64), ks), 256, 0, s>>>(w, cw, 1, round_bf16, s2c, cw, 1, cc, cw, cw, dst,
                                                                                               nullptr, cw, kper);
      if (check_launch("bn_fold_fwd gemm")) return 1;
      if (ks > 1) {
        const int64_t count = (int64_t)cc * cw;
        fold_sum_kernel<float, 0><<<ceil_div(count, 256), 256, 0, s>>>(dst, ks, count, ws2, nullptr);
        if (check_launch("bn_fold_fwd sum")) return 1;
      }
    }
    bn_fold_fwd_kernel<<<cc, 256, 0, s>>>(w, round_bf16, s2, t2, cc, cw, m, gamma, beta, eps, momentum, running_mean, running_var,
                                          num_batches_tracked, mean, invstd, scale, shift, ws2);
    return check_launch("bn_fold_fwd");
  }
  // ws2[c][j] = sum_i W[c][i] S2c[i][j]: the Gram matrix is centred in the product's operand loader, the split-K slices are added by the
  // per-channel kernel behind it -- two launches (round 6; four until round 5: centre, product, slice sum, per-channel)
  int ks, kper;
  const int tiles = ceil_div(cw, 64) * ceil_div(cc, 64);
  fold_split(tiles, cw, &ks, &kper);
  SH_REQUIRE(workspace && workspace_bytes >= simhand_bn_fold_workspace_bytes(cc, cw), "bn_fold_fwd: workspace too small");
  float* dst = ks > 1 ? (float*)workspace : ws2;
  fold_sgemm_kernel<float, 0><<<dim3(ceil_div(cw, 64), ceil_div(cc, 64), ks), 256, 0, s>>>(w, cw, 1, round_bf16, s2, cw, 1, cc, cw, cw, dst,
                                                                                           nullptr, cw, kper, t2, 1.0 / (double)m);
  if (check_launch("bn_fold_fwd gemm")) return 1;
  bn_fold_fwd_kernel<<<cc, 256, 0, s>>>(w, round_bf16, s2, t2, cc, cw, m, gamma, beta, eps, momentum, running_mean, running_var,
                                        num_batches_tracked, mean, invstd, scale, shift, ws2, ks > 1 ? dst : nullptr, ks);
  return check_launch("bn_fold_fwd");
}

int simhand_bn_fold_bwd(const float* w, int round_bf16, const float* gmat, const float* s_, const float* ws2, const float* t2,
                        const float* mean, const float* invstd, const float* gamma, int cc, int cw, int64_t m, float* dgamma,
                        float* dbeta, float* dw, void* wa, float* bw, float* ccoef, void* wm, float* bias, int dtype,
                        void* workspace, size_t workspace_bytes, sh_stream_t stream) {
  SH_REQUIRE(w && gmat && s_ && ws2 && t2 && mean && invstd && dgamma && dbeta && dw && wa && bw && ccoef && wm && bias,
             "bn_fold_bwd: NULL pointer");
  SH_REQUIRE(cc % 32 == 0 && cw % 32 == 0 && m >= 1, "bn_fold_bwd: cc=%d / cw=%d must be multiples of 32", cc, cw);
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SH_PROF_BN, s, 2.0 * cc * (double)cw * cw, 4.0 * (double)cc * cw * 5);
  route_hit(SH_ROUTE_BN_FOLD_BWD);
  if (dtype == SH_F32)
    bn_fold_bwd_kernel<float><<<cc, 256, 0, s>>>(w, round_bf16, gmat, s_, ws2, t2, mean, invstd, gamma, cc, cw, m, dgamma, dbeta, dw, (float*)wa, bw, ccoef);
  else
    bn_fold_bwd_kernel<bf16_t><<<cc, 256, 0, s>>>(w, round_bf16, gmat, s_, ws2, t2, mean, invstd, gamma, cc, cw, m, dgamma, dbeta, dw, (bf16_t*)wa, bw, ccoef);
  if (check_launch("bn_fold_bwd")) return 1;
  // wm[i][j] = -sum_c W[c][i] bw[c][j]  (A(m = i, k = c) = w[c * cw + i])
  int ks, kper;
  fold_split(ceil_div(cw, 64) * ceil_div(cw, 64), cc, &ks, &kper);
  SH_REQUIRE(ks == 1 || (workspace && workspace_bytes >= (size_t)ks * cw * cw * sizeof(float)), "bn_fold_bwd: workspace too small");
  const dim3 grid(ceil_div(cw, 64), ceil_div(cw, 64), ks);
  float* part = (float*)workspace;
  if (dtype == SH_F32)
    fold_sgemm_kernel<float, 1><<<grid, 256, 0, s>>>(w, 1, cw, round_bf16, bw, cw, 1, cw, cw, cc, part, (float*)wm, cw, kper);
  else
    fold_sgemm_kernel<bf16_t, 1><<<grid, 256, 0, s>>>(w, 1, cw, round_bf16, bw, cw, 1, cw, cw, cc, part, (bf16_t*)wm, cw, kper);
  if (check_launch("bn_fold_bwd gemm")) return 1;
  if (sw(SH_SW_FOLD_LEGACY)) {  // the round-5 tail: slice sum and bias as two launches
    if (ks > 1) {
      const int64_t count = (int64_t)cw * cw;
      if (dtype == SH_F32) fold_sum_kernel<float, 1><<<ceil_div(count, 256), 256, 0, s>>>(part, ks, count, nullptr, (float*)wm);
      else fold_sum_kernel<bf16_t, 1><<<ceil_div(count, 256), 256, 0, s>>>(part, ks, count, nullptr, (bf16_t*)wm);
      if (check_launch("bn_fold_bwd sum")) return 1;
    }
    bn_fold_bias_kernel<<<ceil_div(cw, 32), 1024, 0, s>>>(w, round_bf16, ccoef, cc, cw, bias);
    return check_launch("bn_fold_bias");
  }
  // slice sum (when the product was split) + bias = C W in one launch
  const int64_t count = (int64_t)cw * cw;
  const int nsum = ks > 1 ? (int)ceil_div(count, 1024) : 0;
  if (dtype == SH_F32)
    fold_sum_bias_kernel<float><<<nsum + ceil_div(cw, 32), 1024, 0, s>>>(part, ks, count, (float*)wm, nsum, w, round_bf16, ccoef, cc, cw, bias);
  else
    fold_sum_bias_kernel<bf16_t><<<nsum + ceil_div(cw, 32), 1024, 0, s>>>(part, ks, count, (bf16_t*)wm, nsum, w, round_bf16, ccoef, cc, cw, bias);
  return check_launch("bn_fold_bias");
}

int simhand_colsum(const void* x, int64_t m, int c, int dtype, float* partial, float* out, sh_stream_t stream) {
  SH_REQUIRE(x && partial && out, "colsum: NULL pointer");
  SH_REQUIRE(c % (dtype == SH_F32 ? 4 : 8) == 0, "colsum: c=%d not a multiple of the 16-B vector", c);
  int rpb, nblk;
  col_plan(m, &rpb, &nblk);
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SH_PROF_MISC, s, 0, (double)m * c * (dtype == SH_F32 ? 4 : 2));
  if (dtype == SH_F32) colsum_kernel<float><<<nblk, 256, 0, s>>>((const float*)x, m, c, rpb, partial);
  else colsum_kernel<bf16_t><<<nblk, 256, 0, s>>>((const bf16_t*)x, m, c, rpb, partial);
  if (check_launch("colsum")) return 1;
  // reuse the bwd finalize reducer: "dbeta" slot = sum of s1, "dgamma" slot (s2 = 0) goes to scratch inside partial
  bn_bwd_finalize_kernel<float><<<ceil_div(c, 16), 1024, 0, s>>>(partial, nblk, c, partial + (int64_t)nblk * 2 * c, out);
  return check_launch("colsum finalize");
}

}  // extern "C"
