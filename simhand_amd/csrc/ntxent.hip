// Similarity-weighted NT-Xent over the gathered global batch (fwd + closed-form bwd),
// plus the joint-distance passes that feed the adaptive weights.
//
// Replaces (reference, /root/reference):
//   get_weights_linear / get_weights_nonlinear (+_with_pca)   src/models/utils.py:218-388
//   vanila_{,weights_,pos_weights_,neg_weights_}contrastive_loss :157-189,:391-501
//   and their autograd backward.
//
// Design (gfx950): the N x N similarity is never materialised.  A workgroup of
// 4 waves owns 64 local rows i and walks 64-column tiles j of Z_all staged in
// LDS; each wave computes the TRANSPOSED tile T[j][i] = z_j . z_i with the
// exact-f32 MFMA (v_mfma_f32_16x16x4_f32, K = 128 -> 32 MFMAs per 16x16 tile), so
// that lane (i = lane&15, g = lane>>4) holds T[4g..4g+3][i]: the row reduction
// over j is in-lane + two cross-lane adds, and in the backward the same
// registers are directly the A operand of the second product P . Z_j.
// The joint distances D (21 sqrt per pair, VALU-bound) are computed once per
// step into a [rows_loc][N] fp32 row block (HBM is cheaper than recomputing
// them three times) together with the global max/min/sum the weights need.
// Column ranges are split over blockIdx.y so the grid fills 256 CUs; partial
// sums are combined by a deterministic second stage (no atomics).
#include "common.h"

namespace sh {

constexpr int kDim = 128;   // projection width (output_dim)
constexpr int kTJ = 64;     // columns per LDS tile
constexpr int kLds = 132;   // padded row stride (floats): 528 B -> conflict-free ds_read_b128 / b32
constexpr int kMaxF = 64;   // joint features per row (42 = 21 x 2; 14 with PCA)

struct RowMap {
  int B, b_loc, pair_off;
};
__device__ __forceinline__ int global_row(const RowMap& m, int l) {
  return l < m.b_loc ? m.pair_off + l : m.B + m.pair_off + (l - m.b_loc);
}

struct Weighting {
  int wtype;  // sh_weight_type
  float dmax, dmin, mu, lambda;
};
__device__ __forceinline__ float weight_of(float d, const Weighting& w) {
  if (w.wtype == SH_W_EXPLICIT) return d;
  if (w.wtype == SH_W_LINEAR) return (w.dmax - d) / (w.dmax - w.dmin);  // utils.py:235,:259 (no epsilon: NaN if flat)
  return 1.0f / (1.0f + expf(w.lambda * (d - w.mu)));                   // utils.py:321,:344
}

// ------------------------------------------------------------------ distances
template <int MODE>
__device__ __forceinline__ float pair_accum(float acc, float dx, float dy) {
  if (MODE == SH_DIST_MPJPE) return acc + sqrtf(dx * dx + dy * dy);
  if (MODE == SH_DIST_W_ABS) {
    float m = (fabsf(dx) + fabsf(dy)) / 2.0f;
    return acc + m * m;
  }
  float m = (dx + dy) / 2.0f;  // W_O_ABS
  return acc + m * m;
}

// d+_k, src/models/utils.py:219-231 (PCA variants :265-274).  One block.
__global__ __launch_bounds__(1024) void pos_dist_kernel(const float* __restrict__ J, int B, int F, int mode,
                                                        float* __restrict__ dpos, double* __restrict__ stats) {
  __shared__ float smax[16], smin[16];
  __shared__ double ssum[16];
  float vmax = -INFINITY, vmin = INFINITY;
  double vsum = 0.0;
  const int nj = F / 2;
  for (int k = threadIdx.x; k < B; k += blockDim.x) {
    const float* a = J + (size_t)k * F;
    const float* b = J + (size_t)(B + k) * F;
    float d;
    if (mode == SH_DIST_L2) {
      float acc = 0.f;
      for (int f = 0; f < F; ++f) {
        float t = a[f] - b[f];
        acc += t * t;
      }
      d = sqrtf(acc);
    } else if (mode == SH_DIST_MPJPE) {
      float acc = 0.f;
      for (int j = 0; j < nj; ++j) {
        float dx = a[2 * j] - b[2 * j], dy = a[2 * j + 1] - b[2 * j + 1];
        acc += sqrtf(dx * dx + dy * dy);
      }
      d = acc / (float)nj;
    } else {
      float ax = 0.f, ay = 0.f;
      for (int j = 0; j < nj; ++j) {
        float dx = a[2 * j] - b[2 * j], dy = a[2 * j + 1] - b[2 * j + 1];
        if (mode == SH_DIST_W_ABS) {
          dx = fabsf(dx);
          dy = fabsf(dy);
        }
        ax += dx;
        ay += dy;
      }
      ax /= (float)nj;
      ay /= (float)nj;
      d = sqrtf(ax * ax + ay * ay);
    }
    dpos[k] = d;
    vmax = fmaxf(vmax, d);
    vmin = fminf(vmin, d);
    if (d != d) vmax = vmin = d;  // propagate NaN like torch.max/min
    vsum += (double)d;
  }
  vmax = wave_max(vmax);
  vmin = wave_min(vmin);
  for (int o = 32; o > 0; o >>= 1) vsum += __shfl_xor(vsum, o);
  const int wv = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    smax[wv] = vmax;
    smin[wv] = vmin;
    ssum[wv] = vsum;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < (int)(blockDim.x >> 6); ++i) {
      vmax = fmaxf(vmax, smax[i]);
      vmin = fminf(vmin, smin[i]);
      vsum += ssum[i];
    }
    stats[3] = vmax;
    stats[4] = vmin;
    stats[5] = vsum;
  }
}

// One joint distance, evaluated in exactly the order of neg_dist_kernel (accumulate over joints, then / nj or sqrt), so the
// in-tile (fused) and the materialised D agree bit for bit.  ji: this lane's row in registers, jj: the other row in LDS.
template <int MODE>
__device__ __forceinline__ float pair_dist(const float (&ji)[kMaxF], const float* __restrict__ jj, int F) {
  float acc = 0.f;
  if (MODE == SH_DIST_L2) {
#pragma unroll 2
    for (int f = 0; f < F; ++f) {
      const float t = ji[f] - jj[f];
      acc += t * t;
    }
    return sqrtf(acc);
  }
  const int nj = F / 2;
#pragma unroll 3
  for (int j = 0; j < nj; ++j) acc = pair_accum<MODE>(acc, ji[2 * j] - jj[2 * j], ji[2 * j + 1] - jj[2 * j + 1]);
  return MODE == SH_DIST_MPJPE ? acc / (float)nj : sqrtf(acc);
}

// D row block, src/models/utils.py:237-253 (PCA :280-293).  64x64 tile per block,
// 4x4 register block per thread.
template <int MODE>
__global__ __launch_bounds__(256) void neg_dist_kernel(const float* __restrict__ J, RowMap map, int N, int F,
                                                       float* __restrict__ D, double* __restrict__ partial) {
  __shared__ float ji[64][kMaxF + 1];
  __shared__ float jj[64][kMaxF + 1];
  __shared__ float smax[4], smin[4];
  __shared__ double ssum[4];
  const int rows_loc = 2 * map.b_loc;
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  for (int idx = threadIdx.x; idx < 64 * F; idx += 256) {
    int r = idx / F, f = idx - r * F;
    int lr = r0 + r, gc = c0 + r;
    ji[r][f] = lr < rows_loc ? J[(size_t)global_row(map, lr) * F + f] : 0.f;
    jj[r][f] = gc < N ? J[(size_t)gc * F + f] : 0.f;
  }
  __syncthreads();
  const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
  float acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = 0.f;
  if (MODE == SH_DIST_L2) {
    for (int f = 0; f < F; ++f) {
      float vi[4], vj[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) vi[a] = ji[4 * ty + a][f];
#pragma unroll
      for (int b = 0; b < 4; ++b) vj[b] = jj[tx + 16 * b][f];
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          float t = vi[a] - vj[b];
          acc[a][b] += t * t;
        }
    }
  } else {
    const int nj = F / 2;
    for (int j = 0; j < nj; ++j) {
      float xi[4], yi[4], xj[4], yj[4];
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        xi[a] = ji[4 * ty + a][2 * j];
        yi[a] = ji[4 * ty + a][2 * j + 1];
      }
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        xj[b] = jj[tx + 16 * b][2 * j];
        yj[b] = jj[tx + 16 * b][2 * j + 1];
      }
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = pair_accum<MODE>(acc[a][b], xi[a] - xj[b], yi[a] - yj[b]);
    }
  }
  float vmax = -INFINITY, vmin = INFINITY;
  double vsum = 0.0;
  const float inv_nj = (float)(F / 2);
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int lr = r0 + 4 * ty + a;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int gc = c0 + tx + 16 * b;
      float d = MODE == SH_DIST_MPJPE ? acc[a][b] / inv_nj : sqrtf(acc[a][b]);
      if (lr < rows_loc && gc < N) {
        if (D != nullptr) D[(size_t)lr * N + gc] = d;  // null: statistics only (the loss kernels recompute d in their tiles)
        vmax = fmaxf(vmax, d);
        vmin = fminf(vmin, d);
        if (d != d) vmax = vmin = d;
        vsum += (double)d;
      }
    }
  }
  vmax = wave_max(vmax);
  vmin = wave_min(vmin);
  for (int o = 32; o > 0; o >>= 1) vsum += __shfl_xor(vsum, o);
  const int wv = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    smax[wv] = vmax;
    smin[wv] = vmin;
    ssum[wv] = vsum;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < 4; ++i) {
      vmax = fmaxf(vmax, smax[i]);
      vmin = fminf(vmin, smin[i]);
      vsum += ssum[i];
    }
    double* p = partial + 3 * (size_t)(blockIdx.y * gridDim.x + blockIdx.x);
    p[0] = vmax;
    p[1] = vmin;
    p[2] = vsum;
  }
}

__global__ __launch_bounds__(1024) void dist_stats_reduce_kernel(const double* __restrict__ partial, int nblk,
                                                                 double* __restrict__ stats) {
  __shared__ double smax[16], smin[16], ssum[16];
  double vmax = -INFINITY, vmin = INFINITY, vsum = 0.0;
  bool nan = false;
  for (int i = threadIdx.x; i < nblk; i += blockDim.x) {
    double a = partial[3 * (size_t)i], b = partial[3 * (size_t)i + 1];
    nan |= (a != a) | (b != b);
    vmax = fmax(vmax, a);
    vmin = fmin(vmin, b);
    vsum += partial[3 * (size_t)i + 2];
  }
  if (nan) vmax = vmin = NAN;
  for (int o = 32; o > 0; o >>= 1) {
    double a = __shfl_xor(vmax, o), b = __shfl_xor(vmin, o);
    vmax = (a != a || vmax != vmax) ? NAN : fmax(vmax, a);
    vmin = (b != b || vmin != vmin) ? NAN : fmin(vmin, b);
    vsum += __shfl_xor(vsum, o);
  }
  const int wv = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    smax[wv] = vmax;
    smin[wv] = vmin;
    ssum[wv] = vsum;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < (int)(blockDim.x >> 6); ++i) {
      vmax = (smax[i] != smax[i] || vmax != vmax) ? NAN : fmax(vmax, smax[i]);
      vmin = (smin[i] != smin[i] || vmin != vmin) ? NAN : fmin(vmin, smin[i]);
      vsum += ssum[i];
    }
    stats[0] = vmax;
    stats[1] = vmin;
    stats[2] = vsum;
  }
}

// ------------------------------------------------------------------ NT-Xent
struct LossArgs {
  RowMap map;
  int N;
  int csplit;        // column splits (gridDim.y)
  int tiles_per_split;
  int rows_pad;      // rows_loc rounded up to 64
  int use_wneg, use_wpos, wtype;
  float inv_t;       // 1 / temperature
  float lambda_pos, lambda_neg;
};

__device__ __forceinline__ Weighting neg_weighting(const LossArgs& a, const double* stats) {
  Weighting w;
  w.wtype = a.use_wneg ? a.wtype : SH_W_NONE;
  w.dmax = (float)stats[0];
  w.dmin = (float)stats[1];
  w.mu = (float)(stats[2] / ((double)a.N * (double)a.N));
  w.lambda = a.lambda_neg;
  return w;
}
__device__ __forceinline__ Weighting pos_weighting(const LossArgs& a, const double* stats) {
  Weighting w;
  w.wtype = a.use_wpos ? a.wtype : SH_W_NONE;
  w.dmax = (float)stats[3];
  w.dmin = (float)stats[4];
  w.mu = (float)(stats[5] / (double)a.map.B);
  w.lambda = a.lambda_pos;
  return w;
}

__device__ __forceinline__ void stage_z_tile(float* zj, const float* __restrict__ Z, int j0, int N) {
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const int idx = threadIdx.x + 256 * t;
    const int r = idx >> 5, c4 = idx & 31;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (j0 + r < N) v = *reinterpret_cast<const float4*>(Z + (size_t)(j0 + r) * kDim + c4 * 4);
    *reinterpret_cast<float4*>(zj + r * kLds + c4 * 4) = v;
  }
}

// BWD = false: neg_partial[cs][lrow] = sum over this column range of exp(w s / t), j != i
// BWD = true : dz_partial[cs][lrow][128] = sum_j w e (1/neg_i + 1/neg_j) z_j
// DMODE < 0: the joint distances come from the materialised row block D.  DMODE = sh_dist_mode: FUSED -- the distance tile
// is computed here, next to the similarity tile, from the joint rows (J_all [N][F] staged in LDS per column tile, this lane's
// own row in registers); no [rows_loc][N] block exists in HBM (north star: "one LDS-tiled kernel").
template <bool BWD, int DMODE = -1>
__global__ __launch_bounds__(256) void ntxent_tile_kernel(LossArgs a, const float* __restrict__ Z,
                                                          const float* __restrict__ D, const double* __restrict__ stats,
                                                          const float* __restrict__ neg_all, float* __restrict__ out,
                                                          const float* __restrict__ J = nullptr, int F = 0) {
  __shared__ __attribute__((aligned(16))) float zj[kTJ * kLds];
  __shared__ float jjt[DMODE >= 0 ? kTJ * (kMaxF + 1) : 1];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int li = lane & 15, g = lane >> 4;
  const int rows_loc = 2 * a.map.b_loc;
  const int lrow = blockIdx.x * 64 + wave * 16 + li;
  const bool rvalid = lrow < rows_loc;
  const int grow = rvalid ? global_row(a.map, lrow) : 0;
  const Weighting wq = neg_weighting(a, stats);

  float4 zi[8];  // B operand: z_i, k = 16c + 4g + e
#pragma unroll
  for (int c = 0; c < 8; ++c)
    zi[c] = rvalid ? *reinterpret_cast<const float4*>(Z + (size_t)grow * kDim + 16 * c + 4 * g) : make_float4(0, 0, 0, 0);
  float inv_neg_i = 0.f;
  if (BWD) inv_neg_i = rvalid ? 1.0f / neg_all[grow] : 0.f;
  float ji[kMaxF];
  if constexpr (DMODE >= 0) {
#pragma unroll
    for (int f = 0; f < kMaxF; ++f) ji[f] = (rvalid && f < F) ? J[(size_t)grow * F + f] : 0.f;
  }

  float rowsum = 0.f;
  f32x4 dz[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) dz[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int tiles_total = (a.N + kTJ - 1) / kTJ;
  const int t_begin = blockIdx.y * a.tiles_per_split;
  const int t_end = min(tiles_total, t_begin + a.tiles_per_split);
  for (int jt = t_begin; jt < t_end; ++jt) {
    const int j0 = jt * kTJ;
    __syncthreads();
    stage_z_tile(zj, Z, j0, a.N);
    if constexpr (DMODE >= 0) {
      if (wq.wtype != SH_W_NONE)
        for (int idx = threadIdx.x; idx < kTJ * F; idx += 256) {
          const int r = idx / F, f = idx - r * F;
          jjt[r * (kMaxF + 1) + f] = j0 + r < a.N ? J[(size_t)(j0 + r) * F + f] : 0.f;
        }
    }
    __syncthreads();
#pragma unroll 1
    for (int js = 0; js < 4; ++js) {
      f32x4 t = (f32x4){0.f, 0.f, 0.f, 0.f};
      const float* arow = zj + (js * 16 + li) * kLds + 4 * g;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const float4 av = *reinterpret_cast<const float4*>(arow + 16 * c);
        t = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, zi[c].x, t, 0, 0, 0);
        t = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, zi[c].y, t, 0, 0, 0);
        t = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, zi[c].z, t, 0, 0, 0);
        t = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, zi[c].w, t, 0, 0, 0);
      }
      // t[r] = s(i = lrow, j = j0 + 16 js + 4g + r)
      float p[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = j0 + js * 16 + 4 * g + r;
        const bool ok = rvalid && j < a.N && j != grow;
        float w = 1.0f;
        if (wq.wtype != SH_W_NONE) {
          float d;
          if constexpr (DMODE >= 0) d = ok ? pair_dist<DMODE>(ji, jjt + (js * 16 + 4 * g + r) * (kMaxF + 1), F) : 0.f;
          else d = ok ? D[(size_t)lrow * a.N + j] : 0.f;
          w = weight_of(d, wq);
        }
        const float e = expf(t[r] * w * a.inv_t);  // exp(cov * w / temperature), utils.py:412-413
        if (BWD) {
          const float inj = ok ? 1.0f / neg_all[j] : 0.f;
          if (wq.wtype == SH_W_EXPLICIT) {
            // caller-supplied weights need not be symmetric (vanila_*_weights_contrastive_loss takes any (N,N) tensor,
            // src/models/utils.py:391-501): row j's term uses ITS weight of column i.  Single process only, so lrow == grow.
            const float wt = ok ? weight_of(D[(size_t)j * a.N + grow], wq) : 0.f;
            const float et = expf(t[r] * wt * a.inv_t);
            p[r] = ok ? w * e * inv_neg_i + wt * et * inj : 0.f;
          } else {
            p[r] = ok ? w * e * (inv_neg_i + inj) : 0.f;
          }
        } else {
          rowsum += ok ? e : 0.f;
        }
      }
      if (BWD) {
        // dz[i][:] += P[i][j] z_j : A = P (lane (i, g) holds j = 4g + e), B = z_j rows from LDS
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float* brow = zj + (js * 16 + 4 * g + e) * kLds + li;
#pragma unroll
          for (int tt = 0; tt < 8; ++tt) dz[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(p[e], brow[16 * tt], dz[tt], 0, 0, 0);
        }
      }
    }
  }
  if (BWD) {
    // C layout: lane holds column 16t + li of rows (block row0 + 16 wave + 4g + r)
    float* base = out + ((size_t)blockIdx.y * a.rows_pad + blockIdx.x * 64 + wave * 16) * kDim;
#pragma unroll
    for (int tt = 0; tt < 8; ++tt)
#pragma unroll
      for (int r = 0; r < 4; ++r) base[(size_t)(4 * g + r) * kDim + 16 * tt + li] = dz[tt][r];
  } else {
    rowsum += __shfl_xor(rowsum, 16);
    rowsum += __shfl_xor(rowsum, 32);
    if (g == 0) out[(size_t)blockIdx.y * a.rows_pad + blockIdx.x * 64 + wave * 16 + li] = rowsum;
  }
}

// one wave per local row: neg_i, positive term, per-row loss
__global__ __launch_bounds__(256) void ntxent_fwd_finalize_kernel(LossArgs a, const float* __restrict__ Z,
                                                                  const float* __restrict__ dpos,
                                                                  const double* __restrict__ stats,
                                                                  const float* __restrict__ neg_partial,
                                                                  float* __restrict__ neg_loc, float* __restrict__ loss_rows) {
  const int lane = threadIdx.x & 63;
  const int lrow = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int rows_loc = 2 * a.map.b_loc;
  if (lrow >= rows_loc) return;
  const int grow = global_row(a.map, lrow);
  const int k = grow < a.map.B ? grow : grow - a.map.B;
  const int prow = grow < a.map.B ? grow + a.map.B : grow - a.map.B;
  const float2 zi = *reinterpret_cast<const float2*>(Z + (size_t)grow * kDim + 2 * lane);
  const float2 zp = *reinterpret_cast<const float2*>(Z + (size_t)prow * kDim + 2 * lane);
  const float sp = wave_sum(zi.x * zp.x + zi.y * zp.y);
  if (lane == 0) {
    float neg = 0.f;
    for (int c = 0; c < a.csplit; ++c) neg += neg_partial[(size_t)c * a.rows_pad + lrow];
    const Weighting wp = pos_weighting(a, stats);
    const float w = wp.wtype == SH_W_NONE ? 1.0f : weight_of(dpos[k], wp);
    neg_loc[lrow] = neg;
    loss_rows[lrow] = logf(neg) - sp * w * a.inv_t;  // -log(pos/neg), utils.py:420-426
  }
}

__global__ __launch_bounds__(1024) void sum_rows_kernel(const float* __restrict__ x, int n, double scale, float* __restrict__ out) {
  __shared__ double ssum[16];
  double v = 0.0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) v += (double)x[i];
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  if ((threadIdx.x & 63) == 0) ssum[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < (int)(blockDim.x >> 6); ++i) v += ssum[i];
    out[0] = (float)(v * scale);
  }
}

// dZ_loc = dloss/(N t) * (sum_cs partial - 2 w+ z_pair)
__global__ __launch_bounds__(256) void ntxent_bwd_finalize_kernel(LossArgs a, const float* __restrict__ Z,
                                                                  const float* __restrict__ dpos,
                                                                  const double* __restrict__ stats,
                                                                  const float* __restrict__ dz_partial,
                                                                  const float* __restrict__ dloss, float* __restrict__ dZ) {
  const int rows_loc = 2 * a.map.b_loc;
  const int idx = blockIdx.x * 256 + threadIdx.x;  // float4 index
  const int lrow = idx >> 5, c4 = idx & 31;
  if (lrow >= rows_loc) return;
  const int grow = global_row(a.map, lrow);
  const int k = grow < a.map.B ? grow : grow - a.map.B;
  const int prow = grow < a.map.B ? grow + a.map.B : grow - a.map.B;
  float4 acc = make_float4(0, 0, 0, 0);
  for (int c = 0; c < a.csplit; ++c) {
    const float4 v = *reinterpret_cast<const float4*>(dz_partial + ((size_t)c * a.rows_pad + lrow) * kDim + c4 * 4);
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  const Weighting wp = pos_weighting(a, stats);
  const float w = wp.wtype == SH_W_NONE ? 1.0f : weight_of(dpos[k], wp);
  const float4 zp = *reinterpret_cast<const float4*>(Z + (size_t)prow * kDim + c4 * 4);
  const float sc = (dloss ? dloss[0] : 1.0f) * a.inv_t / (float)a.N;
  float4 o;
  o.x = sc * (acc.x - 2.f * w * zp.x);
  o.y = sc * (acc.y - 2.f * w * zp.y);
  o.z = sc * (acc.z - 2.f * w * zp.z);
  o.w = sc * (acc.w - 2.f * w * zp.w);
  *reinterpret_cast<float4*>(dZ + (size_t)lrow * kDim + c4 * 4) = o;
}

// explicit weight tensors for the functional surface (get_weights_* return values)
__global__ __launch_bounds__(256) void weights_from_dist_kernel(const float* __restrict__ d, long long count, int wtype,
                                                                const double* __restrict__ stats, int which, double mean_count,
                                                                float lambda, float* __restrict__ w) {
  Weighting q;
  q.wtype = wtype;
  q.dmax = (float)stats[which ? 3 : 0];
  q.dmin = (float)stats[which ? 4 : 1];
  q.mu = (float)(stats[which ? 5 : 2] / mean_count);
  q.lambda = lambda;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < count; i += (long long)gridDim.x * 256) w[i] = weight_of(d[i], q);
}

static int make_args(const sh_ntxent_params* p, LossArgs* a) {
  SH_REQUIRE(p != nullptr, "ntxent: params is NULL");
  SH_REQUIRE(p->dim == kDim, "ntxent: projection width %d unsupported (kernel is built for output_dim = 128)", p->dim);
  SH_REQUIRE(p->B >= 1 && p->b_loc >= 1 && p->pair_off >= 0 && p->pair_off + p->b_loc <= p->B,
             "ntxent: bad row partition B=%d b_loc=%d pair_off=%d", p->B, p->b_loc, p->pair_off);
  SH_REQUIRE(p->temperature > 0.f, "ntxent: temperature must be > 0");
  SH_REQUIRE(p->weight_type >= SH_W_NONE && p->weight_type <= SH_W_EXPLICIT, "ntxent: bad weight_type %d", p->weight_type);
  a->map = {p->B, p->b_loc, p->pair_off};
  a->N = 2 * p->B;
  const int rb = ceil_div(2 * p->b_loc, 64);
  const int tiles = ceil_div(a->N, kTJ);
  int cs = ceil_div(1024, rb);
  if (cs > tiles) cs = tiles;
  if (cs < 1) cs = 1;
  a->tiles_per_split = ceil_div(tiles, cs);
  a->csplit = ceil_div(tiles, a->tiles_per_split);
  a->rows_pad = rb * 64;
  a->wtype = p->weight_type;
  a->use_wneg = (p->use_wneg && p->weight_type != SH_W_NONE) ? 1 : 0;
  a->use_wpos = (p->use_wpos && p->weight_type != SH_W_NONE) ? 1 : 0;
  a->inv_t = 1.0f / p->temperature;
  a->lambda_pos = p->lambda_pos;
  a->lambda_neg = p->lambda_neg;
  return 0;
}

// launch the tile kernel for the fused-distance mode `dm`
template <bool BWD>
static void launch_tile_fused(int dm, dim3 grid, hipStream_t s, const LossArgs& a, const float* Z, const double* stats, const float* neg_all,
                              float* out, const float* J, int F) {
  switch (dm) {
    case SH_DIST_MPJPE: ntxent_tile_kernel<BWD, SH_DIST_MPJPE><<<grid, 256, 0, s>>>(a, Z, nullptr, stats, neg_all, out, J, F); break;
    case SH_DIST_W_ABS: ntxent_tile_kernel<BWD, SH_DIST_W_ABS><<<grid, 256, 0, s>>>(a, Z, nullptr, stats, neg_all, out, J, F); break;
    case SH_DIST_W_O_ABS: ntxent_tile_kernel<BWD, SH_DIST_W_O_ABS><<<grid, 256, 0, s>>>(a, Z, nullptr, stats, neg_all, out, J, F); break;
    default: ntxent_tile_kernel<BWD, SH_DIST_L2><<<grid, 256, 0, s>>>(a, Z, nullptr, stats, neg_all, out, J, F); break;
  }
}

static int check_fused(const LossArgs& a, const float* J_all, int F, int dist_mode, const char* who) {
  SH_REQUIRE(a.wtype != SH_W_EXPLICIT, "%s: explicit weight tensors have no joints to fuse", who);
  SH_REQUIRE(!a.use_wneg || J_all, "%s: J_all required when negatives are weighted", who);
  SH_REQUIRE(F >= 1 && F <= kMaxF && dist_mode >= 0 && dist_mode <= SH_DIST_L2 && (dist_mode == SH_DIST_L2 || F % 2 == 0), "%s: bad F=%d / dist_mode=%d",
             who, F, dist_mode);
  return 0;
}

}  // namespace sh

using namespace sh;

extern "C" {

int simhand_pos_dist(const float* J_all, int B, int F, int dist_mode, float* d_pos, double* stats, sh_stream_t stream) {
  SH_REQUIRE(J_all && d_pos && stats, "pos_dist: NULL pointer");
  SH_REQUIRE(B >= 1 && F >= 1 && F <= kMaxF, "pos_dist: bad B=%d F=%d", B, F);
  SH_REQUIRE(dist_mode >= 0 && dist_mode <= SH_DIST_L2, "pos_dist: bad dist_mode %d", dist_mode);
  SH_REQUIRE(dist_mode == SH_DIST_L2 || (F % 2) == 0, "pos_dist: joint modes need F even (xy pairs), got %d", F);
  ProfScope ps(SH_PROF_LOSS, (hipStream_t)stream, 0, 0);
  pos_dist_kernel<<<1, 1024, 0, (hipStream_t)stream>>>(J_all, B, F, dist_mode, d_pos, stats);
  return check_launch("pos_dist");
}

size_t simhand_neg_dist_workspace_bytes(int rows_loc, int N) {
  return (size_t)ceil_div(rows_loc, 64) * ceil_div(N, 64) * 3 * sizeof(double);
}

int simhand_neg_dist(const float* J_all, int B, int F, int dist_mode, int b_loc, int pair_off, float* D_loc, double* stats,
                     void* workspace, size_t workspace_bytes, sh_stream_t stream) {
  SH_REQUIRE(J_all && stats && workspace, "neg_dist: NULL pointer");  // D_loc may be NULL: statistics only
  SH_REQUIRE(B >= 1 && F >= 1 && F <= kMaxF, "neg_dist: bad B=%d F=%d", B, F);
  SH_REQUIRE(b_loc >= 1 && pair_off >= 0 && pair_off + b_loc <= B, "neg_dist: bad partition");
  SH_REQUIRE(dist_mode >= 0 && dist_mode <= SH_DIST_L2, "neg_dist: bad dist_mode %d", dist_mode);
  SH_REQUIRE(dist_mode == SH_DIST_L2 || (F % 2) == 0, "neg_dist: joint modes need F even, got %d", F);
  const int N = 2 * B, rows = 2 * b_loc;
  SH_REQUIRE(workspace_bytes >= simhand_neg_dist_workspace_bytes(rows, N), "neg_dist: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  dim3 grid(ceil_div(N, 64), ceil_div(rows, 64));
  RowMap map{B, b_loc, pair_off};
  double* partial = (double*)workspace;
  ProfScope ps(SH_PROF_LOSS, s, 0, D_loc ? (double)rows * N * 4 : 0.0);
  switch (dist_mode) {
    case SH_DIST_MPJPE: neg_dist_kernel<SH_DIST_MPJPE><<<grid, 256, 0, s>>>(J_all, map, N, F, D_loc, partial); break;
    case SH_DIST_W_ABS: neg_dist_kernel<SH_DIST_W_ABS><<<grid, 256, 0, s>>>(J_all, map, N, F, D_loc, partial); break;
    case SH_DIST_W_O_ABS: neg_dist_kernel<SH_DIST_W_O_ABS><<<grid, 256, 0, s>>>(J_all, map, N, F, D_loc, partial); break;
    default: neg_dist_kernel<SH_DIST_L2><<<grid, 256, 0, s>>>(J_all, map, N, F, D_loc, partial); break;
  }
  if (check_launch("neg_dist")) return 1;
  dist_stats_reduce_kernel<<<1, 1024, 0, s>>>(partial, (int)(grid.x * grid.y), stats);
  return check_launch("neg_dist_reduce");
}

int simhand_weights_from_dist(const float* dist, int64_t count, int weight_type, const double* stats, int positive,
                              double mean_count, float lambda, float* weights, sh_stream_t stream) {
  SH_REQUIRE(dist && stats && weights && count >= 1, "weights_from_dist: bad arguments");
  SH_REQUIRE(weight_type == SH_W_LINEAR || weight_type == SH_W_NONLINEAR, "weights_from_dist: bad weight_type %d", weight_type);
  int64_t g = (count + 255) / 256;
  if (g > 4096) g = 4096;
  ProfScope ps(SH_PROF_LOSS, (hipStream_t)stream, 0, (double)count * 8);
  weights_from_dist_kernel<<<(int)g, 256, 0, (hipStream_t)stream>>>(dist, count, weight_type, stats, positive, mean_count, lambda, weights);
  return check_launch("weights_from_dist");
}

size_t simhand_ntxent_workspace_bytes(const sh_ntxent_params* p) {
  LossArgs a;
  if (make_args(p, &a)) return 0;
  const size_t fwd = ((size_t)a.csplit * a.rows_pad + a.rows_pad) * sizeof(float);
  const size_t bwd = (size_t)a.csplit * a.rows_pad * kDim * sizeof(float);
  return fwd > bwd ? fwd : bwd;
}

int simhand_ntxent_fwd(const sh_ntxent_params* p, const float* Z_all, const float* D_loc, const float* d_pos,
                       const double* stats, float* neg_loc, float* loss_part, void* workspace, size_t workspace_bytes,
                       sh_stream_t stream) {
  LossArgs a;
  if (make_args(p, &a)) return 1;
  SH_REQUIRE(Z_all && stats && neg_loc && loss_part && workspace, "ntxent_fwd: NULL pointer");
  SH_REQUIRE(!a.use_wneg || D_loc, "ntxent_fwd: D_loc required when negatives are weighted");
  SH_REQUIRE(!a.use_wpos || d_pos, "ntxent_fwd: d_pos required when positives are weighted");
  SH_REQUIRE(workspace_bytes >= simhand_ntxent_workspace_bytes(p), "ntxent_fwd: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const int rows = 2 * a.map.b_loc;
  float* neg_partial = (float*)workspace;
  float* loss_rows = neg_partial + (size_t)a.csplit * a.rows_pad;
  ProfScope ps(SH_PROF_LOSS, s, 2.0 * rows * a.N * kDim, 0);
  route_hit(SH_ROUTE_NTXENT_FWD);
  dim3 grid(a.rows_pad / 64, a.csplit);
  ntxent_tile_kernel<false><<<grid, 256, 0, s>>>(a, Z_all, D_loc, stats, nullptr, neg_partial);
  if (check_launch("ntxent_fwd tile")) return 1;
  ntxent_fwd_finalize_kernel<<<ceil_div(rows, 4), 256, 0, s>>>(a, Z_all, d_pos, stats, neg_partial, neg_loc, loss_rows);
  if (check_launch("ntxent_fwd finalize")) return 1;
  sum_rows_kernel<<<1, 1024, 0, s>>>(loss_rows, rows, 1.0 / (double)a.N, loss_part);
  return check_launch("ntxent_fwd sum");
}

int simhand_ntxent_fwd_fused(const sh_ntxent_params* p, const float* Z_all, const float* J_all, int F, int dist_mode, const float* d_pos,
                             const double* stats, float* neg_loc, float* loss_part, void* workspace, size_t workspace_bytes, sh_stream_t stream) {
  LossArgs a;
  if (make_args(p, &a)) return 1;
  SH_REQUIRE(Z_all && stats && neg_loc && loss_part && workspace, "ntxent_fwd_fused: NULL pointer");
  if (check_fused(a, J_all, F, dist_mode, "ntxent_fwd_fused")) return 1;
  SH_REQUIRE(!a.use_wpos || d_pos, "ntxent_fwd_fused: d_pos required when positives are weighted");
  SH_REQUIRE(workspace_bytes >= simhand_ntxent_workspace_bytes(p), "ntxent_fwd_fused: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const int rows = 2 * a.map.b_loc;
  float* neg_partial = (float*)workspace;
  float* loss_rows = neg_partial + (size_t)a.csplit * a.rows_pad;
  ProfScope ps(SH_PROF_LOSS, s, 2.0 * rows * a.N * kDim, 0);
  route_hit(SH_ROUTE_NTXENT_FWD);
  route_hit(SH_ROUTE_NTXENT_FUSED_DIST);
  dim3 grid(a.rows_pad / 64, a.csplit);
  launch_tile_fused<false>(dist_mode, grid, s, a, Z_all, stats, nullptr, neg_partial, J_all, F);
  if (check_launch("ntxent_fwd_fused tile")) return 1;
  ntxent_fwd_finalize_kernel<<<ceil_div(rows, 4), 256, 0, s>>>(a, Z_all, d_pos, stats, neg_partial, neg_loc, loss_rows);
  if (check_launch("ntxent_fwd_fused finalize")) return 1;
  sum_rows_kernel<<<1, 1024, 0, s>>>(loss_rows, rows, 1.0 / (double)a.N, loss_part);
  return check_launch("ntxent_fwd_fused sum");
}

int simhand_ntxent_bwd_fused(const sh_ntxent_params* p, const float* Z_all, const float* J_all, int F, int dist_mode, const float* d_pos,
                             const double* stats, const float* neg_all, const float* dloss, float* dZ_loc, void* workspace,
                             size_t workspace_bytes, sh_stream_t stream) {
  LossArgs a;
  if (make_args(p, &a)) return 1;
  SH_REQUIRE(Z_all && stats && neg_all && dZ_loc && workspace, "ntxent_bwd_fused: NULL pointer");
  if (check_fused(a, J_all, F, dist_mode, "ntxent_bwd_fused")) return 1;
  SH_REQUIRE(!a.use_wpos || d_pos, "ntxent_bwd_fused: d_pos required when positives are weighted");
  SH_REQUIRE(workspace_bytes >= simhand_ntxent_workspace_bytes(p), "ntxent_bwd_fused: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const int rows = 2 * a.map.b_loc;
  float* dz_partial = (float*)workspace;
  ProfScope ps(SH_PROF_LOSS, s, 4.0 * rows * a.N * kDim, 0);
  route_hit(SH_ROUTE_NTXENT_BWD);
  route_hit(SH_ROUTE_NTXENT_FUSED_DIST);
  dim3 grid(a.rows_pad / 64, a.csplit);
  launch_tile_fused<true>(dist_mode, grid, s, a, Z_all, stats, neg_all, dz_partial, J_all, F);
  if (check_launch("ntxent_bwd_fused tile")) return 1;
  ntxent_bwd_finalize_kernel<<<ceil_div((int64_t)rows * 32, 256), 256, 0, s>>>(a, Z_all, d_pos, stats, dz_partial, dloss, dZ_loc);
  return check_launch("ntxent_bwd_fused finalize");
}

int simhand_ntxent_bwd(const sh_ntxent_params* p, const float* Z_all, const float* D_loc, const float* d_pos,
                       const double* stats, const float* neg_all, const float* dloss, float* dZ_loc, void* workspace,
                       size_t workspace_bytes, sh_stream_t stream) {
  LossArgs a;
  if (make_args(p, &a)) return 1;
  SH_REQUIRE(Z_all && stats && neg_all && dZ_loc && workspace, "ntxent_bwd: NULL pointer");
  SH_REQUIRE(!a.use_wneg || D_loc, "ntxent_bwd: D_loc required when negatives are weighted");
  SH_REQUIRE(!a.use_wpos || d_pos, "ntxent_bwd: d_pos required when positives are weighted");
  SH_REQUIRE(workspace_bytes >= simhand_ntxent_workspace_bytes(p), "ntxent_bwd: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const int rows = 2 * a.map.b_loc;
  float* dz_partial = (float*)workspace;
  ProfScope ps(SH_PROF_LOSS, s, 4.0 * rows * a.N * kDim, 0);
  route_hit(SH_ROUTE_NTXENT_BWD);
  dim3 grid(a.rows_pad / 64, a.csplit);
  ntxent_tile_kernel<true><<<grid, 256, 0, s>>>(a, Z_all, D_loc, stats, neg_all, dz_partial);
  if (check_launch("ntxent_bwd tile")) return 1;
  ntxent_bwd_finalize_kernel<<<ceil_div((int64_t)rows * 32, 256), 256, 0, s>>>(a, Z_all, d_pos, stats, dz_partial, dloss, dZ_loc);
  return check_launch("ntxent_bwd finalize");
}

}  // extern "C"
