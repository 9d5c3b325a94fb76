// Implicit-GEMM convolution for gfx950: forward and data-gradient, NHWC activations,
// KRSC weights, bf16 (v_mfma_f32_16x16x32_bf16) or exact fp32 (v_mfma_f32_16x16x4_f32).
//
// Replaces (reference): torch.nn.Conv2d / Linear forward + their input-gradient as
// dispatched by torchvision's ResNet inside src/models/resnet_model.py:13-58 and the
// projection head src/models/unsupervised/simclr_model.py:22-39 (cuDNN / cuBLAS there).
//
// GEMM view:  Out[m][n] = sum_{tap,c} A(m,tap,c) * Wt[n][tap][c]
//   forward : m = output pixel, A = x at (ho*stride - pad + r, wo*stride - pad + s), n = cout
//   dgrad   : m = input pixel,  A = dy at ((hi + pad - r)/stride, ...) when divisible, n = cin,
//             Wt = weights permuted to [cin][r][s][cout]
// Tile: 128 (m) x BN (n) x 128 BYTES of k per step (64 bf16 / 32 fp32), 4 waves as 2x2, each
// wave 64 x BN/2 from 16x16 MFMA tiles.  Both operands are k-contiguous rows of 128 B, staged
// global -> VGPR -> LDS (register staging keeps per-row halo masking and, later, the fused
// BN-apply+ReLU prologue possible), double-buffered with one barrier per k-step.  LDS rows
// are XOR-swizzled at 16-B granularity (chunk ^= (row>>1)&7) so every ds_read_b128 lane
// group covers 16 distinct 16-B slots of the 256-B bank row (conflict-free).
// The MFMA is issued "swapped" (weights as the A operand) so a lane ends up with 4
// consecutive output channels of one pixel: 8-B (bf16) / 16-B (fp32) stores, and the
// per-channel BatchNorm partial sums reduce over lanes with 4 xor-shuffles.
// Block ids are remapped so all n-tiles of an m-tile run on one XCD (shared L2 for A rows).
#include "common.h"

namespace sh {

struct IgemmArgs {
  const void* a;      // source activations (x or dy), NHWC
  const void* w;      // [Ng][taps][Ca]
  void* out;          // [Mg][Ng]
  float* bn_partial;  // [m_tiles][2][Ng] or null
  long long Mg;       // destination pixels
  int Ng;             // destination channels
  int Ca;             // source channels
  int R, S, stride, pad;
  int Hd, Wd;         // destination spatial size
  int Hs, Ws;         // source spatial size
  int accumulate;
  int m_tiles, n_tiles;
};

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
  typedef __attribute__((ext_vector_type(8))) __bf16 frag_t;
  // 16 bytes = 8 bf16 = the whole K=32 slice of one lane
  __device__ static __forceinline__ f32x4 run(const uint4& a, const uint4& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(frag_t, a), __builtin_bit_cast(frag_t, b), c, 0, 0, 0);
  }
};
template <> struct Mma<float> {
  // 16 bytes = 4 floats; MFMA step e consumes element e of both operands (same k permutation on
  // both sides, so the dot product is unchanged)
  __device__ static __forceinline__ f32x4 run(const uint4& a, const uint4& b, f32x4 c) {
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), c, 0, 0, 0);
    return c;
  }
};

// bijective XCD-aware remap: hardware places block b on XCD b % 8; give each XCD a contiguous
// range of logical tile ids
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7;
  const int xcd = bid & 7, j = bid >> 3;
  const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + j;
}

__device__ __forceinline__ int swz(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }

template <typename T, bool DGRAD, int BN>
__global__ __launch_bounds__(256, 2) void igemm_kernel(IgemmArgs p) {
  constexpr int KE = 128 / (int)sizeof(T);  // elements of k per step
  constexpr int VE = 16 / (int)sizeof(T);   // elements per 16-B chunk
  constexpr int NB = BN / 32;               // B chunks per thread per step
  constexpr int NI = BN / 32;               // 16-wide n tiles per wave
  __shared__ __attribute__((aligned(16))) char smem[2 * 128 * 128 + 2 * BN * 128];
  char* sA = smem;
  char* sB = smem + 2 * 128 * 128;

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int li = lane & 15, g = lane >> 4;
  const int wm = wave >> 1, wn = wave & 1;
  const int logical = xcd_remap(blockIdx.x, gridDim.x);
  const int n_tile = logical % p.n_tiles;
  const int m_tile = logical / p.n_tiles;
  const long long m0 = (long long)m_tile * 128;
  const int n0 = n_tile * BN;

  // ---- per-thread loader state: 4 A rows (tid/8 + 32 i), chunk tid%8 --------------------
  const int chunk = tid & 7;
  const int lrow = tid >> 3;
  long long a_img[4];  // element offset of the image in the source tensor
  int a_h[4], a_w[4];  // fwd: hs0/ws0 = hd*stride - pad ; dgrad: hd + pad / wd + pad
  bool a_ok[4];
  const int hw_d = p.Hd * p.Wd;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const long long m = m0 + lrow + 32 * i;
    a_ok[i] = m < p.Mg;
    const long long mm = a_ok[i] ? m : 0;
    const int img = (int)(mm / hw_d);
    const int rem = (int)(mm - (long long)img * hw_d);
    const int hd = rem / p.Wd, wd = rem - hd * p.Wd;
    a_img[i] = (long long)img * p.Hs * p.Ws * p.Ca;
    if (DGRAD) {
      a_h[i] = hd + p.pad;
      a_w[i] = wd + p.pad;
    } else {
      a_h[i] = hd * p.stride - p.pad;
      a_w[i] = wd * p.stride - p.pad;
    }
  }
  const T* __restrict__ asrc = reinterpret_cast<const T*>(p.a);
  const T* __restrict__ wsrc = reinterpret_cast<const T*>(p.w);
  const int taps = p.R * p.S;
  const int csteps = p.Ca / KE;
  const int nk = taps * csteps;
  const long long wrow = (long long)taps * p.Ca;  // elements per weight row

  uint4 ra[4], rb[NB];
  auto load_step = [&](int ks) {
    const int tap = ks / csteps;
    const int c0 = (ks - tap * csteps) * KE + chunk * VE;
    const int r = tap / p.S, s = tap - r * p.S;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int hs, ws;
      bool ok = a_ok[i];
      if (DGRAD) {
        const int th = a_h[i] - r, tw = a_w[i] - s;
        if (p.stride == 2) {
          ok = ok && ((th | tw) & 1) == 0;
          hs = th >> 1;
          ws = tw >> 1;
        } else {
          hs = th;
          ws = tw;
        }
        ok = ok && th >= 0 && tw >= 0;
      } else {
        hs = a_h[i] + r;
        ws = a_w[i] + s;
      }
      ok = ok && (unsigned)hs < (unsigned)p.Hs && (unsigned)ws < (unsigned)p.Ws;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (ok) v = *reinterpret_cast<const uint4*>(asrc + a_img[i] + ((long long)hs * p.Ws + ws) * p.Ca + c0);
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int n = n0 + lrow + 32 * i;  // always < Ng (Ng % BN == 0)
      rb[i] = *reinterpret_cast<const uint4*>(wsrc + (long long)n * wrow + (long long)tap * p.Ca + c0);
    }
  };
  auto store_step = [&](int buf) {
    char* dA = sA + buf * (128 * 128);
    char* dB = sB + buf * (BN * 128);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = lrow + 32 * i;
      *reinterpret_cast<uint4*>(dA + row * 128 + swz(row, chunk) * 16) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int row = lrow + 32 * i;
      *reinterpret_cast<uint4*>(dB + row * 128 + swz(row, chunk) * 16) = rb[i];
    }
  };

  f32x4 acc[4][NI];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};

  load_step(0);
  store_step(0);
  __syncthreads();
  for (int ks = 0; ks < nk; ++ks) {
    const int buf = ks & 1;
    if (ks + 1 < nk) load_step(ks + 1);  // global loads in flight under the MFMAs
    const char* cA = sA + buf * (128 * 128) + (wm * 64 + li) * 128;
    const char* cB = sB + buf * (BN * 128) + (wn * (BN / 2) + li) * 128;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      uint4 fa[4], fb[NI];
      const int c = (4 * kk + g) ^ ((li >> 1) & 7);  // rows differ by multiples of 16 -> same swizzle key
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) fa[mi] = *reinterpret_cast<const uint4*>(cA + mi * 16 * 128 + c * 16);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) fb[ni] = *reinterpret_cast<const uint4*>(cB + ni * 16 * 128 + c * 16);
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = Mma<T>::run(fb[ni], fa[mi], acc[mi][ni]);
    }
    if (ks + 1 < nk) store_step(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: lane holds pixel (wm*64 + mi*16 + li), channels n0 + wn*BN/2 + ni*16 + 4g + r
  T* __restrict__ out = reinterpret_cast<T*>(p.out);
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
    const long long m = m0 + wm * 64 + mi * 16 + li;
    if (m < p.Mg) {
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        T* dst = out + m * p.Ng + n0 + wn * (BN / 2) + ni * 16 + 4 * g;
        float v[4] = {acc[mi][ni][0], acc[mi][ni][1], acc[mi][ni][2], acc[mi][ni][3]};
        if (sizeof(T) == 4) {
          float4* d4 = reinterpret_cast<float4*>(dst);
          if (p.accumulate) {
            const float4 o = *d4;
            v[0] += o.x; v[1] += o.y; v[2] += o.z; v[3] += o.w;
          }
          *d4 = make_float4(v[0], v[1], v[2], v[3]);
        } else {
          uint2* d2 = reinterpret_cast<uint2*>(dst);
          if (p.accumulate) {
            const uint2 o = *d2;
            v[0] += __uint_as_float(o.x << 16);
            v[1] += __uint_as_float(o.x & 0xffff0000u);
            v[2] += __uint_as_float(o.y << 16);
            v[3] += __uint_as_float(o.y & 0xffff0000u);
          }
          uint2 w;
          w.x = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
          w.y = (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16);
          *d2 = w;
        }
      }
    }
  }

  // ---- fused BatchNorm partial statistics of the fp32 accumulators ------------------------
  if (p.bn_partial != nullptr) {
    float* red = reinterpret_cast<float*>(smem);  // [2 (wm)][2][BN]; tiles are dead (loop ended with a barrier)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
          const float v = acc[mi][ni][r];  // rows beyond Mg are exact zeros (zero-filled A)
          s1 += v;
          s2 += v * v;
        }
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
          s1 += __shfl_xor(s1, o);
          s2 += __shfl_xor(s2, o);
        }
        if (li == 0) {
          const int c = wn * (BN / 2) + ni * 16 + 4 * g + r;
          red[(wm * 2 + 0) * BN + c] = s1;
          red[(wm * 2 + 1) * BN + c] = s2;
        }
      }
    }
    __syncthreads();
    if (tid < 2 * BN) {
      const int which = tid / BN, c = tid - which * BN;
      const float v = red[(0 * 2 + which) * BN + c] + red[(1 * 2 + which) * BN + c];
      p.bn_partial[((long long)m_tile * 2 + which) * p.Ng + n0 + c] = v;
    }
  }
}

template <typename T, bool DGRAD>
static int launch_igemm(const IgemmArgs& a, hipStream_t s) {
  const int nblk = a.m_tiles * a.n_tiles;
  if (a.Ng % 128 == 0) {
    igemm_kernel<T, DGRAD, 128><<<nblk, 256, 0, s>>>(a);
  } else {
    igemm_kernel<T, DGRAD, 64><<<nblk, 256, 0, s>>>(a);
  }
  return check_launch(DGRAD ? "conv2d_dgrad" : "conv2d_fwd");
}

static int check_desc(const sh_conv_desc* d, const char* who) {
  SH_REQUIRE(d != nullptr, "%s: desc is NULL", who);
  SH_REQUIRE(d->dtype == SH_F32 || d->dtype == SH_BF16, "%s: bad dtype %d", who, d->dtype);
  SH_REQUIRE(d->n >= 1 && d->h >= 1 && d->w >= 1 && d->r >= 1 && d->s >= 1, "%s: bad shape", who);
  SH_REQUIRE(d->stride == 1 || d->stride == 2, "%s: stride %d unsupported (ResNet uses 1 and 2)", who, d->stride);
  SH_REQUIRE(d->ho == (d->h + 2 * d->pad - d->r) / d->stride + 1 && d->wo == (d->w + 2 * d->pad - d->s) / d->stride + 1,
             "%s: ho/wo inconsistent with h/w/r/s/stride/pad", who);
  const int ke = d->dtype == SH_F32 ? 32 : 64;
  SH_REQUIRE(d->cin % ke == 0, "%s: cin=%d must be a multiple of %d for this dtype (pad the stem via im2col)", who, d->cin, ke);
  SH_REQUIRE(d->cout % 64 == 0, "%s: cout=%d must be a multiple of 64", who, d->cout);
  return 0;
}

}  // namespace sh

using namespace sh;

extern "C" {

int simhand_conv2d_fwd_stat_blocks(const sh_conv_desc* d) {
  if (!d) return 0;
  return ceil_div((long long)d->n * d->ho * d->wo, 128);
}

int simhand_conv2d_fwd(const sh_conv_desc* d, const void* x, const void* w, void* y, float* bn_partial, sh_stream_t stream) {
  if (check_desc(d, "conv2d_fwd")) return 1;
  SH_REQUIRE(x && w && y, "conv2d_fwd: NULL pointer");
  IgemmArgs a;
  a.a = x; a.w = w; a.out = y; a.bn_partial = bn_partial;
  a.Mg = (long long)d->n * d->ho * d->wo;
  a.Ng = d->cout; a.Ca = d->cin;
  a.R = d->r; a.S = d->s; a.stride = d->stride; a.pad = d->pad;
  a.Hd = d->ho; a.Wd = d->wo; a.Hs = d->h; a.Ws = d->w;
  a.accumulate = 0;
  a.m_tiles = ceil_div(a.Mg, 128);
  a.n_tiles = a.Ng % 128 == 0 ? a.Ng / 128 : a.Ng / 64;
  const double flops = 2.0 * (double)a.Mg * d->cout * d->cin * d->r * d->s;
  const double es = d->dtype == SH_F32 ? 4 : 2;
  const double bytes = es * ((double)d->n * d->h * d->w * d->cin + (double)a.Mg * d->cout + (double)d->cout * d->cin * d->r * d->s);
  ProfScope ps(SH_PROF_CONV_FWD, (hipStream_t)stream, flops, bytes);
  return d->dtype == SH_F32 ? launch_igemm<float, false>(a, (hipStream_t)stream) : launch_igemm<bf16_t, false>(a, (hipStream_t)stream);
}

int simhand_conv2d_dgrad(const sh_conv_desc* d, const void* dy, const void* wt, void* dx, int accumulate, sh_stream_t stream) {
  if (check_desc(d, "conv2d_dgrad")) return 1;
  SH_REQUIRE(dy && wt && dx, "conv2d_dgrad: NULL pointer");
  const int ke = d->dtype == SH_F32 ? 32 : 64;
  SH_REQUIRE(d->cout % ke == 0, "conv2d_dgrad: cout=%d must be a multiple of %d", d->cout, ke);
  SH_REQUIRE(d->cin % 64 == 0, "conv2d_dgrad: cin=%d must be a multiple of 64", d->cin);
  IgemmArgs a;
  a.a = dy; a.w = wt; a.out = dx; a.bn_partial = nullptr;
  a.Mg = (long long)d->n * d->h * d->w;
  a.Ng = d->cin; a.Ca = d->cout;
  a.R = d->r; a.S = d->s; a.stride = d->stride; a.pad = d->pad;
  a.Hd = d->h; a.Wd = d->w; a.Hs = d->ho; a.Ws = d->wo;
  a.accumulate = accumulate;
  a.m_tiles = ceil_div(a.Mg, 128);
  a.n_tiles = a.Ng % 128 == 0 ? a.Ng / 128 : a.Ng / 64;
  const long long mo = (long long)d->n * d->ho * d->wo;
  const double flops = 2.0 * (double)mo * d->cout * d->cin * d->r * d->s;
  const double es = d->dtype == SH_F32 ? 4 : 2;
  const double bytes = es * ((double)a.Mg * d->cin * (accumulate ? 2 : 1) + (double)mo * d->cout + (double)d->cout * d->cin * d->r * d->s);
  ProfScope ps(SH_PROF_CONV_DGRAD, (hipStream_t)stream, flops, bytes);
  return d->dtype == SH_F32 ? launch_igemm<float, true>(a, (hipStream_t)stream) : launch_igemm<bf16_t, true>(a, (hipStream_t)stream);
}

}  // extern "C"
