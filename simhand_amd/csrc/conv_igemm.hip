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
//   dgrad, stride 2: input pixels are split into the 4 (h%2, w%2) parity classes; a class only
//             visits the taps whose parity can reach it (9 tap-visits in total instead of 36;
//             for 1x1/2 three classes have no tap at all and are just zero-filled).
// Tile: 128 (m) x BN (n) x 128 BYTES of k per step (64 bf16 / 32 fp32), 4 waves as 2x2, each
// wave 64 x BN/2 from 16x16 MFMA tiles.  Both operands are k-contiguous rows of 128 B, staged
// global -> VGPR -> LDS (register staging keeps per-row halo masking and, later, a fused
// BN-apply+ReLU prologue possible); one LDS tile per block and three blocks per CU (see SH_IGEMM_NBUF).  LDS rows
// are XOR-swizzled at 16-B granularity (chunk ^= (row>>1)&7) so every ds_read_b128 lane
// group covers 16 distinct 16-B slots of the 256-B bank row (conflict-free).
// The MFMA is issued "swapped" (weights as the A operand) so a lane ends up with 4
// output channels of one pixel, and the weight rows are fed to the MFMA tiles in a permuted order
// so that those are 4*NI CONSECUTIVE channels (32 B bf16 / 64 B fp32 per lane and pixel): the epilogue
// stores 16-B vectors straight from registers (no LDS staging, no extra barriers), and the per-channel
// BatchNorm partial sums reduce over the 16 pixel lanes with four DPP row rotations.
// Block ids are remapped so all n-tiles of an m-tile run on one XCD (shared L2 for A rows).
#include "common.h"
#include "conv_1x1.h"

#include <stdlib.h>
#include "conv3x3_c64.h"
#include "conv3x3_ring.h"

namespace sh {

struct IgemmArgs {
  const void* a;      // source activations (x or dy), NHWC
  const void* w;      // [Ng][taps][Ca]
  void* out;          // [Mg][Ng]
  float* bn_partial;  // [m_tiles][2][Ng] or null
  long long Mg;       // destination pixels (per parity class when classes == 4)
  int Ng;             // destination channels
  int Ca;             // source channels
  int R, S, stride, pad;
  int Hd, Wd;         // destination spatial size
  int Hs, Ws;         // source spatial size
  int accumulate;     // 0: store, 1: out += result, 2: out = result + res_grad * bit(res_mask)
  const void* res_grad;        // accumulate == 2: gradient of the block output, laid out like out
  const unsigned char* res_mask; // accumulate == 2: its ReLU bit mask [pixel][Ng / VE]
  int m_tiles, n_tiles;
  int classes;        // 1, or 4 = stride-2 dgrad parity classes
  int Hq, Wq;         // class grid (ceil(Hd/2), ceil(Wd/2)) when classes == 4
  FastDiv div_hw;     // pixel-grid size (Hd*Wd, or Hq*Wq with parity classes)
  FastDiv div_w;      // grid width (Wd or Wq)
  int stem_hp, stem_wp;  // > 0: direct 7x7/2 stem from the zero-padded NHWC4 input [N][hp][wp][4] (see stem_fwd below)
  // dgrad only: BatchNorm-backward partial sums of the unit whose incoming gradient this call produces (sh_bn_bwd_fuse)
  const void* fy;              // that unit's raw conv output, laid out like out; null = no fusion
  const float* fscale;         // fmode 2: ReLU mask recomputed from fy * fscale + fshift > 0
  const float* fshift;
  const unsigned char* fmask;  // fmode 3 / 4: ReLU bit mask [pixel][Ng / VE]
  int fmode;                   // 4: the STORED gradient is masked (out = result * bit), sums of it only (fy unused)
  float* fpartial;             // [m_tiles * classes][2][Ng]: sum g, sum g * y; null = no fusion
  const float* bias;           // dgrad: per destination channel, added to the fp32 result before any accumulate / rounding
  // forward only (bf16): epilogue out = act(acc * ep_scale[c] + ep_shift[c] (+ ep_res)) -- BatchNorm with statistics known
  // BEFORE the convolution runs (eval mode, or train mode with the 1x1 Gram-matrix statistics), residual add, ReLU and its
  // 1-bit mask [pixel][Ng / 8]; the raw convolution output is never stored.  ep_scale null = off
  const float* ep_scale;
  const float* ep_shift;
  const void* ep_res;
  unsigned char* ep_mask;
  int ep_relu;
  // dgrad, 1x1 / stride 1 only: a SECOND K segment -- out = a * w^T + a2 * w2^T in one accumulation (the two terms of the folded
  // BatchNorm backward's input gradient).  Ca is then the total reduction length lda + Ca2; the k-steps past lda / KE read
  // a2 [Mg][Ca2] and w2 [Ng][Ca2].  lda = channels per pixel / per tap of a and w (= Ca without a second segment).
  const void* a2 = nullptr;
  const void* w2 = nullptr;
  int Ca2 = 0;
  int lda = 0;
  // 128-row kernel only: this launch covers the m-tiles from m_tile_base on (the ragged last wave of a 256 x 256 launch,
  // see split256) and writes its partial-sum rows from part_row_base on
  int m_tile_base = 0;
  int part_row_base = 0;
  // fp8 forward on the 256 x 256 kernel (BASELINE configs[4]): e4m3 operands, result = acc * x_state[1] * w_state[1] (= 1 / (scale_x scale_w))
  const float* x_state = nullptr;
  const float* w_state = nullptr;
  // dgrad, 1x1 / stride 1, 256 x 256 kernel, masked store (fmode 4) only: out = mask(result + S), S = `sub` [n][Hd/2][Wd/2][Ng] at the even
  // pixels and zero elsewhere (the stride-2 shortcut's dense data gradient: see Gemm1x1Args::sub).  null = off
  const void* sub = nullptr;
};

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
  typedef __attribute__((ext_vector_type(8))) __bf16 frag_t;
  // 16 bytes = 8 bf16 = the whole K=32 slice of one lane
  __device__ static __forceinline__ f32x4 run(const uint4& a, const uint4& b, f32x4 c) {
    return sh_mfma16(a, b, c);
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

__device__ __forceinline__ unsigned add_bf16x2(unsigned a, unsigned b) {
  const float lo = h16_lo(a) + h16_lo(b);
  const float hi = h16_hi(a) + h16_hi(b);
  return pack_bf16x2(lo, hi);
}

// sum over the 16 lanes of a DPP row (lanes sharing lane>>4), result in every lane: four row_ror
// rotations on the VALU instead of ds_bpermute shuffles through the LDS crossbar
__device__ __forceinline__ float row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));  // row_ror:8
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));  // row_ror:4
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));  // row_ror:2
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));  // row_ror:1
  return v;
}

// destination pixel of GEMM row m (parity classes: m runs over the class grid Hq x Wq)
struct Pixel {
  int img, hd, wd;
  bool ok;
};

// LDS tile buffering.  1 (default): one 32-KB operand tile per block, two barriers per k-step, THREE blocks per CU --
// the next step's operands are still prefetched into registers under the MFMAs, and the third resident block hides
// more global-load latency than a second LDS buffer did (measured on MI355X: 3x3 layers 700-820 -> 810-950 TFLOP/s).
// 2: double-buffered tiles, one barrier per k-step, two blocks per CU (kept for experiments).
#ifndef SH_IGEMM_NBUF
#define SH_IGEMM_NBUF 1
#endif
#ifndef SH_IGEMM_MINB
#define SH_IGEMM_MINB 3
#endif
// 128 B of zeros: halo / out-of-range / past-the-end operand loads read it instead of being skipped -- a conditional global
// load in a main loop makes hipcc's waitcnt pass drain the whole queue (it merges the pending state of both paths)
__device__ uint4 g_zero_page[8];

template <typename T, bool DGRAD, int BN>
__global__ __launch_bounds__(256, SH_IGEMM_NBUF == 1 ? SH_IGEMM_MINB : 2) void igemm_kernel(IgemmArgs p) {
  constexpr int KE = 128 / (int)sizeof(T);  // elements of k per step
  constexpr int VE = 16 / (int)sizeof(T);   // elements per 16-B chunk
  constexpr int NB = BN / 32;               // B chunks per thread per step
  constexpr int NI = BN / 32;               // 16-wide n tiles per wave
  constexpr int NBUF = SH_IGEMM_NBUF;
  constexpr int TILE_BYTES = NBUF * 128 * 128 + NBUF * BN * 128;
  __shared__ __attribute__((aligned(16))) char smem[TILE_BYTES];
  char* sA = smem;
  char* sB = smem + NBUF * 128 * 128;

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int li = lane & 15, g = lane >> 4;
  const int wm = wave >> 1, wn = wave & 1;
  int logical = xcd_remap(blockIdx.x, gridDim.x);
  // order [m_tile][class][n_tile]: every XCD range holds all 4 parity classes (their work differs: 1/2/2/4
  // taps for 3x3, 1/0/0/0 for 1x1), and the n-tiles of one m-tile stay adjacent
  const int n_tile = logical % p.n_tiles;
  logical /= p.n_tiles;
  const int cls = logical % p.classes;  // 0 unless classes == 4
  const int m_tile = logical / p.classes + p.m_tile_base;
  const int prow = m_tile - p.m_tile_base + p.part_row_base;  // row of the partial-sum buffers
  const int ph = cls >> 1, pw = cls & 1;
  const unsigned m0 = (unsigned)m_tile * 128u;  // Mg < 2^31 (checked on the host)
  const int n0 = n_tile * BN;
  const bool par = DGRAD && p.classes == 4;

  auto decode = [&](unsigned m) __attribute__((always_inline)) -> Pixel {
    Pixel r;
    r.ok = m < (unsigned)p.Mg;
    const unsigned mm = r.ok ? m : 0u;
    const unsigned img = fdiv(mm, p.div_hw);
    const unsigned rem = mm - img * p.div_hw.d;
    const unsigned hq = fdiv(rem, p.div_w);
    const unsigned wq = rem - hq * p.div_w.d;
    r.img = (int)img;
    if (par) {
      r.hd = 2 * (int)hq + ph;
      r.wd = 2 * (int)wq + pw;
      r.ok = r.ok && r.hd < p.Hd && r.wd < p.Wd;
    } else {
      r.hd = (int)hq;
      r.wd = (int)wq;
    }
    return r;
  };

  // tap enumeration: all R x S taps, or (parity classes) only those with (hd + pad - r) even.
  // Source pixel of tap (tr, ts): (h0 + dh*tr, w0 + dh*ts) with dh = +1 (forward) / -1 (dgrad).
  const int r0 = par ? ((ph + p.pad) & 1) : 0, s0 = par ? ((pw + p.pad) & 1) : 0;
  const int rstep = par ? 2 : 1;
  const int ntr = par ? (p.R - r0 + 1) / 2 : p.R;
  const int nts = par ? (p.S - s0 + 1) / 2 : p.S;
  const int csteps = p.Ca / KE;
  const int nk = (ntr > 0 && nts > 0) ? ntr * nts * csteps : 0;
  constexpr int dh = DGRAD ? -1 : 1;
  // a parity class no filter tap can reach (three of the four for 1x1/2) contributes zeros: with `out += result`
  // there is nothing to do -- the shortcut gradient of a stride-2 block is merged after conv1's, touching only 1/4 of dx
  if (DGRAD && nk == 0 && p.accumulate == 1) return;

  // ---- per-thread loader state: 4 A rows (tid/8 + 32 i), chunk tid%8 --------------------
  const int chunk = tid & 7;
  const int lrow = tid >> 3;
  const T* __restrict__ asrc = reinterpret_cast<const T*>(p.a);
  const T* __restrict__ wsrc = reinterpret_cast<const T*>(p.w);
  const T* pa0; const T* pa1; const T* pa2; const T* pa3;   // row base at tap (0,0), channel chunk 0
  int h00, h01, h02, h03, w00, w01, w02, w03;              // source coords at tap (0,0); h = -2^20 for dead rows
  auto init_row = [&](int i, const T*& pa, int& h0, int& w0) __attribute__((always_inline)) {
    const Pixel px = decode(m0 + lrow + 32 * i);
    if (!DGRAD && p.stem_wp > 0) {
      // filter row r of output pixel (ho, wo) = 8 taps x 4 channels = 32 contiguous elements at padded (2ho + r, 2wo);
      // a 128-B k-step is 2 such rows (bf16) or 1 (fp32); the generic tap walk then advances by whole k-steps
      // (host sets R = k-steps, S = 1, Ca = KE, Ws = wp / 8 so that aoff = l_tr * rows_per_step * wp * 4).
      constexpr int CPR = 32 / VE;
      pa = asrc + (((long long)px.img * p.stem_hp + 2 * px.hd) * p.stem_wp + 2 * px.wd) * 4 + (chunk / CPR) * (p.stem_wp * 4) +
           (chunk % CPR) * VE;
      h0 = px.ok ? 0 : -(1 << 20);
      w0 = 0;
      return;
    }
    if (DGRAD) {
      h0 = par ? (px.hd + p.pad - r0) >> 1 : px.hd + p.pad;
      w0 = par ? (px.wd + p.pad - s0) >> 1 : px.wd + p.pad;
    } else {
      h0 = px.hd * p.stride - p.pad;
      w0 = px.wd * p.stride - p.pad;
    }
    pa = asrc + ((long long)px.img * p.Hs * p.Ws + (long long)h0 * p.Ws + w0) * p.lda + chunk * VE;
    if (!px.ok) h0 = -(1 << 20);  // fails every bounds test below
  };
  init_row(0, pa0, h00, w00);
  init_row(1, pa1, h01, w01);
  init_row(2, pa2, h02, w02);
  init_row(3, pa3, h03, w03);
  const long long wrow = (long long)p.R * p.S * p.lda;  // elements per weight row
  const T* pb0 = wsrc + (long long)(n0 + lrow) * wrow + chunk * VE;  // rows n0 + lrow + 32 i (always < Ng)
  long long wrow32 = 32 * wrow;
  const int cs1 = p.lda / KE;  // k-steps of the first K segment (== csteps without a second one)
  // LDS store offsets of this thread's rows.  The B (weight) tile uses its own swizzle key because its rows are
  // read in a permuted order (see the fragment offsets below).
  // rows lrow + 32 i share one swizzle key (the keys only look at row bits 1..4): offsets differ by 32 * 128 bytes
  const int st0 = lrow * 128 + swz(lrow, chunk) * 16;
  // bf16: rows are read 8q + t (q, t = 0..3) apart -> key from bits 1 and 3..4; fp32: natural row order
  auto key_b = [](int row) __attribute__((always_inline)) -> int {
    return sizeof(T) == 2 ? (((row >> 1) & 1) | (((row >> 3) & 3) << 1)) : ((row >> 1) & 7);
  };
  // weight row (within this wave's half) fed to MFMA tile ni, operand row m -- and therefore the output channel of
  // accumulator register r = m & 3 of lane group g = m >> 2.  bf16: (ni>>1)*32 + g*8 + (ni&1)*4 + r, so the 8 bf16 of
  // tiles (2j, 2j+1) form ONE 16-B vector and the four lane groups cover 64 contiguous bytes per pixel and store
  // instruction.  fp32: ni*16 + 4g + r (the natural order already gives 16 B per lane, 64 B per pixel).
  auto chan_of = [](int ni, int m) __attribute__((always_inline)) -> int {
    return sizeof(T) == 2 ? ((ni >> 1) * 32 + (m >> 2) * 8 + (ni & 1) * 4 + (m & 3)) : (ni * 16 + m);
  };
  const int sb0 = lrow * 128 + (chunk ^ key_b(lrow)) * 16;

  // iteration state of the NEXT load (wave-uniform)
  int l_cs = 0, l_tr = 0, l_ts = 0;
  // staging registers as named scalars: an array here ends up in scratch / promoted to LDS (hipcc 7.2)
  uint4 ra0, ra1, ra2, ra3, rb0, rb1, rb2 = make_uint4(0, 0, 0, 0), rb3 = make_uint4(0, 0, 0, 0);
  const char* zsrc = reinterpret_cast<const char*>(g_zero_page) + (tid & 7) * 16;
  auto load_step = [&](bool live) __attribute__((always_inline)) {
    if constexpr (DGRAD) if (p.a2 != nullptr && l_cs == cs1) {
      // second K segment (1x1 / stride 1: pixel index == m): re-base the row and weight pointers so that the same
      // l_cs * KE offsets walk a2 / w2; dead rows keep their h0 = -2^20
      const T* a2 = reinterpret_cast<const T*>(p.a2);
      const long long back = (long long)chunk * VE - (long long)cs1 * KE;
      pa0 = a2 + (long long)(m0 + lrow) * p.Ca2 + back;
      pa1 = a2 + (long long)(m0 + lrow + 32) * p.Ca2 + back;
      pa2 = a2 + (long long)(m0 + lrow + 64) * p.Ca2 + back;
      pa3 = a2 + (long long)(m0 + lrow + 96) * p.Ca2 + back;
      pb0 = reinterpret_cast<const T*>(p.w2) + (long long)(n0 + lrow) * p.Ca2 + back;
      wrow32 = 32ll * p.Ca2;
    }
    const int hoff = dh * l_tr, woff = dh * l_ts;
    const int aoff = (hoff * p.Ws + woff) * p.lda + l_cs * KE;                              // elements, |.| < 2^31
    const int boff = ((r0 + rstep * l_tr) * p.S + (s0 + rstep * l_ts)) * p.lda + l_cs * KE;
    // branch-free: halo pixels, dead rows and the step past the end (live == false) read the zero page
    auto load_a = [&](const T* pa, int h0, int w0) __attribute__((always_inline)) -> uint4 {
      const bool ok = live && (unsigned)(h0 + hoff) < (unsigned)p.Hs && (unsigned)(w0 + woff) < (unsigned)p.Ws;
      const char* src = ok ? reinterpret_cast<const char*>(pa + aoff) : zsrc;
      return *reinterpret_cast<const uint4*>(src);
    };
    ra0 = load_a(pa0, h00, w00);
    ra1 = load_a(pa1, h01, w01);
    ra2 = load_a(pa2, h02, w02);
    ra3 = load_a(pa3, h03, w03);
    auto load_b = [&](long long row_off) __attribute__((always_inline)) -> uint4 {
      const char* src = live ? reinterpret_cast<const char*>(pb0 + row_off + boff) : zsrc;  // (a re-based pb0 alone is not a valid address)
      return *reinterpret_cast<const uint4*>(src);
    };
    rb0 = load_b(0);
    rb1 = load_b(wrow32);
    if (NB == 4) {
      rb2 = load_b(2 * wrow32);
      rb3 = load_b(3 * wrow32);
    }
    if (++l_cs == csteps) {
      l_cs = 0;
      if (++l_ts == nts) {
        l_ts = 0;
        ++l_tr;
      }
    }
  };
  auto store_step = [&](int buf) __attribute__((always_inline)) {
    char* dA = sA + buf * (128 * 128);
    char* dB = sB + buf * (BN * 128);
    *reinterpret_cast<uint4*>(dA + st0) = ra0;
    *reinterpret_cast<uint4*>(dA + st0 + 32 * 128) = ra1;
    *reinterpret_cast<uint4*>(dA + st0 + 64 * 128) = ra2;
    *reinterpret_cast<uint4*>(dA + st0 + 96 * 128) = ra3;
    *reinterpret_cast<uint4*>(dB + sb0) = rb0;
    *reinterpret_cast<uint4*>(dB + sb0 + 32 * 128) = rb1;
    if (NB == 4) {
      *reinterpret_cast<uint4*>(dB + sb0 + 64 * 128) = rb2;
      *reinterpret_cast<uint4*>(dB + sb0 + 96 * 128) = rb3;
    }
  };

  f32x4 acc[4][NI];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};

  if (nk > 0) {
    load_step(true);
    store_step(0);
  }
  __syncthreads();
  // A fragment offsets: rows differ by multiples of 16 -> one swizzle key per lane
  const int fkey = (li >> 1) & 7;
  const int fo0 = (g ^ fkey) * 16;  // second half of the k-step: chunk 4 + g -> the same offset with bit 6 flipped
  const int fa_base = (wm * 64 + li) * 128;
  // B fragment rows follow chan_of(): the epilogue stores 16-B vectors per lane straight from registers
  // (no LDS staging, no extra barriers) and every store instruction covers whole 64-B sectors.  chan_of(ni, li) =
  // chan_of(0, li) + a constant whose bits (2 and 5; fp32: 4 and 5) the swizzle key ignores: one offset per lane.
  const int rowb0 = wn * (BN / 2) + chan_of(0, li);
  const int fbo = rowb0 * 128 + ((g ^ key_b(rowb0)) * 16);
  for (int ks = 0; ks < nk; ++ks) {
    const int buf = NBUF == 2 ? (ks & 1) : 0;
    load_step(ks + 1 < nk);  // global loads in flight under the MFMAs (unconditional: see g_zero_page)
    const char* cA = sA + buf * (128 * 128) + fa_base;
    const char* cB = sB + buf * (BN * 128);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      uint4 fa[4], fb[NI];
      const int fo = kk == 0 ? fo0 : (fo0 ^ 64);
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) fa[mi] = *reinterpret_cast<const uint4*>(cA + mi * 16 * 128 + fo);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
        fb[ni] = *reinterpret_cast<const uint4*>(cB + ((kk == 0 ? fbo : (fbo ^ 64)) + (chan_of(ni, 0) - chan_of(0, 0)) * 128));
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = Mma<T>::run(fb[ni], fa[mi], acc[mi][ni]);
    }
    if (NBUF == 1) __syncthreads();  // single buffer: every wave is done reading before the tile is overwritten
    store_step(NBUF == 2 ? (buf ^ 1) : 0);  // after the last step: zeros into a tile nobody reads again
    __syncthreads();
  }

  // ---- fused BatchNorm partial statistics of the fp32 accumulators (registers -> LDS -> global) ----
  // lane holds pixel (wm*64 + mi*16 + li), channels wn*BN/2 + chan_of(ni, 4g + r)
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
        s1 = row16_sum(s1);
        s2 = row16_sum(s2);
        if (li == 0) {
          const int c = wn * (BN / 2) + chan_of(ni, 4 * g + r);
          red[(wm * 2 + 0) * BN + c] = s1;
          red[(wm * 2 + 1) * BN + c] = s2;
        }
      }
    }
    __syncthreads();
    if (tid < 2 * BN) {
      const int which = tid / BN, c = tid - which * BN;
      const float v = red[(0 * 2 + which) * BN + c] + red[(1 * 2 + which) * BN + c];
      p.bn_partial[((long long)prow * 2 + which) * p.Ng + n0 + c] = v;
    }
  }

  // ---- epilogue: 16-B vectors straight from registers; chunk j of lane group g = channels j*4*VE + g*VE .. ----
  {
    constexpr int NCH = 4 * NI / VE;  // 16-B chunks per lane per pixel
    T* __restrict__ out = reinterpret_cast<T*>(p.out);
    const int ch0 = n0 + wn * (BN / 2) + g * VE;
    // destination pixel of accumulator row mi (linear NHWC pixel index, or -1)
    auto pixel_of = [&](int mi) __attribute__((always_inline)) -> long long {
      const unsigned mrow = m0 + wm * 64 + mi * 16 + li;
      long long pix = (long long)mrow;
      bool ok = pix < p.Mg;
      if (par) {
        const Pixel px = decode(mrow);
        ok = px.ok;
        pix = ((long long)px.img * p.Hd + px.hd) * p.Wd + px.wd;
      }
      return ok ? pix : -1;
    };
    // value of chunk j of row mi after the accumulate mode, as stored
    auto chunk_out = [&](int mi, int j, long long pix, unsigned keep = 0xffu) __attribute__((always_inline)) -> uint4 {
      uint4 v;
      const int ch = ch0 + j * 4 * VE;
      if (sizeof(T) == 4) {
        f32x4 a0 = acc[mi][j % NI];
        if (DGRAD && p.bias != nullptr) {
          const float4 b0 = *reinterpret_cast<const float4*>(p.bias + ch);
          a0[0] += b0.x; a0[1] += b0.y; a0[2] += b0.z; a0[3] += b0.w;
        }
        v.x = __float_as_uint(a0[0]);
        v.y = __float_as_uint(a0[1]);
        v.z = __float_as_uint(a0[2]);
        v.w = __float_as_uint(a0[3]);
      } else {
        f32x4 lo = acc[mi][(2 * j) % NI], hi = acc[mi][(2 * j + 1) % NI];
        if (DGRAD && p.bias != nullptr) {
          const float4 b0 = *reinterpret_cast<const float4*>(p.bias + ch), b1 = *reinterpret_cast<const float4*>(p.bias + ch + 4);
          lo[0] += b0.x; lo[1] += b0.y; lo[2] += b0.z; lo[3] += b0.w;
          hi[0] += b1.x; hi[1] += b1.y; hi[2] += b1.z; hi[3] += b1.w;
        }
        v.x = pack_bf16x2(lo[0], lo[1]);
        v.y = pack_bf16x2(lo[2], lo[3]);
        v.z = pack_bf16x2(hi[0], hi[1]);
        v.w = pack_bf16x2(hi[2], hi[3]);
      }
      T* dst = out + pix * p.Ng + ch;
      if (p.accumulate == 2) {
        // identity-branch gradient g = res_grad * relu_mask, merged here instead of being materialised
        const uint4 o = *reinterpret_cast<const uint4*>(reinterpret_cast<const T*>(p.res_grad) + pix * p.Ng + ch);
        const unsigned bits = p.res_mask[pix * (p.Ng / VE) + ch / VE];
        if (sizeof(T) == 4) {
          v.x = __float_as_uint(__uint_as_float(v.x) + ((bits & 1u) ? __uint_as_float(o.x) : 0.f));
          v.y = __float_as_uint(__uint_as_float(v.y) + ((bits & 2u) ? __uint_as_float(o.y) : 0.f));
          v.z = __float_as_uint(__uint_as_float(v.z) + ((bits & 4u) ? __uint_as_float(o.z) : 0.f));
          v.w = __float_as_uint(__uint_as_float(v.w) + ((bits & 8u) ? __uint_as_float(o.w) : 0.f));
        } else {
          const unsigned m0w = ((bits & 1u) ? 0x0000ffffu : 0u) | ((bits & 2u) ? 0xffff0000u : 0u);
          const unsigned m1w = ((bits & 4u) ? 0x0000ffffu : 0u) | ((bits & 8u) ? 0xffff0000u : 0u);
          const unsigned m2w = ((bits & 16u) ? 0x0000ffffu : 0u) | ((bits & 32u) ? 0xffff0000u : 0u);
          const unsigned m3w = ((bits & 64u) ? 0x0000ffffu : 0u) | ((bits & 128u) ? 0xffff0000u : 0u);
          v.x = add_bf16x2(v.x, o.x & m0w);
          v.y = add_bf16x2(v.y, o.y & m1w);
          v.z = add_bf16x2(v.z, o.z & m2w);
          v.w = add_bf16x2(v.w, o.w & m3w);
        }
      } else if (p.accumulate) {
        const uint4 o = *reinterpret_cast<const uint4*>(dst);
        if (sizeof(T) == 4) {
          v.x = __float_as_uint(__uint_as_float(v.x) + __uint_as_float(o.x));
          v.y = __float_as_uint(__uint_as_float(v.y) + __uint_as_float(o.y));
          v.z = __float_as_uint(__uint_as_float(v.z) + __uint_as_float(o.z));
          v.w = __float_as_uint(__uint_as_float(v.w) + __uint_as_float(o.w));
        } else {
          v.x = add_bf16x2(v.x, o.x);
          v.y = add_bf16x2(v.y, o.y);
          v.z = add_bf16x2(v.z, o.z);
          v.w = add_bf16x2(v.w, o.w);
        }
      }
      if (DGRAD && p.fmode == 4) {  // the stored gradient is the masked one
        if (sizeof(T) == 4) {
          v.x = (keep & 1u) ? v.x : 0u; v.y = (keep & 2u) ? v.y : 0u; v.z = (keep & 4u) ? v.z : 0u; v.w = (keep & 8u) ? v.w : 0u;
        } else {
          v.x &= ((keep & 1u) ? 0x0000ffffu : 0u) | ((keep & 2u) ? 0xffff0000u : 0u);
          v.y &= ((keep & 4u) ? 0x0000ffffu : 0u) | ((keep & 8u) ? 0xffff0000u : 0u);
          v.z &= ((keep & 16u) ? 0x0000ffffu : 0u) | ((keep & 32u) ? 0xffff0000u : 0u);
          v.w &= ((keep & 64u) ? 0x0000ffffu : 0u) | ((keep & 128u) ? 0xffff0000u : 0u);
        }
      }
      *reinterpret_cast<uint4*>(dst) = v;
      return v;
    };
    if (!DGRAD && sizeof(T) == 2 && p.ep_scale != nullptr) {
      // BatchNorm (statistics known up front) + residual + ReLU applied to the fp32 accumulators; ReLU bit mask out
      const bf16_t* __restrict__ res = reinterpret_cast<const bf16_t*>(p.ep_res);
      bf16_t* __restrict__ outb = reinterpret_cast<bf16_t*>(p.out);
#pragma unroll
      for (int j = 0; j < NCH; ++j) {
        const int ch = ch0 + j * 32;
        float sc[8], sh[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          sc[e] = p.ep_scale[ch + e];
          sh[e] = p.ep_shift[ch + e];
        }
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
          const long long pix = pixel_of(mi);
          if (pix < 0) continue;
          const f32x4 lo = acc[mi][(2 * j) % NI], hi = acc[mi][(2 * j + 1) % NI];
          float o[8];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            o[e] = lo[e] * sc[e] + sh[e];
            o[4 + e] = hi[e] * sc[4 + e] + sh[4 + e];
          }
          if (res != nullptr) {
            float q[8];
            Vec16<bf16_t>::load(res + pix * p.Ng + ch, q);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] += q[e];
          }
          if (p.ep_relu) {
            unsigned bits = 0;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              bits |= (o[e] > 0.f ? 1u : 0u) << e;
              o[e] = o[e] > 0.f ? o[e] : 0.f;
            }
            if (p.ep_mask != nullptr) p.ep_mask[pix * (p.Ng / 8) + ch / 8] = (unsigned char)bits;
          }
          Vec16<bf16_t>::store(outb + pix * p.Ng + ch, o);
        }
      }
    } else if (!DGRAD || p.fpartial == nullptr) {
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) {
        const long long pix = pixel_of(mi);
        if (pix < 0) continue;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
          unsigned keep = 0xffu;
          if (DGRAD && p.fmode == 4) keep = p.fmask[pix * (p.Ng / VE) + (ch0 + j * 4 * VE) / VE];  // masked store, no sums
          chunk_out(mi, j, pix, keep);
        }
      }
    } else {
      // Fused BatchNorm-backward partial sums of the PREVIOUS unit: the stored gradient (rounded as stored) is that
      // unit's incoming gradient da; g = da * relu'(.), sums of g and g * y per channel over this tile's pixels.
      // Chunk-major order keeps one chunk's coefficients / sums live at a time.
      float* red = reinterpret_cast<float*>(smem);  // [2 (wm)][2][BN]; tiles are dead (loop ended with a barrier)
      const T* __restrict__ fy = reinterpret_cast<const T*>(p.fy);
      long long pixs[4];
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) pixs[mi] = pixel_of(mi);
#pragma unroll
      for (int j = 0; j < NCH; ++j) {
        const int ch = ch0 + j * 4 * VE;
        float sc[VE], sh[VE], s1[VE], s2[VE];
#pragma unroll
        for (int e = 0; e < VE; ++e) sc[e] = sh[e] = s1[e] = s2[e] = 0.f;
        if (p.fmode == 2) {  // 16-B loads (the per-element conditional form compiles to one dword load per coefficient)
#pragma unroll
          for (int e = 0; e < VE; e += 4) {
            const float4 a4 = *reinterpret_cast<const float4*>(p.fscale + ch + e), b4 = *reinterpret_cast<const float4*>(p.fshift + ch + e);
            sc[e] = a4.x; sc[e + 1] = a4.y; sc[e + 2] = a4.z; sc[e + 3] = a4.w;
            sh[e] = b4.x; sh[e + 1] = b4.y; sh[e + 2] = b4.z; sh[e + 3] = b4.w;
          }
        }
        // the four y rows / mask bytes of this channel group are requested together, branch-free (rows past the range read row 0)
        uint4 yq[4];
        unsigned bq[4];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
          const long long pc = pixs[mi] < 0 ? 0 : pixs[mi];
          yq[mi] = p.fmode != 4 ? *reinterpret_cast<const uint4*>(fy + pc * p.Ng + ch) : make_uint4(0, 0, 0, 0);
          bq[mi] = p.fmode >= 3 ? (unsigned)p.fmask[pc * (p.Ng / VE) + ch / VE] : 0xffu;
        }
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
          const long long pix = pixs[mi];
          if (pix < 0) continue;
          float yy[VE], gg[VE];
          if (sizeof(T) == 4) {
            yy[0] = __uint_as_float(yq[mi].x); yy[1] = __uint_as_float(yq[mi].y); yy[2] = __uint_as_float(yq[mi].z); yy[3] = __uint_as_float(yq[mi].w);
          } else {
            const unsigned y4[4] = {yq[mi].x, yq[mi].y, yq[mi].z, yq[mi].w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              yy[(2 * i) % VE] = h16_lo(y4[i]);
              yy[(2 * i + 1) % VE] = h16_hi(y4[i]);
            }
          }
          const unsigned bits = bq[mi];
          const uint4 v = chunk_out(mi, j, pix, bits);
          if (sizeof(T) == 4) {
            gg[0] = __uint_as_float(v.x); gg[1] = __uint_as_float(v.y); gg[2] = __uint_as_float(v.z); gg[3] = __uint_as_float(v.w);
          } else {
            const unsigned w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              gg[(2 * i) % VE] = h16_lo(w4[i]);
              gg[(2 * i + 1) % VE] = h16_hi(w4[i]);
            }
          }
#pragma unroll
          for (int e = 0; e < VE; ++e) {
            bool on = true;
            if (p.fmode == 2) on = yy[e] * sc[e] + sh[e] > 0.f;
            else if (p.fmode == 3) on = (bits >> e) & 1u;  // mode 4: the value is already masked
            const float gv = on ? gg[e] : 0.f;
            s1[e] += gv;
            s2[e] += gv * yy[e];
          }
        }
#pragma unroll
        for (int e = 0; e < VE; ++e) {
          const float t1 = row16_sum(s1[e]), t2 = row16_sum(s2[e]);
          if (li == 0) {
            const int c = wn * (BN / 2) + g * VE + j * 4 * VE + e;
            red[(wm * 2 + 0) * BN + c] = t1;
            red[(wm * 2 + 1) * BN + c] = t2;
          }
        }
      }
      __syncthreads();
      if (tid < 2 * BN) {
        const int which = tid / BN, c = tid - which * BN;
        const float v = red[(0 * 2 + which) * BN + c] + red[(1 * 2 + which) * BN + c];
        p.fpartial[(((long long)prow * p.classes + cls) * 2 + which) * p.Ng + n0 + c] = v;
      }
    }
  }
}


// ======================================================================================================================
// 256 x 256 x 64 tile kernel for the MFMA-bound bf16 layers (destination channels a multiple of 256: the 3x3 layers of
// stages 3-4, the long-K 1x1 layers).  The 128 x 128 kernel above moves 256 B of operands from L2 per MFMA; this one
// 128 B, and its operands never pass through registers: every 16-B chunk goes global -> LDS by LDS-DMA
// (global_load_lds_dwordx4), a whole k-step (64 KB) ahead of its use, into the second of two 64-KB stages.
//   * 8 waves as 2 (m) x 4 (n), each 128 x 64 = 8 x 4 MFMA tiles (128 accumulator registers), one block per CU;
//   * a DMA instruction writes 64 lanes x 16 B = 8 consecutive 128-B tile rows; the XOR swizzle of the fragment reads
//     is applied on the SOURCE side (lane l fetches global chunk (l & 7) ^ key(row)), the LDS image stays lane-linear;
//   * halo / out-of-range lanes fetch from a 128-B page of zeros instead of being masked;
//   * per k-step: issue the next step's 8 DMA instructions, s_waitcnt vmcnt(8) (this step has landed), barrier,
//     64 MFMAs per wave from the current stage, barrier.  No ordinary global loads in the loop.
// Fragment layout, weight-row permutation and the epilogues are those of igemm_kernel (MI = 8 row tiles per wave).
// MI = 7: 224-row tiles (each wave 112 x 64).  401 408 pixels (256 channels @ 14^2 at 2048 images) are 1568 tiles of 256 rows =
// 6.125 rounds of the 256 CUs, but exactly 7 rounds of 224-row tiles: no ragged round, no second launch (launch_igemm256).  The
// A region of a stage keeps 256 rows; rows 224.. fetch the zero page (waves 4-7 skip that DMA altogether).
// FP8: both operands are e4m3 bytes (forward; data gradient without a second reduction segment).  A tile row is still 128 B, now 128 k-elements, so the LDS image, the DMA map and
// the swizzle do not change; a lane's two 16-B fragment chunks (g and g + 4) together are ONE operand of
// v_mfma_scale_f32_16x16x128_f8f6f4 (twice the bf16 rate, K = 128 per instruction): 32 instead of 64 matrix instructions per k-step
// for twice the reduction length -- and half the operand bytes per FLOP, which is what bounds this kernel's loop.
template <bool DGRAD, int MI = 8, bool FP8 = false>
__global__ __launch_bounds__(512, 1) void igemm256_kernel(IgemmArgs p) {
  typedef bf16_t T;                                                             // stored results (and bf16 operands)
  typedef typename std::conditional<FP8, unsigned char, bf16_t>::type IT;     // operand element
  constexpr int KE = FP8 ? 128 : 64, VE = 8, IVE = FP8 ? 16 : 8, BM = MI * 32, BN = 256, NI = 4, WR = MI * 16;  // WR: rows per wave
  constexpr int A_BYTES = 256 * 128, B_BYTES = BN * 128, NSA = 3, NSB = 2, B_BASE = NSA * A_BYTES;
  __shared__ __attribute__((aligned(16))) char smem[NSA * A_BYTES + NSB * B_BYTES];  // 160 KB: the CU's whole LDS
  // LDS-DMA as inline asm: hipcc's waitcnt pass makes every ds_read wait for ALL outstanding builtin LDS-DMAs
  // (s_waitcnt vmcnt(0) right after the barrier), which would drain the prefetch; the asm form is invisible to it and
  // the loop counts vmcnt by hand.  m0 = LDS byte address of the wave's 1-KB destination.
  auto dma16 = [](const void* src, unsigned lds_addr) __attribute__((always_inline)) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_addr), "v"(src));
  };
  const unsigned smem_addr = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int li = lane & 15, g = lane >> 4;
  const int wm = wave >> 2, wn = wave & 3;
  int logical = xcd_remap(blockIdx.x, gridDim.x);
  const int n_tile = logical % p.n_tiles;
  logical /= p.n_tiles;
  const int cls = logical % p.classes;
  const int m_tile = logical / p.classes;
  const int ph = cls >> 1, pw = cls & 1;
  const unsigned m0 = (unsigned)m_tile * (unsigned)BM;
  const int n0 = n_tile * BN;
  const bool par = DGRAD && p.classes == 4;

  auto decode = [&](unsigned m) __attribute__((always_inline)) -> Pixel {
    Pixel r;
    r.ok = m < (unsigned)p.Mg;
    const unsigned mm = r.ok ? m : 0u;
    const unsigned img = fdiv(mm, p.div_hw);
    const unsigned rem = mm - img * p.div_hw.d;
    const unsigned hq = fdiv(rem, p.div_w);
    const unsigned wq = rem - hq * p.div_w.d;
    r.img = (int)img;
    if (par) {
      r.hd = 2 * (int)hq + ph;
      r.wd = 2 * (int)wq + pw;
      r.ok = r.ok && r.hd < p.Hd && r.wd < p.Wd;
    } else {
      r.hd = (int)hq;
      r.wd = (int)wq;
    }
    return r;
  };

  const int r0 = par ? ((ph + p.pad) & 1) : 0, s0 = par ? ((pw + p.pad) & 1) : 0;
  const int rstep = par ? 2 : 1;
  const int ntr = par ? (p.R - r0 + 1) / 2 : p.R;
  const int nts = par ? (p.S - s0 + 1) / 2 : p.S;
  const int csteps = p.Ca / KE;
  const int nk = (ntr > 0 && nts > 0) ? ntr * nts * csteps : 0;
  constexpr int dh = DGRAD ? -1 : 1;
  if (DGRAD && nk == 0 && p.accumulate == 1) return;

  // ---- DMA map: instruction i of wave w fills tile rows i*64 + w*8 .. +8; lane l = row (l >> 3), LDS slot l & 7 ----
  const int slot = lane & 7;
  const int lrow = wave * 8 + (lane >> 3);  // 0..63; rows lrow + 64 i share their swizzle keys
  auto key_b = [](int row) __attribute__((always_inline)) -> int { return ((row >> 1) & 1) | (((row >> 3) & 3) << 1); };
  auto chan_of = [](int ni, int m) __attribute__((always_inline)) -> int {
    return (ni >> 1) * 32 + (m >> 2) * 8 + (ni & 1) * 4 + (m & 3);
  };
  const int chunk_a = slot ^ ((lrow >> 1) & 7);
  const int chunk_b = slot ^ key_b(lrow);
  const IT* __restrict__ asrc = reinterpret_cast<const IT*>(p.a);
  const IT* __restrict__ wsrc = reinterpret_cast<const IT*>(p.w);
  const char* zsrc = reinterpret_cast<const char*>(g_zero_page) + slot * 16;
  const IT* pa0; const IT* pa1; const IT* pa2; const IT* pa3;
  int h00, h01, h02, h03, w00, w01, w02, w03;
  auto init_row = [&](int i, const IT*& pa, int& h0, int& w0) __attribute__((always_inline)) {
    const bool in_tile = lrow + 64 * i < BM;  // MI = 7: rows 224.. of the A region belong to no wave
    const Pixel px = decode(in_tile ? m0 + lrow + 64 * i : 0xffffffffu);
    if (DGRAD) {
      h0 = par ? (px.hd + p.pad - r0) >> 1 : px.hd + p.pad;
      w0 = par ? (px.wd + p.pad - s0) >> 1 : px.wd + p.pad;
    } else {
      h0 = px.hd * p.stride - p.pad;
      w0 = px.wd * p.stride - p.pad;
    }
    pa = asrc + ((long long)px.img * p.Hs * p.Ws + (long long)h0 * p.Ws + w0) * p.lda + chunk_a * IVE;
    if (!px.ok) h0 = -(1 << 20);
  };
  init_row(0, pa0, h00, w00);
  init_row(1, pa1, h01, w01);
  init_row(2, pa2, h02, w02);
  init_row(3, pa3, h03, w03);
  const long long wrow = (long long)p.R * p.S * p.lda;
  const IT* pb0 = wsrc + (long long)(n0 + lrow) * wrow + chunk_b * IVE;
  long long wrow64 = 64 * wrow;
  const int cs1 = p.lda / KE;  // k-steps of the first K segment (== csteps without a second one)
  const int dma_row0 = wave * 8 * 128;  // + i * 64 * 128: wave-uniform LDS offset of this wave's 8 rows

  // DMA state (wave-uniform).  TWO walkers over the same k order: the activation walker runs two k-steps ahead of the MFMAs (three
  // 32-KB stages), the weight walker one (two stages) -- 160 KB, the whole LDS of the CU.  k order: taps INNERMOST (a 64-channel
  // slice of the activation rows visits its taps back to back: the re-reads of a slice are at most nts * ntr k-steps apart and stay
  // in L2 whatever cin is; tap-major order measured 5-10 % slower from cin = 512 on, scripts/tile_overhead.py).
  struct Walk { int cs, tr, ts; };
  Walk wa = {0, 0, 0}, wb = {0, 0, 0};
  int n_hoff = 0, n_woff = 0, n_aoff = 0, n_boff = 0;
  unsigned n_dA = 0, n_dB = 0;
  bool a_live = true, b_live = true;  // false past the last k-step: the (branch-free) DMAs then fetch the zero page into an idle stage
  auto advance = [&](Walk& w) __attribute__((always_inline)) {
    if (++w.ts == nts) {
      w.ts = 0;
      if (++w.tr == ntr) {
        w.tr = 0;
        ++w.cs;
      }
    }
  };
  auto next_a = [&](int stage) __attribute__((always_inline)) {
    if constexpr (DGRAD && !FP8) if (p.a2 != nullptr && wa.cs == cs1) {  // (the e4m3 data gradient has no second segment: launch_igemm256_fp8_dgrad refuses one)
      // second K segment (1x1 / stride 1: pixel index == m): re-base the row pointers so that the same cs * KE offsets walk a2;
      // dead rows keep their h0 = -2^20
      const IT* a2 = reinterpret_cast<const IT*>(p.a2);
      const long long back = -(long long)cs1 * KE;
      pa0 = a2 + (long long)(m0 + lrow) * p.Ca2 + chunk_a * VE + back;
      pa1 = a2 + (long long)(m0 + lrow + 64) * p.Ca2 + chunk_a * VE + back;
      pa2 = a2 + (long long)(m0 + lrow + 128) * p.Ca2 + chunk_a * VE + back;
      pa3 = a2 + (long long)(m0 + lrow + 192) * p.Ca2 + chunk_a * VE + back;
    }
    n_hoff = dh * wa.tr;
    n_woff = dh * wa.ts;
    n_aoff = (n_hoff * p.Ws + n_woff) * p.lda + wa.cs * KE;
    n_dA = smem_addr + stage * A_BYTES + dma_row0;
    advance(wa);
  };
  auto next_b = [&](int stage) __attribute__((always_inline)) {
    if constexpr (DGRAD && !FP8) if (p.a2 != nullptr && wb.cs == cs1) {  // second K segment: w2 [Ng][Ca2]
      pb0 = reinterpret_cast<const IT*>(p.w2) + (long long)(n0 + lrow) * p.Ca2 + chunk_b * VE - (long long)cs1 * KE;
      wrow64 = 64ll * p.Ca2;
    }
    n_boff = ((r0 + rstep * wb.tr) * p.S + (s0 + rstep * wb.ts)) * p.lda + wb.cs * KE;
    n_dB = smem_addr + B_BASE + stage * B_BYTES + dma_row0;
    advance(wb);
  };
  // (The round-5 ablation and cycle-stamp builds of this kernel live on the branch r05-instrumented-kernels; what they measured: DESIGN 3b.)
  // Measured and removed again in round 5: non-temporal output stores (neutral), a start skew between the CUs of the first round (neutral).
  // part i (0..3) of a k-step's activation DMAs: tile rows lrow + 64 i (MI = 7: rows 224.. fetch the zero page -- every wave issues the
  // same number of DMAs, the counted waits below depend on it)
  auto dma_a_part = [&](int i) __attribute__((always_inline)) {
    auto dma_a = [&](const IT* pa, int h0, int w0) __attribute__((always_inline)) {
      const bool ok = a_live && (unsigned)(h0 + n_hoff) < (unsigned)p.Hs && (unsigned)(w0 + n_woff) < (unsigned)p.Ws;
      const char* src = ok ? reinterpret_cast<const char*>(pa + n_aoff) : zsrc;
      dma16(src, n_dA + i * 64 * 128);
    };
    if (i == 0) dma_a(pa0, h00, w00);
    else if (i == 1) dma_a(pa1, h01, w01);
    else if (i == 2) dma_a(pa2, h02, w02);
    else dma_a(pa3, h03, w03);
  };
  auto dma_b_part = [&](int i) __attribute__((always_inline)) {
    const char* src = b_live ? reinterpret_cast<const char*>(pb0 + i * wrow64 + n_boff) : zsrc;
    dma16(src, n_dB + i * 64 * 128);
  };
  // the 8 DMA instructions a wave issues per k-step, in queue order: 0-3 = weights of step kt + 1, 4-7 = activations of step kt + 2
  auto dma_part = [&](int q) __attribute__((always_inline)) {
    if (q < 4) dma_b_part(q);
    else dma_a_part(q - 4);
  };

  f32x4 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int fkey = (li >> 1) & 7;
  const int fo0 = (g ^ fkey) * 16;
  const int fa_base = (wm * WR + li) * 128;
  const int rowb0 = wn * 64 + chan_of(0, li);
  const int fbo = rowb0 * 128 + ((g ^ key_b(rowb0)) * 16);

  // ---- main loop: ONE barrier per k-step and a COUNTED wait -- the vector-memory queue never drains.  The CU's L1 holds a bounded
  // number of L2 requests in flight (measured: ~40 lines at ~270 cycles of L2 latency = 19 B/clk, against the 30 B/clk the MFMAs could
  // consume; with every DMA served from L1 the same loop runs at the MFMA rate -- profiles/r05_igemm256_ablation.txt), so what the loop
  // must do is keep that queue non-empty ALL the time: one DMA instruction after every MFMA group, weights of step kt + 1 first (they
  // have the rest of the step to land), then the activations of step kt + 2 (a whole further step).  In-order retirement: at the top
  // of step kt the queue ends with [W(kt) x 4, A(kt + 1) x 4]; vmcnt(4) leaves exactly A(kt + 1) in flight.  The barrier that follows
  // orders (a) every wave's landed DMAs of step kt before anybody's fragment reads and (b) everybody's reads of step kt - 1 before
  // the DMAs into its stages (A stage (kt + 2) % 3 == (kt - 1) % 3, W stage (kt + 1) & 1).  The fragments of MFMA group grp + 1 are
  // read before the MFMAs of group grp are issued.  (Measured alternatives, all slower: two 64-KB stages with the DMAs in the first half
  // of the step and vmcnt(0) -- the round 1-4 form; two barriers per step; ping-pong wave groups: s_barrier costs ~250 cycles.)
  if (nk > 0) {
    next_a(0);
#pragma unroll
    for (int i = 0; i < 4; ++i) dma_a_part(i);
    next_b(0);
#pragma unroll
    for (int i = 0; i < 4; ++i) dma_b_part(i);
    a_live = 1 < nk;
    next_a(1);
#pragma unroll
    for (int i = 0; i < 4; ++i) dma_a_part(i);
  }
  int sa = 0;  // kt % 3
  {
  for (int kt = 0; kt < nk; ++kt) {
    asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    b_live = kt + 1 < nk;
    next_b((kt + 1) & 1);
    a_live = kt + 2 < nk;
    next_a(sa == 0 ? 2 : sa - 1);  // (kt + 2) % 3
    const char* st = smem + sa * A_BYTES;                      // activation stage of this step
    const char* stb = smem + B_BASE + (kt & 1) * B_BYTES;      // weight stage
    sa = sa == 2 ? 0 : sa + 1;
    if constexpr (FP8) {
      // one scaled MFMA per 16 x 16 tile pair and k-step: operand = the lane's chunks g (k 0..63 half) and g + 4 of its tile row
      typedef __attribute__((ext_vector_type(8))) int i32x8;
      uint4 fbl[NI], fbh[NI], fal[2][2], fah[2][2];
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        fbl[ni] = *reinterpret_cast<const uint4*>(stb + fbo + (chan_of(ni, 0) - chan_of(0, 0)) * 128);
        fbh[ni] = *reinterpret_cast<const uint4*>(stb + (fbo ^ 64) + (chan_of(ni, 0) - chan_of(0, 0)) * 128);
      }
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2) {
        fal[0][h2] = *reinterpret_cast<const uint4*>(st + fa_base + h2 * 16 * 128 + fo0);
        fah[0][h2] = *reinterpret_cast<const uint4*>(st + fa_base + h2 * 16 * 128 + (fo0 ^ 64));
      }
      auto pack8 = [](const uint4& lo, const uint4& hi) __attribute__((always_inline)) -> i32x8 {
        i32x8 r = {(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y, (int)hi.z, (int)hi.w};
        return r;
      };
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (q < 3) {
#pragma unroll
          for (int h2 = 0; h2 < 2; ++h2) {
            if (2 * (q + 1) + h2 < MI) {
              fal[(q + 1) & 1][h2] = *reinterpret_cast<const uint4*>(st + fa_base + (2 * (q + 1) + h2) * 16 * 128 + fo0);
              fah[(q + 1) & 1][h2] = *reinterpret_cast<const uint4*>(st + fa_base + (2 * (q + 1) + h2) * 16 * 128 + (fo0 ^ 64));
            }
          }
        }
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
          if (2 * q + h2 < MI) {
            const i32x8 xa = pack8(fal[q & 1][h2], fah[q & 1][h2]);
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
              acc[2 * q + h2][ni] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(pack8(fbl[ni], fbh[ni]), xa, acc[2 * q + h2][ni], 0, 0, 0,
                                                                                     0x7f7f7f7f, 0, 0x7f7f7f7f);
          }
        }
        dma_part(2 * q);
        dma_part(2 * q + 1);
      }
    } else {
    uint4 fb[2][NI], fa[2][2];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) fb[0][ni] = *reinterpret_cast<const uint4*>(stb + fbo + (chan_of(ni, 0) - chan_of(0, 0)) * 128);
    fa[0][0] = *reinterpret_cast<const uint4*>(st + fa_base + fo0);
    fa[0][1] = *reinterpret_cast<const uint4*>(st + fa_base + 16 * 128 + fo0);
#pragma unroll
    for (int grp = 0; grp < 8; ++grp) {
      const int kk = grp >> 2, q = grp & 3;
      // (Measured in round 5 and not kept: the younger wave of each SIMD raised in issue priority in every other MFMA group -- +3-8 % per launch
      // in isolation, nothing in the step, which runs at the socket's power limit: profiles/r05_power_wall.md, branch r05-instrumented-kernels.)
      if (grp < 7) {
        const int nkk = (grp + 1) >> 2, nq = (grp + 1) & 3;
        const int fo = nkk == 0 ? fo0 : (fo0 ^ 64);
        if (grp == 3) {
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            fb[1][ni] = *reinterpret_cast<const uint4*>(stb + (fbo ^ 64) + (chan_of(ni, 0) - chan_of(0, 0)) * 128);
        }
        {
        fa[(grp + 1) & 1][0] = *reinterpret_cast<const uint4*>(st + fa_base + (2 * nq) * 16 * 128 + fo);
        if (2 * nq + 1 < MI) fa[(grp + 1) & 1][1] = *reinterpret_cast<const uint4*>(st + fa_base + (2 * nq + 1) * 16 * 128 + fo);
        }
      }
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) acc[2 * q][ni] = Mma<T>::run(fb[kk][ni], fa[grp & 1][0], acc[2 * q][ni]);
      if (2 * q + 1 < MI) {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[2 * q + 1][ni] = Mma<T>::run(fb[kk][ni], fa[grp & 1][1], acc[2 * q + 1][ni]);
      }
      dma_part(grp);  // one DMA instruction per MFMA group: the L1's request queue stays fed through the whole step
    }
      }
  }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");  // LDS is reused by the epilogues
  if constexpr (FP8) {  // per-tensor scales: one multiply per accumulator, before the statistics and the stores
    const float descale = p.x_state[1] * p.w_state[1];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) acc[mi][ni] *= descale;
  }

  // ---- fused BatchNorm partial statistics (forward): lane holds pixel wm*128 + mi*16 + li, channels wn*64 + chan_of(ni, 4g + r)
  if (!DGRAD && p.bn_partial != nullptr) {
    float* red = reinterpret_cast<float*>(smem);  // [2 (wm)][2][BN]
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          const float v = acc[mi][ni][r];
          s1 += v;
          s2 += v * v;
        }
        s1 = row16_sum(s1);
        s2 = row16_sum(s2);
        if (li == 0) {
          const int c = wn * 64 + chan_of(ni, 4 * g + r);
          red[(wm * 2 + 0) * BN + c] = s1;
          red[(wm * 2 + 1) * BN + c] = s2;
        }
      }
    }
    __syncthreads();
    {
      const int which = tid / BN, c = tid - which * BN;  // 512 threads = 2 x BN
      const float v = red[(0 * 2 + which) * BN + c] + red[(1 * 2 + which) * BN + c];
      p.bn_partial[((long long)m_tile * 2 + which) * p.Ng + n0 + c] = v;
    }
  }

  // ---- epilogue (as igemm_kernel): 16-B vectors from registers, channels ch0 + j*32 .. +8 of pixel row mi ----
  {
    constexpr int NCH = 2;
    T* __restrict__ out = reinterpret_cast<T*>(p.out);
    const int ch0 = n0 + wn * 64 + g * VE;
    auto pixel_of = [&](int mi) __attribute__((always_inline)) -> long long {
      const unsigned mrow = m0 + wm * WR + mi * 16 + li;
      long long pix = (long long)mrow;
      bool ok = pix < p.Mg;
      if (par) {
        const Pixel px = decode(mrow);
        ok = px.ok;
        pix = ((long long)px.img * p.Hd + px.hd) * p.Wd + px.wd;
      }
      return ok ? pix : -1;
    };
    auto chunk_out = [&](int mi, int j, long long pix, unsigned keep = 0xffu, bool do_store = true) __attribute__((always_inline)) -> uint4 {
      uint4 v;
      f32x4 lo = acc[mi][2 * j], hi = acc[mi][2 * j + 1];
      if (DGRAD && p.bias != nullptr) {
        const int chb = ch0 + j * 32;
        const float4 b0 = *reinterpret_cast<const float4*>(p.bias + chb), b1 = *reinterpret_cast<const float4*>(p.bias + chb + 4);
        lo[0] += b0.x; lo[1] += b0.y; lo[2] += b0.z; lo[3] += b0.w;
        hi[0] += b1.x; hi[1] += b1.y; hi[2] += b1.z; hi[3] += b1.w;
      }
      v.x = pack_bf16x2(lo[0], lo[1]);
      v.y = pack_bf16x2(lo[2], lo[3]);
      v.z = pack_bf16x2(hi[0], hi[1]);
      v.w = pack_bf16x2(hi[2], hi[3]);
      const int ch = ch0 + j * 32;
      T* dst = out + pix * p.Ng + ch;
      if (p.accumulate == 2) {
        const uint4 o = *reinterpret_cast<const uint4*>(reinterpret_cast<const T*>(p.res_grad) + pix * p.Ng + ch);
        const unsigned bits = p.res_mask[pix * (p.Ng / VE) + ch / VE];
        const unsigned m0w = ((bits & 1u) ? 0x0000ffffu : 0u) | ((bits & 2u) ? 0xffff0000u : 0u);
        const unsigned m1w = ((bits & 4u) ? 0x0000ffffu : 0u) | ((bits & 8u) ? 0xffff0000u : 0u);
        const unsigned m2w = ((bits & 16u) ? 0x0000ffffu : 0u) | ((bits & 32u) ? 0xffff0000u : 0u);
        const unsigned m3w = ((bits & 64u) ? 0x0000ffffu : 0u) | ((bits & 128u) ? 0xffff0000u : 0u);
        v.x = add_bf16x2(v.x, o.x & m0w);
        v.y = add_bf16x2(v.y, o.y & m1w);
        v.z = add_bf16x2(v.z, o.z & m2w);
        v.w = add_bf16x2(v.w, o.w & m3w);
      } else if (p.accumulate) {
        const uint4 o = *reinterpret_cast<const uint4*>(dst);
        v.x = add_bf16x2(v.x, o.x);
        v.y = add_bf16x2(v.y, o.y);
        v.z = add_bf16x2(v.z, o.z);
        v.w = add_bf16x2(v.w, o.w);
      }
      if (DGRAD && p.sub != nullptr) {  // even pixels: + the shortcut's dense gradient row (odd ones: the zero page, no branch)
        const unsigned pu = (unsigned)pix;
        const unsigned img = fdiv(pu, p.div_hw);
        const unsigned rem = pu - img * p.div_hw.d;
        const unsigned hh = fdiv(rem, p.div_w);
        const unsigned ww = rem - hh * p.div_w.d;
        const unsigned srow = (img * (unsigned)(p.Hd >> 1) + (hh >> 1)) * (unsigned)(p.Wd >> 1) + (ww >> 1);
        const void* q = ((hh | ww) & 1u) == 0 ? (const void*)(reinterpret_cast<const T*>(p.sub) + (unsigned long long)srow * p.Ng + ch)
                                               : (const void*)g_zero_page;
        const uint4 o = *reinterpret_cast<const uint4*>(q);
        v.x = add_bf16x2(v.x, o.x);
        v.y = add_bf16x2(v.y, o.y);
        v.z = add_bf16x2(v.z, o.z);
        v.w = add_bf16x2(v.w, o.w);
      }
      if (DGRAD && p.fmode == 4) {  // the stored gradient is the masked one
        v.x &= ((keep & 1u) ? 0x0000ffffu : 0u) | ((keep & 2u) ? 0xffff0000u : 0u);
        v.y &= ((keep & 4u) ? 0x0000ffffu : 0u) | ((keep & 8u) ? 0xffff0000u : 0u);
        v.z &= ((keep & 16u) ? 0x0000ffffu : 0u) | ((keep & 32u) ? 0xffff0000u : 0u);
        v.w &= ((keep & 64u) ? 0x0000ffffu : 0u) | ((keep & 128u) ? 0xffff0000u : 0u);
      }
      if (do_store) *reinterpret_cast<uint4*>(dst) = v;
      return v;
    };
    if (!DGRAD && sizeof(T) == 2 && p.ep_scale != nullptr) {
      // BatchNorm (statistics known up front) + residual + ReLU applied to the fp32 accumulators; ReLU bit mask out
      const bf16_t* __restrict__ res = reinterpret_cast<const bf16_t*>(p.ep_res);
      bf16_t* __restrict__ outb = reinterpret_cast<bf16_t*>(p.out);
#pragma unroll
      for (int j = 0; j < NCH; ++j) {
        const int ch = ch0 + j * 32;
        float sc[8], sh[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          sc[e] = p.ep_scale[ch + e];
          sh[e] = p.ep_shift[ch + e];
        }
        // all residual rows of this 32-channel group are requested before the first one is used: one exposed global-load
        // round trip per group instead of one per 16-pixel row (the block is alone on its CU, nothing else hides them)
        uint4 rq[MI];
        if (res != nullptr) {
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) {
            const long long pix = pixel_of(mi);
            // (rows past the range read row 0, never used: a conditional load is waited for at the join, one round trip per row)
            // non-temporal (round 6, as in conv_1x1.hip): the residual on its last forward use and the block output below stream past the
            // caches -- forward class -0.35 ms per step in three same-box pairs (profiles/r06_cache_policy_ab.txt)
            rq[mi] = ld16<true>(res + (pix < 0 ? 0 : pix) * p.Ng + ch);
          }
        }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          const long long pix = pixel_of(mi);
          if (pix < 0) continue;
          const f32x4 lo = acc[mi][(2 * j) % NI], hi = acc[mi][(2 * j + 1) % NI];
          float o[8];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            o[e] = lo[e] * sc[e] + sh[e];
            o[4 + e] = hi[e] * sc[4 + e] + sh[4 + e];
          }
          if (res != nullptr) {
            const unsigned w4[4] = {rq[mi].x, rq[mi].y, rq[mi].z, rq[mi].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              o[2 * e] += h16_lo(w4[e]);
              o[2 * e + 1] += h16_hi(w4[e]);
            }
          }
          if (p.ep_relu) {
            unsigned bits = 0;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              bits |= (o[e] > 0.f ? 1u : 0u) << e;
              o[e] = o[e] > 0.f ? o[e] : 0.f;
            }
            if (p.ep_mask != nullptr) p.ep_mask[pix * (p.Ng / 8) + ch / 8] = (unsigned char)bits;
          }
          Vec16<bf16_t>::store<true>(outb + pix * p.Ng + ch, o);
        }
      }
    } else if (!DGRAD || p.fpartial == nullptr) {
      // plain forward store: lanes li / li ^ 1 trade one packed 16-B chunk (DPP quad_perm) so that every store instruction writes whole 128-B lines --
      // 8 rows x 128 B instead of 16 rows x 64 B.  The CU's store path digests the whole-line form in 60 % of the cycles (scripts/probes/store_pattern.hip,
      // profiles/r05_igemm256_tile_stamps.md): -1 us per tile, bit-identical.  (The fused-sums data-gradient epilogue below does the same since round 6; the BatchNorm + residual one and the plain data-gradient stores keep the 64-B form.)
      if (!DGRAD && !par && p.accumulate == 0) {
        auto swap1 = [](unsigned v) __attribute__((always_inline)) -> unsigned {
          return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, false);  // quad_perm [1, 0, 3, 2]
        };
        T* __restrict__ outp = reinterpret_cast<T*>(p.out);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          const f32x4 c0 = acc[mi][0], c1 = acc[mi][1], c2 = acc[mi][2], c3 = acc[mi][3];
          const unsigned a0 = pack_bf16x2(c0[0], c0[1]), a1 = pack_bf16x2(c0[2], c0[3]), a2 = pack_bf16x2(c1[0], c1[1]), a3 = pack_bf16x2(c1[2], c1[3]);
          const unsigned b0 = pack_bf16x2(c2[0], c2[1]), b1 = pack_bf16x2(c2[2], c2[3]), b2 = pack_bf16x2(c3[0], c3[1]), b3 = pack_bf16x2(c3[2], c3[3]);
          const bool odd = (li & 1) != 0;
          // even lanes send their second channel group (b) and keep a; odd lanes send a and keep b
          const unsigned r0 = swap1(odd ? a0 : b0), r1 = swap1(odd ? a1 : b1), r2 = swap1(odd ? a2 : b2), r3 = swap1(odd ? a3 : b3);
          const long long row = (long long)m0 + wm * WR + mi * 16 + li;      // own row; the partner's is row ^ 1
          const long long rowA = odd ? row - 1 : row, rowB = odd ? row : row + 1;
          const int chx = ch0 + (odd ? 32 : 0);
          const uint4 vA = odd ? make_uint4(r0, r1, r2, r3) : make_uint4(a0, a1, a2, a3);
          const uint4 vB = odd ? make_uint4(b0, b1, b2, b3) : make_uint4(r0, r1, r2, r3);
          if (rowA < p.Mg) *reinterpret_cast<uint4*>(outp + rowA * p.Ng + chx) = vA;
          if (rowB < p.Mg) *reinterpret_cast<uint4*>(outp + rowB * p.Ng + chx) = vB;
        }
      } else {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          const long long pix = pixel_of(mi);
          if (pix < 0) continue;
#pragma unroll
          for (int j = 0; j < NCH; ++j) {
            unsigned keep = 0xffu;
            if (DGRAD && p.fmode == 4) keep = p.fmask[pix * (p.Ng / VE) + (ch0 + j * 32) / VE];  // masked store, no sums
            chunk_out(mi, j, pix, keep);
          }
        }
      }
    } else {
      // fused BatchNorm-backward partial sums of the previous unit (arithmetic and summation order of igemm_kernel's; bit-identical to the round 1-5 form).
      // Round 6: (a) the y rows come through LDS -- each wave's 2 MI LDS-DMA instructions go out back to back into lane-private 16-B slots (no swizzle,
      // no barrier: a lane reads what its own DMA lane wrote).  The register form asked for a row's first 64-B half, worked through MI row groups and then
      // asked for the second half: by then the 128-B line had left L2 and HBM delivered it twice (666 MB read per 256-channel 3x3 launch for 472 MB of
      // operands, profiles/r06_i256_dgrad_traffic.txt; 3.1 GB per step).  (b) Row-outer order, both channel halves of a row in hand: lanes li / li ^ 1 trade
      // one packed chunk and every store instruction writes whole 128-B lines, as the plain forward epilogue above.  The 32 registers of the y rows pay for
      // the second half's coefficients and sums.  Data-gradient class -0.13 ms per step in 5 of 5 same-box pairs (profiles/r06_cache_policy_ab.txt).
      float* red = reinterpret_cast<float*>(smem + 8 * MI * NCH * 1024);  // [2 (wm)][2][BN], behind the y slots
      const T* __restrict__ fy = reinterpret_cast<const T*>(p.fy);
      const char* yslot = smem + (wave * MI * NCH) * 1024 + lane * 16;
      if (p.fmode != 4) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          const long long pix = pixel_of(mi);
          const T* src = fy + (pix < 0 ? 0 : pix) * p.Ng + ch0;
#pragma unroll
          for (int j = 0; j < NCH; ++j) dma16(src + j * 32, smem_addr + ((wave * MI + mi) * NCH + j) * 1024);
        }
      }
      float sc0[VE], sh0[VE], sc1[VE], sh1[VE], s10[VE], s20[VE], s11[VE], s21[VE];
#pragma unroll
      for (int e = 0; e < VE; ++e) sc0[e] = sh0[e] = sc1[e] = sh1[e] = s10[e] = s20[e] = s11[e] = s21[e] = 0.f;
      if (p.fmode == 2) {
#pragma unroll
        for (int e = 0; e < VE; e += 4) {
          const float4 a4 = *reinterpret_cast<const float4*>(p.fscale + ch0 + e), b4 = *reinterpret_cast<const float4*>(p.fshift + ch0 + e);
          const float4 c4 = *reinterpret_cast<const float4*>(p.fscale + ch0 + 32 + e), d4 = *reinterpret_cast<const float4*>(p.fshift + ch0 + 32 + e);
          sc0[e] = a4.x; sc0[e + 1] = a4.y; sc0[e + 2] = a4.z; sc0[e + 3] = a4.w;
          sh0[e] = b4.x; sh0[e + 1] = b4.y; sh0[e + 2] = b4.z; sh0[e + 3] = b4.w;
          sc1[e] = c4.x; sc1[e + 1] = c4.y; sc1[e + 2] = c4.z; sc1[e + 3] = c4.w;
          sh1[e] = d4.x; sh1[e + 1] = d4.y; sh1[e + 2] = d4.z; sh1[e + 3] = d4.w;
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      auto swap1 = [](unsigned v) __attribute__((always_inline)) -> unsigned {
        return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, false);  // quad_perm [1, 0, 3, 2]
      };
      auto sums = [&](const uint4& v, const uint4& y, unsigned bits, float (&sc)[VE], float (&sh)[VE], float (&s1)[VE], float (&s2)[VE]) __attribute__((always_inline)) {
        const unsigned w4[4] = {v.x, v.y, v.z, v.w};
        const unsigned y4[4] = {y.x, y.y, y.z, y.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int e = 2 * i + h;
            const float gq = h == 0 ? h16_lo(w4[i]) : h16_hi(w4[i]);
            const float yy = h == 0 ? h16_lo(y4[i]) : h16_hi(y4[i]);
            bool on = true;
            if (p.fmode == 2) on = yy * sc[e] + sh[e] > 0.f;
            else if (p.fmode == 3) on = (bits >> e) & 1u;
            const float gv = on ? gq : 0.f;
            s1[e] += gv;
            s2[e] += gv * yy;
          }
        }
      };
      const bool odd = (li & 1) != 0;
      const int chx = ch0 + (odd ? 32 : 0);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const long long pix = pixel_of(mi);
        const long long ppix = (long long)(((unsigned long long)swap1((unsigned)((unsigned long long)pix >> 32)) << 32) | swap1((unsigned)pix));
        uint4 va = make_uint4(0, 0, 0, 0), vb = make_uint4(0, 0, 0, 0);
        if (pix >= 0) {
          const unsigned b0 = p.fmode >= 3 ? (unsigned)p.fmask[pix * (p.Ng / VE) + ch0 / VE] : 0xffu;
          const unsigned b1 = p.fmode >= 3 ? (unsigned)p.fmask[pix * (p.Ng / VE) + (ch0 + 32) / VE] : 0xffu;
          const uint4 y0 = p.fmode != 4 ? *reinterpret_cast<const uint4*>(yslot + (mi * NCH + 0) * 1024) : make_uint4(0, 0, 0, 0);
          const uint4 y1 = p.fmode != 4 ? *reinterpret_cast<const uint4*>(yslot + (mi * NCH + 1) * 1024) : make_uint4(0, 0, 0, 0);
          va = chunk_out(mi, 0, pix, b0, false);
          vb = chunk_out(mi, 1, pix, b1, false);
          sums(va, y0, b0, sc0, sh0, s10, s20);
          sums(vb, y1, b1, sc1, sh1, s11, s21);
        }
        // even lanes send their second channel group (vb) and keep va; odd lanes send va and keep vb
        const unsigned r0 = swap1(odd ? va.x : vb.x), r1 = swap1(odd ? va.y : vb.y), r2 = swap1(odd ? va.z : vb.z), r3 = swap1(odd ? va.w : vb.w);
        const long long pixA = odd ? ppix : pix, pixB = odd ? pix : ppix;
        const uint4 vA = odd ? make_uint4(r0, r1, r2, r3) : va;
        const uint4 vB = odd ? vb : make_uint4(r0, r1, r2, r3);
        if (pixA >= 0) *reinterpret_cast<uint4*>(out + pixA * p.Ng + chx) = vA;
        if (pixB >= 0) *reinterpret_cast<uint4*>(out + pixB * p.Ng + chx) = vB;
      }
#pragma unroll
      for (int j = 0; j < NCH; ++j) {
#pragma unroll
        for (int e = 0; e < VE; ++e) {
          const float t1 = row16_sum(j == 0 ? s10[e] : s11[e]), t2 = row16_sum(j == 0 ? s20[e] : s21[e]);
          if (li == 0) {
            const int c = wn * 64 + g * VE + j * 32 + e;
            red[(wm * 2 + 0) * BN + c] = t1;
            red[(wm * 2 + 1) * BN + c] = t2;
          }
        }
      }
      __syncthreads();
      {
        const int which = tid / BN, c = tid - which * BN;
        const float v = red[(0 * 2 + which) * BN + c] + red[(1 * 2 + which) * BN + c];
        p.fpartial[(((long long)m_tile * p.classes + cls) * 2 + which) * p.Ng + n0 + c] = v;
      }
    }
  }
}

// ======================================================================================================================
// 1x1 / stride-1 GEMM with exactly 128 destination channels and a long reduction (bf16): out [M][128] = a [M][K1] w^T (+ a2 [M][K2] w2^T).
// Replaces (reference): conv3's input gradient / conv1's forward of torchvision's stage-2 Bottlenecks (src/models/resnet_model.py:13-58;
// cuDNN there): 512 -> 128 @ 28^2 forward, and the folded 128 <- 512 (+ 128) data gradient with bias and the previous unit's BatchNorm-
// backward sums.  The register-staged 128 x 128 kernel runs these as a serial chain (loads, barrier, 32 MFMAs, barrier: 650 us in the
// step for 2.47 GB = 0.45 ms of HBM time + 0.26 ms of matrix time, which add up); the 256-wide LDS-DMA kernel needs 256 channels.
// Here: the 128 x 128 x 64 tile of igemm_kernel with the operand delivery of igemm256_kernel -- both operands global -> LDS by
// LDS-DMA (inline asm), activation rows two k-steps ahead (three 16-KB stages), weights one (two stages), counted vmcnt, ONE barrier per
// k-step, 4 waves as 2 (pixels) x 2 (channels) of 64 x 64 each (64 accumulator registers), and TWO blocks per CU: one block's epilogue (stores, the y rows of the fused sums) runs under the other's k-loop.
// LDS image, swizzles (applied on the DMA source side), weight-row permutation and fragment addressing are igemm256_kernel's.
template <bool DGRAD>
__global__ __launch_bounds__(256, 2) void gemm_n128_kernel(IgemmArgs p) {
  typedef bf16_t T;
  constexpr int BM = 128, BN = 128, MI = 4, NI = 4, VE = 8;
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, NSA = 3, B_BASE = NSA * A_BYTES;
  // THREE activation stages (that operand comes from HBM: two k-steps ahead) and two weight stages (L2-resident: one step ahead):
  // 80 KB per block, two blocks = the CU's 160 KB
  __shared__ __attribute__((aligned(16))) char smem[NSA * A_BYTES + 2 * B_BYTES];
  auto dma16 = [](const void* src, unsigned lds_addr) __attribute__((always_inline)) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_addr), "v"(src));
  };
  const unsigned smem_addr = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int li = lane & 15, g = lane >> 4;
  const int wm = wave >> 1, wn = wave & 1;
  const int m_tile = xcd_remap(blockIdx.x, gridDim.x);
  const long long m0 = (long long)m_tile * BM;

  // ---- DMA map: instruction i of wave w fills tile rows 32 i + 8 w .. + 8; lane l = row (l >> 3), 16-B slot l & 7 ----
  const int slot = lane & 7;
  const int lrow = wave * 8 + (lane >> 3);  // 0..31; rows lrow + 32 i share their swizzle keys
  auto key_b = [](int row) __attribute__((always_inline)) -> int { return ((row >> 1) & 1) | (((row >> 3) & 3) << 1); };
  auto chan_of = [](int ni, int m) __attribute__((always_inline)) -> int { return (ni >> 1) * 32 + (m >> 2) * 8 + (ni & 1) * 4 + (m & 3); };
  const int chunk_a = slot ^ ((lrow >> 1) & 7);
  const int chunk_b = slot ^ key_b(lrow);
  const char* zsrc = reinterpret_cast<const char*>(g_zero_page) + slot * 16;
  const int lda = p.lda;           // channels per row of a and of w (first segment)
  const int cs1 = lda / 64;        // its k-steps
  const int nk = p.Ca / 64;        // all k-steps (Ca = lda + Ca2)
  const T* __restrict__ a1 = reinterpret_cast<const T*>(p.a);
  const T* __restrict__ w1 = reinterpret_cast<const T*>(p.w);
  const T* __restrict__ a2 = reinterpret_cast<const T*>(p.a2);
  const T* __restrict__ w2 = reinterpret_cast<const T*>(p.w2);
  bool rok[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) rok[i] = m0 + lrow + 32 * i < p.Mg;
  // past the end: the zero page (the counted waits below see four instructions per operand and step)
  auto issue_a = [&](int kt) __attribute__((always_inline)) {
    const bool live = kt < nk, seg2 = kt >= cs1;
    const T* ab = seg2 ? a2 : a1;
    const long long ld = seg2 ? p.Ca2 : lda;
    const int k0 = (seg2 ? kt - cs1 : kt) * 64;
    const unsigned dst = smem_addr + (kt % NSA) * A_BYTES + wave * 8 * 128;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const char* src = live && rok[i] ? reinterpret_cast<const char*>(ab + (m0 + lrow + 32 * i) * ld + k0 + chunk_a * VE) : zsrc;
      dma16(src, dst + i * 32 * 128);
    }
  };
  auto issue_b = [&](int kt) __attribute__((always_inline)) {
    const bool live = kt < nk, seg2 = kt >= cs1;
    const T* wb = seg2 ? w2 : w1;
    const long long ld = seg2 ? p.Ca2 : lda;
    const int k0 = (seg2 ? kt - cs1 : kt) * 64;
    const unsigned dst = smem_addr + B_BASE + (kt & 1) * B_BYTES + wave * 8 * 128;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      dma16(live ? reinterpret_cast<const char*>(wb + (long long)(lrow + 32 * i) * ld + k0 + chunk_b * VE) : zsrc, dst + i * 32 * 128);
  };

  f32x4 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int fkey = (li >> 1) & 7;
  const int fo0 = (g ^ fkey) * 16;
  const int fa_base = (wm * 64 + li) * 128;
  const int rowb0 = wn * 64 + chan_of(0, li);
  const int fbo = B_BASE + rowb0 * 128 + ((g ^ key_b(rowb0)) * 16);

  // in-order queue: A0, B0, A1 | step kt issues B(kt+1) then A(kt+2): at the top of step kt everything but A(kt+1) (4 instructions) has landed
  issue_a(0);
  issue_b(0);
  issue_a(1);
  for (int kt = 0; kt < nk; ++kt) {
    asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");  // ... and every wave is past step kt - 1's stages
    issue_b(kt + 1);   // weight stage (kt + 1) & 1 = the one step kt - 1 read
    issue_a(kt + 2);   // activation stage (kt + 2) % 3 = the one step kt - 1 read
    const char* sta = smem + (kt % NSA) * A_BYTES;
    const char* stb = smem + (kt & 1) * B_BYTES;
    uint4 fb[2][NI], fa[2][2];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      fb[0][ni] = *reinterpret_cast<const uint4*>(stb + fbo + (chan_of(ni, 0) - chan_of(0, 0)) * 128);
      fb[1][ni] = *reinterpret_cast<const uint4*>(stb + (fbo ^ 64) + (chan_of(ni, 0) - chan_of(0, 0)) * 128);
    }
    fa[0][0] = *reinterpret_cast<const uint4*>(sta + fa_base + fo0);
    fa[0][1] = *reinterpret_cast<const uint4*>(sta + fa_base + (fo0 ^ 64));
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      if (mi + 1 < MI) {  // the next row group's fragments are read while this one's MFMAs run
        fa[(mi + 1) & 1][0] = *reinterpret_cast<const uint4*>(sta + fa_base + (mi + 1) * 16 * 128 + fo0);
        fa[(mi + 1) & 1][1] = *reinterpret_cast<const uint4*>(sta + fa_base + (mi + 1) * 16 * 128 + (fo0 ^ 64));
      }
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = sh_mfma16(fb[0][ni], fa[mi & 1][0], acc[mi][ni]);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = sh_mfma16(fb[1][ni], fa[mi & 1][1], acc[mi][ni]);
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");  // LDS is reused by the reductions below

  // ---- epilogue: lane holds pixel m0 + wm*64 + mi*16 + li, channels wn*64 + g*8 + 32 j .. + 7 (j = 0, 1) ----
  T* __restrict__ out = reinterpret_cast<T*>(p.out);
  const int ch0 = wn * 64 + g * VE;
  float* red = reinterpret_cast<float*>(smem);  // [2 (wm)][2][BN]
  const bool sums = DGRAD ? p.fpartial != nullptr : p.bn_partial != nullptr;
  {
    // row-outer (round 6, as igemm256_kernel's fused-sums epilogue): both 64-B halves of every y line are requested together, and lanes li / li ^ 1 trade
    // one packed chunk so that every store instruction writes whole 128-B lines.  Arithmetic and summation order are unchanged (bit-identical).  The
    // channel-half-outer form wrote a row's two halves one y round trip apart: 561 MB reached HBM for the 411 MB of the stage-2 folded data gradient
    // (now 414), 670 -> 630 us per launch in isolation (scripts/i256_dgrad_traffic.py n128_fold, profiles/r06_i256_dgrad_traffic.txt).
    float s1[2][VE], s2[2][VE], sc[2][VE], sh[2][VE], bb[2][VE];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < VE; ++e) s1[j][e] = s2[j][e] = sc[j][e] = bb[j][e] = 0.f, sh[j][e] = 1.f;  // (no ReLU: the gate y * 0 + 1 > 0 is open)
    uint4 yq[MI][2];
    if constexpr (DGRAD) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int ch = ch0 + j * 32;
        if (p.bias != nullptr) {
          const float4 b0 = *reinterpret_cast<const float4*>(p.bias + ch), b1 = *reinterpret_cast<const float4*>(p.bias + ch + 4);
          bb[j][0] = b0.x; bb[j][1] = b0.y; bb[j][2] = b0.z; bb[j][3] = b0.w; bb[j][4] = b1.x; bb[j][5] = b1.y; bb[j][6] = b1.z; bb[j][7] = b1.w;
        }
        if (sums && p.fmode == 2) {
#pragma unroll
          for (int e = 0; e < VE; e += 4) {
            const float4 a4 = *reinterpret_cast<const float4*>(p.fscale + ch + e), b4 = *reinterpret_cast<const float4*>(p.fshift + ch + e);
            sc[j][e] = a4.x; sc[j][e + 1] = a4.y; sc[j][e + 2] = a4.z; sc[j][e + 3] = a4.w;
            sh[j][e] = b4.x; sh[j][e + 1] = b4.y; sh[j][e + 2] = b4.z; sh[j][e + 3] = b4.w;
          }
        }
      }
      if (sums) {  // all y rows are requested before the first is used (rows past the range read row 0, branch-free)
        const T* __restrict__ fy = reinterpret_cast<const T*>(p.fy);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          const long long pix = m0 + wm * 64 + mi * 16 + li;
          const T* src = fy + (pix < p.Mg ? pix : 0) * BN + ch0;
          yq[mi][0] = *reinterpret_cast<const uint4*>(src);
          yq[mi][1] = *reinterpret_cast<const uint4*>(src + 32);
        }
      }
    }
    auto swap1 = [](unsigned v) __attribute__((always_inline)) -> unsigned {
      return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, false);  // quad_perm [1, 0, 3, 2]
    };
    const bool odd = (li & 1) != 0;
    const int chx = ch0 + (odd ? 32 : 0);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const long long pix = m0 + wm * 64 + mi * 16 + li;
      const bool ok = pix < p.Mg;
      uint4 oj[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const f32x4 lo = acc[mi][2 * j], hi = acc[mi][2 * j + 1];
        float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        if constexpr (DGRAD) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += bb[j][e];
        }
        uint4 o;
        o.x = pack_bf16x2(v[0], v[1]);
        o.y = pack_bf16x2(v[2], v[3]);
        o.z = pack_bf16x2(v[4], v[5]);
        o.w = pack_bf16x2(v[6], v[7]);
        oj[j] = o;
        if (sums) {
          if constexpr (DGRAD) {  // sums of the STORED gradient (as igemm_kernel): g = bf16(result) gated by the recomputed ReLU mask
            const unsigned w4[4] = {o.x, o.y, o.z, o.w};
            const unsigned y4[4] = {yq[mi][j].x, yq[mi][j].y, yq[mi][j].z, yq[mi][j].w};
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
              for (int h = 0; h < 2; ++h) {
                const int e = 2 * i + h;
                const float gq = h == 0 ? h16_lo(w4[i]) : h16_hi(w4[i]);
                const float yy = h == 0 ? h16_lo(y4[i]) : h16_hi(y4[i]);
                const bool on = ok && yy * sc[j][e] + sh[j][e] > 0.f;
                const float gv = on ? gq : 0.f;
                s1[j][e] += gv;
                s2[j][e] += gv * yy;
              }
          } else {  // forward: BatchNorm partial statistics of the fp32 results
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float ve = ok ? v[e] : 0.f;
              s1[j][e] += ve;
              s2[j][e] += ve * ve;
            }
          }
        }
      }
      // even lanes send their second channel group and keep the first; odd lanes the other way round (rows pix and pix ^ 1 are the pair's: m0 is even)
      const uint4 va = oj[0], vb = oj[1];
      const unsigned r0 = swap1(odd ? va.x : vb.x), r1 = swap1(odd ? va.y : vb.y), r2 = swap1(odd ? va.z : vb.z), r3 = swap1(odd ? va.w : vb.w);
      const long long pixA = odd ? pix - 1 : pix, pixB = odd ? pix : pix + 1;
      const uint4 vA = odd ? make_uint4(r0, r1, r2, r3) : va;
      const uint4 vB = odd ? vb : make_uint4(r0, r1, r2, r3);
      if (pixA < p.Mg) *reinterpret_cast<uint4*>(out + pixA * BN + chx) = vA;
      if (pixB < p.Mg) *reinterpret_cast<uint4*>(out + pixB * BN + chx) = vB;
    }
    if (sums) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < VE; ++e) {
          const float t1 = row16_sum(s1[j][e]), t2 = row16_sum(s2[j][e]);
          if (li == 0) {
            red[(wm * 2 + 0) * BN + ch0 + j * 32 + e] = t1;
            red[(wm * 2 + 1) * BN + ch0 + j * 32 + e] = t2;
          }
        }
    }
  }
  if (sums) {
    __syncthreads();
    const int which = tid >> 7, c = tid & 127;
    const float v = red[(0 * 2 + which) * BN + c] + red[(1 * 2 + which) * BN + c];
    float* dstp = DGRAD ? p.fpartial : p.bn_partial;
    dstp[((long long)m_tile * 2 + which) * BN + c] = v;
  }
}

// 128 destination channels, 1x1 / stride 1, reduction segments in multiples of 64 and >= 256 long, at least 64 tiles
static bool use_n128(const sh_conv_desc* d, int Ng, int K1, int K2, long long Mg) {
  return sw(SH_SW_N128) && d->dtype == SH_BF16 && d->r == 1 && d->s == 1 && d->stride == 1 && d->pad == 0 && Ng == 128 && K1 % 64 == 0 &&
         K2 % 64 == 0 && K1 + K2 >= 256 && Mg >= 128 * 64;
}
template <bool DGRAD>
static int launch_n128(IgemmArgs a, hipStream_t s) {
  if (a.lda == 0) a.lda = a.Ca;
  route_hit(DGRAD ? SH_ROUTE_N128_DGRAD : SH_ROUTE_N128_FWD);
  gemm_n128_kernel<DGRAD><<<ceil_div(a.Mg, 128), 256, 0, s>>>(a);
  return check_launch(DGRAD ? "conv2d_dgrad (1x1, 128 channels)" : "conv2d_fwd (1x1, 128 channels)");
}

// the 256 x 256 kernel takes the bf16 layers with >= 256 destination channels in multiples of 256 and a reduction long
// enough (>= 8 k-steps of 64) to amortise its one-block-per-CU prologue / epilogue
static hook_t g_use_256{1};
static hook_t g_stem_1x1{1};  // bf16 stem forward on the activation-stationary kernel
static hook_t g_fuse_1x1{0};  // BN-backward sums fused into the short-K 1x1 dgrad (slower: tuning hook)
static bool use_256(int dtype, int Ng, int Ca, int taps, long long Mg, int min_k = 512) {
  if (!g_use_256 || dtype != SH_BF16 || Ng % 256 != 0 || Ca % 64 != 0) return false;
  return g_use_256 == 2 || ((long long)taps * Ca >= min_k && Mg >= 256 * 64);  // 2 = forced (tests)
}
// forward: a stride-2 1x1 (the stage-entry shortcuts) already pays at K = 256 -- its A rows are strided gathers the 128-row kernel's
// register loader handles worse ((256 -> 512)/2 @ 56^2 + BN epilogue: 1008 -> 833 us, scripts/layer_table.py --alternates, round 3)
static bool use_256_fwd(const sh_conv_desc* d, long long Mg) {
  return use_256(d->dtype, d->cout, d->cin, d->r * d->s, Mg, d->stride == 2 && d->r == 1 ? 256 : 512);
}

// data gradient: the stride-2 parity classes (uneven work per class, three empty ones for a 1x1/2) run better as many
// small tiles, so they stay on the 128 x 128 kernel unless forced
static bool use_256_dgrad(const sh_conv_desc* d, long long Mg) {
  // stride 2: the 3x3 layers with >= 256 channels gain 5-7 % on the big tile ((256,256)/2 @ 28^2 741 -> 704 us, (512,512)/2 @ 14^2
  // 572 -> 532, round 3); the 1x1 shortcuts (three empty parity classes) lose 30-38 % and stay on the 128-row kernel
  if (d->stride == 2 && g_use_256 != 2 && !(d->r == 3 && d->cout >= 256)) return false;
  return use_256(d->dtype, d->cin, d->cout, d->r * d->s, Mg);
}

template <typename T, bool DGRAD>
static int launch_igemm(const IgemmArgs& a0, hipStream_t s) {
  IgemmArgs a = a0;
  if (a.lda == 0) a.lda = a.Ca;
  const int nblk = a.classes * a.m_tiles * a.n_tiles;
  route_hit(DGRAD ? SH_ROUTE_IGEMM128_DGRAD : SH_ROUTE_IGEMM128_FWD);
  if (a.Ng % 128 == 0) {
    igemm_kernel<T, DGRAD, 128><<<nblk, 256, 0, s>>>(a);
  } else {
    igemm_kernel<T, DGRAD, 64><<<nblk, 256, 0, s>>>(a);
  }
  return check_launch(DGRAD ? "conv2d_dgrad" : "conv2d_fwd");
}

// The 256 x 256 kernel runs one block per CU, so a launch of T tiles takes ceil(T / CUs) rounds: 1568 tiles (256 ch @ 14^2)
// pay 7 rounds for 6.1 rounds of work.  When the last round would be less than a third full, its m-tiles go to a second
// launch of the 128-row kernel instead (bit-identical results: same k order, same fp32 MFMA chain), which takes a
// fraction of a round.  main_m = m-tiles of the 256-row launch, tail128 = 128-row m-tiles of the second one.
static hook_t g_split256{1};
static hook_t g_tile224{1};
static int num_cus() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
              ? prop.multiProcessorCount : 256;
  }
  return cus;
}
static void split256(long long Mg, int Ng, int classes, int* main_m, int* tail128, int* bm = nullptr, bool allow_tail = true) {
  const int cus = num_cus();
  const int tiles_m = ceil_div(Mg, 256), n_tiles = Ng / 256;
  *main_m = tiles_m;
  *tail128 = 0;
  if (bm) *bm = 256;
  const long long total = (long long)tiles_m * n_tiles;
  const int rem = (int)(total % cus);
  // 224-row tiles when they fill whole rounds and the 256-row plan does not: 7 rounds of 7/8-size tiles beat 6 rounds + a ragged one
  if (g_tile224 && bm && classes == 1 && (rem != 0 || g_tile224 == 2) && Mg % 224 == 0) {
    const long long t224 = Mg / 224 * n_tiles;
    const double cost256 = (double)(total / cus) + 0.75, cost224 = (double)(t224 / cus) * 0.9;
    if (g_tile224 == 2 || (t224 % cus == 0 && cost224 < cost256)) {  // 2 = forced (tests)
      *bm = 224;
      *main_m = (int)(Mg / 224);
      return;
    }
  }
  if (!allow_tail || !g_split256 || classes != 1 || total <= cus || rem == 0 || 3 * rem > cus) return;
  const int tail_m = ceil_div(rem, n_tiles);
  *main_m = tiles_m - tail_m;
  *tail128 = ceil_div(Mg - (long long)*main_m * 256, 128);
}

template <bool DGRAD>
static int launch_igemm256(IgemmArgs a, hipStream_t s) {
  if (a.lda == 0) a.lda = a.Ca;
  int main_m, tail128, bm;
  split256(a.Mg, a.Ng, a.classes, &main_m, &tail128, &bm);
  a.m_tiles = main_m;
  a.n_tiles = a.Ng / 256;
  const int nblk = a.classes * a.m_tiles * a.n_tiles;
  route_hit(DGRAD ? SH_ROUTE_IGEMM256_DGRAD : SH_ROUTE_IGEMM256_FWD);
  if (bm == 224) igemm256_kernel<DGRAD, 7><<<nblk, 512, 0, s>>>(a);
  else igemm256_kernel<DGRAD, 8><<<nblk, 512, 0, s>>>(a);
  if (tail128 > 0) {
    route_hit(SH_ROUTE_IGEMM256_TAIL);
    IgemmArgs t = a;
    t.m_tiles = tail128;
    t.n_tiles = a.Ng / 128;
    t.m_tile_base = 2 * main_m;
    t.part_row_base = main_m;
    igemm_kernel<bf16_t, DGRAD, 128><<<t.m_tiles * t.n_tiles, 256, 0, s>>>(t);
  }
  return check_launch(DGRAD ? "conv2d_dgrad (256x256)" : "conv2d_fwd (256x256)");
}

// ---- fp8 forward on the 256 x 256 kernel (entry point: simhand_conv2d_fwd_fp8, conv_fp8.hip) ----------------------------------
// eligibility mirrors use_256: >= 256 output channels in multiples of 256, cin a multiple of 128 (one k-step = 128 e4m3 bytes of a
// row), a reduction of at least 8 k-steps, enough pixels to fill the chip.  No 128-row tail launch (there is no fp8 128-row twin of
// the bf16 tail kernel): the last round of a launch may be ragged.
bool igemm256_fp8_ok(const sh_conv_desc* d) {
  if (!g_use_256 || d->cout % 256 != 0 || d->cin % 128 != 0) return false;
  const long long mg = (long long)d->n * d->ho * d->wo;
  return g_use_256 == 2 || ((long long)d->r * d->s * d->cin >= 1024 && mg >= 256 * 64);
}
int igemm256_fp8_stat_rows(const sh_conv_desc* d) {
  int main_m, tail128, bm;
  split256((long long)d->n * d->ho * d->wo, d->cout, 1, &main_m, &tail128, &bm, false);
  return main_m;
}
int igemm256_fp8_fwd(const sh_conv_desc* d, const void* xq, const void* wq, const float* x_state, const float* w_state, void* y, float* bn_partial,
                     hipStream_t s) {
  IgemmArgs a;
  a.a = xq; a.w = wq; a.out = y; a.bn_partial = bn_partial;
  a.Mg = (long long)d->n * d->ho * d->wo;
  a.Ng = d->cout; a.Ca = d->cin; a.lda = d->cin;
  a.R = d->r; a.S = d->s; a.stride = d->stride; a.pad = d->pad;
  a.Hd = d->ho; a.Wd = d->wo; a.Hs = d->h; a.Ws = d->w;
  a.accumulate = 0; a.res_grad = nullptr; a.res_mask = nullptr;
  a.classes = 1; a.Hq = a.Wq = 0;
  a.stem_hp = a.stem_wp = 0;
  a.fy = nullptr; a.fscale = a.fshift = nullptr; a.fmask = nullptr; a.fmode = 0; a.fpartial = nullptr; a.bias = nullptr;
  a.ep_scale = a.ep_shift = nullptr; a.ep_res = nullptr; a.ep_mask = nullptr; a.ep_relu = 0;
  a.x_state = x_state; a.w_state = w_state;
  a.div_hw = make_fastdiv((unsigned)(a.Hd * a.Wd));
  a.div_w = make_fastdiv((unsigned)a.Wd);
  int main_m, tail128, bm;
  split256(a.Mg, a.Ng, 1, &main_m, &tail128, &bm, false);
  a.m_tiles = main_m;
  a.n_tiles = a.Ng / 256;
  const int nblk = a.m_tiles * a.n_tiles;
  if (bm == 224) igemm256_kernel<false, 7, true><<<nblk, 512, 0, s>>>(a);
  else igemm256_kernel<false, 8, true><<<nblk, 512, 0, s>>>(a);
  return check_launch("conv2d_fwd_fp8 (256x256)");
}

static int stat_rows256(long long Mg, int Ng, int classes);
static bool dgrad_fp8_ok(const sh_conv_desc* d) {  // the 3x3 layers the 256 x 256 kernel takes, K = cout in whole 128-element k-steps
  const long long mg = d->stride == 2 ? (long long)d->n * ((d->h + 1) / 2) * ((d->w + 1) / 2) : (long long)d->n * d->h * d->w;
  return d->dtype == SH_BF16 && d->r == 3 && d->s == 3 && d->cout % 128 == 0 && d->cin % 256 == 0 && use_256_dgrad(d, mg) &&
         (long long)d->r * d->s * d->cout >= 1024;
}
static int launch_igemm256_fp8_dgrad(IgemmArgs a, hipStream_t s) {
  SH_REQUIRE(a.a2 == nullptr, "conv2d_dgrad fp8: no second reduction segment in the e4m3 kernel");
  if (a.lda == 0) a.lda = a.Ca;
  int main_m, tail128, bm;
  split256(a.Mg, a.Ng, a.classes, &main_m, &tail128, &bm, false);
  a.m_tiles = main_m;
  a.n_tiles = a.Ng / 256;
  const int nblk = a.classes * a.m_tiles * a.n_tiles;
  route_hit(SH_ROUTE_IGEMM256_DGRAD);
  route_hit(SH_ROUTE_FP8_DGRAD);
  if (a.fpartial != nullptr) {
    // the caller sized the partial-sum buffer with simhand_conv2d_dgrad_stat_blocks (the bf16 plan, which may hand a ragged last round
    // to a 128-row tail launch = more rows than this single launch writes): rows nobody writes must read as zero
    const size_t rows = (size_t)stat_rows256(a.Mg, a.Ng, a.classes);
    if (hipMemsetAsync(a.fpartial, 0, rows * 2 * a.Ng * sizeof(float), s) != hipSuccess) return check_launch("conv2d_dgrad fp8 memset");
  }
  if (bm == 224) igemm256_kernel<true, 7, true><<<nblk, 512, 0, s>>>(a);
  else igemm256_kernel<true, 8, true><<<nblk, 512, 0, s>>>(a);
  return check_launch("conv2d_dgrad fp8 (256x256)");
}

// rows of the partial-sum buffers a 256 x 256 launch writes
static int stat_rows256(long long Mg, int Ng, int classes) {
  int main_m, tail128, bm;
  split256(Mg, Ng, classes, &main_m, &tail128, &bm);
  return classes * (main_m + tail128);
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

// short-K stride-1 1x1 layers in bf16 go to the activation-stationary kernel (conv_1x1.hip)
static bool use_1x1(const sh_conv_desc* d, int k, int n) {
  return d->dtype == SH_BF16 && d->r == 1 && d->s == 1 && d->stride == 1 && d->pad == 0 && gemm1x1_supported(k, n);
}

void hooks_reset_igemm() {
  g_use_256 = 1;
  g_split256 = 1;
  g_tile224 = 1;
  g_stem_1x1 = 1;
  g_fuse_1x1 = 0;
}

}  // namespace sh

using namespace sh;

extern "C" {

// test / tuning hook: rows per block (64 * mf, mf in {1, 2, 4}; K = 256 supports {1, 2}) of the short-K 1x1 kernel
int simhand_test_conv1x1_set_rows(int k, int mf) {
  SH_REQUIRE((k == 64 || k == 128 || k == 256) && (mf == 1 || mf == 2 || (mf == 4 && k != 256)), "conv1x1_set_rows: bad k=%d mf=%d", k, mf);
  gemm1x1_set_mf(k, mf);
  return 0;
}

// tuning hook: route eligible layers to the 256 x 256 LDS-DMA kernel (0 = never, 1 = default heuristic, 2 = whenever legal)
int simhand_test_igemm256_split_tail(int on) {
  g_split256 = on ? 1 : 0;
  return 0;
}

int simhand_test_igemm256_tile224(int mode) {
  g_tile224 = mode < 0 ? 0 : (mode > 2 ? 2 : mode);
  return 0;
}

int simhand_test_igemm256_enable(int on) {
  g_use_256 = on < 0 ? 0 : (on > 2 ? 2 : on);
  return 0;
}

// 64 -> 64 channel 3x3 / stride 1 on the padded pixel grid with register-resident weights (conv3x3_c64.hip)
static long long c64_q_total(const sh_conv_desc* d) { return (long long)d->n * (d->h + 1) * (d->w + 1); }
static bool use_c64(const sh_conv_desc* d) {
  return c64_supported(d->dtype, d->cin, d->cout, d->r, d->s, d->stride, d->pad, d->w, c64_q_total(d));
}
// the data-gradient form stores only (no accumulate / residual merge / bias) and fuses relu_mode 0 / 2 sums
static bool use_c64_dgrad(const sh_conv_desc* d, int accumulate, int relu_mode, bool has_bias) {
  return use_c64(d) && accumulate == 0 && !has_bias && (relu_mode < 0 || relu_mode == 0 || relu_mode == 2);
}
struct BnIn {  // the previous unit's BatchNorm + ReLU applied inside the ring (simhand_conv2d_fwd_bnin)
  const float* scale;
  const float* shift;
  void* a_out;
};
static int launch_c64_conv(const sh_conv_desc* d, const void* x, const void* w, void* out, float* partial, bool dgrad,
                           const sh_bn_bwd_fuse* fuse, hipStream_t s, const BnIn* bnin = nullptr) {
  C64Args c;
  c.x = (const bf16_t*)x; c.w = (const bf16_t*)w; c.out = (bf16_t*)out; c.partial = partial;
  c.in_scale = bnin ? bnin->scale : nullptr;
  c.in_shift = bnin ? bnin->shift : nullptr;
  c.a_out = bnin ? (bf16_t*)bnin->a_out : nullptr;
  c.fy = fuse ? (const bf16_t*)fuse->y : nullptr;
  c.fscale = fuse ? fuse->scale : nullptr;
  c.fshift = fuse ? fuse->shift : nullptr;
  c.relu = fuse && fuse->relu_mode == 2 ? 1 : 0;
  c.N = d->n; c.H = d->h; c.W = d->w; c.dgrad = dgrad ? 1 : 0;
  c.q_total = c64_q_total(d);
  // the kernels' pixel / element arithmetic is 32-bit: padded positions (boff, q) and, for the BatchNorm-on-load by-product, off[]
  SH_REQUIRE(c.q_total < (1ll << 31), "conv3x3 c64: padded pixel grid exceeds 2^31 positions");
  SH_REQUIRE(bnin == nullptr || (long long)d->n * d->h * d->w * 64 < (1ll << 32), "conv3x3 c64 (bnin): n*h*w*64 exceeds the 32-bit element offsets");
  c.steps_per_block = 0;
  c.div_pp = make_fastdiv((unsigned)((d->h + 1) * (d->w + 1)));
  c.div_wp = make_fastdiv((unsigned)(d->w + 1));
  launch_c64(c, s);
  return check_launch(dgrad ? "conv2d_dgrad (3x3 c64)" : "conv2d_fwd (3x3 c64)");
}

int simhand_test_conv3x3_c64_enable(int on) {
  c64_enable(on);
  return 0;
}

// 128 -> 128 channel 3x3 / stride 1: activation tile staged once per 256 padded positions, weights streamed per tap (conv3x3_ring.hip)
static bool use_r128(const sh_conv_desc* d) {
  return r128_supported(d->dtype, d->cin, d->cout, d->r, d->s, d->stride, d->pad, d->w, c64_q_total(d));
}
// the data-gradient form stores only (no accumulate / residual merge / bias) and fuses relu_mode 0 / 2 sums, as the c64 kernel
static bool use_r128_dgrad(const sh_conv_desc* d, int accumulate, int relu_mode, bool has_bias) {
  return use_r128(d) && accumulate == 0 && !has_bias && (relu_mode < 0 || relu_mode == 0 || relu_mode == 2);
}
static int launch_r128_conv(const sh_conv_desc* d, const void* x, const void* w, void* out, float* partial, bool dgrad,
                            const sh_bn_bwd_fuse* fuse, hipStream_t s, const BnIn* bnin = nullptr) {
  R128Args c;
  c.x = (const bf16_t*)x; c.w = (const bf16_t*)w; c.out = (bf16_t*)out; c.partial = partial;
  c.in_scale = bnin ? bnin->scale : nullptr;
  c.in_shift = bnin ? bnin->shift : nullptr;
  c.a_out = bnin ? (bf16_t*)bnin->a_out : nullptr;
  c.fy = fuse ? (const bf16_t*)fuse->y : nullptr;
  c.fscale = fuse ? fuse->scale : nullptr;
  c.fshift = fuse ? fuse->shift : nullptr;
  c.relu = fuse && fuse->relu_mode == 2 ? 1 : 0;
  c.N = d->n; c.H = d->h; c.W = d->w; c.dgrad = dgrad ? 1 : 0;
  c.q_total = c64_q_total(d);
  SH_REQUIRE(c.q_total < (1ll << 31), "conv3x3 r128: padded pixel grid exceeds 2^31 positions");
  SH_REQUIRE(bnin == nullptr || (long long)d->n * d->h * d->w * 128 < (1ll << 32), "conv3x3 r128 (bnin): n*h*w*128 exceeds the 32-bit element offsets");
  c.tiles = 0;
  c.div_pp = make_fastdiv((unsigned)((d->h + 1) * (d->w + 1)));
  c.div_wp = make_fastdiv((unsigned)(d->w + 1));
  launch_r128(c, s);
  return check_launch(dgrad ? "conv2d_dgrad (3x3 ring)" : "conv2d_fwd (3x3 ring)");
}

int simhand_test_conv3x3_r128_enable(int on) {
  r128_enable(on);
  return 0;
}

int simhand_conv2d_fwd_stat_blocks(const sh_conv_desc* d) {
  if (!d) return 0;
  if (use_c64(d)) return c64_blocks(c64_q_total(d));
  if (use_r128(d)) return r128_blocks(c64_q_total(d));
  const long long m = (long long)d->n * d->ho * d->wo;
  if (use_1x1(d, d->cin, d->cout)) return ceil_div(m, gemm1x1_rows_per_block(d->cin));
  if (use_256_fwd(d, m)) return stat_rows256(m, d->cout, 1);
  return ceil_div(m, 128);
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
  a.accumulate = 0; a.res_grad = nullptr; a.res_mask = nullptr;
  a.classes = 1; a.Hq = a.Wq = 0;
  a.stem_hp = a.stem_wp = 0;
  a.fy = nullptr; a.fscale = a.fshift = nullptr; a.fmask = nullptr; a.fmode = 0; a.fpartial = nullptr; a.bias = nullptr;
  a.ep_scale = a.ep_shift = nullptr; a.ep_res = nullptr; a.ep_mask = nullptr; a.ep_relu = 0;
  SH_REQUIRE(a.Mg < (1ll << 31) - 256, "conv2d_fwd: %lld output pixels exceed the 2^31 index range", a.Mg);
  a.div_hw = make_fastdiv((unsigned)(a.Hd * a.Wd));
  a.div_w = make_fastdiv((unsigned)a.Wd);
  a.m_tiles = ceil_div(a.Mg, 128);
  a.n_tiles = a.Ng % 128 == 0 ? a.Ng / 128 : a.Ng / 64;
  const double flops = 2.0 * (double)a.Mg * d->cout * d->cin * d->r * d->s;
  const double es = d->dtype == SH_F32 ? 4 : 2;
  const double bytes = es * ((double)d->n * d->h * d->w * d->cin + (double)a.Mg * d->cout + (double)d->cout * d->cin * d->r * d->s);
  ProfScope ps(SH_PROF_CONV_FWD, (hipStream_t)stream, flops, bytes);
  if (use_1x1(d, d->cin, d->cout)) {
    Gemm1x1Args g;
    g.a = (const bf16_t*)x; g.w = (const bf16_t*)w; g.out = (bf16_t*)y; g.bn_partial = bn_partial;
    g.M = a.Mg; g.N = d->cout; g.accumulate = 0; g.res_grad = nullptr; g.res_mask = nullptr;
    g.fy = nullptr; g.fscale = g.fshift = nullptr; g.fmask = nullptr; g.fmode = 0; g.fpartial = nullptr; g.bias = nullptr;
    g.ep_scale = g.ep_shift = nullptr; g.ep_res = nullptr; g.ep_mask = nullptr; g.ep_relu = 0;
    launch_gemm1x1(g, d->cin, false, (hipStream_t)stream);
    return check_launch("conv2d_fwd (1x1)");
  }
  if (use_c64(d)) return launch_c64_conv(d, x, w, y, bn_partial, false, nullptr, (hipStream_t)stream);
  if (use_r128(d)) return launch_r128_conv(d, x, w, y, bn_partial, false, nullptr, (hipStream_t)stream);
  if (use_256_fwd(d, a.Mg)) return launch_igemm256<false>(a, (hipStream_t)stream);
  if (use_n128(d, d->cout, d->cin, 0, a.Mg)) return launch_n128<false>(a, (hipStream_t)stream);
  return d->dtype == SH_F32 ? launch_igemm<float, false>(a, (hipStream_t)stream) : launch_igemm<bf16_t, false>(a, (hipStream_t)stream);
}

int simhand_conv2d_fwd_bnin_ok(const sh_conv_desc* d) {
  if (d == nullptr || d->dtype != SH_BF16) return 0;
  // the by-product's element offsets are 32-bit in the kernel
  return (use_c64(d) || use_r128(d)) && (long long)d->n * d->h * d->w * d->cin < (1ll << 32) ? 1 : 0;
}

int simhand_conv2d_fwd_bnin(const sh_conv_desc* d, const void* y_in, const float* in_scale, const float* in_shift, const void* w, void* a_out,
                            void* y, float* bn_partial, sh_stream_t stream) {
  if (check_desc(d, "conv2d_fwd_bnin")) return 1;
  SH_REQUIRE(y_in && in_scale && in_shift && w && a_out && y, "conv2d_fwd_bnin: NULL pointer");
  SH_REQUIRE(simhand_conv2d_fwd_bnin_ok(d), "conv2d_fwd_bnin: no ring kernel for this layer (simhand_conv2d_fwd_bnin_ok)");
  // a tile re-reads its neighbours' halo rows: in place, some of them would already hold the activation
  SH_REQUIRE(a_out != y_in && a_out != y && y != y_in, "conv2d_fwd_bnin: y_in, a_out and y must be three distinct tensors");
  const double mg = (double)d->n * d->ho * d->wo;
  const double flops = 2.0 * mg * d->cout * d->cin * d->r * d->s;
  const double bytes = 2.0 * (2.0 * (double)d->n * d->h * d->w * d->cin + mg * d->cout + (double)d->cout * d->cin * d->r * d->s);
  ProfScope ps(SH_PROF_CONV_FWD, (hipStream_t)stream, flops, bytes);
  const BnIn b = {in_scale, in_shift, a_out};
  if (use_r128(d)) return launch_r128_conv(d, y_in, w, y, bn_partial, false, nullptr, (hipStream_t)stream, &b);
  return launch_c64_conv(d, y_in, w, y, bn_partial, false, nullptr, (hipStream_t)stream, &b);
}

int simhand_conv2d_fwd_bnact(const sh_conv_desc* d, const void* x, const void* w, const float* scale, const float* shift,
                             const void* residual, int relu, void* out, uint8_t* relu_mask, sh_stream_t stream) {
  if (check_desc(d, "conv2d_fwd_bnact")) return 1;
  SH_REQUIRE(x && w && out && scale && shift, "conv2d_fwd_bnact: NULL pointer");
  SH_REQUIRE(d->dtype == SH_BF16, "conv2d_fwd_bnact: bf16 only");
  SH_REQUIRE(!relu_mask || relu, "conv2d_fwd_bnact: a ReLU mask is only produced with relu != 0");
  SH_REQUIRE(d->cout <= 2048, "conv2d_fwd_bnact: cout=%d > 2048 (epilogue coefficient cache)", d->cout);
  IgemmArgs a;
  a.a = x; a.w = w; a.out = out; a.bn_partial = nullptr;
  a.Mg = (long long)d->n * d->ho * d->wo;
  a.Ng = d->cout; a.Ca = d->cin;
  a.R = d->r; a.S = d->s; a.stride = d->stride; a.pad = d->pad;
  a.Hd = d->ho; a.Wd = d->wo; a.Hs = d->h; a.Ws = d->w;
  a.accumulate = 0; a.res_grad = nullptr; a.res_mask = nullptr;
  a.classes = 1; a.Hq = a.Wq = 0;
  a.stem_hp = a.stem_wp = 0;
  a.fy = nullptr; a.fscale = a.fshift = nullptr; a.fmask = nullptr; a.fmode = 0; a.fpartial = nullptr; a.bias = nullptr;
  a.ep_scale = scale; a.ep_shift = shift; a.ep_res = residual; a.ep_mask = relu_mask; a.ep_relu = relu;
  SH_REQUIRE(a.Mg < (1ll << 31) - 256, "conv2d_fwd_bnact: %lld output pixels exceed the 2^31 index range", a.Mg);
  a.div_hw = make_fastdiv((unsigned)(a.Hd * a.Wd));
  a.div_w = make_fastdiv((unsigned)a.Wd);
  a.m_tiles = ceil_div(a.Mg, 128);
  a.n_tiles = a.Ng % 128 == 0 ? a.Ng / 128 : a.Ng / 64;
  const double flops = 2.0 * (double)a.Mg * d->cout * d->cin * d->r * d->s;
  const double bytes = 2.0 * ((double)d->n * d->h * d->w * d->cin + (double)a.Mg * d->cout * (residual ? 2 : 1) + (double)d->cout * d->cin * d->r * d->s);
  ProfScope ps(SH_PROF_CONV_FWD, (hipStream_t)stream, flops, bytes);
  route_hit(SH_ROUTE_FWD_BNACT);
  if (use_1x1(d, d->cin, d->cout) && !(d->cin == 128 && d->cout > 1024)) {  // K = 128: the 1x1 kernel caches 1024 channels' coefficients
    Gemm1x1Args g;
    g.a = (const bf16_t*)x; g.w = (const bf16_t*)w; g.out = (bf16_t*)out; g.bn_partial = nullptr;
    g.M = a.Mg; g.N = d->cout; g.accumulate = 0; g.res_grad = nullptr; g.res_mask = nullptr;
    g.fy = nullptr; g.fscale = g.fshift = nullptr; g.fmask = nullptr; g.fmode = 0; g.fpartial = nullptr; g.bias = nullptr;
    g.ep_scale = scale; g.ep_shift = shift; g.ep_res = (const bf16_t*)residual; g.ep_mask = relu_mask; g.ep_relu = relu;
    launch_gemm1x1(g, d->cin, false, (hipStream_t)stream);
    return check_launch("conv2d_fwd_bnact (1x1)");
  }
  if (use_256_fwd(d, a.Mg)) return launch_igemm256<false>(a, (hipStream_t)stream);
  return launch_igemm<bf16_t, false>(a, (hipStream_t)stream);
}

int simhand_test_conv1x1_chain_mask(int mask) {
  gemm1x1_set_chain(mask);
  return 0;
}

int simhand_conv2d_fwd_chain_ok(const sh_conv_desc* d) {
  return d != nullptr && use_1x1(d, d->cin, d->cout) && gemm1x1_chain_ok(d->cin, d->cout, (long long)d->n * d->ho * d->wo) ? 1 : 0;
}

int simhand_conv2d_fwd_chain_stat_blocks(const sh_conv_desc* d) {
  return d ? ceil_div((long long)d->n * d->ho * d->wo, gemm1x1_chain_rows(d->cin)) : 0;
}

int simhand_conv2d_fwd_bnact_chain(const sh_conv_desc* d, const void* x, const void* w, const float* scale, const float* shift,
                                   const void* residual, void* out, uint8_t* relu_mask, const void* chain_w, void* chain_y,
                                   float* chain_partial, sh_stream_t stream) {
  if (check_desc(d, "conv2d_fwd_bnact_chain")) return 1;
  SH_REQUIRE(x && w && out && scale && shift && residual && relu_mask && chain_w && chain_y && chain_partial,
             "conv2d_fwd_bnact_chain: NULL pointer");
  SH_REQUIRE(simhand_conv2d_fwd_chain_ok(d), "conv2d_fwd_bnact_chain: layer not supported (see simhand_conv2d_fwd_chain_ok)");
  const long long m = (long long)d->n * d->ho * d->wo;
  // the chained conv1's FLOPs are algorithmic work of the forward class too
  ProfScope ps(SH_PROF_CONV_FWD, (hipStream_t)stream, 2.0 * 2.0 * (double)m * d->cout * d->cin,
               2.0 * ((double)m * d->cin * 2 + (double)m * d->cout * 2));
  route_hit(SH_ROUTE_FWD_BNACT);
  Gemm1x1Args g;
  g.a = (const bf16_t*)x; g.w = (const bf16_t*)w; g.out = (bf16_t*)out; g.bn_partial = nullptr;
  g.M = m; g.N = d->cout; g.accumulate = 0; g.res_grad = nullptr; g.res_mask = nullptr;
  g.fy = nullptr; g.fscale = g.fshift = nullptr; g.fmask = nullptr; g.fmode = 0; g.fpartial = nullptr; g.bias = nullptr;
  g.ep_scale = scale; g.ep_shift = shift; g.ep_res = (const bf16_t*)residual; g.ep_mask = relu_mask; g.ep_relu = 1;
  g.chain_w = (const bf16_t*)chain_w; g.chain_y = (bf16_t*)chain_y; g.chain_partial = chain_partial;
  launch_gemm1x1(g, d->cin, false, (hipStream_t)stream);
  return check_launch("conv2d_fwd_bnact_chain");
}

// ---- direct 7x7 / stride 2 / pad 3 / 3 -> 64 stem --------------------------------------------------------------------
// torchvision ResNet conv1 (reference: src/models/resnet_model.py:13-26).  Cin = 3 cannot form a k-contiguous MFMA
// operand, so the input is first repacked (simhand_stem_pad_input) to zero-padded NHWC4 [N][h+8][wp][4]; then filter
// row r of an output pixel is ONE contiguous run of 8 taps x 4 channels (tap 7 and channel 3 carry zero weights) and
// the stem is a K = 8 rows x 32 = 256 GEMM read straight from that buffer: no im2col matrix (9.9 GB at 2048 x 224^2).
int simhand_stem_geometry(int h, int w, int* hp, int* wp, int* ho, int* wo) {
  SH_REQUIRE(h >= 1 && w >= 1 && hp && wp && ho && wo, "stem_geometry: bad arguments");
  *ho = (h + 6 - 7) / 2 + 1;
  *wo = (w + 6 - 7) / 2 + 1;
  *hp = h + 8;
  *wp = (w + 8 + 7) / 8 * 8;
  return 0;
}

int simhand_stem_conv_fwd_stat_blocks(int n, int h, int w, int dtype) {
  int hp, wp, ho, wo;
  if (n < 1 || simhand_stem_geometry(h, w, &hp, &wp, &ho, &wo)) return 0;
  const long long m = (long long)n * ho * wo;
  if (dtype == SH_BF16 && g_stem_1x1 && stem_ring_ok(n, hp, wp, ho, wo)) return n;  // one partial row per image
  return dtype == SH_BF16 && g_stem_1x1 ? gemm1x1_stem_stat_blocks(m) : ceil_div(m, 128);  // partial rows of the kernel the same arguments select
}

int simhand_stem_conv_fwd(const void* xp, const void* wp_, void* y, float* bn_partial, int n, int h, int w, int dtype,
                          sh_stream_t stream) {
  SH_REQUIRE(xp && wp_, "stem_conv_fwd: NULL pointer");
  SH_REQUIRE(dtype == SH_F32 || dtype == SH_BF16, "stem_conv_fwd: bad dtype %d", dtype);
  SH_REQUIRE(n >= 1, "stem_conv_fwd: bad shape");
  int hp, wp, ho, wo;
  if (simhand_stem_geometry(h, w, &hp, &wp, &ho, &wo)) return 1;
  SH_REQUIRE(y != nullptr, "stem_conv_fwd: NULL pointer");
  const int ke = dtype == SH_F32 ? 32 : 64;
  IgemmArgs a;
  a.a = xp; a.w = wp_; a.out = y; a.bn_partial = bn_partial;
  a.Mg = (long long)n * ho * wo;
  a.Ng = 64; a.Ca = ke;
  a.R = 256 / ke; a.S = 1; a.stride = 1; a.pad = 0;
  a.Hd = ho; a.Wd = wo; a.Hs = a.R + 1; a.Ws = wp / 8;
  a.accumulate = 0; a.res_grad = nullptr; a.res_mask = nullptr;
  a.classes = 1; a.Hq = a.Wq = 0;
  a.stem_hp = hp; a.stem_wp = wp;
  a.fy = nullptr; a.fscale = a.fshift = nullptr; a.fmask = nullptr; a.fmode = 0; a.fpartial = nullptr; a.bias = nullptr;
  a.ep_scale = a.ep_shift = nullptr; a.ep_res = nullptr; a.ep_mask = nullptr; a.ep_relu = 0;
  SH_REQUIRE(a.Mg < (1ll << 31) - 256, "stem_conv_fwd: %lld output pixels exceed the 2^31 index range", a.Mg);
  SH_REQUIRE((long long)n * hp * wp * 4 < (1ll << 40), "stem_conv_fwd: input too large");
  a.div_hw = make_fastdiv((unsigned)(ho * wo));
  a.div_w = make_fastdiv((unsigned)wo);
  a.m_tiles = ceil_div(a.Mg, 128);
  a.n_tiles = 1;
  const double flops = 2.0 * (double)a.Mg * 64 * 147;
  const double es = dtype == SH_F32 ? 4 : 2;
  const double bytes = es * ((double)n * hp * wp * 4 + (double)a.Mg * 64 + 64.0 * 256);
  ProfScope ps(SH_PROF_CONV_FWD, (hipStream_t)stream, flops, bytes);
  route_hit(SH_ROUTE_STEM_FWD);
  if (dtype == SH_BF16 && g_stem_1x1 && stem_ring_ok(n, hp, wp, ho, wo)) {
    launch_stem_ring(xp, wp_, y, bn_partial, n, hp, wp, ho, wo, (hipStream_t)stream);
    return check_launch("stem_conv_fwd (input rows in an LDS ring)");
  }
  if (dtype == SH_BF16 && g_stem_1x1) {
    // bf16: the activation-stationary kernel (conv_1x1.hip) with the stem's row addressing -- the layer writes 3.7x what it
    // reads and has only 4 k-steps per tile, the regime that kernel was built for (1.81 -> see DESIGN ms at 2048 x 224^2)
    Gemm1x1Args g;
    g.a = (const bf16_t*)xp; g.w = (const bf16_t*)wp_; g.out = (bf16_t*)y; g.bn_partial = bn_partial;
    g.M = a.Mg; g.N = 64; g.accumulate = 0; g.res_grad = nullptr; g.res_mask = nullptr;
    g.fy = nullptr; g.fscale = g.fshift = nullptr; g.fmask = nullptr; g.fmode = 0; g.fpartial = nullptr; g.bias = nullptr;
    g.ep_scale = g.ep_shift = nullptr; g.ep_res = nullptr; g.ep_mask = nullptr; g.ep_relu = 0;
    g.stem_hp = hp; g.stem_wp = wp;
    g.div_hw = a.div_hw; g.div_w = a.div_w;
    launch_gemm1x1_stem(g, (hipStream_t)stream);
    return check_launch("stem_conv_fwd (activation-stationary)");
  }
  return dtype == SH_F32 ? launch_igemm<float, false>(a, (hipStream_t)stream) : launch_igemm<bf16_t, false>(a, (hipStream_t)stream);
}

int simhand_test_stem_conv_route(int mode) {
  g_stem_1x1 = mode ? 1 : 0;
  gemm1x1_set_stem_persistent(mode != 2);
  return 0;
}

// tiles (= rows of the fused BatchNorm-backward partial buffer) of the data-gradient launch
static int dgrad_stat_blocks(const sh_conv_desc* d, int accumulate, int relu_mode, int c2 = 0) {
  if (use_c64_dgrad(d, accumulate, relu_mode, false)) return c64_blocks(c64_q_total(d));
  if (use_r128_dgrad(d, accumulate, relu_mode, false)) return r128_blocks(c64_q_total(d));
  if (c2 == 0 && use_1x1(d, d->cout, d->cin)) return ceil_div((long long)d->n * d->h * d->w, gemm1x1_rows_per_block(d->cout));
  const long long mg = d->stride == 2 ? (long long)d->n * ((d->h + 1) / 2) * ((d->w + 1) / 2) : (long long)d->n * d->h * d->w;
  if (use_256_dgrad(d, mg)) return stat_rows256(mg, d->cin, d->stride == 2 ? 4 : 1);
  return (d->stride == 2 ? 4 : 1) * ceil_div(mg, 128);
}

// second reduction segment (sh_dgrad_opts.x2): 1x1 / stride-1 bf16 layers, always on the tile kernels (also where a single
// segment would take the activation-stationary kernel: 256 -> 64 @ 56^2 measured 1.4 ms per step faster in one launch)
static bool concat_ok(const sh_conv_desc* d, int c2) {
  return d->dtype == SH_BF16 && d->r == 1 && d->s == 1 && d->stride == 1 && d->pad == 0 && c2 > 0 && c2 % 64 == 0;
}

static int dgrad_impl(const sh_conv_desc* d, const void* dy, const void* wt, void* dx, int accumulate, const void* res_grad,
                      const unsigned char* res_mask, sh_stream_t stream, const sh_bn_bwd_fuse* fuse = nullptr,
                      const float* bias = nullptr, const void* x2 = nullptr, const void* wt2 = nullptr, int c2 = 0,
                      const sh_dy_src* src = nullptr, const sh_dgrad_opts* f8 = nullptr) {
  if (check_desc(d, "conv2d_dgrad")) return 1;
  const sh_dgrad_opts* f8sub = f8;  // (the whole option block: sub_grad is read from it below)
  if (f8 != nullptr && f8->dy_q == nullptr) f8 = nullptr;
  if (f8 != nullptr) {
    SH_REQUIRE(f8->wt_q && f8->dy_state && f8->w_state, "conv2d_dgrad_ex: fp8 operands need wt_q and both scale states");
    SH_REQUIRE(x2 == nullptr && src == nullptr && dgrad_fp8_ok(d), "conv2d_dgrad_ex: fp8 operands only where simhand_conv2d_dgrad_fp8_pays(d)");
    dy = f8->dy_q;
    wt = f8->wt_q;
  }
  if (src != nullptr) {
    SH_REQUIRE(src->da && src->y && src->scale && src->shift && src->coef_a && src->coef_b && src->coef_c && src->dy_out,
               "conv2d_dgrad_ex: dy_src has a NULL member");
    SH_REQUIRE(x2 == nullptr && use_1x1(d, d->cout, d->cin),
               "conv2d_dgrad_ex: dy_src needs a layer simhand_conv2d_dgrad_dysrc_ok accepts and a single reduction segment");
    dy = src->da;
  }
  SH_REQUIRE(dy && wt && dx, "conv2d_dgrad: NULL pointer");
  const void* sub = f8sub != nullptr ? f8sub->sub_grad : nullptr;
  SH_REQUIRE(sub == nullptr || (accumulate == 0 && d->stride == 1 && d->h % 2 == 0 && d->w % 2 == 0 && d->dtype == SH_BF16),
             "conv2d_dgrad_ex: sub_grad needs accumulate 0, a stride-1 bf16 layer and even h / w");
  // the merge where the selected kernel has no epilogue for it: one pass over the even pixels of dx (through the same gate)
  auto sub_fallback = [&]() -> int {
    return simhand_scatter2_add(sub, dx, fuse != nullptr && fuse->relu_mode == 4 ? fuse->mask : nullptr, d->n, d->h, d->w, d->cin, d->dtype, stream);
  };
  SH_REQUIRE(x2 == nullptr || (wt2 != nullptr && concat_ok(d, c2)),
             "conv2d_dgrad_ex: a second reduction segment needs wt2 and a layer simhand_conv2d_dgrad_concat_ok accepts");
  const int ke = d->dtype == SH_F32 ? 32 : 64;
  SH_REQUIRE(d->cout % ke == 0, "conv2d_dgrad: cout=%d must be a multiple of %d", d->cout, ke);
  SH_REQUIRE(d->cin % 64 == 0, "conv2d_dgrad: cin=%d must be a multiple of 64", d->cin);
  IgemmArgs a;
  a.a = dy; a.w = wt; a.out = dx; a.bn_partial = nullptr;
  a.Ng = d->cin; a.Ca = d->cout;
  a.R = d->r; a.S = d->s; a.stride = d->stride; a.pad = d->pad;
  a.Hd = d->h; a.Wd = d->w; a.Hs = d->ho; a.Ws = d->wo;
  a.accumulate = accumulate; a.res_grad = res_grad; a.res_mask = res_mask;
  a.stem_hp = a.stem_wp = 0;
  a.fy = nullptr; a.fscale = a.fshift = nullptr; a.fmask = nullptr; a.fmode = 0; a.fpartial = nullptr; a.bias = nullptr;
  a.ep_scale = a.ep_shift = nullptr; a.ep_res = nullptr; a.ep_mask = nullptr; a.ep_relu = 0;
  if (fuse != nullptr) {
    SH_REQUIRE(fuse->partial || fuse->relu_mode == 4, "conv2d_dgrad_fused: NULL partial (only relu_mode 4 may omit the sums)");
    SH_REQUIRE(fuse->relu_mode == 0 || (fuse->relu_mode >= 2 && fuse->relu_mode <= 4), "conv2d_dgrad_fused: relu_mode %d", fuse->relu_mode);
    SH_REQUIRE(fuse->relu_mode == 4 || fuse->y, "conv2d_dgrad_fused: NULL y");
    SH_REQUIRE(fuse->relu_mode != 2 || (fuse->scale && fuse->shift), "conv2d_dgrad_fused: relu_mode 2 needs scale / shift");
    SH_REQUIRE(fuse->relu_mode < 3 || fuse->mask, "conv2d_dgrad_fused: relu_mode 3 / 4 needs the bit mask");
    // a stride-2 1x1 shortcut that accumulates skips the parity classes no tap reaches: their pixels would be missing
    SH_REQUIRE(fuse->partial == nullptr || !(accumulate == 1 && d->stride == 2 && d->r == 1),
               "conv2d_dgrad_fused: sums are not available for an accumulating stride-2 1x1 (it skips the classes no tap reaches)");
    a.fy = fuse->y; a.fscale = fuse->scale; a.fshift = fuse->shift; a.fmask = fuse->mask; a.fmode = fuse->relu_mode;
    a.fpartial = fuse->partial;
  }
  a.bias = bias;
  if (x2 != nullptr) {
    a.a2 = x2; a.w2 = wt2; a.Ca2 = c2;
    a.lda = d->cout;
    a.Ca = d->cout + c2;
  }
  if (d->stride == 2) {
    a.classes = 4;
    a.Hq = (d->h + 1) / 2;
    a.Wq = (d->w + 1) / 2;
    a.Mg = (long long)d->n * a.Hq * a.Wq;
  } else {
    a.classes = 1;
    a.Hq = a.Wq = 0;
    a.Mg = (long long)d->n * d->h * d->w;
  }
  SH_REQUIRE(a.Mg < (1ll << 31) - 256, "conv2d_dgrad: %lld pixels exceed the 2^31 index range", a.Mg);
  a.div_hw = make_fastdiv((unsigned)(a.classes == 4 ? a.Hq * a.Wq : a.Hd * a.Wd));
  a.div_w = make_fastdiv((unsigned)(a.classes == 4 ? a.Wq : a.Wd));
  a.m_tiles = ceil_div(a.Mg, 128);
  a.n_tiles = a.Ng % 128 == 0 ? a.Ng / 128 : a.Ng / 64;
  const long long mo = (long long)d->n * d->ho * d->wo;
  const double flops = 2.0 * (double)mo * d->cout * d->cin * d->r * d->s;
  const double es = d->dtype == SH_F32 ? 4 : 2;
  const double bytes = es * ((double)d->n * d->h * d->w * d->cin * ((accumulate ? 2 : 1) + (fuse ? 1 : 0)) + (double)mo * d->cout +
                             (double)d->cout * d->cin * d->r * d->s);
  ProfScope ps(SH_PROF_CONV_DGRAD, (hipStream_t)stream, flops, bytes);
  if (x2 != nullptr) route_hit(SH_ROUTE_DGRAD_CONCAT);
  if (fuse != nullptr) route_hit(SH_ROUTE_DGRAD_FUSED_SUMS);
  if (d->stride == 2) route_hit(SH_ROUTE_DGRAD_PARITY);
  if (x2 == nullptr && use_1x1(d, d->cout, d->cin)) {
    Gemm1x1Args g;
    g.a = (const bf16_t*)dy; g.w = (const bf16_t*)wt; g.out = (bf16_t*)dx; g.bn_partial = nullptr;
    g.M = a.Mg; g.N = d->cin; g.accumulate = accumulate; g.res_grad = (const bf16_t*)res_grad; g.res_mask = res_mask;
    g.fy = (const bf16_t*)a.fy; g.fscale = a.fscale; g.fshift = a.fshift; g.fmask = a.fmask; g.fmode = a.fmode; g.fpartial = a.fpartial;
    g.bias = a.bias;
    if (src != nullptr) {
      g.xf_y = (const bf16_t*)src->y; g.xf_s = src->scale; g.xf_h = src->shift; g.xf_a = src->coef_a; g.xf_b = src->coef_b;
      g.xf_c = src->coef_c; g.xf_out = (bf16_t*)src->dy_out; g.xf_relu = src->relu;
      route_hit(SH_ROUTE_DGRAD_DYSRC);
    }
    bool merged = false;
    if (sub != nullptr) {
      g.sub = (const bf16_t*)sub; g.sub_h = d->h; g.sub_w = d->w;
      g.div_hw = make_fastdiv((unsigned)(d->h * d->w));
      g.div_w = make_fastdiv((unsigned)d->w);
      merged = gemm1x1_sub_ok(g, d->cout);
      if (!merged) g.sub = nullptr;
    }
    launch_gemm1x1(g, d->cout, true, (hipStream_t)stream);
    if (check_launch("conv2d_dgrad (1x1)")) return 1;
    return sub != nullptr && !merged ? sub_fallback() : 0;
  }
  // 128 destination channels behind a long reduction (conv3 of the stage-2 Bottlenecks, with the folded second segment): LDS-DMA tiles, two blocks per CU
  if (sub == nullptr && src == nullptr && f8 == nullptr && accumulate == 0 && res_grad == nullptr &&
      (fuse == nullptr || ((fuse->relu_mode == 0 || fuse->relu_mode == 2) && fuse->partial != nullptr)) &&
      use_n128(d, d->cin, d->cout, x2 != nullptr ? c2 : 0, a.Mg))
    return launch_n128<true>(a, (hipStream_t)stream);
  if (sub == nullptr && use_c64_dgrad(d, accumulate, fuse ? fuse->relu_mode : -1, bias != nullptr) && res_grad == nullptr)
    return launch_c64_conv(d, dy, wt, dx, fuse ? fuse->partial : nullptr, true, fuse, (hipStream_t)stream);
  if (sub == nullptr && use_r128_dgrad(d, accumulate, fuse ? fuse->relu_mode : -1, bias != nullptr) && res_grad == nullptr && x2 == nullptr &&
      f8 == nullptr && src == nullptr)
    return launch_r128_conv(d, dy, wt, dx, fuse ? fuse->partial : nullptr, true, fuse, (hipStream_t)stream);
  // 128 -> 128 3x3 / stride 2 (the stage-2 entry block's conv2), plain store: the four parity classes over ONE staged dy tile (conv3x3_ring.hip)
  if (sub == nullptr && fuse == nullptr && accumulate == 0 && bias == nullptr && res_grad == nullptr && x2 == nullptr && src == nullptr &&
      f8 == nullptr &&
      r128_s2dgrad_supported(d->dtype, d->cin, d->cout, d->r, d->s, d->stride, d->pad, d->h, d->w, d->ho, d->wo,
                             (long long)d->n * (d->ho + 1) * (d->wo + 1))) {
    R128Args c;
    c.x = (const bf16_t*)dy; c.w = (const bf16_t*)wt; c.out = (bf16_t*)dx; c.partial = nullptr;
    c.fy = nullptr; c.fscale = c.fshift = nullptr; c.relu = 0;
    c.in_scale = c.in_shift = nullptr; c.a_out = nullptr;
    c.N = d->n; c.H = d->ho; c.W = d->wo; c.dgrad = 1;
    c.q_total = (long long)d->n * (d->ho + 1) * (d->wo + 1);
    c.tiles = 0;
    c.div_pp = make_fastdiv((unsigned)((d->ho + 1) * (d->wo + 1)));
    c.div_wp = make_fastdiv((unsigned)(d->wo + 1));
    launch_r128_s2dgrad(c, (hipStream_t)stream);   // (SH_ROUTE_DGRAD_PARITY was counted above: still four parity classes, one launch)
    return check_launch("conv2d_dgrad (3x3 / stride 2 ring)");
  }
  if (f8 != nullptr) {
    a.x_state = f8->dy_state;
    a.w_state = f8->w_state;
    if (launch_igemm256_fp8_dgrad(a, (hipStream_t)stream)) return 1;
    return sub != nullptr ? sub_fallback() : 0;
  }
  if (use_256_dgrad(d, a.Mg)) {
    // the 256 x 256 kernel merges the shortcut's dense gradient in its masked-store epilogue (1x1 / stride 1, no parity classes)
    const bool merged = sub != nullptr && d->stride == 1 && d->r == 1 && d->s == 1 && a.fmode == 4 && a.fpartial == nullptr && a.Mg < (1ll << 31);
    if (merged) a.sub = sub;
    if (launch_igemm256<true>(a, (hipStream_t)stream)) return 1;
    return sub != nullptr && !merged ? sub_fallback() : 0;
  }
  if (d->dtype == SH_F32 ? launch_igemm<float, true>(a, (hipStream_t)stream) : launch_igemm<bf16_t, true>(a, (hipStream_t)stream)) return 1;
  return sub != nullptr ? sub_fallback() : 0;
}

int simhand_conv2d_dgrad_fp8_pays(const sh_conv_desc* d) { return d != nullptr && check_desc(d, "conv2d_dgrad_fp8_pays") == 0 && dgrad_fp8_ok(d) ? 1 : 0; }

int simhand_conv2d_dgrad(const sh_conv_desc* d, const void* dy, const void* wt, void* dx, int accumulate, sh_stream_t stream) {
  return dgrad_impl(d, dy, wt, dx, accumulate ? 1 : 0, nullptr, nullptr, stream);
}

int simhand_conv2d_dgrad_masked_residual(const sh_conv_desc* d, const void* dy, const void* wt, void* dx, const void* res_grad,
                                         const uint8_t* res_mask, sh_stream_t stream) {
  SH_REQUIRE(res_grad && res_mask, "conv2d_dgrad_masked_residual: NULL pointer");
  return dgrad_impl(d, dy, wt, dx, 2, res_grad, res_mask, stream);
}

// 1 when fusing the previous unit's BatchNorm-backward sums into this data gradient is the faster choice: the
// tile kernel hides the extra read of y behind its other resident blocks; the activation-stationary short-K 1x1
// kernel would pay the read's latency once per 64-channel chunk (measured 2-3x slower), so those layers keep the
// standalone simhand_bn_bwd_partial pass unless forced (simhand_test_conv2d_dgrad_fuse_1x1).
int simhand_test_conv2d_dgrad_fuse_1x1(int on) {
  g_fuse_1x1 = on ? 1 : 0;
  return 0;
}
int simhand_conv2d_dgrad_fuse_pays(const sh_conv_desc* d) {
  if (!d) return 0;
  // stride-2 3x3: each of the four parity-class launches pays the extra read of y against a quarter of the MFMA work; the
  // standalone pass is faster (same-box A/B of the whole step: 126.4 -> 125.9 ms)
  const int fuse_s2 = sw(SH_SW_FUSE_S2);  // A/B timing
  // (SH_SW_FUSE_S2: 1 = all stride-2 3x3 layers, 2 = only those on the 256 x 256 kernel -- round-3 A/B)
  if (d->stride == 2 && d->r == 3 && !g_fuse_1x1 && !(fuse_s2 == 1 || (fuse_s2 == 2 && d->cin >= 256 && d->cout >= 256))) return 0;
  return (!use_1x1(d, d->cout, d->cin) || g_fuse_1x1) ? 1 : 0;
}

int simhand_conv2d_dgrad_stat_blocks(const sh_conv_desc* d, int accumulate, int relu_mode, int c2) {
  if (!d) return 0;
  return dgrad_stat_blocks(d, accumulate, relu_mode, c2);
}

int simhand_conv2d_dgrad_fused(const sh_conv_desc* d, const void* dy, const void* wt, void* dx, int accumulate, const void* res_grad,
                               const uint8_t* res_mask, const sh_bn_bwd_fuse* fuse, sh_stream_t stream) {
  SH_REQUIRE(accumulate >= 0 && accumulate <= 2, "conv2d_dgrad_fused: accumulate mode %d", accumulate);
  SH_REQUIRE(accumulate != 2 || (res_grad && res_mask), "conv2d_dgrad_fused: accumulate 2 needs res_grad / res_mask");
  return dgrad_impl(d, dy, wt, dx, accumulate, res_grad, res_mask, stream, fuse);
}

int simhand_conv2d_dgrad_ex(const sh_conv_desc* d, const void* dy, const void* wt, void* dx, const sh_dgrad_opts* o, sh_stream_t stream) {
  SH_REQUIRE(o != nullptr, "conv2d_dgrad_ex: opts is NULL");
  SH_REQUIRE(o->accumulate >= 0 && o->accumulate <= 2, "conv2d_dgrad_ex: accumulate mode %d", o->accumulate);
  SH_REQUIRE(o->accumulate != 2 || (o->res_grad && o->res_mask), "conv2d_dgrad_ex: accumulate 2 needs res_grad / res_mask");
  return dgrad_impl(d, dy, wt, dx, o->accumulate, o->res_grad, o->res_mask, stream, o->fuse, o->bias, o->x2, o->wt2, o->c2, o->dy_src, o);
}

// the 1x1 layers of the activation-stationary kernel; the 128 -> 128 3x3 / stride-1 layers of the ring kernel (store-only and fused-sums forms)
int simhand_conv2d_dgrad_dysrc_ok(const sh_conv_desc* d) { return d != nullptr && use_1x1(d, d->cout, d->cin) ? 1 : 0; }

int simhand_conv2d_dgrad_concat_ok(const sh_conv_desc* d, int c2) {
  if (!d) return 0;
  return concat_ok(d, c2) ? 1 : 0;
}

}  // extern "C"
