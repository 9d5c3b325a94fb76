// Activation-stationary GEMM for the short-K 1x1 convolutions of the bottleneck blocks (bf16, stride 1):
//   Out[m][n] = sum_k A[m][k] * W[n][k],   K in {64, 128, 256},  N a multiple of 64.
//
// Replaces (reference): the 1x1 nn.Conv2d forward / input-gradient of torchvision's Bottleneck inside
// src/models/resnet_model.py:13-58 (cuDNN there).  Forward: A = x (K = cin), W = KRSC weights, N = cout (the 4x
// expansions 64->256 ... 256->1024 and the reductions with cin <= 256); dgrad: A = dy (K = cout), W = CRSK, N = cin.
//
// These layers are HBM-bound (51-170 FLOP/B against a ~400 FLOP/B ridge) and write up to 4x what they read, so the
// generic 128x128 tile kernel -- one prologue / k-loop / epilogue latency chain per 32 KB of output -- runs at a
// third of the HBM roof on them.  Here a block owns 64*MF pixel rows for ALL output channels:
//   * its A rows are loaded ONCE, straight from global memory into the MFMA operand registers (a lane's 16-B k-slice
//     is contiguous in the NHWC row): no LDS traffic and no re-reads for A, every A byte crosses HBM exactly once;
//   * the weights (<= 512 KB, L2-resident) stream through a double-buffered, XOR-swizzled LDS tile of 64 channels x
//     128 k, one barrier per step, the next tile's global loads in flight under the MFMAs;
//   * every 64 output channels the accumulators leave through 16-B register stores (64 contiguous bytes per pixel
//     and store instruction) while the other waves keep the matrix cores busy -- there is no separate epilogue phase.
// MFMA operands are swapped and the weight rows permuted exactly as in conv_igemm.hip, so a lane owns 8 consecutive
// channels of a pixel per tile pair; BatchNorm partial sums use DPP row reductions + a double-buffered LDS slab.
#include "conv_1x1.h"

#include <stdlib.h>

namespace sh {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16_frag_t;

__device__ __forceinline__ f32x4 mma_bf16(const uint4& a, const uint4& b, f32x4 c) {
  return sh_mfma16(a, b, c);
}

__device__ uint4 g_g1_zero_page[8];  // 128 B of zeros: the PF == 3 lanes of odd pixels fetch it instead of being masked

__device__ __forceinline__ float row16_sum_g1(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));  // row_ror:8
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));  // row_ror:4
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));  // row_ror:2
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));  // row_ror:1
  return v;
}


__device__ __forceinline__ unsigned add_bf16x2_g1(unsigned a, unsigned b) {
  const float lo = h16_lo(a) + h16_lo(b);
  const float hi = h16_hi(a) + h16_hi(b);
  return pack_bf16x2(lo, hi);
}

// DGRAD selects the epilogue: forward = BatchNorm partial sums + plain store; data gradient = the accumulate modes
// FUSE (data gradient only): also emit the previous unit's BatchNorm-backward partial sums (Gemm1x1Args::fy ...)
// EP (forward only): BatchNorm + residual + ReLU epilogue (Gemm1x1Args::ep_*): scale / shift are cached in LDS and the
// residual chunk of the NEXT 64 channels is prefetched into registers while the current chunk's MFMAs run -- otherwise
// every chunk would expose a global-load latency (measured 1.6x the HBM-bound time).
// EP == 2: the same with residual, ReLU and bit mask all present and M a multiple of the block's rows (the conv3 of every
// identity block): no conditional anywhere near a global load or store, see the note on vmcnt below.
// EP == 3: BatchNorm scale / shift ONLY (no residual, no ReLU, no mask: the stage-entry shortcuts), whole blocks, branch-free: the generic
// EP == 1 form took 2.1 ms for 64 -> 256 @ 56^2 (4.1 GB: 0.75 ms of HBM time) -- its conditional residual fetch inside the loop made the
// waitcnt pass drain the queue once per chunk (see "vmcnt" below) although no residual exists on this layer.
// PF == 2: the same without a residual gradient (first block of a stage: masked store only, accumulate 0).
// PF == 3: PF == 2 + the stride-2 shortcut's dense data gradient (Gemm1x1Args::sub) added at the even pixels before the mask: each row's
// pixel is decoded once, its 16-B chunks ride with the chunk's mask bytes (odd pixels fetch a zero page: no branch near a load).
// PF == 1 (data gradient; accumulate 2, masked store, full blocks): residual-gradient rows and both bit masks of the chunk are
// requested when the chunk's MFMAs start (masks as one 8-byte load per pixel row instead of two byte loads per mask).
//
// vmcnt: vector-memory operations retire IN ORDER, and hipcc's waitcnt pass takes the most conservative pending state over all
// paths that reach a point.  A conditional load or store in the loop body therefore turns the counted waits for the weight tile
// (the youngest loads but for the epilogue's) into vmcnt(0): the block drains its queue once per chunk and the HBM latency of
// the prefetched rows is exposed instead of hidden under the MFMAs -- these kernels' MFMA time and HBM time ADDED UP.  The fast
// variants (EP == 2, PF) keep the body branch-free, request the epilogue's rows right AFTER the chunk's first weight-tile fetch
// (so that waiting for the tile does not wait for them), and the last step re-fetches tile 0 instead of skipping the fetch.
// ST (K == 256 plain forward): the direct 7x7 stem -- its eighth k-slice (filter row 7) is all zero weights: neither loaded nor multiplied.
// CH (EP == 2): the chained next conv1 (Gemm1x1Args::chain_*, N -> K channels), fed by every finished output chunk (2 k-slices x
// K / 16 extra MFMAs per 16-row group).  Its weights are used as panels [K rows][64 k] (columns 64 nc .. of chain_w) in the
// 128-B-row tile format: K == 64 (N <= 256) keeps all N / 64 panels in LDS for the block's life; K == 128 (N <= 512, MF == 1)
// streams panel nc + 1 through registers into a second LDS buffer while chunk nc runs.
template <int K, int MF, bool DGRAD, bool FUSE = false, int EP = 0, int PF = 0, bool CH = false, bool ST = false>
#ifndef SH_G1_FUSE_MINB
#define SH_G1_FUSE_MINB 3
#endif
// resident blocks per CU the register budget is set for: three for the FORWARD kernels with K <= 128 (their operand rows and
// accumulators leave room: 64 -> 256 @ 56^2 with the BN epilogue 1619 -> 1527 us, 128 -> 512 @ 28^2 896 -> 867), two elsewhere (the data
// gradients and K = 256 spill at three: 790 -> 1347 us, 532 -> 1268 us)
#ifndef SH_G1_FWD_MINB
#define SH_G1_FWD_MINB 3
#endif
#ifndef SH_G1_CH_MINB
#define SH_G1_CH_MINB 2
#endif
// (the 256-row K = 128 form of the masked + merged data gradient exists for the block-size tuning hook only -- simhand_test_conv1x1_set_rows; with its block's
// mask rows in LDS it fits once per CU, and says so instead of leaving the compiler to miss a target of two)
__global__ __launch_bounds__(256, FUSE ? SH_G1_FUSE_MINB : ((!DGRAD && K <= 128 && !ST) ? (CH ? SH_G1_CH_MINB : SH_G1_FWD_MINB) : ((DGRAD && K == 128 && MF == 4 && PF == 1) ? 1 : 2))) void gemm1x1_kernel(Gemm1x1Args p) {
  constexpr int KC = K < 128 ? K : 128;   // k elements per weight tile
  constexpr int KSTEPS = K / KC;          // weight tiles per 64-channel chunk
  constexpr int KK = KC / 32;             // MFMA k-steps per weight tile
  constexpr int KF = K / 32;              // A fragments per 16-row group
  constexpr int CPR = KC / 8;             // 16-B chunks per weight row
  constexpr int NBL = 64 * CPR / 256;     // staged chunks per thread and step
  constexpr int ROWB = KC * 2;            // bytes per weight-tile row
  constexpr int BT = 64 * ROWB;           // bytes per weight tile
  constexpr bool FAST = EP == 2 || EP == 3 || PF != 0;
  static_assert(!CH || (EP == 2 && (K == 64 || (K == 128 && MF == 1))), "the chained conv1 exists for the K = 64 / 128 fast forward variants");
  constexpr int CT = K / 16;              // chained conv1: 16-channel output tiles
  constexpr int PB = K * 128;             // bytes of one chain panel [K][64]
  constexpr int EPN = CH ? 512 : (K == 128 ? 1024 : 2048);  // channels the epilogue coefficient cache holds (K = 128: three resident
                                                            // blocks of 32 KB weight tiles + the 8-KB store transpose leave 8 KB)
  __shared__ __attribute__((aligned(16))) char sC[CH ? (K == 64 ? 4 * PB : 2 * PB) : 16];  // K == 64: all panels; K == 128: two buffers
  __shared__ __attribute__((aligned(16))) char sB[2 * BT];
  __shared__ float red[2][4][2][64];
  __shared__ __attribute__((aligned(16))) float s_ep[EP ? 2 * EPN : 4];  // [2][N]: scale, shift
  // linear output stores (p.lt): a 16-pixel group's 64-channel chunk is 16 lines of 128 B; from the accumulator layout (lane = pixel li,
  // 16-B chunks g and 4 + g) a store instruction writes 64 separate 16-B pieces -- adjacent lanes are 128..N*2 bytes apart, nothing
  // coalesces.  Through a wave-private 2-KB LDS block (chunks XOR-swizzled by the pixel: conflict-free both ways) adjacent lanes hold
  // adjacent chunks: lane l stores chunk l & 7 of pixel l >> 3 (+ 8), eight full lines per instruction (stem_ring.hip: 1.15 -> 0.94 ms)
  __shared__ __attribute__((aligned(16))) char sT[FUSE ? 16 : 4 * 2048];

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int li = lane & 15, g = lane >> 4;
  const long long mbase = (long long)blockIdx.x * (64 * MF) + wave * (16 * MF);
  const bool lt = !FUSE && p.lt != 0;
  char* tw = sT + (FUSE ? 0 : wave * 2048);
  const int tw0 = li * 128 + (((0 * 4 + g) ^ (li & 7)) * 16), tw1 = li * 128 + (((1 * 4 + g) ^ (li & 7)) * 16);
  const int tr0 = (lane >> 3) * 128 + (((lane & 7) ^ ((lane >> 3) & 7)) * 16);
  const int nch = p.N >> 6, total = nch * KSTEPS;

  // swizzle key of a weight-tile row (rows are read 8q + t apart, q, t = 0..3: see conv_igemm.hip)
  auto keyb = [](int row) __attribute__((always_inline)) -> int {
    return CPR == 16 ? ((row & 3) | (((row >> 3) & 3) << 2)) : (((row >> 1) & 1) | (((row >> 3) & 3) << 1));
  };

  // ---- staging map of this thread: chunks tid + 256 i of the [64][CPR] tile = rows row0 + i*RSTEP, same chunk ---------
  // (named scalars, not arrays: hipcc 7.2 sends small per-thread arrays that live across the loop to scratch)
  constexpr int RSTEP = 256 / CPR;
  const int row0 = tid / CPR, ch0 = tid - row0 * CPR;
  const int st0 = (row0 + 0 * RSTEP) * ROWB + ((ch0 ^ keyb(row0 + 0 * RSTEP)) * 16);
  const int st1 = (row0 + 1 * RSTEP) * ROWB + ((ch0 ^ keyb(row0 + 1 * RSTEP)) * 16);
  const int st2 = (row0 + 2 * RSTEP) * ROWB + ((ch0 ^ keyb(row0 + 2 * RSTEP)) * 16);
  const int st3 = (row0 + 3 * RSTEP) * ROWB + ((ch0 ^ keyb(row0 + 3 * RSTEP)) * 16);
  const bf16_t* wsrc0 = p.w + (long long)row0 * K + ch0 * 8;  // chunk i at + i * RSTEP * K elements
  uint4 rb0, rb1, rb2 = make_uint4(0, 0, 0, 0), rb3 = make_uint4(0, 0, 0, 0);
#define SH_G1_LOAD(OFF)                                                                         \
  do {                                                                                          \
    rb0 = *reinterpret_cast<const uint4*>(wsrc0 + (OFF));                                       \
    rb1 = *reinterpret_cast<const uint4*>(wsrc0 + (OFF) + 1 * RSTEP * K);                       \
    if (NBL == 4) {                                                                             \
      rb2 = *reinterpret_cast<const uint4*>(wsrc0 + (OFF) + 2 * RSTEP * K);                     \
      rb3 = *reinterpret_cast<const uint4*>(wsrc0 + (OFF) + 3 * RSTEP * K);                     \
    }                                                                                           \
  } while (0)
#define SH_G1_STORE(DST)                                  \
  do {                                                    \
    *reinterpret_cast<uint4*>((DST) + st0) = rb0;         \
    *reinterpret_cast<uint4*>((DST) + st1) = rb1;         \
    if (NBL == 4) {                                       \
      *reinterpret_cast<uint4*>((DST) + st2) = rb2;       \
      *reinterpret_cast<uint4*>((DST) + st3) = rb3;       \
    }                                                     \
  } while (0)
  SH_G1_LOAD(0);
  // W2 (fast variants, K = 256: two weight tiles per chunk): a SECOND staging set, so that every tile is fetched TWO k-steps before
  // its use -- the counted wait at the end of a k-step then finds its tile landed long ago instead of exposing an L2 round trip per
  // step (with one set the fetch is issued at the start of the step that ends by waiting for it).  Even tiles travel in rb, odd in rc.
  constexpr bool W2 = FAST && KSTEPS == 2;
  uint4 rc0 = make_uint4(0, 0, 0, 0), rc1 = rc0, rc2 = rc0, rc3 = rc0;
#define SH_G1_LOAD2(OFF)                                                                        \
  do {                                                                                          \
    rc0 = *reinterpret_cast<const uint4*>(wsrc0 + (OFF));                                       \
    rc1 = *reinterpret_cast<const uint4*>(wsrc0 + (OFF) + 1 * RSTEP * K);                       \
    if (NBL == 4) {                                                                             \
      rc2 = *reinterpret_cast<const uint4*>(wsrc0 + (OFF) + 2 * RSTEP * K);                     \
      rc3 = *reinterpret_cast<const uint4*>(wsrc0 + (OFF) + 3 * RSTEP * K);                     \
    }                                                                                           \
  } while (0)
#define SH_G1_STORE2(DST)                                 \
  do {                                                    \
    *reinterpret_cast<uint4*>((DST) + st0) = rc0;         \
    *reinterpret_cast<uint4*>((DST) + st1) = rc1;         \
    if (NBL == 4) {                                       \
      *reinterpret_cast<uint4*>((DST) + st2) = rc2;       \
      *reinterpret_cast<uint4*>((DST) + st3) = rc3;       \
    }                                                     \
  } while (0)
  if constexpr (W2) SH_G1_LOAD2(KC);  // tile 1 = (chunk 0, second k-step)

  // ---- A: this wave's 16*MF rows, whole K, straight into MFMA operand registers ----------------------------------------
  uint4 afr[MF][KF];
#pragma unroll
  for (int mi = 0; mi < MF; ++mi) {
    const long long row = mbase + mi * 16 + li;
    const bool ok = FAST || row < p.M;  // the fast variants only run on whole blocks: no branch per row
    const bf16_t* pr = p.a + (ok ? row : 0) * K + g * 8;
    int jstride = 32;
    if constexpr (K == 256 && !DGRAD && !FUSE && !EP) {
      if (p.stem_wp > 0) {  // direct stem: k-slice j = filter row j of the padded NHWC4 input (see Gemm1x1Args)
        const unsigned m = ok ? (unsigned)row : 0u;
        const unsigned img = fdiv(m, p.div_hw);
        const unsigned rem = m - img * p.div_hw.d;
        const unsigned ho = fdiv(rem, p.div_w);
        const unsigned wo = rem - ho * p.div_w.d;
        pr = p.a + (((long long)img * p.stem_hp + 2 * ho) * p.stem_wp + 2 * wo) * 4 + g * 8;
        jstride = p.stem_wp * 4;
      }
    }
    if (!ST && !FUSE && p.stem_wp == 0) {
      // line-shaped loads: lane l fetches chunk l & 7 of rows l >> 3 and 8 + (l >> 3) of every 128-B column; transposed below
      const long long ra = mbase + mi * 16 + (lane >> 3), rb = ra + 8;
      const bool oka = FAST || ra < p.M, okb = FAST || rb < p.M;
#pragma unroll
      for (int c4 = 0; c4 < K / 64; ++c4) {
        afr[mi][2 * c4] = oka ? *reinterpret_cast<const uint4*>(p.a + ra * K + c4 * 64 + (lane & 7) * 8) : make_uint4(0, 0, 0, 0);
        afr[mi][2 * c4 + 1] = okb ? *reinterpret_cast<const uint4*>(p.a + rb * K + c4 * 64 + (lane & 7) * 8) : make_uint4(0, 0, 0, 0);
      }
      continue;
    }
#pragma unroll
    for (int j = 0; j < (ST ? KF - 1 : KF); ++j) afr[mi][j] = ok ? *reinterpret_cast<const uint4*>(pr + j * jstride) : make_uint4(0, 0, 0, 0);
  }
  if constexpr (!ST && !FUSE) {
    if (p.stem_wp == 0) {  // wave-private transpose: (row l >> 3 (+ 8), chunk l & 7) -> (row li, chunks g and 4 + g): the operand shape
#pragma unroll
      for (int mi = 0; mi < MF; ++mi)
#pragma unroll
        for (int c4 = 0; c4 < K / 64; ++c4) {
          *reinterpret_cast<uint4*>(tw + tr0) = afr[mi][2 * c4];
          *reinterpret_cast<uint4*>(tw + tr0 + 1024) = afr[mi][2 * c4 + 1];
          afr[mi][2 * c4] = *reinterpret_cast<const uint4*>(tw + tw0);
          afr[mi][2 * c4 + 1] = *reinterpret_cast<const uint4*>(tw + tw1);
        }
    }
  }
  if constexpr (DGRAD) {
    if (p.xf_y != nullptr) {  // block-uniform, prologue only: BatchNorm-backward apply on the freshly loaded gradient rows
      // all y rows are requested before the first one is used (one exposed round trip, not MF * KF of them); the per-channel
      // coefficients depend on the k-slice only and are fetched once per slice
      uint4 yfr[MF][KF];
#pragma unroll
      for (int mi = 0; mi < MF; ++mi) {
        if constexpr (!FUSE) {  // line-shaped (rows l >> 3 and + 8, chunk l & 7 of every 128-B column), transposed below
          const long long ra = mbase + mi * 16 + (lane >> 3), rb = ra + 8;
#pragma unroll
          for (int c4 = 0; c4 < K / 64; ++c4) {
            yfr[mi][2 * c4] = *reinterpret_cast<const uint4*>(p.xf_y + (ra < p.M ? ra : 0) * K + c4 * 64 + (lane & 7) * 8);
            yfr[mi][2 * c4 + 1] = *reinterpret_cast<const uint4*>(p.xf_y + (rb < p.M ? rb : 0) * K + c4 * 64 + (lane & 7) * 8);
          }
        } else {
          const long long row = mbase + mi * 16 + li;
          const bf16_t* py = p.xf_y + (row < p.M ? row : 0) * K + g * 8;
#pragma unroll
          for (int j = 0; j < KF; ++j) yfr[mi][j] = *reinterpret_cast<const uint4*>(py + j * 32);
        }
      }
      if constexpr (!FUSE) {
#pragma unroll
        for (int mi = 0; mi < MF; ++mi)
#pragma unroll
          for (int c4 = 0; c4 < K / 64; ++c4) {
            *reinterpret_cast<uint4*>(tw + tr0) = yfr[mi][2 * c4];
            *reinterpret_cast<uint4*>(tw + tr0 + 1024) = yfr[mi][2 * c4 + 1];
            yfr[mi][2 * c4] = *reinterpret_cast<const uint4*>(tw + tw0);
            yfr[mi][2 * c4 + 1] = *reinterpret_cast<const uint4*>(tw + tw1);
          }
      }
      // the five coefficient vectors through LDS (the weight-tile buffers are idle until the first SH_G1_STORE): one global load per
      // thread and vector instead of ten float4 loads per lane and k-slice
      float* s_xf = reinterpret_cast<float*>(sB);  // [5][K]: 5 KB at K = 256 <= 2 * BT
      for (int i = tid; i < K; i += 256) {
        s_xf[i] = p.xf_s[i];
        s_xf[K + i] = p.xf_h[i];
        s_xf[2 * K + i] = p.xf_a[i];
        s_xf[3 * K + i] = p.xf_b[i];
        s_xf[4 * K + i] = p.xf_c[i];
      }
      __syncthreads();
#pragma unroll
      for (int j = 0; j < KF; ++j) {
        const int kc = j * 32 + g * 8;
        float cs[8], ch[8], ca[8], cb[8], cc[8];
#pragma unroll
        for (int e = 0; e < 8; e += 4) {
          const float4 t0 = *reinterpret_cast<const float4*>(s_xf + kc + e), t1 = *reinterpret_cast<const float4*>(s_xf + K + kc + e);
          const float4 t2 = *reinterpret_cast<const float4*>(s_xf + 2 * K + kc + e), t3 = *reinterpret_cast<const float4*>(s_xf + 3 * K + kc + e);
          const float4 t4 = *reinterpret_cast<const float4*>(s_xf + 4 * K + kc + e);
          cs[e] = t0.x; cs[e + 1] = t0.y; cs[e + 2] = t0.z; cs[e + 3] = t0.w;
          ch[e] = t1.x; ch[e + 1] = t1.y; ch[e + 2] = t1.z; ch[e + 3] = t1.w;
          ca[e] = t2.x; ca[e + 1] = t2.y; ca[e + 2] = t2.z; ca[e + 3] = t2.w;
          cb[e] = t3.x; cb[e + 1] = t3.y; cb[e + 2] = t3.z; cb[e + 3] = t3.w;
          cc[e] = t4.x; cc[e + 1] = t4.y; cc[e + 2] = t4.z; cc[e + 3] = t4.w;
        }
#pragma unroll
        for (int mi = 0; mi < MF; ++mi) {
          const long long row = mbase + mi * 16 + li;
          const bool ok = row < p.M;
          const unsigned g4[4] = {afr[mi][j].x, afr[mi][j].y, afr[mi][j].z, afr[mi][j].w};
          const unsigned y4[4] = {yfr[mi][j].x, yfr[mi][j].y, yfr[mi][j].z, yfr[mi][j].w};
          unsigned o4[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            float r2[2];
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
              const int e = 2 * q + hh;
              const float gv = hh == 0 ? h16_lo(g4[q]) : h16_hi(g4[q]);
              const float yy = hh == 0 ? h16_lo(y4[q]) : h16_hi(y4[q]);
              const bool on = !p.xf_relu || (yy * cs[e] + ch[e] > 0.f);
              r2[hh] = ca[e] * (on ? gv : 0.f) - cb[e] * yy + cc[e];  // same expression as the weight-gradient loader's (XFORM 2)
            }
            o4[q] = pack_bf16x2(r2[0], r2[1]);
          }
          afr[mi][j] = ok ? make_uint4(o4[0], o4[1], o4[2], o4[3]) : make_uint4(0, 0, 0, 0);
          if (FUSE && ok) *reinterpret_cast<uint4*>(p.xf_out + row * K + g * 8 + j * 32) = afr[mi][j];
        }
      }
      if constexpr (!FUSE) {  // dy leaves line-shaped as well (the weight gradient that follows reads it)
#pragma unroll
        for (int mi = 0; mi < MF; ++mi) {
          const long long ra = mbase + mi * 16 + (lane >> 3), rb = ra + 8;
#pragma unroll
          for (int c4 = 0; c4 < K / 64; ++c4) {
            *reinterpret_cast<uint4*>(tw + tw0) = afr[mi][2 * c4];
            *reinterpret_cast<uint4*>(tw + tw1) = afr[mi][2 * c4 + 1];
            const uint4 r0 = *reinterpret_cast<const uint4*>(tw + tr0), r1 = *reinterpret_cast<const uint4*>(tw + tr0 + 1024);
            if (ra < p.M) *reinterpret_cast<uint4*>(p.xf_out + ra * K + c4 * 64 + (lane & 7) * 8) = r0;
            if (rb < p.M) *reinterpret_cast<uint4*>(p.xf_out + rb * K + c4 * 64 + (lane & 7) * 8) = r1;
          }
        }
      }
      __syncthreads();  // every wave is done with the coefficients before the first weight tile overwrites them
    }
  }

  // ---- EP: coefficients to LDS (visible after the first barrier below), residual rows of chunk 0 to registers -------
  uint4 rq[EP ? MF : 1][2];
  auto load_res = [&](int nc2) __attribute__((always_inline)) {
#pragma unroll
    for (int mi = 0; mi < (EP ? MF : 1); ++mi) {
      const long long row = mbase + mi * 16 + li;
      const long long ra = mbase + mi * 16 + (lane >> 3), rb = ra + 8;  // line-shaped: rows l >> 3 and + 8, chunk l & 7
      // non-temporal (round 6): the residual is the block INPUT on its last forward use and 3.3 GB wide at stage 1 -- streaming it past the
      // caches (and the output below) leaves L2 to the weight tiles every block re-reads: -0.6 ms per step, three same-box pairs
      // (profiles/r06_cache_policy_ab.txt; the same hint on the data gradient's residual-gradient rows and stores costs +0.9 ms: those tensors are
      // re-read by the very next launches)
      rq[mi][0] = ld16<true>(p.ep_res + (FAST || ra < p.M ? ra : 0) * p.N + nc2 * 64 + (lane & 7) * 8);
      rq[mi][1] = ld16<true>(p.ep_res + (FAST || rb < p.M ? rb : 0) * p.N + nc2 * 64 + (lane & 7) * 8);
    }
  };
  if constexpr (EP) {
    for (int i = tid; i < p.N; i += 256) {
      s_ep[i] = p.ep_scale[i];
      s_ep[EPN + i] = p.ep_shift[i];
    }
    if (EP == 1 && p.ep_res != nullptr) load_res(0);
  }

  // ---- PF: residual gradient + masks of chunk 0
  // Round 6: the block's rows of the 1-bit masks are staged in LDS ONCE.  A mask row is N / 8 bytes (128 at N = 1024) of which every 64-channel
  // chunk uses 8: read from global memory chunk by chunk, a row's line had to survive in L2 for the whole life of the block -- under this
  // kernel's own streams it did not, and rocprofv3 counted 1.75 GB of HBM reads per launch where the operands are 1.13 GB (1024 <- 256 @ 14^2,
  // 2048 images: profiles/r06_g1_dgrad_traffic.txt).  16-B slots XOR-swizzled by the row: the 16 rows a read touches spread over the banks.
  constexpr int MROWS = 64 * MF, MRB = K == 64 ? 32 : 128;   // staged rows, bytes per staged row (N <= 256 at K = 64, <= 1024 otherwise: launch_gemm1x1)
  __shared__ __attribute__((aligned(16))) char sMk[PF ? (PF == 1 ? 2 : 1) * MROWS * MRB : 16];
  const int mnb = p.N >> 3, mnv = mnb >> 4;                  // bytes / 16-B vectors per mask row
  if constexpr (PF != 0) {
    const long long brow = (long long)blockIdx.x * MROWS;
    for (int v = tid; v < MROWS * mnv; v += 256) {
      const int row = v / mnv, c = v - row * mnv;
      const int dst = row * MRB + ((c ^ (row & (mnv - 1))) << 4);
      *reinterpret_cast<uint4*>(sMk + dst) = *reinterpret_cast<const uint4*>(p.fmask + (brow + row) * mnb + c * 16);
      if constexpr (PF == 1)
        *reinterpret_cast<uint4*>(sMk + MROWS * MRB + dst) = *reinterpret_cast<const uint4*>(p.res_mask + (brow + row) * mnb + c * 16);
    }
  }
  uint4 pg[PF ? MF : 1][2];
  uint2 pm[PF ? MF : 1], pk[PF ? MF : 1];
  // PF == 3: the shortcut gradient `sub` lives on the EVEN pixels.  A 16-pixel group starts on an even column (M, W even), so its eight even
  // pixels are eight rows of `sub` -- fetched LINE-shaped by one instruction per chunk (lane l: even pixel l >> 3 of the group, 16-B chunk
  // l & 7 of the 64-channel chunk; pixels on odd image rows: the zero page) and turned into the accumulator shape through the wave-private
  // LDS block in the epilogue.  (First form: two accumulator-shaped loads per chunk with three lanes in four on the zero page -- the merged
  // launch took as long as the unmerged one plus its scatter-add pass.)
  const bf16_t* sp[PF == 3 ? MF : 1];
  if constexpr (PF == 3) {
#pragma unroll
    for (int mi = 0; mi < MF; ++mi) {
      const unsigned ru = (unsigned)(mbase + mi * 16 + 2 * (lane >> 3));  // M < 2^31; full blocks; an even pixel of the row grid
      const unsigned img = fdiv(ru, p.div_hw);
      const unsigned rem = ru - img * p.div_hw.d;
      const unsigned hh = fdiv(rem, p.div_w);
      const unsigned ww = rem - hh * p.div_w.d;
      const unsigned srow = (img * (unsigned)(p.sub_h >> 1) + (hh >> 1)) * (unsigned)(p.sub_w >> 1) + (ww >> 1);
      sp[mi] = (hh & 1u) == 0 ? p.sub + (unsigned long long)srow * p.N + (lane & 7) * 8 : nullptr;
    }
  }
  auto load_pf = [&](int nc2) __attribute__((always_inline)) {
#pragma unroll
    for (int mi = 0; mi < (PF ? MF : 1); ++mi) {
      const long long r = mbase + mi * 16 + li;
      if constexpr (PF == 3) {
        const bf16_t* q = sp[mi] != nullptr ? sp[mi] + nc2 * 64 : reinterpret_cast<const bf16_t*>(g_g1_zero_page);
        pg[mi][0] = *reinterpret_cast<const uint4*>(q);
      }
      if constexpr (PF == 1) {
        const long long ra = mbase + mi * 16 + (lane >> 3);  // line-shaped: rows l >> 3 and + 8, chunk l & 7 (whole blocks: every row exists)
        pg[mi][0] = *reinterpret_cast<const uint4*>(p.res_grad + ra * p.N + nc2 * 64 + (lane & 7) * 8);
        pg[mi][1] = *reinterpret_cast<const uint4*>(p.res_grad + (ra + 8) * p.N + nc2 * 64 + (lane & 7) * 8);
      }
      // the chunk's 8 mask bytes of this lane's pixel row, from the staged copy (row of the block = wave * 16 MF + mi * 16 + li)
      const int lrow = wave * (16 * MF) + mi * 16 + li;
      const int moff = lrow * MRB + ((((nc2 >> 1) ^ (lrow & (mnv - 1))) << 4) | ((nc2 & 1) << 3));
      if constexpr (PF == 1) pm[mi] = *reinterpret_cast<const uint2*>(sMk + MROWS * MRB + moff);
      pk[mi] = *reinterpret_cast<const uint2*>(sMk + moff);
    }
  };

  // ---- fragment read offsets: weight row chan_of(ni, li) = (ni>>1)*32 + (li>>2)*8 + (ni&1)*4 + (li&3); key == li -------
  const int fkey = CPR == 16 ? li : (((li & 3) >> 1) | ((li >> 2) << 1));
  int frow[4];
#pragma unroll
  for (int ni = 0; ni < 4; ++ni) frow[ni] = ((ni >> 1) * 32 + (li >> 2) * 8 + (ni & 1) * 4 + (li & 3)) * ROWB;
  int fo[KK];
#pragma unroll
  for (int kk = 0; kk < KK; ++kk) fo[kk] = ((kk * 4 + g) ^ fkey) * 16;

  f32x4 acc[MF][4];
#pragma unroll
  for (int mi = 0; mi < MF; ++mi)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};
  uint4 vout[CH ? MF : 1][2];  // CH: the chunk's packed bf16 output

  f32x4 cacc[CH ? MF : 1][CH ? CT : 1];
  // chain panel staging map (128-B rows: 8 chunks per row): chunks tid + 256 i = rows tid / 8 + 32 i, chunk tid % 8; key from row bits 1, 3, 4
  const int prow = tid >> 3, pch = tid & 7;
  const int pst = prow * 128 + ((pch ^ (((prow >> 1) & 1) | (((prow >> 3) & 3) << 1))) * 16);  // + i * 32 * 128 (the key ignores bits 5, 6)
  // fragment offsets in a panel: tile ct = channels (ct >> 2) * 64 + chan_of(ct & 3, li); k-slice kk = chunk 4 kk + g
  const int pkey = ((li & 3) >> 1) | ((li >> 2) << 1);
  const int pfo0 = (g ^ pkey) * 16, pfo1 = ((4 + g) ^ pkey) * 16;
  const int pfr0 = ((li >> 2) * 8 + (li & 3)) * 128;  // chan_of(0, li) rows; tile ct adds ((ct >> 2) * 64 + ((ct >> 1) & 1) * 32 + (ct & 1) * 4) * 128
  uint4 pr0, pr1, pr2, pr3;  // K == 128: the next panel on its way to LDS
  auto panel_load = [&](int pn) __attribute__((always_inline)) {
    const bf16_t* src = p.chain_w + (long long)prow * p.N + pn * 64 + pch * 8;
    pr0 = *reinterpret_cast<const uint4*>(src);
    pr1 = *reinterpret_cast<const uint4*>(src + 32ll * p.N);
    if constexpr (K == 128) {
      pr2 = *reinterpret_cast<const uint4*>(src + 64ll * p.N);
      pr3 = *reinterpret_cast<const uint4*>(src + 96ll * p.N);
    }
  };
  auto panel_store = [&](char* dst) __attribute__((always_inline)) {
    *reinterpret_cast<uint4*>(dst + pst) = pr0;
    *reinterpret_cast<uint4*>(dst + pst + 32 * 128) = pr1;
    if constexpr (K == 128) {
      *reinterpret_cast<uint4*>(dst + pst + 64 * 128) = pr2;
      *reinterpret_cast<uint4*>(dst + pst + 96 * 128) = pr3;
    }
  };
  if constexpr (CH) {
#pragma unroll
    for (int mi = 0; mi < MF; ++mi)
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) cacc[mi][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if constexpr (K == 64) {
      for (int pn = 0; pn < nch; ++pn) {
        panel_load(pn);
        panel_store(sC + pn * PB);
      }
    } else {
      panel_load(0);
      panel_store(sC);
    }
  }
  SH_G1_STORE(sB);
  // everything the prologue requested (A rows, coefficients) is waited for HERE, explicitly: a load still pending on the loop's
  // entry path would make every iteration drain the queue down to that path's count (see the note on vmcnt above)
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
  __syncthreads();

  int s = 0;
  for (int nc = 0; nc < nch; ++nc) {
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks, ++s) {
      const int buf = s & 1;
      const bool more = s + 1 < total;
      if constexpr (W2) {  // tile s + 2 = (nc + 1, ks); past the end tile 0 again (unused)
        const long long off2 = nc + 1 < nch ? (long long)(nc + 1) * 64 * K + ks * KC : 0ll;
        if (ks == 0) SH_G1_LOAD(off2);
        else SH_G1_LOAD2(off2);
      } else {  // next weight tile: (nc, ks + 1) or (nc + 1, 0); after the last one tile 0 again (unused)
        const long long off = !more ? 0ll : (ks + 1 < KSTEPS ? (long long)nc * 64 * K + (ks + 1) * KC : (long long)(nc + 1) * 64 * K);
        SH_G1_LOAD(off);
      }
      if constexpr (FAST) {  // this chunk's epilogue rows: younger than the tile fetch above, a whole chunk of MFMAs to land
        __builtin_amdgcn_sched_barrier(0);  // the machine scheduler would hoist them above the tile fetch
        if (ks == 0) {
          if constexpr (EP == 2) load_res(nc);
          else if constexpr (PF != 0) load_pf(nc);
        }
        if constexpr (CH && K == 128) panel_load(nc + 1 < nch ? nc + 1 : 0);  // lands under this chunk's MFMAs (last: panel 0 again, unused)
        __builtin_amdgcn_sched_barrier(0);
      }
      const char* cB = sB + buf * BT;
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) {
        if (ST && ks * KK + kk == KF - 1) continue;  // compile-time after unrolling: the stem's zero filter row
        uint4 fb[4];
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) fb[ni] = *reinterpret_cast<const uint4*>(cB + frow[ni] + fo[kk]);
#pragma unroll
        for (int mi = 0; mi < MF; ++mi)
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = mma_bf16(fb[ni], afr[mi][ks * KK + kk], acc[mi][ni]);
      }
      if (ks == KSTEPS - 1) {
        const int n0 = nc * 64;
        // ---- BatchNorm partial sums of the fp32 accumulators: lane holds pixel li of each 16-row group, channels
        //      chan_of(ni, 4g + r); rows beyond M are exact zeros
        if (!FAST && !DGRAD && p.bn_partial != nullptr) {
#pragma unroll
          for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float s1 = 0.f, s2 = 0.f;
#pragma unroll
              for (int mi = 0; mi < MF; ++mi) {
                const float v = acc[mi][ni][r];
                s1 += v;
                s2 += v * v;
              }
              s1 = row16_sum_g1(s1);
              s2 = row16_sum_g1(s2);
              if (li == 0) {
                const int c = (ni >> 1) * 32 + g * 8 + (ni & 1) * 4 + r;
                red[nc & 1][wave][0][c] = s1;
                red[nc & 1][wave][1][c] = s2;
              }
            }
        }
        // ---- 64 channels out: two 16-B vectors per lane and pixel (channels n0 + j*32 + g*8 .. +8)
        unsigned mb[2] = {0u, 0u};  // lt: the ReLU mask bytes of the lane's two chunks (stored 8 per pixel by the lanes g == 0)
        auto store_chunk = [&](int mi, int j, long long row, unsigned keep = 0xffu) __attribute__((always_inline)) -> uint4 {
          f32x4 lo = acc[mi][2 * j], hi = acc[mi][2 * j + 1];
          if (!FAST && DGRAD && p.bias != nullptr) {
            const int chb = n0 + j * 32 + g * 8;
            const float4 b0 = *reinterpret_cast<const float4*>(p.bias + chb), b1 = *reinterpret_cast<const float4*>(p.bias + chb + 4);
            lo[0] += b0.x; lo[1] += b0.y; lo[2] += b0.z; lo[3] += b0.w;
            hi[0] += b1.x; hi[1] += b1.y; hi[2] += b1.z; hi[3] += b1.w;
          }
          const int ch = n0 + j * 32 + g * 8;
          bf16_t* dst = p.out + row * p.N + ch;
          if constexpr (EP) {
            // BatchNorm (statistics known up front) + residual + ReLU on the fp32 accumulators; ReLU bit mask out
            const float4 s0 = *reinterpret_cast<const float4*>(s_ep + ch), s1 = *reinterpret_cast<const float4*>(s_ep + ch + 4);
            const float4 h0 = *reinterpret_cast<const float4*>(s_ep + EPN + ch), h1 = *reinterpret_cast<const float4*>(s_ep + EPN + ch + 4);
            float o[8] = {lo[0] * s0.x + h0.x, lo[1] * s0.y + h0.y, lo[2] * s0.z + h0.z, lo[3] * s0.w + h0.w,
                          hi[0] * s1.x + h1.x, hi[1] * s1.y + h1.y, hi[2] * s1.z + h1.z, hi[3] * s1.w + h1.w};
            if (EP == 2 || (EP == 1 && p.ep_res != nullptr)) {
              const unsigned w4[4] = {rq[mi][j].x, rq[mi][j].y, rq[mi][j].z, rq[mi][j].w};  // prefetched a chunk ago
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                o[2 * i] += h16_lo(w4[i]);
                o[2 * i + 1] += h16_hi(w4[i]);
              }
            }
            if (EP == 2 || (EP == 1 && p.ep_relu)) {
              unsigned bits = 0;
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                bits |= (o[e] > 0.f ? 1u : 0u) << e;
                o[e] = o[e] > 0.f ? o[e] : 0.f;
              }
              if (EP == 2 || p.ep_mask != nullptr) {
                if (lt) mb[j] = bits;
                else p.ep_mask[row * (p.N >> 3) + (ch >> 3)] = (unsigned char)bits;
              }
            }
            const uint4 pv = make_uint4(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]), pack_bf16x2(o[4], o[5]), pack_bf16x2(o[6], o[7]));
            if (!lt) *reinterpret_cast<uint4*>(dst) = pv;
            return pv;
          }
          uint4 v;
          v.x = pack_bf16x2(lo[0], lo[1]);
          v.y = pack_bf16x2(lo[2], lo[3]);
          v.z = pack_bf16x2(hi[0], hi[1]);
          v.w = pack_bf16x2(hi[2], hi[3]);
          if (DGRAD && (PF == 1 || (PF == 0 && p.accumulate == 2))) {
            uint4 o;
            unsigned bits;
            if constexpr (PF == 1) {
              o = pg[mi][j];
              bits = ((j == 0 ? pm[mi].x : pm[mi].y) >> (8 * g)) & 0xffu;
            } else {
              o = *reinterpret_cast<const uint4*>(p.res_grad + row * p.N + ch);
              bits = p.res_mask[row * (p.N >> 3) + (ch >> 3)];
            }
            const unsigned m0w = ((bits & 1u) ? 0x0000ffffu : 0u) | ((bits & 2u) ? 0xffff0000u : 0u);
            const unsigned m1w = ((bits & 4u) ? 0x0000ffffu : 0u) | ((bits & 8u) ? 0xffff0000u : 0u);
            const unsigned m2w = ((bits & 16u) ? 0x0000ffffu : 0u) | ((bits & 32u) ? 0xffff0000u : 0u);
            const unsigned m3w = ((bits & 64u) ? 0x0000ffffu : 0u) | ((bits & 128u) ? 0xffff0000u : 0u);
            v.x = add_bf16x2_g1(v.x, o.x & m0w);
            v.y = add_bf16x2_g1(v.y, o.y & m1w);
            v.z = add_bf16x2_g1(v.z, o.z & m2w);
            v.w = add_bf16x2_g1(v.w, o.w & m3w);
          } else if (DGRAD && PF == 3) {
            const uint4 o = pg[mi][j];
            v.x = add_bf16x2_g1(v.x, o.x);
            v.y = add_bf16x2_g1(v.y, o.y);
            v.z = add_bf16x2_g1(v.z, o.z);
            v.w = add_bf16x2_g1(v.w, o.w);
          } else if (DGRAD && PF == 0 && p.accumulate) {
            const uint4 o = *reinterpret_cast<const uint4*>(dst);
            v.x = add_bf16x2_g1(v.x, o.x);
            v.y = add_bf16x2_g1(v.y, o.y);
            v.z = add_bf16x2_g1(v.z, o.z);
            v.w = add_bf16x2_g1(v.w, o.w);
          }
          if (DGRAD && (PF || p.fmode == 4)) {  // the stored gradient is the masked one
            v.x &= ((keep & 1u) ? 0x0000ffffu : 0u) | ((keep & 2u) ? 0xffff0000u : 0u);
            v.y &= ((keep & 4u) ? 0x0000ffffu : 0u) | ((keep & 8u) ? 0xffff0000u : 0u);
            v.z &= ((keep & 16u) ? 0x0000ffffu : 0u) | ((keep & 32u) ? 0xffff0000u : 0u);
            v.w &= ((keep & 64u) ? 0x0000ffffu : 0u) | ((keep & 128u) ? 0xffff0000u : 0u);
          }
          if (!lt) *reinterpret_cast<uint4*>(dst) = v;
          return v;
        };
        if (FUSE) {
          // fused BatchNorm-backward partial sums of the previous unit: the stored gradient is its incoming gradient
          // da; g = da * relu'(.); per-channel sums of g and g * y over this block's rows (see conv_igemm.hip)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int ch = n0 + j * 32 + g * 8;
            float sc[8], sh[8], s1[8], s2[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              sc[e] = p.fmode == 2 ? p.fscale[ch + e] : 0.f;
              sh[e] = p.fmode == 2 ? p.fshift[ch + e] : 0.f;
              s1[e] = s2[e] = 0.f;
            }
#pragma unroll
            for (int mi = 0; mi < MF; ++mi) {
              const long long row = mbase + mi * 16 + li;
              if (row < p.M) {
                float yy[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) yy[e] = 0.f;
                if (p.fmode != 4) Vec16<bf16_t>::load(p.fy + row * p.N + ch, yy);
                unsigned bits = 0xffu;
                if (p.fmode >= 3) bits = p.fmask[row * (p.N >> 3) + (ch >> 3)];
                const uint4 v = store_chunk(mi, j, row, bits);
                const unsigned w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
#pragma unroll
                  for (int h = 0; h < 2; ++h) {
                    const int e = 2 * i + h;
                    const float gq = h == 0 ? h16_lo(w4[i]) : h16_hi(w4[i]);
                    bool on = true;
                    if (p.fmode == 2) on = yy[e] * sc[e] + sh[e] > 0.f;
                    else if (p.fmode == 3) on = (bits >> e) & 1u;  // mode 4: the value is already masked
                    const float gv = on ? gq : 0.f;
                    s1[e] += gv;
                    s2[e] += gv * yy[e];
                  }
                }
              }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float t1 = row16_sum_g1(s1[e]), t2 = row16_sum_g1(s2[e]);
              if (li == 0) {
                const int c = j * 32 + g * 8 + e;
                red[nc & 1][wave][0][c] = t1;
                red[nc & 1][wave][1][c] = t2;
              }
            }
          }
        } else {
#pragma unroll
          for (int mi = 0; mi < MF; ++mi) {
            const long long row = mbase + mi * 16 + li;
            uint4 ov[2] = {make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)};
            mb[0] = mb[1] = 0u;
            if constexpr (PF == 1) {  // the prefetched res_grad rows arrive line-shaped: into the accumulator shape
              *reinterpret_cast<uint4*>(tw + tr0) = pg[mi][0];
              *reinterpret_cast<uint4*>(tw + tr0 + 1024) = pg[mi][1];
              pg[mi][0] = *reinterpret_cast<const uint4*>(tw + tw0);
              pg[mi][1] = *reinterpret_cast<const uint4*>(tw + tw1);
            }
            if constexpr (PF == 3) {  // the eight even pixels' `sub` rows arrive line-shaped (1 KB): pixel li = 2 s takes chunks g and 4 + g of row s
              const int ss = lane >> 3, sr = li >> 1;
              *reinterpret_cast<uint4*>(tw + ss * 128 + (((lane & 7) ^ ss) * 16)) = pg[mi][0];
              const uint4 e0 = *reinterpret_cast<const uint4*>(tw + sr * 128 + (((0 * 4 + g) ^ sr) * 16));
              const uint4 e1 = *reinterpret_cast<const uint4*>(tw + sr * 128 + (((1 * 4 + g) ^ sr) * 16));
              const bool ev = (li & 1) == 0;   // odd pixels carry no shortcut gradient
              pg[mi][0] = ev ? e0 : make_uint4(0, 0, 0, 0);
              pg[mi][1] = ev ? e1 : make_uint4(0, 0, 0, 0);
            }
            if constexpr (EP != 0) {
              if (EP == 2 || (EP == 1 && p.ep_res != nullptr)) {  // the prefetched residual rows arrive line-shaped: into the operand shape
                *reinterpret_cast<uint4*>(tw + tr0) = rq[mi][0];
                *reinterpret_cast<uint4*>(tw + tr0 + 1024) = rq[mi][1];
                rq[mi][0] = *reinterpret_cast<const uint4*>(tw + tw0);
                rq[mi][1] = *reinterpret_cast<const uint4*>(tw + tw1);
              }
            }
            if (FAST || row < p.M) {
#pragma unroll
              for (int j = 0; j < 2; ++j) {
                unsigned keep = 0xffu;
                if constexpr (PF) keep = ((j == 0 ? pk[mi].x : pk[mi].y) >> (8 * g)) & 0xffu;
                else if (DGRAD && p.fmode == 4) keep = p.fmask[row * (p.N >> 3) + ((n0 + j * 32 + g * 8) >> 3)];  // masked store, no sums
                ov[j] = store_chunk(mi, j, row, keep);
                if constexpr (CH) vout[mi][j] = ov[j];
              }
            }
            if constexpr (!FUSE) {
              if (lt) {  // wave-uniform; same wave, in-order LDS queue: no barrier
                *reinterpret_cast<uint4*>(tw + tw0) = ov[0];
                *reinterpret_cast<uint4*>(tw + tw1) = ov[1];
                const uint4 r0 = *reinterpret_cast<const uint4*>(tw + tr0), r1 = *reinterpret_cast<const uint4*>(tw + tr0 + 1024);
                const long long rowa = mbase + mi * 16 + (lane >> 3);
                bf16_t* da = p.out + rowa * p.N + n0 + (lane & 7) * 8;
                // linear stores (SH_SW_G1_LT: bit 0 forward, bit 1 data gradient).  Non-temporal when the epilogue is the BatchNorm + residual form (EP != 0:
                // the block output, next read two launches later and far larger than the caches); the plain forward's raw conv output is read by
                // the very next launch and stays cached
                // ... and for the DATA gradient of the stage-1 / stage-2 layers (K <= 128: dx is 3.3 / 1.6 GB, far beyond the caches, although the next
                // launch reads it): -0.6 ms per step in three same-box pairs; at K = 256 (0.8 GB) the hint is neutral (profiles/r06_cache_policy_ab.txt)
                constexpr bool NT_OUT = (EP != 0 && !DGRAD) || (DGRAD && !FUSE && K <= 128);
                if (FAST || rowa < p.M) st16<NT_OUT>(da, r0);
                if (FAST || rowa + 8 < p.M) st16<NT_OUT>(da + 8ll * p.N, r1);
                if constexpr (EP != 0) {
                  if (EP == 2 || (EP == 1 && p.ep_relu && p.ep_mask != nullptr)) {
                    // the pixel's eight mask bytes (chunks g and 4 + g of the four lanes li + 16 g) meet in one lane: one 8-B store per
                    // pixel instead of 128 one-byte stores per 16 pixels
                    unsigned mlo = mb[0] << (8 * g), mhi = mb[1] << (8 * g);
                    mlo |= (unsigned)__shfl_xor((int)mlo, 16); mhi |= (unsigned)__shfl_xor((int)mhi, 16);
                    mlo |= (unsigned)__shfl_xor((int)mlo, 32); mhi |= (unsigned)__shfl_xor((int)mhi, 32);
                    if (g == 0 && (FAST || row < p.M)) *reinterpret_cast<uint2*>(p.ep_mask + row * (p.N >> 3) + (n0 >> 3)) = make_uint2(mlo, mhi);
                  }
                }
              }
            }
          }
          if constexpr (CH) {
            // chained conv1: k-slice j of panel nc x the chunk's packed output (lane = pixel li, channels n0 + 32 j + 8 g .. +8: exactly
            // the A-operand layout) -> the same k order as the stand-alone conv1, so bit-identical accumulators
            const char* cP = sC + (K == 64 ? nc : (nc & 1)) * PB + pfr0;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
              uint4 fc[CT];
#pragma unroll
              for (int ct = 0; ct < CT; ++ct)
                fc[ct] = *reinterpret_cast<const uint4*>(cP + ((ct >> 2) * 64 + ((ct >> 1) & 1) * 32 + (ct & 1) * 4) * 128 + (kk == 0 ? pfo0 : pfo1));
#pragma unroll
              for (int mi = 0; mi < MF; ++mi)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) cacc[mi][ct] = mma_bf16(fc[ct], vout[mi][kk], cacc[mi][ct]);
            }
            if constexpr (K == 128) panel_store(sC + ((nc + 1) & 1) * PB);  // visible after the barrier that ends this step
          }
          if constexpr (EP == 1) {
            if (p.ep_res != nullptr && nc + 1 < nch) load_res(nc + 1);  // lands under the next chunk's MFMAs
          }
        }
#pragma unroll
        for (int mi = 0; mi < MF; ++mi)
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
      {
        char* dB = sB + (buf ^ 1) * BT;
        if constexpr (W2) {  // tile s + 1, fetched a whole k-step ago: rc after an even step, rb after an odd one
          if (ks == 0) SH_G1_STORE2(dB);
          else SH_G1_STORE(dB);
        } else {
          SH_G1_STORE(dB);
        }
      }
      __syncthreads();
      float* stat_out = DGRAD ? (FUSE ? p.fpartial : nullptr) : p.bn_partial;
      if (!FAST && ks == KSTEPS - 1 && stat_out != nullptr && tid < 128) {
        const int which = tid >> 6, c = tid & 63;
        const float v = (red[nc & 1][0][which][c] + red[nc & 1][1][which][c]) + (red[nc & 1][2][which][c] + red[nc & 1][3][which][c]);
        stat_out[((long long)blockIdx.x * 2 + which) * p.N + nc * 64 + c] = v;
      }
    }
  }
  if constexpr (CH) {
    // the chained conv1's raw output: bf16 store (8 consecutive channels per lane) and the BatchNorm partial sums of the fp32
    // accumulators in the layout of the stand-alone forward (one row pair per block), 64 channels (= 4 tiles) at a time
#pragma unroll
    for (int mi = 0; mi < MF; ++mi) {
      const long long row = mbase + mi * 16 + li;
#pragma unroll
      for (int jj = 0; jj < CT / 2; ++jj) {
        const f32x4 lo = cacc[mi][2 * jj], hi = cacc[mi][2 * jj + 1];
        *reinterpret_cast<uint4*>(p.chain_y + row * K + jj * 32 + g * 8) =
            make_uint4(pack_bf16x2(lo[0], lo[1]), pack_bf16x2(lo[2], lo[3]), pack_bf16x2(hi[0], hi[1]), pack_bf16x2(hi[2], hi[3]));
      }
    }
#pragma unroll
    for (int hf = 0; hf < CT / 4; ++hf) {
      __syncthreads();  // red[0] is free (the loop ended with a barrier / the previous half was read)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float s1 = 0.f, s2 = 0.f;
#pragma unroll
          for (int mi = 0; mi < MF; ++mi) {
            const float v = cacc[mi][hf * 4 + ni][r];
            s1 += v;
            s2 += v * v;
          }
          s1 = row16_sum_g1(s1);
          s2 = row16_sum_g1(s2);
          if (li == 0) {
            const int c = (ni >> 1) * 32 + g * 8 + (ni & 1) * 4 + r;
            red[0][wave][0][c] = s1;
            red[0][wave][1][c] = s2;
          }
        }
      __syncthreads();
      if (tid < 128) {
        const int which = tid >> 6, c = tid & 63;
        const float v = (red[0][0][which][c] + red[0][1][which][c]) + (red[0][2][which][c] + red[0][3][which][c]);
        p.chain_partial[((long long)blockIdx.x * 2 + which) * K + hf * 64 + c] = v;
      }
    }
  }
}

#undef SH_G1_LOAD
#undef SH_G1_STORE
#undef SH_G1_LOAD2
#undef SH_G1_STORE2

bool gemm1x1_supported(int k, int n) { return (k == 64 || k == 128 || k == 256) && n % 64 == 0 && n >= 64; }

// rows per block = 64 * MF; tuned per K on MI355X (scripts/conv_bench.py), overridable for experiments
static hook_t g_mf[3] = {{4}, {2}, {2}};  // K = 64, 128, 256
void gemm1x1_set_stem_persistent(int on);
void hooks_reset_1x1() { g_mf[0] = 4; g_mf[1] = 2; g_mf[2] = 2; gemm1x1_set_chain(-1); gemm1x1_set_stem_persistent(1); stem_ring_enable(-1); }
static int mf_of(int k) { return g_mf[k == 64 ? 0 : (k == 128 ? 1 : 2)]; }
void gemm1x1_set_mf(int k, int mf) { g_mf[k == 64 ? 0 : (k == 128 ? 1 : 2)] = mf; }

// the branch-free variants (EP == 2, PF) per K; simhand_test_switch(SH_SW_G1_PF) for A/B timing against the generic kernels
static int pf_of(int k) {
  const int env = sw(SH_SW_G1_PF);  // bit 0: K = 64, 1: 128, 2: 256
  return (env >> (k == 64 ? 0 : (k == 128 ? 1 : 2))) & 1;
}

int gemm1x1_rows_per_block(int k) { return 64 * mf_of(k); }

// the fast data-gradient variants (PF) stage their block's mask rows in LDS: N / 8 bytes per row in whole, XOR-swizzled 16-B slots inside a
// fixed row pitch (32 B at K = 64, 128 B otherwise)
static bool pf_masks_ok(int k, int n) { return n >= 128 && (n & (n - 1)) == 0 && n <= (k == 64 ? 256 : 1024); }

// chained next conv1: K = 64 (128-row blocks, all panels resident: N <= 256) and K = 128 (64-row blocks, streamed panels: N <= 512)
// bit 0: K = 64, bit 1: K = 128; -1 = simhand_test_switch(SH_SW_G1_CHAIN), default 1 (test / tuning hooks).  K = 128 is OFF by default:
// measured at 2048 x 28^2 the chained launch takes 1353 us against 908 + 434 for the two separate ones (its 64-row blocks and the
// streamed panels cost what the saved read of the block output gains); K = 64 @ 56^2: 1816 against 1590 + 750.
static hook_t g_chain{-1};
static int chain_env() {
  const int env = sw(SH_SW_G1_CHAIN);
  const int h = g_chain;
  return h >= 0 ? h : env;
}
void gemm1x1_set_chain(int mask) { g_chain = mask; }
bool gemm1x1_chain_ok(int k, int n, long long m) {
  if (k == 64) return (chain_env() & 1) && n % 64 == 0 && n <= 256 && m % 128 == 0 && pf_of(64);
  if (k == 128) return (chain_env() & 2) && n % 64 == 0 && n <= 512 && m % 64 == 0 && pf_of(128);
  return false;
}
int gemm1x1_chain_rows(int k) { return k == 64 ? 128 : 64; }

bool gemm1x1_sub_ok(const Gemm1x1Args& a, int k) {
  const int mf = mf_of(k);
  return a.fpartial == nullptr && a.accumulate == 0 && a.fmode == 4 && a.fmask != nullptr && a.bias == nullptr && a.M % (64 * mf) == 0 &&
         pf_of(k) && pf_masks_ok(k, a.N) && a.sub_h % 2 == 0 && a.sub_w % 2 == 0 && a.M < (1ll << 31);
}

int launch_gemm1x1(const Gemm1x1Args& a_in, int k, bool dgrad, hipStream_t s) {
  // bit 0: forward launches, bit 1: data gradients (round 4: measured neutral there; round 6, re-measured on the final tree: data-gradient class
  // 34.54 -> 34.32 ms in three same-box pairs, profiles/r06_cache_policy_ab.txt -- on by default now)
  const int lt_env = sw(SH_SW_G1_LT);
  Gemm1x1Args a = a_in;
  a.lt = (lt_env >> (dgrad ? 1 : 0)) & 1;
  if (a.chain_w != nullptr) {  // the caller checked gemm1x1_chain_ok and passes residual + ReLU + mask
    route_hit(SH_ROUTE_GEMM1X1_FWD_BNACT);
    route_hit(SH_ROUTE_FWD_CHAIN);
    if (k == 64) gemm1x1_kernel<64, 2, false, false, 2, false, true><<<ceil_div(a.M, 128), 256, 0, s>>>(a);
    else gemm1x1_kernel<128, 1, false, false, 2, false, true><<<ceil_div(a.M, 64), 256, 0, s>>>(a);
    return 0;
  }
  int mf = mf_of(k);
  // the branch-free BN + residual + ReLU epilogue at K = 64 wants 128-row blocks (its rows are requested one short chunk ahead:
  // more resident blocks hide what the 32 MFMAs of a chunk cannot; measured 1.53 ms against 2.05 with 256 rows, 1.64 generic)
  if (!dgrad && k == 64 && mf == 4 && a.ep_scale != nullptr && a.ep_res != nullptr && a.ep_relu && a.ep_mask != nullptr && a.M % 128 == 0 && pf_of(64))
    mf = 2;
  // the scale / shift-only fast variant (EP == 3) at K = 64: 128-row blocks as well (the 256-row form spills at three blocks per CU)
  if (!dgrad && k == 64 && mf == 4 && a.ep_scale != nullptr && a.ep_res == nullptr && !a.ep_relu && a.ep_mask == nullptr && a.M % 128 == 0 && pf_of(64))
    mf = 2;
  const int nblk = ceil_div(a.M, 64 * mf);
  const bool full = a.M % (64 * mf) == 0;
  route_hit(dgrad ? SH_ROUTE_GEMM1X1_DGRAD : (a.ep_scale != nullptr ? SH_ROUTE_GEMM1X1_FWD_BNACT : SH_ROUTE_GEMM1X1_FWD));
#define SH_G1(KV, MFV)                                                                          \
  do {                                                                                          \
    if (dgrad && a.fpartial != nullptr) gemm1x1_kernel<KV, MFV, true, true><<<nblk, 256, 0, s>>>(a);  \
    else if (dgrad && a.accumulate == 2 && a.fmode == 4 && a.fmask != nullptr && a.bias == nullptr && full && pf_of(KV) && pf_masks_ok(KV, a.N))  \
      gemm1x1_kernel<KV, MFV, true, false, 0, 1><<<nblk, 256, 0, s>>>(a);                       \
    else if (dgrad && a.accumulate == 0 && a.fmode == 4 && a.fmask != nullptr && a.bias == nullptr && full && pf_of(KV) && pf_masks_ok(KV, a.N) && a.sub != nullptr)  \
      gemm1x1_kernel<KV, MFV, true, false, 0, 3><<<nblk, 256, 0, s>>>(a);                       \
    else if (dgrad && a.accumulate == 0 && a.fmode == 4 && a.fmask != nullptr && a.bias == nullptr && full && pf_of(KV) && pf_masks_ok(KV, a.N))  \
      gemm1x1_kernel<KV, MFV, true, false, 0, 2><<<nblk, 256, 0, s>>>(a);                       \
    else if (dgrad) gemm1x1_kernel<KV, MFV, true><<<nblk, 256, 0, s>>>(a);                      \
    else if (a.ep_scale != nullptr && a.ep_res != nullptr && a.ep_relu && a.ep_mask != nullptr && full && pf_of(KV))  \
      gemm1x1_kernel<KV, MFV, false, false, 2><<<nblk, 256, 0, s>>>(a);                         \
    else if (a.ep_scale != nullptr && a.ep_res == nullptr && !a.ep_relu && a.ep_mask == nullptr && full && pf_of(KV))  \
      gemm1x1_kernel<KV, MFV, false, false, 3><<<nblk, 256, 0, s>>>(a);                         \
    else if (a.ep_scale != nullptr) gemm1x1_kernel<KV, MFV, false, false, 1><<<nblk, 256, 0, s>>>(a);  \
    else gemm1x1_kernel<KV, MFV, false><<<nblk, 256, 0, s>>>(a);                                \
  } while (0)
  if (k == 64) {
    if (mf == 4) SH_G1(64, 4); else if (mf == 2) SH_G1(64, 2); else SH_G1(64, 1);
  } else if (k == 128) {
    if (mf == 4) SH_G1(128, 4); else if (mf == 2) SH_G1(128, 2); else SH_G1(128, 1);
  } else {
    if (mf == 2) SH_G1(256, 2); else SH_G1(256, 1);
  }
#undef SH_G1
  return 0;
}

// ---- persistent direct-stem forward ------------------------------------------------------------------------------------
// The stem (N = 64) is ONE output chunk per row tile: in gemm1x1_kernel every block pays a prologue (its A rows' first touch, a
// weight tile, a barrier) per 256 rows and nothing overlaps it.  Here the whole [64][256] weight matrix (32 KB) stays in LDS for the
// life of a block, a block walks row tiles of 64 pixels (grid = a few blocks per CU), and the loop has no barrier and no weight
// traffic at all: a wave requests the NEXT tile's seven k-slices (filter rows 0..6; row 7 is zero weights) before it multiplies
// the current tile, then stores 64 channels per pixel and adds the tile into per-lane BatchNorm partial sums that leave the block
// ONCE, at the end (partial rows = blocks, not tiles).  Same fragment layout, k order and rounding as gemm1x1_kernel<256, ., ST>.
constexpr int STEM_PB = 512;  // persistent blocks: 2 per CU on 256 CUs (230 VGPRs; a 168-register build for 3 per CU spills: 4.0 ms)
#ifndef SH_STEM_MF
#define SH_STEM_MF 1
#endif
__global__ __launch_bounds__(256, 2) void stem_fwd_persistent_kernel(Gemm1x1Args p) {
  constexpr int K = 256, KC = 128, MF = SH_STEM_MF, KF = 7, ROWB = 256, BT = 64 * ROWB, TR = 64 * MF;  // TR: rows per tile
  __shared__ __attribute__((aligned(16))) char sW[2 * BT];
  __shared__ float red[4][2][64];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int li = lane & 15, g = lane >> 4;
  // weights -> LDS once (two [64][128] tiles, 16-B chunks XOR-swizzled by (row & 3) | ((row >> 3) & 3) << 2 as in gemm1x1_kernel)
  for (int id = tid; id < 2 * 64 * 16; id += 256) {
    const int t = id >> 10, r = (id >> 4) & 63, c = id & 15;
    const uint4 v = *reinterpret_cast<const uint4*>(p.w + (long long)r * K + t * KC + c * 8);
    *reinterpret_cast<uint4*>(sW + t * BT + r * ROWB + ((c ^ ((r & 3) | (((r >> 3) & 3) << 2))) * 16)) = v;
  }
  int frow[4];
#pragma unroll
  for (int ni = 0; ni < 4; ++ni) frow[ni] = ((ni >> 1) * 32 + (li >> 2) * 8 + (ni & 1) * 4 + (li & 3)) * ROWB;
  int fo[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) fo[kk] = ((kk * 4 + g) ^ li) * 16;
  const long long ntiles = (p.M + TR - 1) / TR;
  const int jstride = p.stem_wp * 4;
  auto row_ptr = [&](long long row, bool& ok) __attribute__((always_inline)) -> const bf16_t* {
    ok = row < p.M;
    const unsigned m = ok ? (unsigned)row : 0u;
    const unsigned img = fdiv(m, p.div_hw);
    const unsigned rem = m - img * p.div_hw.d;
    const unsigned ho = fdiv(rem, p.div_w);
    const unsigned wo = rem - ho * p.div_w.d;
    return p.a + (((long long)img * p.stem_hp + 2 * ho) * p.stem_wp + 2 * wo) * 4 + g * 8;
  };
  uint4 cur[MF][KF], nxt[MF][KF];
  auto load_tile = [&](long long tile, uint4 (&dst)[MF][KF]) __attribute__((always_inline)) {
#pragma unroll
    for (int mi = 0; mi < MF; ++mi) {
      bool ok;
      const bf16_t* pr = row_ptr(tile * TR + wave * (16 * MF) + mi * 16 + li, ok);
#pragma unroll
      for (int j = 0; j < KF; ++j) {
        const uint4 v = *reinterpret_cast<const uint4*>(pr + j * jstride);  // rows past M read pixel 0 (branch-free) and are zeroed
        dst[mi][j] = ok ? v : make_uint4(0, 0, 0, 0);
      }
    }
  };
  float s1[4][4], s2[4][4];
#pragma unroll
  for (int ni = 0; ni < 4; ++ni)
#pragma unroll
    for (int r = 0; r < 4; ++r) s1[ni][r] = s2[ni][r] = 0.f;
  long long tile = blockIdx.x;
  if (tile < ntiles) load_tile(tile, cur);
  __syncthreads();  // weights visible
  for (; tile < ntiles; tile += gridDim.x) {
    const long long tn = tile + gridDim.x;
    load_tile(tn < ntiles ? tn : tile, nxt);  // in flight under this tile's MFMAs (past the end: this tile again, unused)
    f32x4 acc[MF][4];
#pragma unroll
    for (int mi = 0; mi < MF; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < KF; ++j) {
      const char* cB = sW + (j >> 2) * BT;
      uint4 fb[4];
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) fb[ni] = *reinterpret_cast<const uint4*>(cB + frow[ni] + fo[j & 3]);
#pragma unroll
      for (int mi = 0; mi < MF; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = mma_bf16(fb[ni], cur[mi][j], acc[mi][ni]);
    }
#pragma unroll
    for (int mi = 0; mi < MF; ++mi) {
      const long long row = tile * TR + wave * (16 * MF) + mi * 16 + li;
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v = acc[mi][ni][r];  // rows past M are exact zeros (zeroed operands)
          s1[ni][r] += v;
          s2[ni][r] += v * v;
        }
      if (row < p.M) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const f32x4 lo = acc[mi][2 * j], hi = acc[mi][2 * j + 1];
          *reinterpret_cast<uint4*>(p.out + row * 64 + j * 32 + g * 8) =
              make_uint4(pack_bf16x2(lo[0], lo[1]), pack_bf16x2(lo[2], lo[3]), pack_bf16x2(hi[0], hi[1]), pack_bf16x2(hi[2], hi[3]));
        }
      }
    }
#pragma unroll
    for (int mi = 0; mi < MF; ++mi)
#pragma unroll
      for (int j = 0; j < KF; ++j) cur[mi][j] = nxt[mi][j];
  }
  if (p.bn_partial != nullptr) {
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float t1 = row16_sum_g1(s1[ni][r]), t2 = row16_sum_g1(s2[ni][r]);
        if (li == 0) {
          const int c = (ni >> 1) * 32 + g * 8 + (ni & 1) * 4 + r;
          red[wave][0][c] = t1;
          red[wave][1][c] = t2;
        }
      }
    __syncthreads();
    if (tid < 128) {
      const int which = tid >> 6, c = tid & 63;
      p.bn_partial[((long long)blockIdx.x * 2 + which) * 64 + c] = (red[0][which][c] + red[1][which][c]) + (red[2][which][c] + red[3][which][c]);
    }
  }
}

static hook_t g_stem_persistent{1};
void gemm1x1_set_stem_persistent(int on) { g_stem_persistent = on ? 1 : 0; }
int gemm1x1_stem_stat_blocks(long long m) {
  if (!g_stem_persistent) return ceil_div(m, 256);
  const long long tiles = (m + 64 * SH_STEM_MF - 1) / (64 * SH_STEM_MF);
  return (int)(tiles < STEM_PB ? tiles : STEM_PB);
}

int launch_gemm1x1_stem(const Gemm1x1Args& a, hipStream_t s) {
  if (g_stem_persistent) {
    stem_fwd_persistent_kernel<<<gemm1x1_stem_stat_blocks(a.M), 256, 0, s>>>(a);
    return 0;
  }
  // 256 rows per block: with N = 64 a block has ONE output chunk, so its prologue (the A rows' first touch) is all the latency it
  // can hide; twice the rows per prologue measured 1.76 -> 1.62 ms at 2048 x 224^2 (250 VGPRs, still two blocks per CU)
  gemm1x1_kernel<256, 4, false, false, 0, 0, false, true><<<ceil_div(a.M, 256), 256, 0, s>>>(a);
  return 0;
}

}  // namespace sh
