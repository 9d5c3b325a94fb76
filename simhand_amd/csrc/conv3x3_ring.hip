// 3x3 / stride 1 / pad 1 convolution with 128 input and 128 output channels (bf16): forward and data gradient of the second
// residual stage of ResNet-50/101/152 at 28 x 28 -- the layers where both tile kernels of conv_igemm.hip stop at ~800 TFLOP/s.
//
// Replaces (reference): layer2.*.conv2 of torchvision's Bottleneck (src/models/resnet_model.py:13-58) and its input gradient
// (cuDNN there).
//
// Why a kernel of its own: with 128 destination channels an implicit-GEMM tile re-stages its activation rows for every tap and
// amortises them over only half the matrix work of the 256-wide tile -- per 64-deep k-step the 256 x 128 form of igemm256_kernel
// took 1.12 us for half the MFMAs the 256 x 256 form does in 1.77 us (DESIGN "Round 3": operand delivery, not the staging
// mechanism, is what those loops wait for).  Here the activation tile is staged ONCE per tile: on the zero-padded pixel grid with
// shared padding ((H+1) x (W+1) positions per image, as in conv3x3_c64_kernel / wgrad3x3_kernel) tap (r, s) is the constant row
// shift (r-1)(W+1) + (s-1), so a block that owns 256 consecutive padded positions loads those rows plus a 32-row halo on either
// side into LDS (two 64-channel halves of 320 rows x 128 B = 80 KB, by LDS-DMA, zero page for pad positions) and reads all nine
// tap operands from it at nine row offsets; only the weights move per k-step (tap x 64 input channels: one 16-KB tile, three
// stages, two steps ahead).  864 KB of operands per tile become 368 KB.
//   8 waves as 4 (pixels) x 2 (channels): a wave owns 64 positions x 64 channels = 4 x 4 MFMA tiles (64 accumulator registers; 16
//     fragment reads per 32 MFMAs -- the 2 x 4 split with 8 x 2 tiles per wave needs 20 and measured 525 instead of ... us);
//   the ring's 16-B chunks are XOR-swizzled by row / 2 on the DMA source side, and MFMA column li reads pixel pl(li) of its
//     16-pixel group (even pixels for one half of a ds_read_b128 service group, odd ones for the other: conflict-free for every
//     tap offset -- the analysis is conv3x3_c64_kernel's);
//   k order: input-channel half outermost (the second half of the ring may still be in flight while the first nine taps run);
//   outputs at pad positions are computed and dropped ((H+1)(W+1)/(HW) = 7 % extra MFMAs at 28 x 28);
//   forward: BN partial sums of the fp32 results, one [2][128] row per tile; data gradient: optionally the previous unit's
//     BN-backward sums (sum g, sum g*y, ReLU mask recomputed from y), as conv3x3_c64_kernel.
#include "conv3x3_ring.h"

#include <stdlib.h>

#ifndef SH_R128_YPF
#define SH_R128_YPF 1  // MODE 2: request the previous unit's y rows during the last k-steps (0: in the epilogue -- A/B builds)
#endif

namespace sh {

__device__ uint4 g_r128_zero_page[8];
// BNIN: pad positions fetch NaNs; NaN * scale + shift is NaN and the ReLU (v > 0 ? v : 0) makes it the exact zero the padding needs
#ifdef SH_H16_FP16
#define SH_R128_NAN2 0x7e007e00u
#else
#define SH_R128_NAN2 0x7fc07fc0u
#endif
__device__ uint4 g_r128_nan_page[8] = {
    {SH_R128_NAN2, SH_R128_NAN2, SH_R128_NAN2, SH_R128_NAN2}, {SH_R128_NAN2, SH_R128_NAN2, SH_R128_NAN2, SH_R128_NAN2},
    {SH_R128_NAN2, SH_R128_NAN2, SH_R128_NAN2, SH_R128_NAN2}, {SH_R128_NAN2, SH_R128_NAN2, SH_R128_NAN2, SH_R128_NAN2},
    {SH_R128_NAN2, SH_R128_NAN2, SH_R128_NAN2, SH_R128_NAN2}, {SH_R128_NAN2, SH_R128_NAN2, SH_R128_NAN2, SH_R128_NAN2},
    {SH_R128_NAN2, SH_R128_NAN2, SH_R128_NAN2, SH_R128_NAN2}, {SH_R128_NAN2, SH_R128_NAN2, SH_R128_NAN2, SH_R128_NAN2}};
__device__ uint4 g_r128_bn_sink[512];  // the by-product's stores of halo / pad rows (a neighbour tile owns them / nobody does)

__device__ __forceinline__ float row16_sum_r128(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));  // row_ror:8
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));  // row_ror:4
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));  // row_ror:2
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));  // row_ror:1
  return v;
}

// MODE 0: store only; 1: forward + BN partial statistics; 2: data gradient + BN-backward sums of the previous unit
// BNIN (forward): x is the previous unit's RAW conv output; every lane rewrites the 2 x 5 ring pieces it fetched itself as
// relu(x * in_scale + in_shift) (its own s_waitcnt is the only synchronisation that needs) and stores the ones of the tile's own 256 positions
// to a_out -- the bn_apply pass disappears
// (The backward twin -- this unit's BatchNorm-backward apply rewritten into the data gradient's ring, round 4 -- was exact but VALU-bound:
// ~90 VALU instructions per 16 bytes, 752 us against 548 + a 200-us pass, step unchanged; removed in round 5, see docs/lab-notes.md.)
template <int MODE, bool BNIN = false>
__global__ __launch_bounds__(512, 1) void conv3x3_r128_kernel(R128Args p) {
  static_assert(!BNIN || MODE != 2, "BNIN is a forward form");
  constexpr int BM = 256, BN = 128, MI = 4, NI = 4, HALO = 32;
  constexpr int RROWS = BM + 2 * HALO;  // 320 ring rows: positions m0 - 32 .. m0 + 288
  constexpr int HALF = RROWS * 128;     // one 64-channel half of the ring
  constexpr int A_BYTES = 2 * HALF;
  constexpr int BST = BN * 128;         // one weight stage: 128 destination channels x 64 k
  constexpr int NST = 3;
  constexpr int NK = 18;                // 2 input-channel halves x 9 taps
  __shared__ __attribute__((aligned(16))) char smem[A_BYTES + NST * BST];

  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int li = lane & 15, g = lane >> 4;
  const int wm = wave >> 1, wn = wave & 1;
  // pixel of MFMA column li inside a 16-pixel group (see conv3x3_c64_kernel)
  const int pl = (li & 4) == ((li & 8) >> 1) ? 2 * ((li & 3) + ((li >> 3) << 2)) : 2 * (li - 4) + 1;
  const int WP = p.W + 1;
  // XCD-aware tile order: hardware places block b on XCD b % 8; each XCD gets a contiguous range of tiles
  int tile;
  {
    const int nblk = gridDim.x, q8 = nblk >> 3, r8 = nblk & 7, xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + j;
  }
  const long long m0 = (long long)tile * BM;

  auto dma16 = [](const void* src, unsigned lds_addr) __attribute__((always_inline)) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_addr), "v"(src));
  };
  const unsigned smem_addr = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  const int slot = lane & 7, r8l = lane >> 3;
  const char* zsrc = reinterpret_cast<const char*>(g_r128_zero_page) + slot * 16;
  const char* nsrc = reinterpret_cast<const char*>(BNIN ? g_r128_nan_page : g_r128_zero_page) + slot * 16;
  // BNIN: the lane's 16-B slot holds the same 8 channels in all five rows it fetches per half (the rows' swizzle keys agree): coefficients in
  // registers for the prologue; the loads go out first, ahead of every DMA in the in-order queue
  const int bn_chunk = slot ^ (((wave & 1) * 4 + (r8l >> 1)) & 7);
  float bsc[BNIN ? 2 : 1][8], bsh[BNIN ? 2 : 1][8];
  unsigned boff[BNIN ? 2 : 1][5];
  if constexpr (BNIN) {
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const float4 a = *reinterpret_cast<const float4*>(p.in_scale + h * 64 + bn_chunk * 8 + 4 * i);
        const float4 b = *reinterpret_cast<const float4*>(p.in_shift + h * 64 + bn_chunk * 8 + 4 * i);
        bsc[h][4 * i] = a.x; bsc[h][4 * i + 1] = a.y; bsc[h][4 * i + 2] = a.z; bsc[h][4 * i + 3] = a.w;
        bsh[h][4 * i] = b.x; bsh[h][4 * i + 1] = b.y; bsh[h][4 * i + 2] = b.z; bsh[h][4 * i + 3] = b.w;
      }
  }

  // ---- ring fill: half h = 40 instructions of 8 rows, instruction n of the half by wave n % 8 (5 per wave); lane l = row 8 rb + (l >> 3),
  // 16-B slot l & 7, source chunk slot ^ key(row) --------------------------------------------------------------------------------
  auto fill_half = [&](int h) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int rb = wave + 8 * i;                       // 0..39
      const int j = rb * 8 + r8l;                        // ring row
      const long long q = m0 - HALO + j;
      const bool in = q >= 0 && q < p.q_total;
      const unsigned qu = in ? (unsigned)q : 0u;         // q_total < 2^31 (checked on the host)
      const unsigned img = fdiv(qu, p.div_pp);
      const unsigned rem = qu - img * p.div_pp.d;
      const unsigned hp = fdiv(rem, p.div_wp);
      const unsigned wp = rem - hp * p.div_wp.d;
      const bool ok = in && hp - 1u < (unsigned)p.H && wp - 1u < (unsigned)p.W;
      const unsigned pix = (img * (unsigned)p.H + (hp - 1u)) * (unsigned)p.W + (wp - 1u);
      const int chunk = slot ^ ((j >> 1) & 7);
      const char* src = ok ? reinterpret_cast<const char*>(p.x + (unsigned long long)pix * 128 + h * 64 + chunk * 8) : nsrc;
      dma16(src, smem_addr + h * HALF + rb * 8 * 128);
      if constexpr (BNIN)  // where the rewritten 16 bytes go: own positions (ring rows 32 .. 287) that are real pixels; < 2^32 (checked on the host)
        boff[h][i] = ok && j >= HALO && j < HALO + BM ? pix * 128u + (unsigned)(h * 64 + chunk * 8) : 0xffffffffu;
    }
  };
  auto bn_piece = [&](int h, int i) __attribute__((always_inline)) {
    const int rb = wave + 8 * i;
    char* at = smem + h * HALF + (rb * 8 + r8l) * 128 + slot * 16;
    const uint4 v = *reinterpret_cast<const uint4*>(at);
    const unsigned w4[4] = {v.x, v.y, v.z, v.w};
    float o[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) {  // the arithmetic of bn_apply_kernel (bn.hip), bit for bit
      o[2 * e] = h16_lo(w4[e]) * bsc[BNIN ? h : 0][2 * e] + bsh[BNIN ? h : 0][2 * e];
      o[2 * e + 1] = h16_hi(w4[e]) * bsc[BNIN ? h : 0][2 * e + 1] + bsh[BNIN ? h : 0][2 * e + 1];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = o[e] > 0.f ? o[e] : 0.f;
    uint4 r;
    r.x = pack_bf16x2(o[0], o[1]);
    r.y = pack_bf16x2(o[2], o[3]);
    r.z = pack_bf16x2(o[4], o[5]);
    r.w = pack_bf16x2(o[6], o[7]);
    *reinterpret_cast<uint4*>(at) = r;
    const unsigned bo = boff[BNIN ? h : 0][i];
    uint4* dst = bo != 0xffffffffu ? reinterpret_cast<uint4*>(p.a_out + bo) : &g_r128_bn_sink[tid];
    *dst = r;
  };
  // ---- weight tile of k-step ks (half h = ks / 9, tap t = ks % 9): rows lrow, lrow + 64 of [128][64 k], chunk slot ^ key_b(row) ----
  auto key_b = [](int row) __attribute__((always_inline)) -> int { return ((row >> 1) & 1) | (((row >> 3) & 3) << 1); };
  const int lrow = wave * 8 + r8l;  // 0..63
  const char* wb = reinterpret_cast<const char*>(p.w + (long long)lrow * (9 * 128) + (slot ^ key_b(lrow)) * 8);
  auto dma_w = [&](int ks) __attribute__((always_inline)) {
    const bool live = ks < NK;  // past the end: the zero page into the idle stage (the counted waits see two instructions per step)
    const int h = ks >= 9 ? 1 : 0, t = ks - 9 * h;
    const unsigned dst = smem_addr + A_BYTES + (unsigned)(ks % NST) * BST + wave * 8 * 128;
    const char* s0 = wb + (t * 128 + h * 64) * 2;
    dma16(live ? s0 : zsrc, dst);
    dma16(live ? s0 + 64ll * (9 * 128) * 2 : zsrc, dst + 64 * 128);
  };

  // prologue, in the order the counted waits assume: half 0 (5), weights of steps 0 and 1 (2 + 2), half 1 (5)
  fill_half(0);
  dma_w(0);
  dma_w(1);
  fill_half(1);
  if constexpr (BNIN) {
    // in-order retirement: <= 9 outstanding = the lane's rows of half 0 are in LDS (the compiler's own loads -- y pieces, coefficients -- may sit
    // anywhere in the queue: extra operations only make a counted wait stricter).  Half 1 is rewritten piece by piece under k-steps 0 .. 4.
    asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
#pragma unroll
    for (int i = 0; i < 5; ++i) bn_piece(0, i);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the rewritten rows are in LDS before the loop's first barrier
  }

  // ---- fragment addressing ----------------------------------------------------------------------------------------------------
  int trow[9], tcol[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    int off = (t / 3 - 1) * WP + (t % 3 - 1);
    if (p.dgrad) off = -off;  // dx[q] = sum_t dy[q - off_t] W[.][t][.]
    trow[t] = HALO + pl + off;                    // >= 32 - 31 > 0 (W <= 30)
    tcol[t] = (g ^ ((trow[t] >> 1) & 7)) * 16;    // chunk kk*4 + g of the k-step: kk = 1 flips bit 6
  }
  const int a_base = (wm * 64) * 128;
  // weights: fragment row li of channel tile ni <-> destination channel wn*64 + (ni >> 1)*32 + (li >> 2)*8 + (ni & 1)*4 + (li & 3): a lane's
  // accumulator registers of tiles 2j, 2j + 1 are then 8 CONSECUTIVE channels (one 16-B store per pixel and 32-channel group)
  const int rowb0 = wn * 64 + (li >> 2) * 8 + (li & 3);
  const int fbo = rowb0 * 128 + ((g ^ key_b(rowb0)) * 16);

  f32x4 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // positions -> NHWC pixels of this lane's output rows (decoded inside the loop's tail, see below); MODE 2: the previous unit's y rows
  unsigned pixs[MI];
  bool oks[MI];
  uint4 yq[MODE == 2 ? MI : 1][2];
  const int ch0 = wn * 64 + g * 8;
  auto decode_rows = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const long long q = m0 + wm * 64 + mi * 16 + pl;
      const bool in = q < p.q_total;
      const unsigned qu = in ? (unsigned)q : 0u;
      const unsigned img = fdiv(qu, p.div_pp);
      const unsigned rem = qu - img * p.div_pp.d;
      const unsigned hp = fdiv(rem, p.div_wp);
      const unsigned wp = rem - hp * p.div_wp.d;
      oks[mi] = in && hp - 1u < (unsigned)p.H && wp - 1u < (unsigned)p.W;
      pixs[mi] = (img * (unsigned)p.H + (hp - 1u)) * (unsigned)p.W + (wp - 1u);
      if constexpr (MODE == 2) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const void* src = oks[mi] ? (const void*)(p.fy + (unsigned long long)pixs[mi] * 128 + ch0 + 32 * j) : (const void*)g_r128_zero_page;
          yq[mi][j] = *reinterpret_cast<const uint4*>(src);
        }
      }
    }
  };

  auto kstep = [&](int h, int t) __attribute__((always_inline)) {
    {
      const int ks = h * 9 + t;
      // vector-memory operations retire in order.  Issued so far: half 0, W0, W1, half 1, W2 .. W(ks+1); step ks needs W(ks) (and,
      // from ks = 9 on, half 1 -- older than W2, landed since step 2): after it may fly W1 + half 1 (ks = 0), half 1 + W2 (ks = 1),
      // W(ks+1) (ks >= 2)
      // MODE 2: the 8 loads of the previous unit's y rows go out in step 15 (behind W17), so that their round trip is over when the
      // epilogue wants them: steps 16 and 17 then leave 2 + 8 operations in flight
      if constexpr (BNIN) {
        // queue: half 0 (5), W0, W1 (2 + 2), half 1 (5), the 5 by-product stores of half 0, then per step k: piece k's store (k <= 4), W(k + 2) (2).
        // Step 0 needs W0 and half 1's first instruction (9 younger), step 1 W1 and the second (11), step k >= 2 W(k): piece k - 1's store and
        // W(k + 1) are younger (3; 2 from step 6 on)
        if (ks == 0) asm volatile("s_waitcnt vmcnt(9)\n\ts_barrier" ::: "memory");
        else if (ks == 1) asm volatile("s_waitcnt vmcnt(11)\n\ts_barrier" ::: "memory");
        else if (ks < 6) asm volatile("s_waitcnt vmcnt(3)\n\ts_barrier" ::: "memory");
        else if (MODE == 2 && SH_R128_YPF && h == 1 && t >= 7) asm volatile("s_waitcnt vmcnt(10)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(2)\n\ts_barrier" ::: "memory");
        if (t < 5) {
          if (h == 0) bn_piece(1, t);  // piece t of half 1, under this step's MFMAs of the other waves (first read in step 9: five barriers away)
        }
      } else {
      if (ks < 2) asm volatile("s_waitcnt vmcnt(7)\n\ts_barrier" ::: "memory");
      else if (MODE == 2 && SH_R128_YPF && h == 1 && t >= 7) asm volatile("s_waitcnt vmcnt(10)\n\ts_barrier" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(2)\n\ts_barrier" ::: "memory");
      }
      dma_w(ks + 2);  // stage (ks + 2) % 3 = the one step ks - 1 read: every wave is past it
      if (MODE == 2 && SH_R128_YPF && t == 6 && h == 1) decode_rows();
      const char* sa = smem + h * HALF + a_base;
      const int ob = A_BYTES + (ks % NST) * BST + fbo;  // (A_BYTES, BST multiples of 128: ^ 64 flips the chunk's bit 2 only)
      uint4 fb[2][NI];
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const int on = ob + ((ni >> 1) * 32 + (ni & 1) * 4) * 128;
        fb[0][ni] = *reinterpret_cast<const uint4*>(smem + on);
        fb[1][ni] = *reinterpret_cast<const uint4*>(smem + (on ^ 64));
      }
      const int ra = trow[t] * 128 + tcol[t];
      // the fragments of row group mi + 1 are read while the MFMAs of group mi run (two register sets; the scheduler would otherwise
      // sink the reads to their uses)
      uint4 fa[2][2];
      fa[0][0] = *reinterpret_cast<const uint4*>(sa + ra);
      fa[0][1] = *reinterpret_cast<const uint4*>(sa + (ra ^ 64));
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        if (mi + 1 < MI) {
          fa[(mi + 1) & 1][0] = *reinterpret_cast<const uint4*>(sa + (mi + 1) * 16 * 128 + ra);
          fa[(mi + 1) & 1][1] = *reinterpret_cast<const uint4*>(sa + (mi + 1) * 16 * 128 + (ra ^ 64));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = sh_mfma16(fb[0][ni], fa[mi & 1][0], acc[mi][ni]);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = sh_mfma16(fb[1][ni], fa[mi & 1][1], acc[mi][ni]);
        __builtin_amdgcn_sched_barrier(0);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this step's fragment reads are done before the next barrier
    }
  };
  if constexpr (BNIN) {
    // both halves unrolled: the registers of half 1's pending pieces (steps 0 .. 4 of h = 0) are then dead for the rest of the loop instead of
    // live around its back edge
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int t = 0; t < 9; ++t) kstep(h, t);
  } else {
#pragma unroll 1
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int t = 0; t < 9; ++t) kstep(h, t);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");  // LDS is reused by the reduction below

  // ---- epilogue: lane holds position m0 + wm*64 + mi*16 + pl, channels ch0 + 32 j .. + 7 (j = 0, 1) --------------------------------
  float s1[2][8], s2[2][8], fsc[2][8], fsh[2][8];
  if constexpr (MODE != 0) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 8; ++e) s1[j][e] = s2[j][e] = 0.f;
  }
  if constexpr (MODE == 2) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const float4 a = p.relu ? *reinterpret_cast<const float4*>(p.fscale + ch0 + 32 * j + 4 * i) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 b = p.relu ? *reinterpret_cast<const float4*>(p.fshift + ch0 + 32 * j + 4 * i) : make_float4(1.f, 1.f, 1.f, 1.f);
        fsc[j][4 * i] = a.x; fsc[j][4 * i + 1] = a.y; fsc[j][4 * i + 2] = a.z; fsc[j][4 * i + 3] = a.w;
        fsh[j][4 * i] = b.x; fsh[j][4 * i + 1] = b.y; fsh[j][4 * i + 2] = b.z; fsh[j][4 * i + 3] = b.w;  // no ReLU: y * 0 + 1 > 0 is always open
      }
  }
  if constexpr (MODE != 2 || !SH_R128_YPF) decode_rows();
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const f32x4 lo = acc[mi][2 * j], hi = acc[mi][2 * j + 1];
      const float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      uint4 o;
      o.x = pack_bf16x2(v[0], v[1]);
      o.y = pack_bf16x2(v[2], v[3]);
      o.z = pack_bf16x2(v[4], v[5]);
      o.w = pack_bf16x2(v[6], v[7]);
      if (oks[mi]) *reinterpret_cast<uint4*>(p.out + (unsigned long long)pixs[mi] * 128 + ch0 + 32 * j) = o;
      if constexpr (MODE == 1) {  // BN partial statistics of the fp32 results
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float ve = oks[mi] ? v[e] : 0.f;
          s1[j][e] += ve;
          s2[j][e] = __builtin_fmaf(ve, ve, s2[j][e]);  // spelled out: the instantiations must agree bit for bit
        }
      }
      if constexpr (MODE == 2) {  // BN-backward sums of the previous unit: g = stored gradient * relu'(y), sums of g and g * y
        const unsigned y4[4] = {yq[mi][j].x, yq[mi][j].y, yq[mi][j].z, yq[mi][j].w};
        const unsigned w4[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int hh = 0; hh < 2; ++hh) {
            const int e = 2 * i + hh;
            const float yy = hh == 0 ? h16_lo(y4[i]) : h16_hi(y4[i]);
            const float gq = hh == 0 ? h16_lo(w4[i]) : h16_hi(w4[i]);
            const bool on = oks[mi] && yy * fsc[j][e] + fsh[j][e] > 0.f;
            const float gv = on ? gq : 0.f;
            s1[j][e] += gv;
            s2[j][e] += gv * yy;
          }
      }
    }
  }
  if constexpr (MODE != 0) {
    float* red = reinterpret_cast<float*>(smem);  // [4 (wm)][2][128]
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float t1 = row16_sum_r128(s1[j][e]), t2 = row16_sum_r128(s2[j][e]);
        if (li == 0) {
          red[(wm * 2 + 0) * BN + ch0 + 32 * j + e] = t1;
          red[(wm * 2 + 1) * BN + ch0 + 32 * j + e] = t2;
        }
      }
    __syncthreads();
    if (tid < 2 * BN) {
      const int which = tid >> 7, c = tid & 127;
      p.partial[((long long)tile * 2 + which) * BN + c] = (red[(0 * 2 + which) * BN + c] + red[(1 * 2 + which) * BN + c]) +
                                                         (red[(2 * 2 + which) * BN + c] + red[(3 * 2 + which) * BN + c]);
    }
  }
}

// ---- 3x3 / STRIDE 2 data gradient with 128 channels (the stage-2 entry block's conv2: dy 28 x 28 -> dx 56 x 56) on the same ring ------------
// A stride-2 data gradient is four stride-1 problems over the OUTPUT (dy) grid, one per parity class (p, u) of the dx pixel
// (2 a + p, 2 b + u): x row i = 2 o - 1 + r gives r = i + 1 - 2 o, so p = 0 takes filter row 1 at dy row a, p = 1 filter row 0 at dy row
// a + 1 and filter row 2 at dy row a (columns alike): 1 + 2 + 2 + 4 = 9 tap visits with shifts 0 / +1 -- the MFMA work of the stride-1
// layer one stage later (conv3x3_r128_kernel<2>, 496 us at 2048 x 28^2).  The parity-class launches of the tile kernel ran the same work in
// 1039 us: 50 176 tiles with 2 - 8 k-steps each, two barriers per k-step, every tile re-staging its dy rows per tap.  Here a block owns 256
// padded dy positions: the dy tile (+ halo) is staged ONCE, the four classes run back to back over it (2 + 4 + 4 + 8 = 18 k-steps of
// [tap x 64 channels], weights streamed per k-step as in the stride-1 kernel), each class ends in its own store epilogue into its
// parity plane of dx, issued two stores per k-step UNDER the next class's MFMAs (second accumulator set).  Stores go to a sink for pad
// positions, so every wave issues exactly 8 per class and the counted vmcnt over the in-order queue (weights two steps ahead, the store
// pairs of the last two steps allowed in flight) stays exact.
__device__ uint4 g_r128_sink[64 * 8 * 8];   // [wave-lane][mi, j]: 16-B slots nobody reads

struct R128S2Step { int t, h, dr, ds, last, cls; };
__device__ constexpr R128S2Step kS2Steps[18] = {
    // class 0 (p 0, u 0): tap (1, 1)
    {4, 0, 0, 0, 0, 0}, {4, 1, 0, 0, 1, 0},
    // class 1 (p 0, u 1): taps (1, 0) at column b + 1, (1, 2) at column b
    {3, 0, 0, 1, 0, 1}, {5, 0, 0, 0, 0, 1}, {3, 1, 0, 1, 0, 1}, {5, 1, 0, 0, 1, 1},
    // class 2 (p 1, u 0): taps (0, 1) at row a + 1, (2, 1) at row a
    {1, 0, 1, 0, 0, 2}, {7, 0, 0, 0, 0, 2}, {1, 1, 1, 0, 0, 2}, {7, 1, 0, 0, 1, 2},
    // class 3 (p 1, u 1): taps (0, 0), (0, 2), (2, 0), (2, 2)
    {0, 0, 1, 1, 0, 3}, {2, 0, 1, 0, 0, 3}, {6, 0, 0, 1, 0, 3}, {8, 0, 0, 0, 0, 3},
    {0, 1, 1, 1, 0, 3}, {2, 1, 1, 0, 0, 3}, {6, 1, 0, 1, 0, 3}, {8, 1, 0, 0, 1, 3}};

__global__ __launch_bounds__(512, 1) void conv3x3_r128_s2dgrad_kernel(R128Args p) {
  constexpr int BM = 256, BN = 128, MI = 4, NI = 4, HALO = 32;
  constexpr int RROWS = BM + 2 * HALO;  // 320 ring rows: positions m0 - 32 .. m0 + 288
  constexpr int HALF = RROWS * 128;
  constexpr int A_BYTES = 2 * HALF;
  constexpr int BST = BN * 128;
  constexpr int NST = 3;
  constexpr int NK = 18;
  __shared__ __attribute__((aligned(16))) char smem[A_BYTES + NST * BST];

  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int li = lane & 15, g = lane >> 4;
  const int wm = wave >> 1, wn = wave & 1;
  const int pl = (li & 4) == ((li & 8) >> 1) ? 2 * ((li & 3) + ((li >> 3) << 2)) : 2 * (li - 4) + 1;
  const int WP = p.W + 1;   // p.H, p.W: the dy grid (the ring's grid); dx is 2 H x 2 W
  int tile;
  {
    const int nblk = gridDim.x, q8 = nblk >> 3, r8 = nblk & 7, xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + j;
  }
  const long long m0 = (long long)tile * BM;

  auto dma16 = [](const void* src, unsigned lds_addr) __attribute__((always_inline)) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_addr), "v"(src));
  };
  const unsigned smem_addr = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  const int slot = lane & 7, r8l = lane >> 3;
  const char* zsrc = reinterpret_cast<const char*>(g_r128_zero_page) + slot * 16;

  auto fill_half = [&](int h) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int rb = wave + 8 * i;
      const int j = rb * 8 + r8l;
      const long long q = m0 - HALO + j;
      const bool in = q >= 0 && q < p.q_total;
      const unsigned qu = in ? (unsigned)q : 0u;
      const unsigned img = fdiv(qu, p.div_pp);
      const unsigned rem = qu - img * p.div_pp.d;
      const unsigned hp = fdiv(rem, p.div_wp);
      const unsigned wp = rem - hp * p.div_wp.d;
      const bool ok = in && hp - 1u < (unsigned)p.H && wp - 1u < (unsigned)p.W;
      const unsigned pix = (img * (unsigned)p.H + (hp - 1u)) * (unsigned)p.W + (wp - 1u);
      const int chunk = slot ^ ((j >> 1) & 7);
      const char* src = ok ? reinterpret_cast<const char*>(p.x + (unsigned long long)pix * 128 + h * 64 + chunk * 8) : zsrc;
      dma16(src, smem_addr + h * HALF + rb * 8 * 128);
    }
  };
  auto key_b = [](int row) __attribute__((always_inline)) -> int { return ((row >> 1) & 1) | (((row >> 3) & 3) << 1); };
  const int lrow = wave * 8 + r8l;
  const char* wb = reinterpret_cast<const char*>(p.w + (long long)lrow * (9 * 128) + (slot ^ key_b(lrow)) * 8);
  // weight tile of k-step ks: rows lrow, lrow + 64 of [128 dx channels][64 dy channels] of tap t, half h (past the end: the zero page)
  auto dma_w = [&](int ks, int t, int h) __attribute__((always_inline)) {
    const bool live = ks < NK;
    const unsigned dst = smem_addr + A_BYTES + (unsigned)(ks % NST) * BST + wave * 8 * 128;
    const char* s0 = wb + (t * 128 + h * 64) * 2;
    dma16(live ? s0 : zsrc, dst);
    dma16(live ? s0 + 64ll * (9 * 128) * 2 : zsrc, dst + 64 * 128);
  };

  // prologue: BOTH halves of the dy tile (class 0 needs the second one in its second k-step), then the weights of steps 0 and 1
  fill_half(0);
  fill_half(1);
  dma_w(0, kS2Steps[0].t, kS2Steps[0].h);
  dma_w(1, kS2Steps[1].t, kS2Steps[1].h);

  const int a_base = (wm * 64) * 128;
  const int rowb0 = wn * 64 + (li >> 2) * 8 + (li & 3);
  const int fbo = rowb0 * 128 + ((g ^ key_b(rowb0)) * 16);

  // positions -> dx pixels of class (0, 0) for this lane's output rows (pad positions: the sink)
  unsigned pix0[MI];
  bool oks[MI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const long long q = m0 + wm * 64 + mi * 16 + pl;
    const bool in = q < p.q_total;
    const unsigned qu = in ? (unsigned)q : 0u;
    const unsigned img = fdiv(qu, p.div_pp);
    const unsigned rem = qu - img * p.div_pp.d;
    const unsigned hp = fdiv(rem, p.div_wp);
    const unsigned wp = rem - hp * p.div_wp.d;
    oks[mi] = in && hp - 1u < (unsigned)p.H && wp - 1u < (unsigned)p.W;
    pix0[mi] = (img * (unsigned)(2 * p.H) + 2u * (hp - 1u)) * (unsigned)(2 * p.W) + 2u * (wp - 1u);
  }
  const int ch0 = wn * 64 + g * 8;
  char* sink = reinterpret_cast<char*>(g_r128_sink) + (size_t)lane * 8 * 16;

  // TWO accumulator sets: class c accumulates into set c & 1 while the finished class c - 1 leaves set (c - 1) & 1 two stores per k-step
  // (a store instruction of 64 x 16 B occupies the CU's memory pipeline ~70 cycles: eight of them back to back per wave stalled every
  // class end for ~2 us with the matrix pipes idle -- 770 us per launch in that first form)
  f32x4 acc[2][MI][NI];
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) acc[b][mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // store n (0..7) of finished class cls: output row mi = n >> 1, channel group j = n & 1; pad positions -> the sink (exact store counts)
  auto store_one = [&](int cls, int n) __attribute__((always_inline)) {
    const int b = cls & 1, mi = n >> 1, j = n & 1;
    const unsigned poff = (unsigned)((cls >> 1) * (2 * p.W) + (cls & 1));
    const f32x4 lo = acc[b][mi][2 * j], hi = acc[b][mi][2 * j + 1];
    uint4 o;
    o.x = pack_bf16x2(lo[0], lo[1]);
    o.y = pack_bf16x2(lo[2], lo[3]);
    o.z = pack_bf16x2(hi[0], hi[1]);
    o.w = pack_bf16x2(hi[2], hi[3]);
    char* dst = oks[mi] ? reinterpret_cast<char*>(p.out + (unsigned long long)(pix0[mi] + poff) * 128 + ch0 + 32 * j) : sink + (mi * 2 + j) * 16;
    *reinterpret_cast<uint4*>(dst) = o;
    acc[b][mi][2 * j] = (f32x4){0.f, 0.f, 0.f, 0.f};       // the set is clean again when class cls + 2 takes it
    acc[b][mi][2 * j + 1] = (f32x4){0.f, 0.f, 0.f, 0.f};
  };

#pragma unroll
  for (int ks = 0; ks < NK; ++ks) {
    const R128S2Step st = kS2Steps[ks];
    const int cur = st.cls & 1;
    // the class that finished before this one leaves in steps first + 0 .. first + 3 of this class, two stores per step (n = 2 i, 2 i + 1)
    const int first = st.cls == 1 ? 2 : (st.cls == 2 ? 6 : 10);
    const int si = st.cls >= 1 && ks - first >= 0 && ks - first < 4 ? ks - first : -1;   // this step's store pair, or none
    // vector-memory operations retire in order.  Behind the weights of step ks the queue may hold: the weights of step ks + 1 (2) and the
    // store pairs of steps ks - 1 and ks - 2 (steps 2 .. 13 carry one)
    const int allow = 2 + (ks - 1 >= 2 && ks - 1 <= 13 ? 2 : 0) + (ks - 2 >= 2 && ks - 2 <= 13 ? 2 : 0);
    if (allow == 2) asm volatile("s_waitcnt vmcnt(2)\n\ts_barrier" ::: "memory");
    else if (allow == 4) asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(6)\n\ts_barrier" ::: "memory");
    {
      const int kn = ks + 2 < NK ? ks + 2 : 0;
      dma_w(ks + 2, kS2Steps[kn].t, kS2Steps[kn].h);
    }
    const char* sa = smem + st.h * HALF + a_base;
    const int ob = A_BYTES + (ks % NST) * BST + fbo;
    uint4 fb[2][NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int on = ob + ((ni >> 1) * 32 + (ni & 1) * 4) * 128;
      fb[0][ni] = *reinterpret_cast<const uint4*>(smem + on);
      fb[1][ni] = *reinterpret_cast<const uint4*>(smem + (on ^ 64));
    }
    const int trow = HALO + pl + st.dr * WP + st.ds;
    const int ra = trow * 128 + ((g ^ ((trow >> 1) & 7)) * 16);
    uint4 fa[2][2];
    fa[0][0] = *reinterpret_cast<const uint4*>(sa + ra);
    fa[0][1] = *reinterpret_cast<const uint4*>(sa + (ra ^ 64));
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      if (mi + 1 < MI) {
        fa[(mi + 1) & 1][0] = *reinterpret_cast<const uint4*>(sa + (mi + 1) * 16 * 128 + ra);
        fa[(mi + 1) & 1][1] = *reinterpret_cast<const uint4*>(sa + (mi + 1) * 16 * 128 + (ra ^ 64));
      }
      if (si >= 0 && (mi == 1 || mi == 3)) store_one(st.cls - 1, 2 * si + (mi >> 1));   // under the MFMAs of this row group
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) acc[cur][mi][ni] = sh_mfma16(fb[0][ni], fa[mi & 1][0], acc[cur][mi][ni]);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) acc[cur][mi][ni] = sh_mfma16(fb[1][ni], fa[mi & 1][1], acc[cur][mi][ni]);
      __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  // the last class (3) leaves at the end; its stores drain under the next block's prologue
#pragma unroll
  for (int n = 0; n < 8; ++n) store_one(3, n);
}

static hook_t g_use_r128{-1};  // -1 = simhand_test_switch(SH_SW_R128) (default on), 0 / 1 forced
void r128_enable(int on) { g_use_r128 = on < 0 ? -1 : (on ? 1 : 0); }
void hooks_reset_r128() { g_use_r128 = -1; }

bool r128_supported(int dtype, int cin, int cout, int r, int s, int stride, int pad, int w, long long q_total) {
  const int env = sw(SH_SW_R128);
  const int h = g_use_r128;
  return (h >= 0 ? h : env) && dtype == SH_BF16 && cin == 128 && cout == 128 && r == 3 && s == 3 && stride == 1 && pad == 1 && w + 2 <= 32 &&
         q_total < (1ll << 31) && q_total >= 256 * 64;
}

int r128_blocks(long long q_total) { return (int)((q_total + 255) / 256); }

// the stride-2 data gradient: dy grid H x W (a0.H, a0.W), dx 2 H x 2 W
bool r128_s2dgrad_supported(int dtype, int cin, int cout, int r, int s, int stride, int pad, int h, int w, int ho, int wo, long long q_total) {
  const int env = sw(SH_SW_R128);
  const int hk = g_use_r128;
  return (hk >= 0 ? hk : env) && dtype == SH_BF16 && cin == 128 && cout == 128 && r == 3 && s == 3 && stride == 2 && pad == 1 && h == 2 * ho &&
         w == 2 * wo && wo + 2 <= 32 && q_total < (1ll << 31) && q_total >= 256 * 64 && (long long)q_total * 4 < (1ll << 31);
}

int launch_r128_s2dgrad(const R128Args& a0, hipStream_t s) {
  R128Args a = a0;
  a.tiles = r128_blocks(a.q_total);
  route_hit(SH_ROUTE_R128_DGRAD);
  conv3x3_r128_s2dgrad_kernel<<<a.tiles, 512, 0, s>>>(a);
  return 0;
}

int launch_r128(const R128Args& a0, hipStream_t s) {
  R128Args a = a0;
  a.tiles = r128_blocks(a.q_total);
  route_hit(a.dgrad ? SH_ROUTE_R128_DGRAD : SH_ROUTE_R128_FWD);
  if (a.in_scale != nullptr) {
    route_hit(SH_ROUTE_FWD_BNIN);
    if (a.partial == nullptr) conv3x3_r128_kernel<0, true><<<a.tiles, 512, 0, s>>>(a);
    else conv3x3_r128_kernel<1, true><<<a.tiles, 512, 0, s>>>(a);
  } else if (a.partial == nullptr) conv3x3_r128_kernel<0><<<a.tiles, 512, 0, s>>>(a);
  else if (!a.dgrad) conv3x3_r128_kernel<1><<<a.tiles, 512, 0, s>>>(a);
  else conv3x3_r128_kernel<2><<<a.tiles, 512, 0, s>>>(a);
  return 0;
}

}  // namespace sh
