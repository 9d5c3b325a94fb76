// 3x3 / stride 1 / pad 1 convolution with 64 input and 64 output channels (bf16): forward and data gradient of the
// first residual stage of ResNet-50/101/152 at 56 x 56 -- the layer where the generic implicit GEMM is worst (a 128 x 64
// tile re-stages 24 KB of operands per 64 MFMAs and runs at ~0.5 PFLOP/s against a 0.33 ms HBM bound).
//
// Replaces (reference): layer1.*.conv2 of torchvision's Bottleneck (src/models/resnet_model.py:13-58) and its
// input-gradient (cuDNN there).
//
// Same idea as wgrad3x3_kernel (conv_wgrad.hip): work on the ZERO-PADDED pixel grid with shared padding ((H+1) x (W+1) per image, images back to
// back), where tap (r, s) is the constant row shift (r-1)(W+2) + (s-1).  A persistent block walks a range of padded pixels
// in steps of 64: the x rows go global -> LDS ring once (LDS-DMA, two steps ahead, zero page for pad positions), the nine
// tap operands are read from the ring at nine row offsets -- and the WEIGHTS never move: with K = 64 the whole
// 64 x 9 x 64 filter is 73 KB, each wave keeps its 32-channel slice of it (36 fragments = 144 VGPRs) in registers for
// the lifetime of the block.  Per step a wave issues 72 MFMAs against 36 ds_read_b128; nothing is staged per tap.
//   block = 4 waves as 2 (pixels) x 2 (channels): a wave owns 32 of the step's 64 pixels x 32 channels;
//   ring = 512 rows x 128 B, 16-B chunks XOR-swizzled by row/2 on the DMA source side (conflict-free fragment reads);
//   outputs at pad positions are computed and dropped ((H+1)(W+1)/(HW) = 3.6 % extra MFMAs at 56 x 56);
//   forward: BN partial sums of the fp32 results ride in registers across the block's steps (one [2][64] row per block);
//   data gradient: optionally the previous unit's BN-backward sums (sum g, sum g*y, ReLU mask recomputed from y).
#include "conv3x3_c64.h"

namespace sh {

__device__ uint4 g_c64_zero_page[8];
// results at pad positions are stored here: every wave then issues exactly two stores per step, which the counted
// s_waitcnt of the DMA ring relies on (a skipped store would shift the count)
__device__ uint4 g_c64_sink[64 * 64];
// BNIN: pad positions fetch NaNs instead of zeros -- NaN * scale + shift is NaN and the ReLU (v > 0 ? v : 0) turns it into the exact zero a
// pad position has to hold after the activation, whatever the channel's scale and shift are
#ifdef SH_H16_FP16
#define SH_C64_NAN2 0x7e007e00u
#else
#define SH_C64_NAN2 0x7fc07fc0u
#endif
__device__ uint4 g_c64_nan_page[8] = {
    {SH_C64_NAN2, SH_C64_NAN2, SH_C64_NAN2, SH_C64_NAN2}, {SH_C64_NAN2, SH_C64_NAN2, SH_C64_NAN2, SH_C64_NAN2},
    {SH_C64_NAN2, SH_C64_NAN2, SH_C64_NAN2, SH_C64_NAN2}, {SH_C64_NAN2, SH_C64_NAN2, SH_C64_NAN2, SH_C64_NAN2},
    {SH_C64_NAN2, SH_C64_NAN2, SH_C64_NAN2, SH_C64_NAN2}, {SH_C64_NAN2, SH_C64_NAN2, SH_C64_NAN2, SH_C64_NAN2},
    {SH_C64_NAN2, SH_C64_NAN2, SH_C64_NAN2, SH_C64_NAN2}, {SH_C64_NAN2, SH_C64_NAN2, SH_C64_NAN2, SH_C64_NAN2}};

typedef __attribute__((ext_vector_type(8))) __bf16 c64_frag_t;

__device__ __forceinline__ float row16_sum_c64(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));  // row_ror:8
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));  // row_ror:4
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));  // row_ror:2
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));  // row_ror:1
  return v;
}

// MODE 0: store only; 1: forward + BN partial statistics; 2: data gradient + BN-backward sums of the previous unit
// BNIN (forward): x is the previous unit's RAW conv output; every wave rewrites the ring rows it fetched itself as relu(x * in_scale + in_shift)
// (its own s_waitcnt is all the synchronisation that takes: the rewrite of chunk j + 2 runs at the end of step j, one barrier before its first
// reader) and stores them to a_out on the way
template <int MODE, bool BNIN = false>
__global__ __launch_bounds__(256, 2) void conv3x3_c64_kernel(C64Args p) {
  static_assert(!BNIN || MODE != 2, "BNIN is a forward form");
  constexpr int RING = 512;  // ring rows (8 chunks of 64)
  constexpr int D = 2;       // DMA distance in steps
  __shared__ __attribute__((aligned(16))) char ring[RING * 128];
  __shared__ float red[2][2][64];
  __shared__ __attribute__((aligned(16))) float s_coef[2][64];

  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int li = lane & 15, g = lane >> 4;
  // pixel of MFMA column li inside a 16-pixel group.  A ds_read_b128 is served in four groups of 16 lanes that are NOT
  // contiguous: {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... -- columns {0-3, 12-15} of k-chunk g next to columns 4-11 of
  // chunk g ^ 1.  Tap offsets shift the rows by arbitrary amounts, so no row-keyed XOR alone keeps such a group on 16 distinct
  // bank quads; giving the first column set the even pixels and the second the odd ones does (row parity picks the
  // 128-B half of the 256-B bank line, the row/2 key separates the 8 rows of one parity) for every offset.
  const int pl = (li & 4) == ((li & 8) >> 1) ? 2 * ((li & 3) + ((li >> 3) << 2)) : 2 * (li - 4) + 1;
  const int wm = wave >> 1, wn = wave & 1;
  const int WP = p.W + 1;  // SHARED padding as in wgrad3x3_kernel: one zero column between rows, one zero row between images
  const long long total_steps = (p.q_total + 63) / 64;
  const long long step0 = (long long)blockIdx.x * p.steps_per_block;
  long long step1 = step0 + p.steps_per_block;
  if (step1 > total_steps) step1 = total_steps;
  const int nsteps = step0 < step1 ? (int)(step1 - step0) : 0;
  const long long q0 = step0 * 64;

  // ---- weights: this wave's 32 destination channels, all taps, whole K, resident in registers -----------------------------
  // fragment row m of tile ni <-> destination channel wn*32 + (m >> 2)*8 + ni*4 + (m & 3): a lane's accumulator registers of the
  // two tiles are then 8 CONSECUTIVE channels (one 16-B store per pixel)
  uint4 wf[9][2][2];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        const int ch = wn * 32 + (li >> 2) * 8 + ni * 4 + (li & 3);
        wf[t][kk][ni] = *reinterpret_cast<const uint4*>(p.w + ((long long)ch * 9 + t) * 64 + kk * 32 + g * 8);
      }

  // ---- ring DMA: chunk c = padded pixels q0 + 64 c .. +64 -> ring rows ((c + 1) * 64 ..) & 511; wave w fills rows 16w..16w+15
  // of the chunk with two instructions (8 rows each); lane l = row (l >> 3), 16-B slot l & 7, source chunk slot ^ key(row) ----
  auto dma16 = [](const void* src, unsigned lds_addr) __attribute__((always_inline)) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_addr), "v"(src));
  };
  const unsigned ring_addr = (unsigned)(size_t)(__attribute__((address_space(3))) char*)ring;
  const int slot = lane & 7;
  const char* zsrc = reinterpret_cast<const char*>(BNIN ? g_c64_nan_page : g_c64_zero_page) + slot * 16;
  // off[h] (BNIN): element offset of the 16 bytes the lane fetched (a_out gets the rewritten ones at the same place), ~0 for a pad position
  auto dma_chunk = [&](int c, unsigned (&off)[2]) __attribute__((always_inline)) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int row = wave * 16 + h * 8 + (lane >> 3);          // row inside the 64-row chunk
      const int q = (int)q0 + c * 64 + row;                      // q_total < 2^31 (checked on the host)
      const bool in = q >= 0 && q < (int)p.q_total;
      const unsigned qu = in ? (unsigned)q : 0u;
      const unsigned img = fdiv(qu, p.div_pp);
      const unsigned rem = qu - img * p.div_pp.d;
      const unsigned hp = fdiv(rem, p.div_wp);
      const unsigned wp = rem - hp * p.div_wp.d;
      const bool ok = in && hp - 1u < (unsigned)p.H && wp - 1u < (unsigned)p.W;
      const unsigned pix = (img * (unsigned)p.H + (hp - 1u)) * (unsigned)p.W + (wp - 1u);
      int chunk = slot ^ ((row >> 1) & 7);                      // ring row = 64 (c + 1) + row: same key
      // opaque to the optimiser: left visible, `p.x + chunk * 8` is hoisted out of the step loop as two 64-bit VGPR pairs, which the data-gradient
      // instantiation (256 registers: the filter alone is 144) spills -- and behind a scratch reload hipcc waits vmcnt(0), draining the counted DMA queue
      asm volatile("" : "+v"(chunk));
      const char* src = ok ? reinterpret_cast<const char*>(p.x + (unsigned long long)pix * 64 + chunk * 8) : zsrc;
      dma16(src, ring_addr + ((((c + 1) * 64) & (RING - 1)) + wave * 16 + h * 8) * 128);
      if constexpr (BNIN) off[h] = ok ? pix * 64u + (unsigned)chunk * 8u : 0xffffffffu;  // 32-bit element offset: n*h*w*cin < 2^32 (simhand_conv2d_fwd_bnin_ok, re-checked in launch_c64_conv)
    }
  };
  // BNIN: rewrite the lane's own 2 x 16 bytes of chunk c in place; own = the chunk belongs to this block's range (else a neighbour stores it)
  auto bn_chunk = [&](int c, const unsigned (&off)[2], bool own) __attribute__((always_inline)) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int row = wave * 16 + h * 8 + (lane >> 3);
      const int chunk = slot ^ ((row >> 1) & 7);
      char* at = ring + ((((c + 1) * 64) & (RING - 1)) + row) * 128 + slot * 16;
      const uint4 v = *reinterpret_cast<const uint4*>(at);
      const unsigned w4[4] = {v.x, v.y, v.z, v.w};
      float sc[8], sh[8], o[8];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const float4 a = *reinterpret_cast<const float4*>(&s_coef[0][chunk * 8 + 4 * i]);
        const float4 b = *reinterpret_cast<const float4*>(&s_coef[1][chunk * 8 + 4 * i]);
        sc[4 * i] = a.x; sc[4 * i + 1] = a.y; sc[4 * i + 2] = a.z; sc[4 * i + 3] = a.w;
        sh[4 * i] = b.x; sh[4 * i + 1] = b.y; sh[4 * i + 2] = b.z; sh[4 * i + 3] = b.w;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {  // the arithmetic of bn_apply_kernel (bn.hip), bit for bit
        o[2 * i] = h16_lo(w4[i]) * sc[2 * i] + sh[2 * i];
        o[2 * i + 1] = h16_hi(w4[i]) * sc[2 * i + 1] + sh[2 * i + 1];
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = o[e] > 0.f ? o[e] : 0.f;
      uint4 r;
      r.x = pack_bf16x2(o[0], o[1]);
      r.y = pack_bf16x2(o[2], o[3]);
      r.z = pack_bf16x2(o[4], o[5]);
      r.w = pack_bf16x2(o[6], o[7]);
      *reinterpret_cast<uint4*>(at) = r;
      uint4* dst = own && off[h] != 0xffffffffu ? reinterpret_cast<uint4*>(p.a_out + off[h]) : &g_c64_sink[(blockIdx.x & 63) * 64 + lane];
      *dst = r;  // always one store: the counted waits below rely on it
    }
  };

  // ---- fragment addressing: pixel row of lane = 64 (j + 1) + wm*32 + mi*16 + pl + off_t; key and tap offsets do not depend on j ----
  // ONE register per tap: ring byte offset of the lane's row (row * 128) plus its swizzled 16-B column (bits 4..6) -- kept as separate row / column
  // arrays the eighteen values pushed the data-gradient instantiation over its 256 registers (scratch reloads inside the step loop)
  int tadd[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    int off = (t / 3 - 1) * WP + (t % 3 - 1);
    if (p.dgrad) off = -off;  // dx[p] = sum_t dy[p - off_t] W[.][t][.]
    const int trow = 64 + wm * 32 + pl + off;      // >= 64 - 59 > 0
    tadd[t] = trow * 128 + (g ^ ((trow >> 1) & 7)) * 16;  // chunk kk*4 + g of the k-step: kk = 1 flips bit 6
  }

  // output / statistics state
  const int ch0 = wn * 32 + g * 8;  // this lane's 8 consecutive channels
  float s1[8], s2[8];
  if constexpr (MODE != 0) {
#pragma unroll
    for (int e = 0; e < 8; ++e) s1[e] = s2[e] = 0.f;
  }
  if constexpr (MODE == 2) {  // ReLU gate coefficients live in LDS (the registers hold the filter); visible after the loop's first barrier
    if (tid < 64) {
      s_coef[0][tid] = p.relu ? p.fscale[tid] : 0.f;
      s_coef[1][tid] = p.relu ? p.fshift[tid] : 1.f;  // no ReLU: the gate y * 0 + 1 > 0 is always open
    }
  }
  // padded pixel of this lane's output row mi of step j -> (NHWC pixel index, is it a real pixel)
  auto decode = [&](int j, int mi, unsigned& pix) __attribute__((always_inline)) -> bool {
    const unsigned qraw = (unsigned)q0 + (unsigned)(j * 64 + wm * 32 + mi * 16 + pl);  // q_total < 2^31
    const bool in = qraw < (unsigned)p.q_total;
    const unsigned qu = in ? qraw : 0u;
    const unsigned img = fdiv(qu, p.div_pp);
    const unsigned rem = qu - img * p.div_pp.d;
    const unsigned hp = fdiv(rem, p.div_wp);
    const unsigned wp = rem - hp * p.div_wp.d;
    pix = (img * (unsigned)p.H + (hp - 1u)) * (unsigned)p.W + (wp - 1u);
    return in && hp - 1u < (unsigned)p.H && wp - 1u < (unsigned)p.W;  // else a pad position
  };

  // prologue: chunks -1 .. 2 in the order the counted waits assume
  unsigned off_cur[2] = {0u, 0u};
  if constexpr (BNIN) {
    if (tid < 64) {
      s_coef[0][tid] = p.in_scale[tid];
      s_coef[1][tid] = p.in_shift[tid];
    }
    __syncthreads();
    unsigned o0[2], o1[2], o2[2];
    dma_chunk(-1, o0);
    dma_chunk(0, o1);
    dma_chunk(1, o2);
    dma_chunk(2, off_cur);
    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");  // the lane's own rows of chunks -1 .. 1 have landed
    bn_chunk(-1, o0, false);
    bn_chunk(0, o1, nsteps > 0);
    bn_chunk(1, o2, nsteps > 1);
  } else {
#pragma unroll
    for (int c = -1; c <= D; ++c) dma_chunk(c, off_cur);
  }

  for (int j = 0; j < nsteps; ++j) {
    unsigned off_new[2];
    if constexpr (BNIN) {
      // chunk j + 1 was fetched AND rewritten by its owners before they arrived here (end of step j - 1 / prologue): the barrier only needs their
      // LDS writes to be done
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      dma_chunk(j + 1 + D, off_new);
    } else {
    // outstanding, in issue order: DMA chunk j + 1 (2 per wave), stores of step j - 2 (2), DMA chunk j + 2 (2), stores of step
    // j - 1 (2); vector-memory operations retire in order, so <= 6 outstanding means chunk j + 1 has landed.  After the
    // barrier it is visible to every wave and every wave is done with step j - 1
    // (steps 0 and 1 have no older stores in the queue yet: the stricter count is the safe one there)
    if (j < 2) asm volatile("s_waitcnt vmcnt(2)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(6)\n\ts_barrier" ::: "memory");
    dma_chunk(j + 1 + D, off_new);  // ring slot (j + 4) & 7: outside the window (chunks j - 1 .. j + 1) and the chunk in flight (j + 2)
    }

    // MODE 2: this step's rows of the previous unit's conv output are fetched now and consumed in the epilogue
    unsigned pixe[2];
    bool vale[2];
    uint4 yq[2];
    if constexpr (MODE == 2) {
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        vale[mi] = decode(j, mi, pixe[mi]);
        const void* src = vale[mi] ? (const void*)(p.fy + (unsigned long long)pixe[mi] * 64 + ch0) : (const void*)g_c64_zero_page;
        yq[mi] = *reinterpret_cast<const uint4*>(src);
      }
    }

    f32x4 acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // fragments of tap t + 1 are read while the MFMAs of tap t run (two register sets)
    const int jb = (j * 64 * 128) & (RING * 128 - 1);
    uint4 fr[2][4];
    auto load_tap = [&](int t, uint4 (&f)[4]) __attribute__((always_inline)) {
      const int b = jb + tadd[t];  // (the column bits ride below the ring mask's lowest row bit)
      const int a0 = b & (RING * 128 - 1), a1 = (b + 16 * 128) & (RING * 128 - 1);  // rows +16: same key
      f[0] = *reinterpret_cast<const uint4*>(ring + a0);
      f[1] = *reinterpret_cast<const uint4*>(ring + a1);
      f[2] = *reinterpret_cast<const uint4*>(ring + (a0 ^ 64));
      f[3] = *reinterpret_cast<const uint4*>(ring + (a1 ^ 64));
    };
    load_tap(0, fr[0]);
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      if (t + 1 < 9) load_tap(t + 1, fr[(t + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);  // keep the reads ahead of this tap's MFMAs (the scheduler would sink them to their uses)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
          for (int mi = 0; mi < 2; ++mi)
            acc[mi][ni] = sh_mfma16(wf[t][kk][ni], fr[t & 1][kk * 2 + mi], acc[mi][ni]);
      __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this step's fragment reads are done before the next barrier

    // ---- epilogue of the step: lane holds pixel q0 + 64 j + wm*32 + mi*16 + pl, channels ch0 .. ch0 + 7 ----
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      unsigned pix32;
      bool valid;
      if constexpr (MODE == 2) {
        pix32 = pixe[mi];
        valid = vale[mi];
      } else {
        valid = decode(j, mi, pix32);
      }
      const unsigned long long pix = pix32;
      const f32x4 lo = acc[mi][0], hi = acc[mi][1];
      const float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      uint4 o;
      o.x = pack_bf16x2(v[0], v[1]);
      o.y = pack_bf16x2(v[2], v[3]);
      o.z = pack_bf16x2(v[4], v[5]);
      o.w = pack_bf16x2(v[6], v[7]);
      uint4* dst = valid ? reinterpret_cast<uint4*>(p.out + pix * 64 + ch0) : &g_c64_sink[(blockIdx.x & 63) * 64 + lane];
      *dst = o;
      if constexpr (MODE == 1) {  // BN partial statistics of the fp32 results
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float ve = valid ? v[e] : 0.f;
          s1[e] += ve;
          s2[e] = __builtin_fmaf(ve, ve, s2[e]);  // spelled out: left to the compiler, the two forward instantiations contracted it differently
        }
      }
      if constexpr (MODE == 2) {
        {  // BN-backward sums of the previous unit: g = stored gradient * relu'(y), sums of g and g * y (pad rows: y = 0, g dropped)
          const unsigned y4[4] = {yq[mi].x, yq[mi].y, yq[mi].z, yq[mi].w};
          float yy[8], fsc[8], fsh[8];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            yy[2 * i] = h16_lo(y4[i]);
            yy[2 * i + 1] = h16_hi(y4[i]);
          }
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const float4 a = *reinterpret_cast<const float4*>(&s_coef[0][ch0 + 4 * i]);
            const float4 b = *reinterpret_cast<const float4*>(&s_coef[1][ch0 + 4 * i]);
            fsc[4 * i] = a.x; fsc[4 * i + 1] = a.y; fsc[4 * i + 2] = a.z; fsc[4 * i + 3] = a.w;
            fsh[4 * i] = b.x; fsh[4 * i + 1] = b.y; fsh[4 * i + 2] = b.z; fsh[4 * i + 3] = b.w;
          }
          const unsigned w4[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              const int e = 2 * i + h;
              const float gq = h == 0 ? h16_lo(w4[i]) : h16_hi(w4[i]);
              const bool on = valid && yy[e] * fsc[e] + fsh[e] > 0.f;
              const float gv = on ? gq : 0.f;
              s1[e] += gv;
              s2[e] += gv * yy[e];
            }
        }
      }
    }
    if constexpr (BNIN) {
      // issued after the DMAs of chunk j + 2 (top of step j - 1, or the prologue): stores of step j - 1 (2), rewritten rows of chunk j + 1 (2), DMAs of
      // chunk j + 3 (2), stores of step j (2) -- in-order retirement: <= 8 outstanding means the lane's rows of chunk j + 2 are in LDS
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      bn_chunk(j + 2, off_cur, j + 2 < nsteps);
      off_cur[0] = off_new[0];
      off_cur[1] = off_new[1];
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  if constexpr (MODE != 0) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float t1 = row16_sum_c64(s1[e]), t2 = row16_sum_c64(s2[e]);
      if (li == 0) {
        red[wm][0][ch0 + e] = t1;
        red[wm][1][ch0 + e] = t2;
      }
    }
    __syncthreads();
    if (tid < 128) {
      const int which = tid >> 6, c = tid & 63;
      p.partial[((long long)blockIdx.x * 2 + which) * 64 + c] = red[0][which][c] + red[1][which][c];
    }
  }
}

static hook_t g_use_c64{1};
void hooks_reset_c64() { g_use_c64 = 1; }

bool c64_supported(int dtype, int cin, int cout, int r, int s, int stride, int pad, int w, long long q_total) {
  return g_use_c64 && dtype == SH_BF16 && cin == 64 && cout == 64 && r == 3 && s == 3 && stride == 1 && pad == 1 && w + 3 <= 64 &&
         q_total < (1ll << 31) && q_total >= 64 * 64;
}

void c64_enable(int on) { g_use_c64 = on ? 1 : 0; }

// persistent blocks: ~2 per CU, at least 8 steps each
int c64_blocks(long long q_total) {
  const long long steps = (q_total + 63) / 64;
  long long b = steps / 8;
#ifndef SH_C64_BLOCKS
#define SH_C64_BLOCKS 512  // two resident blocks per CU: one round
#endif
  if (b > SH_C64_BLOCKS) b = SH_C64_BLOCKS;
  if (b < 1) b = 1;
  const long long per = (steps + b - 1) / b;
  return (int)((steps + per - 1) / per);
}

int launch_c64(const C64Args& a0, hipStream_t s) {
  C64Args a = a0;
  const long long steps = (a.q_total + 63) / 64;
  const int b = c64_blocks(a.q_total);
  a.steps_per_block = (int)((steps + b - 1) / b);
  route_hit(a.dgrad ? SH_ROUTE_C64_DGRAD : SH_ROUTE_C64_FWD);
  if (a.in_scale != nullptr) {
    route_hit(SH_ROUTE_FWD_BNIN);
    if (a.partial == nullptr) conv3x3_c64_kernel<0, true><<<b, 256, 0, s>>>(a);
    else conv3x3_c64_kernel<1, true><<<b, 256, 0, s>>>(a);
  } else if (a.partial == nullptr) conv3x3_c64_kernel<0><<<b, 256, 0, s>>>(a);
  else if (!a.dgrad) conv3x3_c64_kernel<1><<<b, 256, 0, s>>>(a);
  else conv3x3_c64_kernel<2><<<b, 256, 0, s>>>(a);
  return 0;
}

}  // namespace sh
