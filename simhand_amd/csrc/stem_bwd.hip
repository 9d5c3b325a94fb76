// Fused backward of the ResNet stem (conv1 7x7/2 -> bn1 -> relu -> maxpool 3/2/1) at 224 x 224, 16-bit storage: ONE kernel that
//   1. recomputes the raw conv1 output y of an image, two rows at a time, from the zero-padded NHWC4 input rows held in an LDS ring
//      (the forward's stem_ring_fwd_kernel arithmetic, same MFMA order: the same bits),
//   2. routes the pooled gradient dz to the conv pixels through the winner index by SCATTER: every pooled element adds its gradient to
//      exactly one pixel of an fp32 LDS image of three conv rows (ds_add_f32; windows of one parity of ow in one phase, so no two lanes
//      of a phase touch one address and the order of the <= 4 addends of a pixel is fixed: deterministic), the MFMA lanes then read their
//      pixels' sums, round them as maxpool_bwd would have stored them, gate by the ReLU and form dy = cA * g + (y * cP + cQ) -- the
//      BatchNorm-backward apply of pool_bn_bwd_blk_kernel (bn.hip) -- in registers, rounded to the storage type as that kernel stores it,
//   3. accumulates dW[64][7 x 32] += dy^T x from a 40-KB LDS image of the two dy rows and the SAME input rows (transpose reads).
// Neither y (3.3 GB at 2048 x 224^2) nor dy (3.3 GB) exists in HBM: the round-3 chain read y, wrote dy (pool_bn_bwd_blk_kernel,
// 1.50 ms) and read dy again next to the input (wgrad_kernel<.., STEM>, 1.58 ms).
//
// Replaces (reference): autograd's max_pool2d_backward -> threshold_backward -> native_batch_norm_backward -> the weight gradient of
// conv1 of torchvision's ResNet (src/models/resnet_model.py:13-26); conv1 has no data gradient (its input is the image).
//
// Shape of the kernel.  256 threads = 4 waves, ONE wave per SIMD (the register budget: 112 registers of resident filter + 96
// accumulators + 80 BatchNorm coefficients), persistent over images (block b: images b, b + grid, ...).  Iteration q = 0 .. 56 handles
// pooled row q and the conv rows 2 q - 1, 2 q (the rows whose gradient is complete once pooled row q has been scattered):
//   * everything requested in iteration q - 1 has landed (vmcnt(0)): the four input rows of this iteration, the (dz, idx) registers;
//   * scatter of pooled row q in two phases (even / odd ow; thread t owns windows (t >> 3) and 32 + (t >> 3), chunk t & 7), barrier each;
//   * LDS-DMA of the next iteration's input rows, register loads of pooled row q + 1;
//   * wave w, row r, m-tiles t = 0, 1: pixels 32 w + 16 t + li (pixels >= 112 are padding: dy = 0), all 64 channels: 2 x 28 MFMAs, then
//     its pixels' gradient sums out of the LDS image (zeroed behind the read), BatchNorm arithmetic, dy -> dybuf ([2 x 128 px][64 ch],
//     160-B rows); barrier;
//   * weight gradient of both rows: wave w owns filter rows 2 w, 2 w + 1 (row 7 is the zero row of the [64][256] layout: skipped):
//     64 channels x 32 k-elements = 8 accumulator tiles per filter row, K = 2 x 128 pixels = 8 k-steps; both operands by
//     ds_read_b64_tr_b16 (pixel = k: the permutation inside a k-step is the same for both), the x operand straight from the ring
//     (pixel px of filter row r starts at byte 16 px of ring row 2 ho + r).
// Round 4 measured the first form of this kernel (per-pixel GATHER of the <= 4 candidate windows in the MFMA lanes) at 2.95 ms: 800
// VALU instructions per wave and row, 41 % of the wave cycles issuing VALU, 23 % MFMA (profiles/r04_stem_kernels_v1.md): with one wave
// per SIMD nothing hides VALU issue.  The scatter does O(1) work per pooled element instead of 32 compare-selects per candidate window.
// The per-block partial [64][224] goes to a workspace; stem_bwd_reduce_kernel sums the blocks in a fixed order and writes the
// reference's OIHW fp32 weight.grad.
#include "conv_1x1.h"

namespace sh {

__device__ uint4 g_sb_zero_page[8];

struct StemBwdArgs {
  const bf16_t* xp;          // [n][hp][wp][4] zero-padded input
  const bf16_t* w;           // [64][256] forward weights (stem_pack_weights)
  const bf16_t* dz;          // [n][56][56][64] gradient of the pooled output
  const unsigned char* idx;  // [n][56][56][64] winner taps
  const float* scale;        // [64] BatchNorm forward scale / shift (the ReLU gate)
  const float* shift;
  const float* mean;         // [64]
  const float* invstd;
  const float* gamma;        // [64] or null (= 1)
  const float* dgamma;       // [64] finalized BatchNorm-backward sums
  const float* dbeta;
  float inv_m;               // 1 / (n * 112 * 112)
  float* part;               // [grid][64][224] fp32
  int n, hp, wp;
};

typedef __attribute__((ext_vector_type(4))) short sb_s16x4;

// MFMA operand for k = 32 pixels from an LDS image whose "row" of pixel q starts at base + q * stride: lane (p, g) receives column
// col0 + p of pixels {4 g .. 4 g + 3, 16 + 4 g .. 16 + 4 g + 3} (the k permutation both operands share)
__device__ __forceinline__ uint4 sb_frag_tr(const char* base, int stride, int col0_bytes, int lane) {
  const int p = lane & 15, g = lane >> 4;
  const char* a0 = base + (4 * g + (p >> 2)) * stride + col0_bytes + (p & 3) * 8;
  typedef sb_s16x4 __attribute__((address_space(3))) * lds_ptr;
  const sb_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(a0));
  const sb_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(a0 + 16 * stride));
  uint4 r;
  r.x = (unsigned)(unsigned short)lo[0] | ((unsigned)(unsigned short)lo[1] << 16);
  r.y = (unsigned)(unsigned short)lo[2] | ((unsigned)(unsigned short)lo[3] << 16);
  r.z = (unsigned)(unsigned short)hi[0] | ((unsigned)(unsigned short)hi[1] << 16);
  r.w = (unsigned)(unsigned short)hi[2] | ((unsigned)(unsigned short)hi[3] << 16);
  return r;
}

__global__ __launch_bounds__(256, 1) void stem_bwd_fused_kernel(StemBwdArgs p) {
  constexpr int SLOT = 2048, NSLOT = 14;            // input ring: rows 4 q - 2 .. 4 q + 10 are live in iteration q (13 consecutive rows)
  constexpr int HO = 112, WO = 112, PH = 56, PW = 56;
  constexpr int DYS = 160, DYROW = 128 * DYS;        // dy image of one conv row: 128 pixel rows of 128 B + 32 B (conflict-free transpose reads)
  constexpr int GP = 272, GROW = WO * GP;            // gradient image: 64 fp32 per pixel + 16 B (16 consecutive pixels on 16 distinct bank quads)
  __shared__ __attribute__((aligned(16))) char ring[NSLOT * SLOT + 128];   // (+ the overrun of the padding pixels 112..127)
  __shared__ __attribute__((aligned(16))) char gbuf[3 * GROW];             // conv rows r -> slot r % 3
  __shared__ __attribute__((aligned(16))) char dybuf[2 * DYROW];
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int li = lane & 15, g = lane >> 4;
  const int row_bytes = p.wp * 8;            // 1856 at 224^2
  const int nchunk = row_bytes >> 4;

  auto dma16 = [](const void* src, unsigned lds_addr) __attribute__((always_inline)) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_addr), "v"(src));
  };
  const unsigned ring_addr = (unsigned)(size_t)(__attribute__((address_space(3))) char*)ring;
  const char* zsrc = reinterpret_cast<const char*>(g_sb_zero_page);

  // ---- conv1 weights: all 64 channels x 7 filter rows, resident (fragment row li of channel tile ni <-> channel
  // (ni >> 1)*32 + (li >> 2)*8 + (ni & 1)*4 + (li & 3): accumulator registers of tiles 2j, 2j + 1 are 8 consecutive channels) ----------------
  uint4 wf[7][4];
#pragma unroll
  for (int r = 0; r < 7; ++r)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      const int ch = (ni >> 1) * 32 + (li >> 2) * 8 + (ni & 1) * 4 + (li & 3);
      wf[r][ni] = *reinterpret_cast<const uint4*>(p.w + ch * 256 + r * 32 + g * 8);
    }
  // ---- BatchNorm coefficients of the lane's channels j*32 + g*8 + e: gate (sc, sh) and dy = cA * gv + (y * cP + cQ) --------------------------
  float sc[2][8], sh[2][8], cA[2][8], cP[2][8], cQ[2][8];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int ch = j * 32 + g * 8 + e;
      sc[j][e] = p.scale[ch];
      sh[j][e] = p.shift[ch];
      const float m_ = p.mean[ch], i_ = p.invstd[ch];
      const float a_ = (p.gamma ? p.gamma[ch] : 1.0f) * i_;
      const float k2 = p.dbeta[ch] * p.inv_m, k3 = a_ * p.dgamma[ch] * p.inv_m;
      cA[j][e] = a_;
      cP[j][e] = -i_ * k3;
      cQ[j][e] = m_ * i_ * k3 - a_ * k2;
    }
  // every global load above is waited for HERE, with the builtin (the compiler's waitcnt pass sees it): a load it still counts as pending
  // on the loop's entry path would make it drain the whole vector-memory queue in every iteration
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
  // ---- weight-gradient accumulators: filter rows 2 wave + f (f = 0, 1), channel tiles mt, k-element tiles nt ----------------------------------
  f32x4 dw[2][4][2];
#pragma unroll
  for (int f = 0; f < 2; ++f)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) dw[f][mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // zero for the kernel's life: the ring's overrun pad (read as "pixels 112..127" of the last slot, always against zero dy rows: must be
  // finite), the 16 padding pixel rows 112..127 of both dy rows; the gradient image starts at zero and every reader zeroes what it read
  if (tid < 8) *reinterpret_cast<uint4*>(ring + NSLOT * SLOT + tid * 16) = make_uint4(0, 0, 0, 0);
  for (int i = tid; i < 2 * 16 * (DYS / 16); i += 256) {
    const int b = i / (16 * (DYS / 16)), r = i % (16 * (DYS / 16));
    *reinterpret_cast<uint4*>(dybuf + b * DYROW + 112 * DYS + r * 16) = make_uint4(0, 0, 0, 0);
  }
  for (int i = tid; i < 3 * GROW / 16; i += 256) *reinterpret_cast<uint4*>(gbuf + i * 16) = make_uint4(0, 0, 0, 0);

  // scatter role of this thread (t < 224): chunk sc_c of the windows ow = 2 sc_k (even phase) and 2 sc_k + 1 (odd phase) of every pooled row
  const int sc_c = tid & 7, sc_k = tid >> 3;
  const bool sc_on = sc_k < PW / 2;
  // weight gradient of the two conv rows in dybuf (row slot b <-> conv row rowa + b; a missing row's dy image is all zero):
  // 8 k-steps of 32 pixels, 16 MFMAs each (8 for wave 3: filter row 7 is padding)
  auto wgrad_rows = [&](int rowa, bool has_a, bool has_b) __attribute__((always_inline)) {
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      if (!(b == 0 ? has_a : has_b)) continue;   // uniform
      const int row = rowa + b;
      const char* dyb = dybuf + b * DYROW;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        uint4 af[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) af[mt] = sb_frag_tr(dyb + ks * 32 * DYS, DYS, mt * 32, lane);
#pragma unroll
        for (int f = 0; f < 2; ++f) {
          const int r = 2 * wave + f;
          if (r < 7) {
            const char* xr = ring + ((2 * row + r) % NSLOT) * SLOT + ks * 32 * 16;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
              const uint4 bf = sb_frag_tr(xr, 16, nt * 32, lane);
#pragma unroll
              for (int mt = 0; mt < 4; ++mt) dw[f][mt][nt] = sh_mfma16(af[mt], bf, dw[f][mt][nt]);
            }
          }
        }
      }
    }
  };

  for (int img = blockIdx.x; img < p.n; img += gridDim.x) {
    const char* xbase = reinterpret_cast<const char*>(p.xp) + (long long)img * p.hp * row_bytes;
    const bf16_t* dzimg = p.dz + (long long)img * PH * PW * 64;
    const unsigned char* iximg = p.idx + (long long)img * PH * PW * 64;
    // half hf of padded input row y -> ring slot y % NSLOT (chunks past the row's end / rows past the image: the zero page)
    auto dma_in = [&](int y, int hf) __attribute__((always_inline)) {
      const int c = 64 * hf + lane;
      const bool ok = y < p.hp && c < nchunk;
      dma16(ok ? xbase + (long long)y * row_bytes + c * 16 : zsrc, ring_addr + (unsigned)(y % NSLOT) * SLOT + hf * 1024);
    };
    // this thread's two (dz, idx) items of pooled row q: windows 2 sc_k, 2 sc_k + 1 (rows past the end / idle threads: clamped, never used)
    uint4 dzr[2];
    uint2 ixr[2];
    auto load_pooled = [&](int q) __attribute__((always_inline)) {
      const int qc = q < PH ? q : PH - 1;
      const int kc = sc_on ? sc_k : 0;
      const long long o = ((long long)qc * PW + 2 * kc) * 64 + sc_c * 8;
      dzr[0] = *reinterpret_cast<const uint4*>(dzimg + o);
      ixr[0] = *reinterpret_cast<const uint2*>(iximg + o);
      dzr[1] = *reinterpret_cast<const uint4*>(dzimg + o + 64);
      ixr[1] = *reinterpret_cast<const uint2*>(iximg + o + 64);
    };

    // ---- prologue of an image: input rows 0 .. 6 (conv row 0), pooled row 0 --------------------------------------------------------------
    __syncthreads();  // (the previous image's last weight-gradient reads of the ring / dy images are done)
    for (int k = wave; k < 14; k += 4) dma_in(k >> 1, k & 1);
    load_pooled(0);

    for (int q = 0; q <= PH; ++q) {
      const int ra = 2 * q - 1, rb = 2 * q;            // conv rows of this iteration
      const bool has_a = q >= 1, has_b = q < PH;      // uniform
      // everything requested in the previous iteration (or the prologue) has landed
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      // ---- scatter pooled row q: element e of chunk sc_c of window ow adds dz to pixel (2 q - 1 + kh, 2 ow - 1 + kw) of the image.  Plain
      // read-modify-write (LDS float atomics retire about one LANE per three cycles on this part -- the first scatter form spent 4.6 x
      // the gather form's LDS cycles in ds_add_f32): inside a phase no two lanes touch one address.  The byte offset of tap t's pixel comes
      // out of a 9-entry table held across lanes 0..8 (ds_bpermute: one crossbar read instead of a divide by 3 and two selects) ---------------
      if (has_b) {
        const int s0 = (2 * q + 2) % 3;                // slot of conv row 2 q - 1 (= (2 q - 1) mod 3)
        const int lt = lane < 9 ? lane : 0;
        const int tab = ((s0 + lt / 3) % 3) * GROW + (lt % 3) * GP;   // lane t: row slot of tap row t / 3, pixel column offset t % 3
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
          if (sc_on) {
            const unsigned d4[4] = {dzr[ph].x, dzr[ph].y, dzr[ph].z, dzr[ph].w};
            const int gb = (2 * (2 * sc_k + ph) - 1) * GP + sc_c * 32;
            int ad[8];
            float cur[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const unsigned t4 = (((e < 4 ? ixr[ph].x : ixr[ph].y) >> (8 * (e & 3))) & 0xffu) << 2;
              ad[e] = __builtin_amdgcn_ds_bpermute((int)t4, tab) + gb + e * 4;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) cur[e] = *reinterpret_cast<const float*>(gbuf + ad[e]);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float dv = (e & 1) ? h16_hi(d4[e >> 1]) : h16_lo(d4[e >> 1]);
              *reinterpret_cast<float*>(gbuf + ad[e]) = cur[e] + dv;
            }
          }
          // phase boundary: the other parity's windows overlap these by one pixel column; after the second phase: the image is complete
          asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
      } else {
        asm volatile("s_barrier" ::: "memory");  // (keeps every iteration's barrier count equal; nothing to order)
        asm volatile("s_barrier" ::: "memory");
      }
      // ---- requests of the next iteration: input rows 4 q + 7 .. 4 q + 10 (8 half rows, two per wave), pooled row q + 1 ----------------------
      dma_in(4 * q + 7 + wave, 0);
      dma_in(4 * q + 7 + wave, 1);
      load_pooled(q + 1);

      // ---- conv1 of rows ra, rb for the wave's 2 x 16 pixels; gradient sums -> dy ---------------------------------------------------------------
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        if (!(b == 0 ? has_a : has_b)) continue;   // uniform
        const int ho = b == 0 ? ra : rb;
        char* dyw = dybuf + b * DYROW;
        char* gr = gbuf + (ho % 3) * GROW;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int px = wave * 32 + t * 16 + li;
          const int a_off = 16 * (px + g);
          f32x4 acc[4];
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) acc[ni] = (f32x4){0.f, 0.f, 0.f, 0.f};
          uint4 fa[7];
#pragma unroll
          for (int r = 0; r < 7; ++r) fa[r] = *reinterpret_cast<const uint4*>(ring + ((2 * ho + r) % NSLOT) * SLOT + a_off);
          // this pixel's gradient sums (fp32, two 8-channel chunks), zeroed behind the read for the row that takes the slot next
          const bool pvalid = px < WO;
          const int pxc = pvalid ? px : WO - 1;
          float4 gs[2][2];
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            char* gp = gr + pxc * GP + (j * 4 + g) * 32;
            gs[j][0] = *reinterpret_cast<const float4*>(gp);
            gs[j][1] = *reinterpret_cast<const float4*>(gp + 16);
          }
#pragma unroll
          for (int r = 0; r < 7; ++r)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) acc[ni] = sh_mfma16(wf[r][ni], fa[r], acc[ni]);
          if (pvalid) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              char* gp = gr + px * GP + (j * 4 + g) * 32;
              *reinterpret_cast<float4*>(gp) = make_float4(0.f, 0.f, 0.f, 0.f);
              *reinterpret_cast<float4*>(gp + 16) = make_float4(0.f, 0.f, 0.f, 0.f);
            }
          }
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int c = j * 4 + g;
            const f32x4 lo = acc[2 * j], hi = acc[2 * j + 1];
            const float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            const float gq[8] = {gs[j][0].x, gs[j][0].y, gs[j][0].z, gs[j][0].w, gs[j][1].x, gs[j][1].y, gs[j][1].z, gs[j][1].w};
            unsigned dyo[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const unsigned yw = pack_bf16x2(v[2 * i], v[2 * i + 1]);
              const unsigned gr2 = pack_bf16x2(gq[2 * i], gq[2 * i + 1]);            // "rounded as maxpool_bwd would have stored it"
              const float y0 = h16_lo(yw), y1 = h16_hi(yw);
              const float g0 = y0 * sc[j][2 * i] + sh[j][2 * i] > 0.f ? h16_lo(gr2) : 0.f;
              const float g1 = y1 * sc[j][2 * i + 1] + sh[j][2 * i + 1] > 0.f ? h16_hi(gr2) : 0.f;
              const float o0 = cA[j][2 * i] * g0 + (y0 * cP[j][2 * i] + cQ[j][2 * i]);
              const float o1 = cA[j][2 * i + 1] * g1 + (y1 * cP[j][2 * i + 1] + cQ[j][2 * i + 1]);
              dyo[i] = pack_bf16x2(o0, o1);
            }
            if (pvalid) *reinterpret_cast<uint4*>(dyw + px * DYS + c * 16) = make_uint4(dyo[0], dyo[1], dyo[2], dyo[3]);
          }
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // both dy rows complete
      wgrad_rows(ra, has_a, has_b);
      // (the next iteration's scatter touches only the gradient image; its dy writes come behind two more barriers: no barrier here)
    }
  }

  // ---- this block's partial: part[block][ch][r * 32 + kk]; accumulator lane (li, g) of tile (mt, nt) = channel 16 mt + 4 g + e, k-element 16 nt + li
  float* out = p.part + (long long)blockIdx.x * 64 * 224;
#pragma unroll
  for (int f = 0; f < 2; ++f) {
    const int r = 2 * wave + f;
    if (r < 7) {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int e = 0; e < 4; ++e) out[(16 * mt + 4 * g + e) * 224 + r * 32 + 16 * nt + li] = dw[f][mt][nt][e];
    }
  }
}

// ---- the stem's weight gradient alone (one-pass chain: dy was written by the BatchNorm-backward apply), HBM-bound ------------------------------
// dW[64][7 x 32] = sum over pixels of dy^T x with BOTH operands in LDS rings: the padded input rows as in the forward (a pixel's 8 taps x 4
// channels of filter row r start at byte 16 px of ring row 2 ho + r: the x operand is read straight from there), the dy rows
// ([112 px][64 ch], 14 KB each) three deep, every byte of dy and of the input crossing HBM -> LDS once by LDS-DMA.  The tile kernel it
// replaces (wgrad_kernel<.., 64, 256, STEM>, 1.6 ms at 2048 x 224^2 for 4.2 GB) staged both operands through registers per 32-pixel
// k-step with the input re-fetched per filter row; here a step is one conv row: 4 k-steps of 32 pixels (pixels 112..127: zero dy rows),
// wave w owns filter rows 2 w, 2 w + 1 (row 7 is the layout's zero row: skipped), 64 MFMAs per wave and row, both operands by
// ds_read_b64_tr_b16.  dy rows keep their 128-B pitch in LDS (the DMA image is lane-linear); the 32-B channel groups are rotated by
// (px >> 1) & 3 on the DMA source side, which keeps the eight pixel rows a transpose read touches on distinct bank slots
// (wgrad3x3_kernel's layout).  256 threads, 78 KB of LDS: two blocks per CU, persistent over images; deterministic per-block partials.
struct StemWgradArgs {
  const bf16_t* xp;   // [n][hp][wp][4]
  const bf16_t* dy;   // [n][112][112][64]
  float* part;        // [grid][64][224]
  int n, hp, wp;
};

__global__ __launch_bounds__(256, 2) void stem_wgrad_ring_kernel(StemWgradArgs p) {
  constexpr int SLOT = 2048, NSLOT = 14, D = 2;   // rows 2 ho .. 2 ho + 10 live (11 consecutive rows); 14 slots keep two blocks per CU
  constexpr int HO = 112, WO = 112;
  constexpr int DYSLOT = 128 * 128;                    // 112 real pixel rows of 128 B + 16 rows that stay zero
  __shared__ __attribute__((aligned(16))) char ring[NSLOT * SLOT + 128];
  __shared__ __attribute__((aligned(16))) char dyr[3 * DYSLOT];
  __shared__ __attribute__((aligned(16))) char sink[1024];
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int li = lane & 15, g = lane >> 4;
  const int row_bytes = p.wp * 8;
  const int nchunk = row_bytes >> 4;
  auto dma16 = [](const void* src, unsigned lds_addr) __attribute__((always_inline)) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_addr), "v"(src));
  };
  const unsigned ring_addr = (unsigned)(size_t)(__attribute__((address_space(3))) char*)ring;
  const unsigned dyr_addr = (unsigned)(size_t)(__attribute__((address_space(3))) char*)dyr;
  const unsigned sink_addr = (unsigned)(size_t)(__attribute__((address_space(3))) char*)sink;
  const char* zsrc = reinterpret_cast<const char*>(g_sb_zero_page);

  f32x4 dw[2][4][2];
#pragma unroll
  for (int f = 0; f < 2; ++f)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) dw[f][mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // zero for the block's life: the ring's overrun pad, pixel rows 112..127 of the three dy slots
  if (tid < 8) *reinterpret_cast<uint4*>(ring + NSLOT * SLOT + tid * 16) = make_uint4(0, 0, 0, 0);
  for (int i = tid; i < 3 * 16 * 8; i += 256) *reinterpret_cast<uint4*>(dyr + (i / 128) * DYSLOT + 112 * 128 + (i % 128) * 16) = make_uint4(0, 0, 0, 0);

  // the lane's fixed part of the dy source address inside a 1-KB piece: pixel (lane >> 3) of the piece, LDS chunk lane & 7 = rotated group
  // (lane & 7) >> 1, half lane & 1; piece k holds pixels 8 k .. 8 k + 7, whose rotation (px >> 1) & 3 = (4 k + (lane >> 4)) & 3
  const int dpix = lane >> 3, dch = lane & 7;

  for (int img = blockIdx.x; img < p.n; img += gridDim.x) {
    const char* xbase = reinterpret_cast<const char*>(p.xp) + (long long)img * p.hp * row_bytes;
    const char* dybase = reinterpret_cast<const char*>(p.dy) + (long long)img * HO * WO * 128;
    auto dma_in = [&](int y, int hf) __attribute__((always_inline)) {
      const int c = 64 * hf + lane;
      const bool ok = y < p.hp && c < nchunk;
      dma16(ok ? xbase + (long long)y * row_bytes + c * 16 : zsrc, ring_addr + (unsigned)(y % NSLOT) * SLOT + hf * 1024);
    };
    // piece k (0..13) of dy row `row` -> slot row % 3
    auto dma_dy = [&](int row, int k) __attribute__((always_inline)) {
      const int px = 8 * k + dpix;
      const int grp = ((dch >> 1) - ((px >> 1) & 3)) & 3;                 // LDS group = (source group + rotation) & 3
      const char* src = dybase + ((long long)row * WO + px) * 128 + (grp * 2 + (dch & 1)) * 16;
      dma16(row < HO ? src : zsrc, dyr_addr + (unsigned)(row % 3) * DYSLOT + k * 1024);
    };
    // every wave issues FIVE DMAs per step: one input half row, four dy pieces (14 real ones over the four waves + two into the sink)
    auto dma_step = [&](int ho) __attribute__((always_inline)) {
      dma_in(2 * (ho + D) + 5 + (wave >> 1), wave & 1);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int k = wave * 4 + i;
        if (k < 14) dma_dy(ho + D, k);
        else dma16(zsrc, sink_addr);
      }
    };

    __syncthreads();  // (the previous image's last reads are done)
    for (int k = wave; k < 2 * (2 * D + 5); k += 4) dma_in(k >> 1, k & 1);
#pragma unroll
    for (int r0 = 0; r0 < D; ++r0)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int k = wave * 4 + i;
        if (k < 14) dma_dy(r0, k);
        else dma16(zsrc, sink_addr);
      }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int ho = 0; ho < HO; ++ho) {
      // only LDS-DMAs in the vector-memory queue, five per wave and step, in order: everything but the previous step's five has landed
      asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      dma_step(ho);
      const char* dyb = dyr + (ho % 3) * DYSLOT;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        // A = dy^T: channel tile mt = 32-B group mt of the pixel row, rotated by (px >> 1) & 3; lane (p, g) supplies pixel 32 ks + 4 g + (p >> 2)
        uint4 af[4];
        {
          const int pr = lane & 15, gg = lane >> 4;
          const int q0 = 32 * ks + 4 * gg + (pr >> 2), q1 = q0 + 16;
          typedef sb_s16x4 __attribute__((address_space(3))) * lds_ptr;
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) {
            const char* a0 = dyb + q0 * 128 + (((mt + (q0 >> 1)) & 3) * 32) + (pr & 3) * 8;
            const char* a1 = dyb + q1 * 128 + (((mt + (q1 >> 1)) & 3) * 32) + (pr & 3) * 8;
            const sb_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(a0));
            const sb_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(a1));
            af[mt].x = (unsigned)(unsigned short)lo[0] | ((unsigned)(unsigned short)lo[1] << 16);
            af[mt].y = (unsigned)(unsigned short)lo[2] | ((unsigned)(unsigned short)lo[3] << 16);
            af[mt].z = (unsigned)(unsigned short)hi[0] | ((unsigned)(unsigned short)hi[1] << 16);
            af[mt].w = (unsigned)(unsigned short)hi[2] | ((unsigned)(unsigned short)hi[3] << 16);
          }
        }
#pragma unroll
        for (int f = 0; f < 2; ++f) {
          const int r = 2 * wave + f;
          if (r < 7) {
            const char* xr = ring + ((2 * ho + r) % NSLOT) * SLOT + ks * 32 * 16;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
              const uint4 bf = sb_frag_tr(xr, 16, nt * 32, lane);
#pragma unroll
              for (int mt = 0; mt < 4; ++mt) dw[f][mt][nt] = sh_mfma16(af[mt], bf, dw[f][mt][nt]);
            }
          }
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  }
  float* out = p.part + (long long)blockIdx.x * 64 * 224;
#pragma unroll
  for (int f = 0; f < 2; ++f) {
    const int r = 2 * wave + f;
    if (r < 7) {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
          for (int e = 0; e < 4; ++e) out[(16 * mt + 4 * g + e) * 224 + r * 32 + 16 * nt + li] = dw[f][mt][nt][e];
    }
  }
}

// dw_oihw[ch][c][r][tap] = sum over blocks of part[b][ch][r * 32 + tap * 4 + c]  (fixed order: deterministic)
__global__ __launch_bounds__(256) void stem_bwd_reduce_kernel(const float* __restrict__ part, int nblk, float* __restrict__ dw) {
  const int i = blockIdx.x * 256 + threadIdx.x;  // over 64 * 3 * 7 * 7
  if (i >= 64 * 147) return;
  const int ch = i / 147, rem = i % 147, c = rem / 49, r = (rem % 49) / 7, tap = rem % 7;
  const float* src = part + ch * 224 + r * 32 + tap * 4 + c;
  double s = 0.0;
  for (int b = 0; b < nblk; ++b) s += (double)src[(long long)b * 64 * 224];
  dw[i] = (float)s;
}

int stem_bwd_blocks(int n) { return n < 256 ? n : 256; }
int stem_wgrad_ring_blocks(int n) { return n < 512 ? n : 512; }   // two blocks per CU

int launch_stem_wgrad_ring(const void* xp, const void* dy, float* dw_oihw, float* workspace, int n, int hp, int wp, hipStream_t s) {
  StemWgradArgs a;
  a.xp = (const bf16_t*)xp; a.dy = (const bf16_t*)dy; a.part = workspace; a.n = n; a.hp = hp; a.wp = wp;
  const int grid = stem_wgrad_ring_blocks(n);
  stem_wgrad_ring_kernel<<<grid, 256, 0, s>>>(a);
  stem_bwd_reduce_kernel<<<(64 * 147 + 255) / 256, 256, 0, s>>>(workspace, grid, dw_oihw);
  return 0;
}

int launch_stem_bwd_fused(const void* xp, const void* w, const void* dz, const unsigned char* idx, const float* scale, const float* shift,
                          const float* mean, const float* invstd, const float* gamma, const float* dgamma, const float* dbeta, float* dw_oihw,
                          float* workspace, int n, int hp, int wp, hipStream_t s) {
  StemBwdArgs a;
  a.xp = (const bf16_t*)xp; a.w = (const bf16_t*)w; a.dz = (const bf16_t*)dz; a.idx = idx;
  a.scale = scale; a.shift = shift; a.mean = mean; a.invstd = invstd; a.gamma = gamma; a.dgamma = dgamma; a.dbeta = dbeta;
  a.inv_m = (float)(1.0 / ((double)n * 112 * 112));
  a.part = workspace; a.n = n; a.hp = hp; a.wp = wp;
  const int grid = stem_bwd_blocks(n);
  stem_bwd_fused_kernel<<<grid, 256, 0, s>>>(a);
  stem_bwd_reduce_kernel<<<(64 * 147 + 255) / 256, 256, 0, s>>>(workspace, grid, dw_oihw);
  return 0;
}

}  // namespace sh

using namespace sh;

extern "C" {

size_t simhand_stem_bwd_fused_workspace_bytes(int n) { return n < 1 ? 0 : (size_t)stem_bwd_blocks(n) * 64 * 224 * sizeof(float); }

int simhand_stem_bwd_fused(const void* xp, const void* wp_, const void* dz, const uint8_t* idx, const float* scale, const float* shift,
                           const float* mean, const float* invstd, const float* gamma, const float* dgamma, const float* dbeta,
                           float* dw_oihw, void* workspace, size_t workspace_bytes, int n, int h, int w, int dtype, sh_stream_t stream) {
  SH_REQUIRE(xp && wp_ && dz && idx && scale && shift && mean && invstd && dgamma && dbeta && dw_oihw && workspace, "stem_bwd_fused: NULL pointer");
  SH_REQUIRE(simhand_stem_two_pass_ok(n, h, w, dtype), "stem_bwd_fused: 16-bit storage at 224 x 224 only (n=%d h=%d w=%d dtype=%d)", n, h, w, dtype);
  SH_REQUIRE(workspace_bytes >= simhand_stem_bwd_fused_workspace_bytes(n), "stem_bwd_fused: workspace too small");
  int hp, wp, ho, wo;
  if (simhand_stem_geometry(h, w, &hp, &wp, &ho, &wo)) return 1;
  const double mo = (double)n * ho * wo;
  // algorithmic work: the weight gradient's FLOPs (the recomputed forward is overhead, not credit); bytes: input + pooled gradient + winner index
  ProfScope ps(SH_PROF_CONV_WGRAD, (hipStream_t)stream, 2.0 * mo * 64 * 147, 2.0 * ((double)n * hp * wp * 4 + mo / 4 * 64) + mo / 4 * 64);
  route_hit(SH_ROUTE_STEM_BWD_FUSED);
  launch_stem_bwd_fused(xp, wp_, dz, idx, scale, shift, mean, invstd, gamma, dgamma, dbeta, dw_oihw, (float*)workspace, n, hp, wp, (hipStream_t)stream);
  return check_launch("stem_bwd_fused");
}

}  // extern "C"
