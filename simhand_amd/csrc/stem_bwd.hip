// Fused backward of the ResNet stem (conv1 7x7/2 -> bn1 -> relu -> maxpool 3/2/1) at 224 x 224, 16-bit storage: ONE kernel that
//   1. recomputes the raw conv1 output y of an image, row by row, from the zero-padded NHWC4 input rows held in an LDS ring
//      (the forward's stem_ring_fwd_kernel arithmetic, same MFMA order: the same bits),
//   2. gathers the pooled gradient dz through the winner index (each conv pixel is a tap of <= 4 windows), gates it by the ReLU
//      and forms dy = cA * g + (y * cP + cQ) -- the BatchNorm-backward apply of pool_bn_bwd_blk_kernel (bn.hip), bit for bit --
//      in the MFMA lanes' registers, rounds it to the storage type as that kernel would have stored it,
//   3. accumulates dW[64][7 x 32] += dy^T x from a 20-KB LDS image of the dy row and the SAME input rows (transpose reads).
// Neither y (3.3 GB at 2048 x 224^2) nor dy (3.3 GB) exists in HBM: the round-3 chain read y, wrote dy (pool_bn_bwd_blk_kernel,
// 1.42 ms) and read dy again next to the input (wgrad_kernel<.., STEM>, 1.55 ms).
//
// Replaces (reference): autograd's max_pool2d_backward -> threshold_backward -> native_batch_norm_backward -> the weight gradient of
// conv1 of torchvision's ResNet (src/models/resnet_model.py:13-26); conv1 has no data gradient (its input is the image).
//
// Shape of the kernel.  256 threads = 4 waves, ONE wave per SIMD (the register budget: 112 registers of resident filter + 96
// accumulators + 80 BatchNorm coefficients), persistent over images (block b: images b, b + grid, ...).  Per conv row ho:
//   * barrier; LDS-DMA: two more input rows (as the forward ring), on even rows one more pooled row of (dz, idx);
//   * wave w, m-tiles t = 0, 1: pixels 32 w + 16 t + li (pixels >= 112 are padding: dy = 0), all 64 channels: 2 x 28 MFMAs, then the
//     gather + BatchNorm arithmetic on its 2 x 2 chunks of 8 channels, dy -> dybuf[ho & 1] ([128 px][64 ch], 160-B rows);
//   * the weight gradient of the PREVIOUS row (its dy is complete behind this row's barrier, its input rows are still in the ring):
//     wave w owns filter rows 2 w, 2 w + 1 (row 7 is the zero row of the [64][256] layout: skipped): 64 channels x 32 k-elements
//     = 8 accumulator tiles per filter row, K = 128 pixels = 4 k-steps; both operands by ds_read_b64_tr_b16 (pixel = k: the
//     permutation inside a k-step is the same for both), the x operand straight from the ring (pixel px of filter row r starts at
//     byte 16 px of ring row 2 ho + r).
// One barrier per conv row.  The per-block partial [64][224] goes to a workspace; stem_bwd_reduce_kernel sums the blocks in a fixed
// order and writes the reference's OIHW fp32 weight.grad.
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
  constexpr int SLOT = 2048, NSLOT = 16, D = 2;     // input ring as stem_ring_fwd_kernel<2>
  constexpr int HO = 112, WO = 112, PH = 56, PW = 56;
  constexpr int DZROW = PW * 128, IXROW = 4096;      // one pooled row of dz (7168 B) / idx (3584 B in a 4-KB slot)
  constexpr int DYS = 160, DYBUF = 128 * DYS;        // dy image: 128 pixel rows of 128 B + 32 B (conflict-free transpose reads)
  __shared__ __attribute__((aligned(16))) char ring[NSLOT * SLOT + 128];   // (+ the overrun of the padding pixels 112..127)
  __shared__ __attribute__((aligned(16))) char pdz[4 * DZROW];
  __shared__ __attribute__((aligned(16))) char pix[4 * IXROW];
  __shared__ __attribute__((aligned(16))) char dybuf[2 * DYBUF];
  __shared__ __attribute__((aligned(16))) char sink[1024];                 // target of the count-balancing dummy DMA
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int li = lane & 15, g = lane >> 4;
  const int row_bytes = p.wp * 8;            // 1856 at 224^2
  const int nchunk = row_bytes >> 4;

  auto dma16 = [](const void* src, unsigned lds_addr) __attribute__((always_inline)) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_addr), "v"(src));
  };
  const unsigned ring_addr = (unsigned)(size_t)(__attribute__((address_space(3))) char*)ring;
  const unsigned pdz_addr = (unsigned)(size_t)(__attribute__((address_space(3))) char*)pdz;
  const unsigned pix_addr = (unsigned)(size_t)(__attribute__((address_space(3))) char*)pix;
  const unsigned sink_addr = (unsigned)(size_t)(__attribute__((address_space(3))) char*)sink;
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
  // on the loop's entry path would make it drain the whole vector-memory queue -- the LDS-DMAs in flight -- in every iteration
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
  // ---- weight-gradient accumulators: filter rows 2 wave + f (f = 0, 1), channel tiles mt, k-element tiles nt ----------------------------------
  f32x4 dw[2][4][2];
#pragma unroll
  for (int f = 0; f < 2; ++f)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) dw[f][mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // the ring's overrun pad (read as "pixels 112..127" of the last slot, always against zero dy rows: must be finite) and
  // the 16 padding pixel rows 112..127 of both dy images stay zero for the kernel's life
  if (tid < 8) *reinterpret_cast<uint4*>(ring + NSLOT * SLOT + tid * 16) = make_uint4(0, 0, 0, 0);
  for (int i = tid; i < 2 * 16 * (DYS / 16); i += 256) {
    const int b = i / (16 * (DYS / 16)), r = i % (16 * (DYS / 16));
    *reinterpret_cast<uint4*>(dybuf + b * DYBUF + 112 * DYS + r * 16) = make_uint4(0, 0, 0, 0);
  }

  // weight gradient of conv row `row` (dy image buffer row & 1): 4 k-steps of 32 pixels, 16 MFMAs each (8 for wave 3: filter row 7 is padding)
  auto wgrad_row = [&](int row) __attribute__((always_inline)) {
    const char* dyb = dybuf + (row & 1) * DYBUF;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      uint4 af[4];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) af[mt] = sb_frag_tr(dyb + ks * 32 * DYS, DYS, mt * 32, lane);
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        const int r = 2 * wave + f;
        if (r < 7) {
          const char* xr = ring + ((2 * row + r) & (NSLOT - 1)) * SLOT + ks * 32 * 16;
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) {
            const uint4 bf = sb_frag_tr(xr, 16, nt * 32, lane);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) dw[f][mt][nt] = sh_mfma16(af[mt], bf, dw[f][mt][nt]);
          }
        }
      }
    }
  };

  for (int img = blockIdx.x; img < p.n; img += gridDim.x) {
    const char* xbase = reinterpret_cast<const char*>(p.xp) + (long long)img * p.hp * row_bytes;
    const char* dzbase = reinterpret_cast<const char*>(p.dz) + (long long)img * PH * DZROW;
    const char* ixbase = reinterpret_cast<const char*>(p.idx) + (long long)img * PH * (PW * 64);
    // half hf of padded input row y -> ring slot y & 15 (chunks past the row's end / rows past the image: the zero page)
    auto dma_in = [&](int y, int hf) __attribute__((always_inline)) {
      const int c = 64 * hf + lane;
      const bool ok = y < p.hp && c < nchunk;
      dma16(ok ? xbase + (long long)y * row_bytes + c * 16 : zsrc, ring_addr + (unsigned)(y & (NSLOT - 1)) * SLOT + hf * 1024);
    };
    // kilobyte k (0..6) of pooled dz row q / kilobyte k (0..3) of its winner-index row -> slot q & 3
    auto dma_dz = [&](int q, int k) __attribute__((always_inline)) {
      dma16(q < PH ? dzbase + (long long)q * DZROW + k * 1024 + lane * 16 : zsrc, pdz_addr + (unsigned)(q & 3) * DZROW + k * 1024);
    };
    auto dma_ix = [&](int q, int k) __attribute__((always_inline)) {
      const int b = k * 1024 + lane * 16;
      dma16(q < PH && b < PW * 64 ? ixbase + (long long)q * (PW * 64) + b : zsrc, pix_addr + (unsigned)(q & 3) * IXROW + k * 1024);
    };
    // every wave issues exactly FOUR pooled-row DMAs per even step (7 + 4 real ones + one into the sink): the counted waits below
    auto dma_pooled = [&](int q) __attribute__((always_inline)) {
      if (wave < 3) {
        dma_dz(q, 2 * wave);
        dma_dz(q, 2 * wave + 1);
        dma_ix(q, wave);
        dma16(zsrc, sink_addr);
      } else {
        dma_dz(q, 6);
        dma_ix(q, 3);
        dma16(zsrc, sink_addr);
        dma16(zsrc, sink_addr);
      }
    };

    // ---- prologue of an image: input rows 0 .. 2 D + 4, pooled rows 0 and 1; waited for in full ---------------------------------------------
    __syncthreads();  // (the previous image's last weight-gradient reads of the ring / dy images are done)
    for (int k = wave; k < 2 * (2 * D + 5); k += 4) dma_in(k >> 1, k & 1);
    dma_pooled(0);
    dma_pooled(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int ho = 0; ho < HO; ++ho) {
      // vector-memory operations of this kernel are LDS-DMAs only and retire in order: an even step issues 1 + 4 per wave, an odd one 1.
      // The rows of step ho were requested in step ho - 2, the pooled row first used in step 2 q - 1 in step 2 q - 4: everything but the
      // previous step's requests must have landed.  lgkmcnt(0): this wave's dy-image writes of the previous row are in LDS.
      if (ho & 1) asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      dma_in(2 * (ho + D) + 5 + (wave >> 1), wave & 1);
      if (!(ho & 1)) dma_pooled((ho >> 1) + 2);

      // ---- conv1 of row ho for the wave's 2 x 16 pixels ---------------------------------------------------------------------------------------
      const int par = ho & 1, p0 = ho >> 1;
      char* dyw = dybuf + (ho & 1) * DYBUF;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int px = wave * 32 + t * 16 + li;
        const int a_off = 16 * (px + g);
        f32x4 acc[4];
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[ni] = (f32x4){0.f, 0.f, 0.f, 0.f};
        uint4 fa[7];
#pragma unroll
        for (int r = 0; r < 7; ++r) fa[r] = *reinterpret_cast<const uint4*>(ring + ((2 * ho + r) & (NSLOT - 1)) * SLOT + a_off);
#pragma unroll
        for (int r = 0; r < 7; ++r)
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) acc[ni] = sh_mfma16(wf[r][ni], fa[r], acc[ni]);
        // ---- gather + BatchNorm backward for chunks j * 4 + g of pixel px -------------------------------------------------------------------
        const bool pvalid = px < WO;
        const int pxc = pvalid ? px : WO - 1;
        const int ow0 = pxc >> 1, odd = pxc & 1;
        const int ow1 = ow0 + 1 < PW ? ow0 + 1 : PW - 1;
        const bool w1ok = odd && ow0 + 1 < PW;
        const unsigned kw0 = (unsigned)(odd + 1);       // tap column of window ow0; window ow1 sees this pixel as tap column 0
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int c = j * 4 + g;
          const f32x4 lo = acc[2 * j], hi = acc[2 * j + 1];
          const float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          unsigned yw[4];
          yw[0] = pack_bf16x2(v[0], v[1]);
          yw[1] = pack_bf16x2(v[2], v[3]);
          yw[2] = pack_bf16x2(v[4], v[5]);
          yw[3] = pack_bf16x2(v[6], v[7]);
          float gq[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) gq[e] = 0.f;
          // windows (oh, ow): even row: oh = p0 (tap row 1); odd row: oh = p0 (tap row 2) and p0 + 1 (tap row 0); accumulation order as
          // pool_bn_bwd_blk_kernel: window rows ascending, window columns ascending inside
#pragma unroll
          for (int wi = 0; wi < 2; ++wi) {
            if (wi == 1 && !par) continue;                      // uniform per row
            const int oh = p0 + wi;
            const bool hok = oh < PH;
            const unsigned kh = par ? (wi == 0 ? 2u : 0u) : 1u;
            const char* zb = pdz + (oh & 3) * DZROW + c * 16;
            const char* ib = pix + (oh & 3) * IXROW + c * 8;
#pragma unroll
            for (int wj = 0; wj < 2; ++wj) {
              const int ow = wj == 0 ? ow0 : ow1;
              const bool ok = hok && pvalid && (wj == 0 || w1ok);
              const unsigned me = ok ? kh * 3u + (wj == 0 ? kw0 : 0u) : 0xffu;   // 0xff: matches no tap
              const uint4 d = *reinterpret_cast<const uint4*>(zb + ow * 128);
              const uint2 ix = *reinterpret_cast<const uint2*>(ib + ow * 64);
              const unsigned d4[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                const unsigned tap = ((e < 4 ? ix.x : ix.y) >> (8 * (e & 3))) & 0xffu;
                const float dv = (e & 1) ? h16_hi(d4[e >> 1]) : h16_lo(d4[e >> 1]);
                gq[e] += tap == me ? dv : 0.f;
              }
            }
          }
          unsigned dyo[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const unsigned gr = pack_bf16x2(gq[2 * i], gq[2 * i + 1]);            // "rounded as maxpool_bwd would have stored it"
            const float y0 = h16_lo(yw[i]), y1 = h16_hi(yw[i]);
            const float g0 = y0 * sc[j][2 * i] + sh[j][2 * i] > 0.f ? h16_lo(gr) : 0.f;
            const float g1 = y1 * sc[j][2 * i + 1] + sh[j][2 * i + 1] > 0.f ? h16_hi(gr) : 0.f;
            const float o0 = cA[j][2 * i] * g0 + (y0 * cP[j][2 * i] + cQ[j][2 * i]);
            const float o1 = cA[j][2 * i + 1] * g1 + (y1 * cP[j][2 * i + 1] + cQ[j][2 * i + 1]);
            dyo[i] = pvalid ? pack_bf16x2(o0, o1) : 0u;
          }
          if (pvalid) *reinterpret_cast<uint4*>(dyw + px * DYS + c * 16) = make_uint4(dyo[0], dyo[1], dyo[2], dyo[3]);
        }
      }
      // ---- weight gradient of the previous row (its dy image and its input rows are complete and still in place) -------------------------------
      if (ho > 0) wgrad_row(ho - 1);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    wgrad_row(HO - 1);
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
