// The stem's backward at 224 x 224, 16-bit storage (conv1 7x7/2 of torchvision's ResNet, src/models/resnet_model.py:13-26; conv1 has no data
// gradient -- its input is the image): the weight gradient with both operands in LDS rings (stem_wgrad_ring_kernel) and the deterministic
// reduction of its per-block partials.  Round 4 also carried a FUSED backward here (conv1 recomputed, pooled gradient routed through the
// winner index by an LDS scatter, dy formed in registers, dW accumulated in the same kernel): exact, 3.22 ms against 1.75 + 1.63 in
// isolation, +1.0 ms inside the step; removed in round 5 (docs/lab-notes.md round 4 / 5, git history).
#include "conv_1x1.h"

namespace sh {

__device__ uint4 g_sb_zero_page[8];

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
  const bf16_t* dy;   // [n][WO][WO][64]
  float* part;        // [grid][64][224]
  int n, hp, wp;
};

// Round 6: the output width is a template parameter -- 112 (224 x 224 inputs) and 64 (128 x 128: the reference's `--resize` recipe,
// training_config.json:38-41): KS = ceil(WO / 32) k-steps per conv row, WO / 8 dy pieces of 1 KB per row dealt to the four waves (PPW each,
// the surplus into the sink), 1 + PPW DMAs per wave and step -- the counted wait follows.
template <int WO>
__global__ __launch_bounds__(256, 2) void stem_wgrad_ring_kernel(StemWgradArgs p) {
  constexpr int SLOT = 2048, NSLOT = 14, D = 2;   // rows 2 ho .. 2 ho + 10 live (11 consecutive rows); 14 slots keep two blocks per CU
  constexpr int HO = WO;
  constexpr int KS = (WO + 31) / 32;                   // 32-pixel k-steps per conv row
  constexpr int PIECES = WO / 8, PPW = (PIECES + 3) / 4;
  constexpr int ZR = KS * 32 - WO;                     // pixel rows of a dy slot that stay zero (16 at WO = 112)
  constexpr int DYSLOT = KS * 32 * 128;                // WO real pixel rows of 128 B + ZR rows that stay zero
  static_assert(WO % 8 == 0 && WO * 16 + 8 * 16 <= SLOT, "a padded input row must fit a ring slot");
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
  // zero for the block's life: the ring's overrun pad, pixel rows WO .. 32 KS - 1 of the three dy slots
  if (tid < 8) *reinterpret_cast<uint4*>(ring + NSLOT * SLOT + tid * 16) = make_uint4(0, 0, 0, 0);
  if (ZR > 0)
    for (int i = tid; i < 3 * ZR * 8; i += 256) *reinterpret_cast<uint4*>(dyr + (i / (ZR * 8)) * DYSLOT + WO * 128 + (i % (ZR * 8)) * 16) = make_uint4(0, 0, 0, 0);

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
    // piece k (0 .. PIECES - 1) of dy row `row` -> slot row % 3
    auto dma_dy = [&](int row, int k) __attribute__((always_inline)) {
      const int px = 8 * k + dpix;
      const int grp = ((dch >> 1) - ((px >> 1) & 3)) & 3;                 // LDS group = (source group + rotation) & 3
      const char* src = dybase + ((long long)row * WO + px) * 128 + (grp * 2 + (dch & 1)) * 16;
      dma16(row < HO ? src : zsrc, dyr_addr + (unsigned)(row % 3) * DYSLOT + k * 1024);
    };
    // every wave issues 1 + PPW DMAs per step: one input half row, PPW dy pieces (WO = 112: FIVE -- 14 real pieces over the four waves + two into the sink)
    auto dma_step = [&](int ho) __attribute__((always_inline)) {
      dma_in(2 * (ho + D) + 5 + (wave >> 1), wave & 1);
#pragma unroll
      for (int i = 0; i < PPW; ++i) {
        const int k = wave * PPW + i;
        if (k < PIECES) dma_dy(ho + D, k);
        else dma16(zsrc, sink_addr);
      }
    };

    __syncthreads();  // (the previous image's last reads are done)
    for (int k = wave; k < 2 * (2 * D + 5); k += 4) dma_in(k >> 1, k & 1);
#pragma unroll
    for (int r0 = 0; r0 < D; ++r0)
#pragma unroll
      for (int i = 0; i < PPW; ++i) {
        const int k = wave * PPW + i;
        if (k < PIECES) dma_dy(r0, k);
        else dma16(zsrc, sink_addr);
      }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int ho = 0; ho < HO; ++ho) {
      // only LDS-DMAs in the vector-memory queue, 1 + PPW per wave and step, in order: everything but the previous step's has landed
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(1 + PPW) : "memory");
      dma_step(ho);
      const char* dyb = dyr + (ho % 3) * DYSLOT;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
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

int stem_wgrad_ring_blocks(int n) { return n < 512 ? n : 512; }   // two blocks per CU

int launch_stem_wgrad_ring(const void* xp, const void* dy, float* dw_oihw, float* workspace, int n, int hp, int wp, int wo, hipStream_t s) {
  StemWgradArgs a;
  a.xp = (const bf16_t*)xp; a.dy = (const bf16_t*)dy; a.part = workspace; a.n = n; a.hp = hp; a.wp = wp;
  const int grid = stem_wgrad_ring_blocks(n);
  route_hit(SH_ROUTE_STEM_RING_WGRAD);
  if (wo == 64) stem_wgrad_ring_kernel<64><<<grid, 256, 0, s>>>(a);
  else stem_wgrad_ring_kernel<112><<<grid, 256, 0, s>>>(a);
  stem_bwd_reduce_kernel<<<(64 * 147 + 255) / 256, 256, 0, s>>>(workspace, grid, dw_oihw);
  return 0;
}

}  // namespace sh
