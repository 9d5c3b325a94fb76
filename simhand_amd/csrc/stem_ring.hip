// Direct 7x7 / stride-2 stem forward (3 -> 64 channels, bf16) with the INPUT ROWS staged in an LDS ring: one block per image.
//
// Replaces (reference): conv1 of torchvision's ResNet (src/models/resnet_model.py:13-58; cuDNN there).
//
// The activation-stationary form (conv_1x1.hip, ST) fetches, for every output pixel, seven 64-B runs of the zero-padded NHWC4 input
// straight into MFMA operand registers: 448 B per pixel = 11.5 GB through L2 at 2048 x 224^2 for 0.87 GB of input -- that traffic, not
// HBM (3.3 GB of output) and not the matrix pipes, is what its 1.39 ms are made of.  Neighbouring output pixels share 3/4 of every run
// and neighbouring output rows 5 of their 7 input rows, so here a block walks the 112 output rows of ONE image and keeps the input rows
// in a 32-slot LDS ring (row y of the padded image in slot y & 31; two new rows per step, fetched D = 2 steps ahead by LDS-DMA): every
// input byte crosses L2 -> LDS once.  (Vector-memory operations retire in order, so waiting for the rows requested D steps ago also
// waits for the output stores older than that request; D = 6 -- seven steps of stores in flight instead of three -- measured 6 % SLOWER:
// the kernel is not held by store latency but by the write rate itself, 3.3 GB in 1.12 ms next to 0.9 GB of reads.)
// In the padded layout filter row r of output pixel (ho, wo) is the run of 8 taps x 4 channels at padded pixel (2 ho + r, 2 wo): the 8 k-elements of MFMA lane (li, g) -- taps 2g, 2g + 1 -- are the 16 bytes at byte 16 (wo + g) of
// ring row 2 ho + r: ONE aligned ds_read_b128, consecutive lanes on consecutive chunks (conflict-free), no transposition, no gather.
//   NW waves, one 16-pixel m-tile each (NW x 16 = the output row: 7 x 16 = 112 at 224^2), all 64 output channels: 28 MFMAs per wave and output
//   row against 7 fragment reads; the 64 x 224 filter lives in registers (28 fragments = 112 VGPRs per lane) for the block's life;
//   BatchNorm partial sums of the fp32 results ride in registers across the image: one [2][64] row per block (= per image).
// Round 6: NW is a template parameter -- 7 (224 x 224: BASELINE's geometry) and 4 (128 x 128: the reference's own `--resize` recipe,
// src/experiments/config/training_config.json:38-41, two blocks per CU); other sizes keep the activation-stationary kernel (a padded row
// of a 256 x 256 input no longer fits a 2-KB ring slot).
//
// (Round 4 also carried a two-pass / recompute form of the stem -- statistics-only and BN + ReLU + MaxPool-epilogue variants of this
// kernel plus a fused backward: built, bit-exact, +1.0 ms in the step; removed in round 5, see docs/lab-notes.md and git history.)
#include "conv_1x1.h"

#include <stdlib.h>

namespace sh {

__device__ uint4 g_sr_zero_page[8];

__device__ __forceinline__ float row16_sum_sr(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));  // row_ror:8
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));  // row_ror:4
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));  // row_ror:2
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));  // row_ror:1
  return v;
}

struct StemRingArgs {
  const bf16_t* xp;   // [n][hp][wp][4] zero-padded input
  const bf16_t* w;    // [64][256]: column r*32 + tap*4 + c (stem_pack_weights)
  bf16_t* y;          // [n][ho][wo][64]
  float* partial;     // [n][2][64] or null
  int hp, wp, ho, wo;
};

// LT: the wave's 16 pixels x 128 B of an output row are 2 KB CONTIGUOUS in memory; with LT the packed chunks go through a wave-private
// 2-KB LDS block (16-B chunks XOR-swizzled by the pixel, conflict-free both ways) and leave as two fully linear 1-KB store instructions
// (lane l: bytes 16 l), the shape of the BatchNorm streaming passes, instead of 16 segments of 64 B per instruction.
template <bool LT, int NW = 7>
__global__ __launch_bounds__(NW * 64, NW <= 4 ? 2 : 1) void stem_ring_fwd_kernel(StemRingArgs p) {
  static_assert(NW >= 4 && NW <= 8, "waves 0-3 are the loader waves; a block has at most 512 threads");
  constexpr int SLOT = 2048, NSLOT = 32, D = 2;  // D: steps between a row's request and its use (2 D + 7 <= NSLOT rows)
  __shared__ __attribute__((aligned(16))) char ring[NSLOT * SLOT];
  __shared__ __attribute__((aligned(16))) char tbuf[LT ? NW * 2048 : 16];
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int li = lane & 15, g = lane >> 4;
  const int img = blockIdx.x;
  const int row_bytes = p.wp * 8;            // 1856 at 224^2
  const int nchunk = row_bytes >> 4;         // 16-B chunks per input row
  const char* xbase = reinterpret_cast<const char*>(p.xp) + (long long)img * p.hp * row_bytes;

  auto dma16 = [](const void* src, unsigned lds_addr) __attribute__((always_inline)) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_addr), "v"(src));
  };
  const unsigned ring_addr = (unsigned)(size_t)(__attribute__((address_space(3))) char*)ring;
  const char* zsrc = reinterpret_cast<const char*>(g_sr_zero_page);
  // half hf (0 / 1) of padded row y -> ring slot y & (NSLOT - 1): lane l = 16-B chunk 64 hf + l (chunks past the row's end: the zero page)
  auto dma_half = [&](int y, int hf) __attribute__((always_inline)) {
    const int c = 64 * hf + lane;
    const bool ok = y < p.hp && c < nchunk;
    dma16(ok ? xbase + (long long)y * row_bytes + c * 16 : zsrc, ring_addr + (unsigned)(y & (NSLOT - 1)) * SLOT + hf * 1024);
  };

  // ---- weights: all 64 channels x 7 filter rows, resident in registers.  Fragment row li of channel tile ni <-> channel
  // (ni >> 1)*32 + (li >> 2)*8 + (ni & 1)*4 + (li & 3): a lane's accumulator registers of tiles 2j, 2j + 1 are 8 CONSECUTIVE channels --------
  uint4 wf[7][4];
#pragma unroll
  for (int r = 0; r < 7; ++r)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      const int ch = (ni >> 1) * 32 + (li >> 2) * 8 + (ni & 1) * 4 + (li & 3);
      wf[r][ni] = *reinterpret_cast<const uint4*>(p.w + ch * 256 + r * 32 + g * 8);
    }

  // ---- prologue: rows 0 .. 2 D + 4 (steps 0 .. D - 1), two half-row instructions each, over the 7 waves; waited for in full --------------
  for (int k = wave; k < 2 * (2 * D + 5); k += NW) dma_half(k >> 1, k & 1);
  // everything the prologue requested (filter fragments, rows) is waited for HERE, with the builtin: a load the compiler still counts as
  // pending on the loop's entry path would make its waitcnt pass drain the whole queue in every iteration (see conv_1x1.hip on vmcnt)
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
  __syncthreads();

  const int px = wave * 16 + li;                 // this lane's output pixel (wo) of every row
  const int a_off = 16 * (px + g);               // byte offset of its 8 k-elements in a ring row
  float s1[2][8], s2[2][8];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int e = 0; e < 8; ++e) s1[j][e] = s2[j][e] = 0.f;
  bf16_t* yrow = p.y + ((long long)img * p.ho * p.wo + px) * 64 + g * 8;
  // LT: write offsets of the lane's two chunks (pixel li, chunk j * 4 + g), read offset of linear position l (pixel l >> 3, chunk l & 7)
  char* tw = tbuf + (LT ? wave * 2048 : 0);
  const int tw0 = li * 128 + (((0 * 4 + g) ^ (li & 7)) * 16), tw1 = li * 128 + (((1 * 4 + g) ^ (li & 7)) * 16);
  const int tr0 = (lane >> 3) * 128 + (((lane & 7) ^ ((lane >> 3) & 7)) * 16);  // second kilobyte: pixels 8..15, same keys -> + 1024
  char* ylin = reinterpret_cast<char*>(p.y + ((long long)img * p.ho * p.wo + wave * 16) * 64) + lane * 16;

  for (int ho = 0; ho < p.ho; ++ho) {
    // loader waves 0-3, in issue order (vector-memory operations retire in order): ... DMA(ho-D) [rows of this step], 2 stores(ho-D), then per
    // later step one DMA + 2 stores: <= 3 D - 1 outstanding means this step's rows have landed (the other waves have only stores in flight)
    if (ho >= D) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(3 * D - 1) : "memory");
    else asm volatile("s_barrier" ::: "memory");
    // rows 2 (ho + D) + 5, + 6 (the last rows of step ho + D) -> slots outside the windows of steps ho .. ho + D - 1 (2 D + 7 <= NSLOT rows)
    if (wave < 4) dma_half(2 * (ho + D) + 5 + (wave >> 1), wave & 1);
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
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const f32x4 lo = acc[2 * j], hi = acc[2 * j + 1];
      const float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      uint4 o;
      o.x = pack_bf16x2(v[0], v[1]);
      o.y = pack_bf16x2(v[2], v[3]);
      o.z = pack_bf16x2(v[4], v[5]);
      o.w = pack_bf16x2(v[6], v[7]);
      if (LT) *reinterpret_cast<uint4*>(tw + (j == 0 ? tw0 : tw1)) = o;
      else *reinterpret_cast<uint4*>(yrow + j * 32) = o;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        s1[j][e] += v[e];
        s2[j][e] += v[e] * v[e];
      }
    }
    if (LT) {  // same wave, in-order LDS queue: the reads see the writes above; the next row's writes follow these reads
      const uint4 r0 = *reinterpret_cast<const uint4*>(tw + tr0), r1 = *reinterpret_cast<const uint4*>(tw + tr0 + 1024);
      st16<true>(ylin, r0);  // non-temporal (round 6): 3.3 GB of whole-line stores that the next launch streams back in; forward class -0.17 ms
      st16<true>(ylin + 1024, r1);
      ylin += (long long)p.wo * 128;
    }
    yrow += (long long)p.wo * 64;
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  if (p.partial != nullptr) {
    float* red = reinterpret_cast<float*>(ring);  // [NW waves][2][64]
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float t1 = row16_sum_sr(s1[j][e]), t2 = row16_sum_sr(s2[j][e]);
        if (li == 0) {
          red[(wave * 2 + 0) * 64 + j * 32 + g * 8 + e] = t1;
          red[(wave * 2 + 1) * 64 + j * 32 + g * 8 + e] = t2;
        }
      }
    __syncthreads();
    if (tid < 128) {
      const int which = tid >> 6, c = tid & 63;
      float t = 0.f;
#pragma unroll
      for (int wv = 0; wv < NW; ++wv) t += red[(wv * 2 + which) * 64 + c];
      p.partial[((long long)img * 2 + which) * 64 + c] = t;
    }
  }
}

static hook_t g_stem_ring{-1};  // -1 = simhand_test_switch(SH_SW_STEM_RING) (default on), 0 / 1 forced
void stem_ring_enable(int on) { g_stem_ring = on < 0 ? -1 : (on ? 1 : 0); }

// one 16-pixel m-tile per wave (wo = 16 NW with NW in {4, 7}: 128^2 and 224^2 inputs), a padded row inside a 2-KB ring slot
static bool stem_ring_wo_ok(int wo) { return wo == 64 || wo == 112; }
bool stem_ring_geometry_ok(int hp, int wp, int ho, int wo) { return stem_ring_wo_ok(wo) && ho == wo && wp * 8 <= 2048 && hp >= 2 * ho + 5; }

bool stem_ring_ok(int n, int hp, int wp, int ho, int wo) {
  const int env = sw(SH_SW_STEM_RING);
  const int h = g_stem_ring;
  return (h >= 0 ? h : env) && stem_ring_wo_ok(wo) && wp * 8 <= 2048 && hp >= 2 * ho + 5 && n >= 1;
}

int launch_stem_ring(const void* xp, const void* w, void* y, float* partial, int n, int hp, int wp, int ho, int wo, hipStream_t s) {
  StemRingArgs a = {};
  a.xp = (const bf16_t*)xp; a.w = (const bf16_t*)w; a.y = (bf16_t*)y; a.partial = partial;
  a.hp = hp; a.wp = wp; a.ho = ho; a.wo = wo;
  const int lt = sw(SH_SW_STEM_RING_LT);
  route_hit(SH_ROUTE_STEM_RING_FWD);
#define SH_SR(NW)                                                         \
  do {                                                                    \
    if (lt) stem_ring_fwd_kernel<true, NW><<<n, NW * 64, 0, s>>>(a);      \
    else stem_ring_fwd_kernel<false, NW><<<n, NW * 64, 0, s>>>(a);        \
  } while (0)
  if (wo == 64) SH_SR(4);
  else SH_SR(7);
#undef SH_SR
  return 0;
}

}  // namespace sh
