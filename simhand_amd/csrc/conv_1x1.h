// Internal interface of conv_1x1.hip (activation-stationary short-K 1x1 GEMM); used by the conv2d entry points.
#pragma once
#include "common.h"

namespace sh {

struct Gemm1x1Args {
  const bf16_t* a;    // [M][K] activations (x, or dy for the data gradient), NHWC rows
  const bf16_t* w;    // [N][K] weights (KRSC forward, CRSK data gradient)
  bf16_t* out;        // [M][N]
  float* bn_partial;  // [ceil(M / rows_per_block)][2][N] or null
  long long M;
  int N;
  int accumulate;     // 0: store, 1: out += result, 2: out = result + res_grad * bit(res_mask)
  const bf16_t* res_grad;
  const unsigned char* res_mask;
  // dgrad only: fused BatchNorm-backward partial sums of the previous unit (see sh_bn_bwd_fuse); fy null = off
  const bf16_t* fy;
  const float* fscale;
  const float* fshift;
  const unsigned char* fmask;
  int fmode;
  float* fpartial;    // [ceil(M / rows_per_block)][2][N]; null = off
  const float* bias;  // dgrad: per output channel, added to the fp32 result (null = none)
  // forward: out = act(acc * ep_scale[c] + ep_shift[c] (+ ep_res)), ReLU bit mask out (see IgemmArgs); ep_scale null = off
  const float* ep_scale;
  const float* ep_shift;
  const bf16_t* ep_res;
  unsigned char* ep_mask;
  int ep_relu;
  // K = 256 forward only: the direct 7x7/2 stem.  a = zero-padded NHWC4 input [n][stem_hp][stem_wp][4]; row m = output pixel
  // (img, ho, wo); its k-slice j (32 elements) is filter row j: 8 taps x 4 channels contiguous at padded (2 ho + j, 2 wo).
  // stem_wp = 0: off
  // data gradient only: the A operand is DERIVED on load instead of read -- dy = xf_a * (a [y*xf_s + xf_h > 0]) - xf_b * y + xf_c per
  // channel (the BatchNorm-backward apply of the unit whose conv this is the data gradient of; a = incoming gradient, xf_y = the
  // unit's raw conv output) -- and written to xf_out for the weight gradient that follows.  xf_y null = off
  const bf16_t* xf_y = nullptr;
  const float* xf_s = nullptr;
  const float* xf_h = nullptr;
  const float* xf_a = nullptr;
  const float* xf_b = nullptr;
  const float* xf_c = nullptr;
  bf16_t* xf_out = nullptr;
  int xf_relu = 0;
  // forward, EP == 2 fast variant with K == 64 and N <= 256 only: CHAINED second 1x1 convolution (the next Bottleneck's conv1, N -> K
  // channels): chain_y = out * chain_w^T computed from the bf16 output chunks while they are still in registers (a chunk's packed
  // output IS the next MFMA's A operand) -- the block output is written but not read back.  chain_w: [K][N] bf16 (KRSC of the next
  // conv1), chain_y: [M][K] bf16 raw conv output, chain_partial: [blocks][2][K] BatchNorm partial sums of it.  chain_w null = off
  const bf16_t* chain_w = nullptr;
  bf16_t* chain_y = nullptr;
  float* chain_partial = nullptr;
  int stem_hp = 0, stem_wp = 0;
  FastDiv div_hw = {1, 0, 0}, div_w = {1, 0, 0};  // ho * wo, wo (stem); h * w, w of the rows' pixel grid (sub)
  // data gradient, masked-store fast variant only (see gemm1x1_sub_ok): out = mask(result + S) where S is `sub` [n][h/2][w/2][N] at the
  // EVEN pixels of the rows' [n][h][w] grid and zero elsewhere -- the data gradient of a stride-2 1x1 shortcut, computed densely at the
  // output resolution, merged here instead of by a scatter-add pass over out.  sub null = off
  const bf16_t* sub = nullptr;
  int sub_h = 0, sub_w = 0;
  int lt = 0;  // linear output stores through a wave-private LDS transpose (set by launch_gemm1x1: env SIMHAND_G1_LT, default on)
};

bool gemm1x1_supported(int k, int n);
bool gemm1x1_chain_ok(int k, int n, long long m);  // chained next conv1 available (the K = 64 / 128 fast forward variants)
void gemm1x1_set_chain(int mask);                   // bit 0: K = 64, bit 1: K = 128, -1: default
int gemm1x1_chain_rows(int k);                      // rows per block of that launch (= rows per BatchNorm partial-sum pair)
int gemm1x1_rows_per_block(int k);
void gemm1x1_set_mf(int k, int mf);
int launch_gemm1x1(const Gemm1x1Args& a, int k, bool dgrad, hipStream_t s);
bool gemm1x1_sub_ok(const Gemm1x1Args& a, int k);  // the launch these arguments select merges a.sub (else: the caller scatter-adds)
int launch_gemm1x1_stem(const Gemm1x1Args& a, hipStream_t s);  // persistent direct-stem forward (or 256 rows per block)
int gemm1x1_stem_stat_blocks(long long m);                      // rows of the BatchNorm partial-sum buffer that launch fills
void gemm1x1_set_stem_persistent(int on);
// stem_ring.hip: the 224 x 224 stem forward with the input rows staged in an LDS ring (one block per image, one partial row per image)
bool stem_ring_ok(int n, int hp, int wp, int ho, int wo);
bool stem_ring_geometry_ok(int hp, int wp, int ho, int wo);  // the geometry alone (no test hook)
int launch_stem_ring(const void* xp, const void* w, void* y, float* partial, int n, int hp, int wp, int ho, int wo, hipStream_t s);
void stem_ring_enable(int on);
// stem_bwd.hip: the stem's weight gradient with both operands in LDS rings (224 x 224, 16-bit storage); workspace = blocks x 64 x 224 floats
int stem_wgrad_ring_blocks(int n);
int launch_stem_wgrad_ring(const void* xp, const void* dy, float* dw_oihw, float* workspace, int n, int hp, int wp, int wo, hipStream_t s);

}  // namespace sh
