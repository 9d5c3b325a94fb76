// Internal interface of conv3x3_ring.hip (128 -> 128 channel 3x3 / stride-1 convolution: activation tile staged ONCE per 256 padded
// pixels in an LDS ring, weights streamed per tap).
#pragma once
#include "common.h"

namespace sh {

struct R128Args {
  const bf16_t* x;      // source activations [N][H][W][128] (x for forward, dy for the data gradient)
  const bf16_t* w;      // [128 dest][9 taps][128 src]: KRSC (forward) / CRSK (data gradient)
  bf16_t* out;          // [N][H][W][128]
  float* partial;       // forward: BN partial sums [tiles][2][128] of the fp32 results (sum, sum of squares) or null;
                        // dgrad: BN-backward sums of the previous unit (sum g, sum g*y) or null
  const bf16_t* fy;     // dgrad + partial: the previous unit's raw conv output (ReLU mask recomputed from it)
  const float* fscale;  // ... y * fscale + fshift > 0
  const float* fshift;
  int relu;             // dgrad + partial: 1 = gate by the recomputed ReLU mask, 0 = no ReLU
  // forward with the PREVIOUS unit's BatchNorm + ReLU applied on the way in (in_scale != null; see C64Args): x is that unit's raw conv output,
  // the staged tile is rewritten in place as relu(x * in_scale + in_shift), the activation leaves as a by-product (a_out)
  const float* in_scale;
  const float* in_shift;
  bf16_t* a_out;
  int N, H, W;
  int dgrad;            // 1: tap offsets are mirrored
  long long q_total;    // N * (H+1) * (W+1): padded grid with shared pad rows / columns
  int tiles;            // 256-position tiles
  FastDiv div_pp, div_wp;
};

bool r128_supported(int dtype, int cin, int cout, int r, int s, int stride, int pad, int w, long long q_total);
int r128_blocks(long long q_total);
int launch_r128(const R128Args& a, hipStream_t s);
// stride-2 data gradient (128 -> 128, 3x3): a.x = dy [N][H][W][128], a.out = dx [N][2H][2W][128], a.w = CRSK; a.H, a.W, q_total, div_*: the dy grid
bool r128_s2dgrad_supported(int dtype, int cin, int cout, int r, int s, int stride, int pad, int h, int w, int ho, int wo, long long q_total);
int launch_r128_s2dgrad(const R128Args& a, hipStream_t s);
void r128_enable(int on);

}  // namespace sh
