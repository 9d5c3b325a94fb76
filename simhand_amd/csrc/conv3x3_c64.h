// Internal interface of conv3x3_c64.hip (64 -> 64 channel 3x3 / stride-1 convolution with register-resident weights).
#pragma once
#include "common.h"

namespace sh {

struct C64Args {
  const bf16_t* x;      // source activations [N][H][W][64] (x for forward, dy for the data gradient)
  const bf16_t* w;      // [64 dest][9 taps][64 src]: KRSC (forward) / CRSK (data gradient)
  bf16_t* out;          // [N][H][W][64]
  float* partial;       // forward: BN partial sums [blocks][2][64] of the fp32 results (sum, sum of squares) or null;
                        // dgrad: BN-backward sums of the previous unit (sum g, sum g*y) or null
  const bf16_t* fy;     // dgrad + partial: the previous unit's raw conv output (ReLU mask recomputed from it)
  const float* fscale;  // ... y * fscale + fshift > 0
  const float* fshift;
  int relu;             // dgrad + partial: 1 = gate by the recomputed ReLU mask, 0 = no ReLU
  // forward with the PREVIOUS unit's BatchNorm + ReLU applied on the way in (in_scale != null): x is that unit's raw conv output, the ring rows
  // are rewritten in place as relu(x * in_scale + in_shift) before any tap reads them, and the activation leaves as a by-product (a_out: the
  // weight gradient's operand) -- the stand-alone bn_apply pass (one read, one write of the tensor) disappears
  const float* in_scale;
  const float* in_shift;
  bf16_t* a_out;
  int N, H, W;
  int dgrad;            // 1: tap offsets are mirrored
  long long q_total;    // N * (H+2) * (W+2)
  int steps_per_block;  // k... 64-pixel steps per block
  FastDiv div_pp, div_wp;
};

bool c64_supported(int dtype, int cin, int cout, int r, int s, int stride, int pad, int w, long long n_h_w_padded);
int c64_blocks(long long q_total);
int launch_c64(const C64Args& a, hipStream_t s);
void c64_enable(int on);

}  // namespace sh
