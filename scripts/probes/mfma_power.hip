// Probe: sustained MFMA rate on RANDOM vs ZERO register operands, bf16 16x16x32 vs 32x32x16 vs the scaled e4m3 16x16x128 (the chip clocks to its power budget: which
// shape buys more FLOP per joule?).  One wave per SIMD x 2 waves, operands in registers, no memory traffic.
// build: hipcc --offload-arch=gfx950 -O3 -o scripts/probes/mfma_power scripts/probes/mfma_power.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) int i32x8;

template <int SHAPE>
__global__ __launch_bounds__(512, 1) void k(const uint4* in, float* out, int iters) {
  const int tid = threadIdx.x;
  uint4 a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = in[(tid * 8 + i) & 4095]; b[i] = in[(tid * 8 + 4 + i) & 4095]; }
  float s = 0.f;
  if (SHAPE == 16) {
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = (f32x4){0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i)
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[i & 3]), __builtin_bit_cast(bf16x8, b[(i >> 2) & 3]), acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  } else if (SHAPE == 128) {  // v_mfma_scale_f32_16x16x128_f8f6f4 on e4m3 bytes, unit scales: the fp8 configuration's instruction (65 536 FLOP each)
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = (f32x4){0, 0, 0, 0};
    i32x8 a8[2], b8[2];
    for (int i = 0; i < 4; ++i) {  // no e4m3 NaN bytes (0x7f / 0xff): clear the exponent's low bit, keep sign and the rest random
      a[i].x &= 0xF7F7F7F7u; a[i].y &= 0xF7F7F7F7u; a[i].z &= 0xF7F7F7F7u; a[i].w &= 0xF7F7F7F7u;
      b[i].x &= 0xF7F7F7F7u; b[i].y &= 0xF7F7F7F7u; b[i].z &= 0xF7F7F7F7u; b[i].w &= 0xF7F7F7F7u;
    }
    for (int i = 0; i < 2; ++i) {
      a8[i] = (i32x8){(int)a[2 * i].x, (int)a[2 * i].y, (int)a[2 * i].z, (int)a[2 * i].w, (int)a[2 * i + 1].x, (int)a[2 * i + 1].y, (int)a[2 * i + 1].z, (int)a[2 * i + 1].w};
      b8[i] = (i32x8){(int)b[2 * i].x, (int)b[2 * i].y, (int)b[2 * i].z, (int)b[2 * i].w, (int)b[2 * i + 1].x, (int)b[2 * i + 1].y, (int)b[2 * i + 1].z, (int)b[2 * i + 1].w};
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i)
        acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8[i & 1], b8[(i >> 1) & 1], acc[i], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    }
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  } else {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[(i + r) & 3]), __builtin_bit_cast(bf16x8, b[(i >> 1) + r]), acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
  }
  out[blockIdx.x * 512 + tid] = s;
}

int main() {
  const int nblk = 256 * 8, iters = 4000;
  uint4* in; float* out;
  hipMalloc(&in, 4096 * 16); hipMalloc(&out, nblk * 512 * 4);
  std::vector<unsigned> h(4096 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 2; ++mode) {
    for (auto& v : h) {  // bf16 pairs: random values ~ N(0,1)-ish magnitudes (random sign / exponent near 0 / mantissa), or zeros
      unsigned lo = 0x3f00u + (rand() & 0xff) + ((rand() & 1) << 15) , hi = 0x3f00u + (rand() & 0xff) + ((rand() & 1) << 15);
      v = mode == 0 ? (lo | (hi << 16)) : 0u;
    }
    hipMemcpy(in, h.data(), 4096 * 16, hipMemcpyHostToDevice);
    for (int shape : {16, 32, 128, 16, 32, 128}) {
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        for (int l = 0; l < 20; ++l) {
          if (shape == 16) k<16><<<nblk, 512>>>(in, out, iters); else if (shape == 32) k<32><<<nblk, 512>>>(in, out, iters); else k<128><<<nblk, 512>>>(in, out, iters);
        }
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        // flops per block-iteration: 16x16x32: 16 MFMA x 16384 x 8 waves ; 32x32x16: 8 MFMA x 32768 x 8 waves -- equal
        const double fl = 20.0 * nblk * (double)iters * 8 * 16 * (shape == 128 ? 65536.0 : 16384.0);
        if (rep == 1) printf("%s operands, %s MFMA: %7.1f ms  %6.0f TFLOP/s\n", mode == 0 ? "random" : "zero  ", shape == 16 ? "16x16x32 bf16" : shape == 32 ? "32x32x16 bf16" : "16x16x128 e4m3 (scaled)", ms, fl / (ms * 1e-3) / 1e12);
      }
    }
  }
  return 0;
}
