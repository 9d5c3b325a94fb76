// Micro-benchmark: what does the memory system give for the access pattern of the activation-stationary 1x1 kernels'
// epilogues?  A block owns R rows of a [M][C] bf16 tensor and visits them chunk by chunk (CW bytes of every row per visit,
// 1 read + 1 write), like gemm1x1_kernel's 64-channel (128-B) chunks.  Sweep CW = 128 .. full row at row pitches 512 B .. 4 KB.
// build: hipcc --offload-arch=gfx950 -O3 scripts/micro/pattern_copy.hip -o scripts/micro/pattern_copy ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int CW>
__global__ __launch_bounds__(256) void pattern_copy(const uint4* __restrict__ src, uint4* __restrict__ dst, long long M, int row_bytes, int R, int spin) {
  const long long r0 = (long long)blockIdx.x * R;
  constexpr int PPR = CW / 16;          // 16-B pieces per row and visit
  const int rows_per_pass = 256 / PPR;  // rows covered by one pass of the block
  const int piece = threadIdx.x % PPR, rsub = threadIdx.x / PPR;
  for (int c = 0; c < row_bytes / CW; ++c) {
    for (int r = rsub; r < R; r += rows_per_pass) {
      const long long row = r0 + r;
      if (row < M) {
        const long long off = (row * row_bytes + (long long)c * CW) / 16 + piece;
        uint4 v = src[off];
        v.x += 1u;
        dst[off] = v;
      }
    }
    for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(16);  // stand-in for the MFMA phase between visits
    __syncthreads();
  }
}

int main() {
  const long long bytes = 3ll << 30;  // 3 GiB per tensor
  uint4 *a, *b;
  hipMalloc(&a, bytes);
  hipMalloc(&b, bytes);
  hipMemset(a, 1, bytes);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int row_bytes : {512, 1024, 2048, 4096}) {
    const long long M = bytes / row_bytes;
    for (int R : {64, 256}) {
      for (int spin : {0, 4}) {
        printf("pitch %4d B, %3d rows/block, spin %d:", row_bytes, R, spin);
        for (int cw : {128, 256, 512, 1024, 2048, 4096}) {
          if (cw > row_bytes) continue;
          const int grid = (int)((M + R - 1) / R);
          float best = 1e9f;
          for (int it = 0; it < 4; ++it) {
            hipEventRecord(e0);
            switch (cw) {
              case 128: pattern_copy<128><<<grid, 256>>>(a, b, M, row_bytes, R, spin); break;
              case 256: pattern_copy<256><<<grid, 256>>>(a, b, M, row_bytes, R, spin); break;
              case 512: pattern_copy<512><<<grid, 256>>>(a, b, M, row_bytes, R, spin); break;
              case 1024: pattern_copy<1024><<<grid, 256>>>(a, b, M, row_bytes, R, spin); break;
              case 2048: pattern_copy<2048><<<grid, 256>>>(a, b, M, row_bytes, R, spin); break;
              default: pattern_copy<4096><<<grid, 256>>>(a, b, M, row_bytes, R, spin); break;
            }
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (it > 0 && ms < best) best = ms;
          }
          printf("  CW %4d: %5.2f TB/s", cw, 2.0 * bytes / best / 1e9);
        }
        printf("\n");
      }
    }
  }
  return 0;
}
