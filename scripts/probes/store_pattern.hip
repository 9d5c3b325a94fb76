// Probe: cycles a CU needs to ISSUE the output stores of a 256 x 256 bf16 tile (128 KB, 16 global_store_dwordx4 per wave, 8 waves), by the
// lane -> address pattern of one store instruction.  s_memtime stamps inside igemm256_kernel (profiles/r05_igemm256_tile_stamps.txt) show the
// epilogue's store ISSUE taking 4.6 k cycles for the first wave of each SIMD and 7.9 k for the second -- 250 cycles per instruction and SIMD --
// while the drain afterwards is 770: the cost is in how the CU's store path digests the instruction, not in L2 / HBM.
//   P0  the kernel's: lane (g, li) writes 16 B at row li, byte g*16 (+ 64 for the second channel group): 16 rows x 64 B per instruction
//   P1  linear: lane l writes 16 B at l*16: two whole 512-B tile rows per instruction (needs an LDS transpose in the kernel)
//   P2  lane pairs trade a chunk (abl53): 8 rows x 128 B, lanes 2a / 2a+1 of a quarter-wave write bytes g*16 and 64 + g*16 of row a
//   P3  8 lanes per row: lane l writes row l >> 3, byte (l & 7)*16: 8 rows x 128 B with whole lines per 8 consecutive lanes
// build: hipcc --offload-arch=gfx950 -O3 -o scripts/probes/store_pattern scripts/probes/store_pattern.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int P>
__global__ __launch_bounds__(512, 1) void k(char* out, unsigned long long* stamps, int tiles_per_block, int total_tiles) {
  __shared__ char pad[150 * 1024];  // one block per CU, as the kernel
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, g = lane >> 4, wm = wave >> 2, wn = wave & 3;
  if (tid == 0) pad[0] = 0;
  uint4 v = make_uint4(tid * 2654435761u, tid ^ 0x5bd1e995u, tid * 40503u, tid + 77u);
  unsigned long long issue = 0, drain = 0;
  for (int t = 0; t < tiles_per_block; ++t) {
    const int tile = (blockIdx.x + t * gridDim.x) % total_tiles;
    char* base = out + (size_t)tile * 256 * 512;  // [256 rows][256 channels] bf16, contiguous (Ng == 256)
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int mi = 0; mi < 8; ++mi)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        size_t off;
        if (P == 0) off = (size_t)(wm * 128 + mi * 16 + li) * 512 + wn * 128 + j * 64 + g * 16;
        else if (P == 1) off = (size_t)(wave * 32 + (mi * 2 + j) * 2) * 512 + lane * 16;
        else if (P == 2) off = (size_t)(wm * 128 + mi * 16 + (li & ~1) + j) * 512 + wn * 128 + (li & 1) * 64 + g * 16;
        else off = (size_t)(wm * 128 + mi * 16 + j * 8 + (lane >> 3)) * 512 + wn * 128 + (lane & 7) * 16;
        v.x += mi;
        *reinterpret_cast<uint4*>(base + off) = v;
      }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t2 = __builtin_amdgcn_s_memtime();
    issue += t1 - t0;
    drain += t2 - t1;
  }
  if (lane == 0) {
    stamps[(blockIdx.x * 8 + wave) * 2 + 0] = issue;
    stamps[(blockIdx.x * 8 + wave) * 2 + 1] = drain;
  }
}

int main() {
  const int nblk = 256, tpb = 14;
  char* out; unsigned long long* st;
  hipMalloc(&out, (size_t)4096 * 256 * 512);
  hipMalloc(&st, nblk * 8 * 2 * 8);
  std::vector<unsigned long long> h(nblk * 8 * 2);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 4; ++rep)
    for (int p = 0; p < 4; ++p) {
      const int total = rep < 2 ? 4096 : 128;  // 512 MB of output (every store reaches HBM) | 16 MB: every block rewrites one of 128 tiles, L2-resident
      hipEventRecord(e0);
      if (p == 0) k<0><<<nblk, 512>>>(out, st, tpb, total);
      else if (p == 1) k<1><<<nblk, 512>>>(out, st, tpb, total);
      else if (p == 2) k<2><<<nblk, 512>>>(out, st, tpb, total);
      else k<3><<<nblk, 512>>>(out, st, tpb, total);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
      double is[8] = {0}, dr[8] = {0};
      for (int b = 0; b < nblk; ++b)
        for (int w = 0; w < 8; ++w) { is[w] += h[(b * 8 + w) * 2] / (double)tpb / nblk; dr[w] += h[(b * 8 + w) * 2 + 1] / (double)tpb / nblk; }
      printf("%s P%d: %.1f us per tile round (%.0f GB/s);  issue cycles / tile, waves 0..7: %.0f %.0f %.0f %.0f | %.0f %.0f %.0f %.0f;  drain: %.0f .. %.0f\n", total == 4096 ? "to HBM" : "L2-resident", p,
             ms * 1e3 / tpb, (double)nblk * tpb * 131072 / (ms * 1e-3) / 1e9, is[0], is[1], is[2], is[3], is[4], is[5], is[6], is[7], dr[0], dr[7]);
    }
  return 0;
}
