// Stand-alone probe (not part of libsimhand): what does the LANE -> ADDRESS shape of a 16-B global store / load cost on gfx950?
//
// Background (DESIGN 3, round 3 "linear epilogue stores"): in an MFMA accumulator layout lane (li = lane & 15, g = lane >> 4) owns
// pixel li and the 16-B chunks g and 4 + g of a 64-channel group, so a store instruction's adjacent lanes are a whole pixel row apart
// and nothing coalesces; the same bytes stored with adjacent lanes on adjacent chunks made the store-bound stem kernel 19 % faster.
// This program times the shapes in isolation over a [rows][row_bytes] bf16 tensor (rows of 128 B .. 4 KB, 16 rows per wave-instruction
// pair), stores and loads separately:
//   acc     lane (li, g): row li, bytes 16 g (+ 64)            two instructions per 16 rows x 128 B, 4 NON-adjacent lanes per 64-B segment
//   line8   lane l: row l >> 3 (+ 8), bytes 16 (l & 7)         two instructions, 8 adjacent lanes per 128-B line
//   linear  lane l: bytes 16 l (+ 1024) of the 2-KB block      only meaningful for row_bytes == 128 (the block is contiguous)
// build: hipcc --offload-arch=gfx950 -O3 request_shape.hip -o request_shape      run: ./request_shape [GiB = 3]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                       \
  do {                                                                              \
    hipError_t e_ = (x);                                                            \
    if (e_ != hipSuccess) {                                                         \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));     \
      exit(1);                                                                      \
    }                                                                               \
  } while (0)

enum Shape { ACC = 0, LINE8 = 1, LINEAR = 2 };

// one wave handles groups of 16 rows x 128 B (one 64-channel chunk of 16 pixels); `chunk` selects the 128-B column of the row
template <int SHAPE, bool STORE>
__global__ __launch_bounds__(256) void probe(char* base, long long groups, int row_bytes, int chunks, unsigned* sink) {
  const int lane = threadIdx.x & 63, li = lane & 15, g = lane >> 4;
  const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long long)gridDim.x * 4;
  uint4 v = make_uint4(lane, 2 * lane, 3 * lane, 4 * lane);
  unsigned acc = 0;
  for (long long grp = wave; grp < groups; grp += nwaves) {
    const long long g16 = grp / chunks;       // which 16 rows
    const int chunk = (int)(grp % chunks);   // which 128-B column
    char* blk = base + g16 * 16 * (long long)row_bytes + chunk * 128;
    char *p0, *p1;
    if (SHAPE == ACC) {
      p0 = blk + (long long)li * row_bytes + g * 16;
      p1 = p0 + 64;
    } else if (SHAPE == LINE8) {
      p0 = blk + (long long)(lane >> 3) * row_bytes + (lane & 7) * 16;
      p1 = p0 + 8ll * row_bytes;
    } else {
      p0 = blk + lane * 16;
      p1 = p0 + 1024;
    }
    if (STORE) {
      *reinterpret_cast<uint4*>(p0) = v;
      *reinterpret_cast<uint4*>(p1) = v;
    } else {
      const uint4 a = *reinterpret_cast<const uint4*>(p0), b = *reinterpret_cast<const uint4*>(p1);
      acc += a.x ^ b.y ^ a.z ^ b.w;
    }
  }
  if (!STORE && acc == 0x9e3779b9u) *sink = acc;
}

template <int SHAPE, bool STORE>
static double run(char* buf, long long bytes, int row_bytes, unsigned* sink) {
  const int chunks = row_bytes / 128;
  const long long rows = bytes / row_bytes, groups = rows / 16 * chunks;
  const int blocks = 256 * 8;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  probe<SHAPE, STORE><<<blocks, 256>>>(buf, groups, row_bytes, chunks, sink);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  const int iters = 5;
  for (int i = 0; i < iters; ++i) probe<SHAPE, STORE><<<blocks, 256>>>(buf, groups, row_bytes, chunks, sink);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return (double)groups * 2048.0 * iters / (ms * 1e-3) / 1e12;  // TB/s
}

int main(int argc, char** argv) {
  const double gib = argc > 1 ? atof(argv[1]) : 3.0;
  const long long bytes = (long long)(gib * (1ll << 30)) / 65536 * 65536;
  char* buf;
  unsigned* sink;
  CK(hipMalloc(&buf, bytes));
  CK(hipMalloc(&sink, 4));
  CK(hipMemset(buf, 1, bytes));
  printf("%.2f GiB tensor; TB/s per shape (store | load)\n", gib);
  printf("%10s %22s %22s %22s\n", "row bytes", "acc (4 x 16 B apart)", "line8 (8 adjacent)", "linear (2 KB block)");
  for (int rb : {128, 256, 512, 1024, 2048, 4096}) {
    const double sa = run<ACC, true>(buf, bytes, rb, sink), sl = run<LINE8, true>(buf, bytes, rb, sink);
    const double la = run<ACC, false>(buf, bytes, rb, sink), ll = run<LINE8, false>(buf, bytes, rb, sink);
    if (rb == 128) {
      const double s2 = run<LINEAR, true>(buf, bytes, rb, sink), l2 = run<LINEAR, false>(buf, bytes, rb, sink);
      printf("%10d %10.2f | %-9.2f %10.2f | %-9.2f %10.2f | %-9.2f\n", rb, sa, la, sl, ll, s2, l2);
    } else {
      printf("%10d %10.2f | %-9.2f %10.2f | %-9.2f %22s\n", rb, sa, la, sl, ll, "-");
    }
  }
  CK(hipFree(buf));
  CK(hipFree(sink));
  return 0;
}
