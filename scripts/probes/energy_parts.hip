// Probe: where do the joules of the 256 x 256 x 64 tile loop go?  The socket's power limit (1.4 kW) is what bounds the matrix-bound
// kernels of the step (profiles/r05_power_wall.md); a bare 16x16x32 bf16 MFMA loop sustains 2064 TFLOP/s at that limit, the tile kernel
// 1000-1160.  This probe rebuilds the tile loop's SKELETON piece by piece on random operands -- MFMAs only; + the LDS fragment reads at the
// kernel's ratio (24 ds_read_b128 per 64 MFMAs per wave); + the LDS-DMA refill from L2 (8 global_load_lds_dwordx4 per 64 MFMAs per wave);
// + both with the loop's two barriers -- and the same FLOPs arranged as 4 waves x (128 x 128) (32 reads + 16 DMAs per 128 MFMAs), each
// for ~2 s with the socket power and gfx clock sampled from sysfs: TFLOP/s at the cap = FLOP per joule of that arrangement.
// build: hipcc --offload-arch=gfx950 -O3 -o scripts/probes/energy_parts scripts/probes/energy_parts.hip -lpthread ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dirent.h>
#include <string>
#include <thread>
#include <unistd.h>
#include <vector>
#include <limits.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// WM x WN MFMA tiles per wave (8 x 4 = the kernel's 128 x 64; 8 x 8 = 128 x 128), NW waves, LDSR: fragments from LDS, DMA: refill by LDS-DMA,
// BAR: the loop's two barriers.  One block per CU (LDS footprint forces it).
template <int NW, int WM, int WN, bool LDSR, bool DMA, bool BAR>
__global__ __launch_bounds__(NW * 64, 1) void k(const uint4* __restrict__ in, const char* __restrict__ src, float* out, int ksteps, size_t phase_stride) {
  constexpr int FRAG_BYTES = 96 * 1024, DMA_BYTES = 64 * 1024;  // fragment image (read only) + DMA landing zone (write only): 160 KB
  __shared__ __attribute__((aligned(16))) char smem[FRAG_BYTES + DMA_BYTES];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned smem_addr = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  for (int i = tid; i < FRAG_BYTES / 16; i += NW * 64) ((uint4*)smem)[i] = in[i & 4095];
  __syncthreads();
  uint4 a[WM], b[WN];
  for (int i = 0; i < WM; ++i) a[i] = in[(tid * 16 + i) & 4095];
  for (int i = 0; i < WN; ++i) b[i] = in[(tid * 16 + 8 + i) & 4095];
  f32x4 acc[WM][WN];
  for (int i = 0; i < WM; ++i) for (int j = 0; j < WN; ++j) acc[i][j] = (f32x4){0, 0, 0, 0};
  // DMA: each wave refills its share of a 64-KB stage per k-step: 64 KB / NW / 1 KB instructions
  constexpr int NDMA = 64 / NW;
  const char* gsrc = src + (size_t)(blockIdx.x & 255) * DMA_BYTES + (size_t)lane * 16;
  for (int ks = 0; ks < ksteps; ++ks) {
    if (BAR) { asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      if (LDSR) {
        // conflict-free 16 B per lane, the rows another wave's DMA never touches; (ks, kk) move the window so the data toggles
        const char* base = smem + ((wave * 8 + kk * 4 + (ks & 3)) * 1024) % (FRAG_BYTES - 16 * 1024) + lane * 16;
#pragma unroll
        for (int i = 0; i < WM; ++i) a[i] = *(const uint4*)(base + i * 1024);
#pragma unroll
        for (int j = 0; j < WN; ++j) b[j] = *(const uint4*)(base + (8 + j) * 1024);
      }
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[i]), __builtin_bit_cast(bf16x8, b[j]), acc[i][j], 0, 0, 0);
          if (DMA) {
            constexpr int every = WM * WN * 2 / NDMA;  // spread the wave's NDMA instructions over its 2 * WM * WN MFMAs
            const int n = kk * WM * WN + i * WN + j;
            if (n % every == 0) {
              const int d = n / every;
              const unsigned dst = smem_addr + FRAG_BYTES + (unsigned)((wave * NDMA + d) * 1024);
              const char* s = gsrc + (size_t)((wave * NDMA + d) * 1024) + (size_t)(ks & 7) * phase_stride;
              asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(dst), "v"(s));
            }
          }
        }
    }
    if (BAR) __builtin_amdgcn_s_barrier();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float s = 0.f;
  for (int i = 0; i < WM; ++i) for (int j = 0; j < WN; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  out[blockIdx.x * NW * 64 + tid] = s;
}

static std::string hwmon_dir() {
  char bus[64] = {0};
  hipDeviceGetPCIBusId(bus, sizeof bus, 0);
  for (char* p = bus; *p; ++p) *p = tolower(*p);
  std::string pick;
  for (int c = 0; c < 64; ++c) {
    std::string dev = "/sys/class/drm/card" + std::to_string(c) + "/device";
    char real[PATH_MAX];
    if (!realpath(dev.c_str(), real)) continue;
    if (access((dev + "/pp_dpm_sclk").c_str(), R_OK)) continue;
    const char* base = strrchr(real, '/');
    if (pick.empty() || (base && !strcmp(base + 1, bus))) {
      DIR* d = opendir((dev + "/hwmon").c_str());
      if (!d) continue;
      while (dirent* e = readdir(d))
        if (!strncmp(e->d_name, "hwmon", 5)) pick = dev + "/hwmon/" + e->d_name;
      closedir(d);
      if (base && !strcmp(base + 1, bus)) break;
    }
  }
  return pick;
}
static double read_num(const std::string& p) {
  FILE* f = fopen(p.c_str(), "r");
  if (!f) return -1;
  double v = -1;
  if (fscanf(f, "%lf", &v) != 1) v = -1;
  fclose(f);
  return v;
}

int main() {
  const int nblk = 256, ksteps = 2000;  // one block per CU, one wave of blocks per launch: 2000 k-steps = 2000 x 8.4 MFLOP x 256 blocks
  uint4* in; char* src; float* out;
  hipMalloc(&in, 4096 * 16);
  hipMalloc(&src, (size_t)8 * 256 * 64 * 1024);  // 128 MB; stride 0: every block re-reads its own 64 KB (L2 resident); stride 16 MB: 8 phases, every DMA misses L2 (MALL / HBM)
  hipMalloc(&out, nblk * 512 * 4);
  std::vector<unsigned> h(4096 * 4);
  for (auto& v : h) {
    unsigned lo = 0x3f00u + (rand() & 0xff) + ((rand() & 1) << 15), hi = 0x3f00u + (rand() & 0xff) + ((rand() & 1) << 15);
    v = lo | (hi << 16);
  }
  hipMemcpy(in, h.data(), 4096 * 16, hipMemcpyHostToDevice);
  std::vector<unsigned> big((size_t)8 * 256 * 64 * 1024 / 4);
  for (size_t i = 0; i < big.size(); ++i) big[i] = h[(i * 2654435761u >> 7) & 16383];
  hipMemcpy(src, big.data(), big.size() * 4, hipMemcpyHostToDevice);
  const std::string hw = hwmon_dir();
  printf("hwmon: %s\n", hw.c_str());
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  struct V { const char* name; void (*launch)(const uint4*, const char*, float*, int, int, size_t); size_t stride; };
#define L(NW, WM, WN, LDSR, DMA, BAR) [](const uint4* i, const char* s, float* o, int ks, int nb, size_t st) { k<NW, WM, WN, LDSR, DMA, BAR><<<nb, NW * 64>>>(i, s, o, ks, st); }
  const V vs[] = {
      {"8 waves x 128x64: MFMA only                      ", L(8, 8, 4, false, false, false), 0},
      {"8 waves x 128x64: + LDS fragment reads           ", L(8, 8, 4, true, false, false), 0},
      {"8 waves x 128x64: + LDS-DMA refill               ", L(8, 8, 4, false, true, false), 0},
      {"8 waves x 128x64: + reads + DMA                  ", L(8, 8, 4, true, true, false), 0},
      {"8 waves x 128x64: + reads + DMA + 2 barriers     ", L(8, 8, 4, true, true, true), 0},
      {"8 waves x 128x64: + reads + DMA (L2 MISSES) + bar", L(8, 8, 4, true, true, true), (size_t)256 * 64 * 1024},
      {"4 waves x 128x128: MFMA only                     ", L(4, 8, 8, false, false, false), 0},
      {"4 waves x 128x128: + LDS fragment reads          ", L(4, 8, 8, true, false, false), 0},
      {"4 waves x 128x128: + reads + DMA                 ", L(4, 8, 8, true, true, false), 0},
      {"4 waves x 128x128: + reads + DMA + 2 barriers    ", L(4, 8, 8, true, true, true), 0},
  };
  for (int round = 0; round < 2; ++round)
    for (const V& v : vs) {
      std::atomic<bool> stop{false};
      std::vector<double> pw, ck;
      // warm (and let the power manager settle on this arrangement) 0.7 s, then measure ~1.5 s
      for (int phase = 0; phase < 2; ++phase) {
        std::thread sampler;
        if (phase == 1 && !hw.empty())
          sampler = std::thread([&] {
            while (!stop.load()) {
              double p = read_num(hw + "/power1_average");
              if (p < 0) p = read_num(hw + "/power1_input");
              double c = read_num(hw + "/freq1_input");
              if (p > 0) pw.push_back(p / 1e6);
              if (c > 0) ck.push_back(c / 1e6);
              usleep(20000);
            }
          });
        const int launches = phase == 0 ? 60 : 130;
        hipEventRecord(e0);
        for (int l = 0; l < launches; ++l) v.launch(in, src, out, ksteps, nblk, v.stride);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (phase == 1) {
          stop.store(true);
          if (sampler.joinable()) sampler.join();
          const double fl = (double)launches * nblk * ksteps * 2.0 * 256 * 256 * 64;
          double p = 0, c = 0;
          for (double x : pw) p += x;
          for (double x : ck) c += x;
          p = pw.empty() ? 0 : p / pw.size(); c = ck.empty() ? 0 : c / ck.size();
          const double tf = fl / (ms * 1e-3) / 1e12;
          printf("%s %7.1f ms  %6.0f TFLOP/s  %6.0f W  %5.0f MHz  %5.2f pJ/FLOP (socket)  %5.2f pJ/FLOP (above 250 W idle)\n", v.name, ms, tf, p, c,
                 p / tf, (p - 250) / tf);
          fflush(stdout);
        }
      }
    }
  return 0;
}
