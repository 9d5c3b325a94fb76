// Error plumbing, device check and the per-kernel-class HIP-event profiler.
#include <stdarg.h>

#include <atomic>
#include <mutex>
#include <vector>

#include "common.h"

namespace sh {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return 1;
  }
  return 0;
}

static std::atomic<long long> g_routes[SH_ROUTE_COUNT];
void route_hit(int route) {
  if (route >= 0 && route < SH_ROUTE_COUNT) g_routes[route].fetch_add(1, std::memory_order_relaxed);
}

// ---- profiler: one event pair per launch, recorded on the launch stream ----
struct ProfRec {
  hipEvent_t a, b;
  int cls;
  double flops, bytes;
};
static std::mutex g_prof_mu;
static bool g_prof_on = false;
static unsigned g_prof_mask = ~0u;  // classes that record events
static std::vector<ProfRec> g_recs;
static std::vector<std::pair<hipEvent_t, hipEvent_t>> g_pool;

ProfScope::ProfScope(int c, hipStream_t s, double flops, double bytes) : cls(c), stream(s), on(false), slot(-1) {
  if (!g_prof_on || !((g_prof_mask >> c) & 1u)) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  ProfRec r;
  if (!g_pool.empty()) {
    r.a = g_pool.back().first;
    r.b = g_pool.back().second;
    g_pool.pop_back();
  } else {
    if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return;
  }
  r.cls = c;
  r.flops = flops;
  r.bytes = bytes;
  if (hipEventRecord(r.a, s) != hipSuccess) {  // profiling is best effort: drop the record, keep the launch
    g_pool.push_back({r.a, r.b});
    return;
  }
  g_recs.push_back(r);
  slot = (int)g_recs.size() - 1;
  on = true;
}

ProfScope::~ProfScope() {
  if (!on) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  (void)hipEventRecord(g_recs[slot].b, stream);  // a failed record surfaces as an error in prof_collect's synchronize
}

static hook_t g_sw[SH_SW_COUNT];
static const int g_sw_default[SH_SW_COUNT] = {131072, 131072, 1, 7, 1, 3, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0};
static struct SwInit { SwInit() { for (auto& h : g_sw) h.store(-1, std::memory_order_relaxed); } } g_sw_init;
int sw(int which) {
  const int v = g_sw[which].load(std::memory_order_relaxed);
  return v < 0 ? g_sw_default[which] : v;
}

}  // namespace sh

extern "C" {

int simhand_test_switch(int which, int value) {
  if (which < 0 || which >= SH_SW_COUNT) {
    sh::set_error("test_switch: which=%d out of range", which);
    return 1;
  }
  sh::g_sw[which].store(value < 0 ? -1 : value, std::memory_order_relaxed);
  return 0;
}

int simhand_abi_version(void) { return SH_ABI_VERSION; }
// 0: this build's 16-bit storage type (enum SH_BF16) is bfloat16; 1: IEEE fp16 (libsimhand_hip_f16.so)
int simhand_half_format(void) { return SH_H16_FORMAT; }

const char* simhand_last_error(void) { return sh::g_err; }

int simhand_device_check(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n == 0) {
    sh::set_error("no HIP device visible");
    return 1;
  }
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) {
    sh::set_error("hipGetDevice failed");
    return 1;
  }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) {
    sh::set_error("hipGetDeviceProperties failed");
    return 1;
  }
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    sh::set_error("device is %s; this library is built for gfx950 only", prop.gcnArchName);
    return 2;
  }
  return 0;
}

int simhand_route_counts(int64_t* out) {
  if (!out) {
    sh::set_error("route_counts: NULL");
    return 1;
  }
  for (int i = 0; i < SH_ROUTE_COUNT; ++i) out[i] = sh::g_routes[i].load(std::memory_order_relaxed);
  return 0;
}

int simhand_route_reset(void) {
  for (int i = 0; i < SH_ROUTE_COUNT; ++i) sh::g_routes[i].store(0, std::memory_order_relaxed);
  return 0;
}

int simhand_test_hooks_reset(void) {
  sh::hooks_reset_igemm();
  sh::hooks_reset_c64();
  sh::hooks_reset_r128();
  sh::hooks_reset_1x1();
  sh::hooks_reset_wgrad();
  sh::hooks_reset_bn();
  for (auto& h : sh::g_sw) h.store(-1, std::memory_order_relaxed);
  return 0;
}

int simhand_prof_enable(int on) {
  std::lock_guard<std::mutex> lk(sh::g_prof_mu);
  sh::g_prof_on = on != 0;
  return 0;
}

int simhand_prof_set_classes(uint32_t mask) {
  std::lock_guard<std::mutex> lk(sh::g_prof_mu);
  sh::g_prof_mask = mask;
  return 0;
}

int simhand_prof_reset(void) {
  std::lock_guard<std::mutex> lk(sh::g_prof_mu);
  for (auto& r : sh::g_recs) {
    (void)hipEventSynchronize(r.b);  // the pair goes back to the pool either way
    sh::g_pool.push_back({r.a, r.b});
  }
  sh::g_recs.clear();
  return 0;
}

// per-launch records in issue order (diagnostic: which launch sits furthest above the time its own algorithmic bytes / FLOPs allow);
// does NOT clear the records (simhand_prof_collect / simhand_prof_reset do)
int simhand_prof_records(int max_records, int* cls, double* ms, double* flops, double* bytes, int* n_out) {
  std::lock_guard<std::mutex> lk(sh::g_prof_mu);
  if (!cls || !ms || !flops || !bytes || !n_out || max_records < 0) {
    sh::set_error("prof_records: bad arguments");
    return 1;
  }
  int n = 0;
  for (auto& r : sh::g_recs) {
    if (n >= max_records) break;
    if (hipEventSynchronize(r.b) != hipSuccess) {
      sh::set_error("hipEventSynchronize failed in prof_records");
      return 1;
    }
    float t = 0;
    if (hipEventElapsedTime(&t, r.a, r.b) != hipSuccess) {
      sh::set_error("hipEventElapsedTime failed in prof_records");
      return 1;
    }
    cls[n] = r.cls; ms[n] = t; flops[n] = r.flops; bytes[n] = r.bytes;
    ++n;
  }
  *n_out = n;
  return 0;
}

int simhand_prof_collect(double* out_ms, double* out_flops, double* out_bytes, int64_t* out_count) {
  std::lock_guard<std::mutex> lk(sh::g_prof_mu);
  for (int i = 0; i < SH_PROF_NCLASS; ++i) {
    out_ms[i] = 0;
    out_flops[i] = 0;
    out_bytes[i] = 0;
    out_count[i] = 0;
  }
  for (auto& r : sh::g_recs) {
    if (hipEventSynchronize(r.b) != hipSuccess) {
      sh::set_error("hipEventSynchronize failed in prof_collect");
      return 1;
    }
    float ms = 0;
    if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) {
      sh::set_error("hipEventElapsedTime failed in prof_collect");
      return 1;
    }
    out_ms[r.cls] += ms;
    out_flops[r.cls] += r.flops;
    out_bytes[r.cls] += r.bytes;
    out_count[r.cls] += 1;
    sh::g_pool.push_back({r.a, r.b});
  }
  sh::g_recs.clear();
  return 0;
}

}  // extern "C"
