// Shared device/host helpers for libsimhand_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>
#include <atomic>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/simhand_hip.h"
#include "../../include/simhand_hip_test.h"  // route counters, event profiler, test hooks: implemented here, not part of the product surface

namespace sh {

// ---- error plumbing -------------------------------------------------------
void set_error(const char* fmt, ...);
int check_launch(const char* what);

#define SH_REQUIRE(cond, ...)          \
  do {                                 \
    if (!(cond)) {                     \
      sh::set_error(__VA_ARGS__);      \
      return 1;                        \
    }                                  \
  } while (0)

// ---- route counters (core.hip): one relaxed atomic increment per launch ----
void route_hit(int route);
// test / tuning hooks are relaxed atomics (they only choose between kernels that compute the same result); every file that
// owns some puts them back to their defaults here (simhand_test_hooks_reset)
typedef std::atomic<int> hook_t;
int sw(int which);  // current value of a simhand_test_switch (core.hip): the hook if set, else the built-in default
void hooks_reset_igemm();
void hooks_reset_c64();
void hooks_reset_r128();
void hooks_reset_1x1();
void hooks_reset_wgrad();
void hooks_reset_bn();

// ---- profiler hooks (prof.hip) ---------------------------------------------
struct ProfScope {
  int cls;
  hipStream_t stream;
  bool on;
  int slot;
  ProfScope(int cls, hipStream_t s, double flops, double bytes);
  ~ProfScope();
};

// ---- types -----------------------------------------------------------------
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) short bf16x8;  // 8 bf16 in 4 VGPRs
typedef unsigned short bf16_t;                              // raw bf16 bits

// The library is built twice from the same sources (csrc/Makefile): libsimhand_hip.so stores 16-bit tensors as bf16 (8-bit
// significand, fp32's exponent range: BASELINE's benchmark dtype), libsimhand_hip_f16.so (-DSH_H16_FP16) as IEEE fp16 (11-bit
// significand, 5-bit exponent: the storage type of the reference's precision=16 / native AMP, src/experiments/main.py:158-159).  Everything
// that depends on the format goes through the helpers below -- unpack (h16_lo / h16_hi / bf16_to_f32), round-to-nearest-even pack
// (pack_bf16x2 / f32_to_bf16) and the 16x16x32 MFMA (sh_mfma16); the names keep "bf16" for "the build's 16-bit storage type".  Layouts,
// tiles, swizzles, LDS-DMA and the transpose reads move raw 16-bit words and do not care.
typedef float hw_f32x2_t __attribute__((ext_vector_type(2)));
#ifdef SH_H16_FP16
typedef _Float16 hw_h16x2_t __attribute__((ext_vector_type(2)));
typedef __attribute__((ext_vector_type(8))) _Float16 hw_h16x8_t;
__device__ __forceinline__ float h16_lo(unsigned w) { return (float)__builtin_bit_cast(hw_h16x2_t, w)[0]; }  // v_cvt_f32_f16
__device__ __forceinline__ float h16_hi(unsigned w) { return (float)__builtin_bit_cast(hw_h16x2_t, w)[1]; }  // v_cvt_f32_f16 (SDWA word 1)
__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return h16_lo((unsigned)v); }
// round-to-nearest-even, overflow -> inf (what the loss scaler's overflow check looks for): gfx950's v_cvt_pk_f16_f32
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
  const hw_f32x2_t v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, hw_h16x2_t));
}
__device__ __forceinline__ f32x4 sh_mfma16(const uint4& a, const uint4& b, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(hw_h16x8_t, a), __builtin_bit_cast(hw_h16x8_t, b), c, 0, 0, 0);
}
#define SH_H16_FORMAT 1
#else
typedef __bf16 hw_bf16x2_t __attribute__((ext_vector_type(2)));
typedef __attribute__((ext_vector_type(8))) __bf16 hw_h16x8_t;
__device__ __forceinline__ float h16_lo(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float h16_hi(unsigned w) { return __uint_as_float(w & 0xffff0000u); }
__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((unsigned)v) << 16); }
// round-to-nearest-even, NaN stays NaN (same rounding as torch's .to(bfloat16)): gfx950's v_cvt_pk_bf16_f32
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
  const hw_f32x2_t v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, hw_bf16x2_t));
}
__device__ __forceinline__ f32x4 sh_mfma16(const uint4& a, const uint4& b, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(hw_h16x8_t, a), __builtin_bit_cast(hw_h16x8_t, b), c, 0, 0, 0);
}
#define SH_H16_FORMAT 0
#endif
__device__ __forceinline__ bf16_t f32_to_bf16(float f) { return (bf16_t)(pack_bf16x2(f, 0.f) & 0xffffu); }

template <typename T> struct Elem;
template <> struct Elem<float> {
  static constexpr int kDtype = SH_F32;
  static constexpr int kVec = 4;  // elements per 16 bytes
  __device__ static __forceinline__ float load(const float* p) { return *p; }
  __device__ static __forceinline__ void store(float* p, float v) { *p = v; }
};
template <> struct Elem<bf16_t> {
  static constexpr int kDtype = SH_BF16;
  static constexpr int kVec = 8;
  __device__ static __forceinline__ float load(const bf16_t* p) { return bf16_to_f32(*p); }
  __device__ static __forceinline__ void store(bf16_t* p, float v) { *p = f32_to_bf16(v); }
};

// 16-byte vector of T <-> fp32 lanes.  NT = non-temporal (streaming) access: the tensors of the BatchNorm / pooling
// passes are far larger than L2 + Infinity Cache and are touched once per pass.
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
template <bool NT> __device__ __forceinline__ uint4 ld16(const void* p) {
  if (NT) {
    const u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
  }
  return *reinterpret_cast<const uint4*>(p);
}
template <bool NT> __device__ __forceinline__ void st16(void* p, const uint4& v) {
  if (NT) {
    u32x4_t o = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(o, reinterpret_cast<u32x4_t*>(p));
  } else {
    *reinterpret_cast<uint4*>(p) = v;
  }
}
template <typename T> struct Vec16;
template <> struct Vec16<float> {
  static constexpr int N = 4;
  template <bool NT = false> __device__ static __forceinline__ void load(const float* p, float (&o)[4]) {
    const uint4 v = ld16<NT>(p);
    o[0] = __uint_as_float(v.x); o[1] = __uint_as_float(v.y); o[2] = __uint_as_float(v.z); o[3] = __uint_as_float(v.w);
  }
  template <bool NT = false> __device__ static __forceinline__ void store(float* p, const float (&o)[4]) {
    st16<NT>(p, make_uint4(__float_as_uint(o[0]), __float_as_uint(o[1]), __float_as_uint(o[2]), __float_as_uint(o[3])));
  }
};
template <> struct Vec16<bf16_t> {
  static constexpr int N = 8;
  template <bool NT = false> __device__ static __forceinline__ void load(const bf16_t* p, float (&o)[8]) {
    const uint4 v = ld16<NT>(p);
    unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      o[2 * i] = h16_lo(w[i]);
      o[2 * i + 1] = h16_hi(w[i]);
    }
  }
  template <bool NT = false> __device__ static __forceinline__ void store(bf16_t* p, const float (&o)[8]) {
    unsigned w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = pack_bf16x2(o[2 * i], o[2 * i + 1]);
    st16<NT>(p, make_uint4(w[0], w[1], w[2], w[3]));
  }
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o));
  return v;
}

// 4 floats -> 4 e4m3 bytes (round-to-nearest-even; the clamp saturates: the hardware convert alone would give NaN past 448 in the
// OCP "fn" encoding)
__device__ __forceinline__ unsigned fp8_pack4(float a, float b, float c, float d) {
  auto cl = [](float v) { return fminf(fmaxf(v, -448.f), 448.f); };
  unsigned r = 0;
  r = __builtin_amdgcn_cvt_pk_fp8_f32(cl(a), cl(b), r, false);
  r = __builtin_amdgcn_cvt_pk_fp8_f32(cl(c), cl(d), r, true);
  return r;
}

inline int ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// 32-bit division by an invariant divisor (host-precomputed magic number), valid for n < 2^31
struct FastDiv {
  unsigned d, mul, shr;
};
inline FastDiv make_fastdiv(unsigned d) {
  FastDiv f;
  f.d = d ? d : 1;
  if (f.d == 1) {
    f.mul = 0;
    f.shr = 0;
    return f;
  }
  unsigned lg = 31 - __builtin_clz(f.d);
  if (f.d & (f.d - 1)) lg += 1;
  const unsigned pw = 31 + lg;
  f.mul = (unsigned)(((1ull << pw) + f.d - 1) / f.d);
  f.shr = pw - 32;
  return f;
}
__device__ __forceinline__ unsigned fdiv(unsigned n, const FastDiv& f) { return f.d == 1 ? n : (__umulhi(n, f.mul) >> f.shr); }

}  // namespace sh
