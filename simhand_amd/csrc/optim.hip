// LARSWrapper(Adam) parameter update, one fused elementwise launch per tensor.
//
// Replaces (reference): pl_bolts 0.2.2 `LARSWrapper.step` around `torch.optim.Adam.step`
// as configured in src/models/base_model.py:59-106 (Adam lr = 1e-4*sqrt(1024*k), LARS
// eta 0.02, clip True, eps 1e-8).  pl_bolts / torch.optim are not vendored in the
// reference: semantics restated from their published sources -- PARITY UNPINNED.
//   LARS (per tensor with grad): if ||p|| != 0 and ||g|| != 0:
//        lr' = eta*||p|| / (||g|| + wd*||p|| + eps);  if clip: lr' = min(lr'/lr, 1)
//        g <- (g + wd*p) * lr'        (the group's weight_decay is zeroed for the inner Adam step)
//   Adam: m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ;
//        p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
// Norms come from a deterministic two-stage sum of squares (partials -> fold in-kernel).
#include "common.h"

namespace sh {

__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ x, int64_t count, float* __restrict__ partial) {
  __shared__ float red[4];
  float s = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256) s += x[i] * x[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void lars_adam_kernel(float* __restrict__ param, const float* __restrict__ grad,
                                                        float* __restrict__ exp_avg, float* __restrict__ exp_avg_sq, int64_t count,
                                                        const float* __restrict__ p_part, const float* __restrict__ g_part, int nblk,
                                                        float lr, float beta1, float beta2, float adam_eps, float weight_decay,
                                                        float lars_eta, float lars_eps, int lars_clip, int use_lars, float bc1,
                                                        float bc2_sqrt) {
  float scale = 1.0f, wd = weight_decay;
  if (use_lars) {
    double ps = 0.0, gs = 0.0;
    for (int i = 0; i < nblk; ++i) {
      ps += (double)p_part[i];
      gs += (double)g_part[i];
    }
    const float pn = (float)sqrt(ps), gn = (float)sqrt(gs);
    if (pn != 0.f && gn != 0.f) {
      float l = lars_eta * pn / (gn + pn * weight_decay + lars_eps);
      if (lars_clip) l = fminf(l / lr, 1.0f);
      scale = l;
    } else {
      wd = 0.f;  // untouched gradient when either norm is zero (pl_bolts skips the tensor)
    }
  }
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256) {
    float p = param[i];
    float g = grad[i];
    if (use_lars) g = (g + wd * p) * scale;
    else g = g + wd * p;  // plain Adam (L2 form of torch.optim.Adam weight_decay)
    const float m = beta1 * exp_avg[i] + (1.0f - beta1) * g;
    const float v = beta2 * exp_avg_sq[i] + (1.0f - beta2) * g * g;
    exp_avg[i] = m;
    exp_avg_sq[i] = v;
    const float denom = sqrtf(v) / bc2_sqrt + adam_eps;
    param[i] = p - (lr / bc1) * (m / denom);
  }
}

// ---- multi-tensor form: the whole parameter list in two launches (norm partials, update) --------------------------------
// A chunk is OPT_CHUNK consecutive elements of one tensor; the chunk table and the tensor table live in device memory.
constexpr int OPT_CHUNK = 16384;

__global__ __launch_bounds__(256) void opt_norms_kernel(const sh_opt_tensor* __restrict__ tensors, const int2* __restrict__ chunks,
                                                        float* __restrict__ p_part, float* __restrict__ g_part) {
  __shared__ float red[8];
  const int2 ck = chunks[blockIdx.x];
  const sh_opt_tensor t = tensors[ck.x];
  float ps = 0.f, gs = 0.f;
  if (t.use_lars) {
    const int64_t off = (int64_t)ck.y * OPT_CHUNK;
    const int n = (int)((t.count - off) < OPT_CHUNK ? (t.count - off) : OPT_CHUNK);
    const float* __restrict__ p = t.param + off;
    const float* __restrict__ g = t.grad + off;
    for (int i = threadIdx.x; i < n; i += 256) {
      const float a = p[i], b = g[i];
      ps += a * a;
      gs += b * b;
    }
  }
  ps = wave_sum(ps);
  gs = wave_sum(gs);
  if ((threadIdx.x & 63) == 0) {
    red[threadIdx.x >> 6] = ps;
    red[4 + (threadIdx.x >> 6)] = gs;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    p_part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
    g_part[blockIdx.x] = red[4] + red[5] + red[6] + red[7];
  }
}

__global__ __launch_bounds__(256) void opt_update_kernel(const sh_opt_tensor* __restrict__ tensors, const int2* __restrict__ chunks,
                                                         const float* __restrict__ p_part, const float* __restrict__ g_part,
                                                         float beta1, float beta2, float adam_eps, float lars_eta, float lars_eps,
                                                         int lars_clip, const float* __restrict__ found_inf) {
  __shared__ float coef[2];
  // loss-scaled (fp16) training: the gradients of this step overflowed -> the step is skipped, decided here on the device
  // (torch's fused optimizers do the same with GradScaler's found_inf): parameters and both moments stay as they are
  if (found_inf != nullptr && found_inf[0] != 0.f) return;
  const int2 ck = chunks[blockIdx.x];
  const sh_opt_tensor t = tensors[ck.x];
  if (threadIdx.x == 0) {
    float scale = 1.0f, wd = t.weight_decay;
    if (t.use_lars) {
      double ps = 0.0, gs = 0.0;
      for (int i = 0; i < t.n_chunks; ++i) {  // fixed order: deterministic
        ps += (double)p_part[t.first_chunk + i];
        gs += (double)g_part[t.first_chunk + i];
      }
      const float pn = (float)sqrt(ps), gn = (float)sqrt(gs);
      if (pn != 0.f && gn != 0.f) {
        float l = lars_eta * pn / (gn + pn * t.weight_decay + lars_eps);
        if (lars_clip) l = fminf(l / t.lr, 1.0f);
        scale = l;
      } else {
        wd = 0.f;
      }
    }
    coef[0] = scale;
    coef[1] = wd;
  }
  __syncthreads();
  const float scale = coef[0], wd = coef[1];
  const int64_t off = (int64_t)ck.y * OPT_CHUNK;
  const int n = (int)((t.count - off) < OPT_CHUNK ? (t.count - off) : OPT_CHUNK);
  float* __restrict__ param = t.param + off;
  const float* __restrict__ grad = t.grad + off;
  float* __restrict__ exp_avg = t.exp_avg + off;
  float* __restrict__ exp_avg_sq = t.exp_avg_sq + off;
  for (int i = threadIdx.x; i < n; i += 256) {
    const float p = param[i];
    float g = grad[i];
    if (t.use_lars) g = (g + wd * p) * scale;
    else g = g + wd * p;
    const float m = beta1 * exp_avg[i] + (1.0f - beta1) * g;
    const float v = beta2 * exp_avg_sq[i] + (1.0f - beta2) * g * g;
    exp_avg[i] = m;
    exp_avg_sq[i] = v;
    const float denom = sqrtf(v) / t.bc2_sqrt + adam_eps;
    param[i] = p - (t.lr / t.bc1) * (m / denom);
  }
}

}  // namespace sh

using namespace sh;

extern "C" {

int simhand_sumsq_partial(const float* x, int64_t count, float* partial, int nblk, sh_stream_t stream) {
  SH_REQUIRE(x && partial && count >= 0 && nblk >= 1, "sumsq_partial: bad arguments");
  ProfScope ps(SH_PROF_OPT, (hipStream_t)stream, 0, (double)count * 4);
  sumsq_partial_kernel<<<nblk, 256, 0, (hipStream_t)stream>>>(x, count, partial);
  return check_launch("sumsq_partial");
}

int simhand_lars_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t count,
                           const float* p_sumsq, const float* g_sumsq, int nblk_norm, float lr, float beta1, float beta2,
                           float adam_eps, float weight_decay, float lars_eta, float lars_eps, int lars_clip, int use_lars,
                           int step, sh_stream_t stream) {
  SH_REQUIRE(param && grad && exp_avg && exp_avg_sq, "lars_adam_step: NULL pointer");
  SH_REQUIRE(!use_lars || (p_sumsq && g_sumsq && nblk_norm >= 1), "lars_adam_step: LARS needs the norm partials");
  SH_REQUIRE(step >= 1, "lars_adam_step: step counts from 1");
  const float bc1 = 1.0f - powf(beta1, (float)step);
  const float bc2_sqrt = sqrtf(1.0f - powf(beta2, (float)step));
  int64_t g = (count + 255) / 256;
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  ProfScope ps(SH_PROF_OPT, (hipStream_t)stream, 0, (double)count * 28);
  lars_adam_kernel<<<(int)g, 256, 0, (hipStream_t)stream>>>(param, grad, exp_avg, exp_avg_sq, count, p_sumsq, g_sumsq, nblk_norm, lr,
                                                            beta1, beta2, adam_eps, weight_decay, lars_eta, lars_eps, lars_clip, use_lars,
                                                            bc1, bc2_sqrt);
  return check_launch("lars_adam_step");
}

int simhand_opt_chunk_elems(void) { return OPT_CHUNK; }

int simhand_lars_adam_multi_guarded(const sh_opt_tensor* tensors, int n_tensors, const int32_t* chunks, int n_chunks, float* norm_partials,
                                    float beta1, float beta2, float adam_eps, float lars_eta, float lars_eps, int lars_clip,
                                    int64_t total_elems, const float* found_inf, sh_stream_t stream) {
  SH_REQUIRE(tensors && chunks && norm_partials, "lars_adam_multi: NULL pointer");
  SH_REQUIRE(n_tensors >= 1 && n_chunks >= 1, "lars_adam_multi: empty tables");
  hipStream_t st = (hipStream_t)stream;
  ProfScope ps(SH_PROF_OPT, st, 0, (double)total_elems * 36);
  opt_norms_kernel<<<n_chunks, 256, 0, st>>>(tensors, (const int2*)chunks, norm_partials, norm_partials + n_chunks);
  opt_update_kernel<<<n_chunks, 256, 0, st>>>(tensors, (const int2*)chunks, norm_partials, norm_partials + n_chunks, beta1, beta2,
                                              adam_eps, lars_eta, lars_eps, lars_clip, found_inf);
  return check_launch("lars_adam_multi");
}

int simhand_lars_adam_multi(const sh_opt_tensor* tensors, int n_tensors, const int32_t* chunks, int n_chunks, float* norm_partials,
                            float beta1, float beta2, float adam_eps, float lars_eta, float lars_eps, int lars_clip,
                            int64_t total_elems, sh_stream_t stream) {
  return simhand_lars_adam_multi_guarded(tensors, n_tensors, chunks, n_chunks, norm_partials, beta1, beta2, adam_eps, lars_eta, lars_eps,
                                         lars_clip, total_elems, nullptr, stream);
}

}  // extern "C"
