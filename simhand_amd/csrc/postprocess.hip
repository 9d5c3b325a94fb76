// Projection post-process: normalize -> translate -> rotate -> normalize on rows of
// 128 floats viewed as 64 2-D points, forward + backward, and the projection stats.
//
// Replaces (reference): HandCLR_W/PeCLR_W.get_transformed_projections
// simhand_w_model.py:56-93 (peclr_w_model.py:53-90), translate_encodings
// src/models/utils.py:661-684, rotate_encoding + get_rotation_2D_matrix :606-658,
// get_projection_stats simhand_w_model.py:138-151.
//
// One wavefront per row: lane k owns point k = (P[row][2k], P[row][2k+1]); the
// four per-row reductions (two norms, range, centre) are 64-lane butterflies, no LDS.
#include "common.h"

namespace sh {

constexpr float kNormEps = 1e-12f;  // F.normalize default eps

struct RowXform {
  float inv_d1;        // 1 / max(||p||, eps)
  float qx, qy;        // after first normalize
  float ux, uy;        // after translate + rotate (before second normalize)
  float a, b;          // cos, sin of theta (fp32-rounded like the reference's rot_mat)
  float n2, d2;        // ||u||, max(||u||, eps)
  float n1;            // ||p||
};

struct PPArgs {
  const int64_t* jx;   // raw collated jitters (int64) or null
  const int64_t* jy;
  const float* tx;     // OR: ready translation factors (translate_encodings' arguments) or null
  const float* ty;
  const double* angle; // degrees or null
  int img_h, img_w;
  int flags;           // SH_PP_* bits
};

__device__ __forceinline__ RowXform row_forward(float px, float py, int row, const PPArgs& a) {
  RowXform r;
  r.inv_d1 = 1.0f;
  r.qx = px;
  r.qy = py;
  r.n1 = 1.0f;
  if (a.flags & SH_PP_NORM_IN) {
    const float n1 = sqrtf(wave_sum(px * px + py * py));
    const float d1 = fmaxf(n1, kNormEps);
    r.n1 = n1;
    r.inv_d1 = 1.0f / d1;
    r.qx = px / d1;
    r.qy = py / d1;
  }
  float x = r.qx, y = r.qy;
  if (a.jx != nullptr || a.tx != nullptr) {
    // -(jitter / float(size)) * (max - min), simhand_w_model.py:68-83, utils.py:674-682
    const float tx = a.tx ? a.tx[row] : -((float)a.jx[row] / (float)a.img_h);
    const float ty = a.ty ? a.ty[row] : -((float)a.jy[row] / (float)a.img_w);
    const float rx = wave_max(x) - wave_min(x);
    const float ry = wave_max(y) - wave_min(y);
    x += tx * rx;
    y += ty * ry;
  }
  r.a = 1.0f;
  r.b = 0.0f;
  if (a.angle != nullptr) {
    // theta = (-angle) * pi / 180 in float64 (collated dtype), matrix rounded to fp32: utils.py:622-631
    const double deg = (a.flags & SH_PP_ANGLE_AS_GIVEN) ? a.angle[row] : -a.angle[row];
    const double th = deg * 3.141592653589793 / 180.0;
    const double alpha = cos(th), beta = sin(th);
    const float cx = wave_sum(x) / 64.0f;
    const float cy = wave_sum(y) / 64.0f;
    const float m20 = (float)((1.0 - alpha) * (double)cx - beta * (double)cy);
    const float m21 = (float)((1.0 - alpha) * (double)cy + beta * (double)cx);
    r.a = (float)alpha;
    r.b = (float)beta;
    const float xr = x * r.a + y * r.b + m20;
    const float yr = -(x * r.b) + y * r.a + m21;
    x = xr;
    y = yr;
  }
  r.ux = x;
  r.uy = y;
  r.n2 = 1.0f;
  r.d2 = 1.0f;
  if (a.flags & SH_PP_NORM_OUT) {
    r.n2 = sqrtf(wave_sum(x * x + y * y));
    r.d2 = fmaxf(r.n2, kNormEps);
  }
  return r;
}

__global__ __launch_bounds__(256) void postprocess_fwd_kernel(const float* __restrict__ P, int N, PPArgs a,
                                                              float* __restrict__ Z) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= N) return;
  const float2 p = *reinterpret_cast<const float2*>(P + (size_t)row * 128 + 2 * lane);
  const RowXform r = row_forward(p.x, p.y, row, a);
  *reinterpret_cast<float2*>(Z + (size_t)row * 128 + 2 * lane) = make_float2(r.ux / r.d2, r.uy / r.d2);
}

__global__ __launch_bounds__(256) void postprocess_bwd_kernel(const float* __restrict__ P, int N, PPArgs a,
                                                              const float* __restrict__ dZ, float* __restrict__ dP) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= N) return;
  const float2 p = *reinterpret_cast<const float2*>(P + (size_t)row * 128 + 2 * lane);
  const float2 g = *reinterpret_cast<const float2*>(dZ + (size_t)row * 128 + 2 * lane);
  const RowXform r = row_forward(p.x, p.y, row, a);
  // second normalize: z = u / d2
  const float zx = r.ux / r.d2, zy = r.uy / r.d2;
  float dux = g.x / r.d2, duy = g.y / r.d2;
  if ((a.flags & SH_PP_NORM_OUT) && r.n2 > kNormEps) {
    const float dot = wave_sum(zx * g.x + zy * g.y);
    dux -= zx * dot / r.d2;
    duy -= zy * dot / r.d2;
  }
  // rotate (centre detached) then translate (range detached): linear part only
  const float dqx = r.a * dux - r.b * duy;
  const float dqy = r.b * dux + r.a * duy;
  // first normalize: q = p / d1
  float dpx = dqx * r.inv_d1, dpy = dqy * r.inv_d1;
  if ((a.flags & SH_PP_NORM_IN) && r.n1 > kNormEps) {
    const float dot = wave_sum(r.qx * dqx + r.qy * dqy);
    dpx -= r.qx * dot * r.inv_d1;
    dpy -= r.qy * dot * r.inv_d1;
  }
  *reinterpret_cast<float2*>(dP + (size_t)row * 128 + 2 * lane) = make_float2(dpx, dpy);
}

__device__ __forceinline__ float wave_lower_median(float v, int lane) {
  // rank = number of elements ordered before this one (ties broken by lane id);
  // torch.median returns the lower of the two middle values -> rank 31 of 64
  int rank = 0;
#pragma unroll 8
  for (int m = 0; m < 64; ++m) {
    const float o = __shfl(v, m);
    rank += (o < v || (o == v && m < lane)) ? 1 : 0;
  }
  const unsigned long long ball = __ballot(rank == 31);
  const int src = __ffsll((long long)ball) - 1;
  return __shfl(v, src < 0 ? 0 : src);
}

__global__ __launch_bounds__(256) void proj_row_stats_kernel(const float* __restrict__ P, int N, float* __restrict__ ws) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= N) return;
  const float2 p = *reinterpret_cast<const float2*>(P + (size_t)row * 128 + 2 * lane);
  float o[8];
  o[0] = wave_sum(p.x) / 64.0f;
  o[1] = wave_lower_median(p.x, lane);
  o[2] = wave_min(p.x);
  o[3] = wave_max(p.x);
  o[4] = wave_sum(p.y) / 64.0f;
  o[5] = wave_lower_median(p.y, lane);
  o[6] = wave_min(p.y);
  o[7] = wave_max(p.y);
  if (lane < 8) {
    float v = o[0];
#pragma unroll
    for (int i = 1; i < 8; ++i) v = lane == i ? o[i] : v;
    ws[(size_t)row * 8 + lane] = v;
  }
}

__global__ __launch_bounds__(1024) void proj_stats_reduce_kernel(const float* __restrict__ ws, int N, float* __restrict__ out) {
  __shared__ double part[128][8];
  const int col = threadIdx.x & 7, grp = threadIdx.x >> 3;
  double v = 0.0;
  for (int r = grp; r < N; r += 128) v += (double)ws[(size_t)r * 8 + col];
  part[grp][col] = v;
  __syncthreads();
  if (threadIdx.x < 8) {
    double s = 0.0;
    for (int i = 0; i < 128; ++i) s += part[i][threadIdx.x];
    out[threadIdx.x] = (float)(s / (double)N);
  }
}

}  // namespace sh

using namespace sh;

extern "C" {

static int pp_make(const void* a, const void* b, int N, const int64_t* jx, const int64_t* jy, const float* tx, const float* ty,
                   const double* angle, int h, int w, int flags, PPArgs* out, const char* who) {
  SH_REQUIRE(a && b, "%s: NULL pointer", who);
  SH_REQUIRE(N >= 1, "%s: N must be >= 1", who);
  SH_REQUIRE((jx == nullptr) == (jy == nullptr), "%s: jitter_x and jitter_y must both be given or both NULL", who);
  SH_REQUIRE((tx == nullptr) == (ty == nullptr), "%s: translate_x and translate_y must both be given or both NULL", who);
  SH_REQUIRE(!(jx && tx), "%s: give raw jitters OR ready translation factors, not both", who);
  SH_REQUIRE(!jx || (h > 0 && w > 0), "%s: bad image size %dx%d", who, h, w);
  out->jx = jx; out->jy = jy; out->tx = tx; out->ty = ty; out->angle = angle;
  out->img_h = h; out->img_w = w; out->flags = flags;
  return 0;
}

int simhand_proj_postprocess_fwd(const float* P, int N, int width, const int64_t* jitter_x, const int64_t* jitter_y, const float* translate_x,
                                 const float* translate_y, const double* angle, int img_h, int img_w, int flags, float* Z,
                                 sh_stream_t stream) {
  PPArgs a;
  SH_REQUIRE(width == SH_PROJ_DIM, "proj_postprocess_fwd: rows must be %d floats wide (64 2-D points), got %d", SH_PROJ_DIM, width);
  if (pp_make(P, Z, N, jitter_x, jitter_y, translate_x, translate_y, angle, img_h, img_w, flags, &a, "proj_postprocess_fwd")) return 1;
  ProfScope ps(SH_PROF_MISC, (hipStream_t)stream, 0, (double)N * 128 * 8);
  postprocess_fwd_kernel<<<ceil_div(N, 4), 256, 0, (hipStream_t)stream>>>(P, N, a, Z);
  return check_launch("proj_postprocess_fwd");
}

int simhand_proj_postprocess_bwd(const float* P, int N, int width, const int64_t* jitter_x, const int64_t* jitter_y, const float* translate_x,
                                 const float* translate_y, const double* angle, int img_h, int img_w, int flags, const float* dZ,
                                 float* dP, sh_stream_t stream) {
  PPArgs a;
  SH_REQUIRE(width == SH_PROJ_DIM, "proj_postprocess_bwd: rows must be %d floats wide (64 2-D points), got %d", SH_PROJ_DIM, width);
  if (pp_make(P, dP, N, jitter_x, jitter_y, translate_x, translate_y, angle, img_h, img_w, flags, &a, "proj_postprocess_bwd")) return 1;
  SH_REQUIRE(dZ, "proj_postprocess_bwd: dZ is NULL");
  ProfScope ps(SH_PROF_MISC, (hipStream_t)stream, 0, (double)N * 128 * 12);
  postprocess_bwd_kernel<<<ceil_div(N, 4), 256, 0, (hipStream_t)stream>>>(P, N, a, dZ, dP);
  return check_launch("proj_postprocess_bwd");
}

int simhand_proj_stats(const float* P, int N, int width, float* row_ws, float* out, sh_stream_t stream) {
  SH_REQUIRE(P && row_ws && out, "proj_stats: NULL pointer");
  SH_REQUIRE(width == SH_PROJ_DIM, "proj_stats: rows must be %d floats wide (64 2-D points), got %d", SH_PROJ_DIM, width);
  SH_REQUIRE(N >= 1, "proj_stats: N must be >= 1");
  ProfScope ps(SH_PROF_MISC, (hipStream_t)stream, 0, (double)N * 128 * 4);
  proj_row_stats_kernel<<<ceil_div(N, 4), 256, 0, (hipStream_t)stream>>>(P, N, row_ws);
  if (check_launch("proj_row_stats")) return 1;
  proj_stats_reduce_kernel<<<1, 1024, 0, (hipStream_t)stream>>>(row_ws, N, out);
  return check_launch("proj_stats_reduce");
}

}  // extern "C"
