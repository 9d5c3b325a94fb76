// Pooling, layout / dtype transforms and the stem im2col.  All HBM-bound streaming kernels
// with 16-B vector accesses on the NHWC side.
//
// Replaces (reference): nn.MaxPool2d(3, 2, 1) and nn.AdaptiveAvgPool2d(1) of torchvision's
// ResNet (src/models/resnet_model.py:17-26), `z.flatten(start_dim=1)` (:53), and the implicit
// NCHW / OIHW tensor layouts of torch.nn.Conv2d (the 7x7 stride-2 stem is lowered to
// im2col + GEMM because its 3 input channels cannot feed a k-contiguous MFMA operand).
#include "common.h"

namespace sh {

static inline int stream_grid(int64_t n) {
  int64_t g = (n + 255) / 256;
  if (g > 16384) g = 16384;
  if (g < 1) g = 1;
  return (int)g;
}

// ---- MaxPool 3x3 s2 p1: forward stores the winning tap (0..8) per element ------------------
template <typename T>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, uint8_t* __restrict__ idx,
                                                          int n, int h, int w, int c, int ho, int wo) {
  constexpr int VE = Vec16<T>::N;
  const int cvecs = c / VE;
  const int64_t total = (int64_t)n * ho * wo * cvecs;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int cv = (int)(i % cvecs);
    int64_t t = i / cvecs;
    const int ow = (int)(t % wo);
    t /= wo;
    const int oh = (int)(t % ho);
    const int img = (int)(t / ho);
    float best[VE];
    int bi[VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) {
      best[e] = -INFINITY;
      bi[e] = 0;
    }
    // scan order kh then kw, strict '>' (first maximum wins) like ATen's max_pool2d
    for (int kh = 0; kh < 3; ++kh) {
      const int ih = oh * 2 - 1 + kh;
      if ((unsigned)ih >= (unsigned)h) continue;
      for (int kw = 0; kw < 3; ++kw) {
        const int iw = ow * 2 - 1 + kw;
        if ((unsigned)iw >= (unsigned)w) continue;
        float v[VE];
        Vec16<T>::load(x + (((int64_t)img * h + ih) * w + iw) * c + cv * VE, v);
#pragma unroll
        for (int e = 0; e < VE; ++e)
          if (v[e] > best[e] || v[e] != v[e]) {
            best[e] = v[e];
            bi[e] = kh * 3 + kw;
          }
      }
    }
    Vec16<T>::store(y + i * VE, best);
    uint8_t* ip = idx + i * VE;
#pragma unroll
    for (int e = 0; e < VE; ++e) ip[e] = (uint8_t)bi[e];
  }
}

// gather form: every input element sums dy of the (<= 4) windows whose stored winner is itself
template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const T* __restrict__ dy, const uint8_t* __restrict__ idx,
                                                          T* __restrict__ dx, int n, int h, int w, int c, int ho, int wo) {
  constexpr int VE = Vec16<T>::N;
  const int cvecs = c / VE;
  const int64_t total = (int64_t)n * h * w * cvecs;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int cv = (int)(i % cvecs);
    int64_t t = i / cvecs;
    const int iw = (int)(t % w);
    t /= w;
    const int ih = (int)(t % h);
    const int img = (int)(t / h);
    float acc[VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) acc[e] = 0.f;
    // windows oh with oh*2-1+kh == ih  ->  kh = ih + 1 - 2*oh in [0,2]
    for (int oh = ih / 2; oh <= (ih + 1) / 2; ++oh) {
      const int kh = ih + 1 - 2 * oh;
      if (kh < 0 || kh > 2 || oh >= ho) continue;
      for (int ow = iw / 2; ow <= (iw + 1) / 2; ++ow) {
        const int kw = iw + 1 - 2 * ow;
        if (kw < 0 || kw > 2 || ow >= wo) continue;
        const int64_t o = ((((int64_t)img * ho + oh) * wo + ow) * cvecs + cv) * VE;
        float g[VE];
        Vec16<T>::load(dy + o, g);
        const uint8_t* ip = idx + o;
        const int me = kh * 3 + kw;
#pragma unroll
        for (int e = 0; e < VE; ++e) acc[e] += ip[e] == me ? g[e] : 0.f;
      }
    }
    Vec16<T>::store(dx + i * VE, acc);
  }
}

// ---- every second pixel of every second row (the input a stride-2 1x1 convolution actually reads) -------------------
template <typename T>
__global__ __launch_bounds__(256) void subsample2_kernel(const T* __restrict__ x, T* __restrict__ y, int n, int h, int w, int c, int ho,
                                                         int wo) {
  constexpr int VE = Vec16<T>::N;
  const int cvecs = c / VE;
  const int64_t total = (int64_t)n * ho * wo * cvecs;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int cv = (int)(i % cvecs);
    int64_t t = i / cvecs;
    const int ow = (int)(t % wo);
    t /= wo;
    const int oh = (int)(t % ho);
    const int img = (int)(t / ho);
    *reinterpret_cast<uint4*>(y + i * VE) = *reinterpret_cast<const uint4*>(x + (((int64_t)img * h + 2 * oh) * w + 2 * ow) * c + cv * VE);
  }
}

// dx[n][2i][2j][:] = gate(dx[n][2i][2j][:] + src[n][i][j][:]): the transpose of subsample2 with accumulation; gate = the
// consumer block's ReLU bit mask [pixel of dx][c / VE] (optional).  Used by the folded shortcut backward: its input
// gradient is computed densely at the output resolution and lands on the even pixels of the block-input gradient.
template <typename T>
__global__ __launch_bounds__(256) void scatter2_add_kernel(const T* __restrict__ src, T* __restrict__ dx, const uint8_t* __restrict__ mask,
                                                           int n, int h, int w, int c, int ho, int wo) {
  constexpr int VE = Vec16<T>::N;
  const int cvecs = c / VE;
  const int64_t total = (int64_t)n * ho * wo * cvecs;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int cv = (int)(i % cvecs);
    int64_t t = i / cvecs;
    const int ow = (int)(t % wo);
    t /= wo;
    const int oh = (int)(t % ho);
    const int img = (int)(t / ho);
    const int64_t pix = ((int64_t)img * h + 2 * oh) * w + 2 * ow;
    float a[VE], b[VE];
    Vec16<T>::load(src + i * VE, a);
    Vec16<T>::load(dx + pix * c + cv * VE, b);
    const unsigned bits = mask ? mask[pix * cvecs + cv] : 0xffu;
#pragma unroll
    for (int e = 0; e < VE; ++e) {
      const float v = Elem<T>::kDtype == SH_BF16 ? bf16_to_f32(f32_to_bf16(a[e] + b[e])) : a[e] + b[e];
      b[e] = (bits >> e) & 1u ? v : 0.f;
    }
    Vec16<T>::store(dx + pix * c + cv * VE, b);
  }
}

// ---- global average pool ---------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void avgpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int n, int hw, int c) {
  constexpr int VE = Vec16<T>::N;
  const int cvecs = c / VE;
  const int64_t total = (int64_t)n * cvecs;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int cv = (int)(i % cvecs);
    const int img = (int)(i / cvecs);
    float acc[VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) acc[e] = 0.f;
    for (int p = 0; p < hw; ++p) {
      float v[VE];
      Vec16<T>::load(x + ((int64_t)img * hw + p) * c + cv * VE, v);
#pragma unroll
      for (int e = 0; e < VE; ++e) acc[e] += v[e];
    }
#pragma unroll
    for (int e = 0; e < VE; ++e) acc[e] /= (float)hw;
    Vec16<T>::store(y + i * VE, acc);
  }
}

// mask (optional): ReLU bit mask [n * hw][c / VE] of the tensor the pooled map came from -- the gradient leaves already gated (what the
// folded BatchNorm backward of the last block wants: avgpool_bwd + apply_bitmask in one pass, same bits)
template <typename T>
__global__ __launch_bounds__(256) void avgpool_bwd_kernel(const T* __restrict__ dy, const uint8_t* __restrict__ mask, T* __restrict__ dx, int n,
                                                          int hw, int c) {
  constexpr int VE = Vec16<T>::N;
  const int cvecs = c / VE;
  const int64_t total = (int64_t)n * hw * cvecs;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int cv = (int)(i % cvecs);
    const int img = (int)(i / ((int64_t)hw * cvecs));
    float g[VE];
    Vec16<T>::load(dy + ((int64_t)img * cvecs + cv) * VE, g);
    const unsigned bits = mask != nullptr ? mask[i] : 0xffu;
#pragma unroll
    for (int e = 0; e < VE; ++e) {
      g[e] /= (float)hw;
      g[e] = (bits >> e) & 1u ? g[e] : 0.f;   // (the store rounds: the same bits apply_bitmask would have kept)
    }
    Vec16<T>::store(dx + i * VE, g);
  }
}

// ---- im2col for small-cin convolutions (the stem): NCHW fp32 image -> [N*Ho*Wo][k_pad] ----------
// column k = (c*R + r)*S + s  (= the OIHW flattening of the filter, so the weight matrix is the parameter
// itself, row-padded), zero-filled for k >= Cin*R*S and for the halo.  RS > 0: compile-time R = S = RS
// (the divisions become multiply-shifts; the 7x7 stem), RS = 0: runtime R, S.
template <typename T, int RS>
__global__ __launch_bounds__(256) void im2col_nchw_kernel(const float* __restrict__ x, T* __restrict__ col, int n, int cin, int h,
                                                          int w, int Rr, int Sr, int stride, int pad, int ho, int wo, int k_pad) {
  constexpr int VE = Vec16<T>::N;
  const int R = RS > 0 ? RS : Rr, S = RS > 0 ? RS : Sr;
  const unsigned kvecs = (unsigned)(k_pad / VE);
  const int kreal = R * S * cin;
  const unsigned hw_o = (unsigned)(ho * wo);
  const int64_t total = (int64_t)n * hw_o * kvecs;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const unsigned pixrow = (unsigned)(i / kvecs);          // < 2^31 (checked on the host)
    const int kv = (int)((unsigned)i - pixrow * kvecs);
    const unsigned img = pixrow / hw_o;
    const unsigned rem = pixrow - img * hw_o;
    const int oh = (int)(rem / (unsigned)wo), ow = (int)(rem - (unsigned)oh * (unsigned)wo);
    const int ih0 = oh * stride - pad, iw0 = ow * stride - pad;
    const float* __restrict__ ximg = x + (int64_t)img * cin * h * w;
    float v[VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) {
      const int k = kv * VE + e;
      float val = 0.f;
      if (k < kreal) {
        const int c = k / (R * S);
        const int rs = k - c * (R * S);
        const int r = rs / S, s2 = rs - r * S;
        const int ih = ih0 + r, iw = iw0 + s2;
        if ((unsigned)ih < (unsigned)h && (unsigned)iw < (unsigned)w) val = ximg[(c * h + ih) * w + iw];
      }
      v[e] = val;
    }
    Vec16<T>::store(col + i * VE, v);
  }
}

// 7x7 / stride 2 / pad 3 / cin 3 stem, k_pad = 192: the gather runs with LANES ALONG THE OUTPUT ROW (adjacent lanes read
// addresses 8 B apart: a wave touches 4-5 cache lines per load instead of 64), the [64 pixel][192] tile is transposed
// through LDS (odd dword row stride -> conflict-free 2/4-B writes) and leaves as whole 384/768-B rows of 16-B stores.
template <typename T>
__global__ __launch_bounds__(256) void im2col_stem7_kernel(const float* __restrict__ x, T* __restrict__ col, int n, int h, int w,
                                                           int ho, int wo) {
  constexpr int KP = 192, KR = 147, TP = 64;
  constexpr int ROW_DW = KP * (int)sizeof(T) / 4 + 1;  // odd -> lanes (= pixels) hit distinct banks
  __shared__ unsigned tile[TP * ROW_DW];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const long long npix = (long long)n * ho * wo;
  const long long p0 = (long long)blockIdx.x * TP;
  const long long pp = p0 + lane;
  const bool pok = pp < npix;
  const unsigned pr = (unsigned)(pok ? pp : 0);
  const unsigned hw_o = (unsigned)(ho * wo);
  const unsigned img = pr / hw_o;
  const unsigned rem = pr - img * hw_o;
  const int oh = (int)(rem / (unsigned)wo), ow = (int)(rem - (unsigned)oh * (unsigned)wo);
  const int ih0 = oh * 2 - 3, iw0 = ow * 2 - 3;
  const float* __restrict__ ximg = x + (long long)img * 3 * h * w;
  T* trow = reinterpret_cast<T*>(tile + lane * ROW_DW);
  for (int k = wv; k < KP; k += 4) {  // k is wave-uniform: (c, r, s) are scalars
    float val = 0.f;
    if (k < KR) {
      const int c = k / 49, rs = k - c * 49, r = rs / 7, s2 = rs - r * 7;
      const int ih = ih0 + r, iw = iw0 + s2;
      if (pok && (unsigned)ih < (unsigned)h && (unsigned)iw < (unsigned)w) val = ximg[(c * h + ih) * w + iw];
    }
    Elem<T>::store(trow + k, val);
  }
  __syncthreads();
  constexpr int CPR = KP * (int)sizeof(T) / 16;  // 16-B chunks per row
  for (int id = threadIdx.x; id < TP * CPR; id += 256) {
    const int pix = id / CPR, ch = id - pix * CPR;
    if (p0 + pix >= npix) continue;
    const unsigned* src = tile + pix * ROW_DW + ch * 4;
    const uint4 v = make_uint4(src[0], src[1], src[2], src[3]);
    *reinterpret_cast<uint4*>(reinterpret_cast<char*>(col) + ((p0 + pix) * CPR + ch) * 16) = v;
  }
}

// ---- weights: OIHW fp32 -> [K][k_pad] (KRSC rows, zero padded) / [C][R][S][K] ; and back ---------------
template <typename T>
__global__ __launch_bounds__(256) void oihw_to_krsc_kernel(const float* __restrict__ src, T* __restrict__ dst, int k, int c, int r,
                                                           int s, int k_pad) {
  const int64_t total = (int64_t)k * k_pad;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int col = (int)(i % k_pad);
    const int ko = (int)(i / k_pad);
    float v = 0.f;
    if (col < r * s * c) {
      const int ci = col % c;
      const int rs = col / c;
      v = src[(((int64_t)ko * c + ci) * r + rs / s) * s + rs % s];
    }
    Elem<T>::store(dst + i, v);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void oihw_to_crsk_kernel(const float* __restrict__ src, T* __restrict__ dst, int k, int c, int r,
                                                           int s) {
  const int64_t total = (int64_t)k * c * r * s;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    // dst index = ((ci*R + rr)*S + ss)*K + ko
    const int ko = (int)(i % k);
    int64_t t = i / k;
    const int ss = (int)(t % s);
    t /= s;
    const int rr = (int)(t % r);
    const int ci = (int)(t / r);
    Elem<T>::store(dst + i, src[(((int64_t)ko * c + ci) * r + rr) * s + ss]);
  }
}

// every conv weight of the net in ONE launch (was two launches per convolution and step: ~0.5 ms of 5-us kernels): block b works
// on PACK_CHUNK consecutive KRSC elements of tensor chunks[b].x, starting at element chunks[b].y * PACK_CHUNK, and writes the
// KRSC copy and -- when the item has one -- the CRSK copy (the data-gradient operand) from a single read of the fp32 master
constexpr int PACK_CHUNK = 4096;
template <typename T>
__global__ __launch_bounds__(256) void pack_multi_kernel(const sh_pack_item* __restrict__ items, const int2* __restrict__ chunks) {
  const int2 ck = chunks[blockIdx.x];
  const sh_pack_item it = items[ck.x];
  const int c = it.c, r = it.r, s = it.s, k = it.k;
  const int row = c * r * s;
  const int64_t total = (int64_t)k * row;
  const float* __restrict__ src = it.src;
  T* __restrict__ krsc = reinterpret_cast<T*>(it.krsc);
  T* __restrict__ crsk = reinterpret_cast<T*>(it.crsk);
  const int64_t base = (int64_t)ck.y * PACK_CHUNK;
  for (int j = threadIdx.x; j < PACK_CHUNK; j += 256) {
    const int64_t i = base + j;
    if (i >= total) break;
    const int col = (int)(i % row), ko = (int)(i / row);
    const int ci = col % c, rs = col / c;
    const float v = src[(((int64_t)ko * c + ci) * r + rs / s) * s + rs % s];
    Elem<T>::store(krsc + i, v);
  }
}

// CRSK copies from the KRSC copies just written, as 64 x 64 tile transposes through LDS (both sides in 128-B+ runs; a scatter from
// the KRSC index space costs 0.3 ms per step): block = (item, tile); tile = (tap, 64 output channels, 64 input channels)
template <typename T>
__global__ __launch_bounds__(256) void pack_crsk_tiles_kernel(const sh_pack_item* __restrict__ items, const int2* __restrict__ tiles) {
  __shared__ T tile[64][64 + 2];
  const int2 tk = tiles[blockIdx.x];
  const sh_pack_item it = items[tk.x];
  const int c = it.c, k = it.k, rs_n = it.r * it.s;
  const int ct = c / 64, kt = k / 64;
  int t = tk.y;
  const int ci0 = (t % ct) * 64;
  t /= ct;
  const int ko0 = (t % kt) * 64;
  const int rs = t / kt;
  const T* __restrict__ krsc = reinterpret_cast<const T*>(it.krsc);
  T* __restrict__ crsk = reinterpret_cast<T*>(it.crsk);
  const int col = threadIdx.x & 63, row4 = threadIdx.x >> 6;
#pragma unroll 4
  for (int rr = row4; rr < 64; rr += 4) tile[rr][col] = krsc[((int64_t)(ko0 + rr) * rs_n + rs) * c + ci0 + col];
  __syncthreads();
#pragma unroll 4
  for (int rr = row4; rr < 64; rr += 4) crsk[((int64_t)(ci0 + rr) * rs_n + rs) * k + ko0 + col] = tile[col][rr];
}

__global__ __launch_bounds__(256) void krsc_to_oihw_kernel(const float* __restrict__ src, float* __restrict__ dst, int k, int c,
                                                           int r, int s, int k_pad) {
  const int64_t total = (int64_t)k * c * r * s;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    // dst index = ((ko*C + ci)*R + rr)*S + ss
    const int ss = (int)(i % s);
    int64_t t = i / s;
    const int rr = (int)(t % r);
    t /= r;
    const int ci = (int)(t % c);
    const int ko = (int)(t / c);
    dst[i] = src[(int64_t)ko * k_pad + ((int64_t)rr * s + ss) * c + ci];
  }
}

template <typename TS, typename TD>
__global__ __launch_bounds__(256) void cast_kernel(const TS* __restrict__ src, TD* __restrict__ dst, int64_t count) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256)
    Elem<TD>::store(dst + i, Elem<TS>::load(src + i));
}

template <typename T>
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ src, T* __restrict__ dst, int n, int c, int h,
                                                           int w, int c_pad) {
  const int64_t total = (int64_t)n * h * w * c_pad;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int ci = (int)(i % c_pad);
    int64_t t = i / c_pad;
    const int iw = (int)(t % w);
    t /= w;
    const int ih = (int)(t % h);
    const int img = (int)(t / h);
    Elem<T>::store(dst + i, ci < c ? src[(((int64_t)img * c + ci) * h + ih) * w + iw] : 0.f);
  }
}

}  // namespace sh

using namespace sh;

// ---- direct stem (conv_igemm.hip simhand_stem_conv_fwd): NCHW fp32 -> zero-padded NHWC4 [N][hp][wp][4] ------------------
// one thread per padded pixel; input pixel (ih, iw) lands at (ih + 3, iw + 3); borders and channel 3 are zeros
template <typename T>
__global__ __launch_bounds__(256) void stem_pad_kernel(const float* __restrict__ x, T* __restrict__ xp, int n, int h, int w, int hp,
                                                       int wp) {
  const int64_t total = (int64_t)n * hp * wp;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int pw = (int)(i % wp);
    const int64_t t = i / wp;
    const int ph = (int)(t % hp);
    const int64_t img = t / hp;
    const int ih = ph - 3, iw = pw - 3;
    float v0 = 0.f, v1 = 0.f, v2 = 0.f;
    if ((unsigned)ih < (unsigned)h && (unsigned)iw < (unsigned)w) {
      const float* src = x + ((img * 3) * h + ih) * (int64_t)w + iw;
      v0 = src[0];
      v1 = src[(int64_t)h * w];
      v2 = src[2 * (int64_t)h * w];
    }
    if (sizeof(T) == 4) {
      *reinterpret_cast<float4*>(xp + i * 4) = make_float4(v0, v1, v2, 0.f);
    } else {
      uint2 o;
      o.x = pack_bf16x2(v0, v1);
      o.y = (unsigned)f32_to_bf16(v2);
      *reinterpret_cast<uint2*>(xp + i * 4) = o;
    }
  }
}

// OIHW fp32 [64][3][7][7] -> [64][256]: column r*32 + tap*4 + c (rows r = 7, taps 7 and channel 3 are zeros)
template <typename T>
__global__ __launch_bounds__(256) void stem_pack_w_kernel(const float* __restrict__ w, T* __restrict__ wp) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= 64 * 256) return;
  const int k = i >> 8, col = i & 255, r = col >> 5, t = (col & 31) >> 2, c = col & 3;
  Elem<T>::store(wp + i, (r < 7 && t < 7 && c < 3) ? w[((k * 3 + c) * 7 + r) * 7 + t] : 0.f);
}

#define SH_DISPATCH(dtype, CALL_F32, CALL_BF16) \
  do {                                          \
    if ((dtype) == SH_F32) { CALL_F32; }        \
    else { CALL_BF16; }                         \
  } while (0)

extern "C" {

static int vec_ok(int c, int dtype, const char* who) {
  SH_REQUIRE(dtype == SH_F32 || dtype == SH_BF16, "%s: bad dtype %d", who, dtype);
  const int ve = dtype == SH_F32 ? 4 : 8;
  SH_REQUIRE(c % ve == 0, "%s: channel count %d not a multiple of %d", who, c, ve);
  return 0;
}

int simhand_maxpool3x3s2_fwd(const void* x, void* y, uint8_t* idx, int n, int h, int w, int c, int dtype, sh_stream_t stream) {
  SH_REQUIRE(x && y && idx, "maxpool_fwd: NULL pointer");
  if (vec_ok(c, dtype, "maxpool_fwd")) return 1;
  const int ho = (h + 2 - 3) / 2 + 1, wo = (w + 2 - 3) / 2 + 1;
  const int ve = dtype == SH_F32 ? 4 : 8;
  const int64_t total = (int64_t)n * ho * wo * (c / ve);
  hipStream_t s = (hipStream_t)stream;
  const double es = dtype == SH_F32 ? 4 : 2;
  ProfScope ps(SH_PROF_POOL, s, 0, es * ((double)n * h * w * c + (double)n * ho * wo * c) + (double)n * ho * wo * c);
  SH_DISPATCH(dtype, (maxpool_fwd_kernel<float><<<stream_grid(total), 256, 0, s>>>((const float*)x, (float*)y, idx, n, h, w, c, ho, wo)),
              (maxpool_fwd_kernel<bf16_t><<<stream_grid(total), 256, 0, s>>>((const bf16_t*)x, (bf16_t*)y, idx, n, h, w, c, ho, wo)));
  return check_launch("maxpool_fwd");
}

int simhand_maxpool3x3s2_bwd(const void* dy, const uint8_t* idx, void* dx, int n, int h, int w, int c, int dtype, sh_stream_t stream) {
  SH_REQUIRE(dy && idx && dx, "maxpool_bwd: NULL pointer");
  if (vec_ok(c, dtype, "maxpool_bwd")) return 1;
  const int ho = (h + 2 - 3) / 2 + 1, wo = (w + 2 - 3) / 2 + 1;
  const int ve = dtype == SH_F32 ? 4 : 8;
  const int64_t total = (int64_t)n * h * w * (c / ve);
  hipStream_t s = (hipStream_t)stream;
  const double es = dtype == SH_F32 ? 4 : 2;
  ProfScope ps(SH_PROF_POOL, s, 0, es * ((double)n * h * w * c + (double)n * ho * wo * c) + (double)n * ho * wo * c);
  SH_DISPATCH(dtype, (maxpool_bwd_kernel<float><<<stream_grid(total), 256, 0, s>>>((const float*)dy, idx, (float*)dx, n, h, w, c, ho, wo)),
              (maxpool_bwd_kernel<bf16_t><<<stream_grid(total), 256, 0, s>>>((const bf16_t*)dy, idx, (bf16_t*)dx, n, h, w, c, ho, wo)));
  return check_launch("maxpool_bwd");
}

int simhand_subsample2(const void* x, void* y, int n, int h, int w, int c, int dtype, sh_stream_t stream) {
  SH_REQUIRE(x && y, "subsample2: NULL pointer");
  if (vec_ok(c, dtype, "subsample2")) return 1;
  const int ho = (h + 1) / 2, wo = (w + 1) / 2;
  const int ve = dtype == SH_F32 ? 4 : 8;
  const int64_t total = (int64_t)n * ho * wo * (c / ve);
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SH_PROF_MISC, s, 0, 2.0 * (double)n * ho * wo * c * (dtype == SH_F32 ? 4 : 2));
  SH_DISPATCH(dtype, (subsample2_kernel<float><<<stream_grid(total), 256, 0, s>>>((const float*)x, (float*)y, n, h, w, c, ho, wo)),
              (subsample2_kernel<bf16_t><<<stream_grid(total), 256, 0, s>>>((const bf16_t*)x, (bf16_t*)y, n, h, w, c, ho, wo)));
  return check_launch("subsample2");
}

int simhand_scatter2_add(const void* src, void* dx, const uint8_t* mask, int n, int h, int w, int c, int dtype, sh_stream_t stream) {
  SH_REQUIRE(src && dx, "scatter2_add: NULL pointer");
  if (vec_ok(c, dtype, "scatter2_add")) return 1;
  const int ho = (h + 1) / 2, wo = (w + 1) / 2;
  const int ve = dtype == SH_F32 ? 4 : 8;
  const int64_t total = (int64_t)n * ho * wo * (c / ve);
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SH_PROF_MISC, s, 0, 3.0 * (double)n * ho * wo * c * (dtype == SH_F32 ? 4 : 2));
  SH_DISPATCH(dtype, (scatter2_add_kernel<float><<<stream_grid(total), 256, 0, s>>>((const float*)src, (float*)dx, mask, n, h, w, c, ho, wo)),
              (scatter2_add_kernel<bf16_t><<<stream_grid(total), 256, 0, s>>>((const bf16_t*)src, (bf16_t*)dx, mask, n, h, w, c, ho, wo)));
  return check_launch("scatter2_add");
}

int simhand_avgpool_fwd(const void* x, void* y, int n, int hw, int c, int dtype, sh_stream_t stream) {
  SH_REQUIRE(x && y, "avgpool_fwd: NULL pointer");
  if (vec_ok(c, dtype, "avgpool_fwd")) return 1;
  const int ve = dtype == SH_F32 ? 4 : 8;
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SH_PROF_POOL, s, 0, (double)n * hw * c * (dtype == SH_F32 ? 4 : 2));
  const int64_t total = (int64_t)n * (c / ve);
  SH_DISPATCH(dtype, (avgpool_fwd_kernel<float><<<stream_grid(total), 256, 0, s>>>((const float*)x, (float*)y, n, hw, c)),
              (avgpool_fwd_kernel<bf16_t><<<stream_grid(total), 256, 0, s>>>((const bf16_t*)x, (bf16_t*)y, n, hw, c)));
  return check_launch("avgpool_fwd");
}

int simhand_avgpool_bwd(const void* dy, void* dx, int n, int hw, int c, int dtype, sh_stream_t stream) {
  return simhand_avgpool_bwd_masked(dy, nullptr, dx, n, hw, c, dtype, stream);
}

int simhand_avgpool_bwd_masked(const void* dy, const uint8_t* mask, void* dx, int n, int hw, int c, int dtype, sh_stream_t stream) {
  SH_REQUIRE(dy && dx, "avgpool_bwd: NULL pointer");
  if (vec_ok(c, dtype, "avgpool_bwd")) return 1;
  const int ve = dtype == SH_F32 ? 4 : 8;
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SH_PROF_POOL, s, 0, (double)n * hw * c * (dtype == SH_F32 ? 4 : 2));
  const int64_t total = (int64_t)n * hw * (c / ve);
  SH_DISPATCH(dtype, (avgpool_bwd_kernel<float><<<stream_grid(total), 256, 0, s>>>((const float*)dy, mask, (float*)dx, n, hw, c)),
              (avgpool_bwd_kernel<bf16_t><<<stream_grid(total), 256, 0, s>>>((const bf16_t*)dy, mask, (bf16_t*)dx, n, hw, c)));
  return check_launch("avgpool_bwd");
}

int simhand_im2col_nchw_f32(const float* x, void* col, int n, int cin, int h, int w, int r, int s, int stride, int pad, int k_pad,
                            int dtype, sh_stream_t stream) {
  SH_REQUIRE(x && col, "im2col: NULL pointer");
  if (vec_ok(k_pad, dtype, "im2col")) return 1;
  SH_REQUIRE(k_pad >= r * s * cin, "im2col: k_pad=%d < r*s*cin=%d", k_pad, r * s * cin);
  const int ho = (h + 2 * pad - r) / stride + 1, wo = (w + 2 * pad - s) / stride + 1;
  const int ve = dtype == SH_F32 ? 4 : 8;
  const int64_t total = (int64_t)n * ho * wo * (k_pad / ve);
  hipStream_t st = (hipStream_t)stream;
  ProfScope ps(SH_PROF_MISC, st, 0, (double)n * ho * wo * k_pad * (dtype == SH_F32 ? 4 : 2) + (double)n * cin * h * w * 4);
  SH_REQUIRE((int64_t)n * ho * wo < (1ll << 31), "im2col: %lld output pixels exceed the 2^31 index range", (long long)n * ho * wo);
  if (r == 7 && s == 7 && cin == 3 && stride == 2 && pad == 3 && k_pad == 192) {
    const int grid = (int)(((int64_t)n * ho * wo + 63) / 64);
    SH_DISPATCH(dtype, (im2col_stem7_kernel<float><<<grid, 256, 0, st>>>(x, (float*)col, n, h, w, ho, wo)),
                (im2col_stem7_kernel<bf16_t><<<grid, 256, 0, st>>>(x, (bf16_t*)col, n, h, w, ho, wo)));
  } else if (r == 7 && s == 7) {
    SH_DISPATCH(dtype, (im2col_nchw_kernel<float, 7><<<stream_grid(total), 256, 0, st>>>(x, (float*)col, n, cin, h, w, r, s, stride, pad, ho, wo, k_pad)),
                (im2col_nchw_kernel<bf16_t, 7><<<stream_grid(total), 256, 0, st>>>(x, (bf16_t*)col, n, cin, h, w, r, s, stride, pad, ho, wo, k_pad)));
  } else {
    SH_DISPATCH(dtype, (im2col_nchw_kernel<float, 0><<<stream_grid(total), 256, 0, st>>>(x, (float*)col, n, cin, h, w, r, s, stride, pad, ho, wo, k_pad)),
                (im2col_nchw_kernel<bf16_t, 0><<<stream_grid(total), 256, 0, st>>>(x, (bf16_t*)col, n, cin, h, w, r, s, stride, pad, ho, wo, k_pad)));
  }
  return check_launch("im2col");
}

int simhand_stem_pad_input(const float* x, void* xp, int n, int h, int w, int dtype, sh_stream_t stream) {
  SH_REQUIRE(x && xp, "stem_pad_input: NULL pointer");
  SH_REQUIRE(n >= 1 && h >= 1 && w >= 1, "stem_pad_input: bad shape");
  const int hp = h + 8, wp = (w + 8 + 7) / 8 * 8;
  const int64_t total = (int64_t)n * hp * wp;
  hipStream_t st = (hipStream_t)stream;
  ProfScope ps(SH_PROF_MISC, st, 0, (double)total * 4 * (dtype == SH_F32 ? 4 : 2) + (double)n * 3 * h * w * 4);
  SH_DISPATCH(dtype, (stem_pad_kernel<float><<<stream_grid(total), 256, 0, st>>>(x, (float*)xp, n, h, w, hp, wp)),
              (stem_pad_kernel<bf16_t><<<stream_grid(total), 256, 0, st>>>(x, (bf16_t*)xp, n, h, w, hp, wp)));
  return check_launch("stem_pad_input");
}

int simhand_stem_pack_weights(const float* w_oihw, void* wp, int dtype, sh_stream_t stream) {
  SH_REQUIRE(w_oihw && wp, "stem_pack_weights: NULL pointer");
  hipStream_t st = (hipStream_t)stream;
  SH_DISPATCH(dtype, (stem_pack_w_kernel<float><<<64, 256, 0, st>>>(w_oihw, (float*)wp)),
              (stem_pack_w_kernel<bf16_t><<<64, 256, 0, st>>>(w_oihw, (bf16_t*)wp)));
  return check_launch("stem_pack_weights");
}

int simhand_nchw_f32_to_nhwc(const float* src, void* dst, int n, int c, int h, int w, int c_pad, int dtype, sh_stream_t stream) {
  SH_REQUIRE(src && dst && c_pad >= c, "nchw_to_nhwc: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  const int64_t total = (int64_t)n * h * w * c_pad;
  ProfScope ps(SH_PROF_MISC, s, 0, (double)total * 6);
  SH_DISPATCH(dtype, (nchw_to_nhwc_kernel<float><<<stream_grid(total), 256, 0, s>>>(src, (float*)dst, n, c, h, w, c_pad)),
              (nchw_to_nhwc_kernel<bf16_t><<<stream_grid(total), 256, 0, s>>>(src, (bf16_t*)dst, n, c, h, w, c_pad)));
  return check_launch("nchw_to_nhwc");
}

int simhand_oihw_f32_to_krsc(const float* src, void* dst, int k, int c, int r, int s, int k_pad, int dtype, sh_stream_t stream) {
  SH_REQUIRE(src && dst && k_pad >= c * r * s, "oihw_to_krsc: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const int64_t total = (int64_t)k * k_pad;
  ProfScope ps(SH_PROF_MISC, st, 0, (double)total * 6);
  SH_DISPATCH(dtype, (oihw_to_krsc_kernel<float><<<stream_grid(total), 256, 0, st>>>(src, (float*)dst, k, c, r, s, k_pad)),
              (oihw_to_krsc_kernel<bf16_t><<<stream_grid(total), 256, 0, st>>>(src, (bf16_t*)dst, k, c, r, s, k_pad)));
  return check_launch("oihw_to_krsc");
}

int simhand_pack_chunk_elems(void) { return PACK_CHUNK; }

int simhand_pack_weights_multi(const sh_pack_item* items, const int32_t* chunks, int n_chunks, const int32_t* crsk_tiles, int n_tiles,
                               int dtype, sh_stream_t stream) {
  SH_REQUIRE(items && chunks && n_chunks >= 1 && (n_tiles == 0 || crsk_tiles), "pack_weights_multi: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  ProfScope ps(SH_PROF_MISC, st, 0, (double)n_chunks * PACK_CHUNK * 8);
  SH_DISPATCH(dtype, (pack_multi_kernel<float><<<n_chunks, 256, 0, st>>>(items, (const int2*)chunks)),
              (pack_multi_kernel<bf16_t><<<n_chunks, 256, 0, st>>>(items, (const int2*)chunks)));
  if (n_tiles > 0) {
    SH_DISPATCH(dtype, (pack_crsk_tiles_kernel<float><<<n_tiles, 256, 0, st>>>(items, (const int2*)crsk_tiles)),
                (pack_crsk_tiles_kernel<bf16_t><<<n_tiles, 256, 0, st>>>(items, (const int2*)crsk_tiles)));
  }
  return check_launch("pack_weights_multi");
}

int simhand_oihw_f32_to_crsk(const float* src, void* dst, int k, int c, int r, int s, int dtype, sh_stream_t stream) {
  SH_REQUIRE(src && dst, "oihw_to_crsk: NULL pointer");
  hipStream_t st = (hipStream_t)stream;
  const int64_t total = (int64_t)k * c * r * s;
  ProfScope ps(SH_PROF_MISC, st, 0, (double)total * 6);
  SH_DISPATCH(dtype, (oihw_to_crsk_kernel<float><<<stream_grid(total), 256, 0, st>>>(src, (float*)dst, k, c, r, s)),
              (oihw_to_crsk_kernel<bf16_t><<<stream_grid(total), 256, 0, st>>>(src, (bf16_t*)dst, k, c, r, s)));
  return check_launch("oihw_to_crsk");
}

int simhand_krsc_f32_to_oihw(const float* src, float* dst, int k, int c, int r, int s, int k_pad, sh_stream_t stream) {
  SH_REQUIRE(src && dst && k_pad >= c * r * s, "krsc_to_oihw: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const int64_t total = (int64_t)k * c * r * s;
  ProfScope ps(SH_PROF_MISC, st, 0, (double)total * 8);
  krsc_to_oihw_kernel<<<stream_grid(total), 256, 0, st>>>(src, dst, k, c, r, s, k_pad);
  return check_launch("krsc_to_oihw");
}

int simhand_cast(const void* src, int src_dtype, void* dst, int dst_dtype, int64_t count, sh_stream_t stream) {
  SH_REQUIRE(src && dst && count >= 0, "cast: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  ProfScope ps(SH_PROF_MISC, s, 0, (double)count * 6);
  if (src_dtype == SH_F32 && dst_dtype == SH_BF16) cast_kernel<float, bf16_t><<<stream_grid(count), 256, 0, s>>>((const float*)src, (bf16_t*)dst, count);
  else if (src_dtype == SH_BF16 && dst_dtype == SH_F32) cast_kernel<bf16_t, float><<<stream_grid(count), 256, 0, s>>>((const bf16_t*)src, (float*)dst, count);
  else if (src_dtype == SH_F32 && dst_dtype == SH_F32) cast_kernel<float, float><<<stream_grid(count), 256, 0, s>>>((const float*)src, (float*)dst, count);
  else if (src_dtype == SH_BF16 && dst_dtype == SH_BF16) cast_kernel<bf16_t, bf16_t><<<stream_grid(count), 256, 0, s>>>((const bf16_t*)src, (bf16_t*)dst, count);
  else {
    sh::set_error("cast: bad dtypes %d -> %d", src_dtype, dst_dtype);
    return 1;
  }
  return check_launch("cast");
}

}  // extern "C"
