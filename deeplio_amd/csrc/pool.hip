// Max pooling (3x3, floor or ceil output size, ATen "first max wins" tie rule), global
// average pooling and the SELayer channel re-weighting.  HBM-bound streaming kernels.
//
// Replaces nn.MaxPool2d at pointseg_net.py:21-46, resnet.py:40, lidar_feat_nets.py:281-299;
// adaptive_avg_pool2d at lidar_feat_nets.py:84-85,258, resnet.py:49, pointseg_modules.py:218;
// the SELayer product pointseg_modules.py:220.
#include "common.h"
#include "pool_strip.h"
#include <stdlib.h>

namespace {

__global__ __launch_bounds__(256) void maxpool_fwd_kernel(
    const float* __restrict__ x, const float* __restrict__ xs, float* __restrict__ y,
    uint8_t* __restrict__ idx, int N, int C, int H, int W, int OH, int OW, int K, int SH, int SW,
    int PH, int PW) {
  const int64_t total = (int64_t)N * C * OH * OW;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int ow = (int)(i % OW);
    int64_t t = i / OW;
    const int oh = (int)(t % OH);
    const int64_t pl = t / OH;
    const float* xp = x + (size_t)pl * H * W;
    const float s = xs ? xs[pl] : 1.f;
    float best = -INFINITY;
    int bi = 0;
    bool first = true;
    for (int ky = 0; ky < K; ++ky) {
      const int ih = oh * SH - PH + ky;
      if (ih < 0 || ih >= H) continue;
      for (int kx = 0; kx < K; ++kx) {
        const int iw = ow * SW - PW + kx;
        if (iw < 0 || iw >= W) continue;
        float v = xp[(size_t)ih * W + iw];
        if (xs) v *= s;
        if (first || v > best || v != v) { best = v; bi = ky * K + kx; first = false; }
      }
    }
    y[i] = best;
    if (idx) idx[i] = (uint8_t)bi;
  }
}

__global__ __launch_bounds__(256) void maxpool_bwd_kernel(
    const float* __restrict__ dy, const uint8_t* __restrict__ idx, const float* __restrict__ xs,
    float* __restrict__ dx, int N, int C, int H, int W, int OH, int OW, int K, int SH, int SW,
    int PH, int PW) {
  const int64_t total = (int64_t)N * C * H * W;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int iw = (int)(i % W);
    int64_t t = i / W;
    const int ih = (int)(t % H);
    const int64_t pl = t / H;
    const float* dyp = dy + (size_t)pl * OH * OW;
    const uint8_t* ip = idx + (size_t)pl * OH * OW;
    // windows covering (ih, iw): oh*SH - PH <= ih <= oh*SH - PH + K - 1
    int oh_lo = ih + PH - K + 1;
    oh_lo = oh_lo <= 0 ? 0 : (oh_lo + SH - 1) / SH;
    int oh_hi = (ih + PH) / SH;
    if (oh_hi > OH - 1) oh_hi = OH - 1;
    int ow_lo = iw + PW - K + 1;
    ow_lo = ow_lo <= 0 ? 0 : (ow_lo + SW - 1) / SW;
    int ow_hi = (iw + PW) / SW;
    if (ow_hi > OW - 1) ow_hi = OW - 1;
    float g = 0.f;
    for (int oh = oh_lo; oh <= oh_hi; ++oh) {
      const int ky = ih + PH - oh * SH;
      for (int ow = ow_lo; ow <= ow_hi; ++ow) {
        const int kx = iw + PW - ow * SW;
        if (ip[(size_t)oh * OW + ow] == (uint8_t)(ky * K + kx)) g += dyp[(size_t)oh * OW + ow];
      }
    }
    if (xs) g *= xs[pl];
    dx[i] = g;
  }
}

// ---- 3x3 / pad 1 / stride (SH,2) fast paths (every pool on the PointSeg and ResNet paths) ----
// forward: one thread = 2 adjacent outputs = input columns 4b-1 .. 4b+3: per window row one
// aligned float4 + one scalar instead of 6 scalar loads; same scan order as the generic kernel
// (ky major, kx minor, strict '>'), so values and argmax are identical.
template <int SH>
__global__ __launch_bounds__(256) void maxpool3_fwd_sw2(const float* __restrict__ x,
                                                        const float* __restrict__ xs,
                                                        float* __restrict__ y,
                                                        uint8_t* __restrict__ idx, int64_t planes,
                                                        int H, int W, int OH, int OW) {
  const int OW2 = OW >> 1;
  const int64_t total = planes * OH * OW2;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int b = (int)(i % OW2);
    int64_t t = i / OW2;
    const int oh = (int)(t % OH);
    const int64_t pl = t / OH;
    const float* xp = x + (size_t)pl * H * W;
    const float s = xs ? xs[pl] : 1.f;
    float best[2] = {-INFINITY, -INFINITY};
    int bi[2] = {0, 0};
    bool first[2] = {true, true};
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int ih = oh * SH - 1 + ky;
      if (ih < 0 || ih >= H) continue;
      const float* row = xp + (size_t)ih * W + 4 * b;
      const float4 v4 = *reinterpret_cast<const float4*>(row);
      float v[5];
      v[0] = b > 0 ? row[-1] : 0.f;
      v[1] = v4.x; v[2] = v4.y; v[3] = v4.z; v[4] = v4.w;
      if (xs) {
#pragma unroll
        for (int k = 0; k < 5; ++k) v[k] *= s;
      }
#pragma unroll
      for (int o = 0; o < 2; ++o) {
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int c = 2 * o + kx;          // column index into v (input col 4b-1+c)
          if (c == 0 && b == 0) continue;    // left padding
          const float val = v[c];
          if (first[o] || val > best[o] || val != val) { best[o] = val; bi[o] = ky * 3 + kx; first[o] = false; }
        }
      }
    }
    const size_t oo = ((size_t)pl * OH + oh) * OW + 2 * b;
    *reinterpret_cast<float2*>(y + oo) = make_float2(best[0], best[1]);
    if (idx) { idx[oo] = (uint8_t)bi[0]; idx[oo + 1] = (uint8_t)bi[1]; }
  }
}

// backward gather for 4 adjacent input columns 4b..4b+3 of row ih <- output columns 2b..2b+2
// ---- forward as a rolling vertical window (same idea as the strip backward below) ------------
// A thread owns 2 output columns x FR output rows.  Every input row it needs is read ONCE
// (float4 + left halo) and reduced to a (row maximum, kx) pair per output column; an output is
// the first maximum over its three row results.  Row-then-column "first strictly greater wins,
// NaN always wins" is the same total order as ATen's flat ky-major scan, so values and argmax
// codes are bit-identical to maxpool3_fwd_sw2.  10 row loads per 16 outputs instead of 48.
template <int SH>
__global__ __launch_bounds__(256) void maxpool3_fwd_strip(const float* __restrict__ x,
                                                          const float* __restrict__ xs,
                                                          float* __restrict__ y,
                                                          uint8_t* __restrict__ idx, int64_t planes,
                                                          int H, int W, int OH, int OW,
                                                          const float* __restrict__ aff = nullptr, int C = 1) {
  // aff (apply-on-load, [3][C] = mean, scale, shift): x is a convolution's raw output, the pool takes the maximum of
  // max(0, (x - mean[c]) * scale[c] + shift[c]) -- the BatchNorm + ReLU in front of the pool is never written
  constexpr int FR = SH == 1 ? 8 : 4;              // output rows per thread
  constexpr int NIN = (FR - 1) * SH + 3;           // input rows they touch
  const int OW2 = OW >> 1, strips = (OH + FR - 1) / FR;
  const int64_t total = planes * strips * OW2;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int b = (int)(i % OW2);
    int64_t t = i / OW2;
    const int oh0 = (int)(t % strips) * FR;
    const int64_t pl = t / strips;
    const float* xp = x + (size_t)pl * H * W;
    const float s = xs ? xs[pl] : 1.f;
    float amu = 0.f, asc = 1.f, ash = 0.f;
    if (aff) { const int c = (int)(pl % C); amu = aff[c]; asc = aff[C + c]; ash = aff[2 * C + c]; }
    float rb[NIN][2];
    int rk[NIN][2];
    bool rv[NIN];
    // all row loads first, unconditionally (row index clamped, halo address clamped): a load under
    // a branch is followed by s_waitcnt vmcnt(0), i.e. one HBM round trip per row
    float4 v4s[NIN];
    float hl[NIN];
#pragma unroll
    for (int j = 0; j < NIN; ++j) {
      const int ih = oh0 * SH - 1 + j;
      rv[j] = ih >= 0 && ih < H;
      const float* row = xp + (size_t)min(max(ih, 0), H - 1) * W + 4 * b;
      v4s[j] = *reinterpret_cast<const float4*>(row);
      hl[j] = row[b > 0 ? -1 : 0];
    }
#pragma unroll
    for (int j = 0; j < NIN; ++j) {
      rb[j][0] = rb[j][1] = 0.f; rk[j][0] = rk[j][1] = 0;
      if (!rv[j]) continue;
      const float4 v4 = v4s[j];
      float v[5];
      v[0] = b > 0 ? hl[j] : 0.f;
      v[1] = v4.x; v[2] = v4.y; v[3] = v4.z; v[4] = v4.w;
      if (aff) {
#pragma unroll
        for (int k = 0; k < 5; ++k) v[k] = fmaxf((v[k] - amu) * asc + ash, 0.f);
      }
      if (xs) {
#pragma unroll
        for (int k = 0; k < 5; ++k) v[k] *= s;
      }
#pragma unroll
      for (int o = 0; o < 2; ++o) {
        bool first = true;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int c = 2 * o + kx;
          if (c == 0 && b == 0) continue;          // left padding
          const float val = v[c];
          if (first || val > rb[j][o] || val != val) { rb[j][o] = val; rk[j][o] = kx; first = false; }
        }
      }
    }
#pragma unroll
    for (int r = 0; r < FR; ++r) {
      const int oh = oh0 + r;
      if (oh >= OH) continue;
      float best[2] = {-INFINITY, -INFINITY};
      int bi[2] = {0, 0};
#pragma unroll
      for (int o = 0; o < 2; ++o) {
        bool first = true;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          const int j = r * SH + ky;
          if (!rv[j]) continue;
          const float val = rb[j][o];
          if (first || val > best[o] || val != val) { best[o] = val; bi[o] = ky * 3 + rk[j][o]; first = false; }
        }
      }
      const size_t oo = ((size_t)pl * OH + oh) * OW + 2 * b;
      *reinterpret_cast<float2*>(y + oo) = make_float2(best[0], best[1]);
      if (idx) *reinterpret_cast<unsigned short*>(idx + oo) = (unsigned short)(bi[0] | (bi[1] << 8));
    }
  }
}

template <int SH>
__device__ __forceinline__ void pool3_bwd_gather(const float* __restrict__ dyp,
                                                 const uint8_t* __restrict__ ip, int ih, int b, int OH,
                                                 int OW, float (&g)[4]) {
  g[0] = g[1] = g[2] = g[3] = 0.f;
  int ohs[3], kys[3], nc;
  if (SH == 1) { nc = 3; ohs[0] = ih - 1; kys[0] = 2; ohs[1] = ih; kys[1] = 1; ohs[2] = ih + 1; kys[2] = 0; }
  else if ((ih & 1) == 0) { nc = 1; ohs[0] = ih >> 1; kys[0] = 1; ohs[1] = ohs[2] = -1; kys[1] = kys[2] = 0; }
  else { nc = 2; ohs[0] = ih >> 1; kys[0] = 2; ohs[1] = (ih >> 1) + 1; kys[1] = 0; ohs[2] = -1; kys[2] = 0; }
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    if (q >= nc) continue;
    const int oh = ohs[q];
    if (oh < 0 || oh >= OH) continue;
    const int base = kys[q] * 3;
    const size_t ro = (size_t)oh * OW + 2 * b;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      if (2 * b + c >= OW) continue;
      const int kx = (int)ip[ro + c] - base;
      const float v = dyp[ro + c];
      if (c == 0) { if (kx == 1) g[0] += v; else if (kx == 2) g[1] += v; }
      else if (c == 1) { if (kx == 0) g[1] += v; else if (kx == 1) g[2] += v; else if (kx == 2) g[3] += v; }
      else { if (kx == 0) g[3] += v; }
    }
  }
}

// dx = gather * xs[plane] + xadd[plane]   (xs / xadd nullable: SELayer scale, GAP-backward constant)
template <int SH>
__global__ __launch_bounds__(256) void maxpool3_bwd_sw2(const float* __restrict__ dy,
                                                        const uint8_t* __restrict__ idx,
                                                        const float* __restrict__ xs,
                                                        const float* __restrict__ xadd,
                                                        float* __restrict__ dx, int64_t planes, int H,
                                                        int W, int OH, int OW) {
  const int W4 = W >> 2;
  const int64_t total = planes * H * W4;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int b = (int)(i % W4);
    int64_t t = i / W4;
    const int ih = (int)(t % H);
    const int64_t pl = t / H;
    float g[4];
    pool3_bwd_gather<SH>(dy + (size_t)pl * OH * OW, idx + (size_t)pl * OH * OW, ih, b, OH, OW, g);
    if (xs) { const float s = xs[pl]; g[0] *= s; g[1] *= s; g[2] *= s; g[3] *= s; }
    if (xadd) { const float a = xadd[pl]; g[0] += a; g[1] += a; g[2] += a; g[3] += a; }
    *reinterpret_cast<float4*>(dx + ((size_t)pl * H + ih) * W + 4 * b) = make_float4(g[0], g[1], g[2], g[3]);
  }
}

// ds[plane] = sum over the plane of gather * x  (the SELayer scale gradient) without ever
// writing the pooled-gradient plane: one block per plane, fp64 block reduction
// ---- stride-(1,2) backward as a rolling vertical window -------------------------------------
// A thread owns 4 input columns x PR input rows.  Every output row it needs (PR+2 of them) is
// read ONCE (float2 + 1 halo value, 3 argmax bytes) and routed to up to three of the PR register
// rows -- the per-element gather above reads each dY element 3x (18 loads per float4 of dX; this
// form: 4 loads per output row, 40 per 8 float4).  Contributions arrive in the same order
// (output row ascending, then column), so results are bit-identical to pool3_bwd_gather.
template <int SH>
__global__ __launch_bounds__(256) void maxpool3_bwd_strip(const float* __restrict__ dy,
                                                             const uint8_t* __restrict__ idx,
                                                             const float* __restrict__ xs,
                                                             const float* __restrict__ xadd,
                                                             float* __restrict__ dx, int64_t planes,
                                                             int H, int W, int OH, int OW) {
  const int W4 = W >> 2, strips = (H + PR - 1) / PR;
  const int64_t total = planes * strips * W4;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int b = (int)(i % W4);
    int64_t t = i / W4;
    const int ih0 = (int)(t % strips) * PR;
    const int64_t pl = t / strips;
    float G[PR][4];
    pool3_strip<SH>(dy + (size_t)pl * OH * OW, idx + (size_t)pl * OH * OW, ih0, b, OH, OW, G);
    const float s = xs ? xs[pl] : 1.f, a = xadd ? xadd[pl] : 0.f;
#pragma unroll
    for (int r = 0; r < PR; ++r) {
      if (ih0 + r >= H) continue;
      float4 o = make_float4(G[r][0], G[r][1], G[r][2], G[r][3]);
      if (xs) { o.x *= s; o.y *= s; o.z *= s; o.w *= s; }
      if (xadd) { o.x += a; o.y += a; o.z += a; o.w += a; }
      st4<32>(dx + ((size_t)pl * H + ih0 + r) * W + 4 * b, o);
    }
  }
}

template <int SH>
__global__ __launch_bounds__(256) void maxpool3_bwd_dot_strip(const float* __restrict__ dy,
                                                                 const uint8_t* __restrict__ idx,
                                                                 const float* __restrict__ x,
                                                                 float* __restrict__ ds, int planes,
                                                                 int H, int W, int OH, int OW) {
  __shared__ double sm[16];
  const int W4 = W >> 2, strips = (H + PR - 1) / PR;
  for (int pl = blockIdx.x; pl < planes; pl += gridDim.x) {
    const float* xp = x + (size_t)pl * H * W;
    double acc = 0.0;
    for (int i = threadIdx.x; i < strips * W4; i += 256) {
      const int b = i % W4, ih0 = (i / W4) * PR;
      float G[PR][4];
      pool3_strip<SH>(dy + (size_t)pl * OH * OW, idx + (size_t)pl * OH * OW, ih0, b, OH, OW, G);
      float4 xvs[PR];
#pragma unroll
      for (int r = 0; r < PR; ++r)
        xvs[r] = *reinterpret_cast<const float4*>(xp + (size_t)min(ih0 + r, H - 1) * W + 4 * b);
#pragma unroll
      for (int r = 0; r < PR; ++r) {
        if (ih0 + r >= H) continue;
        const float4 xv = xvs[r];
        acc += (double)((G[r][0] * xv.x + G[r][1] * xv.y) + (G[r][2] * xv.z + G[r][3] * xv.w));
      }
    }
    const double rsum = block_sum_d(acc, sm);
    if (threadIdx.x == 0) ds[pl] = (float)rsum;
  }
}

template <int SH>
__global__ __launch_bounds__(256) void maxpool3_bwd_dot_sw2(const float* __restrict__ dy,
                                                            const uint8_t* __restrict__ idx,
                                                            const float* __restrict__ x,
                                                            float* __restrict__ ds, int planes, int H,
                                                            int W, int OH, int OW) {
  __shared__ double sm[16];
  const int W4 = W >> 2;
  for (int pl = blockIdx.x; pl < planes; pl += gridDim.x) {
    const float* dyp = dy + (size_t)pl * OH * OW;
    const uint8_t* ip = idx + (size_t)pl * OH * OW;
    const float* xp = x + (size_t)pl * H * W;
    double acc = 0.0;
    for (int i = threadIdx.x; i < H * W4; i += 256) {
      const int b = i % W4, ih = i / W4;
      float g[4];
      pool3_bwd_gather<SH>(dyp, ip, ih, b, OH, OW, g);
      const float4 xv = *reinterpret_cast<const float4*>(xp + (size_t)ih * W + 4 * b);
      acc += (double)((g[0] * xv.x + g[1] * xv.y) + (g[2] * xv.z + g[3] * xv.w));
    }
    const double r = block_sum_d(acc, sm);
    if (threadIdx.x == 0) ds[pl] = (float)r;
  }
}

// one block per (n, c) plane
__global__ __launch_bounds__(256) void gap_fwd_kernel(const float* __restrict__ x, int ctot,
                                                      int coff, float* __restrict__ out, int N,
                                                      int C, int HW) {
  __shared__ double sm[16];
  for (int pl = blockIdx.x; pl < N * C; pl += gridDim.x) {
    const int n = pl / C, c = pl - n * C;
    const float* xp = x + ((size_t)n * ctot + coff + c) * HW;
    double s = 0.0;
    if ((HW & 3) == 0) {
      for (int i = threadIdx.x; i < (HW >> 2); i += 256) {
        const float4 v = *reinterpret_cast<const float4*>(xp + ((size_t)i << 2));
        s += (double)((v.x + v.y) + (v.z + v.w));
      }
    } else {
      for (int i = threadIdx.x; i < HW; i += 256) s += xp[i];
    }
    const double r = block_sum_d(s, sm);
    if (threadIdx.x == 0) out[pl] = (float)(r / (double)HW);
  }
}

// out[plane] = (sum_hw a * b) / div[plane]: the SELayer scale gradient behind a max-pool from POOLED tensors.  With
// y = maxpool(x * s), s > 0 per plane: d loss / d s = sum_o dy[o] * x[argmax(o)] = sum_o dy[o] * y[o] / s -- the pooled
// gradient and the pooled output instead of the full-resolution x and the arg-max map (dlio_maxpool2d_bwd_dot reads 1.6-2.6x
// the bytes).  s == 0 (a sigmoid that underflowed): y is all zero, the result is 0.  UN float4 pairs in flight per thread.
__global__ __launch_bounds__(256) void plane_dot_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                        const float* __restrict__ div, float* __restrict__ out, int planes,
                                                        int HW) {
  __shared__ double sm[16];
  for (int pl = blockIdx.x; pl < planes; pl += gridDim.x) {
    const float* ap = a + (size_t)pl * HW;
    const float* bp = b + (size_t)pl * HW;
    double s = 0.0;
    if ((HW & 3) == 0) {
      const int n4 = HW >> 2;
      int i = threadIdx.x;
      for (; i + 256 < n4; i += 512) {
        const float4 u0 = *reinterpret_cast<const float4*>(ap + ((size_t)i << 2)), v0 = *reinterpret_cast<const float4*>(bp + ((size_t)i << 2));
        const float4 u1 = *reinterpret_cast<const float4*>(ap + ((size_t)(i + 256) << 2)), v1 = *reinterpret_cast<const float4*>(bp + ((size_t)(i + 256) << 2));
        s += (double)((u0.x * v0.x + u0.y * v0.y) + (u0.z * v0.z + u0.w * v0.w));
        s += (double)((u1.x * v1.x + u1.y * v1.y) + (u1.z * v1.z + u1.w * v1.w));
      }
      for (; i < n4; i += 256) {
        const float4 u = *reinterpret_cast<const float4*>(ap + ((size_t)i << 2)), v = *reinterpret_cast<const float4*>(bp + ((size_t)i << 2));
        s += (double)((u.x * v.x + u.y * v.y) + (u.z * v.z + u.w * v.w));
      }
    } else {
      for (int i = threadIdx.x; i < HW; i += 256) s += (double)ap[i] * bp[i];
    }
    const double r = block_sum_d(s, sm);
    if (threadIdx.x == 0) {
      const float dv = div ? div[pl] : 1.f;
      out[pl] = dv != 0.f ? (float)(r / (double)dv) : 0.f;
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void gap_bwd_kernel(const float* __restrict__ dout,
                                                      float* __restrict__ dx, int64_t planes,
                                                      int HW, int accumulate) {
  const int64_t total = planes * HW;
  const float inv = 1.0f / (float)HW;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const float g = dout[i / HW] * inv;
    dx[i] = accumulate ? dx[i] + g : g;
  }
}

__global__ __launch_bounds__(256) void chan_scale_fwd_kernel(const float* __restrict__ x,
                                                             const float* __restrict__ s,
                                                             float* __restrict__ y,
                                                             int64_t planes, int HW) {
  const int64_t total = planes * HW;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x)
    y[i] = x[i] * s[i / HW];
}

// the same, four float4 in flight per thread (the SELayer scale over the pooled maximum of a block that pooled its own
// output: csrc/bn_stream.hip)
__global__ __launch_bounds__(256) void chan_scale_fwd_v4_kernel(const float4* __restrict__ x, const float* __restrict__ s,
                                                                float4* __restrict__ y, int64_t total4, int HW4) {
  constexpr int UN = 4;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total4; i += UN * stride) {
    float4 v[UN];
    float sc[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int64_t k = i + u * stride < total4 ? i + u * stride : total4 - 1;
      v[u] = x[k];
      sc[u] = s[k / HW4];
    }
#pragma unroll
    for (int u = 0; u < UN; ++u)
      if (i + u * stride < total4) y[i + u * stride] = make_float4(v[u].x * sc[u], v[u].y * sc[u], v[u].z * sc[u], v[u].w * sc[u]);
  }
}

// one block per plane: dx = dy*s, ds = sum dy*x
__global__ __launch_bounds__(256) void chan_scale_bwd_kernel(const float* __restrict__ dy,
                                                             const float* __restrict__ x,
                                                             const float* __restrict__ s,
                                                             float* __restrict__ dx,
                                                             float* __restrict__ ds, int planes,
                                                             int HW) {
  __shared__ double sm[16];
  for (int pl = blockIdx.x; pl < planes; pl += gridDim.x) {
    const float sv = s[pl];
    const float* gp = dy + (size_t)pl * HW;
    const float* xp = x + (size_t)pl * HW;
    float* op = dx + (size_t)pl * HW;
    double acc = 0.0;
    for (int i = threadIdx.x; i < HW; i += 256) {
      const float g = gp[i];
      acc += (double)(g * xp[i]);
      op[i] = g * sv;
    }
    const double r = block_sum_d(acc, sm);
    if (threadIdx.x == 0) ds[pl] = (float)r;
  }
}

}  // namespace

extern "C" int dlio_maxpool2d_fwd_aff(const float* x, const float* aff, float* y, uint8_t* idx, int N, int C, int H,
                                      int W, int OH, int OW, int K, int SH, int SW, int PH, int PW,
                                      dlio_stream_t stream) {
  if (!x || !aff || !y || N <= 0 || C <= 0) return DLIO_EINVAL;
  if (!(K == 3 && SW == 2 && PH == 1 && PW == 1 && (SH == 1 || SH == 2) && (W & 3) == 0 && OW * 2 == W &&
        OH == (H + 2 - 3) / SH + 1))
    return DLIO_EUNSUP;
  const int64_t work = (int64_t)N * C * OH * OW / 2;
  DlioProfScope prof(10, as_stream(stream), 0.0, (double)N * C * (4.0 * H * W + (idx ? 5.0 : 4.0) * OH * OW));
  if (SH == 1)
    hipLaunchKernelGGL(maxpool3_fwd_strip<1>, dim3(ew_grid(cdiv64(work, 8), 256)), dim3(256), 0, as_stream(stream), x,
                       (const float*)nullptr, y, idx, (int64_t)N * C, H, W, OH, OW, aff, C);
  else
    hipLaunchKernelGGL(maxpool3_fwd_strip<2>, dim3(ew_grid(cdiv64(work, 4), 256)), dim3(256), 0, as_stream(stream), x,
                       (const float*)nullptr, y, idx, (int64_t)N * C, H, W, OH, OW, aff, C);
  return dlio_check_launch();
}

extern "C" int dlio_maxpool2d_fwd(const float* x, const float* x_scale, float* y, uint8_t* idx,
                                  int N, int C, int H, int W, int OH, int OW, int K, int SH,
                                  int SW, int PH, int PW, dlio_stream_t stream) {
  if (!x || !y || N <= 0 || C <= 0 || K <= 0 || K > 15) return DLIO_EINVAL;
  const int64_t total = (int64_t)N * C * OH * OW;
  DlioProfScope prof(10, as_stream(stream), 0.0, (double)N * C * (4.0 * H * W + (idx ? 5.0 : 4.0) * OH * OW));
  if (K == 3 && SW == 2 && PH == 1 && PW == 1 && (SH == 1 || SH == 2) && (W & 3) == 0 && OW * 2 == W &&
      OH == (H + 2 - 3) / SH + 1) {
    const int64_t work = total / 2;
    static const int strip = 1;   // tuning knob
    if (strip && SH == 1)
      hipLaunchKernelGGL(maxpool3_fwd_strip<1>, dim3(ew_grid(cdiv64(work, 8), 256)), dim3(256), 0,
                         as_stream(stream), x, x_scale, y, idx, (int64_t)N * C, H, W, OH, OW, (const float*)nullptr, 1);
    else if (strip)
      hipLaunchKernelGGL(maxpool3_fwd_strip<2>, dim3(ew_grid(cdiv64(work, 4), 256)), dim3(256), 0,
                         as_stream(stream), x, x_scale, y, idx, (int64_t)N * C, H, W, OH, OW, (const float*)nullptr, 1);
    else if (SH == 1)
      hipLaunchKernelGGL(maxpool3_fwd_sw2<1>, dim3(ew_grid(work, 256)), dim3(256), 0, as_stream(stream),
                         x, x_scale, y, idx, (int64_t)N * C, H, W, OH, OW);
    else
      hipLaunchKernelGGL(maxpool3_fwd_sw2<2>, dim3(ew_grid(work, 256)), dim3(256), 0, as_stream(stream),
                         x, x_scale, y, idx, (int64_t)N * C, H, W, OH, OW);
    return dlio_check_launch();
  }
  hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(ew_grid(total, 256)), dim3(256), 0,
                     as_stream(stream), x, x_scale, y, idx, N, C, H, W, OH, OW, K, SH, SW, PH, PW);
  return dlio_check_launch();
}

static bool pool_fast(int H, int W, int OH, int OW, int K, int SH, int SW, int PH, int PW) {
  return K == 3 && SW == 2 && PH == 1 && PW == 1 && (SH == 1 || SH == 2) && (W & 3) == 0 && OW * 2 == W &&
         OH == (H + 2 - 3) / SH + 1;
}

extern "C" int dlio_maxpool2d_bwd_dot(const float* dy, const uint8_t* idx, const float* x, float* ds,
                                      int N, int C, int H, int W, int OH, int OW, int K, int SH,
                                      int SW, int PH, int PW, dlio_stream_t stream) {
  if (!dy || !idx || !x || !ds || N <= 0 || C <= 0) return DLIO_EINVAL;
  if (!pool_fast(H, W, OH, OW, K, SH, SW, PH, PW)) return DLIO_EUNSUP;
  DlioProfScope prof(10, as_stream(stream), 0.0, (double)N * C * (4.0 * H * W + 5.0 * OH * OW));
  int grid = N * C;
  if (grid > 65535) grid = 65535;
  static const int strip = 1;   // tuning knob
  if (SH == 1 && strip)
    hipLaunchKernelGGL(maxpool3_bwd_dot_strip<1>, dim3(grid), dim3(256), 0, as_stream(stream), dy, idx,
                       x, ds, N * C, H, W, OH, OW);
  else if (strip && (H & 1) == 0)
    hipLaunchKernelGGL(maxpool3_bwd_dot_strip<2>, dim3(grid), dim3(256), 0, as_stream(stream), dy, idx,
                       x, ds, N * C, H, W, OH, OW);
  else if (SH == 1)
    hipLaunchKernelGGL(maxpool3_bwd_dot_sw2<1>, dim3(grid), dim3(256), 0, as_stream(stream), dy, idx, x,
                       ds, N * C, H, W, OH, OW);
  else
    hipLaunchKernelGGL(maxpool3_bwd_dot_sw2<2>, dim3(grid), dim3(256), 0, as_stream(stream), dy, idx, x,
                       ds, N * C, H, W, OH, OW);
  return dlio_check_launch();
}

extern "C" int dlio_maxpool2d_bwd(const float* dy, const uint8_t* idx, const float* x_scale,
                                  const float* x_add, float* dx, int N, int C, int H, int W, int OH,
                                  int OW, int K, int SH, int SW, int PH, int PW,
                                  dlio_stream_t stream) {
  if (!dy || !idx || !dx || N <= 0 || C <= 0 || K <= 0) return DLIO_EINVAL;
  const int64_t total = (int64_t)N * C * H * W;
  DlioProfScope prof(10, as_stream(stream), 0.0, (double)N * C * (4.0 * H * W + 5.0 * OH * OW));
  if (K == 3 && SW == 2 && PH == 1 && PW == 1 && (SH == 1 || SH == 2) && (W & 3) == 0 && OW * 2 == W &&
      OH == (H + 2 - 3) / SH + 1) {
    const int64_t work = total / 4;
    static const int strip = 1;   // tuning knob
    if (SH == 1 && strip)
      hipLaunchKernelGGL(maxpool3_bwd_strip<1>, dim3(ew_grid(cdiv64(work, PR), 256)), dim3(256), 0,
                         as_stream(stream), dy, idx, x_scale, x_add, dx, (int64_t)N * C, H, W, OH, OW);
    else if (strip && (H & 1) == 0)
      hipLaunchKernelGGL(maxpool3_bwd_strip<2>, dim3(ew_grid(cdiv64(work, PR), 256)), dim3(256), 0,
                         as_stream(stream), dy, idx, x_scale, x_add, dx, (int64_t)N * C, H, W, OH, OW);
    else if (SH == 1)
      hipLaunchKernelGGL(maxpool3_bwd_sw2<1>, dim3(ew_grid(work, 256)), dim3(256), 0, as_stream(stream),
                         dy, idx, x_scale, x_add, dx, (int64_t)N * C, H, W, OH, OW);
    else
      hipLaunchKernelGGL(maxpool3_bwd_sw2<2>, dim3(ew_grid(work, 256)), dim3(256), 0, as_stream(stream),
                         dy, idx, x_scale, x_add, dx, (int64_t)N * C, H, W, OH, OW);
    return dlio_check_launch();
  }
  if (x_add) return DLIO_EUNSUP;   // the fused add exists on the 3x3 / stride-(.,2) fast path only
  hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(ew_grid(total, 256)), dim3(256), 0,
                     as_stream(stream), dy, idx, x_scale, dx, N, C, H, W, OH, OW, K, SH, SW, PH, PW);
  return dlio_check_launch();
}

extern "C" int dlio_gap_fwd(const float* x, int ctot, int coff, float* out, int N, int C, int HW,
                            dlio_stream_t stream) {
  if (!x || !out || N <= 0 || C <= 0 || HW <= 0) return DLIO_EINVAL;
  int grid = N * C;
  if (grid > 65535) grid = 65535;
  hipLaunchKernelGGL(gap_fwd_kernel, dim3(grid), dim3(256), 0, as_stream(stream), x, ctot, coff,
                     out, N, C, HW);
  return dlio_check_launch();
}

extern "C" int dlio_plane_dot(const float* a, const float* b, const float* div, float* out, int planes, int HW,
                              dlio_stream_t stream) {
  if (!a || !b || !out || planes <= 0 || HW <= 0) return DLIO_EINVAL;
  if ((HW & 3) == 0 && ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) & 15)) return DLIO_EUNSUP;
  hipStream_t s = as_stream(stream);
  DlioProfScope prof(10, s, 0.0, 8.0 * planes * (double)HW);
  hipLaunchKernelGGL(plane_dot_kernel, dim3(planes > 65535 ? 65535 : planes), dim3(256), 0, s, a, b, div, out, planes, HW);
  return dlio_check_launch();
}

extern "C" int dlio_gap_bwd(const float* dout, float* dx, int N, int C, int HW, int accumulate,
                            dlio_stream_t stream) {
  if (!dout || !dx || N <= 0 || C <= 0 || HW <= 0) return DLIO_EINVAL;
  const int64_t total = (int64_t)N * C * HW;
  hipLaunchKernelGGL(gap_bwd_kernel, dim3(ew_grid(total, 256)), dim3(256), 0, as_stream(stream),
                     dout, dx, (int64_t)N * C, HW, accumulate);
  return dlio_check_launch();
}

extern "C" int dlio_chan_scale_fwd(const float* x, const float* s, float* y, int N, int C, int HW,
                                   dlio_stream_t stream) {
  if (!x || !s || !y || N <= 0 || C <= 0 || HW <= 0) return DLIO_EINVAL;
  const int64_t total = (int64_t)N * C * HW;
  DlioProfScope prof(10, as_stream(stream), 0.0, 8.0 * (double)total);
  if ((HW & 3) == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0)
    hipLaunchKernelGGL(chan_scale_fwd_v4_kernel, dim3(ew_grid(total / 16, 256)), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const float4*>(x), s, reinterpret_cast<float4*>(y), total / 4, HW / 4);
  else
    hipLaunchKernelGGL(chan_scale_fwd_kernel, dim3(ew_grid(total, 256)), dim3(256), 0,
                       as_stream(stream), x, s, y, (int64_t)N * C, HW);
  return dlio_check_launch();
}

extern "C" int dlio_chan_scale_bwd(const float* dy, const float* x, const float* s, float* dx,
                                   float* ds, int N, int C, int HW, dlio_stream_t stream) {
  if (!dy || !x || !s || !dx || !ds || N <= 0 || C <= 0 || HW <= 0) return DLIO_EINVAL;
  int grid = N * C;
  if (grid > 65535) grid = 65535;
  hipLaunchKernelGGL(chan_scale_bwd_kernel, dim3(grid), dim3(256), 0, as_stream(stream), dy, x, s,
                     dx, ds, N * C, HW);
  return dlio_check_launch();
}
