// The two bias-free fully connected layers of an SELayer (pointseg_modules.py:203-221: Linear(C, C/r) -> ReLU ->
// Linear(C/r, C) -> Sigmoid on the [N, C] plane averages) as ONE launch forward and TWO backward (data path, weight
// gradients) instead of 2 and 7 generic dense launches: N <= 64 rows against 2 x 512 KB of weights at most, sitting on the
// serial chain between a Fire block's BatchNorm and the max-pool that applies the scale.
//
// Forward / backward data path: one workgroup per image (the weights are L2 hits after the first image), the image's
// vectors in LDS.  A row of a weight matrix is read by a wave (16-byte loads, wave reduction) where the product runs
// along the row (h = W1 g, s = W2 h) and by consecutive threads where it runs down the columns (dh = W2^T dz2,
// dg = W1^T dz1: thread = column, the rows cut into parts that are added in a fixed order through LDS).
#include "common.h"

namespace {

constexpr int SE_T = 1024;          // threads per workgroup
constexpr int SE_MAXC = 1024;       // channels
constexpr int SE_MAXR = 512;        // reduced width

// y[j] = act(sum_k w[j][k] x[k]) for j < J, x in LDS; K % 4 == 0
template <int ACT>      // 0 relu, 1 sigmoid
__device__ __forceinline__ void rows_dot(const float* __restrict__ w, const float* xs, float* ys, float* yg, int J, int K) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = SE_T / 64;
  for (int j0 = wave * 4; j0 < J; j0 += nw * 4) {          // four rows at a time: four independent load streams
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k = lane * 4; k < K; k += 256) {
      const float4 xv = *reinterpret_cast<const float4*>(xs + k);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int j = min(j0 + u, J - 1);
        const float4 wv = *reinterpret_cast<const float4*>(w + (size_t)j * K + k);
        acc[u] += wv.x * xv.x + wv.y * xv.y + wv.z * xv.z + wv.w * xv.w;
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) acc[u] = wave_sum(acc[u]);
    if (lane == 0) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (j0 + u < J) {
          const float v = ACT == 0 ? fmaxf(acc[u], 0.f) : sigmoidf_(acc[u]);
          ys[j0 + u] = v;
          yg[j0 + u] = v;
        }
    }
  }
}

__global__ __launch_bounds__(SE_T) void se_fc_fwd_kernel(const float* __restrict__ g, const float* __restrict__ w1,
                                                         const float* __restrict__ w2, float* __restrict__ h,
                                                         float* __restrict__ s, int C, int R) {
  __shared__ __attribute__((aligned(16))) float gs[SE_MAXC];
  __shared__ __attribute__((aligned(16))) float hs[SE_MAXR];
  __shared__ float ss[SE_MAXC];
  const int n = blockIdx.x;
  for (int c = threadIdx.x; c < C; c += SE_T) gs[c] = g[(size_t)n * C + c];
  __syncthreads();
  rows_dot<0>(w1, gs, hs, h + (size_t)n * R, R, C);
  __syncthreads();
  rows_dot<1>(w2, hs, ss, s + (size_t)n * C, C, R);
}

// sum_i v[i] w[i * ldw + col] (thread = column, the I rows cut into `parts` parts), v in LDS; the partial sums go
// through `red` ([parts][Jp], slot = the thread's own (part, jj)) and are added in part order; result in part 0
__device__ __forceinline__ float cols_dot(const float* __restrict__ w, const float* vs, float* red, int I, int ldw, int col,
                                          bool valid, int jj, int part, int parts, int Jp) {
  float acc = 0.f;
  if (valid) {
    const int per = (I + parts - 1) / parts;
    const int i0 = part * per, i1 = min(I, i0 + per);
    int i = i0;
    for (; i + 8 <= i1; i += 8) {
      float t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = w[(size_t)(i + u) * ldw + col];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += vs[i + u] * t[u];
    }
    for (; i < i1; ++i) acc += vs[i] * w[(size_t)i * ldw + col];
  }
  red[part * Jp + jj] = acc;
  __syncthreads();
  float r = 0.f;
  if (part == 0 && valid)
    for (int p = 0; p < parts; ++p) r += red[p * Jp + jj];
  __syncthreads();
  return r;
}

// per image: dz2 = ds * s (1 - s); dh = W2^T dz2; dz1 = dh [h > 0]; dg = W1^T dz1 * dg_scale
__global__ __launch_bounds__(SE_T) void se_fc_bwd_data_kernel(const float* __restrict__ ds, const float* __restrict__ s,
                                                              const float* __restrict__ h, const float* __restrict__ w1,
                                                              const float* __restrict__ w2, float* __restrict__ dz2,
                                                              float* __restrict__ dz1, float* __restrict__ dg,
                                                              float dg_scale, int C, int R) {
  __shared__ float z2[SE_MAXC];
  __shared__ float z1[SE_MAXR];
  __shared__ float red[SE_T];
  const int n = blockIdx.x;
  for (int c = threadIdx.x; c < C; c += SE_T) {
    const float sv = s[(size_t)n * C + c];
    const float v = ds[(size_t)n * C + c] * sv * (1.f - sv);
    z2[c] = v;
    dz2[(size_t)n * C + c] = v;
  }
  __syncthreads();
  {   // dh[r] = sum_c dz2[c] w2[c][r]
    int Jp = 64;
    while (Jp < R) Jp <<= 1;                 // columns padded to a power of two <= SE_T
    const int parts = SE_T / Jp, j = threadIdx.x % Jp, part = threadIdx.x / Jp;
    const float v = cols_dot(w2, z2, red, C, R, j, j < R, j, part, parts, Jp);
    if (part == 0 && j < R) {
      const float o = h[(size_t)n * R + j] > 0.f ? v : 0.f;
      z1[j] = o;
      dz1[(size_t)n * R + j] = o;
    }
  }
  __syncthreads();
  for (int c0 = 0; c0 < C; c0 += SE_T) {     // dg[c] = sum_r dz1[r] w1[r][c]
    const int Jc = min(SE_T, C - c0);
    int Jp = 64;
    while (Jp < Jc) Jp <<= 1;
    const int parts = SE_T / Jp, j = threadIdx.x % Jp, part = threadIdx.x / Jp;
    const float v = cols_dot(w1, z1, red, R, C, c0 + j, j < Jc, j, part, parts, Jp);
    if (part == 0 && j < Jc) dg[(size_t)n * C + c0 + j] = v * dg_scale;
  }
}

// dw[j][k] (+)= sum_n a[n][j] b[n][k]  (N <= 64 rows: an outer-product sum per element)
__global__ __launch_bounds__(256) void se_fc_bwd_weight_kernel(const float* __restrict__ a1, const float* __restrict__ b1,
                                                               float* __restrict__ dw1, int J1, int K1,
                                                               const float* __restrict__ a2, const float* __restrict__ b2,
                                                               float* __restrict__ dw2, int J2, int K2, int N, int accumulate) {
  const int64_t n1 = (int64_t)J1 * K1, total = n1 + (int64_t)J2 * K2;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const bool first = i < n1;
    const int64_t e = first ? i : i - n1;
    const int K = first ? K1 : K2, J = first ? J1 : J2;
    const int j = (int)(e / K), k = (int)(e - (int64_t)j * K);
    const float* a = first ? a1 : a2;
    const float* b = first ? b1 : b2;
    float acc = 0.f;
    for (int n = 0; n < N; ++n) acc += a[(size_t)n * J + j] * b[(size_t)n * K + k];
    float* o = (first ? dw1 : dw2) + e;
    *o = accumulate ? *o + acc : acc;
  }
}

}  // namespace

extern "C" int dlio_se_fc_ok(int N, int C, int R) {
  return N >= 1 && N <= 4096 && C >= 4 && C <= SE_MAXC && R >= 4 && R <= SE_MAXR && (C & 3) == 0 && (R & 3) == 0;
}

extern "C" int dlio_se_fc_fwd(const float* g, const float* w1, const float* w2, float* h, float* s, int N, int C, int R,
                              dlio_stream_t stream) {
  if (!g || !w1 || !w2 || !h || !s || N <= 0 || C <= 0 || R <= 0) return DLIO_EINVAL;
  if (!dlio_se_fc_ok(N, C, R) || ((reinterpret_cast<uintptr_t>(w1) | reinterpret_cast<uintptr_t>(w2)) & 15)) return DLIO_EUNSUP;
  hipLaunchKernelGGL(se_fc_fwd_kernel, dim3((unsigned)N), dim3(SE_T), 0, as_stream(stream), g, w1, w2, h, s, C, R);
  return dlio_check_launch();
}

extern "C" int dlio_se_fc_bwd(const float* ds, const float* s, const float* h, const float* g, const float* w1,
                              const float* w2, float* dz2, float* dz1, float* dg, float dg_scale, float* dw1, float* dw2,
                              int accumulate, int N, int C, int R, dlio_stream_t stream) {
  if (!ds || !s || !h || !g || !w1 || !w2 || !dz2 || !dz1 || !dg || !dw1 || !dw2 || N <= 0 || C <= 0 || R <= 0)
    return DLIO_EINVAL;
  if (!dlio_se_fc_ok(N, C, R)) return DLIO_EUNSUP;
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(se_fc_bwd_data_kernel, dim3((unsigned)N), dim3(SE_T), 0, st, ds, s, h, w1, w2, dz2, dz1, dg, dg_scale, C, R);
  int rc = dlio_check_launch();
  if (rc) return rc;
  // dw1 [R][C] = dz1^T g, dw2 [C][R] = dz2^T h
  const int64_t total = 2 * (int64_t)R * C;
  hipLaunchKernelGGL(se_fc_bwd_weight_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, dz1, g, dw1, R, C, dz2, h,
                     dw2, C, R, N, accumulate);
  return dlio_check_launch();
}
