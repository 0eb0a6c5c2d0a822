// Train-mode BatchNorm2d (+ ReLU, + residual) of SMALL feature maps in one launch, forward and backward
// (pointseg_modules.py:98-106 in fire_blk4 / fire_blk5: 32 x 64 and 16 x 32 maps).  bn.hip's two-launch form
// (statistics partials, then an apply kernel that finalises them) reads the tensor twice and costs two dependent
// launches of 5-15 us; these blocks are the serial valley of the step (DESIGN: 14 % of the time for 5 % of the work),
// where launch count is what matters.  Here ONE workgroup of 16 waves owns a channel: wave w holds image n = w of the
// channel in registers (V float4 per lane: N <= 16, H * W = 256 V, V in {1, 2, 4, 8}), the statistics are a wave
// reduction + 16 partials through LDS, the apply runs on the registers.  Each element is read ONCE.
// A launch may cover two layers' channels at once (the expand1x1 / expand3x3 halves of a Fire block's concat buffer:
// channels [0, C1) take the first parameter set, [C1, C) the second), so a Fire block's two expand BatchNorms are one launch.
// Accumulation: per float4 in fp32, across float4 / lanes / waves in fp64 (as bn.hip), fixed order.
#include "common.h"

namespace {

struct BnSet {            // one layer's parameters (device pointers, nullable where noted)
  const float* gamma;     // nullable
  const float* beta;      // nullable
  float* running_mean;    // nullable
  float* running_var;     // nullable
  float* dgamma;          // backward, nullable
  float* dbeta;           // backward, nullable
};

__device__ __forceinline__ double block16_sum(double v, double* sm, int wave, int lane) {
  v = wave_sum_d(v);
  __syncthreads();
  if (lane == 0) sm[wave] = v;
  __syncthreads();
  double r = 0.0;
#pragma unroll
  for (int i = 0; i < 16; ++i) r += sm[i];
  return r;                                         // every thread has the total
}

template <int V>
__global__ __launch_bounds__(1024) void bn_small_fwd_kernel(
    const float* __restrict__ x, int x_ctot, int x_coff, int N, int C1, BnSet s1, BnSet s2, float eps, float momentum,
    float* __restrict__ mean_o, float* __restrict__ invstd_o, float* __restrict__ scale_o, float* __restrict__ shift_o,
    const float* residual, int r_ctot, int r_coff, const float* __restrict__ r_mean, const float* __restrict__ r_scale,
    const float* __restrict__ r_shift, float* y, int y_ctot, int y_coff, float* __restrict__ gap_out, int gap_ctot,
    int gap_coff, int post_relu) {
  constexpr int HW = 256 * V;
  __shared__ double sm[2][16];
  const int c = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const BnSet& ps = c < C1 ? s1 : s2;
  const int cl = c < C1 ? c : c - C1;
  const bool have = wave < N;
  float4 v[V];
  const float* xp = x + ((size_t)(have ? wave : 0) * x_ctot + x_coff + c) * HW;
  double a = 0.0, b = 0.0;
#pragma unroll
  for (int j = 0; j < V; ++j) {
    v[j] = have ? *reinterpret_cast<const float4*>(xp + 4 * (lane + 64 * j)) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float f0 = (v[j].x + v[j].y) + (v[j].z + v[j].w);
    const float f1 = (v[j].x * v[j].x + v[j].y * v[j].y) + (v[j].z * v[j].z + v[j].w * v[j].w);
    a += f0; b += f1;
  }
  a = block16_sum(a, sm[0], wave, lane);
  b = block16_sum(b, sm[1], wave, lane);
  const double count = (double)N * HW;
  const double m = a / count;
  double var = b / count - m * m;
  if (var < 0.0) var = 0.0;
  const float is = (float)(1.0 / sqrt(var + (double)eps));
  const float mu = (float)m, sc = (ps.gamma ? ps.gamma[cl] : 1.f) * is, be = ps.beta ? ps.beta[cl] : 0.f;
  if (threadIdx.x == 0) {
    mean_o[c] = mu; invstd_o[c] = is; scale_o[c] = sc;
    if (shift_o) shift_o[c] = be;
    if (ps.running_mean) ps.running_mean[cl] = (1.f - momentum) * ps.running_mean[cl] + momentum * mu;
    if (ps.running_var) {
      const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
      ps.running_var[cl] = (1.f - momentum) * ps.running_var[cl] + momentum * (float)unb;
    }
  }
  if (!y || !have) return;                          // statistics only (apply-on-load consumers)
  const float* rp = residual ? residual + ((size_t)wave * r_ctot + r_coff + c) * HW : nullptr;
  const bool raff = rp && r_scale;
  const float rmu = raff ? r_mean[r_coff + c] : 0.f, rsc = raff ? r_scale[r_coff + c] : 1.f, rsh = raff ? r_shift[r_coff + c] : 0.f;
  float* yp = y + ((size_t)wave * y_ctot + y_coff + c) * HW;
  double gs = 0.0;
#pragma unroll
  for (int j = 0; j < V; ++j) {
    float e[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
    float4 rv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (rp) rv = *reinterpret_cast<const float4*>(rp + 4 * (lane + 64 * j));
    float re[4] = {rv.x, rv.y, rv.z, rv.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float o = (e[k] - mu) * sc + be;
      if (post_relu) o = fmaxf(o, 0.f);
      if (raff) re[k] = fmaxf((re[k] - rmu) * rsc + rsh, 0.f);
      e[k] = o + re[k];
    }
    *reinterpret_cast<float4*>(yp + 4 * (lane + 64 * j)) = make_float4(e[0], e[1], e[2], e[3]);
    gs += (double)((e[0] + e[1]) + (e[2] + e[3]));
  }
  if (gap_out) {                                     // plane average of the OUTPUT (the SELayer behind the block): a wave = a plane
    gs = wave_sum_d(gs);
    if (lane == 0) gap_out[(size_t)wave * gap_ctot + gap_coff + c] = (float)(gs / (double)HW);
  }
}

// dx = scale * (g - mean(g) - xhat * mean(g * xhat)), g = dy where the activated output was > 0 (post_relu);
// channels [0, C1) -> dx1 [N][C1][HW], [C1, C) -> dx2 [N][C - C1][HW]; dgamma / dbeta per set (accumulated when asked)
template <int V>
__global__ __launch_bounds__(1024) void bn_small_bwd_kernel(
    const float* __restrict__ dy, int dy_ctot, int dy_coff, const float* __restrict__ x, int x_ctot, int x_coff, int N, int C,
    int C1, BnSet s1, BnSet s2, const float* __restrict__ mean, const float* __restrict__ invstd,
    const float* __restrict__ scale, float* __restrict__ dx1, float* __restrict__ dx2, int accumulate, int post_relu) {
  constexpr int HW = 256 * V;
  __shared__ double sm[2][16];
  const int c = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const BnSet& ps = c < C1 ? s1 : s2;
  const int cl = c < C1 ? c : c - C1;
  const bool have = wave < N;
  const float mu = mean[c], is = invstd[c], sc = scale[c], be = ps.beta ? ps.beta[cl] : 0.f;
  const float* gp = dy + ((size_t)(have ? wave : 0) * dy_ctot + dy_coff + c) * HW;
  const float* xp = x + ((size_t)(have ? wave : 0) * x_ctot + x_coff + c) * HW;
  float4 g[V], xh[V];
  double sg = 0.0, sgx = 0.0;
#pragma unroll
  for (int j = 0; j < V; ++j) {
    g[j] = have ? *reinterpret_cast<const float4*>(gp + 4 * (lane + 64 * j)) : make_float4(0.f, 0.f, 0.f, 0.f);
    xh[j] = have ? *reinterpret_cast<const float4*>(xp + 4 * (lane + 64 * j)) : make_float4(mu, mu, mu, mu);
  }
#pragma unroll
  for (int j = 0; j < V; ++j) {
    float ge[4] = {g[j].x, g[j].y, g[j].z, g[j].w};
    float xe[4] = {xh[j].x, xh[j].y, xh[j].z, xh[j].w};
    float f0 = 0.f, f1 = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (post_relu && !((xe[k] - mu) * sc + be > 0.f)) ge[k] = 0.f;
      xe[k] = (xe[k] - mu) * is;
      f0 += ge[k]; f1 += ge[k] * xe[k];
    }
    g[j] = make_float4(ge[0], ge[1], ge[2], ge[3]);
    xh[j] = make_float4(xe[0], xe[1], xe[2], xe[3]);
    sg += f0; sgx += f1;
  }
  sg = block16_sum(sg, sm[0], wave, lane);
  sgx = block16_sum(sgx, sm[1], wave, lane);
  if (threadIdx.x == 0) {
    if (ps.dbeta) ps.dbeta[cl] = accumulate ? ps.dbeta[cl] + (float)sg : (float)sg;
    if (ps.dgamma) ps.dgamma[cl] = accumulate ? ps.dgamma[cl] + (float)sgx : (float)sgx;
  }
  if (!have) return;
  const double inv_cnt = 1.0 / ((double)N * HW);
  const float mg = (float)(sg * inv_cnt), mgx = (float)(sgx * inv_cnt);
  float* op = c < C1 ? dx1 + ((size_t)wave * C1 + c) * HW : dx2 + ((size_t)wave * (C - C1) + cl) * HW;
#pragma unroll
  for (int j = 0; j < V; ++j) {
    float4 o;
    o.x = sc * (g[j].x - mg - xh[j].x * mgx); o.y = sc * (g[j].y - mg - xh[j].y * mgx);
    o.z = sc * (g[j].z - mg - xh[j].z * mgx); o.w = sc * (g[j].w - mg - xh[j].w * mgx);
    *reinterpret_cast<float4*>(op + 4 * (lane + 64 * j)) = o;
  }
}

int small_v(int N, int HW) {
  if (N < 1 || N > 16) return 0;
  return HW == 256 ? 1 : HW == 512 ? 2 : HW == 1024 ? 4 : HW == 2048 ? 8 : 0;
}

}  // namespace

extern "C" int dlio_bn_small_ok(int N, int HW) { return small_v(N, HW) != 0; }

extern "C" int dlio_bn_small_fwd(const float* x, int N, int x_ctot, int x_coff, int C, int C1, int HW, int post_relu,
                                 const float* gamma1, const float* beta1, float* running_mean1, float* running_var1,
                                 const float* gamma2, const float* beta2, float* running_mean2, float* running_var2,
                                 float eps, float momentum, float* mean, float* invstd, float* scale, float* shift_out,
                                 const float* residual, int r_ctot, int r_coff, const float* r_mean, const float* r_scale,
                                 const float* r_shift, float* y, int y_ctot, int y_coff, float* gap_out, int gap_ctot,
                                 int gap_coff, dlio_stream_t stream) {
  if (!x || !mean || !invstd || !scale || C <= 0 || C1 < 0 || C1 > C || N <= 0 || HW <= 0) return DLIO_EINVAL;
  const int v = small_v(N, HW);
  if (!v) return DLIO_EUNSUP;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(residual)) & 15) return DLIO_EUNSUP;
  hipStream_t s = as_stream(stream);
  const BnSet s1{gamma1, beta1, running_mean1, running_var1, nullptr, nullptr};
  const BnSet s2{gamma2, beta2, running_mean2, running_var2, nullptr, nullptr};
  DlioProfScope prof(7, s, 0.0, 4.0 * N * (double)C * HW * (y ? (residual ? 3.0 : 2.0) : 1.0));
#define BNS(VV) hipLaunchKernelGGL(bn_small_fwd_kernel<VV>, dim3((unsigned)C), dim3(1024), 0, s, x, x_ctot, x_coff, N, C1, s1, s2, eps, \
                                   momentum, mean, invstd, scale, shift_out, residual, r_ctot, r_coff, r_mean, r_scale, r_shift, y,     \
                                   y_ctot, y_coff, gap_out, gap_ctot, gap_coff, post_relu)
  if (v == 1) BNS(1); else if (v == 2) BNS(2); else if (v == 4) BNS(4); else BNS(8);
#undef BNS
  return dlio_check_launch();
}

extern "C" int dlio_bn_small_bwd(const float* dy, int dy_ctot, int dy_coff, const float* x, int x_ctot, int x_coff,
                                 const float* mean, const float* invstd, const float* scale, const float* beta1,
                                 const float* beta2, float* dx1, float* dx2, float* dgamma1, float* dbeta1, float* dgamma2,
                                 float* dbeta2, int accumulate, int N, int C, int C1, int HW, int post_relu,
                                 dlio_stream_t stream) {
  if (!dy || !x || !mean || !invstd || !scale || C <= 0 || C1 < 0 || C1 > C || N <= 0 || HW <= 0) return DLIO_EINVAL;
  if ((C1 > 0 && !dx1) || (C1 < C && !dx2)) return DLIO_EINVAL;
  const int v = small_v(N, HW);
  if (!v) return DLIO_EUNSUP;
  if ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dx1) |
       reinterpret_cast<uintptr_t>(dx2)) & 15)
    return DLIO_EUNSUP;
  hipStream_t s = as_stream(stream);
  const BnSet s1{nullptr, beta1, nullptr, nullptr, dgamma1, dbeta1};
  const BnSet s2{nullptr, beta2, nullptr, nullptr, dgamma2, dbeta2};
  DlioProfScope prof(9, s, 0.0, 3.0 * 4.0 * N * (double)C * HW);
#define BNS(VV) hipLaunchKernelGGL(bn_small_bwd_kernel<VV>, dim3((unsigned)C), dim3(1024), 0, s, dy, dy_ctot, dy_coff, x, x_ctot, x_coff, \
                                   N, C, C1, s1, s2, mean, invstd, scale, dx1, dx2, accumulate, post_relu)
  if (v == 1) BNS(1); else if (v == 2) BNS(2); else if (v == 4) BNS(4); else BNS(8);
#undef BNS
  return dlio_check_launch();
}
