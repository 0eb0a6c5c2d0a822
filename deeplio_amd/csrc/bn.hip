// Per-channel statistics and BatchNorm2d (train + eval, forward + backward) over NCHW
// channel slices.  Memory-bound streaming kernels: float4 loads where HW % 4 == 0,
// fp64 accumulation (free under the HBM bound) so that E[x^2]-E[x]^2 keeps fp32 accuracy.
//
// Replaces nn.BatchNorm2d at pointseg_net.py:19, pointseg_modules.py:98-106,
// base_net.py:63, resnet.py:38, lidar_feat_nets.py:280-301 and the conv-bias gradient
// reduction of nn.Conv2d backward.
#include "common.h"
#include "pool_strip.h"

namespace {

constexpr int RB = 256;  // reduction block

// split the (n, hw) domain of one channel over `splits` blocks.
// mode 0: stats (sum x', sum x'^2)    mode 1: bn backward (sum g, sum g*xh)   mode 2: sum only
// MM (mode 0): the block also leaves the smallest and the largest value it saw in mm[c][split][2] (floats) -- the exact range of
// a channel, from which the squeeze BatchNorm's two-piece planes take their scale (fire_expand.hip) instead of the analytic bound
template <int MODE, int UN = 1, bool MM = false>
__global__ __launch_bounds__(RB) void chan_reduce_kernel(
    const float* __restrict__ a, int a_ctot, int a_coff, const float* __restrict__ x, int x_ctot,
    int x_coff, const float* __restrict__ mean, const float* __restrict__ invstd,
    const float* __restrict__ scale, const float* __restrict__ beta, int N, int C, int HW,
    int pre_relu, int post_relu, int splits, double* __restrict__ part, float* __restrict__ mm = nullptr) {
  __shared__ double sm[2][16];
  const int c = blockIdx.x / splits;
  const int sp = blockIdx.x % splits;
  double s0 = 0.0, s1 = 0.0;
  float vmin = 3.0e38f, vmax = -3.0e38f;
  float mu = 0.f, is = 0.f, sc = 0.f, be = 0.f;
  if (MODE == 1) { mu = mean[c]; is = invstd[c]; sc = scale[c]; be = beta ? beta[c] : 0.f; }
  const int64_t total = (int64_t)N * HW;
  const bool vec = (HW & 3) == 0;
  if (vec) {
    // (n, p) cursor advanced by the block stride (one division up front) and UN independent loads
    // per operand in flight: beside a resident MFMA kernel only a few wave slots per SIMD are
    // free, and bytes in flight per wave are what keeps the HBM pipe full then
    const int64_t total4 = total >> 2;
    const int hw4 = HW >> 2;
    const int64_t stride = (int64_t)splits * RB;
    const int q = (int)(stride / hw4), r = (int)(stride - (int64_t)q * hw4);
    int64_t i = (int64_t)sp * RB + threadIdx.x;
    int n = (int)(i / hw4), p = (int)(i - (int64_t)n * hw4);
    const float* ab = a + ((size_t)a_coff + c) * HW;
    const float* xb = MODE == 1 ? x + ((size_t)x_coff + c) * HW : nullptr;
    const size_t an = (size_t)a_ctot * HW, xn = (size_t)x_ctot * HW;
    auto acc = [&](const float4& av, const float4& xv) {
      const float ae[4] = {av.x, av.y, av.z, av.w};
      float f0 = 0.f, f1 = 0.f;
      if (MODE == 1) {
        const float xe[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float xx = pre_relu ? fmaxf(xe[k], 0.f) : xe[k];
          const float xh = (xx - mu) * is;
          float g = ae[k];
          if (post_relu && !((xx - mu) * sc + be > 0.f)) g = 0.f;
          f0 += g; f1 += g * xh;
        }
        s0 += f0; s1 += f1;
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float xx = (MODE == 0 && pre_relu) ? fmaxf(ae[k], 0.f) : ae[k];
          f0 += xx; f1 += xx * xx;
          if constexpr (MM) { vmin = fminf(vmin, xx); vmax = fmaxf(vmax, xx); }
        }
        s0 += f0; s1 += (MODE == 0) ? (double)f1 : 0.0;
      }
    };
    if (UN > 1) {
      for (; i + (UN - 1) * stride < total4; i += UN * stride) {
        float4 av[UN], xv[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
          av[u] = *reinterpret_cast<const float4*>(ab + (size_t)n * an + ((size_t)p << 2));
          if (MODE == 1) xv[u] = *reinterpret_cast<const float4*>(xb + (size_t)n * xn + ((size_t)p << 2));
          else xv[u] = av[u];
          n += q; p += r;
          if (p >= hw4) { p -= hw4; ++n; }
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) acc(av[u], xv[u]);
      }
    }
    for (; i < total4; i += stride) {
      const float4 av = *reinterpret_cast<const float4*>(ab + (size_t)n * an + ((size_t)p << 2));
      float4 xv = av;
      if (MODE == 1) xv = *reinterpret_cast<const float4*>(xb + (size_t)n * xn + ((size_t)p << 2));
      acc(av, xv);
      n += q; p += r;
      if (p >= hw4) { p -= hw4; ++n; }
    }
  } else {
    for (int64_t i = (int64_t)sp * RB + threadIdx.x; i < total; i += (int64_t)splits * RB) {
      const int n = (int)(i / HW);
      const int p = (int)(i - (int64_t)n * HW);
      float av = a[((size_t)n * a_ctot + a_coff + c) * HW + p];
      if (MODE == 1) {
        float xx = x[((size_t)n * x_ctot + x_coff + c) * HW + p];
        if (pre_relu) xx = fmaxf(xx, 0.f);
        const float xh = (xx - mu) * is;
        if (post_relu && !((xx - mu) * sc + be > 0.f)) av = 0.f;
        s0 += av; s1 += (double)av * xh;
      } else {
        if (MODE == 0 && pre_relu) av = fmaxf(av, 0.f);
        s0 += av;
        if (MODE == 0) s1 += (double)av * av;
        if constexpr (MM) { vmin = fminf(vmin, av); vmax = fmaxf(vmax, av); }
      }
    }
  }
  double r0 = block_sum_d(s0, sm[0]);
  double r1 = block_sum_d(s1, sm[1]);
  if (threadIdx.x == 0) {
    part[((size_t)c * splits + sp) * 2 + 0] = r0;
    part[((size_t)c * splits + sp) * 2 + 1] = r1;
  }
  if constexpr (MM) {
    __shared__ float smm[2][RB / 64];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      vmin = fminf(vmin, __shfl_xor(vmin, o, 64));
      vmax = fmaxf(vmax, __shfl_xor(vmax, o, 64));
    }
    if ((threadIdx.x & 63) == 0) { smm[0][threadIdx.x >> 6] = vmin; smm[1][threadIdx.x >> 6] = vmax; }
    __syncthreads();
    if (threadIdx.x == 0 && mm) {
      float lo = smm[0][0], hi = smm[1][0];
#pragma unroll
      for (int w = 1; w < RB / 64; ++w) { lo = fminf(lo, smm[0][w]); hi = fmaxf(hi, smm[1][w]); }
      mm[((size_t)c * splits + sp) * 2 + 0] = lo;
      mm[((size_t)c * splits + sp) * 2 + 1] = hi;
    }
  }
}

__global__ void chan_reduce_final(const double* __restrict__ part, int C, int splits,
                                  double* __restrict__ o0, double* __restrict__ o1,
                                  float* __restrict__ f0, float* __restrict__ f1, int accumulate) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double a = 0.0, b = 0.0;
  for (int s = 0; s < splits; ++s) {
    a += part[((size_t)c * splits + s) * 2 + 0];
    b += part[((size_t)c * splits + s) * 2 + 1];
  }
  if (o0) o0[c] = a;
  if (o1) o1[c] = b;
  if (f0) f0[c] = accumulate ? f0[c] + (float)a : (float)a;
  if (f1) f1[c] = accumulate ? f1[c] + (float)b : (float)b;
}

// statistics epilogue of a train-mode BatchNorm: sums the split partials of one channel and
// finalises mean / invstd / scale and the running statistics in the same launch
__global__ void chan_stats_finalize_kernel(const double* __restrict__ part, int C, int splits,
                                           double count, const float* __restrict__ gamma, float eps,
                                           float momentum, float* running_mean, float* running_var,
                                           float* mean, float* invstd, float* scale,
                                           const float* __restrict__ beta = nullptr, float* shift_out = nullptr) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  if (shift_out) shift_out[c] = beta ? beta[c] : 0.f;     // third row of an apply-on-load table (mean, scale, beta)
  double a = 0.0, b = 0.0;
  for (int s = 0; s < splits; ++s) {
    a += part[((size_t)c * splits + s) * 2 + 0];
    b += part[((size_t)c * splits + s) * 2 + 1];
  }
  const double m = a / count;
  double var = b / count - m * m;
  if (var < 0.0) var = 0.0;
  const float is = (float)(1.0 / sqrt(var + (double)eps));
  mean[c] = (float)m;
  invstd[c] = is;
  scale[c] = (gamma ? gamma[c] : 1.f) * is;
  if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)m;
  if (running_var) {
    const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
  }
}

int pick_splits(int N, int C, int HW) {
  int64_t per_chan = (int64_t)N * HW;
  int64_t want = cdiv64(2048, C);                 // ~2048 blocks in total
  int64_t max_sp = cdiv64(per_chan, (int64_t)RB * 8);  // >= 8 elements (or float4s) per thread
  if (want > max_sp) want = max_sp;
  if (want < 1) want = 1;
  if (want > 512) want = 512;
  return (int)want;
}

__global__ void bn_finalize_kernel(const double* __restrict__ sum, const double* __restrict__ sumsq,
                                   int C, double count, const float* __restrict__ gamma, float eps,
                                   float momentum, float* running_mean, float* running_var,
                                   float* mean, float* invstd, float* scale) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const double m = sum[c] / count;
  double var = sumsq[c] / count - m * m;
  if (var < 0.0) var = 0.0;
  const float is = (float)(1.0 / sqrt(var + (double)eps));
  mean[c] = (float)m;
  invstd[c] = is;
  scale[c] = (gamma ? gamma[c] : 1.f) * is;
  if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)m;
  if (running_var) {
    const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
  }
}

__global__ void bn_eval_params_kernel(const float* __restrict__ rm, const float* __restrict__ rv,
                                      const float* __restrict__ gamma, float eps, int C,
                                      float* mean, float* invstd, float* scale) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float is = 1.0f / sqrtf(rv[c] + eps);
  mean[c] = rm[c];
  invstd[c] = is;
  scale[c] = (gamma ? gamma[c] : 1.f) * is;
}

// elementwise over (n, c, hw); one block row per (n, c) plane chunk
template <bool VEC>
__global__ __launch_bounds__(256) void bn_apply_kernel(
    const float* __restrict__ x, int x_ctot, int x_coff, const float* __restrict__ mean,
    const float* __restrict__ scale, const float* __restrict__ beta, const float* residual,
    int r_ctot, int r_coff, float* y, int y_ctot, int y_coff, int N, int C, int HW, int pre_relu,
    int post_relu) {
  const int64_t planes = (int64_t)N * C;
  const int per = VEC ? (HW >> 2) : HW;
  const int64_t total = planes * per;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t pl = i / per;
    const int p = (int)(i - pl * per);
    const int n = (int)(pl / C), c = (int)(pl - (int64_t)n * C);
    const float mu = mean[c], sc = scale[c], be = beta ? beta[c] : 0.f;
    if (VEC) {
      const size_t xo = ((size_t)n * x_ctot + x_coff + c) * HW + ((size_t)p << 2);
      const size_t yo = ((size_t)n * y_ctot + y_coff + c) * HW + ((size_t)p << 2);
      float4 v = *reinterpret_cast<const float4*>(x + xo);
      float e[4] = {v.x, v.y, v.z, v.w};
      float4 rv = make_float4(0.f, 0.f, 0.f, 0.f);
      if (residual)
        rv = *reinterpret_cast<const float4*>(residual + ((size_t)n * r_ctot + r_coff + c) * HW + ((size_t)p << 2));
      float re[4] = {rv.x, rv.y, rv.z, rv.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float xx = pre_relu ? fmaxf(e[k], 0.f) : e[k];
        float o = (xx - mu) * sc + be;
        if (post_relu) o = fmaxf(o, 0.f);
        e[k] = o + re[k];
      }
      *reinterpret_cast<float4*>(y + yo) = make_float4(e[0], e[1], e[2], e[3]);
    } else {
      const size_t xo = ((size_t)n * x_ctot + x_coff + c) * HW + p;
      const size_t yo = ((size_t)n * y_ctot + y_coff + c) * HW + p;
      float xx = x[xo];
      if (pre_relu) xx = fmaxf(xx, 0.f);
      float o = (xx - mu) * sc + be;
      if (post_relu) o = fmaxf(o, 0.f);
      if (residual) o += residual[((size_t)n * r_ctot + r_coff + c) * HW + p];
      y[yo] = o;
    }
  }
}

template <bool VEC>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(
    const float* __restrict__ dy, int dy_ctot, int dy_coff, const float* __restrict__ x,
    int x_ctot, int x_coff, const float* __restrict__ mean, const float* __restrict__ invstd,
    const float* __restrict__ scale, const float* __restrict__ beta,
    const double* __restrict__ sum_g, const double* __restrict__ sum_gx, float* dx, int dx_ctot,
    int dx_coff, int N, int C, int HW, int pre_relu, int post_relu, int use_batch_stats) {
  const int64_t planes = (int64_t)N * C;
  const int per = VEC ? (HW >> 2) : HW;
  const int64_t total = planes * per;
  const double inv_cnt = 1.0 / ((double)N * HW);
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t pl = i / per;
    const int p = (int)(i - pl * per);
    const int n = (int)(pl / C), c = (int)(pl - (int64_t)n * C);
    const float mu = mean[c], is = invstd[c], sc = scale[c], be = beta ? beta[c] : 0.f;
    float mg = 0.f, mgx = 0.f;
    if (use_batch_stats) { mg = (float)(sum_g[c] * inv_cnt); mgx = (float)(sum_gx[c] * inv_cnt); }
    constexpr int V = VEC ? 4 : 1;
    const size_t off = VEC ? ((size_t)p << 2) : (size_t)p;
    const size_t go = ((size_t)n * dy_ctot + dy_coff + c) * HW + off;
    const size_t xo = ((size_t)n * x_ctot + x_coff + c) * HW + off;
    const size_t oo = ((size_t)n * dx_ctot + dx_coff + c) * HW + off;
    float ge[4], xe[4];
    if (VEC) {
      float4 gv = *reinterpret_cast<const float4*>(dy + go);
      float4 xv = *reinterpret_cast<const float4*>(x + xo);
      ge[0] = gv.x; ge[1] = gv.y; ge[2] = gv.z; ge[3] = gv.w;
      xe[0] = xv.x; xe[1] = xv.y; xe[2] = xv.z; xe[3] = xv.w;
    } else {
      ge[0] = dy[go]; xe[0] = x[xo];
    }
#pragma unroll
    for (int k = 0; k < V; ++k) {
      const float xraw = xe[k];
      const float xx = pre_relu ? fmaxf(xraw, 0.f) : xraw;
      float g = ge[k];
      if (post_relu && !((xx - mu) * sc + be > 0.f)) g = 0.f;
      const float xh = (xx - mu) * is;
      float o = sc * (g - mg - xh * mgx);
      if (pre_relu && !(xraw > 0.f)) o = 0.f;
      ge[k] = o;
    }
    if (VEC) *reinterpret_cast<float4*>(dx + oo) = make_float4(ge[0], ge[1], ge[2], ge[3]);
    else dx[oo] = ge[0];
  }
}

__global__ void bn_param_grads_kernel(const double* __restrict__ sum_g,
                                      const double* __restrict__ sum_gx, float* dgamma,
                                      float* dbeta, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  if (dgamma) dgamma[c] = (float)sum_gx[c];
  if (dbeta) dbeta[c] = (float)sum_g[c];
}

// ---- plane-structured apply kernels: one workgroup per (n, c) plane (or plane chunk) ---------
// The workgroup first sums the split partials of ITS channel (<= 512 doubles, fixed order: every
// workgroup of the channel computes bit-identical statistics) and finalises them itself, so the
// separate finalise launches (74 + 74 tiny kernels per training step) disappear without any
// cross-workgroup hand-off; the workgroup of plane (n=0, chunk 0) publishes mean / invstd / scale,
// the running statistics and dgamma / dbeta.  Optional by-product: the global average of every
// output plane (the SELayer that follows a Fire block needs it) -- same thread/element mapping
// and fp64 accumulation as gap_fwd_kernel, so bit-identical to a separate pass.
__device__ __forceinline__ void plane_partials(const double* __restrict__ part, int c, int splits,
                                               double* sm0, double* sm1, double& a, double& b) {
  double ta = 0.0, tb = 0.0;
  for (int q = threadIdx.x; q < splits; q += 256) {
    ta += part[((size_t)c * splits + q) * 2 + 0];
    tb += part[((size_t)c * splits + q) * 2 + 1];
  }
  __shared__ double bc[2];
  ta = block_sum_d(ta, sm0);
  tb = block_sum_d(tb, sm1);
  if (threadIdx.x == 0) { bc[0] = ta; bc[1] = tb; }
  __syncthreads();
  a = bc[0]; b = bc[1];
}

template <bool VEC, int UN = 1>
__global__ __launch_bounds__(256) void bn_plane_apply_kernel(
    const float* __restrict__ x, int x_ctot, int x_coff, const double* __restrict__ part, int splits,
    double count, const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
    float momentum, float* running_mean, float* running_var, float* mean_o, float* invstd_o,
    float* scale_o, const float* residual, int r_ctot, int r_coff, float* y, int y_ctot, int y_coff,
    int N, int C, int HW, int pre_relu, int post_relu, int chunks, int chunk_len,
    float* __restrict__ gap_out, int gap_ctot, int gap_coff, const float* __restrict__ r_mean = nullptr,
    const float* __restrict__ r_scale = nullptr, const float* __restrict__ r_shift = nullptr,
    float* __restrict__ amax_out = nullptr) {
  __shared__ double sm[2][16];
  __shared__ unsigned s_amax;
  float amax = 0.f;                    // largest |y| this thread wrote (amax_out: the next layer's two-piece operand scale)
  const int chunk = blockIdx.x % chunks;
  const int pl = blockIdx.x / chunks;
  const int n = pl / C, c = pl - n * C;
  double a, b;
  plane_partials(part, c, splits, sm[0], sm[1], a, b);
  const double m = a / count;
  double var = b / count - m * m;
  if (var < 0.0) var = 0.0;
  const float is = (float)(1.0 / sqrt(var + (double)eps));
  const float mu = (float)m, sc = (gamma ? gamma[c] : 1.f) * is, be = beta ? beta[c] : 0.f;
  if (n == 0 && chunk == 0 && threadIdx.x == 0) {
    mean_o[c] = mu; invstd_o[c] = is; scale_o[c] = sc;
    if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mu;
    if (running_var) {
      const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
    }
  }
  const float* xp = x + ((size_t)n * x_ctot + x_coff + c) * HW;
  float* yp = y + ((size_t)n * y_ctot + y_coff + c) * HW;
  const float* rp = residual ? residual + ((size_t)n * r_ctot + r_coff + c) * HW : nullptr;
  // residual stored before ITS BatchNorm + ReLU (apply-on-load): r' = max(0, (r - r_mean) * r_scale + r_shift)
  const bool raff = rp && r_scale;
  const float rmu = raff ? r_mean[r_coff + c] : 0.f, rsc = raff ? r_scale[r_coff + c] : 1.f,
              rsh = raff ? r_shift[r_coff + c] : 0.f;
  const int per = VEC ? (HW >> 2) : HW;
  const int i1 = min(per, (chunk + 1) * chunk_len);
  double gs = 0.0;
  int i = chunk * chunk_len + threadIdx.x;
  if (VEC && UN > 1) {
    for (; i + (UN - 1) * 256 < i1; i += UN * 256) {
      float4 v[UN], rv[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        v[u] = *reinterpret_cast<const float4*>(xp + ((size_t)(i + u * 256) << 2));
        rv[u] = rp ? *reinterpret_cast<const float4*>(rp + ((size_t)(i + u * 256) << 2)) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        float e[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
        float re[4] = {rv[u].x, rv[u].y, rv[u].z, rv[u].w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float xx = pre_relu ? fmaxf(e[k], 0.f) : e[k];
          float o = (xx - mu) * sc + be;
          if (post_relu) o = fmaxf(o, 0.f);
          if (raff) re[k] = fmaxf((re[k] - rmu) * rsc + rsh, 0.f);
          e[k] = o + re[k];
        }
        *reinterpret_cast<float4*>(yp + ((size_t)(i + u * 256) << 2)) = make_float4(e[0], e[1], e[2], e[3]);
        gs += (double)((e[0] + e[1]) + (e[2] + e[3]));
        amax = amax4(amax, e[0], e[1], e[2], e[3]);
      }
    }
  }
  for (; i < i1; i += 256) {
    if (VEC) {
      const float4 v = *reinterpret_cast<const float4*>(xp + ((size_t)i << 2));
      float e[4] = {v.x, v.y, v.z, v.w};
      float4 rv = make_float4(0.f, 0.f, 0.f, 0.f);
      if (rp) rv = *reinterpret_cast<const float4*>(rp + ((size_t)i << 2));
      float re[4] = {rv.x, rv.y, rv.z, rv.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float xx = pre_relu ? fmaxf(e[k], 0.f) : e[k];
        float o = (xx - mu) * sc + be;
        if (post_relu) o = fmaxf(o, 0.f);
        if (raff) re[k] = fmaxf((re[k] - rmu) * rsc + rsh, 0.f);
        e[k] = o + re[k];
      }
      *reinterpret_cast<float4*>(yp + ((size_t)i << 2)) = make_float4(e[0], e[1], e[2], e[3]);
      gs += (double)((e[0] + e[1]) + (e[2] + e[3]));
      amax = amax4(amax, e[0], e[1], e[2], e[3]);
    } else {
      float xx = xp[i];
      if (pre_relu) xx = fmaxf(xx, 0.f);
      float o = (xx - mu) * sc + be;
      if (post_relu) o = fmaxf(o, 0.f);
      if (rp) o += raff ? fmaxf((rp[i] - rmu) * rsc + rsh, 0.f) : rp[i];
      yp[i] = o;
      gs += o;
      amax = fmaxf(amax, fabsf(o));
    }
  }
  if (gap_out) {                       // chunks == 1 whenever gap_out is set
    const double r = block_sum_d(gs, sm[0]);
    if (threadIdx.x == 0) gap_out[(size_t)n * gap_ctot + gap_coff + c] = (float)(r / (double)HW);
  }
  if (amax_out) block_amax_commit(amax, amax_out, &s_amax);
}

template <bool VEC, int UN = 1>
__global__ __launch_bounds__(256) void bn_plane_bwd_kernel(
    const float* __restrict__ dy, int dy_ctot, int dy_coff, const float* __restrict__ x, int x_ctot,
    int x_coff, const float* __restrict__ mean, const float* __restrict__ invstd,
    const float* __restrict__ scale, const float* __restrict__ beta, const double* __restrict__ part,
    const double* __restrict__ lpart, double inv_cnt,
    int splits, float* dx, int dx_ctot, int dx_coff, float* dgamma, float* dbeta, int accumulate,
    int N, int C, int HW, int pre_relu, int post_relu, int use_batch_stats, int chunks, int chunk_len,
    float* __restrict__ amax_out = nullptr) {
  __shared__ double sm[2][16];
  __shared__ unsigned s_amax;
  float amax = 0.f;                    // largest |dx| this thread wrote (amax_out: the data gradient's two-piece operand scale)
  const int chunk = blockIdx.x % chunks;
  const int pl = blockIdx.x / chunks;
  const int n = pl / C, c = pl - n * C;
  double sg, sgx;
  plane_partials(part, c, splits, sm[0], sm[1], sg, sgx);
  if (n == 0 && chunk == 0) {
    // dgamma / dbeta are THIS replica's sums (the gradient all-reduce adds the others); with
    // synchronised statistics `part` holds the global sums and `lpart` the local ones
    double lg = sg, lgx = sgx;
    if (lpart) plane_partials(lpart, c, splits, sm[0], sm[1], lg, lgx);
    if (threadIdx.x == 0) {
      if (dbeta) dbeta[c] = accumulate ? dbeta[c] + (float)lg : (float)lg;
      if (dgamma) dgamma[c] = accumulate ? dgamma[c] + (float)lgx : (float)lgx;
    }
  }
  const float mu = mean[c], is = invstd[c], sc = scale[c], be = beta ? beta[c] : 0.f;
  float mg = 0.f, mgx = 0.f;
  if (use_batch_stats) { mg = (float)(sg * inv_cnt); mgx = (float)(sgx * inv_cnt); }
  const float* gp = dy + ((size_t)n * dy_ctot + dy_coff + c) * HW;
  const float* xp = x + ((size_t)n * x_ctot + x_coff + c) * HW;
  float* op = dx + ((size_t)n * dx_ctot + dx_coff + c) * HW;
  const int per = VEC ? (HW >> 2) : HW;
  const int i1 = min(per, (chunk + 1) * chunk_len);
  constexpr int V = VEC ? 4 : 1;
  int i = chunk * chunk_len + threadIdx.x;
  if (VEC && UN > 1) {
    for (; i + (UN - 1) * 256 < i1; i += UN * 256) {
      float4 gv[UN], xv[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        gv[u] = *reinterpret_cast<const float4*>(gp + ((size_t)(i + u * 256) << 2));
        xv[u] = *reinterpret_cast<const float4*>(xp + ((size_t)(i + u * 256) << 2));
      }
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        float ge[4] = {gv[u].x, gv[u].y, gv[u].z, gv[u].w};
        const float xe[4] = {xv[u].x, xv[u].y, xv[u].z, xv[u].w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float xraw = xe[k];
          const float xx = pre_relu ? fmaxf(xraw, 0.f) : xraw;
          float g = ge[k];
          if (post_relu && !((xx - mu) * sc + be > 0.f)) g = 0.f;
          const float xh = (xx - mu) * is;
          float o = sc * (g - mg - xh * mgx);
          if (pre_relu && !(xraw > 0.f)) o = 0.f;
          ge[k] = o;
        }
        *reinterpret_cast<float4*>(op + ((size_t)(i + u * 256) << 2)) = make_float4(ge[0], ge[1], ge[2], ge[3]);
        amax = amax4(amax, ge[0], ge[1], ge[2], ge[3]);
      }
    }
  }
  for (; i < i1; i += 256) {
    float ge[4], xe[4];
    if (VEC) {
      const float4 gv = *reinterpret_cast<const float4*>(gp + ((size_t)i << 2));
      const float4 xv = *reinterpret_cast<const float4*>(xp + ((size_t)i << 2));
      ge[0] = gv.x; ge[1] = gv.y; ge[2] = gv.z; ge[3] = gv.w;
      xe[0] = xv.x; xe[1] = xv.y; xe[2] = xv.z; xe[3] = xv.w;
    } else {
      ge[0] = gp[i]; xe[0] = xp[i];
    }
#pragma unroll
    for (int k = 0; k < V; ++k) {
      const float xraw = xe[k];
      const float xx = pre_relu ? fmaxf(xraw, 0.f) : xraw;
      float g = ge[k];
      if (post_relu && !((xx - mu) * sc + be > 0.f)) g = 0.f;
      const float xh = (xx - mu) * is;
      float o = sc * (g - mg - xh * mgx);
      if (pre_relu && !(xraw > 0.f)) o = 0.f;
      ge[k] = o;
    }
    if (VEC) { *reinterpret_cast<float4*>(op + ((size_t)i << 2)) = make_float4(ge[0], ge[1], ge[2], ge[3]); amax = amax4(amax, ge[0], ge[1], ge[2], ge[3]); }
    else { op[i] = ge[0]; amax = fmaxf(amax, fabsf(ge[0])); }
  }
  if (amax_out) block_amax_commit(amax, amax_out, &s_amax);
}

// chunks per plane: one when planes alone fill the chip (or the plane average is wanted), else
// enough for ~2048 workgroups; chunk_len is a multiple of 256 work items
// ---- BatchNorm backward behind a 3x3 / pad 1 / stride (SH, 2) max-pool: the gradient of the activated tensor is never
// materialised -- both passes gather it from the pooled gradient and the arg-max map (dx[h][w] = sum over the <= 3 x 2
// windows whose arg-max is (h, w): maxpool3_bwd's rule) for the four columns of a float4 while they stream x.
// Columns w0 .. w0+3 (w0 % 4 == 0) see the outputs ow0 = w0 / 2 .. ow0 + 2: k = 0 -> (ow0, kx 1); 1 -> (ow0, 2), (ow0+1, 0);
// 2 -> (ow0+1, 1); 3 -> (ow0+1, 2), (ow0+2, 0).
template <int SH>
__device__ __forceinline__ void pool_grad4(const float* __restrict__ gp, const uint8_t* __restrict__ ip, int h, int w0,
                                           int OH, int OW, float (&g)[4]) {
  g[0] = g[1] = g[2] = g[3] = 0.f;
  const int ow0 = w0 >> 1;
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const int num = h + 1 - ky;                 // = oh * SH
    if (num < 0 || (SH == 2 && (num & 1))) continue;
    const int oh = SH == 2 ? num >> 1 : num;
    if (oh >= OH) continue;
    const size_t ro = (size_t)oh * OW + ow0;
    const float2 d01 = *reinterpret_cast<const float2*>(gp + ro);
    const unsigned short i01 = *reinterpret_cast<const unsigned short*>(ip + ro);
    const bool v2 = ow0 + 2 < OW;
    const float d2 = v2 ? gp[ro + 2] : 0.f;
    const int i0 = i01 & 255, i1 = i01 >> 8, i2 = v2 ? (int)ip[ro + 2] : -1;
    const int b = ky * 3;
    if (i0 == b + 1) g[0] += d01.x;
    if (i0 == b + 2) g[1] += d01.x;
    if (i1 == b + 0) g[1] += d01.y;
    if (i1 == b + 1) g[2] += d01.y;
    if (i1 == b + 2) g[3] += d01.y;
    if (i2 == b + 0) g[3] += d2;
  }
}

template <int SH>
__global__ __launch_bounds__(RB) void bn_pool_bwd_reduce_kernel(
    const float* __restrict__ dyp, const uint8_t* __restrict__ idx, const float* __restrict__ x,
    const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ scale,
    const float* __restrict__ beta, int N, int C, int H, int W, int OH, int OW, int splits, double* __restrict__ part) {
  __shared__ double sm[2][16];
  const int c = blockIdx.x / splits, sp = blockIdx.x % splits;
  const float mu = mean[c], is = invstd[c], sc = scale[c], be = beta ? beta[c] : 0.f;
  const int w4 = W >> 2;
  const int64_t per = (int64_t)H * w4, total = (int64_t)N * per;
  double s0 = 0.0, s1 = 0.0;
  for (int64_t i = (int64_t)sp * RB + threadIdx.x; i < total; i += (int64_t)splits * RB) {
    const int n = (int)(i / per);
    const int r = (int)(i - (int64_t)n * per);
    const int h = r / w4, w0 = (r - h * w4) << 2;
    const size_t pl = (size_t)n * C + c;
    const float4 xv = *reinterpret_cast<const float4*>(x + (pl * H + h) * W + w0);
    float g[4];
    pool_grad4<SH>(dyp + pl * OH * OW, idx + pl * OH * OW, h, w0, OH, OW, g);
    const float xe[4] = {xv.x, xv.y, xv.z, xv.w};
    float f0 = 0.f, f1 = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float xh = (xe[k] - mu) * is;
      const float gg = ((xe[k] - mu) * sc + be > 0.f) ? g[k] : 0.f;
      f0 += gg; f1 += gg * xh;
    }
    s0 += f0; s1 += f1;
  }
  const double r0 = block_sum_d(s0, sm[0]);
  const double r1 = block_sum_d(s1, sm[1]);
  if (threadIdx.x == 0) {
    part[((size_t)c * splits + sp) * 2 + 0] = r0;
    part[((size_t)c * splits + sp) * 2 + 1] = r1;
  }
}

template <int SH>
__global__ __launch_bounds__(256) void bn_pool_bwd_apply_kernel(
    const float* __restrict__ dyp, const uint8_t* __restrict__ idx, const float* __restrict__ x,
    const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ scale,
    const float* __restrict__ beta, const double* __restrict__ part, double inv_cnt, int splits, float* __restrict__ dx,
    float* dgamma, float* dbeta, int accumulate, int N, int C, int H, int W, int OH, int OW, int chunks, int chunk_len) {
  __shared__ double sm[2][16];
  const int chunk = blockIdx.x % chunks;
  const int pl = blockIdx.x / chunks;
  const int n = pl / C, c = pl - n * C;
  double sg, sgx;
  plane_partials(part, c, splits, sm[0], sm[1], sg, sgx);
  if (n == 0 && chunk == 0 && threadIdx.x == 0) {
    if (dbeta) dbeta[c] = accumulate ? dbeta[c] + (float)sg : (float)sg;
    if (dgamma) dgamma[c] = accumulate ? dgamma[c] + (float)sgx : (float)sgx;
  }
  const float mu = mean[c], is = invstd[c], sc = scale[c], be = beta ? beta[c] : 0.f;
  const float mg = (float)(sg * inv_cnt), mgx = (float)(sgx * inv_cnt);
  const int w4 = W >> 2, per = H * w4;
  const int i1 = min(per, (chunk + 1) * chunk_len);
  const float* gp = dyp + (size_t)pl * OH * OW;
  const uint8_t* ip = idx + (size_t)pl * OH * OW;
  for (int i = chunk * chunk_len + threadIdx.x; i < i1; i += 256) {
    const int h = i / w4, w0 = (i - h * w4) << 2;
    const size_t o = ((size_t)pl * H + h) * W + w0;
    const float4 xv = *reinterpret_cast<const float4*>(x + o);
    float g[4];
    pool_grad4<SH>(gp, ip, h, w0, OH, OW, g);
    const float xe[4] = {xv.x, xv.y, xv.z, xv.w};
    float oe[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float xh = (xe[k] - mu) * is;
      const float gg = ((xe[k] - mu) * sc + be > 0.f) ? g[k] : 0.f;
      oe[k] = sc * (gg - mg - xh * mgx);
    }
    *reinterpret_cast<float4*>(dx + o) = make_float4(oe[0], oe[1], oe[2], oe[3]);
  }
}

// the same two kernels as rolling windows (pool_strip.h): a thread owns 4 columns x PR rows, reads each pooled-gradient row
// the strip can be the arg-max of ONCE (the per-element gather above reads it up to three times, under branches) and its
// PR float4 of x; item = (strip, float4 column) of a plane
template <int SH>
__global__ __launch_bounds__(RB) void bn_pool_bwd_reduce_strip(
    const float* __restrict__ dyp, const uint8_t* __restrict__ idx, const float* __restrict__ x,
    const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ scale,
    const float* __restrict__ beta, int N, int C, int H, int W, int OH, int OW, int splits, double* __restrict__ part) {
  __shared__ double sm[2][16];
  const int c = blockIdx.x / splits, sp = blockIdx.x % splits;
  const float mu = mean[c], is = invstd[c], sc = scale[c], be = beta ? beta[c] : 0.f;
  const int w4 = W >> 2, strips = (H + PR - 1) / PR;
  const int64_t per = (int64_t)strips * w4, total = (int64_t)N * per;
  double s0 = 0.0, s1 = 0.0;
  for (int64_t i = (int64_t)sp * RB + threadIdx.x; i < total; i += (int64_t)splits * RB) {
    const int n = (int)(i / per);
    const int r = (int)(i - (int64_t)n * per);
    const int ih0 = (r / w4) * PR, b = r - (r / w4) * w4;
    const size_t pl = (size_t)n * C + c;
    float G[PR][4];
    pool3_strip<SH>(dyp + pl * OH * OW, idx + pl * OH * OW, ih0, b, OH, OW, G);
    float4 xvs[PR];
#pragma unroll
    for (int q = 0; q < PR; ++q) xvs[q] = *reinterpret_cast<const float4*>(x + (pl * H + min(ih0 + q, H - 1)) * W + 4 * b);
#pragma unroll
    for (int q = 0; q < PR; ++q) {
      if (ih0 + q >= H) continue;
      const float xe[4] = {xvs[q].x, xvs[q].y, xvs[q].z, xvs[q].w};
      float f0 = 0.f, f1 = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float xh = (xe[k] - mu) * is;
        const float gg = ((xe[k] - mu) * sc + be > 0.f) ? G[q][k] : 0.f;
        f0 += gg; f1 += gg * xh;
      }
      s0 += f0; s1 += f1;
    }
  }
  const double r0 = block_sum_d(s0, sm[0]);
  const double r1 = block_sum_d(s1, sm[1]);
  if (threadIdx.x == 0) {
    part[((size_t)c * splits + sp) * 2 + 0] = r0;
    part[((size_t)c * splits + sp) * 2 + 1] = r1;
  }
}

template <int SH>
__global__ __launch_bounds__(256) void bn_pool_bwd_apply_strip(
    const float* __restrict__ dyp, const uint8_t* __restrict__ idx, const float* __restrict__ x,
    const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ scale,
    const float* __restrict__ beta, const double* __restrict__ part, double inv_cnt, int splits, float* __restrict__ dx,
    float* dgamma, float* dbeta, int accumulate, int N, int C, int H, int W, int OH, int OW, int chunks, int chunk_len) {
  __shared__ double sm[2][16];
  const int chunk = blockIdx.x % chunks;
  const int pl = blockIdx.x / chunks;
  const int n = pl / C, c = pl - n * C;
  double sg, sgx;
  plane_partials(part, c, splits, sm[0], sm[1], sg, sgx);
  if (n == 0 && chunk == 0 && threadIdx.x == 0) {
    if (dbeta) dbeta[c] = accumulate ? dbeta[c] + (float)sg : (float)sg;
    if (dgamma) dgamma[c] = accumulate ? dgamma[c] + (float)sgx : (float)sgx;
  }
  const float mu = mean[c], is = invstd[c], sc = scale[c], be = beta ? beta[c] : 0.f;
  const float mg = (float)(sg * inv_cnt), mgx = (float)(sgx * inv_cnt);
  const int w4 = W >> 2, strips = (H + PR - 1) / PR, per = strips * w4;
  const int i1 = min(per, (chunk + 1) * chunk_len);
  const float* gp = dyp + (size_t)pl * OH * OW;
  const uint8_t* ip = idx + (size_t)pl * OH * OW;
  for (int i = chunk * chunk_len + threadIdx.x; i < i1; i += 256) {
    const int ih0 = (i / w4) * PR, b = i - (i / w4) * w4;
    float G[PR][4];
    pool3_strip<SH>(gp, ip, ih0, b, OH, OW, G);
    float4 xvs[PR];
#pragma unroll
    for (int q = 0; q < PR; ++q) xvs[q] = *reinterpret_cast<const float4*>(x + ((size_t)pl * H + min(ih0 + q, H - 1)) * W + 4 * b);
#pragma unroll
    for (int q = 0; q < PR; ++q) {
      if (ih0 + q >= H) continue;
      const float xe[4] = {xvs[q].x, xvs[q].y, xvs[q].z, xvs[q].w};
      float oe[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float xh = (xe[k] - mu) * is;
        const float gg = ((xe[k] - mu) * sc + be > 0.f) ? G[q][k] : 0.f;
        oe[k] = sc * (gg - mg - xh * mgx);
      }
      st4<32>(dx + ((size_t)pl * H + ih0 + q) * W + 4 * b, make_float4(oe[0], oe[1], oe[2], oe[3]));
    }
  }
}

static void plane_chunks(int planes, int per, bool whole_plane, int& chunks, int& chunk_len) {
  chunks = 1;
  if (!whole_plane && planes < 2048) {
    chunks = cdiv(2048, planes);
    const int maxc = cdiv(per, 512);
    if (chunks > maxc) chunks = maxc;
    if (chunks < 1) chunks = 1;
  }
  chunk_len = cdiv(cdiv(per, chunks), 256) * 256;
  chunks = cdiv(per, chunk_len);
}

}  // namespace

// loads in flight per thread in the streaming kernels: 4 (statistics, apply) / 2 per operand (backward);
// beside a resident MFMA kernel few wave slots are free, so bandwidth has to come from bytes in
// flight per wave (measured on the full step: 1 -> 33.1 ms, 4 -> 32.8 ms, 8 -> 33.1 ms)
static int bn_unroll() {
  static const int v = 4;
  return v;
}

extern "C" int dlio_chan_stats_splits(int N, int C, int HW) {
  if (N <= 0 || C <= 0 || HW <= 0) return 0;
  return pick_splits(N, C, HW);
}

extern "C" size_t dlio_chan_stats_ws_bytes(int N, int C, int HW) {
  if (N <= 0 || C <= 0 || HW <= 0) return 0;
  // [C][splits][2] doubles (sum, sum of squares) + [C][splits][2] floats (min, max: dlio_bn_split16's two-piece planes)
  return (size_t)C * pick_splits(N, C, HW) * (2 * sizeof(double) + 2 * sizeof(float));
}

// internal (fire_expand.hip): the train-mode statistics partials of dlio_bn_train_apply's first phase + the per-split range of
// every channel behind them in the same workspace; -> the range array, or nullptr when the workspace has no room for it (the
// plain partials are written either way)
float* dlio_internal_stats_partials_mm(const float* x, int N, int x_ctot, int x_coff, int C, int HW, void* ws, size_t ws_bytes,
                                       hipStream_t s, int* rc) {
  const int splits = pick_splits(N, C, HW);
  *rc = DLIO_OK;
  if (ws_bytes < (size_t)C * splits * 2 * sizeof(double)) { *rc = DLIO_EWS; return nullptr; }
  double* part = reinterpret_cast<double*>(ws);
  float* mm = ws_bytes >= (size_t)C * splits * (2 * sizeof(double) + 2 * sizeof(float)) ? reinterpret_cast<float*>(part + (size_t)C * splits * 2)
                                                                                     : nullptr;
  DlioProfScope prof(6, s, 0.0, 4.0 * N * (double)C * HW);
  if (mm)
    hipLaunchKernelGGL((chan_reduce_kernel<0, 4, true>), dim3((unsigned)(C * splits)), dim3(RB), 0, s, x, x_ctot, x_coff,
                       (const float*)nullptr, 0, 0, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr,
                       (const float*)nullptr, N, C, HW, 0, 0, splits, part, mm);
  else
    hipLaunchKernelGGL((chan_reduce_kernel<0, 4>), dim3((unsigned)(C * splits)), dim3(RB), 0, s, x, x_ctot, x_coff,
                       (const float*)nullptr, 0, 0, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr,
                       (const float*)nullptr, N, C, HW, 0, 0, splits, part);
  *rc = dlio_check_launch();
  return mm;
}

static int chan_reduce(int mode, const float* a, int a_ctot, int a_coff, const float* x, int x_ctot,
                       int x_coff, const float* mean, const float* invstd, const float* scale,
                       const float* beta, int N, int C, int HW, int pre_relu, int post_relu,
                       double* o0, double* o1, float* f0, float* f1, int accumulate, void* ws,
                       size_t ws_bytes, hipStream_t s) {
  if (!a || N <= 0 || C <= 0 || HW <= 0 || !ws) return DLIO_EINVAL;
  const int splits = pick_splits(N, C, HW);
  if (ws_bytes < (size_t)C * splits * 2 * sizeof(double)) return DLIO_EWS;
  double* part = reinterpret_cast<double*>(ws);
  dim3 grid((unsigned)(C * splits));
  if (mode == 0)
    hipLaunchKernelGGL(chan_reduce_kernel<0>, grid, dim3(RB), 0, s, a, a_ctot, a_coff, x, x_ctot,
                       x_coff, mean, invstd, scale, beta, N, C, HW, pre_relu, post_relu, splits, part);
  else if (mode == 1)
    hipLaunchKernelGGL(chan_reduce_kernel<1>, grid, dim3(RB), 0, s, a, a_ctot, a_coff, x, x_ctot,
                       x_coff, mean, invstd, scale, beta, N, C, HW, pre_relu, post_relu, splits, part);
  else
    hipLaunchKernelGGL(chan_reduce_kernel<2>, grid, dim3(RB), 0, s, a, a_ctot, a_coff, x, x_ctot,
                       x_coff, mean, invstd, scale, beta, N, C, HW, pre_relu, post_relu, splits, part);
  int rc = dlio_check_launch();
  if (rc) return rc;
  hipLaunchKernelGGL(chan_reduce_final, dim3(cdiv(C, 128)), dim3(128), 0, s, part, C, splits, o0,
                     o1, f0, f1, accumulate);
  return dlio_check_launch();
}

extern "C" int dlio_chan_stats(const float* x, int N, int ctot, int coff, int C, int HW,
                               int pre_relu, double* sum, double* sumsq, void* ws,
                               size_t ws_bytes, dlio_stream_t stream) {
  if (!sum || !sumsq) return DLIO_EINVAL;
  return chan_reduce(0, x, ctot, coff, nullptr, 0, 0, nullptr, nullptr, nullptr, nullptr, N, C, HW,
                     pre_relu, 0, sum, sumsq, nullptr, nullptr, 0, ws, ws_bytes, as_stream(stream));
}

extern "C" int dlio_bn_train_stats(const float* x, int N, int ctot, int coff, int C, int HW,
                                   int pre_relu, const float* gamma, float eps, float momentum,
                                   float* running_mean, float* running_var, float* mean,
                                   float* invstd, float* scale, void* ws, size_t ws_bytes,
                                   const float* beta, float* shift_out, int phase, double count_scale,
                                   dlio_stream_t stream) {
  if (!x || !mean || !invstd || !scale || N <= 0 || C <= 0 || HW <= 0 || !ws || phase < 0 || phase > 2 ||
      !(count_scale >= 1.0))
    return DLIO_EINVAL;
  const int splits = pick_splits(N, C, HW);
  if (ws_bytes < (size_t)C * splits * 2 * sizeof(double)) return DLIO_EWS;
  hipStream_t s = as_stream(stream);
  double* part = reinterpret_cast<double*>(ws);
  if (phase != 2) {
    DlioProfScope prof(6, s, 0.0, 4.0 * N * (double)C * HW);
    hipLaunchKernelGGL(chan_reduce_kernel<0>, dim3((unsigned)(C * splits)), dim3(RB), 0, s, x, ctot, coff,
                       (const float*)nullptr, 0, 0, (const float*)nullptr, (const float*)nullptr,
                       (const float*)nullptr, (const float*)nullptr, N, C, HW, pre_relu, 0, splits, part);
    const int rc = dlio_check_launch();
    if (rc || phase == 1) return rc;
  }
  hipLaunchKernelGGL(chan_stats_finalize_kernel, dim3(cdiv(C, 128)), dim3(128), 0, s, part, C, splits,
                     (double)N * HW * count_scale, gamma, eps, momentum, running_mean, running_var, mean, invstd,
                     scale, beta, shift_out);
  return dlio_check_launch();
}

extern "C" int dlio_bn_bwd_pool(const float* dy_pool, const uint8_t* idx, const float* x, const float* mean,
                                const float* invstd, const float* scale, const float* beta, float* dx,
                                float* dgamma, float* dbeta, int accumulate, int N, int C, int H, int W, int OH,
                                int OW, int SH, void* ws, size_t ws_bytes, dlio_stream_t stream) {
  if (!dy_pool || !idx || !x || !mean || !invstd || !scale || !dx || !ws || N <= 0 || C <= 0) return DLIO_EINVAL;
  if (!((SH == 1 || SH == 2) && (W & 3) == 0 && OW * 2 == W && OH == (H + 2 - 3) / SH + 1)) return DLIO_EUNSUP;
  if (((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dx) | reinterpret_cast<uintptr_t>(dy_pool)) & 15) != 0)
    return DLIO_EUNSUP;
  const int HW = H * W;
  const int splits = pick_splits(N, C, HW);
  if (ws_bytes < (size_t)C * splits * 2 * sizeof(double)) return DLIO_EWS;
  hipStream_t s = as_stream(stream);
  double* part = reinterpret_cast<double*>(ws);
  const double tensor_bytes = 4.0 * N * (double)C * HW;
  static const int strip = 1;
  if (strip) {
    {
      DlioProfScope prof(8, s, 0.0, tensor_bytes * 1.6);
      if (SH == 1)
        hipLaunchKernelGGL(bn_pool_bwd_reduce_strip<1>, dim3((unsigned)(C * splits)), dim3(RB), 0, s, dy_pool, idx, x, mean,
                           invstd, scale, beta, N, C, H, W, OH, OW, splits, part);
      else
        hipLaunchKernelGGL(bn_pool_bwd_reduce_strip<2>, dim3((unsigned)(C * splits)), dim3(RB), 0, s, dy_pool, idx, x, mean,
                           invstd, scale, beta, N, C, H, W, OH, OW, splits, part);
      const int rc = dlio_check_launch();
      if (rc) return rc;
    }
    int chunks, chunk_len;
    plane_chunks(N * C, cdiv(H, PR) * (W / 4), false, chunks, chunk_len);
    const dim3 grid((unsigned)(N * C * chunks));
    const double inv_cnt = 1.0 / ((double)N * HW);
    DlioProfScope prof(9, s, 0.0, tensor_bytes * 2.6);
    if (SH == 1)
      hipLaunchKernelGGL(bn_pool_bwd_apply_strip<1>, grid, dim3(256), 0, s, dy_pool, idx, x, mean, invstd, scale, beta, part,
                         inv_cnt, splits, dx, dgamma, dbeta, accumulate, N, C, H, W, OH, OW, chunks, chunk_len);
    else
      hipLaunchKernelGGL(bn_pool_bwd_apply_strip<2>, grid, dim3(256), 0, s, dy_pool, idx, x, mean, invstd, scale, beta, part,
                         inv_cnt, splits, dx, dgamma, dbeta, accumulate, N, C, H, W, OH, OW, chunks, chunk_len);
    return dlio_check_launch();
  }
  {
    DlioProfScope prof(8, s, 0.0, tensor_bytes * 1.6);
    if (SH == 1)
      hipLaunchKernelGGL(bn_pool_bwd_reduce_kernel<1>, dim3((unsigned)(C * splits)), dim3(RB), 0, s, dy_pool, idx, x, mean,
                         invstd, scale, beta, N, C, H, W, OH, OW, splits, part);
    else
      hipLaunchKernelGGL(bn_pool_bwd_reduce_kernel<2>, dim3((unsigned)(C * splits)), dim3(RB), 0, s, dy_pool, idx, x, mean,
                         invstd, scale, beta, N, C, H, W, OH, OW, splits, part);
    const int rc = dlio_check_launch();
    if (rc) return rc;
  }
  int chunks, chunk_len;
  plane_chunks(N * C, HW / 4, false, chunks, chunk_len);
  const dim3 grid((unsigned)(N * C * chunks));
  const double inv_cnt = 1.0 / ((double)N * HW);
  DlioProfScope prof(9, s, 0.0, tensor_bytes * 2.6);
  if (SH == 1)
    hipLaunchKernelGGL(bn_pool_bwd_apply_kernel<1>, grid, dim3(256), 0, s, dy_pool, idx, x, mean, invstd, scale, beta, part,
                       inv_cnt, splits, dx, dgamma, dbeta, accumulate, N, C, H, W, OH, OW, chunks, chunk_len);
  else
    hipLaunchKernelGGL(bn_pool_bwd_apply_kernel<2>, grid, dim3(256), 0, s, dy_pool, idx, x, mean, invstd, scale, beta, part,
                       inv_cnt, splits, dx, dgamma, dbeta, accumulate, N, C, H, W, OH, OW, chunks, chunk_len);
  return dlio_check_launch();
}

extern "C" int dlio_chan_sum(const float* x, int N, int ctot, int coff, int C, int HW, float* out,
                             int accumulate, void* ws, size_t ws_bytes, dlio_stream_t stream) {
  if (!out) return DLIO_EINVAL;
  return chan_reduce(2, x, ctot, coff, nullptr, 0, 0, nullptr, nullptr, nullptr, nullptr, N, C, HW,
                     0, 0, nullptr, nullptr, out, nullptr, accumulate, ws, ws_bytes, as_stream(stream));
}

extern "C" int dlio_bn_bwd_reduce(const float* dy, int dy_ctot, int dy_coff, const float* x,
                                  int x_ctot, int x_coff, const float* mean, const float* invstd,
                                  const float* scale, const float* beta, int N, int C, int HW,
                                  int pre_relu, int post_relu, double* sum_g, double* sum_gx,
                                  float* dgamma, float* dbeta, int accumulate, void* ws,
                                  size_t ws_bytes, dlio_stream_t stream) {
  if (!x || !mean || !invstd || !scale || !sum_g || !sum_gx) return DLIO_EINVAL;
  return chan_reduce(1, dy, dy_ctot, dy_coff, x, x_ctot, x_coff, mean, invstd, scale, beta, N, C,
                     HW, pre_relu, post_relu, sum_g, sum_gx, dbeta, dgamma, accumulate, ws, ws_bytes,
                     as_stream(stream));
}

extern "C" int dlio_bn_finalize(const double* sum, const double* sumsq, int C, double count,
                                const float* gamma, float eps, float momentum,
                                float* running_mean, float* running_var, float* mean,
                                float* invstd, float* scale, dlio_stream_t stream) {
  if (!sum || !sumsq || !mean || !invstd || !scale || C <= 0 || count <= 0) return DLIO_EINVAL;
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(C, 128)), dim3(128), 0, as_stream(stream), sum,
                     sumsq, C, count, gamma, eps, momentum, running_mean, running_var, mean,
                     invstd, scale);
  return dlio_check_launch();
}

extern "C" int dlio_bn_eval_params(const float* running_mean, const float* running_var,
                                   const float* gamma, float eps, int C, float* mean,
                                   float* invstd, float* scale, dlio_stream_t stream) {
  if (!running_mean || !running_var || !mean || !invstd || !scale || C <= 0) return DLIO_EINVAL;
  hipLaunchKernelGGL(bn_eval_params_kernel, dim3(cdiv(C, 128)), dim3(128), 0, as_stream(stream),
                     running_mean, running_var, gamma, eps, C, mean, invstd, scale);
  return dlio_check_launch();
}

extern "C" int dlio_bn_apply(const float* x, int x_ctot, int x_coff, const float* mean,
                             const float* scale, const float* beta, const float* residual,
                             int r_ctot, int r_coff, float* y, int y_ctot, int y_coff, int N,
                             int C, int HW, int pre_relu, int post_relu, dlio_stream_t stream) {
  if (!x || !mean || !scale || !y || N <= 0 || C <= 0 || HW <= 0) return DLIO_EINVAL;
  const bool vec = (HW & 3) == 0;
  const int64_t total = (int64_t)N * C * (vec ? HW / 4 : HW);
  dim3 grid(ew_grid(total, 256));
  if (vec)
    hipLaunchKernelGGL(bn_apply_kernel<true>, grid, dim3(256), 0, as_stream(stream), x, x_ctot,
                       x_coff, mean, scale, beta, residual, r_ctot, r_coff, y, y_ctot, y_coff, N, C,
                       HW, pre_relu, post_relu);
  else
    hipLaunchKernelGGL(bn_apply_kernel<false>, grid, dim3(256), 0, as_stream(stream), x, x_ctot,
                       x_coff, mean, scale, beta, residual, r_ctot, r_coff, y, y_ctot, y_coff, N, C,
                       HW, pre_relu, post_relu);
  return dlio_check_launch();
}

extern "C" int dlio_bn_bwd_apply(const float* dy, int dy_ctot, int dy_coff, const float* x,
                                 int x_ctot, int x_coff, const float* mean, const float* invstd,
                                 const float* scale, const float* beta, const double* sum_g,
                                 const double* sum_gx, float* dx, int dx_ctot, int dx_coff,
                                 float* dgamma, float* dbeta, int N, int C, int HW, int pre_relu,
                                 int post_relu, int use_batch_stats, dlio_stream_t stream) {
  if (!dy || !x || !mean || !invstd || !scale || !sum_g || !sum_gx || !dx) return DLIO_EINVAL;
  if (N <= 0 || C <= 0 || HW <= 0) return DLIO_EINVAL;
  hipStream_t s = as_stream(stream);
  const bool vec = (HW & 3) == 0;
  const int64_t total = (int64_t)N * C * (vec ? HW / 4 : HW);
  dim3 grid(ew_grid(total, 256));
  if (vec)
    hipLaunchKernelGGL(bn_bwd_apply_kernel<true>, grid, dim3(256), 0, s, dy, dy_ctot, dy_coff, x,
                       x_ctot, x_coff, mean, invstd, scale, beta, sum_g, sum_gx, dx, dx_ctot,
                       dx_coff, N, C, HW, pre_relu, post_relu, use_batch_stats);
  else
    hipLaunchKernelGGL(bn_bwd_apply_kernel<false>, grid, dim3(256), 0, s, dy, dy_ctot, dy_coff, x,
                       x_ctot, x_coff, mean, invstd, scale, beta, sum_g, sum_gx, dx, dx_ctot,
                       dx_coff, N, C, HW, pre_relu, post_relu, use_batch_stats);
  int rc = dlio_check_launch();
  if (rc) return rc;
  if (dgamma || dbeta) {
    hipLaunchKernelGGL(bn_param_grads_kernel, dim3(cdiv(C, 128)), dim3(128), 0, s, sum_g, sum_gx,
                       dgamma, dbeta, C);
    rc = dlio_check_launch();
  }
  return rc;
}

// ---- statistics + apply / backward in two launches (reduce, plane kernel) --------------------
extern "C" int dlio_bn_train_apply(const float* x, int N, int x_ctot, int x_coff, int C, int HW,
                                   int pre_relu, int post_relu, const float* gamma, const float* beta,
                                   float eps, float momentum, float* running_mean, float* running_var,
                                   float* mean, float* invstd, float* scale, const float* residual,
                                   int r_ctot, int r_coff, float* y, int y_ctot, int y_coff,
                                   float* gap_out, int gap_ctot, int gap_coff, void* ws,
                                   size_t ws_bytes, int phase, double count_scale,
                                   const float* r_mean, const float* r_scale, const float* r_shift,
                                   float* amax_out, dlio_stream_t stream) {
  if (!x || !y || !mean || !invstd || !scale || N <= 0 || C <= 0 || HW <= 0 || !ws || phase < 0 ||
      phase > 2 || !(count_scale >= 1.0))
    return DLIO_EINVAL;
  const int splits = pick_splits(N, C, HW);
  if (ws_bytes < (size_t)C * splits * 2 * sizeof(double)) return DLIO_EWS;
  hipStream_t s = as_stream(stream);
  double* part = reinterpret_cast<double*>(ws);
  int rc = DLIO_OK;
  const double tensor_bytes = 4.0 * N * (double)C * HW;
  if (phase != 2) {
    DlioProfScope prof(6, s, 0.0, tensor_bytes);
    if (bn_unroll() > 1)
      hipLaunchKernelGGL((chan_reduce_kernel<0, 4>), dim3((unsigned)(C * splits)), dim3(RB), 0, s, x, x_ctot, x_coff,
                       (const float*)nullptr, 0, 0, (const float*)nullptr, (const float*)nullptr,
                       (const float*)nullptr, (const float*)nullptr, N, C, HW, pre_relu, 0, splits, part);
    else
      hipLaunchKernelGGL(chan_reduce_kernel<0>, dim3((unsigned)(C * splits)), dim3(RB), 0, s, x, x_ctot, x_coff,
                       (const float*)nullptr, 0, 0, (const float*)nullptr, (const float*)nullptr,
                       (const float*)nullptr, (const float*)nullptr, N, C, HW, pre_relu, 0, splits, part);
    rc = dlio_check_launch();
    if (rc || phase == 1) return rc;
  }
  const bool vec = (HW & 3) == 0;
  int chunks, chunk_len;
  plane_chunks(N * C, vec ? HW / 4 : HW, gap_out != nullptr, chunks, chunk_len);
  const dim3 grid((unsigned)(N * C * chunks));
  DlioProfScope prof(7, s, 0.0, tensor_bytes * (residual ? 3.0 : 2.0));
  if (vec && bn_unroll() > 1)
    hipLaunchKernelGGL((bn_plane_apply_kernel<true, 4>), grid, dim3(256), 0, s, x, x_ctot, x_coff, part, splits,
                       (double)N * HW * count_scale, gamma, beta, eps, momentum, running_mean, running_var, mean, invstd,
                       scale, residual, r_ctot, r_coff, y, y_ctot, y_coff, N, C, HW, pre_relu, post_relu,
                       chunks, chunk_len, gap_out, gap_ctot, gap_coff, r_mean, r_scale, r_shift, amax_out);
  else if (vec)
    hipLaunchKernelGGL(bn_plane_apply_kernel<true>, grid, dim3(256), 0, s, x, x_ctot, x_coff, part, splits,
                       (double)N * HW * count_scale, gamma, beta, eps, momentum, running_mean, running_var, mean, invstd,
                       scale, residual, r_ctot, r_coff, y, y_ctot, y_coff, N, C, HW, pre_relu, post_relu,
                       chunks, chunk_len, gap_out, gap_ctot, gap_coff, r_mean, r_scale, r_shift, amax_out);
  else
    hipLaunchKernelGGL(bn_plane_apply_kernel<false>, grid, dim3(256), 0, s, x, x_ctot, x_coff, part, splits,
                       (double)N * HW * count_scale, gamma, beta, eps, momentum, running_mean, running_var, mean, invstd,
                       scale, residual, r_ctot, r_coff, y, y_ctot, y_coff, N, C, HW, pre_relu, post_relu,
                       chunks, chunk_len, gap_out, gap_ctot, gap_coff, r_mean, r_scale, r_shift, amax_out);
  return dlio_check_launch();
}

extern "C" int dlio_bn_bwd(const float* dy, int dy_ctot, int dy_coff, const float* x, int x_ctot,
                           int x_coff, const float* mean, const float* invstd, const float* scale,
                           const float* beta, float* dx, int dx_ctot, int dx_coff, float* dgamma,
                           float* dbeta, int accumulate, int N, int C, int HW, int pre_relu,
                           int post_relu, int use_batch_stats, void* ws, size_t ws_bytes, int phase,
                           double count_scale, const void* local_ws, float* amax_out, dlio_stream_t stream) {
  if (!dy || !x || !mean || !invstd || !scale || !dx || N <= 0 || C <= 0 || HW <= 0 || !ws ||
      phase < 0 || phase > 2 || !(count_scale >= 1.0))
    return DLIO_EINVAL;
  const int splits = pick_splits(N, C, HW);
  if (ws_bytes < (size_t)C * splits * 2 * sizeof(double)) return DLIO_EWS;
  hipStream_t s = as_stream(stream);
  double* part = reinterpret_cast<double*>(ws);
  int rc = DLIO_OK;
  const double tensor_bytes = 4.0 * N * (double)C * HW;
  if (phase != 2) {
    DlioProfScope prof(8, s, 0.0, 2.0 * tensor_bytes);
    if (bn_unroll() > 1)
      hipLaunchKernelGGL((chan_reduce_kernel<1, 2>), dim3((unsigned)(C * splits)), dim3(RB), 0, s, dy, dy_ctot,
                       dy_coff, x, x_ctot, x_coff, mean, invstd, scale, beta, N, C, HW, pre_relu, post_relu,
                       splits, part);
    else
      hipLaunchKernelGGL(chan_reduce_kernel<1>, dim3((unsigned)(C * splits)), dim3(RB), 0, s, dy, dy_ctot,
                       dy_coff, x, x_ctot, x_coff, mean, invstd, scale, beta, N, C, HW, pre_relu, post_relu,
                       splits, part);
    rc = dlio_check_launch();
    if (rc || phase == 1) return rc;
  }
  const bool vec = (HW & 3) == 0;
  int chunks, chunk_len;
  plane_chunks(N * C, vec ? HW / 4 : HW, false, chunks, chunk_len);
  const dim3 grid((unsigned)(N * C * chunks));
  const double inv_cnt = 1.0 / ((double)N * HW * count_scale);
  const double* lpart = reinterpret_cast<const double*>(local_ws);
  DlioProfScope prof(9, s, 0.0, 3.0 * tensor_bytes);
  if (vec && bn_unroll() > 1)
    hipLaunchKernelGGL((bn_plane_bwd_kernel<true, 2>), grid, dim3(256), 0, s, dy, dy_ctot, dy_coff, x, x_ctot,
                       x_coff, mean, invstd, scale, beta, part, lpart, inv_cnt, splits, dx, dx_ctot, dx_coff,
                       dgamma, dbeta, accumulate, N, C, HW, pre_relu, post_relu, use_batch_stats, chunks,
                       chunk_len, amax_out);
  else if (vec)
    hipLaunchKernelGGL(bn_plane_bwd_kernel<true>, grid, dim3(256), 0, s, dy, dy_ctot, dy_coff, x, x_ctot,
                       x_coff, mean, invstd, scale, beta, part, lpart, inv_cnt, splits, dx, dx_ctot, dx_coff,
                       dgamma, dbeta, accumulate, N, C, HW, pre_relu, post_relu, use_batch_stats, chunks,
                       chunk_len, amax_out);
  else
    hipLaunchKernelGGL(bn_plane_bwd_kernel<false>, grid, dim3(256), 0, s, dy, dy_ctot, dy_coff, x, x_ctot,
                       x_coff, mean, invstd, scale, beta, part, lpart, inv_cnt, splits, dx, dx_ctot, dx_coff,
                       dgamma, dbeta, accumulate, N, C, HW, pre_relu, post_relu, use_batch_stats, chunks,
                       chunk_len, amax_out);
  return dlio_check_launch();
}
