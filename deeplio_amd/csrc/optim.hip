// Optimizer step over ONE flat fp32 parameter buffer (41-65 M parameters on this path, 87 %
// of them the odometry LSTM): a single HBM-bound float4 sweep, 28 B per parameter for Adam
// (read p,g,m,v; write p,m,v), instead of ~600 per-tensor launches.
//
// Replaces torch.optim.Adam / SGD / RMSprop / Adadelta as built by create_optimizer
// (optimizer.py:4-16; weight decay is L2 added to the gradient) and calc_grad_norm (trainer.py:481-486).
#include "common.h"
#include <math.h>

namespace {

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v,
                                                   int64_t n, float lr_over_bc1, float beta1,
                                                   float beta2, float eps, float wd,
                                                   float inv_sqrt_bc2, float gscale) {
  const int64_t n4 = n >> 2;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4;
       i += (int64_t)gridDim.x * blockDim.x) {
    float4 pv = reinterpret_cast<float4*>(p)[i];
    const float4 gv = reinterpret_cast<const float4*>(g)[i];
    float4 mv = reinterpret_cast<float4*>(m)[i];
    float4 vv = reinterpret_cast<float4*>(v)[i];
    float pe[4] = {pv.x, pv.y, pv.z, pv.w}, ge[4] = {gv.x, gv.y, gv.z, gv.w};
    float me[4] = {mv.x, mv.y, mv.z, mv.w}, ve[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float gg = ge[k] * gscale + wd * pe[k];
      me[k] = beta1 * me[k] + (1.f - beta1) * gg;
      ve[k] = beta2 * ve[k] + (1.f - beta2) * gg * gg;
      const float denom = sqrtf(ve[k]) * inv_sqrt_bc2 + eps;
      pe[k] -= lr_over_bc1 * (me[k] / denom);
    }
    reinterpret_cast<float4*>(p)[i] = make_float4(pe[0], pe[1], pe[2], pe[3]);
    reinterpret_cast<float4*>(m)[i] = make_float4(me[0], me[1], me[2], me[3]);
    reinterpret_cast<float4*>(v)[i] = make_float4(ve[0], ve[1], ve[2], ve[3]);
  }
  // tail
  const int64_t base = n4 << 2;
  const int64_t i = base + blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) {
    const float gg = g[i] * gscale + wd * p[i];
    const float mm = beta1 * m[i] + (1.f - beta1) * gg;
    const float vv = beta2 * v[i] + (1.f - beta2) * gg * gg;
    m[i] = mm; v[i] = vv;
    p[i] -= lr_over_bc1 * (mm / (sqrtf(vv) * inv_sqrt_bc2 + eps));
  }
}

__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                  float* __restrict__ buf, int64_t n, float lr,
                                                  float momentum, float wd, int first,
                                                  float gscale) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    float gg = g[i] * gscale + wd * p[i];
    if (momentum != 0.f) {
      const float b = first ? gg : momentum * buf[i] + gg;
      buf[i] = b;
      gg = b;
    }
    p[i] -= lr * gg;
  }
}

// torch.optim.RMSprop (alpha, eps, optional momentum / centered): one sweep, 20-28 B per parameter
__global__ __launch_bounds__(256) void rmsprop_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                      float* __restrict__ sq, float* __restrict__ buf,
                                                      float* __restrict__ gavg, int64_t n, float lr,
                                                      float alpha, float eps, float wd, float momentum,
                                                      float gscale) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const float gg = g[i] * gscale + wd * p[i];
    const float s = alpha * sq[i] + (1.f - alpha) * gg * gg;
    sq[i] = s;
    float avg;
    if (gavg) {
      const float ga = alpha * gavg[i] + (1.f - alpha) * gg;     // lerp as torch: ga + (g - ga)(1 - alpha)
      gavg[i] = ga;
      avg = sqrtf(s - ga * ga) + eps;
    } else {
      avg = sqrtf(s) + eps;
    }
    if (buf) {
      const float b = momentum * buf[i] + gg / avg;
      buf[i] = b;
      p[i] -= lr * b;
    } else {
      p[i] -= lr * (gg / avg);
    }
  }
}

// torch.optim.Adadelta (rho, eps): 28 B per parameter
__global__ __launch_bounds__(256) void adadelta_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                       float* __restrict__ sq, float* __restrict__ acc,
                                                       int64_t n, float lr, float rho, float eps, float wd,
                                                       float gscale) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const float gg = g[i] * gscale + wd * p[i];
    const float s = rho * sq[i] + (1.f - rho) * gg * gg;
    sq[i] = s;
    const float a = acc[i];
    const float delta = sqrtf(a + eps) / sqrtf(s + eps) * gg;
    acc[i] = rho * a + (1.f - rho) * delta * delta;
    p[i] -= lr * delta;
  }
}

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, int64_t n,
                                                    double* out) {
  __shared__ double sm[16];
  double s = 0.0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    s += (double)g[i] * g[i];
  const double r = block_sum_d(s, sm);
  if (threadIdx.x == 0) atomicAdd(out, r);
}

// workgroups an optimizer sweep may use (0: the default of every memory-bound sweep, 8 per CU).  A sweep issued from inside the
// backward pass (FlatOptimizer.step_early: the tail bucket's update under the encoder backward) is background work: at the full
// grid its 2048 resident workgroups starve the 3-30 us launches of the critical chain for the sweep's whole length.
int g_max_blocks = 0;
inline int opt_grid(int64_t items) {
  int g = ew_grid(items, 256);
  if (g_max_blocks > 0 && g > g_max_blocks) g = g_max_blocks;
  return g;
}

}  // namespace

extern "C" int dlio_optim_set_max_blocks(int blocks) {
  if (blocks < 0) return DLIO_EINVAL;
  g_max_blocks = blocks;
  return DLIO_OK;
}

extern "C" int dlio_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr,
                              float beta1, float beta2, float eps, float weight_decay, int step,
                              float grad_scale, dlio_stream_t stream) {
  if (!p || !g || !m || !v || n <= 0 || step < 1) return DLIO_EINVAL;
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  const float lr_over_bc1 = (float)((double)lr / bc1);
  const float inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
  int grid = opt_grid(cdiv64(n, 4));
  hipLaunchKernelGGL(adam_kernel, dim3(grid), dim3(256), 0, as_stream(stream), p, g, m, v, n,
                     lr_over_bc1, beta1, beta2, eps, weight_decay, inv_sqrt_bc2, grad_scale);
  return dlio_check_launch();
}

extern "C" int dlio_sgd_step(float* p, const float* g, float* buf, int64_t n, float lr,
                             float momentum, float weight_decay, int step, float grad_scale,
                             dlio_stream_t stream) {
  if (!p || !g || n <= 0 || step < 1 || (momentum != 0.f && !buf)) return DLIO_EINVAL;
  hipLaunchKernelGGL(sgd_kernel, dim3(opt_grid(n)), dim3(256), 0, as_stream(stream), p, g, buf,
                     n, lr, momentum, weight_decay, step == 1 ? 1 : 0, grad_scale);
  return dlio_check_launch();
}

extern "C" int dlio_rmsprop_step(float* p, const float* g, float* square_avg, float* momentum_buf,
                                 float* grad_avg, int64_t n, float lr, float alpha, float eps,
                                 float weight_decay, float momentum, float grad_scale,
                                 dlio_stream_t stream) {
  if (!p || !g || !square_avg || n <= 0 || (momentum != 0.f && !momentum_buf)) return DLIO_EINVAL;
  hipLaunchKernelGGL(rmsprop_kernel, dim3(opt_grid(n)), dim3(256), 0, as_stream(stream), p, g,
                     square_avg, momentum != 0.f ? momentum_buf : nullptr, grad_avg, n, lr, alpha, eps,
                     weight_decay, momentum, grad_scale);
  return dlio_check_launch();
}

extern "C" int dlio_adadelta_step(float* p, const float* g, float* square_avg, float* acc_delta,
                                  int64_t n, float lr, float rho, float eps, float weight_decay,
                                  float grad_scale, dlio_stream_t stream) {
  if (!p || !g || !square_avg || !acc_delta || n <= 0) return DLIO_EINVAL;
  hipLaunchKernelGGL(adadelta_kernel, dim3(opt_grid(n)), dim3(256), 0, as_stream(stream), p, g,
                     square_avg, acc_delta, n, lr, rho, eps, weight_decay, grad_scale);
  return dlio_check_launch();
}

extern "C" int dlio_sumsq(const float* g, int64_t n, double* out, dlio_stream_t stream) {
  if (!g || !out || n <= 0) return DLIO_EINVAL;
  hipStream_t s = as_stream(stream);
  if (hipMemsetAsync(out, 0, sizeof(double), s) != hipSuccess) return DLIO_ELAUNCH;
  hipLaunchKernelGGL(sumsq_kernel, dim3(ew_grid(n, 256)), dim3(256), 0, s, g, n, out);
  return dlio_check_launch();
}
