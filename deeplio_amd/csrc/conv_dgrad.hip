// Data gradient of STRIDED convolutions (stride-1 convs go through conv_fwd.hip with
// tap-reversed weights).  Production route: dlio_zero_upsample2d inserts the stride's zeros into
// dY and the stride-1 MFMA kernel of conv_fwd.hip runs on that (SH*SW x the minimal MFMA work,
// but ~100 TF/s instead of a scalar gather; FlowNet step 900 -> see DESIGN).  The gather kernel
// below is kept as the shape-agnostic reference implementation behind the C-ABI.  Only the non-PointSeg encoders have strided convs below the stem
// (FlowNet conv2..conv6, ResNet stage heads + 1x1 downsamples); this is a direct gather
// kernel: one thread per dx element, coalesced along W.
// Replaces the input-gradient half of nn.Conv2d backward at base_net.py:55-71, resnet.py
// (_make_layer strides) for those layers.
#include "common.h"

namespace {
__global__ __launch_bounds__(256) void dgrad_strided_kernel(const float* __restrict__ dy,
                                                            const float* __restrict__ w,
                                                            float* __restrict__ dx,
                                                            DlioConvDesc d) {
  const int64_t total = (int64_t)d.N * d.Cin * d.H * d.W;
  const int taps = d.KH * d.KW;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int iw = (int)(i % d.W);
    int64_t t = i / d.W;
    const int ih = (int)(t % d.H); t /= d.H;
    const int ci = (int)(t % d.Cin);
    const int n = (int)(t / d.Cin);
    float acc = 0.f;
    for (int ky = 0; ky < d.KH; ++ky) {
      const int a = ih + d.PH - ky;
      if (a < 0 || a % d.SH) continue;
      const int oh = a / d.SH;
      if (oh >= d.OH) continue;
      for (int kx = 0; kx < d.KW; ++kx) {
        const int b = iw + d.PW - kx;
        if (b < 0 || b % d.SW) continue;
        const int ow = b / d.SW;
        if (ow >= d.OW) continue;
        const float* dyp = dy + (((size_t)n * d.out_ctot + d.out_coff) * d.OH + oh) * d.OW + ow;
        const float* wp = w + ((size_t)ci * taps) + ky * d.KW + kx;
        const size_t ohw = (size_t)d.OH * d.OW;
        const size_t wstride = (size_t)d.Cin * taps;
        for (int co = 0; co < d.Cout; ++co) acc += dyp[co * ohw] * wp[co * wstride];
      }
    }
    dx[(((size_t)n * d.in_ctot + d.in_coff + ci) * d.H + ih) * d.W + iw] = acc;
  }
}
// dst[pl][ih][iw] = (ih % SH == 0 && iw % SW == 0 && in range) ? src[pl][ih/SH][iw/SW] : 0
__global__ void zero_upsample_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                     int64_t planes, int OH, int OW, int HU, int WU, int SH, int SW) {
  const int64_t total = planes * HU * WU;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int iw = (int)(i % WU);
    int64_t t = i / WU;
    const int ih = (int)(t % HU);
    const int64_t pl = t / HU;
    float v = 0.f;
    if (ih % SH == 0 && iw % SW == 0) {
      const int oh = ih / SH, ow = iw / SW;
      if (oh < OH && ow < OW) v = src[(pl * OH + oh) * OW + ow];
    }
    dst[i] = v;
  }
}
// dx[n][c][ih][iw] = phase[(ih % SH) * SW + iw % SW][n][c][ih / SH][iw / SW] (+ residual): the phases
// of a strided data gradient, each computed as a stride-1 convolution of dY with the taps of
// that phase (functional.conv_dgrad), woven back into the input grid.  A null phase pointer is
// a phase without taps (all zeros).  One thread per output element, coalesced along W.
struct PhasePtrs { const float* p[4]; };
__global__ __launch_bounds__(256) void phase_interleave_kernel(PhasePtrs ph, const float* __restrict__ residual,
                                                               float* __restrict__ dx, int N, int C, int H, int W,
                                                               int SH, int SW, int dx_ctot, int dx_coff,
                                                               int r_ctot, int r_coff) {
  const int64_t total = (int64_t)N * C * H * W;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int iw = (int)(i % W);
    int64_t t = i / W;
    const int ih = (int)(t % H); t /= H;
    const int c = (int)(t % C);
    const int n = (int)(t / C);
    const int a = ih % SH, b = iw % SW;
    const float* src = ph.p[a * SW + b];
    const int Hp = (H - a + SH - 1) / SH, Wp = (W - b + SW - 1) / SW;
    float v = src ? src[(((size_t)n * C + c) * Hp + ih / SH) * Wp + iw / SW] : 0.f;
    if (residual) v += residual[(((size_t)n * r_ctot + r_coff + c) * H + ih) * W + iw];
    dx[(((size_t)n * dx_ctot + dx_coff + c) * H + ih) * W + iw] = v;
  }
}
}  // namespace

extern "C" int dlio_phase_interleave2d(const float* const* phases, int SH, int SW, const float* residual,
                                       int r_ctot, int r_coff, float* dx, int dx_ctot, int dx_coff, int N,
                                       int C, int H, int W, dlio_stream_t stream) {
  if (!phases || !dx || N <= 0 || C <= 0 || H <= 0 || W <= 0 || SH < 1 || SW < 1 || SH * SW > 4 ||
      dx_ctot < dx_coff + C || (residual && r_ctot < r_coff + C))
    return DLIO_EINVAL;
  PhasePtrs ph;
  for (int i = 0; i < 4; ++i) ph.p[i] = i < SH * SW ? phases[i] : nullptr;
  hipLaunchKernelGGL(phase_interleave_kernel, dim3(ew_grid((int64_t)N * C * H * W, 256)), dim3(256), 0,
                     as_stream(stream), ph, residual, dx, N, C, H, W, SH, SW, dx_ctot, dx_coff, r_ctot, r_coff);
  return dlio_check_launch();
}

extern "C" int dlio_zero_upsample2d(const float* src, float* dst, int64_t planes, int OH, int OW,
                                    int HU, int WU, int SH, int SW, dlio_stream_t stream) {
  if (!src || !dst || planes <= 0 || OH <= 0 || OW <= 0 || SH <= 0 || SW <= 0 ||
      HU < (OH - 1) * SH + 1 || WU < (OW - 1) * SW + 1)
    return DLIO_EINVAL;
  hipLaunchKernelGGL(zero_upsample_kernel, dim3(ew_grid(planes * HU * WU, 256)), dim3(256), 0,
                     as_stream(stream), src, dst, planes, OH, OW, HU, WU, SH, SW);
  return dlio_check_launch();
}

extern "C" int dlio_conv2d_dgrad_strided(const float* dy, const float* w, float* dx,
                                         const DlioConvDesc* dp, dlio_stream_t stream) {
  if (!dy || !w || !dx || !dp) return DLIO_EINVAL;
  const DlioConvDesc& d = *dp;
  const int64_t total = (int64_t)d.N * d.Cin * d.H * d.W;
  if (total <= 0) return DLIO_EINVAL;
  hipLaunchKernelGGL(dgrad_strided_kernel, dim3(ew_grid(total, 256)), dim3(256), 0,
                     as_stream(stream), dy, w, dx, d);
  return dlio_check_launch();
}
