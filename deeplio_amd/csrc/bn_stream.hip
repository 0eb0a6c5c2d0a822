// Train-mode BatchNorm2d + ReLU (+ bypass residual) of a Fire block's concat buffer as a plain STREAMING apply, for blocks
// whose statistics are already known when the pass starts: dlio_fire_expand_fwd_stats leaves (mean, scale, shift) from the
// expand launch's epilogue (pointseg_modules.py:100-106,126-133), so nothing has to be exchanged between workgroups and no
// launch spins for partners (round 4 ran bn_coop_fwd_kernel here: 150-343 us per launch on 104 of 256 CUs).
//
//   bn_aff_apply_kernel     y = max(0, (x - mean) * scale + shift) + r   (r: the bypass input, itself apply-on-load when it is
//                           a deferred block's raw output), optional plane averages of y (one workgroup per plane: fixed order)
//   bn_aff_pool_kernel<SH>  the same followed by the SELayer's MaxPool2d(3, stride (SH, 2), padding 1) of pointseg_net.py:27-46
//                           WITHOUT writing y: with s = sigmoid(...) > 0 per plane, maxpool(s * y) = s * maxpool(y) and the
//                           arg-max is the same, so the pool can run before the SELayer's scale exists.  Outputs: the pooled
//                           maximum of y (unscaled), the arg-max map, the plane averages of y (what the SELayer squeezes).
//                           The full-resolution block output is never written nor read back: per block 2 passes over the
//                           largest tensors of the encoder less (blk1: 268 MB each), the scale is one pass over the pooled
//                           tensor (dlio_chan_scale_fwd).  Backward needs neither y nor the scaled tensor at full
//                           resolution (dlio_bn_coop_bwd_pool routes the pooled gradient through the arg-max map).
// Tie rule and scan order are maxpool3_fwd_strip's (ky major, kx minor, first strictly greater wins, NaN wins): values and
// arg-max codes are bit-identical to dlio_maxpool2d_fwd over the materialised y.
#include "common.h"

namespace {

struct AffRows {              // per-channel rows of the transforms (device pointers)
  const float* mean; const float* scale; const float* shift;            // of x (required)
  const float* r_mean; const float* r_scale; const float* r_shift;      // of the residual (null: residual taken as stored)
};

__device__ __forceinline__ float act1(float v, float mu, float sc, float sh) { return fmaxf((v - mu) * sc + sh, 0.f); }

template <int UN>
__global__ __launch_bounds__(256) void bn_aff_apply_kernel(
    const float* __restrict__ x, int x_ctot, int x_coff, AffRows a, const float* __restrict__ residual, int r_ctot, int r_coff,
    float* __restrict__ y, int y_ctot, int y_coff, int C, int HW4, int chunks, int chunk_len, float* __restrict__ gap_out,
    int gap_ctot, int gap_coff) {
  __shared__ double sm[16];
  const int chunk = blockIdx.x % chunks, pl = blockIdx.x / chunks, n = pl / C, c = pl - n * C;
  const float mu = a.mean[c], sc = a.scale[c], sh = a.shift[c];
  const bool raff = residual && a.r_scale;
  const float rmu = raff ? a.r_mean[r_coff + c] : 0.f, rsc = raff ? a.r_scale[r_coff + c] : 1.f, rsh = raff ? a.r_shift[r_coff + c] : 0.f;
  const float4* xp = reinterpret_cast<const float4*>(x + ((size_t)n * x_ctot + x_coff + c) * HW4 * 4);
  const float4* rp = residual ? reinterpret_cast<const float4*>(residual + ((size_t)n * r_ctot + r_coff + c) * HW4 * 4) : nullptr;
  float4* yp = reinterpret_cast<float4*>(y + ((size_t)n * y_ctot + y_coff + c) * HW4 * 4);
  const int i1 = min(HW4, (chunk + 1) * chunk_len);
  double gs = 0.0;
  for (int i = chunk * chunk_len + threadIdx.x; i < i1; i += UN * 256) {
    float4 v[UN], rv[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int k = min(i + u * 256, i1 - 1);          // (clamped: every load unconditional)
      v[u] = xp[k];
      rv[u] = rp ? rp[k] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      if (i + u * 256 >= i1) continue;
      float e[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
      float re[4] = {rv[u].x, rv[u].y, rv[u].z, rv[u].w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float o = act1(e[k], mu, sc, sh);
        if (raff) re[k] = act1(re[k], rmu, rsc, rsh);
        e[k] = o + re[k];
      }
      st4<16>(reinterpret_cast<float*>(yp + i + u * 256), make_float4(e[0], e[1], e[2], e[3]));
      gs += (double)((e[0] + e[1]) + (e[2] + e[3]));
    }
  }
  if (gap_out) {                                       // (chunks == 1 then)
    const double r = block_sum_d(gs, sm);
    if (threadIdx.x == 0) gap_out[(size_t)n * gap_ctot + gap_coff + c] = (float)(r / (4.0 * HW4));
  }
}

// one workgroup per (n, c) plane; a work item = 2 output columns x FR output rows (input columns 4 b - 1 .. 4 b + 3)
template <int SH>
__global__ __launch_bounds__(256) void bn_aff_pool_kernel(
    const float* __restrict__ x, int x_ctot, int x_coff, AffRows a, const float* __restrict__ residual, int r_ctot, int r_coff,
    float* __restrict__ yp_out, uint8_t* __restrict__ idx, float* __restrict__ gap_out, int gap_ctot, int gap_coff, int C,
    int H, int W, int OH, int OW) {
  constexpr int FR = SH == 1 ? 4 : 2;              // output rows per item
  constexpr int NIN = (FR - 1) * SH + 3;           // input rows they touch (6 / 5); rows j = 1 .. 4 are owned (plane sum)
  __shared__ double sm[16];
  const int pl = blockIdx.x, n = pl / C, c = pl - n * C;
  const float mu = a.mean[c], sc = a.scale[c], sh = a.shift[c];
  const bool raff = residual && a.r_scale;
  const float rmu = raff ? a.r_mean[r_coff + c] : 0.f, rsc = raff ? a.r_scale[r_coff + c] : 1.f, rsh = raff ? a.r_shift[r_coff + c] : 0.f;
  const float* xp = x + ((size_t)n * x_ctot + x_coff + c) * H * W;
  const float* rp = residual ? residual + ((size_t)n * r_ctot + r_coff + c) * H * W : nullptr;
  const int OW2 = OW >> 1, strips = (OH + FR - 1) / FR, items = strips * OW2;
  double gs = 0.0;
  for (int i = threadIdx.x; i < items; i += 256) {
    const int b = i % OW2, oh0 = (i / OW2) * FR;
    float4 v4[NIN], r4[NIN];
    float hl[NIN], rl[NIN];
    bool rv[NIN];
#pragma unroll
    for (int j = 0; j < NIN; ++j) {                 // all loads first, unconditional (row clamped, halo address clamped)
      const int ih = oh0 * SH - 1 + j;
      rv[j] = ih >= 0 && ih < H;
      const size_t ro = (size_t)min(max(ih, 0), H - 1) * W + 4 * b;
      v4[j] = *reinterpret_cast<const float4*>(xp + ro);
      hl[j] = xp[ro - (b > 0 ? 1 : 0)];
      if (rp) { r4[j] = *reinterpret_cast<const float4*>(rp + ro); rl[j] = rp[ro - (b > 0 ? 1 : 0)]; }
    }
    float rb[NIN][2];
    int rk[NIN][2];
#pragma unroll
    for (int j = 0; j < NIN; ++j) {
      rb[j][0] = rb[j][1] = 0.f; rk[j][0] = rk[j][1] = 0;
      if (!rv[j]) continue;
      float v[5] = {hl[j], v4[j].x, v4[j].y, v4[j].z, v4[j].w};
#pragma unroll
      for (int k = 0; k < 5; ++k) v[k] = act1(v[k], mu, sc, sh);
      if (rp) {
        float r[5] = {rl[j], r4[j].x, r4[j].y, r4[j].z, r4[j].w};
#pragma unroll
        for (int k = 0; k < 5; ++k) v[k] += raff ? act1(r[k], rmu, rsc, rsh) : r[k];
      }
      // plane sum: the rows this item owns (input rows oh0 SH .. oh0 SH + 3 = j 1 .. 4), its own four columns
      if (j >= 1 && j <= 4 && oh0 * SH - 1 + j < (oh0 + FR) * SH) gs += (double)((v[1] + v[2]) + (v[3] + v[4]));
#pragma unroll
      for (int o = 0; o < 2; ++o) {
        bool first = true;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int cc = 2 * o + kx;
          if (cc == 0 && b == 0) continue;          // left padding
          const float val = v[cc];
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
      *reinterpret_cast<float2*>(yp_out + oo) = make_float2(best[0], best[1]);
      *reinterpret_cast<unsigned short*>(idx + oo) = (unsigned short)(bi[0] | (bi[1] << 8));
    }
  }
  if (gap_out) {
    const double r = block_sum_d(gs, sm);
    if (threadIdx.x == 0) gap_out[(size_t)n * gap_ctot + gap_coff + c] = (float)(r / ((double)H * W));
  }
}

}  // namespace

extern "C" int dlio_bn_aff_apply(const float* x, int N, int x_ctot, int x_coff, int C, int HW, const float* mean,
                                 const float* scale, const float* shift, const float* residual, int r_ctot, int r_coff,
                                 const float* r_mean, const float* r_scale, const float* r_shift, float* y, int y_ctot,
                                 int y_coff, float* gap_out, int gap_ctot, int gap_coff, dlio_stream_t stream) {
  if (!x || !y || !mean || !scale || !shift || N <= 0 || C <= 0 || HW <= 0) return DLIO_EINVAL;
  if ((HW & 3) || ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(residual)) & 15))
    return DLIO_EUNSUP;
  if (r_scale && (!r_mean || !r_shift)) return DLIO_EINVAL;
  hipStream_t s = as_stream(stream);
  const int HW4 = HW >> 2;
  // one workgroup per plane when the plane averages are wanted (fixed summation order) or the planes alone fill the chip
  int chunks = 1;
  if (!gap_out) {
    const int64_t planes = (int64_t)N * C;
    while (planes * chunks < 2048 && HW4 / (chunks * 2) >= 1024) chunks *= 2;
  }
  const int chunk_len = ((HW4 + chunks - 1) / chunks + 255) / 256 * 256;
  chunks = (HW4 + chunk_len - 1) / chunk_len;
  const int64_t blocks = (int64_t)N * C * chunks;
  if (blocks > 0x7fffffff) return DLIO_EINVAL;
  const AffRows a{mean, scale, shift, r_mean, r_scale, r_shift};
  DlioProfScope prof(7, s, 0.0, 4.0 * N * (double)C * HW * (residual ? 3.0 : 2.0));
  hipLaunchKernelGGL(bn_aff_apply_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, s, x, x_ctot, x_coff, a, residual, r_ctot, r_coff,
                     y, y_ctot, y_coff, C, HW4, chunks, chunk_len, gap_out, gap_ctot, gap_coff);
  return dlio_check_launch();
}

extern "C" int dlio_bn_aff_pool_ok(int H, int W, int SH) {
  return (SH == 1 || SH == 2) && H > 0 && W >= 8 && (W & 3) == 0 && (SH == 1 || (H & 1) == 0);
}

extern "C" int dlio_bn_aff_pool_fwd(const float* x, int N, int x_ctot, int x_coff, int C, int H, int W, int SH,
                                    const float* mean, const float* scale, const float* shift, const float* residual,
                                    int r_ctot, int r_coff, const float* r_mean, const float* r_scale, const float* r_shift,
                                    float* y_pooled, unsigned char* idx, float* gap_out, int gap_ctot, int gap_coff,
                                    dlio_stream_t stream) {
  if (!x || !y_pooled || !idx || !mean || !scale || !shift || N <= 0 || C <= 0) return DLIO_EINVAL;
  if (!dlio_bn_aff_pool_ok(H, W, SH)) return DLIO_EUNSUP;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(residual)) & 15) return DLIO_EUNSUP;
  if ((reinterpret_cast<uintptr_t>(y_pooled) & 7) || (reinterpret_cast<uintptr_t>(idx) & 1)) return DLIO_EUNSUP;
  if (r_scale && (!r_mean || !r_shift)) return DLIO_EINVAL;
  const int OH = (H + 2 - 3) / SH + 1, OW = W / 2;
  const int64_t planes = (int64_t)N * C;
  if (planes > 0x7fffffff) return DLIO_EINVAL;
  hipStream_t s = as_stream(stream);
  const AffRows a{mean, scale, shift, r_mean, r_scale, r_shift};
  DlioProfScope prof(7, s, 0.0, (double)planes * (4.0 * H * W * (residual ? 2.0 : 1.0) + 5.0 * OH * OW));
  if (SH == 1)
    hipLaunchKernelGGL(bn_aff_pool_kernel<1>, dim3((unsigned)planes), dim3(256), 0, s, x, x_ctot, x_coff, a, residual, r_ctot,
                       r_coff, y_pooled, idx, gap_out, gap_ctot, gap_coff, C, H, W, OH, OW);
  else
    hipLaunchKernelGGL(bn_aff_pool_kernel<2>, dim3((unsigned)planes), dim3(256), 0, s, x, x_ctot, x_coff, a, residual, r_ctot,
                       r_coff, y_pooled, idx, gap_out, gap_ctot, gap_coff, C, H, W, OH, OW);
  return dlio_check_launch();
}
