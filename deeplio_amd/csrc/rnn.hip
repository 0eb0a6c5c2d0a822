// LSTM / GRU sequence recurrences (one layer, one direction per call).
//
// Persistent path (H = 32/64/128): one 4H-thread workgroup per chunk of 1 / 2 / 8 batch rows (lstm_rows_per_wg) walks
// all T steps inside ONE launch.  Forward: thread j keeps row j of W_hh (H floats) in VGPRs
// for the whole sequence, h_{t-1}/c live in LDS and are read as wave-uniform ds_read_b128
// broadcasts, gx of step t+1 is prefetched into registers while step t computes.
// Backward: thread (q,k) keeps column k of gate block q of W_hh in VGPRs and produces the
// partial dh_{t-1}[k] of that block; four partials are summed in a fixed order through LDS.
// Streamed path (any H, e.g. the 1024-wide odometry LSTM whose W_hh is 16 MB): per step one
// weight-streaming dlio_linear_* call plus one pointwise cell kernel.
//
// Replaces nn.LSTM / nn.GRU at imu_feat_nets.py:63-70,79-83 and odom_feat_nets.py:61-68,80.
#include "common.h"
#include <stdlib.h>

namespace {

// batch rows per persistent LSTM / GRU workgroup (template parameter BCT): the recurrent product is VALU work (H FMAs per row and thread and step), so a
// small batch is spread over more workgroups -- with 8 rows in ONE workgroup the headline's IMU net (B = 8) ran every
// recurrence on a single CU at 5.9 us per step; one row per workgroup (8 CUs): same results bit for bit
static int lstm_rows_per_wg(int B) {
  static const int forced = 0;
  if (forced == 1 || forced == 2 || forced == 4 || forced == 8) return forced;
  return B <= 16 ? 1 : B <= 64 ? 2 : 8;   // (B = 8: IMU forward chain 2.16 / 1.35 / 1.24 / 0.92 ms at 8 / 4 / 2 / 1 rows)
}

__device__ __forceinline__ float sigm(float x) { return 1.0f / (1.0f + expf(-x)); }

// ------------------------------------------------------------------ LSTM forward
template <int H, int BCT>
__global__ __launch_bounds__(4 * H) void lstm_persist_fwd(
    const float* __restrict__ gx, const float* __restrict__ w_hh, const float* __restrict__ b_hh,
    const float* __restrict__ h0, const float* __restrict__ c0, float* __restrict__ hs, int ldhs,
    float* __restrict__ cs, float* __restrict__ hp, float* __restrict__ gates,
    float* __restrict__ hT, float* __restrict__ cT, int T, int B, int rst, int rsb, int reverse) {
  constexpr int G = 4 * H;
  __shared__ __attribute__((aligned(16))) float hl[BCT * H];
  __shared__ __attribute__((aligned(16))) float cl[BCT * H];
  __shared__ __attribute__((aligned(16))) float gl[BCT * G];
  const int j = threadIdx.x;
  const int b0 = blockIdx.x * BCT;
  const int nb = min(BCT, B - b0);

  float w[H];
#pragma unroll
  for (int k = 0; k < H; k += 4) {
    const float4 v = *reinterpret_cast<const float4*>(w_hh + (size_t)j * H + k);
    w[k] = v.x; w[k + 1] = v.y; w[k + 2] = v.z; w[k + 3] = v.w;
  }
  const float bj = b_hh ? b_hh[j] : 0.f;

  for (int e = j; e < BCT * H; e += G) {
    const int b = e / H, k = e - b * H;
    const bool live = b < nb;
    hl[e] = (live && h0) ? h0[(size_t)(b0 + b) * H + k] : 0.f;
    cl[e] = (live && c0) ? c0[(size_t)(b0 + b) * H + k] : 0.f;
  }

  float gcur[BCT], gnext[BCT];
  {
    const int t = reverse ? T - 1 : 0;
#pragma unroll
    for (int bb = 0; bb < BCT; ++bb)
      gcur[bb] = bb < nb ? gx[((size_t)t * rst + (size_t)(b0 + bb) * rsb) * G + j] : 0.f;
  }
  __syncthreads();

  for (int step = 0; step < T; ++step) {
    const int t = reverse ? T - 1 - step : step;
    if (step + 1 < T) {
      const int tn = reverse ? t - 1 : t + 1;
#pragma unroll
      for (int bb = 0; bb < BCT; ++bb)
        gnext[bb] = bb < nb ? gx[((size_t)tn * rst + (size_t)(b0 + bb) * rsb) * G + j] : 0.f;
    }
    // phase 1: gate pre-activations
    float acc[BCT];
#pragma unroll
    for (int bb = 0; bb < BCT; ++bb) acc[bb] = gcur[bb] + bj;
#pragma unroll
    for (int k = 0; k < H; k += 4) {
#pragma unroll
      for (int bb = 0; bb < BCT; ++bb) {
        const float4 hv = *reinterpret_cast<const float4*>(&hl[bb * H + k]);
        acc[bb] = fmaf(w[k], hv.x, acc[bb]);
        acc[bb] = fmaf(w[k + 1], hv.y, acc[bb]);
        acc[bb] = fmaf(w[k + 2], hv.z, acc[bb]);
        acc[bb] = fmaf(w[k + 3], hv.w, acc[bb]);
      }
    }
#pragma unroll
    for (int bb = 0; bb < BCT; ++bb) gl[bb * G + j] = acc[bb];
    __syncthreads();
    // phase 2: cell update
    for (int e = j; e < nb * H; e += G) {
      const int b = e / H, k = e - b * H;
      const size_t r = (size_t)t * rst + (size_t)(b0 + b) * rsb;
      const float ig = sigm(gl[b * G + k]);
      const float fg = sigm(gl[b * G + H + k]);
      const float gg = tanhf(gl[b * G + 2 * H + k]);
      const float og = sigm(gl[b * G + 3 * H + k]);
      const float cprev = cl[e], hprev = hl[e];
      const float c = fmaf(fg, cprev, ig * gg);
      const float h = og * tanhf(c);
      hp[r * H + k] = hprev;
      cs[r * H + k] = c;
      float* gr = gates + r * G;
      gr[k] = ig; gr[H + k] = fg; gr[2 * H + k] = gg; gr[3 * H + k] = og;
      hs[r * ldhs + k] = h;
      cl[e] = c;
      hl[e] = h;
    }
#pragma unroll
    for (int bb = 0; bb < BCT; ++bb) gcur[bb] = gnext[bb];
    __syncthreads();
  }
  for (int e = j; e < nb * H; e += G) {
    const int b = e / H, k = e - b * H;
    if (hT) hT[(size_t)(b0 + b) * H + k] = hl[e];
    if (cT) cT[(size_t)(b0 + b) * H + k] = cl[e];
  }
}

// ------------------------------------------------------------------ LSTM backward
template <int H, int BCT>
__global__ __launch_bounds__(4 * H) void lstm_persist_bwd(
    const float* __restrict__ dhs, int lddhs, const float* __restrict__ dhT,
    const float* __restrict__ dcT, const float* __restrict__ gates, const float* __restrict__ cs,
    const float* __restrict__ c0, const float* __restrict__ w_hh, float* __restrict__ dgates,
    float* __restrict__ dh0, float* __restrict__ dc0, int T, int B, int rst, int rsb,
    int reverse) {
  constexpr int G = 4 * H;
  constexpr int EPT = (BCT * H + G - 1) / G;  // pointwise elements per thread (2 for 8 rows, 1 below 5)
  __shared__ __attribute__((aligned(16))) float dGl[BCT * G];
  __shared__ float part[4][BCT * H];
  __shared__ float dhl[BCT * H];
  __shared__ float dcl[BCT * H];
  const int tid = threadIdx.x;
  const int q = tid / H, k = tid - q * H;
  const int b0 = blockIdx.x * BCT;
  const int nb = min(BCT, B - b0);

  float wc[H];  // column k of gate block q:  W_hh[q*H + jj][k]
#pragma unroll
  for (int jj = 0; jj < H; ++jj) wc[jj] = w_hh[((size_t)q * H + jj) * H + k];

  for (int e = tid; e < BCT * H; e += G) {
    const int b = e / H, kk = e - b * H;
    const bool live = b < nb;
    dhl[e] = (live && dhT) ? dhT[(size_t)(b0 + b) * H + kk] : 0.f;
    dcl[e] = (live && dcT) ? dcT[(size_t)(b0 + b) * H + kk] : 0.f;
  }
  for (int e = tid; e < BCT * G; e += G) dGl[e] = 0.f;

  // per-thread prefetch registers for the pointwise phase
  float p_i[EPT], p_f[EPT], p_g[EPT], p_o[EPT], p_c[EPT], p_cp[EPT], p_dh[EPT];
  auto prefetch = [&](int step) {
    const int t = reverse ? step : T - 1 - step;       // backward walks fwd order in reverse
    const bool first_fwd = reverse ? (t == T - 1) : (t == 0);
    const int tprev = reverse ? t + 1 : t - 1;
#pragma unroll
    for (int u = 0; u < EPT; ++u) {
      const int e = tid + u * G;
      const int b = e / H, kk = e - b * H;
      if (e < BCT * H && b < nb) {
        const size_t r = (size_t)t * rst + (size_t)(b0 + b) * rsb;
        const float* gr = gates + r * G;
        p_i[u] = gr[kk]; p_f[u] = gr[H + kk]; p_g[u] = gr[2 * H + kk]; p_o[u] = gr[3 * H + kk];
        p_c[u] = cs[r * H + kk];
        if (first_fwd) p_cp[u] = c0 ? c0[(size_t)(b0 + b) * H + kk] : 0.f;
        else p_cp[u] = cs[((size_t)tprev * rst + (size_t)(b0 + b) * rsb) * H + kk];
        p_dh[u] = dhs ? dhs[r * lddhs + kk] : 0.f;
      }
    }
  };
  prefetch(0);
  __syncthreads();

  for (int step = 0; step < T; ++step) {
    const int t = reverse ? step : T - 1 - step;
    // phase A: pointwise gate gradients
#pragma unroll
    for (int u = 0; u < EPT; ++u) {
      const int e = tid + u * G;
      const int b = e / H, kk = e - b * H;
      if (e < BCT * H && b < nb) {
        const size_t r = (size_t)t * rst + (size_t)(b0 + b) * rsb;
        const float ig = p_i[u], fg = p_f[u], gg = p_g[u], og = p_o[u];
        const float dh = p_dh[u] + dhl[e];
        const float tc = tanhf(p_c[u]);
        const float dc = fmaf(dh * og, 1.f - tc * tc, dcl[e]);
        const float dai = dc * gg * ig * (1.f - ig);
        const float daf = dc * p_cp[u] * fg * (1.f - fg);
        const float dag = dc * ig * (1.f - gg * gg);
        const float dao = dh * tc * og * (1.f - og);
        dcl[e] = dc * fg;
        float* go = dgates + r * G;
        go[kk] = dai; go[H + kk] = daf; go[2 * H + kk] = dag; go[3 * H + kk] = dao;
        float* gl = dGl + b * G;
        gl[kk] = dai; gl[H + kk] = daf; gl[2 * H + kk] = dag; gl[3 * H + kk] = dao;
      }
    }
    if (step + 1 < T) prefetch(step + 1);
    __syncthreads();
    // phase B: partial dh_{prev}[b][k] over gate block q
    {
      float acc[BCT];
#pragma unroll
      for (int bb = 0; bb < BCT; ++bb) acc[bb] = 0.f;
#pragma unroll
      for (int jj = 0; jj < H; jj += 4) {
#pragma unroll
        for (int bb = 0; bb < BCT; ++bb) {
          const float4 gv = *reinterpret_cast<const float4*>(&dGl[bb * G + q * H + jj]);
          acc[bb] = fmaf(wc[jj], gv.x, acc[bb]);
          acc[bb] = fmaf(wc[jj + 1], gv.y, acc[bb]);
          acc[bb] = fmaf(wc[jj + 2], gv.z, acc[bb]);
          acc[bb] = fmaf(wc[jj + 3], gv.w, acc[bb]);
        }
      }
#pragma unroll
      for (int bb = 0; bb < BCT; ++bb) part[q][bb * H + k] = acc[bb];
    }
    __syncthreads();
    // phase C: fixed-order sum of the four blocks
    for (int e = tid; e < BCT * H; e += G)
      dhl[e] = ((part[0][e] + part[1][e]) + part[2][e]) + part[3][e];
    __syncthreads();
  }
  for (int e = tid; e < nb * H; e += G) {
    const int b = e / H, kk = e - b * H;
    if (dh0) dh0[(size_t)(b0 + b) * H + kk] = dhl[e];
    if (dc0) dc0[(size_t)(b0 + b) * H + kk] = dcl[e];
  }
}

// ------------------------------------------------------------------ streamed cells
// zero_state (first step of a sequence that starts from zeros): the recurrent product is zero, the pre-activations are
// gx[t] + b_hh read straight from `pre` = gx with row stride ldpre and the bias `bh` -- no state initialisation launch, no
// weight-streaming launch for that step
__global__ void lstm_cell_fwd_kernel(const float* __restrict__ pre, int ldpre, const float* __restrict__ bh,
                                     float* __restrict__ hcur, float* __restrict__ ccur, float* __restrict__ hs, int ldhs,
                                     float* __restrict__ cs, float* __restrict__ hp,
                                     float* __restrict__ gates, int t, int B, int H, int rst,
                                     int rsb, int zero_state) {
  const int G = 4 * H;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < B * H; e += gridDim.x * blockDim.x) {
    const int b = e / H, k = e - b * H;
    const size_t r = (size_t)t * rst + (size_t)b * rsb;
    const float* p = pre + (size_t)b * ldpre;
    const float b0 = bh ? bh[k] : 0.f, b1 = bh ? bh[H + k] : 0.f, b2 = bh ? bh[2 * H + k] : 0.f, b3 = bh ? bh[3 * H + k] : 0.f;
    const float ig = sigm(p[k] + b0), fg = sigm(p[H + k] + b1), gg = tanhf(p[2 * H + k] + b2),
                og = sigm(p[3 * H + k] + b3);
    const float cprev = zero_state ? 0.f : ccur[e], hprev = zero_state ? 0.f : hcur[e];
    const float c = fmaf(fg, cprev, ig * gg);
    const float h = og * tanhf(c);
    hp[r * H + k] = hprev;
    cs[r * H + k] = c;
    float* gr = gates + r * G;
    gr[k] = ig; gr[H + k] = fg; gr[2 * H + k] = gg; gr[3 * H + k] = og;
    hs[r * ldhs + k] = h;
    ccur[e] = c;
    hcur[e] = h;
  }
}

__global__ void lstm_cell_bwd_kernel(const float* __restrict__ dhs, int lddhs,
                                     const float* __restrict__ dhrec, float* __restrict__ dccur,
                                     const float* __restrict__ gates, const float* __restrict__ cs,
                                     const float* __restrict__ c0, float* __restrict__ dgates,
                                     float* __restrict__ dG, int t, int tprev, int first_fwd,
                                     int B, int H, int rst, int rsb) {
  const int G = 4 * H;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < B * H; e += gridDim.x * blockDim.x) {
    const int b = e / H, k = e - b * H;
    const size_t r = (size_t)t * rst + (size_t)b * rsb;
    const float* gr = gates + r * G;
    const float ig = gr[k], fg = gr[H + k], gg = gr[2 * H + k], og = gr[3 * H + k];
    const float c = cs[r * H + k];
    float cp;
    if (first_fwd) cp = c0 ? c0[e] : 0.f;
    else cp = cs[((size_t)tprev * rst + (size_t)b * rsb) * H + k];
    const float dh = (dhs ? dhs[r * lddhs + k] : 0.f) + dhrec[e];
    const float tc = tanhf(c);
    const float dc = fmaf(dh * og, 1.f - tc * tc, dccur[e]);
    const float dai = dc * gg * ig * (1.f - ig);
    const float daf = dc * cp * fg * (1.f - fg);
    const float dag = dc * ig * (1.f - gg * gg);
    const float dao = dh * tc * og * (1.f - og);
    dccur[e] = dc * fg;
    float* go = dgates + r * G;
    go[k] = dai; go[H + k] = daf; go[2 * H + k] = dag; go[3 * H + k] = dao;
    float* gl = dG + (size_t)b * G;
    gl[k] = dai; gl[H + k] = daf; gl[2 * H + k] = dag; gl[3 * H + k] = dao;
  }
}

__global__ void init_state_kernel(const float* __restrict__ src, float* __restrict__ dst, int n) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    dst[i] = src ? src[i] : 0.f;
}

// two state vectors (h and c, or dh and dc) in one launch; a null dst is skipped
__global__ void init_state2_kernel(const float* __restrict__ src_a, float* __restrict__ dst_a,
                                   const float* __restrict__ src_b, float* __restrict__ dst_b, int n) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    if (dst_a) dst_a[i] = src_a ? src_a[i] : 0.f;
    if (dst_b) dst_b[i] = src_b ? src_b[i] : 0.f;
  }
}

// ------------------------------------------------------------------ GRU (streamed)
// gates saved: r, z, n, hn(= W_hn h + b_hn)
__global__ void gru_cell_fwd_kernel(const float* __restrict__ gx, const float* __restrict__ gh,
                                    float* __restrict__ hcur, float* __restrict__ hs, int ldhs,
                                    float* __restrict__ hp, float* __restrict__ gates, int t,
                                    int B, int H, int rst, int rsb) {
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < B * H; e += gridDim.x * blockDim.x) {
    const int b = e / H, k = e - b * H;
    const size_t r = (size_t)t * rst + (size_t)b * rsb;
    const float* x = gx + r * 3 * H;
    const float* g = gh + (size_t)b * 3 * H;
    const float rg = sigm(x[k] + g[k]);
    const float zg = sigm(x[H + k] + g[H + k]);
    const float hn = g[2 * H + k];
    const float ng = tanhf(x[2 * H + k] + rg * hn);
    const float hprev = hcur[e];
    const float h = (1.f - zg) * ng + zg * hprev;
    hp[r * H + k] = hprev;
    float* gr = gates + r * 4 * H;
    gr[k] = rg; gr[H + k] = zg; gr[2 * H + k] = ng; gr[3 * H + k] = hn;
    hs[r * ldhs + k] = h;
    hcur[e] = h;
  }
}

__global__ void gru_cell_bwd_kernel(const float* __restrict__ dhs, int lddhs,
                                    float* __restrict__ dhcur, const float* __restrict__ gates,
                                    const float* __restrict__ hp, float* __restrict__ dgx,
                                    float* __restrict__ dgh, float* __restrict__ dGh, int t, int B,
                                    int H, int rst, int rsb) {
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < B * H; e += gridDim.x * blockDim.x) {
    const int b = e / H, k = e - b * H;
    const size_t r = (size_t)t * rst + (size_t)b * rsb;
    const float* gr = gates + r * 4 * H;
    const float rg = gr[k], zg = gr[H + k], ng = gr[2 * H + k], hn = gr[3 * H + k];
    const float hprev = hp[r * H + k];
    const float dh = (dhs ? dhs[r * lddhs + k] : 0.f) + dhcur[e];
    const float dn = dh * (1.f - zg);
    const float dz = dh * (hprev - ng);
    const float dan = dn * (1.f - ng * ng);
    const float dar = dan * hn * rg * (1.f - rg);
    const float daz = dz * zg * (1.f - zg);
    float* ox = dgx + r * 3 * H;
    ox[k] = dar; ox[H + k] = daz; ox[2 * H + k] = dan;
    float* oh = dgh + r * 3 * H;
    const float dhn = dan * rg;
    oh[k] = dar; oh[H + k] = daz; oh[2 * H + k] = dhn;
    float* gl = dGh + (size_t)b * 3 * H;
    gl[k] = dar; gl[H + k] = daz; gl[2 * H + k] = dhn;
    dhcur[e] = dh * zg;  // direct path; the W_hh path is accumulated by linear_bwd_data
  }
}

// ------------------------------------------------------------------ GRU (persistent)
// Same scheme as lstm_persist_*: ONE launch per (layer, direction, sequence), 3H threads, thread j
// keeps row j of W_hh (forward) / a column of one gate block (backward) in registers for all T
// steps, h and the gate pre-activations live in LDS, the next step's gx is prefetched.
// gates saved per row: r, z, n, hn (= W_hn h + b_hn), as the streamed cells do.
template <int H, int BCT>
__global__ __launch_bounds__(3 * H) void gru_persist_fwd(
    const float* __restrict__ gx, const float* __restrict__ w_hh, const float* __restrict__ b_hh,
    const float* __restrict__ h0, float* __restrict__ hs, int ldhs, float* __restrict__ hp,
    float* __restrict__ gates, float* __restrict__ hT, int T, int B, int rst, int rsb, int reverse) {
  constexpr int G = 3 * H;
  __shared__ __attribute__((aligned(16))) float hl[BCT * H];
  __shared__ __attribute__((aligned(16))) float gl[BCT * G];    // r,z: gx+gh ; n: gh only
  __shared__ float gxn[BCT * H];                                // gx of the n gate
  const int j = threadIdx.x;
  const int b0 = blockIdx.x * BCT;
  const int nb = min(BCT, B - b0);

  float w[H];
#pragma unroll
  for (int k = 0; k < H; k += 4) {
    const float4 v = *reinterpret_cast<const float4*>(w_hh + (size_t)j * H + k);
    w[k] = v.x; w[k + 1] = v.y; w[k + 2] = v.z; w[k + 3] = v.w;
  }
  const float bj = b_hh ? b_hh[j] : 0.f;
  for (int e = j; e < BCT * H; e += G) {
    const int b = e / H, k = e - b * H;
    hl[e] = (b < nb && h0) ? h0[(size_t)(b0 + b) * H + k] : 0.f;
  }
  float gcur[BCT], gnext[BCT];
  {
    const int t = reverse ? T - 1 : 0;
#pragma unroll
    for (int bb = 0; bb < BCT; ++bb)
      gcur[bb] = bb < nb ? gx[((size_t)t * rst + (size_t)(b0 + bb) * rsb) * G + j] : 0.f;
  }
  __syncthreads();

  for (int step = 0; step < T; ++step) {
    const int t = reverse ? T - 1 - step : step;
    if (step + 1 < T) {
      const int tn = reverse ? t - 1 : t + 1;
#pragma unroll
      for (int bb = 0; bb < BCT; ++bb)
        gnext[bb] = bb < nb ? gx[((size_t)tn * rst + (size_t)(b0 + bb) * rsb) * G + j] : 0.f;
    }
    float acc[BCT];
#pragma unroll
    for (int bb = 0; bb < BCT; ++bb) acc[bb] = bj;
#pragma unroll
    for (int k = 0; k < H; k += 4) {
#pragma unroll
      for (int bb = 0; bb < BCT; ++bb) {
        const float4 hv = *reinterpret_cast<const float4*>(&hl[bb * H + k]);
        acc[bb] = fmaf(w[k], hv.x, acc[bb]);
        acc[bb] = fmaf(w[k + 1], hv.y, acc[bb]);
        acc[bb] = fmaf(w[k + 2], hv.z, acc[bb]);
        acc[bb] = fmaf(w[k + 3], hv.w, acc[bb]);
      }
    }
    if (j < 2 * H) {
#pragma unroll
      for (int bb = 0; bb < BCT; ++bb) gl[bb * G + j] = gcur[bb] + acc[bb];
    } else {
#pragma unroll
      for (int bb = 0; bb < BCT; ++bb) { gl[bb * G + j] = acc[bb]; gxn[bb * H + j - 2 * H] = gcur[bb]; }
    }
    __syncthreads();
    for (int e = j; e < nb * H; e += G) {
      const int b = e / H, k = e - b * H;
      const size_t r = (size_t)t * rst + (size_t)(b0 + b) * rsb;
      const float rg = sigm(gl[b * G + k]);
      const float zg = sigm(gl[b * G + H + k]);
      const float hn = gl[b * G + 2 * H + k];
      const float ng = tanhf(gxn[e] + rg * hn);
      const float hprev = hl[e];
      const float h = (1.f - zg) * ng + zg * hprev;
      hp[r * H + k] = hprev;
      float* gr = gates + r * 4 * H;
      gr[k] = rg; gr[H + k] = zg; gr[2 * H + k] = ng; gr[3 * H + k] = hn;
      hs[r * ldhs + k] = h;
      hl[e] = h;
    }
#pragma unroll
    for (int bb = 0; bb < BCT; ++bb) gcur[bb] = gnext[bb];
    __syncthreads();
  }
  if (hT)
    for (int e = j; e < nb * H; e += G) hT[(size_t)(b0 + e / H) * H + e % H] = hl[e];
}

template <int H, int BCT>
__global__ __launch_bounds__(3 * H) void gru_persist_bwd(
    const float* __restrict__ dhs, int lddhs, const float* __restrict__ dhT,
    const float* __restrict__ gates, const float* __restrict__ hp, const float* __restrict__ w_hh,
    float* __restrict__ dgx, float* __restrict__ dgh, float* __restrict__ dh0, int T, int B, int rst,
    int rsb, int reverse) {
  constexpr int G = 3 * H;
  __shared__ __attribute__((aligned(16))) float dGl[BCT * G];
  __shared__ float part[3][BCT * H];
  __shared__ float dhl[BCT * H];      // gradient arriving from the later step
  __shared__ float dir[BCT * H];      // dh * z : the path that bypasses W_hh
  const int tid = threadIdx.x;
  const int q = tid / H, k = tid - q * H;
  const int b0 = blockIdx.x * BCT;
  const int nb = min(BCT, B - b0);

  float wc[H];  // column k of gate block q:  W_hh[q*H + jj][k]
#pragma unroll
  for (int jj = 0; jj < H; ++jj) wc[jj] = w_hh[((size_t)q * H + jj) * H + k];
  for (int e = tid; e < BCT * H; e += G) {
    const int b = e / H, kk = e - b * H;
    dhl[e] = (b < nb && dhT) ? dhT[(size_t)(b0 + b) * H + kk] : 0.f;
    dir[e] = 0.f;
  }
  for (int e = tid; e < BCT * G; e += G) dGl[e] = 0.f;
  __syncthreads();

  for (int step = 0; step < T; ++step) {
    const int t = reverse ? step : T - 1 - step;
    for (int e = tid; e < nb * H; e += G) {
      const int b = e / H, kk = e - b * H;
      const size_t r = (size_t)t * rst + (size_t)(b0 + b) * rsb;
      const float* gr = gates + r * 4 * H;
      const float rg = gr[kk], zg = gr[H + kk], ng = gr[2 * H + kk], hn = gr[3 * H + kk];
      const float hprev = hp[r * H + kk];
      const float dh = (dhs ? dhs[r * lddhs + kk] : 0.f) + dhl[e];
      const float dn = dh * (1.f - zg);
      const float dz = dh * (hprev - ng);
      const float dan = dn * (1.f - ng * ng);
      const float dar = dan * hn * rg * (1.f - rg);
      const float daz = dz * zg * (1.f - zg);
      const float dhn = dan * rg;
      float* ox = dgx + r * G;
      ox[kk] = dar; ox[H + kk] = daz; ox[2 * H + kk] = dan;
      float* oh = dgh + r * G;
      oh[kk] = dar; oh[H + kk] = daz; oh[2 * H + kk] = dhn;
      float* gl = dGl + b * G;
      gl[kk] = dar; gl[H + kk] = daz; gl[2 * H + kk] = dhn;
      dir[e] = dh * zg;
    }
    __syncthreads();
    {
      float acc[BCT];
#pragma unroll
      for (int bb = 0; bb < BCT; ++bb) acc[bb] = 0.f;
#pragma unroll
      for (int jj = 0; jj < H; jj += 4) {
#pragma unroll
        for (int bb = 0; bb < BCT; ++bb) {
          const float4 gv = *reinterpret_cast<const float4*>(&dGl[bb * G + q * H + jj]);
          acc[bb] = fmaf(wc[jj], gv.x, acc[bb]);
          acc[bb] = fmaf(wc[jj + 1], gv.y, acc[bb]);
          acc[bb] = fmaf(wc[jj + 2], gv.z, acc[bb]);
          acc[bb] = fmaf(wc[jj + 3], gv.w, acc[bb]);
        }
      }
#pragma unroll
      for (int bb = 0; bb < BCT; ++bb) part[q][bb * H + k] = acc[bb];
    }
    __syncthreads();
    for (int e = tid; e < BCT * H; e += G) dhl[e] = dir[e] + ((part[0][e] + part[1][e]) + part[2][e]);
    __syncthreads();
  }
  if (dh0)
    for (int e = tid; e < nb * H; e += G) dh0[(size_t)(b0 + e / H) * H + e % H] = dhl[e];
}

int ew_blocks(int n) { return n < 256 ? 1 : (n + 255) / 256 > 1024 ? 1024 : (n + 255) / 256; }

}  // namespace

extern "C" size_t dlio_rnn_ws_bytes(int T, int B, int H) {
  (void)T;
  if (B <= 0 || H <= 0) return 0;
  // state + gate scratch (16 B*H floats) followed by the weight-streaming scratch of the
  // recurrent data-gradient GEMV
  return (size_t)B * H * 16 * sizeof(float) + dlio_linear_bwd_data_ws_bytes(B, 4 * H, H);
}

extern "C" int dlio_lstm_seq_fwd(const float* gx, const float* w_hh, const float* b_hh,
                                 const float* h0, const float* c0, float* hs, int ldhs, float* cs,
                                 float* hp, float* gates, float* hT, float* cT, int T, int B, int H,
                                 int rst, int rsb, int reverse, void* ws, size_t ws_bytes,
                                 dlio_stream_t stream) {
  if (!gx || !w_hh || !hs || !cs || !hp || !gates || T <= 0 || B <= 0 || H <= 0 || ldhs < H)
    return DLIO_EINVAL;
  hipStream_t s = as_stream(stream);
  const int rows = lstm_rows_per_wg(B);
#define LSTM_FWD_B(HH, RB)                                                                   \
  if (rows == RB)                                                                            \
    hipLaunchKernelGGL((lstm_persist_fwd<HH, RB>), dim3(cdiv(B, RB)), dim3(4 * HH), 0, s, gx, w_hh, b_hh, h0, c0, hs, ldhs, \
                       cs, hp, gates, hT, cT, T, B, rst, rsb, reverse);
#define LSTM_FWD_P(HH)                                                                         \
  if (H == HH) {                                                                               \
    LSTM_FWD_B(HH, 1) else LSTM_FWD_B(HH, 2) else LSTM_FWD_B(HH, 4) else LSTM_FWD_B(HH, 8)            \
    return dlio_check_launch();                                                                \
  }
  LSTM_FWD_P(32)
  LSTM_FWD_P(64)
  LSTM_FWD_P(128)
#undef LSTM_FWD_P
#undef LSTM_FWD_B
  // streamed path
  if (!ws || ws_bytes < dlio_rnn_ws_bytes(T, B, H)) return DLIO_EWS;
  float* f = reinterpret_cast<float*>(ws);
  float* hcur = f;
  float* ccur = f + (size_t)B * H;
  float* pre = f + (size_t)2 * B * H;  // [B][4H]
  const int n = B * H;
  const bool from_zero = !h0 && !c0;           // (the odometry LSTM: every sequence starts from zeros)
  if (!from_zero) hipLaunchKernelGGL(init_state2_kernel, dim3(ew_blocks(n)), dim3(256), 0, s, h0, hcur, c0, ccur, n);
  for (int step = 0; step < T; ++step) {
    const int t = reverse ? T - 1 - step : step;
    if (step == 0 && from_zero) {
      // the sum a + b_hh + 0 the linear launch would have produced, formed in the cell kernel: fl(gx + b) in both cases
      hipLaunchKernelGGL(lstm_cell_fwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, s, gx + (size_t)t * rst * 4 * H,
                         rsb * 4 * H, b_hh, hcur, ccur, hs, ldhs, cs, hp, gates, t, B, H, rst, rsb, 1);
      continue;
    }
    int rc = dlio_linear_fwd(hcur, H, w_hh, b_hh, gx + (size_t)t * rst * 4 * H, rsb * 4 * H, pre,
                             4 * H, B, 4 * H, H, 0, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(lstm_cell_fwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, s, (const float*)pre, 4 * H,
                       (const float*)nullptr, hcur, ccur, hs, ldhs, cs, hp, gates, t, B, H, rst, rsb, 0);
  }
  if (hT || cT)
    hipLaunchKernelGGL(init_state2_kernel, dim3(ew_blocks(n)), dim3(256), 0, s, (const float*)hcur, hT,
                       (const float*)ccur, cT, n);
  return dlio_check_launch();
}

extern "C" int dlio_lstm_seq_bwd(const float* dhs, int lddhs, const float* dhT, const float* dcT,
                                 const float* gates, const float* cs, const float* c0,
                                 const float* w_hh, float* dgates, float* dh0, float* dc0, int T,
                                 int B, int H, int rst, int rsb, int reverse, void* ws,
                                 size_t ws_bytes, dlio_stream_t stream) {
  if (!gates || !cs || !w_hh || !dgates || T <= 0 || B <= 0 || H <= 0) return DLIO_EINVAL;
  hipStream_t s = as_stream(stream);
  const int rows = lstm_rows_per_wg(B);
#define LSTM_BWD_B(HH, RB)                                                                  \
  if (rows == RB)                                                                           \
    hipLaunchKernelGGL((lstm_persist_bwd<HH, RB>), dim3(cdiv(B, RB)), dim3(4 * HH), 0, s, dhs, lddhs, dhT, dcT, gates, cs, \
                       c0, w_hh, dgates, dh0, dc0, T, B, rst, rsb, reverse);
#define LSTM_BWD_P(HH)                                                                        \
  if (H == HH) {                                                                              \
    LSTM_BWD_B(HH, 1) else LSTM_BWD_B(HH, 2) else LSTM_BWD_B(HH, 4) else LSTM_BWD_B(HH, 8)           \
    return dlio_check_launch();                                                               \
  }
  LSTM_BWD_P(32)
  LSTM_BWD_P(64)
  LSTM_BWD_P(128)
#undef LSTM_BWD_P
#undef LSTM_BWD_B
  if (!ws || ws_bytes < dlio_rnn_ws_bytes(T, B, H)) return DLIO_EWS;
  float* f = reinterpret_cast<float*>(ws);
  float* dhrec = f;
  float* dccur = f + (size_t)B * H;
  float* dG = f + (size_t)2 * B * H;  // [B][4H]
  const int n = B * H;
  hipLaunchKernelGGL(init_state2_kernel, dim3(ew_blocks(n)), dim3(256), 0, s, dhT, dhrec, dcT, dccur, n);
  for (int step = 0; step < T; ++step) {
    const int t = reverse ? step : T - 1 - step;
    const int first_fwd = reverse ? (t == T - 1) : (t == 0);
    const int tprev = reverse ? t + 1 : t - 1;
    hipLaunchKernelGGL(lstm_cell_bwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, s, dhs, lddhs,
                       dhrec, dccur, gates, cs, c0, dgates, dG, t, tprev, first_fwd, B, H, rst,
                       rsb);
    if (step == T - 1 && !dh0) break;      // the last recurrent data gradient is dh0: nobody asked
    int rc = dlio_linear_bwd_data(dG, 4 * H, w_hh, dhrec, H, B, 4 * H, H, 0, f + (size_t)16 * B * H,
                                  ws_bytes - (size_t)16 * B * H * sizeof(float), stream);
    if (rc) return rc;
  }
  if (dh0 || dc0)
    hipLaunchKernelGGL(init_state2_kernel, dim3(ew_blocks(n)), dim3(256), 0, s, (const float*)dhrec, dh0,
                       (const float*)dccur, dc0, n);
  return dlio_check_launch();
}

extern "C" int dlio_gru_seq_fwd(const float* gx, const float* w_hh, const float* b_hh,
                                const float* h0, float* hs, int ldhs, float* hp, float* gates,
                                float* hT, int T, int B, int H, int rst, int rsb, int reverse,
                                void* ws, size_t ws_bytes, dlio_stream_t stream) {
  if (!gx || !w_hh || !hs || !hp || !gates || T <= 0 || B <= 0 || H <= 0 || ldhs < H)
    return DLIO_EINVAL;
  hipStream_t s = as_stream(stream);
  static const int persist = 1;
#define GRU_FWD_P(HH)                                                                          \
  if (persist && H == HH) {                                                                    \
    const int rows = lstm_rows_per_wg(B);                                                       \
    if (rows == 1) hipLaunchKernelGGL((gru_persist_fwd<HH, 1>), dim3(B), dim3(3 * HH), 0, s, gx, w_hh,   \
                       b_hh, h0, hs, ldhs, hp, gates, hT, T, B, rst, rsb, reverse); \
    else if (rows == 2) hipLaunchKernelGGL((gru_persist_fwd<HH, 2>), dim3(cdiv(B, 2)), dim3(3 * HH), 0, s, gx, w_hh,   \
                       b_hh, h0, hs, ldhs, hp, gates, hT, T, B, rst, rsb, reverse); \
    else if (rows == 4) hipLaunchKernelGGL((gru_persist_fwd<HH, 4>), dim3(cdiv(B, 4)), dim3(3 * HH), 0, s, gx, w_hh,   \
                       b_hh, h0, hs, ldhs, hp, gates, hT, T, B, rst, rsb, reverse); \
    else hipLaunchKernelGGL((gru_persist_fwd<HH, 8>), dim3(cdiv(B, 8)), dim3(3 * HH), 0, s, gx, w_hh,   \
                       b_hh, h0, hs, ldhs, hp, gates, hT, T, B, rst, rsb, reverse);            \
    return dlio_check_launch();                                                                \
  }
  GRU_FWD_P(32)
  GRU_FWD_P(64)
  GRU_FWD_P(128)
#undef GRU_FWD_P
  if (!ws || ws_bytes < dlio_rnn_ws_bytes(T, B, H)) return DLIO_EWS;
  float* f = reinterpret_cast<float*>(ws);
  float* hcur = f;
  float* gh = f + (size_t)B * H;  // [B][3H]
  const int n = B * H;
  hipLaunchKernelGGL(init_state_kernel, dim3(ew_blocks(n)), dim3(256), 0, s, h0, hcur, n);
  for (int step = 0; step < T; ++step) {
    const int t = reverse ? T - 1 - step : step;
    int rc = dlio_linear_fwd(hcur, H, w_hh, b_hh, nullptr, 0, gh, 3 * H, B, 3 * H, H, 0, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(gru_cell_fwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, s, gx, gh, hcur, hs,
                       ldhs, hp, gates, t, B, H, rst, rsb);
  }
  if (hT) hipLaunchKernelGGL(init_state_kernel, dim3(ew_blocks(n)), dim3(256), 0, s, hcur, hT, n);
  return dlio_check_launch();
}

extern "C" int dlio_gru_seq_bwd(const float* dhs, int lddhs, const float* dhT, const float* gates,
                                const float* hp, const float* w_hh, float* dgx, float* dgh,
                                float* dh0, int T, int B, int H, int rst, int rsb, int reverse,
                                void* ws, size_t ws_bytes, dlio_stream_t stream) {
  if (!gates || !hp || !w_hh || !dgx || !dgh || T <= 0 || B <= 0 || H <= 0) return DLIO_EINVAL;
  hipStream_t s = as_stream(stream);
  static const int persist = 1;
#define GRU_BWD_P(HH)                                                                          \
  if (persist && H == HH) {                                                                    \
    const int rows = lstm_rows_per_wg(B);                                                      \
    if (rows == 1) hipLaunchKernelGGL((gru_persist_bwd<HH, 1>), dim3(B), dim3(3 * HH), 0, s, dhs, lddhs, \
                       dhT, gates, hp, w_hh, dgx, dgh, dh0, T, B, rst, rsb, reverse); \
    else if (rows == 2) hipLaunchKernelGGL((gru_persist_bwd<HH, 2>), dim3(cdiv(B, 2)), dim3(3 * HH), 0, s, dhs, lddhs, \
                       dhT, gates, hp, w_hh, dgx, dgh, dh0, T, B, rst, rsb, reverse); \
    else if (rows == 4) hipLaunchKernelGGL((gru_persist_bwd<HH, 4>), dim3(cdiv(B, 4)), dim3(3 * HH), 0, s, dhs, lddhs, \
                       dhT, gates, hp, w_hh, dgx, dgh, dh0, T, B, rst, rsb, reverse); \
    else hipLaunchKernelGGL((gru_persist_bwd<HH, 8>), dim3(cdiv(B, 8)), dim3(3 * HH), 0, s, dhs, lddhs, \
                       dhT, gates, hp, w_hh, dgx, dgh, dh0, T, B, rst, rsb, reverse);          \
    return dlio_check_launch();                                                                \
  }
  GRU_BWD_P(32)
  GRU_BWD_P(64)
  GRU_BWD_P(128)
#undef GRU_BWD_P
  if (!ws || ws_bytes < dlio_rnn_ws_bytes(T, B, H)) return DLIO_EWS;
  float* f = reinterpret_cast<float*>(ws);
  float* dhcur = f;
  float* dGh = f + (size_t)B * H;  // [B][3H]
  const int n = B * H;
  hipLaunchKernelGGL(init_state_kernel, dim3(ew_blocks(n)), dim3(256), 0, s, dhT, dhcur, n);
  for (int step = 0; step < T; ++step) {
    const int t = reverse ? step : T - 1 - step;
    hipLaunchKernelGGL(gru_cell_bwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, s, dhs, lddhs, dhcur,
                       gates, hp, dgx, dgh, dGh, t, B, H, rst, rsb);
    if (step == T - 1 && !dh0) break;      // only dh0 is left to compute and nobody asked for it
    int rc = dlio_linear_bwd_data(dGh, 3 * H, w_hh, dhcur, H, B, 3 * H, H, 1, f + (size_t)16 * B * H,
                                  ws_bytes - (size_t)16 * B * H * sizeof(float), stream);
    if (rc) return rc;
  }
  if (dh0) hipLaunchKernelGGL(init_state_kernel, dim3(ew_blocks(n)), dim3(256), 0, s, dhcur, dh0, n);
  return dlio_check_launch();
}
