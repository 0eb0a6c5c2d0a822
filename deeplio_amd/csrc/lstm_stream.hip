// Streamed LSTM LAYER for a wide hidden state (the 1024-wide odometry bi-LSTM over the S axis: odom_feat_nets.py:61-68,80;
// W_ih / W_hh of one direction are 4-34 MB, the sequence is S = 2..5 steps, the batch 8 rows): one call = one layer, BOTH
// directions, the whole sequence, zero initial state.
//
// The layer is bound by streaming its weights from HBM (143 MB forward, as much again for the data gradients, and as much
// written by the weight gradients), the activations are a few hundred KB.  rnn.hip's streamed path (one direction per call,
// dlio_linear_* + a cell kernel per step, the reverse direction on a companion stream) moved them at 1.1-1.6 TB/s: a
// workgroup of the skinny-M kernel staged ALL K columns of x (128 KB of LDS: one workgroup per CU, as many bytes of x from L2
// per workgroup as of W from HBM) before its first product.  Here
//   * gemv_splitk_kernel: K is cut into slices of 512 (256) columns -- 32 KB of x per workgroup, several workgroups per CU --,
//     a workgroup owns 64 weight rows x one slice of one direction (grid.z), the product itself runs on the fp32 MFMA
//     (16 samples x 16 weight rows per wave, exact fp32), partial sums go to the workspace [D][KQ][rows][4H]; fixed
//     summation order (slice order, in the consumer);
//   * the cell kernels sum the slices, add both biases and do the pointwise LSTM update for both directions at once;
//     h_t is read by the next step's product straight from the layer's output buffer;
//   * backward: the transposed products (dh_{t-1} = dgates W_hh, dx = dgates W_ih) cut N = 4H into slabs as
//     dense.hip's linear_bwd_data_split_kernel does, both directions in one launch, the slab partials are summed by the
//     consumer (next cell step / one reduce launch for dx over both directions); all four weight gradients + biases
//     of the layer in ONE launch.
// Launches per layer: forward 2 T, backward 2 T + 2 (rnn.hip: 8 T + 4 per layer and direction pair, on two streams).
#include "common.h"

namespace {

__device__ __forceinline__ float sigm_(float x) { return 1.0f / (1.0f + expf(-x)); }

// p[0] + p[stride] + ... (n terms, in that order) with up to eight loads in flight
__device__ __forceinline__ float sum_strided(const float* __restrict__ p, int n, size_t stride) {
  float s = 0.f;
  int i = 0;
  for (; i + 8 <= n; i += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = p[(size_t)(i + u) * stride];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  if (i + 4 <= n) {
    float v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = p[(size_t)(i + u) * stride];
#pragma unroll
    for (int u = 0; u < 4; ++u) s += v[u];
    i += 4;
  }
  for (; i < n; ++i) s += p[(size_t)i * stride];
  return s;
}

// ------------------------------------------------------------------------------------------------- y_part = x W^T (slice of K)
// grid (ceil(N / 64), KQ, D); block 256 = 4 waves; wave w owns the 16 weight rows blockIdx.x * 64 + 16 w .. + 15 over one K slice
// of KS * 256 columns of one direction.  The product runs on v_mfma_f32_16x16x4_f32 (exact fp32: an fmaf chain) with the
// <= 16 samples as A and the 16 weight rows as B: lane (i = l & 15, kq = l >> 4) loads 32 contiguous bytes of weight row i
// (k = kb + 8 kq .. + 7: four lanes cover 128 contiguous bytes of a row) and the same columns of sample i from LDS, eight
// MFMAs consume them (any pairing of k between A and B is a valid order of the sum); no cross-lane reduction, 4 accumulator
// registers (two chains, joined at the end: a dependent MFMA waits 40 cycles), ~60 VGPRs.
template <int KS>
__global__ __launch_bounds__(256) void gemv_splitk_kernel(const float* __restrict__ x0, const float* __restrict__ x1, int ldx,
                                                          const float* __restrict__ w0, const float* __restrict__ w1,
                                                          float* __restrict__ part, int M, int Mtot, int m0, int N, int K) {
  constexpr int KQW = KS * 256;                                   // columns of a slice
  constexpr int XLD = KQW + 4;                                    // LDS row stride (floats): rows land on different banks
  constexpr int NB = KQW / 32;                                    // 32-column blocks of the slice
  constexpr int PF = 4;                                           // blocks in flight per lane (8 x 16 bytes)
  __shared__ __attribute__((aligned(16))) float xs[16 * XLD];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int q = blockIdx.y, d = blockIdx.z, KQ = gridDim.y;
  const float* __restrict__ x = d ? x1 : x0;
  const float* __restrict__ w = d ? w1 : w0;
  const int kbase = q * KQW;
  const int n0 = blockIdx.x * 64 + wave * 16;
  const int li = lane & 15, kq = lane >> 4;
  const float* __restrict__ wr = w + (size_t)min(n0 + li, N - 1) * K + kbase + 8 * kq;
  float4 wv[PF][2];
#pragma unroll
  for (int p = 0; p < PF; ++p) {                                   // in flight while x is staged
    wv[p][0] = *reinterpret_cast<const float4*>(wr + 32 * p);
    wv[p][1] = *reinterpret_cast<const float4*>(wr + 32 * p + 4);
  }
  constexpr int K4 = KQW / 4;
  for (int i = threadIdx.x; i < 16 * K4; i += 256) {
    const int m = i / K4, k4 = i - m * K4;
    *reinterpret_cast<float4*>(xs + m * XLD + 4 * k4) =
        m < M ? *reinterpret_cast<const float4*>(x + (size_t)m * ldx + kbase + 4 * k4) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  __syncthreads();
  const float* __restrict__ xr = xs + li * XLD + 8 * kq;
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int blk = 0; blk < NB; ++blk) {
    const float4 b0 = wv[blk % PF][0], b1 = wv[blk % PF][1];
    if (blk + PF < NB) {
      wv[blk % PF][0] = *reinterpret_cast<const float4*>(wr + 32 * (blk + PF));
      wv[blk % PF][1] = *reinterpret_cast<const float4*>(wr + 32 * (blk + PF) + 4);
    }
    const float4 a0 = *reinterpret_cast<const float4*>(xr + 32 * blk);
    const float4 a1 = *reinterpret_cast<const float4*>(xr + 32 * blk + 4);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, b0.x, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, b0.y, acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, b0.z, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, b0.w, acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, b1.x, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, b1.y, acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, b1.z, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, b1.w, acc1, 0, 0, 0);
  }
  // C: column (weight row) l & 15, sample row 4 (l >> 4) + r
  const int n = n0 + li;
  if (n < N) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = 4 * kq + r;
      if (m < M) part[(((size_t)d * KQ + q) * Mtot + m0 + m) * N + n] = acc0[r] + acc1[r];
    }
  }
}

// ------------------------------------------------------------------------------------------------- forward cell, both directions
// element (d, b, k); step s: direction 0 is at time s, direction 1 at time T - 1 - s; row(t, b) = b T + t
__global__ void lstm_cell_fwd2_kernel(const float* __restrict__ gxp, int KQi, const float* __restrict__ recp, int KQh,
                                      const float* __restrict__ bi0, const float* __restrict__ bh0,
                                      const float* __restrict__ bi1, const float* __restrict__ bh1,
                                      float* __restrict__ hs, int ldhs, float* __restrict__ cs, float* __restrict__ hp,
                                      float* __restrict__ gates, int step, int T, int B, int H, int D) {
  const int G = 4 * H, rows = B * T;
  const int total = D * B * H;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
    const int d = e / (B * H), r2 = e - d * (B * H), b = r2 / H, k = r2 - b * H;
    const int t = d ? T - 1 - step : step;
    const int tp = d ? t + 1 : t - 1;                              // the time the state comes from (step > 0)
    const int row = b * T + t, rowp = b * T + tp;
    const float* bi = d ? bi1 : bi0;
    const float* bh = d ? bh1 : bh0;
    float pre[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int n = g * H + k;
      const float* __restrict__ gp = gxp + ((size_t)d * KQi * rows + row) * G + n;
      const float* __restrict__ rp = recp + ((size_t)d * KQh * B + b) * G + n;
      pre[g] = (sum_strided(gp, KQi, (size_t)rows * G) + (bi ? bi[n] : 0.f)) +
               ((step > 0 ? sum_strided(rp, KQh, (size_t)B * G) : 0.f) + (bh ? bh[n] : 0.f));
    }
    const float ig = sigm_(pre[0]), fg = sigm_(pre[1]), gg = tanhf(pre[2]), og = sigm_(pre[3]);
    const float cprev = step > 0 ? cs[((size_t)d * rows + rowp) * H + k] : 0.f;
    const float hprev = step > 0 ? hs[(size_t)rowp * ldhs + d * H + k] : 0.f;
    const float c = fmaf(fg, cprev, ig * gg);
    const float h = og * tanhf(c);
    const size_t dr = (size_t)d * rows + row;
    hp[dr * H + k] = hprev;
    cs[dr * H + k] = c;
    float* gr = gates + dr * G;
    gr[k] = ig; gr[H + k] = fg; gr[2 * H + k] = gg; gr[3 * H + k] = og;
    hs[(size_t)row * ldhs + d * H + k] = h;
  }
}

// ------------------------------------------------------------------------------------------------- backward cell, both directions
// backward step s: direction 0 is at time T - 1 - s, direction 1 at time s; the recurrent gradient arrives as NS slab partials
__global__ void lstm_cell_bwd2_kernel(const float* __restrict__ dhs, int lddhs, const float* __restrict__ dhp, int NS,
                                      float* __restrict__ dccur, const float* __restrict__ gates, const float* __restrict__ cs,
                                      float* __restrict__ dgates, int step, int T, int B, int H, int D) {
  const int G = 4 * H, rows = B * T;
  const int total = D * B * H;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
    const int d = e / (B * H), r2 = e - d * (B * H), b = r2 / H, k = r2 - b * H;
    const int t = d ? step : T - 1 - step;
    const bool first_fwd = d ? (t == T - 1) : (t == 0);           // the step that started from the zero state
    const int tp = d ? t + 1 : t - 1;
    const int row = b * T + t;
    const size_t dr = (size_t)d * rows + row;
    const float* gr = gates + dr * G;
    const float ig = gr[k], fg = gr[H + k], gg = gr[2 * H + k], og = gr[3 * H + k];
    const float c = cs[dr * H + k];
    const float cp = first_fwd ? 0.f : cs[((size_t)d * rows + b * T + tp) * H + k];
    float dh = dhs ? dhs[(size_t)row * lddhs + d * H + k] : 0.f;
    if (step > 0) dh += sum_strided(dhp + ((size_t)d * NS * B + b) * H + k, NS, (size_t)B * H);
    const float tc = tanhf(c);
    const float dcin = step > 0 ? dccur[e] : 0.f;
    const float dc = fmaf(dh * og, 1.f - tc * tc, dcin);
    dccur[e] = dc * fg;
    float* go = dgates + dr * G;
    go[k] = dc * gg * ig * (1.f - ig);
    go[H + k] = dc * cp * fg * (1.f - fg);
    go[2 * H + k] = dc * ig * (1.f - gg * gg);
    go[3 * H + k] = dh * tc * og * (1.f - og);
  }
}

// ------------------------------------------------------------------------------------------------- dx_part = dz W (slab of N)
// grid (ceil(K / 256), NS, D): a block owns 256 columns k (one float4 per lane) and a slab of `per` <= 128 weight rows of one
// direction; its four waves take the slab's rows in interleaved groups of eight; ONE partial per block: part[D][NS][Mtot][K]
constexpr int SLAB = 128;
template <int MROWS>
__global__ __launch_bounds__(256) void gemv_t_splitn_kernel(const float* __restrict__ dz0, const float* __restrict__ dz1, int lddz,
                                                            const float* __restrict__ w0, const float* __restrict__ w1,
                                                            float* __restrict__ part, int M, int Mtot, int m0, int N, int K) {
  __shared__ __attribute__((aligned(16))) float sdz[MROWS][SLAB];
  __shared__ float4 red[3][MROWS][64];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int d = blockIdx.z, sp = blockIdx.y, NS = gridDim.y;
  const float* __restrict__ dz = d ? dz1 : dz0;
  const float* __restrict__ w = d ? w1 : w0;
  const int k = (blockIdx.x * 64 + lane) * 4;
  const int kc = k < K ? k : K - 4;
  const int per = (N + NS - 1) / NS;                              // <= SLAB (launcher)
  const int n_lo = sp * per, n_hi = min(N, n_lo + per);
  float4 wa[8], wb[8];
  auto load_rows = [&](float4 (&wv)[8], int r) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int n = n_lo + r + u;
      wv[u] = n < n_hi ? *reinterpret_cast<const float4*>(w + (size_t)n * K + kc) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  load_rows(wa, wave * 8);
  for (int i = threadIdx.x; i < MROWS * SLAB; i += 256) {
    const int j = i / SLAB, r = i - j * SLAB, n = n_lo + r;
    sdz[j][r] = (j < M && n < n_hi) ? dz[(size_t)j * lddz + n] : 0.f;
  }
  float4 acc[MROWS];
#pragma unroll
  for (int j = 0; j < MROWS; ++j) acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  __syncthreads();
  auto mul_rows = [&](const float4 (&wv)[8], int r) {
#pragma unroll
    for (int j = 0; j < MROWS; ++j) {
      const float4 d0 = *reinterpret_cast<const float4*>(&sdz[j][r]);
      const float4 d1 = *reinterpret_cast<const float4*>(&sdz[j][r + 4]);
      const float dd[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        acc[j].x += dd[u] * wv[u].x; acc[j].y += dd[u] * wv[u].y; acc[j].z += dd[u] * wv[u].z; acc[j].w += dd[u] * wv[u].w;
      }
    }
  };
  for (int r = wave * 8; r < per; r += 64) {
    if (r + 32 < per) load_rows(wb, r + 32);
    mul_rows(wa, r);
    if (r + 32 < per) {
      if (r + 64 < per) load_rows(wa, r + 64);
      mul_rows(wb, r + 32);
    }
  }
  if (wave) {
#pragma unroll
    for (int j = 0; j < MROWS; ++j) red[wave - 1][j][lane] = acc[j];
  }
  __syncthreads();
  if (wave == 0 && k < K) {
#pragma unroll
    for (int j = 0; j < MROWS; ++j) {
      if (j < M) {
        const float4 a = red[0][j][lane], b = red[1][j][lane], c = red[2][j][lane];
        float4 o;
        o.x = ((acc[j].x + a.x) + b.x) + c.x; o.y = ((acc[j].y + a.y) + b.y) + c.y;
        o.z = ((acc[j].z + a.z) + b.z) + c.z; o.w = ((acc[j].w + a.w) + b.w) + c.w;
        *reinterpret_cast<float4*>(part + (((size_t)d * NS + sp) * Mtot + m0 + j) * K + k) = o;
      }
    }
  }
}

// dx[m][k] = sum over (direction, slab) of part, fixed order
__global__ void slab_reduce_kernel(const float* __restrict__ part, float* __restrict__ dx, int lddx, int M, int K, int NP) {
  const int total = M * (K >> 2);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int m = i / (K >> 2), k = (i - m * (K >> 2)) * 4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    int p = 0;
    for (; p + 4 <= NP; p += 4) {
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(part + ((size_t)(p + u) * M + m) * K + k);
#pragma unroll
      for (int u = 0; u < 4; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
    }
    for (; p < NP; ++p) {
      const float4 v = *reinterpret_cast<const float4*>(part + ((size_t)p * M + m) * K + k);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    *reinterpret_cast<float4*>(dx + (size_t)m * lddx + k) = s;
  }
}

// ------------------------------------------------------------------------------------------------- all weight gradients of a layer
// grid (ceil(max(I, H) / 1024), 4H / 4, 2 D): z = 2 d + which (0: W_ih against x, 1: W_hh against hp); thread = (4 rows n, 4 k);
// the bias gradients (b_ih and b_hh both receive the column sums of dgates) ride in the first k block
struct WgSet { const float* dz; const float* x; int ldx; int K; float* dw; float* db; };
struct WgArgs { WgSet s[4]; };
__global__ __launch_bounds__(256) void lstm_wgrad_kernel(WgArgs a, int lddz, int M, int N, int accumulate) {
  const WgSet& s = a.s[blockIdx.z];
  const int K = s.K;
  const int k = (blockIdx.x * 256 + threadIdx.x) * 4;
  const int n0 = blockIdx.y * 4;
  const float* __restrict__ dz = s.dz;
  if (s.db && blockIdx.x == 0 && threadIdx.x >= 252 && n0 + (int)threadIdx.x - 252 < N) {
    const int n = n0 + (int)threadIdx.x - 252;
    float sb = 0.f;
    for (int m = 0; m < M; ++m) sb += dz[(size_t)m * lddz + n];
    s.db[n] = accumulate ? s.db[n] + sb : sb;
  }
  if (k >= K) return;
  const float* __restrict__ x = s.x;
  float4 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  // (the old values of an accumulating launch are requested first: they arrive under the product)
  float4 old[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
    old[j] = (accumulate && n0 + j < N) ? *reinterpret_cast<const float4*>(s.dw + (size_t)(n0 + j) * K + k)
                                        : make_float4(0.f, 0.f, 0.f, 0.f);
  const int nr = min(4, N - n0);
  auto fma_row = [&](int m, const float4& xv) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float g = j < nr ? dz[(size_t)m * lddz + n0 + j] : 0.f;
      acc[j].x += g * xv.x; acc[j].y += g * xv.y; acc[j].z += g * xv.z; acc[j].w += g * xv.w;
    }
  };
  int m = 0;
  for (; m + 4 <= M; m += 4) {
    float4 xv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) xv[u] = *reinterpret_cast<const float4*>(x + (size_t)(m + u) * s.ldx + k);
#pragma unroll
    for (int u = 0; u < 4; ++u) fma_row(m + u, xv[u]);
  }
  for (; m < M; ++m) fma_row(m, *reinterpret_cast<const float4*>(x + (size_t)m * s.ldx + k));
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (j < nr) {
      float4 o = old[j];
      o.x += acc[j].x; o.y += acc[j].y; o.z += acc[j].z; o.w += acc[j].w;
      *reinterpret_cast<float4*>(s.dw + (size_t)(n0 + j) * K + k) = o;
    }
  }
}

static inline int slabs_of(int N) { return (N + SLAB - 1) / SLAB; }
static inline int kq_of(int K) { return (K % 512 == 0) ? K / 512 : K / 256; }

static size_t fwd_ws_floats(int T, int B, int I, int H, int D) {
  return (size_t)D * kq_of(I) * B * T * 4 * H + (size_t)D * kq_of(H) * B * 4 * H;
}
static size_t bwd_ws_floats(int T, int B, int I, int H, int D) {
  const int NS = slabs_of(4 * H);
  return (size_t)D * NS * B * H + (size_t)D * B * H + (size_t)D * NS * B * T * I;
}

static void launch_gemv(hipStream_t s, const float* x0, const float* x1, int ldx, const float* w0, const float* w1, float* part,
                        int M, int Mtot, int m0, int N, int K, int D) {
  if (K % 512 == 0)
    hipLaunchKernelGGL((gemv_splitk_kernel<2>), dim3(cdiv(N, 64), K / 512, D), dim3(256), 0, s, x0, x1, ldx, w0, w1, part, M, Mtot,
                       m0, N, K);
  else
    hipLaunchKernelGGL((gemv_splitk_kernel<1>), dim3(cdiv(N, 64), K / 256, D), dim3(256), 0, s, x0, x1, ldx, w0, w1, part, M, Mtot,
                       m0, N, K);
}

extern "C" int dlio_lstm_layer_ok(int T, int B, int I, int H, int D);

static void launch_wgrad(hipStream_t s, const float* dgates, const float* x, int ldx, const float* hp, float* dw_ih0, float* dw_hh0,
                         float* db_ih0, float* db_hh0, float* dw_ih1, float* dw_hh1, float* db_ih1, float* db_hh1, int accumulate,
                         int T, int B, int I, int H, int D) {
  const int rows = B * T, G = 4 * H;
  WgArgs a;
  for (int d = 0; d < 2; ++d) {
    const bool on = d < D;
    const float* dz = dgates + (size_t)(on ? d : 0) * rows * G;
    a.s[2 * d + 0] = WgSet{dz, x, ldx, on ? I : 0, d ? dw_ih1 : dw_ih0, d ? db_ih1 : db_ih0};
    a.s[2 * d + 1] = WgSet{dz, hp + (size_t)(on ? d : 0) * rows * H, H, on ? H : 0, d ? dw_hh1 : dw_hh0, d ? db_hh1 : db_hh0};
  }
  const int kmax = I > H ? I : H;
  hipLaunchKernelGGL(lstm_wgrad_kernel, dim3(cdiv(kmax / 4, 256), G / 4, 2 * D), dim3(256), 0, s, a, G, rows, G, accumulate);
}

}  // namespace

extern "C" int dlio_lstm_layer_wgrad(const float* dgates, const float* x, int ldx, const float* hp, float* dw_ih0, float* dw_hh0,
                                     float* db_ih0, float* db_hh0, float* dw_ih1, float* dw_hh1, float* db_ih1, float* db_hh1,
                                     int accumulate, int T, int B, int I, int H, int D, dlio_stream_t stream) {
  if (!dgates || !x || !hp || !dw_ih0 || !dw_hh0 || (D == 2 && (!dw_ih1 || !dw_hh1))) return DLIO_EINVAL;
  if (!dlio_lstm_layer_ok(T, B, I, H, D) || ldx < I || (ldx & 3)) return DLIO_EUNSUP;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(hp) | reinterpret_cast<uintptr_t>(dw_ih0) |
       reinterpret_cast<uintptr_t>(dw_hh0) | reinterpret_cast<uintptr_t>(dw_ih1) | reinterpret_cast<uintptr_t>(dw_hh1)) & 15)
    return DLIO_EUNSUP;
  launch_wgrad(as_stream(stream), dgates, x, ldx, hp, dw_ih0, dw_hh0, db_ih0, db_hh0, dw_ih1, dw_hh1, db_ih1, db_hh1, accumulate, T,
               B, I, H, D);
  return dlio_check_launch();
}

extern "C" int dlio_lstm_layer_ok(int T, int B, int I, int H, int D) {
  return T >= 1 && B >= 1 && B <= 8 && (D == 1 || D == 2) && H >= 256 && H % 256 == 0 && I >= 256 && I % 256 == 0;
}

extern "C" size_t dlio_lstm_layer_ws_bytes(int T, int B, int I, int H, int D) {
  if (!dlio_lstm_layer_ok(T, B, I, H, D)) return 0;
  const size_t f = fwd_ws_floats(T, B, I, H, D), b = bwd_ws_floats(T, B, I, H, D);
  return (f > b ? f : b) * sizeof(float);
}

extern "C" int dlio_lstm_layer_fwd(const float* x, int ldx, const float* w_ih0, const float* w_hh0, const float* b_ih0,
                                   const float* b_hh0, const float* w_ih1, const float* w_hh1, const float* b_ih1,
                                   const float* b_hh1, float* hs, int ldhs, float* cs, float* hp, float* gates, int T, int B,
                                   int I, int H, int D, void* ws, size_t ws_bytes, dlio_stream_t stream) {
  if (!x || !w_ih0 || !w_hh0 || !hs || !cs || !hp || !gates || !ws || (D == 2 && (!w_ih1 || !w_hh1))) return DLIO_EINVAL;
  if (!dlio_lstm_layer_ok(T, B, I, H, D) || ldx < I || ldhs < D * H || (ldx & 3) || (ldhs & 3)) return DLIO_EUNSUP;
  if (ws_bytes < dlio_lstm_layer_ws_bytes(T, B, I, H, D)) return DLIO_EWS;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w_ih0) | reinterpret_cast<uintptr_t>(w_hh0) |
       reinterpret_cast<uintptr_t>(w_ih1) | reinterpret_cast<uintptr_t>(w_hh1) | reinterpret_cast<uintptr_t>(hs)) & 15)
    return DLIO_EUNSUP;
  hipStream_t s = as_stream(stream);
  const int rows = B * T, G = 4 * H;
  const int KQi = kq_of(I), KQh = kq_of(H);
  float* gxp = reinterpret_cast<float*>(ws);
  float* recp = gxp + (size_t)D * KQi * rows * G;
  // the input projection of every step of both directions: rows in passes of 16
  for (int m0 = 0; m0 < rows; m0 += 16) {
    const int M = rows - m0 < 16 ? rows - m0 : 16;
    launch_gemv(s, x + (size_t)m0 * ldx, x + (size_t)m0 * ldx, ldx, w_ih0, w_ih1, gxp, M, rows, m0, G, I, D);
  }
  const int cgrid = cdiv(D * B * H, 256);
  for (int step = 0; step < T; ++step) {
    if (step > 0) {
      // h of the previous step, straight from the output buffer: direction 0 at time step - 1, direction 1 at T - step
      const float* h0p = hs + (size_t)(step - 1) * ldhs;
      const float* h1p = hs + (size_t)(T - step) * ldhs + H;
      launch_gemv(s, h0p, h1p, T * ldhs, w_hh0, w_hh1, recp, B, B, 0, G, H, D);
    }
    hipLaunchKernelGGL(lstm_cell_fwd2_kernel, dim3(cgrid), dim3(256), 0, s, (const float*)gxp, KQi, (const float*)recp, KQh,
                       b_ih0, b_hh0, b_ih1, b_hh1, hs, ldhs, cs, hp, gates, step, T, B, H, D);
  }
  return dlio_check_launch();
}

extern "C" int dlio_lstm_layer_bwd(const float* dhs, int lddhs, const float* x, int ldx, const float* hp, const float* gates,
                                   const float* cs, const float* w_ih0, const float* w_hh0, const float* w_ih1,
                                   const float* w_hh1, float* dgates, float* dw_ih0, float* dw_hh0, float* db_ih0,
                                   float* db_hh0, float* dw_ih1, float* dw_hh1, float* db_ih1, float* db_hh1, int accumulate,
                                   float* dx, int lddx, int T, int B, int I, int H, int D, void* ws, size_t ws_bytes,
                                   dlio_stream_t stream) {
  const bool with_wgrad = dw_ih0 != nullptr;                       // NULL: the caller runs dlio_lstm_layer_wgrad itself (another stream)
  if (!dhs || !x || !hp || !gates || !cs || !w_ih0 || !w_hh0 || !dgates || !ws || (D == 2 && (!w_ih1 || !w_hh1)) ||
      (with_wgrad && (!dw_hh0 || (D == 2 && (!dw_ih1 || !dw_hh1)))))
    return DLIO_EINVAL;
  if (!dlio_lstm_layer_ok(T, B, I, H, D) || ldx < I || lddhs < D * H || (ldx & 3) || (dx && (lddx < I || (lddx & 3))))
    return DLIO_EUNSUP;
  if (ws_bytes < dlio_lstm_layer_ws_bytes(T, B, I, H, D)) return DLIO_EWS;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(hp) | reinterpret_cast<uintptr_t>(w_ih0) |
       reinterpret_cast<uintptr_t>(w_hh0) | reinterpret_cast<uintptr_t>(w_ih1) | reinterpret_cast<uintptr_t>(w_hh1) |
       reinterpret_cast<uintptr_t>(dw_ih0) | reinterpret_cast<uintptr_t>(dw_hh0) | reinterpret_cast<uintptr_t>(dw_ih1) |
       reinterpret_cast<uintptr_t>(dw_hh1) | reinterpret_cast<uintptr_t>(dx) | reinterpret_cast<uintptr_t>(dgates)) & 15)
    return DLIO_EUNSUP;
  hipStream_t s = as_stream(stream);
  const int rows = B * T, G = 4 * H;
  const int NS = slabs_of(G);
  float* dhp = reinterpret_cast<float*>(ws);                       // [D][NS][B][H]
  float* dccur = dhp + (size_t)D * NS * B * H;                     // [D][B][H]
  float* dxp = dccur + (size_t)D * B * H;                          // [D][NS][rows][I]
  const int cgrid = cdiv(D * B * H, 256);
  float* dg0 = dgates;
  float* dg1 = dgates + (size_t)rows * G;
  for (int step = 0; step < T; ++step) {
    hipLaunchKernelGGL(lstm_cell_bwd2_kernel, dim3(cgrid), dim3(256), 0, s, dhs, lddhs, (const float*)dhp, NS, dccur, gates, cs,
                       dgates, step, T, B, H, D);
    if (step == T - 1) break;
    // dh of the step before (in forward order): dgates of THIS step's time rows times W_hh; rows of one time are T * G apart
    const float* z0 = dg0 + (size_t)(T - 1 - step) * G;
    const float* z1 = dg1 + (size_t)step * G;
    hipLaunchKernelGGL((gemv_t_splitn_kernel<8>), dim3(cdiv(H / 4, 64), NS, D), dim3(256), 0, s, z0, z1, T * G, w_hh0, w_hh1, dhp,
                       B, B, 0, G, H);
  }
  if (with_wgrad)
    launch_wgrad(s, dgates, x, ldx, hp, dw_ih0, dw_hh0, db_ih0, db_hh0, dw_ih1, dw_hh1, db_ih1, db_hh1, accumulate, T, B, I, H, D);
  if (dx) {
    for (int m0 = 0; m0 < rows; m0 += 16) {
      const int M = rows - m0 < 16 ? rows - m0 : 16;
      const float* z0 = dg0 + (size_t)m0 * G;
      const float* z1 = dg1 + (size_t)m0 * G;
      if (M <= 8)
        hipLaunchKernelGGL((gemv_t_splitn_kernel<8>), dim3(cdiv(I / 4, 64), NS, D), dim3(256), 0, s, z0, z1, G, w_ih0, w_ih1, dxp, M,
                           rows, m0, G, I);
      else
        hipLaunchKernelGGL((gemv_t_splitn_kernel<16>), dim3(cdiv(I / 4, 64), NS, D), dim3(256), 0, s, z0, z1, G, w_ih0, w_ih1, dxp,
                           M, rows, m0, G, I);
    }
    hipLaunchKernelGGL(slab_reduce_kernel, dim3(cdiv(rows * (I / 4), 256)), dim3(256), 0, s, (const float*)dxp, dx, lddx, rows, I,
                       D * NS);
  }
  return dlio_check_launch();
}
