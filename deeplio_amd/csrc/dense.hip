// Dense (nn.Linear-shaped) layers with skinny M (M = B*S = 16 on the headline config; the
// weights, not the activations, are the HBM traffic), activations, dropout and small
// elementwise helpers.  Weight-streaming kernels: every weight byte is read once per
// 8/16-row tile of M in 1 KiB wave-wide float4 requests; no atomics, fixed summation
// order (deterministic).
//
// Replaces nn.Linear / F.relu / F.leaky_relu / torch.sigmoid / nn.Dropout at
// lidar_feat_nets.py:66,94-97, pointseg_modules.py:209-214, imu_feat_nets.py:27-49,
// fusion_nets.py:59-73, odom_feat_nets.py:18-37, deeplio_nets.py:57-90 and the input /
// recurrent projections of nn.LSTM / nn.GRU (imu_feat_nets.py:63-70, odom_feat_nets.py:61-68).
#include "common.h"
#include <stdlib.h>

namespace {

__device__ __forceinline__ float act_fwd(float v, int act) {
  switch (act) {
    case 1: return fmaxf(v, 0.f);
    case 2: return v > 0.f ? v : 0.01f * v;
    case 3: return 1.0f / (1.0f + expf(-v));
    case 4: return tanhf(v);
    default: return v;
  }
}

// ---- forward: one wave per output feature n, 8 rows of M per pass -------------
constexpr int MT = 8;

__global__ __launch_bounds__(256) void linear_fwd_kernel(
    const float* __restrict__ x, int ldx, const float* __restrict__ w, const float* __restrict__ b,
    const float* __restrict__ addend, int ldadd, float* __restrict__ y, int ldy, int M, int N,
    int K, int act) {
  const int lane = threadIdx.x & 63;
  const int wave_in_blk = threadIdx.x >> 6;
  const int waves_total = gridDim.x * 4;
  const bool vec = ((K & 3) == 0) && ((ldx & 3) == 0);
  for (int n = blockIdx.x * 4 + wave_in_blk; n < N; n += waves_total) {
    const float* wr = w + (size_t)n * K;
    const float bv = b ? b[n] : 0.f;
    for (int m0 = 0; m0 < M; m0 += MT) {
      float acc[MT];
#pragma unroll
      for (int j = 0; j < MT; ++j) acc[j] = 0.f;
      if (vec) {
        for (int k = lane * 4; k < K; k += 256) {
          const float4 wv = *reinterpret_cast<const float4*>(wr + k);
#pragma unroll
          for (int j = 0; j < MT; ++j) {
            if (m0 + j < M) {
              const float4 xv = *reinterpret_cast<const float4*>(x + (size_t)(m0 + j) * ldx + k);
              acc[j] += wv.x * xv.x + wv.y * xv.y + wv.z * xv.z + wv.w * xv.w;
            }
          }
        }
      } else {
        for (int k = lane; k < K; k += 64) {
          const float wv = wr[k];
#pragma unroll
          for (int j = 0; j < MT; ++j)
            if (m0 + j < M) acc[j] += wv * x[(size_t)(m0 + j) * ldx + k];
        }
      }
#pragma unroll
      for (int j = 0; j < MT; ++j) acc[j] = wave_sum(acc[j]);
      if (lane == 0) {
#pragma unroll
        for (int j = 0; j < MT; ++j) {
          const int m = m0 + j;
          if (m < M) {
            float v = acc[j] + bv;
            if (addend) v += addend[(size_t)m * ldadd + n];
            y[(size_t)m * ldy + n] = act_fwd(v, act);
          }
        }
      }
    }
  }
}

// ---- skinny-M forward with the activations in LDS (the recurrent / projection layers of the odometry LSTM, M = 8 / 16 rows
// against 4096 x 1024..2048 weights): the wave-per-feature kernel above re-reads all M rows of x from L1 / L2 for every
// output feature (8x the weight bytes); here a workgroup stages x once ([M][K] floats, <= 128 KB), a wave streams the
// weight rows of FOUR features at a time (four 16-byte loads in flight per lane and k block) and the 4 MROWS partial sums
// of a lane are reduced over the wave by halving exchanges (reduce-scatter: 4 MROWS - 1 shuffles instead of 6 per value;
// lane i ends with value i), fixed order.
template <int MROWS>
__global__ __launch_bounds__(256) void linear_fwd_skinny_kernel(
    const float* __restrict__ x, int ldx, const float* __restrict__ w, const float* __restrict__ b,
    const float* __restrict__ addend, int ldadd, float* __restrict__ y, int ldy, int M, int N, int K, int act) {
  extern __shared__ __attribute__((aligned(16))) float xs[];          // [MROWS][K]
  constexpr int NV = 4 * MROWS;                                         // values per lane: (feature u, row m)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int K4 = K >> 2;
  const int groups = (N + 3) >> 2;
  // weight rows: two k steps (2 x 4 features x 16 bytes per lane) per register set, the next set prefetched while the
  // current one is multiplied; the first set of the first group is in flight while x is staged
  float4 wa[2][4], wb[2][4];
  const float* wr[4];
  auto rows_of = [&](int g) {
#pragma unroll
    for (int u = 0; u < 4; ++u) wr[u] = w + (size_t)min(g * 4 + u, N - 1) * K;
  };
  auto load_w = [&](float4 (&wv)[2][4], int kb) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int k = kb + c * 256 + lane * 4;
#pragma unroll
      for (int u = 0; u < 4; ++u)
        wv[c][u] = k < K ? *reinterpret_cast<const float4*>(wr[u] + k) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  int g = blockIdx.x * 4 + wave;
  if (g < groups) { rows_of(g); load_w(wa, 0); }
  for (int i = threadIdx.x; i < MROWS * K4; i += 256) {
    const int m = i / K4, k4 = i - m * K4;
    reinterpret_cast<float4*>(xs)[i] = m < M ? *reinterpret_cast<const float4*>(x + (size_t)m * ldx + 4 * k4)
                                             : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  __syncthreads();
  float v[NV];
  auto mul_w = [&](const float4 (&wv)[2][4], int kb) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int k = min(kb + c * 256 + lane * 4, K - 4);   // past K the weights are zero, the read stays inside xs
#pragma unroll
      for (int m = 0; m < MROWS; ++m) {
        const float4 xv = *reinterpret_cast<const float4*>(xs + m * K + k);
#pragma unroll
        for (int u = 0; u < 4; ++u)
          v[u * MROWS + m] += (wv[c][u].x * xv.x + wv[c][u].y * xv.y) + (wv[c][u].z * xv.z + wv[c][u].w * xv.w);
      }
    }
  };
  bool first = true;
  for (; g < groups; g += gridDim.x * 4) {
    const int n0 = g * 4;
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = 0.f;
    if (!first) { rows_of(g); load_w(wa, 0); }
    first = false;
    for (int kb = 0; kb < K; kb += 1024) {
      if (kb + 512 < K) load_w(wb, kb + 512);
      mul_w(wa, kb);
      if (kb + 512 < K) {
        if (kb + 1024 < K) load_w(wa, kb + 1024);
        mul_w(wb, kb + 512);
      }
    }
    // reduce-scatter over the wave: after the step with offset o a lane keeps the half of its values selected by its
    // bit o; NV = 64: lane i ends with the wave sum of value i; NV = 32: offsets 16..1, then one pair sum over offset 32
    int cnt = NV;
#pragma unroll
    for (int o = (NV == 64 ? 32 : 16); o >= 1; o >>= 1) {
      const int hcnt = cnt >> 1;
      const bool up = (lane & o) != 0;
#pragma unroll
      for (int i = 0; i < NV / 2; ++i) {
        if (i < hcnt) {
          const float send = up ? v[i] : v[i + hcnt], keep = up ? v[i + hcnt] : v[i];
          v[i] = keep + __shfl_xor(send, o, 64);
        }
      }
      cnt = hcnt;
    }
    float r = v[0];
    if (NV == 32) r += __shfl_xor(r, 32, 64);
    const int idx = lane & (NV - 1), u = idx / MROWS, m = idx - u * MROWS, n = n0 + u;
    if ((NV == 64 || lane < 32) && n < N && m < M) {
      float o = r + (b ? b[n] : 0.f);
      if (addend) o += addend[(size_t)m * ldadd + n];
      y[(size_t)m * ldy + n] = act_fwd(o, act);
    }
  }
}

__global__ void act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                               float* __restrict__ dz, int64_t n, int act) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const float g = dy[i], o = y[i];
    float r;
    switch (act) {
      case 1: r = o > 0.f ? g : 0.f; break;
      case 2: r = o > 0.f ? g : 0.01f * g; break;
      case 3: r = g * o * (1.f - o); break;
      case 4: r = g * (1.f - o * o); break;
      default: r = g;
    }
    dz[i] = r;
  }
}

// ---- data gradient: block = 64 columns of K x all N, 4 waves interleave n --------
constexpr int BM = 16;

__global__ __launch_bounds__(256) void linear_bwd_data_kernel(
    const float* __restrict__ dz, int lddz, const float* __restrict__ w, float* __restrict__ dx,
    int lddx, int M, int N, int K, int accumulate) {
  // dx[m][k] = sum_n dz[m][n] w[n][k].  A block owns 64 columns k and BM rows m; the dz tile of
  // a 256-wide n chunk is staged in LDS (read back as broadcast float4), the 4 waves take 64
  // consecutive n each and keep 8 coalesced weight loads in flight; partials are combined in a
  // fixed order.
  constexpr int NC = 256;
  __shared__ __attribute__((aligned(16))) float sdz[BM][NC];
  __shared__ float red[4][BM][64];
  const int tx = threadIdx.x & 63;
  const int ty = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int k = blockIdx.x * 64 + tx;
  const int kc = k < K ? k : K - 1;
  const int m0 = blockIdx.y * BM;
  float acc[BM];
#pragma unroll
  for (int j = 0; j < BM; ++j) acc[j] = 0.f;
  for (int c0 = 0; c0 < N; c0 += NC) {
    __syncthreads();
    for (int i = threadIdx.x; i < BM * NC; i += 256) {
      const int j = i / NC, n = c0 + (i - j * NC);
      sdz[j][i - j * NC] = (m0 + j < M && n < N) ? dz[(size_t)(m0 + j) * lddz + n] : 0.f;
    }
    __syncthreads();
    const int nb = ty * 64;
#pragma unroll 1
    for (int g = 0; g < 64; g += 8) {
      float wv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int n = c0 + nb + g + u;
        wv[u] = n < N ? w[(size_t)n * K + kc] : 0.f;
      }
#pragma unroll
      for (int j = 0; j < BM; ++j) {
        const float4 d0 = *reinterpret_cast<const float4*>(&sdz[j][nb + g]);
        const float4 d1 = *reinterpret_cast<const float4*>(&sdz[j][nb + g + 4]);
        acc[j] += d0.x * wv[0]; acc[j] += d0.y * wv[1]; acc[j] += d0.z * wv[2]; acc[j] += d0.w * wv[3];
        acc[j] += d1.x * wv[4]; acc[j] += d1.y * wv[5]; acc[j] += d1.z * wv[6]; acc[j] += d1.w * wv[7];
      }
    }
  }
#pragma unroll
  for (int j = 0; j < BM; ++j) red[ty][j][tx] = acc[j];
  __syncthreads();
  // fixed-order sum of the 4 partials; 256 threads cover BM*64 = 1024 outputs
  for (int o = threadIdx.x; o < BM * 64; o += 256) {
    const int j = o >> 6, c = o & 63;
    const int kk = blockIdx.x * 64 + c, m = m0 + j;
    if (kk < K && m < M) {
      float v = ((red[0][j][c] + red[1][j][c]) + red[2][j][c]) + red[3][j][c];
      float* p = dx + (size_t)m * lddx + kk;
      *p = accumulate ? *p + v : v;
    }
  }
}

// ---- data gradient, weight-streaming form for large N*K (the 1024-wide odometry LSTM):
// a block owns 256 columns k (one float4 per lane) and a slab of `per` weight rows; its four waves take the slab's rows in
// interleaved groups of eight (eight coalesced 16-byte row loads in flight per lane, the next group prefetched while the
// current one is multiplied), the dz slab sits in LDS (broadcast reads), the four wave partials are combined through LDS in
// a fixed order and ONE partial per block goes to the caller's workspace ([nsplit][M][K]); linear_bwd_data_reduce_kernel
// sums the slabs in slab order.  Each weight byte is read once; the workspace traffic is nsplit*M*K*8 bytes.
constexpr int DG_PER_MAX = 128;
template <int MROWS>
__global__ __launch_bounds__(256) void linear_bwd_data_split_kernel(
    const float* __restrict__ dz, int lddz, const float* __restrict__ w, float* __restrict__ part,
    int M, int N, int K, int m0, int nsplit) {
  __shared__ __attribute__((aligned(16))) float sdz[MROWS][DG_PER_MAX];
  __shared__ float4 red[3][MROWS][64];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int k = (blockIdx.x * 64 + lane) * 4;
  const int kc = k < K ? k : K - 4;
  const int sp = blockIdx.y;
  const int per = (N + nsplit - 1) / nsplit;           // <= DG_PER_MAX (launcher)
  const int n_lo = sp * per, n_hi = min(N, n_lo + per);
  float4 wa[8], wb[8];
  auto load_rows = [&](float4 (&wv)[8], int r) {        // r = slab-relative first row of the group
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int n = n_lo + r + u;
      wv[u] = n < n_hi ? *reinterpret_cast<const float4*>(w + (size_t)n * K + kc) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  load_rows(wa, wave * 8);                              // in flight while the dz slab is staged
  for (int i = threadIdx.x; i < MROWS * DG_PER_MAX; i += 256) {
    const int j = i / DG_PER_MAX, r = i - j * DG_PER_MAX, n = n_lo + r;
    sdz[j][r] = (m0 + j < M && n < n_hi) ? dz[(size_t)(m0 + j) * lddz + n] : 0.f;
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
      const float d[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        acc[j].x += d[u] * wv[u].x; acc[j].y += d[u] * wv[u].y; acc[j].z += d[u] * wv[u].z; acc[j].w += d[u] * wv[u].w;
      }
    }
  };
  for (int r = wave * 8; r < per; r += 64) {             // two groups per trip (register double buffer, static indices)
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
    const int rows = min(MROWS, M - m0);
#pragma unroll
    for (int j = 0; j < MROWS; ++j) {
      if (j < rows) {
        const float4 a = red[0][j][lane], b = red[1][j][lane], c = red[2][j][lane];
        float4 o;
        o.x = ((acc[j].x + a.x) + b.x) + c.x; o.y = ((acc[j].y + a.y) + b.y) + c.y;
        o.z = ((acc[j].z + a.z) + b.z) + c.z; o.w = ((acc[j].w + a.w) + b.w) + c.w;
        *reinterpret_cast<float4*>(part + ((size_t)sp * rows + j) * K + k) = o;
      }
    }
  }
}

__global__ void linear_bwd_data_reduce_kernel(const float* __restrict__ part, float* __restrict__ dx,
                                              int lddx, int rows, int K, int m0, int nsplit,
                                              int accumulate) {
  const int total = rows * K;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int j = i / K, k = i - j * K;
    float s = 0.f;
    int sp = 0;
    for (; sp + 8 <= nsplit; sp += 8) {        // 8 loads in flight, same summation order
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = part[((size_t)(sp + u) * rows + j) * K + k];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; sp < nsplit; ++sp) s += part[((size_t)sp * rows + j) * K + k];
    float* p = dx + (size_t)(m0 + j) * lddx + k;
    *p = accumulate ? *p + s : s;
  }
}

static int bwd_data_nsplit(int N, int K) {
  if ((K & 3) != 0 || (int64_t)N * K < (1 << 18)) return 0;   // small problems: direct kernel
  const int kblocks = (K / 4 + 63) / 64;
  int ns = (2 * dlio_num_cus()) / kblocks;                     // about two workgroups per CU
  if (ns > N / 32) ns = N / 32;                                // at least eight rows per wave
  const int floor_ns = (N + DG_PER_MAX - 1) / DG_PER_MAX;      // slab fits the LDS tile
  if (ns < floor_ns) ns = floor_ns;
  return ns < 2 ? 0 : ns;
}

// ---- weight gradient: thread per (4 n, k) --------------------------------------
__global__ __launch_bounds__(256) void linear_bwd_weight_kernel(
    const float* __restrict__ dz, int lddz, const float* __restrict__ x, int ldx,
    float* __restrict__ dw, int M, int N, int K, int accumulate) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  const int n0 = blockIdx.y * 4;
  if (k >= K) return;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int m = 0; m < M; ++m) {
    const float xv = x[(size_t)m * ldx + k];
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (n0 + j < N) acc[j] += dz[(size_t)m * lddz + n0 + j] * xv;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (n0 + j < N) {
      float* p = dw + (size_t)(n0 + j) * K + k;
      *p = accumulate ? *p + acc[j] : acc[j];
    }
  }
}

// 16-byte form (K % 4 == 0): thread per (4 n, 4 k) -- the x row is one float4 load per sample and serves four weight rows,
// dW is read (accumulate) and written as float4
__global__ __launch_bounds__(256) void linear_bwd_weight_v4_kernel(
    const float* __restrict__ dz, int lddz, const float* __restrict__ x, int ldx,
    float* __restrict__ dw, float* __restrict__ db, int M, int N, int K, int accumulate) {
  const int k = (blockIdx.x * 256 + threadIdx.x) * 4;
  const int n0 = blockIdx.y * 4;
  // the bias gradient of the block's four rows rides along (first k block, one lane per row; same m order as
  // col_sum_kernel): one launch less per weight in the serial middle of the step
  if (db && blockIdx.x == 0 && threadIdx.x >= 252 && n0 + (int)threadIdx.x - 252 < N) {
    const int n = n0 + (int)threadIdx.x - 252;
    float sb = 0.f;
    for (int m = 0; m < M; ++m) sb += dz[(size_t)m * lddz + n];
    db[n] = accumulate ? db[n] + sb : sb;
  }
  if (k >= K) return;
  float4 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int m = 0; m < M; ++m) {
    const float4 xv = *reinterpret_cast<const float4*>(x + (size_t)m * ldx + k);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float a = n0 + j < N ? dz[(size_t)m * lddz + n0 + j] : 0.f;
      acc[j].x += a * xv.x; acc[j].y += a * xv.y; acc[j].z += a * xv.z; acc[j].w += a * xv.w;
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (n0 + j < N) {
      float4* p = reinterpret_cast<float4*>(dw + (size_t)(n0 + j) * K + k);
      if (accumulate) { const float4 o = *p; acc[j].x += o.x; acc[j].y += o.y; acc[j].z += o.z; acc[j].w += o.w; }
      *p = acc[j];
    }
  }
}

__global__ void col_sum_kernel(const float* __restrict__ dz, int lddz, float* __restrict__ db,
                               int M, int N, int accumulate) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  float s = 0.f;
  for (int m = 0; m < M; ++m) s += dz[(size_t)m * lddz + n];
  db[n] = accumulate ? db[n] + s : s;
}

__global__ void ew_binary_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                 float* __restrict__ y, int64_t n, int op) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const float u = a[i], v = b[i];
    y[i] = op == 0 ? u + v : (op == 1 ? u - v : (op == 2 ? u * v : fmaxf(u + v, 0.f)));
  }
}

// the same on float4 with four loads in flight per operand and (optionally) the largest |y| written: the tail of a
// BasicBlock, relu(out + identity) over 16-268 MB tensors, whose consumer is a two-piece fp16 3x3 convolution
__global__ __launch_bounds__(256) void ew_binary_v4_kernel(const float4* __restrict__ a, const float4* __restrict__ b,
                                                           float4* __restrict__ y, int64_t n4, int op, float* __restrict__ amax_out) {
  constexpr int UN = 4;
  __shared__ unsigned s_amax;
  float amax = 0.f;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += UN * stride) {
    float4 u[UN], v[UN];
#pragma unroll
    for (int k = 0; k < UN; ++k) {
      const int64_t j = i + k * stride < n4 ? i + k * stride : n4 - 1;
      u[k] = a[j]; v[k] = b[j];
    }
#pragma unroll
    for (int k = 0; k < UN; ++k) {
      if (i + k * stride >= n4) continue;
      const float ue[4] = {u[k].x, u[k].y, u[k].z, u[k].w}, ve[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
      float o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e)
        o[e] = op == 0 ? ue[e] + ve[e] : (op == 1 ? ue[e] - ve[e] : (op == 2 ? ue[e] * ve[e] : fmaxf(ue[e] + ve[e], 0.f)));
      y[i + k * stride] = make_float4(o[0], o[1], o[2], o[3]);
      amax = amax4(amax, o[0], o[1], o[2], o[3]);
    }
  }
  if (amax_out) block_amax_commit(amax, amax_out, &s_amax);
}

// y[g][c] = sum_r x[g][r][c]   /   dx[g][r][c] = dy[g][c]
__global__ void seg_sum_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int groups,
                                   int rows, int cols) {
  const int64_t total = (int64_t)groups * cols;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int g = (int)(i / cols), c = (int)(i - (int64_t)g * cols);
    const float* p = x + (size_t)g * rows * cols + c;
    float s = 0.f;
    for (int r = 0; r < rows; ++r) s += p[(size_t)r * cols];
    y[i] = s;
  }
}
__global__ void seg_sum_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int groups,
                                   int rows, int cols) {
  const int64_t total = (int64_t)groups * rows * cols;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % cols);
    const int g = (int)(i / ((int64_t)rows * cols));
    dx[i] = dy[(size_t)g * cols + c];
  }
}

__global__ void ew_scale_kernel(const float* __restrict__ a, float alpha, float* __restrict__ y,
                                int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    y[i] = alpha * a[i];
}

__global__ void copy2d_kernel(const float* __restrict__ src, int lds, float* __restrict__ dst,
                              int ldd, int rows, int cols, int accumulate) {
  const int64_t total = (int64_t)rows * cols;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / cols), c = (int)(i - (int64_t)r * cols);
    const float v = src[(size_t)r * lds + c];
    float* p = dst + (size_t)r * ldd + c;
    *p = accumulate ? *p + v : v;
  }
}

// Philox4x32-10 (Salmon et al. 2011), counter = (offset + i/4), key = seed
__device__ __forceinline__ void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}

__global__ void dropout_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                   uint8_t* __restrict__ mask, int64_t n, float p, uint64_t seed,
                                   uint64_t offset) {
  const float inv = 1.0f / (1.0f - p);
  const int64_t groups = (n + 3) >> 2;
  for (int64_t g = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; g < groups;
       g += (int64_t)gridDim.x * blockDim.x) {
    const uint64_t ctr = offset + (uint64_t)g;
    uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u};
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t i = g * 4 + j;
      if (i < n) {
        const float u = (float)(c[j] >> 8) * (1.0f / 16777216.0f);  // [0,1)
        const uint8_t keep = u >= p ? 1 : 0;
        mask[i] = keep;
        y[i] = keep ? x[i] * inv : 0.f;
      }
    }
  }
}

__global__ void dropout_bwd_kernel(const float* __restrict__ dy, const uint8_t* __restrict__ mask,
                                   float* __restrict__ dx, int64_t n, float p) {
  const float inv = 1.0f / (1.0f - p);
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    dx[i] = mask[i] ? dy[i] * inv : 0.f;
}

__global__ void nonfinite_kernel(const float* __restrict__ x, int64_t n, int32_t* flag) {
  bool bad = false;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const float v = x[i];
    bad |= !(fabsf(v) <= 3.402823466e38f);  // NaN or Inf
  }
  if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}


// ---- tall-M dense layers on the fp32 MFMA (v_mfma_f32_32x32x2_f32) --------------------------------
// The IMU branch calls nn.Linear-shaped ops with M = B*T = 400..1600 rows (input projections of the
// bi-LSTM / GRU over whole windows, the ImuFeatFC MLP): GEMMs of 0.05-0.2 GFLOP that the weight-streaming
// kernels above (built for M = B*S = 16) execute at ~1 TFLOP/s, 50-90 us per launch, ~170 launches per
// step.  Same results to fp32 round-off (an fmaf chain in k order per output), fixed summation order.
//
// forward: y[M][N] = act(x[M][K] W[N][K]^T + b + addend).  Workgroup = 2 x 2 waves = 64 x 64 outputs; a
// lane (row l&31, half l>>5) reads FOUR consecutive k of its x row / W row per 16-byte load and feeds
// four MFMA k-steps (k slot `half` of step e = element 8j + 4*half + e for both operands).
__global__ __launch_bounds__(256) void linear_fwd_mfma_kernel(
    const float* __restrict__ x, int ldx, const float* __restrict__ w, const float* __restrict__ b,
    const float* __restrict__ addend, int ldadd, float* __restrict__ y, int ldy, int M, int N, int K, int act,
    int kvec) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, half = lane >> 5;
  const int m0 = blockIdx.y * 64 + (wave >> 1) * 32, n0 = blockIdx.x * 64 + (wave & 1) * 32;
  if (m0 >= M || n0 >= N) return;
  const float* xr = x + (size_t)min(m0 + l31, M - 1) * ldx;
  const float* wr = w + (size_t)min(n0 + l31, N - 1) * K;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  if (kvec) {
#pragma unroll 2
    for (int k0 = 4 * half; k0 < K; k0 += 8) {
      const float4 a = *reinterpret_cast<const float4*>(xr + k0);
      const float4 c = *reinterpret_cast<const float4*>(wr + k0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, c.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, c.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, c.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, c.w, acc, 0, 0, 0);
    }
  } else {
    for (int k0 = 0; k0 < K; k0 += 2) {
      const int k = k0 + half;
      const float a = k < K ? xr[k] : 0.f, c = k < K ? wr[k] : 0.f;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, c, acc, 0, 0, 0);
    }
  }
  const int n = n0 + l31;
  if (n >= N) return;
  const float bv = b ? b[n] : 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * half;
    if (m >= M) continue;
    float v = acc[r] + bv;
    if (addend) v += addend[(size_t)m * ldadd + n];
    y[(size_t)m * ldy + n] = act_fwd(v, act);
  }
}

// weight gradient: dW[N][K] (+)= dz[M][N]^T x[M][K], db[N] (+)= column sums of dz.  Workgroup = one
// 32 x 32 tile of dW; its 4 waves take every 4th pair of rows (k slot `half` = row 2s + half), partial
// tiles summed through LDS in wave order; the k-tile-0 workgroups also produce db.
__global__ __launch_bounds__(256) void linear_wgrad_mfma_kernel(
    const float* __restrict__ dz, int lddz, const float* __restrict__ x, int ldx, float* __restrict__ dw,
    float* __restrict__ db, int M, int N, int K, int accumulate) {
  __shared__ float red[16 * 64];
  __shared__ float bred[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, half = lane >> 5;
  const int k0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
  const int n = n0 + l31, k = k0 + l31;
  const bool vn = n < N, vk = k < K;
  const float* ap = dz + (vn ? n : 0);
  const float* bp = x + (vk ? k : 0);
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float bsum = 0.f;
  const int steps = (M + 1) >> 1;
  int s = wave;
  for (; s + 12 < steps; s += 16) {                 // 4 row pairs per iteration: 8 loads in flight
    float a[4], c[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int m = 2 * (s + 4 * u) + half;
      const bool vm = m < M;
      a[u] = (vm && vn) ? ap[(size_t)m * lddz] : 0.f;
      c[u] = (vm && vk) ? bp[(size_t)m * ldx] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], c[u], acc, 0, 0, 0);
      bsum += a[u];
    }
  }
  for (; s < steps; s += 4) {
    const int m = 2 * s + half;
    const bool vm = m < M;
    const float a = (vm && vn) ? ap[(size_t)m * lddz] : 0.f;
    const float c = (vm && vk) ? bp[(size_t)m * ldx] : 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, c, acc, 0, 0, 0);
    bsum += a;
  }
  bred[wave][lane] = bsum;
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (w == 0) red[r * 64 + lane] = acc[r];
        else red[r * 64 + lane] += acc[r];
      }
    }
    __syncthreads();
  }
  // D tile: row (n) = (r & 3) + 8 (r >> 2) + 4 half, col (k) = lane & 31
  for (int idx = threadIdx.x; idx < 32 * 32; idx += 256) {
    const int row = idx >> 5, col = idx & 31;
    if (n0 + row >= N || k0 + col >= K) continue;
    const int hf = (row >> 2) & 1, r = (row & 3) + 4 * (row >> 3);
    const float v = red[r * 64 + hf * 32 + col];
    float* p = dw + (size_t)(n0 + row) * K + k0 + col;
    *p = accumulate ? *p + v : v;
  }
  if (db && blockIdx.x == 0 && threadIdx.x < 32 && n0 + threadIdx.x < N) {
    const int j = threadIdx.x;
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) t += bred[w][j] + bred[w][32 + j];
    db[n0 + j] = accumulate ? db[n0 + j] + t : t;
  }
}

// ---- the small layers of the step's serial middle as single launches ---------------------------------------------------------
// DeepLIOFusionSoft (fusion_nets.py:64-75): cat = [a | b]; s1 = sigmoid(cat W1^T + b1), s2 = sigmoid(cat W2^T + b2);
// out = [a s1 | b s2] -- ten launches as cat + 2 x linear + 2 x multiply + cat.  One wave per output column j (rows of the
// stacked [W1; W2], F = Fa + Fb columns each), all R sample rows in passes of 16.
__global__ __launch_bounds__(256) void soft_fusion_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                              const float* __restrict__ w1, const float* __restrict__ b1,
                                                              const float* __restrict__ w2, const float* __restrict__ b2,
                                                              float* __restrict__ out, float* __restrict__ gate, int R, int Fa,
                                                              int Fb, int lda, int ldb) {
  const int F = Fa + Fb;
  const int lane = threadIdx.x & 63;
  const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (j >= F) return;
  const float* __restrict__ wr = j < Fa ? w1 + (size_t)j * F : w2 + (size_t)(j - Fa) * F;
  const float bias = j < Fa ? (b1 ? b1[j] : 0.f) : (b2 ? b2[j - Fa] : 0.f);
  for (int m0 = 0; m0 < R; m0 += 16) {
    float acc[16];
#pragma unroll
    for (int mm = 0; mm < 16; ++mm) acc[mm] = 0.f;
    for (int k = lane; k < F; k += 64) {
      const float wv = wr[k];
      // (one pointer / stride per column, rows past R re-read the last row: 16 independent loads in flight, no branch)
      const float* __restrict__ src = k < Fa ? a + k : b + (k - Fa);
      const int st = k < Fa ? lda : ldb;
      float xv[16];
#pragma unroll
      for (int mm = 0; mm < 16; ++mm) xv[mm] = src[(size_t)min(m0 + mm, R - 1) * st];
#pragma unroll
      for (int mm = 0; mm < 16; ++mm) acc[mm] += wv * xv[mm];
    }
#pragma unroll
    for (int mm = 0; mm < 16; ++mm) acc[mm] = wave_sum(acc[mm]);
    if (lane < 16 && m0 + lane < R) {
      const int m = m0 + lane;
      float v = 0.f;
#pragma unroll
      for (int mm = 0; mm < 16; ++mm) v = lane == mm ? acc[mm] : v;
      const float sg = 1.0f / (1.0f + expf(-(v + bias)));
      const float src = j < Fa ? a[(size_t)m * lda + j] : b[(size_t)m * ldb + (j - Fa)];
      out[(size_t)m * F + j] = src * sg;
      gate[(size_t)m * F + j] = sg;
    }
  }
}

// backward of the above: grid F / 16; workgroup g owns rows j in [16 g, 16 g + 16) of the stacked weight (their gradient:
// dW[j][k] = sum_m dpre[m][j] cat[m][k], db[j]) and columns k in the same range of d cat (= sum_j dpre[m][j] W[j][k] +
// dout gate); dpre = dout src gate (1 - gate) is formed by every workgroup in LDS (16 x F floats per pass of 16 samples)
template <int FMAX>
__global__ __launch_bounds__(256) void soft_fusion_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ a,
                                                              const float* __restrict__ b, const float* __restrict__ gate,
                                                              const float* __restrict__ w1, const float* __restrict__ w2,
                                                              float* __restrict__ da, float* __restrict__ db_,
                                                              float* __restrict__ dw1, float* __restrict__ dbias1,
                                                              float* __restrict__ dw2, float* __restrict__ dbias2, int R, int Fa,
                                                              int Fb, int lda, int ldb, int accumulate) {
  __shared__ float sdp[16][FMAX], scat[16][FMAX];
  const int F = Fa + Fb;
  const int g0 = blockIdx.x * 16;
  const int t = threadIdx.x, jj = t >> 4, kg = t & 15;
  constexpr int NI = FMAX / 16;
  float acc[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) acc[i] = 0.f;
  float bsum = 0.f;
  for (int m0 = 0; m0 < R; m0 += 16) {
    __syncthreads();
    for (int i = t; i < 16 * F; i += 256) {
      const int mm = i / F, c = i - mm * F, m = m0 + mm;
      float dp = 0.f, cv = 0.f;
      if (m < R) {
        cv = c < Fa ? a[(size_t)m * lda + c] : b[(size_t)m * ldb + (c - Fa)];
        const float sg = gate[(size_t)m * F + c];
        dp = dout[(size_t)m * F + c] * cv * sg * (1.f - sg);
      }
      sdp[mm][c] = dp;
      scat[mm][c] = cv;
    }
    __syncthreads();
    // weight-gradient rows g0 + jj
    const int j = g0 + jj;
    if (j < F) {
#pragma unroll
      for (int mm = 0; mm < 16; ++mm) {
        const float dp = sdp[mm][j];
        if (kg == 0) bsum += dp;
#pragma unroll
        for (int i = 0; i < NI; ++i)
          if (kg + 16 * i < F) acc[i] += dp * scat[mm][kg + 16 * i];
      }
    }
    // d cat, columns g0 + kg, sample m0 + jj
    const int k = g0 + kg, m = m0 + jj;
    if (k < F && m < R) {
      float s = 0.f;
      int jx = 0;
      for (; jx + 8 <= F; jx += 8) {                 // eight weight loads in flight (a column walk: 64 B per 16 lanes and row)
        float wv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int r = jx + u;
          wv[u] = r < Fa ? w1[(size_t)r * F + k] : w2[(size_t)(r - Fa) * F + k];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) s += sdp[jj][jx + u] * wv[u];
      }
      for (; jx < F; ++jx) s += sdp[jj][jx] * (jx < Fa ? w1[(size_t)jx * F + k] : w2[(size_t)(jx - Fa) * F + k]);
      const float v = s + dout[(size_t)m * F + k] * gate[(size_t)m * F + k];
      if (k < Fa) da[(size_t)m * Fa + k] = v;
      else db_[(size_t)m * Fb + (k - Fa)] = v;
    }
  }
  const int j = g0 + jj;
  if (j < F) {
    float* __restrict__ dwr = j < Fa ? dw1 + (size_t)j * F : dw2 + (size_t)(j - Fa) * F;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int k = kg + 16 * i;
      if (k < F) dwr[k] = accumulate ? dwr[k] + acc[i] : acc[i];
    }
    if (kg == 0) {
      float* __restrict__ p = j < Fa ? dbias1 + j : dbias2 + (j - Fa);
      *p = accumulate ? *p + bsum : bsum;
    }
  }
}

// the same on v_mfma_f32_16x16x4_f32 (F and Fa multiples of 16): the scalar kernel above walks a 256-long dependent dot product per
// thread (86 us beside the early optimizer sweep, ~30 alone); here a workgroup's four waves form
//   dW[j][k]  = sum_m dpre[m][j] cat[m][k]   as 16 x 16 tiles over k with the 16 samples as the MFMA's k dimension (4 MFMAs a tile),
//   dcat[m][k] = sum_j dpre[m][j] W[j][k]     as ONE 16 x 16 tile whose j range is cut over the four waves (F / 16 MFMAs each, the
//                                             weight elements loaded up front), the four partial tiles summed through LDS in wave order.
template <int FMAX>
__global__ __launch_bounds__(256) void soft_fusion_bwd_mfma_kernel(const float* __restrict__ dout, const float* __restrict__ a,
                                                                   const float* __restrict__ b, const float* __restrict__ gate,
                                                                   const float* __restrict__ w1, const float* __restrict__ w2,
                                                                   float* __restrict__ da, float* __restrict__ db_,
                                                                   float* __restrict__ dw1, float* __restrict__ dbias1,
                                                                   float* __restrict__ dw2, float* __restrict__ dbias2, int R,
                                                                   int Fa, int Fb, int lda, int ldb, int accumulate) {
  constexpr int LD = FMAX + 4;                                     // LDS row stride: the 16 rows of a fragment read on different banks
  __shared__ float sdp[16][LD], scat[16][LD];
  __shared__ float red[4][16][17];
  const int F = Fa + Fb;
  const int g0 = blockIdx.x * 16;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, li = lane & 15, lq = lane >> 4;
  constexpr int NTW = FMAX / 64;                                   // k tiles of the weight gradient per wave
  f32x4 accw[NTW];
#pragma unroll
  for (int i = 0; i < NTW; ++i) accw[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;
  const int ntiles = F >> 4;
  // this wave's slice of the j range for dcat: rows [jw0, jw1) of the stacked weight, in steps of 4; column g0 + li
  const int jper = ((F / 4 + 3) / 4) * 4;                          // per wave, a multiple of 4
  const int jw0 = min(wave * jper, F), jw1 = min(jw0 + jper, F);
  for (int m0 = 0; m0 < R; m0 += 16) {
    __syncthreads();
    for (int i = t; i < 16 * F; i += 256) {
      const int mm = i / F, c = i - mm * F, m = m0 + mm;
      float dp = 0.f, cv = 0.f;
      if (m < R) {
        cv = c < Fa ? a[(size_t)m * lda + c] : b[(size_t)m * ldb + (c - Fa)];
        const float sg = gate[(size_t)m * F + c];
        dp = dout[(size_t)m * F + c] * cv * sg * (1.f - sg);
      }
      sdp[mm][c] = dp;
      scat[mm][c] = cv;
    }
    __syncthreads();
    // ---- weight-gradient rows g0 .. g0 + 15: A[i = row j][kk = sample], B[kk = sample][n = column k]
    {
      float av[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) av[q] = sdp[4 * q + lq][g0 + li];
#pragma unroll
      for (int i = 0; i < NTW; ++i) {
        const int kt = wave + 4 * i;
        if (kt < ntiles) {
#pragma unroll
          for (int q = 0; q < 4; ++q)
            accw[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q], scat[4 * q + lq][16 * kt + li], accw[i], 0, 0, 0);
        }
      }
      if (t < 16) {
        float sb = 0.f;
#pragma unroll
        for (int mm = 0; mm < 16; ++mm) sb += sdp[mm][g0 + t];
        bsum += sb;
      }
    }
    // ---- d cat, columns g0 .. g0 + 15: A[i = sample][kk = row j], B[kk = row j][n = column]
    {
      f32x4 accc = {0.f, 0.f, 0.f, 0.f};
      constexpr int CH = 16;                                        // weight elements in flight per lane
      for (int j0 = jw0; j0 < jw1; j0 += 4 * CH) {
        float wv[CH];
#pragma unroll
        for (int u = 0; u < CH; ++u) {
          const int j = min(j0 + 4 * u + lq, F - 1);
          wv[u] = j < Fa ? w1[(size_t)j * F + g0 + li] : w2[(size_t)(j - Fa) * F + g0 + li];
        }
#pragma unroll
        for (int u = 0; u < CH; ++u) {
          const int jb = j0 + 4 * u;
          if (jb < jw1) accc = __builtin_amdgcn_mfma_f32_16x16x4f32(sdp[li][min(jb + lq, F - 1)], wv[u], accc, 0, 0, 0);
        }
      }
      // C: column (l & 15) = k, row 4 (l >> 4) + r = sample
#pragma unroll
      for (int r = 0; r < 4; ++r) red[wave][4 * lq + r][li] = accc[r];
      __syncthreads();
      {
        const int mm = t >> 4, kk = t & 15, m = m0 + mm, k = g0 + kk;
        if (m < R && k < F) {
          const float s = ((red[0][mm][kk] + red[1][mm][kk]) + red[2][mm][kk]) + red[3][mm][kk];
          const float v = s + dout[(size_t)m * F + k] * gate[(size_t)m * F + k];
          if (k < Fa) da[(size_t)m * Fa + k] = v;
          else db_[(size_t)m * Fb + (k - Fa)] = v;
        }
      }
    }
  }
  // ---- write the weight gradient: tile column (l & 15) = k, row 4 (l >> 4) + r = j
#pragma unroll
  for (int i = 0; i < NTW; ++i) {
    const int kt = wave + 4 * i;
    if (kt >= ntiles) continue;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int j = g0 + 4 * lq + r, k = 16 * kt + li;
      if (j < F) {
        float* p = (j < Fa ? dw1 + (size_t)j * F : dw2 + (size_t)(j - Fa) * F) + k;
        *p = accumulate ? *p + accw[i][r] : accw[i][r];
      }
    }
  }
  if (t < 16 && g0 + t < F) {
    const int j = g0 + t;
    float* p = j < Fa ? dbias1 + j : dbias2 + (j - Fa);
    *p = accumulate ? *p + bsum : bsum;
  }
}

// deeplio_nets.py:84-90: dropout(p) in front of the two heads fc_pos / fc_ori (Linear(K, 3) each) -- slice copy + dropout +
// two linear launches.  One workgroup per sample row: the row (read with its own row stride: the forward half of the
// odometry LSTM's [.., 2H] output, no slice copy) is masked into LDS, the six dot products are taken by its four waves.
// The mask comes from the same Philox stream position a separate dropout launch over the contiguous [R, K] tensor would use.
__global__ __launch_bounds__(256) void heads_fwd_kernel(const float* __restrict__ x, int ldx, uint8_t* __restrict__ mask,
                                                        const float* __restrict__ wp, const float* __restrict__ bp,
                                                        const float* __restrict__ wo, const float* __restrict__ bo,
                                                        float* __restrict__ pos, float* __restrict__ ori, int R, int K, float p,
                                                        uint64_t seed, uint64_t offset) {
  extern __shared__ __attribute__((aligned(16))) float ys[];       // [K]
  const int m = blockIdx.x;
  const float inv = 1.0f / (1.0f - p);
  const float* __restrict__ xr = x + (size_t)m * ldx;
  for (int g = threadIdx.x; g < (K >> 2); g += 256) {
    float4 v = *reinterpret_cast<const float4*>(xr + 4 * g);
    if (mask) {
      const uint64_t ctr = offset + (uint64_t)(((size_t)m * K) >> 2) + (uint64_t)g;
      uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u};
      philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
      float* vv = reinterpret_cast<float*>(&v);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float u = (float)(c[j] >> 8) * (1.0f / 16777216.0f);
        const uint8_t keep = u >= p ? 1 : 0;
        mask[(size_t)m * K + 4 * g + j] = keep;
        vv[j] = keep ? vv[j] * inv : 0.f;
      }
    }
    *reinterpret_cast<float4*>(ys + 4 * g) = v;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int c = wave; c < 6; c += 4) {
    const float* __restrict__ wr = c < 3 ? wp + (size_t)c * K : wo + (size_t)(c - 3) * K;
    float s = 0.f;
    for (int k = lane * 4; k < K; k += 256) {
      const float4 wv = *reinterpret_cast<const float4*>(wr + k);
      const float4 yv = *reinterpret_cast<const float4*>(ys + k);
      s += (wv.x * yv.x + wv.y * yv.y) + (wv.z * yv.z + wv.w * yv.w);
    }
    s = wave_sum(s);
    if (lane == 0) {
      if (c < 3) pos[m * 3 + c] = s + (bp ? bp[c] : 0.f);
      else ori[m * 3 + (c - 3)] = s + (bo ? bo[c - 3] : 0.f);
    }
  }
}

// backward: thread = column k; dW[c][k] = sum_m d[m][c] y[m][k], dx[m][k] = keep / (1 - p) sum_c d[m][c] W[c][k]; the columns
// [K, lddx) of dx (the discarded reverse half of the LSTM output) are written as zeros; biases by workgroup 0
__global__ __launch_bounds__(256) void heads_bwd_kernel(const float* __restrict__ dpos, const float* __restrict__ dori,
                                                        const float* __restrict__ x, int ldx, const uint8_t* __restrict__ mask,
                                                        const float* __restrict__ wp, const float* __restrict__ wo,
                                                        float* __restrict__ dx, int lddx, float* __restrict__ dwp,
                                                        float* __restrict__ dbp, float* __restrict__ dwo, float* __restrict__ dbo,
                                                        int R, int K, float p, int accumulate) {
  extern __shared__ float sd[];                                    // [R][6]
  const float inv = 1.0f / (1.0f - p);
  for (int i = threadIdx.x; i < R * 6; i += 256) {
    const int m = i / 6, c = i - m * 6;
    sd[i] = c < 3 ? dpos[m * 3 + c] : dori[m * 3 + (c - 3)];
  }
  __syncthreads();
  if (blockIdx.x == 0 && threadIdx.x < 6) {
    const int c = threadIdx.x;
    float s = 0.f;
    for (int m = 0; m < R; ++m) s += sd[m * 6 + c];
    float* q = c < 3 ? (dbp ? dbp + c : nullptr) : (dbo ? dbo + (c - 3) : nullptr);
    if (q) *q = accumulate ? *q + s : s;
  }
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= lddx) return;
  if (k >= K) {
    if (dx) for (int m = 0; m < R; ++m) dx[(size_t)m * lddx + k] = 0.f;
    return;
  }
  float w6[6], dw[6];
#pragma unroll
  for (int c = 0; c < 6; ++c) { w6[c] = c < 3 ? wp[(size_t)c * K + k] : wo[(size_t)(c - 3) * K + k]; dw[c] = 0.f; }
  for (int m = 0; m < R; ++m) {
    const float kp = mask ? (mask[(size_t)m * K + k] ? inv : 0.f) : 1.f;
    const float y = x[(size_t)m * ldx + k] * kp;
    float g = 0.f;
#pragma unroll
    for (int c = 0; c < 6; ++c) { const float dd = sd[m * 6 + c]; dw[c] += dd * y; g += dd * w6[c]; }
    if (dx) dx[(size_t)m * lddx + k] = g * kp;
  }
#pragma unroll
  for (int c = 0; c < 6; ++c) {
    float* q = c < 3 ? dwp + (size_t)c * K + k : dwo + (size_t)(c - 3) * K + k;
    *q = accumulate ? *q + dw[c] : dw[c];
  }
}

}  // namespace

extern "C" int dlio_linear_fwd(const float* x, int ldx, const float* w, const float* b,
                               const float* addend, int ldadd, float* y, int ldy, int M, int N,
                               int K, int act, dlio_stream_t stream) {
  if (!x || !w || !y || M <= 0 || N <= 0 || K <= 0 || ldx < K || ldy < N || act < 0 || act > 4)
    return DLIO_EINVAL;
  static const int mfma_from = 128;
  if (M >= mfma_from) {                 // tall M (IMU windows): GEMM on the fp32 MFMA
    const int kvec = (K % 8 == 0) && (ldx % 4 == 0) &&
                     ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w)) & 15) == 0;
    hipLaunchKernelGGL(linear_fwd_mfma_kernel, dim3(cdiv(N, 64), cdiv(M, 64)), dim3(256), 0, as_stream(stream), x, ldx,
                       w, b, addend, ldadd, y, ldy, M, N, K, act, kvec);
    return dlio_check_launch();
  }
  // skinny M against a large weight matrix (the odometry LSTM): x staged once per workgroup
  static const int skinny = 1;
  if (skinny && M <= 16 && (K & 3) == 0 && (ldx & 3) == 0 && (int64_t)N * K >= (1 << 18) &&
      ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w)) & 15) == 0) {
    const int mrows = M <= 8 ? 8 : 16;
    const size_t lds = (size_t)mrows * K * sizeof(float);
    if (lds <= 128 * 1024) {
      int grid = cdiv(cdiv(N, 4), 4);
      if (grid > 2 * dlio_num_cus()) grid = 2 * dlio_num_cus();
      auto kern = mrows == 8 ? linear_fwd_skinny_kernel<8> : linear_fwd_skinny_kernel<16>;
      dlio_set_max_lds(reinterpret_cast<const void*>(kern), 128 * 1024);
      hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, as_stream(stream), x, ldx, w, b, addend, ldadd, y, ldy, M, N, K, act);
      return dlio_check_launch();
    }
  }
  int grid = cdiv(N, 4);
  if (grid > 8192) grid = 8192;
  hipLaunchKernelGGL(linear_fwd_kernel, dim3(grid), dim3(256), 0, as_stream(stream), x, ldx, w, b,
                     addend, ldadd, y, ldy, M, N, K, act);
  return dlio_check_launch();
}

extern "C" int dlio_act_bwd(const float* dy, const float* y, float* dz, int64_t n, int act,
                            dlio_stream_t stream) {
  if (!dy || !y || !dz || n <= 0 || act < 0 || act > 4) return DLIO_EINVAL;
  hipLaunchKernelGGL(act_bwd_kernel, dim3(ew_grid(n, 256)), dim3(256), 0, as_stream(stream), dy, y,
                     dz, n, act);
  return dlio_check_launch();
}

extern "C" size_t dlio_linear_bwd_data_ws_bytes(int M, int N, int K) {
  (void)M;
  const int ns = bwd_data_nsplit(N, K);
  return ns ? (size_t)ns * BM * K * sizeof(float) : 0;
}

extern "C" int dlio_linear_bwd_data(const float* dz, int lddz, const float* w, float* dx, int lddx,
                                    int M, int N, int K, int accumulate, void* ws, size_t ws_bytes,
                                    dlio_stream_t stream) {
  if (!dz || !w || !dx || M <= 0 || N <= 0 || K <= 0 || lddz < N || lddx < K) return DLIO_EINVAL;
  hipStream_t s = as_stream(stream);
  const int ns = bwd_data_nsplit(N, K);
  const bool aligned = ((reinterpret_cast<uintptr_t>(w) & 15) == 0);
  if (ns && aligned && ws && ws_bytes >= (size_t)ns * BM * K * sizeof(float)) {
    float* part = reinterpret_cast<float*>(ws);
    for (int m0 = 0; m0 < M; m0 += BM) {
      const int rows = M - m0 < BM ? M - m0 : BM;
      if (rows <= 8)
        hipLaunchKernelGGL(linear_bwd_data_split_kernel<8>, dim3((K / 4 + 63) / 64, ns), dim3(256), 0, s, dz, lddz, w, part,
                           M, N, K, m0, ns);
      else
        hipLaunchKernelGGL(linear_bwd_data_split_kernel<BM>, dim3((K / 4 + 63) / 64, ns), dim3(256), 0, s, dz, lddz, w, part,
                           M, N, K, m0, ns);
      hipLaunchKernelGGL(linear_bwd_data_reduce_kernel, dim3(cdiv(rows * K, 256)), dim3(256), 0, s, part,
                         dx, lddx, rows, K, m0, ns, accumulate);
    }
    return dlio_check_launch();
  }
  hipLaunchKernelGGL(linear_bwd_data_kernel, dim3(cdiv(K, 64), cdiv(M, BM)), dim3(256), 0, s, dz, lddz,
                     w, dx, lddx, M, N, K, accumulate);
  return dlio_check_launch();
}

extern "C" int dlio_linear_bwd_weight(const float* dz, int lddz, const float* x, int ldx,
                                      float* dw, float* db, int M, int N, int K, int accumulate,
                                      dlio_stream_t stream) {
  if (!dz || !x || !dw || M <= 0 || N <= 0 || K <= 0 || lddz < N || ldx < K) return DLIO_EINVAL;
  hipStream_t s = as_stream(stream);
  static const int mfma_from = 128;
  if (M >= mfma_from) {                 // tall M: one launch for dW and db
    hipLaunchKernelGGL(linear_wgrad_mfma_kernel, dim3(cdiv(K, 32), cdiv(N, 32)), dim3(256), 0, s, dz, lddz, x, ldx, dw,
                       db, M, N, K, accumulate);
    return dlio_check_launch();
  }
  if ((K & 3) == 0 && (ldx & 3) == 0 && K >= 256 &&
      ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dw)) & 15) == 0)
  {
    hipLaunchKernelGGL(linear_bwd_weight_v4_kernel, dim3(cdiv(K / 4, 256), cdiv(N, 4)), dim3(256), 0, s, dz, lddz, x, ldx, dw,
                       db, M, N, K, accumulate);
    return dlio_check_launch();
  }
  else
    hipLaunchKernelGGL(linear_bwd_weight_kernel, dim3(cdiv(K, 256), cdiv(N, 4)), dim3(256), 0, s, dz,
                       lddz, x, ldx, dw, M, N, K, accumulate);
  int rc = dlio_check_launch();
  if (rc) return rc;
  if (db) {
    hipLaunchKernelGGL(col_sum_kernel, dim3(cdiv(N, 256)), dim3(256), 0, s, dz, lddz, db, M, N,
                       accumulate);
    rc = dlio_check_launch();
  }
  return rc;
}

extern "C" int dlio_ew_binary(const float* a, const float* b, float* y, int64_t n, int op, float* amax_out,
                              dlio_stream_t stream) {
  if (!a || !b || !y || n <= 0 || op < 0 || op > 3) return DLIO_EINVAL;
  const bool v4 = (n & 3) == 0 && ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(y)) & 15) == 0;
  if (amax_out && !v4) return DLIO_EUNSUP;
  if (v4 && (n >= 4096 || amax_out))
    hipLaunchKernelGGL(ew_binary_v4_kernel, dim3(ew_grid(n / 16, 256)), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const float4*>(a), reinterpret_cast<const float4*>(b), reinterpret_cast<float4*>(y), n / 4,
                       op, amax_out);
  else
    hipLaunchKernelGGL(ew_binary_kernel, dim3(ew_grid(n, 256)), dim3(256), 0, as_stream(stream), a,
                       b, y, n, op);
  return dlio_check_launch();
}

extern "C" int dlio_seg_sum_fwd(const float* x, float* y, int groups, int rows, int cols,
                                dlio_stream_t stream) {
  if (!x || !y || groups <= 0 || rows <= 0 || cols <= 0) return DLIO_EINVAL;
  hipLaunchKernelGGL(seg_sum_fwd_kernel, dim3(ew_grid((int64_t)groups * cols, 256)), dim3(256), 0,
                     as_stream(stream), x, y, groups, rows, cols);
  return dlio_check_launch();
}

extern "C" int dlio_seg_sum_bwd(const float* dy, float* dx, int groups, int rows, int cols,
                                dlio_stream_t stream) {
  if (!dy || !dx || groups <= 0 || rows <= 0 || cols <= 0) return DLIO_EINVAL;
  hipLaunchKernelGGL(seg_sum_bwd_kernel, dim3(ew_grid((int64_t)groups * rows * cols, 256)),
                     dim3(256), 0, as_stream(stream), dy, dx, groups, rows, cols);
  return dlio_check_launch();
}

// largest |x| of a tensor -> *amax_out (zero before the launch): the operand scale of a two-piece fp16 consumer whose input
// has no producer kernel to leave it (the stem's range images)
__global__ __launch_bounds__(256) void abs_max_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ amax_out) {
  __shared__ unsigned s_amax;
  float amax = 0.f;
  const int64_t n4 = n >> 2, stride = (int64_t)gridDim.x * blockDim.x;
  const bool al = (reinterpret_cast<uintptr_t>(x) & 15) == 0;
  if (al) {
    const float4* x4 = reinterpret_cast<const float4*>(x);
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += 4 * stride) {
      float4 v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = x4[i + k * stride < n4 ? i + k * stride : n4 - 1];
#pragma unroll
      for (int k = 0; k < 4; ++k) amax = amax4(amax, v[k].x, v[k].y, v[k].z, v[k].w);
    }
  }
  for (int64_t i = (al ? (n4 << 2) : 0) + blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += stride)
    amax = fmaxf(amax, fabsf(x[i]));
  block_amax_commit(amax, amax_out, &s_amax);
}

extern "C" int dlio_abs_max(const float* x, int64_t n, float* amax_out, dlio_stream_t stream) {
  if (!x || !amax_out || n <= 0) return DLIO_EINVAL;
  hipLaunchKernelGGL(abs_max_kernel, dim3(ew_grid(n / 16 + 1, 256)), dim3(256), 0, as_stream(stream), x, n, amax_out);
  return dlio_check_launch();
}

extern "C" int dlio_ew_scale(const float* a, float alpha, float* y, int64_t n,
                             dlio_stream_t stream) {
  if (!a || !y || n <= 0) return DLIO_EINVAL;
  hipLaunchKernelGGL(ew_scale_kernel, dim3(ew_grid(n, 256)), dim3(256), 0, as_stream(stream), a,
                     alpha, y, n);
  return dlio_check_launch();
}

extern "C" int dlio_copy2d(const float* src, int lds, float* dst, int ldd, int rows, int cols,
                           int accumulate, dlio_stream_t stream) {
  if (!src || !dst || rows <= 0 || cols <= 0 || lds < cols || ldd < cols) return DLIO_EINVAL;
  hipLaunchKernelGGL(copy2d_kernel, dim3(ew_grid((int64_t)rows * cols, 256)), dim3(256), 0,
                     as_stream(stream), src, lds, dst, ldd, rows, cols, accumulate);
  return dlio_check_launch();
}

extern "C" int dlio_dropout_fwd(const float* x, float* y, uint8_t* mask, int64_t n, float p,
                                uint64_t seed, uint64_t offset, dlio_stream_t stream) {
  if (!x || !y || !mask || n <= 0 || !(p >= 0.f && p < 1.f)) return DLIO_EINVAL;
  hipLaunchKernelGGL(dropout_fwd_kernel, dim3(ew_grid((n + 3) / 4, 256)), dim3(256), 0,
                     as_stream(stream), x, y, mask, n, p, seed, offset);
  return dlio_check_launch();
}

extern "C" int dlio_dropout_bwd(const float* dy, const uint8_t* mask, float* dx, int64_t n,
                                float p, dlio_stream_t stream) {
  if (!dy || !mask || !dx || n <= 0 || !(p >= 0.f && p < 1.f)) return DLIO_EINVAL;
  hipLaunchKernelGGL(dropout_bwd_kernel, dim3(ew_grid(n, 256)), dim3(256), 0, as_stream(stream),
                     dy, mask, dx, n, p);
  return dlio_check_launch();
}

extern "C" int dlio_nonfinite_flag(const float* x, int64_t n, int32_t* flag,
                                   dlio_stream_t stream) {
  if (!x || !flag || n <= 0) return DLIO_EINVAL;
  hipLaunchKernelGGL(nonfinite_kernel, dim3(ew_grid(n, 256)), dim3(256), 0, as_stream(stream), x,
                     n, flag);
  return dlio_check_launch();
}

extern "C" int dlio_soft_fusion_ok(int R, int Fa, int Fb) { return R > 0 && Fa > 0 && Fb > 0 && Fa + Fb <= 512; }

extern "C" int dlio_soft_fusion_fwd(const float* a, int lda, const float* b, int ldb, const float* w1, const float* b1,
                                    const float* w2, const float* b2, float* out, float* gate, int R, int Fa, int Fb,
                                    dlio_stream_t stream) {
  if (!a || !b || !w1 || !w2 || !out || !gate || lda < Fa || ldb < Fb) return DLIO_EINVAL;
  if (!dlio_soft_fusion_ok(R, Fa, Fb)) return DLIO_EUNSUP;
  hipLaunchKernelGGL(soft_fusion_fwd_kernel, dim3(cdiv(Fa + Fb, 4)), dim3(256), 0, as_stream(stream), a, b, w1, b1, w2, b2, out,
                     gate, R, Fa, Fb, lda, ldb);
  return dlio_check_launch();
}

extern "C" int dlio_soft_fusion_bwd(const float* dout, const float* a, int lda, const float* b, int ldb, const float* gate,
                                    const float* w1, const float* w2, float* da, float* db, float* dw1, float* dbias1, float* dw2,
                                    float* dbias2, int R, int Fa, int Fb, int accumulate, dlio_stream_t stream) {
  if (!dout || !a || !b || !gate || !w1 || !w2 || !da || !db || !dw1 || !dbias1 || !dw2 || !dbias2 || lda < Fa || ldb < Fb)
    return DLIO_EINVAL;
  if (!dlio_soft_fusion_ok(R, Fa, Fb)) return DLIO_EUNSUP;
  const int F = Fa + Fb;
  static const int mfma_on = getenv("DLIO_SOFT_FUSION_MFMA") ? atoi(getenv("DLIO_SOFT_FUSION_MFMA")) : 1;
  if (mfma_on && (F & 15) == 0 && (Fa & 15) == 0) {
    if (F <= 256)
      hipLaunchKernelGGL((soft_fusion_bwd_mfma_kernel<256>), dim3(F / 16), dim3(256), 0, as_stream(stream), dout, a, b, gate, w1,
                         w2, da, db, dw1, dbias1, dw2, dbias2, R, Fa, Fb, lda, ldb, accumulate);
    else
      hipLaunchKernelGGL((soft_fusion_bwd_mfma_kernel<512>), dim3(F / 16), dim3(256), 0, as_stream(stream), dout, a, b, gate, w1,
                         w2, da, db, dw1, dbias1, dw2, dbias2, R, Fa, Fb, lda, ldb, accumulate);
    return dlio_check_launch();
  }
  if (F <= 256)
    hipLaunchKernelGGL((soft_fusion_bwd_kernel<256>), dim3(cdiv(F, 16)), dim3(256), 0, as_stream(stream), dout, a, b, gate, w1, w2,
                       da, db, dw1, dbias1, dw2, dbias2, R, Fa, Fb, lda, ldb, accumulate);
  else
    hipLaunchKernelGGL((soft_fusion_bwd_kernel<512>), dim3(cdiv(F, 16)), dim3(256), 0, as_stream(stream), dout, a, b, gate, w1, w2,
                       da, db, dw1, dbias1, dw2, dbias2, R, Fa, Fb, lda, ldb, accumulate);
  return dlio_check_launch();
}

extern "C" int dlio_heads_ok(int R, int K, int ldx) {
  return R > 0 && R <= 1024 && K >= 4 && (K & 3) == 0 && K <= 8192 && ldx >= K && (ldx & 3) == 0;
}

extern "C" int dlio_heads_fwd(const float* x, int ldx, uint8_t* mask, const float* wp, const float* bp, const float* wo,
                              const float* bo, float* pos, float* ori, int R, int K, float p, uint64_t seed, uint64_t offset,
                              dlio_stream_t stream) {
  if (!x || !wp || !wo || !pos || !ori || !(p >= 0.f && p < 1.f)) return DLIO_EINVAL;
  if (!dlio_heads_ok(R, K, ldx) || (reinterpret_cast<uintptr_t>(x) & 15)) return DLIO_EUNSUP;
  hipLaunchKernelGGL(heads_fwd_kernel, dim3(R), dim3(256), (size_t)K * sizeof(float), as_stream(stream), x, ldx, mask, wp, bp, wo,
                     bo, pos, ori, R, K, p, seed, offset);
  return dlio_check_launch();
}

extern "C" int dlio_heads_bwd(const float* dpos, const float* dori, const float* x, int ldx, const uint8_t* mask,
                              const float* wp, const float* wo, float* dx, int lddx, float* dwp, float* dbp, float* dwo,
                              float* dbo, int R, int K, float p, int accumulate, dlio_stream_t stream) {
  if (!dpos || !dori || !x || !wp || !wo || !dwp || !dwo || !(p >= 0.f && p < 1.f)) return DLIO_EINVAL;
  if (!dlio_heads_ok(R, K, ldx) || (dx && lddx < K)) return DLIO_EUNSUP;
  hipLaunchKernelGGL(heads_bwd_kernel, dim3(cdiv(dx ? lddx : K, 256)), dim3(256), (size_t)R * 6 * sizeof(float), as_stream(stream),
                     dpos, dori, x, ldx, mask, wp, wo, dx, dx ? lddx : K, dwp, dbp, dwo, dbo, R, K, p, accumulate);
  return dlio_check_launch();
}
