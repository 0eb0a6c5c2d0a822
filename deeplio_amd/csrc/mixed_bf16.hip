// HBM-bound streaming kernels of the mixed-precision PointSeg path (BASELINE configs[4]) over bf16
// NCHW tensors: train-mode BatchNorm (statistics, apply, backward reductions, backward apply), the
// 3x3 max-pool with the fused SELayer scale (forward, backward, scale gradient), global average pool
// and the fp32 <-> bf16 casts at the path's two ends.  Arithmetic is fp32 (fp64 for the statistics),
// exactly as in bn.hip / pool.hip; only the storage type differs: eight elements per 16-byte access,
// half the bytes per element -- which is the whole point on these kernels.
// Same reference sites as bn.hip / pool.hip (pointseg_modules.py:98-106,216-221, pointseg_net.py:27-46).
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void ld8(const __bf16* p, float (&v)[8]) {
  const bf16x8 t = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = (float)t[j];
}
__device__ __forceinline__ void st8(__bf16* p, const float (&v)[8]) {
  bf16x8 t;
#pragma unroll
  for (int j = 0; j < 8; ++j) t[j] = (__bf16)v[j];
  *reinterpret_cast<bf16x8*>(p) = t;
}

constexpr int RB = 256;

// partial sums of one channel over a slice of its (n, hw) domain -> part[c][split][2] (fp64)
// MODE 0: sum x, sum x^2      MODE 1: sum g, sum g*xhat with g = dy masked by the ReLU behind the BN
template <int MODE>
__global__ __launch_bounds__(RB) void bn16_reduce_kernel(
    const __bf16* __restrict__ a, int a_ctot, int a_coff, const __bf16* __restrict__ x, int x_ctot, int x_coff,
    const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ scale,
    const float* __restrict__ beta, int N, int C, int HW, int post_relu, int splits, double* __restrict__ part) {
  __shared__ double sm[2][16];
  const int c = blockIdx.x / splits, sp = blockIdx.x % splits;
  float mu = 0.f, is = 0.f, sc = 0.f, be = 0.f;
  if (MODE == 1) { mu = mean[c]; is = invstd[c]; sc = scale[c]; be = beta ? beta[c] : 0.f; }
  const int hw8 = HW >> 3;
  const int64_t total8 = (int64_t)N * hw8;
  const int64_t stride = (int64_t)splits * RB;
  const int q = (int)(stride / hw8), r = (int)(stride - (int64_t)q * hw8);
  int64_t i = (int64_t)sp * RB + threadIdx.x;
  int n = (int)(i / hw8), p = (int)(i - (int64_t)n * hw8);
  const __bf16* ab = a + ((size_t)a_coff + c) * HW;
  const __bf16* xb = MODE == 1 ? x + ((size_t)x_coff + c) * HW : nullptr;
  const size_t an = (size_t)a_ctot * HW, xn = (size_t)x_ctot * HW;
  double s0 = 0.0, s1 = 0.0;
  constexpr int UN = 2;
  auto acc = [&](const float (&av)[8], const float (&xv)[8]) {
    float f0 = 0.f, f1 = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (MODE == 1) {
        float g = av[k];
        if (post_relu && !((xv[k] - mu) * sc + be > 0.f)) g = 0.f;
        f0 += g; f1 += g * ((xv[k] - mu) * is);
      } else {
        f0 += av[k]; f1 += av[k] * av[k];
      }
    }
    s0 += f0; s1 += f1;
  };
  for (; i + (UN - 1) * stride < total8; i += UN * stride) {
    float av[UN][8], xv[UN][8];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      ld8(ab + (size_t)n * an + ((size_t)p << 3), av[u]);
      if (MODE == 1) ld8(xb + (size_t)n * xn + ((size_t)p << 3), xv[u]);
      n += q; p += r;
      if (p >= hw8) { p -= hw8; ++n; }
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) acc(av[u], MODE == 1 ? xv[u] : av[u]);
  }
  for (; i < total8; i += stride) {
    float av[8], xv[8];
    ld8(ab + (size_t)n * an + ((size_t)p << 3), av);
    if (MODE == 1) ld8(xb + (size_t)n * xn + ((size_t)p << 3), xv);
    acc(av, MODE == 1 ? xv : av);
    n += q; p += r;
    if (p >= hw8) { p -= hw8; ++n; }
  }
  const double r0 = block_sum_d(s0, sm[0]);
  const double r1 = block_sum_d(s1, sm[1]);
  if (threadIdx.x == 0) {
    part[((size_t)c * splits + sp) * 2 + 0] = r0;
    part[((size_t)c * splits + sp) * 2 + 1] = r1;
  }
}

__device__ __forceinline__ void plane_partials16(const double* __restrict__ part, int c, int splits, double* sm0,
                                                 double* sm1, double& a, double& b) {
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

// one workgroup per (n, c) plane (or plane chunk): finalises its channel's statistics from the partials
// itself (fixed order -> bit-identical in every workgroup of the channel), then streams its plane:
// y = relu?((x - mean) * scale + beta) + residual, rounded once to bf16; by-product: the plane average
// of the STORED (rounded) outputs for the SELayer behind a Fire block.
__global__ __launch_bounds__(256) void bn16_plane_apply_kernel(
    const __bf16* __restrict__ x, int x_ctot, int x_coff, const double* __restrict__ part, int splits, double count,
    const float* __restrict__ gamma, const float* __restrict__ beta, float eps, float momentum, float* running_mean,
    float* running_var, float* mean_o, float* invstd_o, float* scale_o, const __bf16* residual, int r_ctot, int r_coff,
    __bf16* y, int y_ctot, int y_coff, int N, int C, int HW, int post_relu, int chunks, int chunk_len,
    float* __restrict__ gap_out, int gap_ctot, int gap_coff, int eval_mode) {
  __shared__ double sm[2][16];
  const int chunk = blockIdx.x % chunks;
  const int pl = blockIdx.x / chunks;
  const int n = pl / C, c = pl - n * C;
  float mu, is, sc;
  const float be = beta ? beta[c] : 0.f;
  if (eval_mode) {                                  // running statistics (mean_o / invstd_o / scale_o precomputed)
    mu = mean_o[c]; is = invstd_o[c]; sc = scale_o[c];
  } else {
    double a, b;
    plane_partials16(part, c, splits, sm[0], sm[1], a, b);
    const double m = a / count;
    double var = b / count - m * m;
    if (var < 0.0) var = 0.0;
    is = (float)(1.0 / sqrt(var + (double)eps));
    mu = (float)m; sc = (gamma ? gamma[c] : 1.f) * is;
    if (n == 0 && chunk == 0 && threadIdx.x == 0) {
      mean_o[c] = mu; invstd_o[c] = is; scale_o[c] = sc;
      if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mu;
      if (running_var) {
        const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
      }
    }
  }
  const __bf16* xp = x + ((size_t)n * x_ctot + x_coff + c) * HW;
  __bf16* yp = y + ((size_t)n * y_ctot + y_coff + c) * HW;
  const __bf16* rp = residual ? residual + ((size_t)n * r_ctot + r_coff + c) * HW : nullptr;
  const int per = HW >> 3;
  const int i1 = min(per, (chunk + 1) * chunk_len);
  double gs = 0.0;
  for (int i = chunk * chunk_len + threadIdx.x; i < i1; i += 256) {
    float v[8], rv[8];
    ld8(xp + ((size_t)i << 3), v);
    if (rp) ld8(rp + ((size_t)i << 3), rv);
    bf16x8 o;
    float fs = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float t = (v[k] - mu) * sc + be;
      if (post_relu) t = fmaxf(t, 0.f);
      if (rp) t += rv[k];
      o[k] = (__bf16)t;
      fs += (float)o[k];
    }
    *reinterpret_cast<bf16x8*>(yp + ((size_t)i << 3)) = o;
    gs += fs;
  }
  if (gap_out) {                       // chunks == 1 whenever gap_out is set
    const double r = block_sum_d(gs, sm[0]);
    if (threadIdx.x == 0) gap_out[(size_t)n * gap_ctot + gap_coff + c] = (float)(r / (double)HW);
  }
}

__global__ __launch_bounds__(256) void bn16_plane_bwd_kernel(
    const __bf16* __restrict__ dy, int dy_ctot, int dy_coff, const __bf16* __restrict__ x, int x_ctot, int x_coff,
    const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ scale,
    const float* __restrict__ beta, const double* __restrict__ part, const double* __restrict__ lpart, double inv_cnt,
    int splits, __bf16* dx, int dx_ctot, int dx_coff, float* dgamma, float* dbeta, int accumulate, int N, int C, int HW,
    int post_relu, int use_batch_stats, int chunks, int chunk_len) {
  __shared__ double sm[2][16];
  const int chunk = blockIdx.x % chunks;
  const int pl = blockIdx.x / chunks;
  const int n = pl / C, c = pl - n * C;
  double sg, sgx;
  if (n == 0 && chunk == 0) {
    // synchronised statistics: `part` holds the sums over all replicas, `lpart` this replica's -- dgamma / dbeta are
    // local sums (the gradient all-reduce adds the replicas')
    double lg, lgx;
    plane_partials16(lpart ? lpart : part, c, splits, sm[0], sm[1], lg, lgx);
    if (threadIdx.x == 0) {
      if (dbeta) dbeta[c] = accumulate ? dbeta[c] + (float)lg : (float)lg;
      if (dgamma) dgamma[c] = accumulate ? dgamma[c] + (float)lgx : (float)lgx;
    }
    __syncthreads();
  }
  plane_partials16(part, c, splits, sm[0], sm[1], sg, sgx);
  const float mu = mean[c], is = invstd[c], sc = scale[c], be = beta ? beta[c] : 0.f;
  float mg = 0.f, mgx = 0.f;
  if (use_batch_stats) { mg = (float)(sg * inv_cnt); mgx = (float)(sgx * inv_cnt); }
  const __bf16* gp = dy + ((size_t)n * dy_ctot + dy_coff + c) * HW;
  const __bf16* xp = x + ((size_t)n * x_ctot + x_coff + c) * HW;
  __bf16* op = dx + ((size_t)n * dx_ctot + dx_coff + c) * HW;
  const int per = HW >> 3;
  const int i1 = min(per, (chunk + 1) * chunk_len);
  for (int i = chunk * chunk_len + threadIdx.x; i < i1; i += 256) {
    float g[8], xv[8];
    ld8(gp + ((size_t)i << 3), g);
    ld8(xp + ((size_t)i << 3), xv);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float gg = g[k];
      if (post_relu && !((xv[k] - mu) * sc + be > 0.f)) gg = 0.f;
      g[k] = sc * (gg - mg - (xv[k] - mu) * is * mgx);
    }
    st8(op + ((size_t)i << 3), g);
  }
}

int splits16(int N, int C, int HW) {
  const int64_t per_chan8 = (int64_t)N * HW / 8;
  int64_t want = cdiv64(2048, C);
  const int64_t max_sp = cdiv64(per_chan8, (int64_t)RB * 4);
  if (want > max_sp) want = max_sp;
  if (want < 1) want = 1;
  if (want > 512) want = 512;
  return (int)want;
}

void plane_chunks16(int planes, int per, bool whole_plane, int& chunks, int& chunk_len) {
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

// ---- 3x3 max-pool, padding 1, stride (SH, 2), optional per-plane scale (the fused SELayer) ------------
// thread = 8 consecutive output pixels of one row: per input row two aligned 16-byte loads + the left
// neighbour; first maximum in row-major window order wins (ATen's tie rule), idx = kh*3 + kw
template <int SH>
__global__ __launch_bounds__(256) void pool16_fwd_kernel(const __bf16* __restrict__ x, const float* __restrict__ xs,
                                                         __bf16* __restrict__ y, uint8_t* __restrict__ idx,
                                                         int64_t planes, int H, int W, int OH, int OW) {
  const int ow8 = OW >> 3;
  const int64_t total = planes * OH * ow8;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c8 = (int)(i % ow8);
    int64_t t = i / ow8;
    const int oh = (int)(t % OH);
    const int64_t pl = t / OH;
    const float s = xs ? xs[pl] : 1.f;
    const __bf16* xp = x + pl * (int64_t)H * W;
    const int iw0 = c8 * 16;                       // first input column of the 16 aligned ones
    float best[8];
    int bi[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { best[k] = -INFINITY; bi[k] = 0; }
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int ih = oh * SH - 1 + kh;
      if (ih < 0 || ih >= H) continue;
      const __bf16* row = xp + (int64_t)ih * W + iw0;
      float v[17];
      v[0] = iw0 > 0 ? (float)row[-1] : -INFINITY;
      float a[8], b[8];
      ld8(row, a);
      ld8(row + 8, b);
#pragma unroll
      for (int k = 0; k < 8; ++k) { v[1 + k] = a[k]; v[9 + k] = b[k]; }
#pragma unroll
      for (int k = 0; k < 8; ++k)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const float sv = v[2 * k + kw] * s;           // left padding: -inf (s > 0: a sigmoid)
          if (sv > best[k]) { best[k] = sv; bi[k] = kh * 3 + kw; }
        }
    }
    const int64_t o = (pl * OH + oh) * (int64_t)OW + c8 * 8;
    st8(y + o, best);
    if (idx) {
      uint2 pk;
      pk.x = (unsigned)bi[0] | ((unsigned)bi[1] << 8) | ((unsigned)bi[2] << 16) | ((unsigned)bi[3] << 24);
      pk.y = (unsigned)bi[4] | ((unsigned)bi[5] << 8) | ((unsigned)bi[6] << 16) | ((unsigned)bi[7] << 24);
      *reinterpret_cast<uint2*>(idx + o) = pk;
    }
  }
}

// dx[h][w] = scale * sum over the (<= 2 x 3 / 2 x 2) windows whose arg-max is (h, w) of dy + add[plane]
template <int SH>
__global__ __launch_bounds__(256) void pool16_bwd_kernel(const __bf16* __restrict__ dy, const uint8_t* __restrict__ idx,
                                                         const float* __restrict__ xs, const float* __restrict__ xadd,
                                                         __bf16* __restrict__ dx, int64_t planes, int H, int W, int OH,
                                                         int OW) {
  const int w8 = W >> 3;
  const int64_t total = planes * H * w8;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c8 = (int)(i % w8);
    int64_t t = i / w8;
    const int h = (int)(t % H);
    const int64_t pl = t / H;
    const float s = xs ? xs[pl] : 1.f, ad = xadd ? xadd[pl] : 0.f;
    const int w0 = c8 * 8, ow0 = c8 * 4;          // outputs ow0 .. ow0+4 can point into these 8 columns
    float acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = 0.f;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int num = h + 1 - kh;                 // = oh * SH
      if (num < 0 || num % SH) continue;
      const int oh = num / SH;
      if (oh >= OH) continue;
      const int64_t ro = (pl * OH + oh) * (int64_t)OW + ow0;
      float g[5];
      int id[5];
      // outputs ow0 .. ow0+3 as one 8-byte / one 4-byte load (OW % 8 == 0, ow0 % 4 == 0: aligned, all inside the row),
      // the fifth (first of the next thread's group) on its own -- 12 loads per thread instead of 30 scalar ones
      const uint2 gv = *reinterpret_cast<const uint2*>(dy + ro);
      const unsigned iv = *reinterpret_cast<const unsigned*>(idx + ro);
      g[0] = __builtin_bit_cast(float, gv.x << 16); g[1] = __builtin_bit_cast(float, gv.x & 0xffff0000u);
      g[2] = __builtin_bit_cast(float, gv.y << 16); g[3] = __builtin_bit_cast(float, gv.y & 0xffff0000u);
      id[0] = (int)(iv & 255u); id[1] = (int)((iv >> 8) & 255u); id[2] = (int)((iv >> 16) & 255u); id[3] = (int)(iv >> 24);
      const bool v4 = ow0 + 4 < OW;
      g[4] = v4 ? (float)dy[ro + 4] : 0.f;
      id[4] = v4 ? (int)idx[ro + 4] : -1;
      // column w0 + k: even k -> output ow0 + k/2 with kw = 1; odd k -> outputs ow0 + (k-1)/2 (kw = 2), ow0 + (k+1)/2 (kw = 0)
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if ((k & 1) == 0) {
          if (id[k >> 1] == kh * 3 + 1) acc[k] += g[k >> 1];
        } else {
          if (id[k >> 1] == kh * 3 + 2) acc[k] += g[k >> 1];
          if (id[(k + 1) >> 1] == kh * 3 + 0) acc[k] += g[(k + 1) >> 1];
        }
      }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = acc[k] * s + ad;
    st8(dx + (pl * H + h) * (int64_t)W + w0, acc);
  }
}

// ds[plane] = sum over outputs of dy * x[arg-max] (gradient of the SELayer scale through the fused pool)
template <int SH>
__global__ __launch_bounds__(256) void pool16_bwd_dot_kernel(const __bf16* __restrict__ dy, const uint8_t* __restrict__ idx,
                                                             const __bf16* __restrict__ x, float* __restrict__ ds,
                                                             int planes, int H, int W, int OH, int OW) {
  __shared__ double sm[16];
  for (int pl = blockIdx.x; pl < planes; pl += gridDim.x) {
    const __bf16* xp = x + (int64_t)pl * H * W;
    const int64_t ob = (int64_t)pl * OH * OW;
    double acc = 0.0;
    for (int o = threadIdx.x; o < OH * OW; o += 256) {
      const int oh = o / OW, ow = o - oh * OW;
      const int id = idx[ob + o];
      const int ih = oh * SH - 1 + id / 3, iw = ow * 2 - 1 + id % 3;
      acc += (double)((float)dy[ob + o] * (float)xp[(int64_t)ih * W + iw]);
    }
    const double r = block_sum_d(acc, sm);
    if (threadIdx.x == 0) ds[pl] = (float)r;
    __syncthreads();
  }
}

// ---- the same three pools as rolling windows (pool.hip's strip kernels on bf16 storage): the kernels above read every
// input row once per output row that touches it (3x for row stride 1) and the scale gradient one element per thread and trip;
// here a thread owns a strip of rows and reads each row it needs ONCE.
// forward: 4 output columns (8 input columns = one 16-byte load + the left neighbour) x FR output rows; a row is reduced to
// (row maximum, kw) per output column, an output is the first maximum over its three row results -- row-then-column "first
// strictly greater wins" is the same total order as the flat kh-major scan of pool16_fwd_kernel: bit-identical values / codes
template <int SH>
__global__ __launch_bounds__(256) void pool16_fwd_strip(const __bf16* __restrict__ x, const float* __restrict__ xs,
                                                        __bf16* __restrict__ y, uint8_t* __restrict__ idx, int64_t planes,
                                                        int H, int W, int OH, int OW) {
  constexpr int FR = SH == 1 ? 8 : 4, NIN = (FR - 1) * SH + 3;
  const int ow4 = OW >> 2, strips = (OH + FR - 1) / FR;
  const int64_t total = planes * strips * ow4;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int b = (int)(i % ow4);
    int64_t t = i / ow4;
    const int oh0 = (int)(t % strips) * FR;
    const int64_t pl = t / strips;
    const float s = xs ? xs[pl] : 1.f;
    const __bf16* xp = x + pl * (int64_t)H * W;
    bf16x8 rows[NIN];
    __bf16 hal[NIN];
    bool rv[NIN];
#pragma unroll
    for (int j = 0; j < NIN; ++j) {                       // loads first, unconditional (clamped addresses)
      const int ih = oh0 * SH - 1 + j;
      rv[j] = ih >= 0 && ih < H;
      const __bf16* row = xp + (int64_t)min(max(ih, 0), H - 1) * W + 8 * b;
      rows[j] = *reinterpret_cast<const bf16x8*>(row);
      hal[j] = row[b > 0 ? -1 : 0];
    }
    float rb[NIN][4];
    int rk[NIN][4];
#pragma unroll
    for (int j = 0; j < NIN; ++j) {
      float v[9];
      v[0] = b > 0 ? (float)hal[j] * s : -INFINITY;       // left padding never wins (s > 0: a sigmoid)
#pragma unroll
      for (int k = 0; k < 8; ++k) v[1 + k] = (float)rows[j][k] * s;
#pragma unroll
      for (int o = 0; o < 4; ++o) {
        rb[j][o] = -INFINITY; rk[j][o] = 0;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
          if (v[2 * o + kw] > rb[j][o]) { rb[j][o] = v[2 * o + kw]; rk[j][o] = kw; }
      }
    }
#pragma unroll
    for (int r = 0; r < FR; ++r) {
      const int oh = oh0 + r;
      if (oh >= OH) continue;
      float best[4];
      unsigned code = 0;
#pragma unroll
      for (int o = 0; o < 4; ++o) {
        best[o] = -INFINITY;
        int bi = 0;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
          const int j = r * SH + kh;
          if (rv[j] && rb[j][o] > best[o]) { best[o] = rb[j][o]; bi = kh * 3 + rk[j][o]; }
        }
        code |= (unsigned)bi << (8 * o);
      }
      const int64_t oo = (pl * OH + oh) * (int64_t)OW + 4 * b;
      typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
      *reinterpret_cast<bf16x4*>(y + oo) = bf16x4{(__bf16)best[0], (__bf16)best[1], (__bf16)best[2], (__bf16)best[3]};
      if (idx) *reinterpret_cast<unsigned*>(idx + oo) = code;
    }
  }
}

// routed gradient of a strip: G[r][c] = sum of the dy whose arg-max is input element (ih0 + r, 8 b + c); every output row
// the strip's rows can be the arg-max of is read once (4 dy + 4 codes + the first output of the next thread's group)
constexpr int PR16 = 8;
template <int SH>
__device__ __forceinline__ void pool16_strip(const __bf16* __restrict__ dyp, const uint8_t* __restrict__ ip, int ih0, int b,
                                             int OH, int OW, float (&G)[PR16][8]) {
#pragma unroll
  for (int r = 0; r < PR16; ++r)
#pragma unroll
    for (int c = 0; c < 8; ++c) G[r][c] = 0.f;
  constexpr int J0 = SH == 1 ? -1 : 0, J1 = SH == 1 ? PR16 : PR16 / 2, NJ = J1 - J0 + 1;
  const bool has4 = 4 * b + 4 < OW;
  uint2 gv[NJ];
  unsigned iv[NJ];
  __bf16 g4[NJ];
  uint8_t i4[NJ];
  const int c4 = has4 ? 4 : 3;
#pragma unroll
  for (int j = J0; j <= J1; ++j) {
    const int oh = min(max(ih0 / SH + j, 0), OH - 1);
    const int64_t ro = (int64_t)oh * OW + 4 * b;
    gv[j - J0] = *reinterpret_cast<const uint2*>(dyp + ro);
    iv[j - J0] = *reinterpret_cast<const unsigned*>(ip + ro);
    g4[j - J0] = dyp[ro + c4];
    i4[j - J0] = ip[ro + c4];
  }
#pragma unroll
  for (int j = J0; j <= J1; ++j) {
    const int oh = ih0 / SH + j;
    if (oh < 0 || oh >= OH) continue;
    const uint2 g2 = gv[j - J0];
    const unsigned id = iv[j - J0];
    const float v[5] = {__builtin_bit_cast(float, g2.x << 16), __builtin_bit_cast(float, g2.x & 0xffff0000u),
                        __builtin_bit_cast(float, g2.y << 16), __builtin_bit_cast(float, g2.y & 0xffff0000u),
                        has4 ? (float)g4[j - J0] : 0.f};
    const int k[5] = {(int)(id & 255u), (int)((id >> 8) & 255u), (int)((id >> 16) & 255u), (int)(id >> 24),
                      has4 ? (int)i4[j - J0] : 4};              // 4 = (kh 1, kw 1): never column 7 of this group
#pragma unroll
    for (int c = 0; c < 5; ++c) {
      const int kh = k[c] / 3, kw = k[c] - 3 * kh;
#pragma unroll
      for (int khv = 0; khv < 3; ++khv) {
        const int r = j * SH - 1 + khv;                         // input row inside the strip (compile time)
        if (r < 0 || r >= PR16) continue;
        const float val = kh == khv ? v[c] : 0.f;
        // output c covers input columns 2c - 1 + kw (relative to 8 b): c = 0 -> (-1), 0, 1; c = 4 -> 7 only
        if (c == 0) { G[r][0] += kw == 1 ? val : 0.f; G[r][1] += kw == 2 ? val : 0.f; }
        else if (c < 4) { G[r][2 * c - 1] += kw == 0 ? val : 0.f; G[r][2 * c] += kw == 1 ? val : 0.f; G[r][2 * c + 1] += kw == 2 ? val : 0.f; }
        else { G[r][7] += kw == 0 ? val : 0.f; }
      }
    }
  }
}

template <int SH>
__global__ __launch_bounds__(256) void pool16_bwd_strip(const __bf16* __restrict__ dy, const uint8_t* __restrict__ idx,
                                                        const float* __restrict__ xs, const float* __restrict__ xadd,
                                                        __bf16* __restrict__ dx, int64_t planes, int H, int W, int OH, int OW) {
  const int w8 = W >> 3, strips = (H + PR16 - 1) / PR16;
  const int64_t total = planes * strips * w8;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int b = (int)(i % w8);
    int64_t t = i / w8;
    const int ih0 = (int)(t % strips) * PR16;
    const int64_t pl = t / strips;
    float G[PR16][8];
    pool16_strip<SH>(dy + pl * (int64_t)OH * OW, idx + pl * (int64_t)OH * OW, ih0, b, OH, OW, G);
    const float s = xs ? xs[pl] : 1.f, ad = xadd ? xadd[pl] : 0.f;
#pragma unroll
    for (int r = 0; r < PR16; ++r) {
      if (ih0 + r >= H) continue;
      float o[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) o[c] = G[r][c] * s + ad;
      st8(dx + (pl * H + ih0 + r) * (int64_t)W + 8 * b, o);
    }
  }
}

template <int SH>
__global__ __launch_bounds__(256) void pool16_bwd_dot_strip(const __bf16* __restrict__ dy, const uint8_t* __restrict__ idx,
                                                            const __bf16* __restrict__ x, float* __restrict__ ds, int planes,
                                                            int H, int W, int OH, int OW) {
  __shared__ double sm[16];
  const int w8 = W >> 3, strips = (H + PR16 - 1) / PR16;
  for (int pl = blockIdx.x; pl < planes; pl += gridDim.x) {
    const __bf16* xp = x + (int64_t)pl * H * W;
    double acc = 0.0;
    for (int i = threadIdx.x; i < strips * w8; i += 256) {
      const int b = i % w8, ih0 = (i / w8) * PR16;
      float G[PR16][8];
      pool16_strip<SH>(dy + (int64_t)pl * OH * OW, idx + (int64_t)pl * OH * OW, ih0, b, OH, OW, G);
      bf16x8 xv[PR16];
#pragma unroll
      for (int r = 0; r < PR16; ++r) xv[r] = *reinterpret_cast<const bf16x8*>(xp + (int64_t)min(ih0 + r, H - 1) * W + 8 * b);
#pragma unroll
      for (int r = 0; r < PR16; ++r) {
        if (ih0 + r >= H) continue;
        float f = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) f += G[r][c] * (float)xv[r][c];
        acc += (double)f;
      }
    }
    const double rsum = block_sum_d(acc, sm);
    if (threadIdx.x == 0) ds[pl] = (float)rsum;
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void gap16_fwd_kernel(const __bf16* __restrict__ x, int ctot, int coff,
                                                        float* __restrict__ out, int N, int C, int HW) {
  __shared__ double sm[16];
  for (int pl = blockIdx.x; pl < N * C; pl += gridDim.x) {
    const int n = pl / C, c = pl - n * C;
    const __bf16* xp = x + ((size_t)n * ctot + coff + c) * HW;
    double s = 0.0;
    for (int i = threadIdx.x; i < (HW >> 3); i += 256) {
      float v[8];
      ld8(xp + ((size_t)i << 3), v);
      s += (double)(((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7])));
    }
    const double r = block_sum_d(s, sm);
    if (threadIdx.x == 0) out[pl] = (float)(r / (double)HW);
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void gap16_bwd_kernel(const float* __restrict__ dout, __bf16* __restrict__ dx,
                                                        int64_t planes, int HW) {
  const int hw8 = HW >> 3;
  const int64_t total = planes * hw8;
  const float inv = 1.f / (float)HW;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const float g = dout[i / hw8] * inv;
    const float v[8] = {g, g, g, g, g, g, g, g};
    st8(dx + (i << 3), v);
  }
}

__global__ __launch_bounds__(256) void cast_f32_bf16_kernel(const float* __restrict__ x, __bf16* __restrict__ y, int64_t n8) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
    const float4 a = reinterpret_cast<const float4*>(x)[2 * i], b = reinterpret_cast<const float4*>(x)[2 * i + 1];
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    st8(y + (i << 3), v);
  }
}

__global__ __launch_bounds__(256) void cast_bf16_f32_kernel(const __bf16* __restrict__ x, float* __restrict__ y, int64_t n8) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
    float v[8];
    ld8(x + (i << 3), v);
    reinterpret_cast<float4*>(y)[2 * i] = make_float4(v[0], v[1], v[2], v[3]);
    reinterpret_cast<float4*>(y)[2 * i + 1] = make_float4(v[4], v[5], v[6], v[7]);
  }
}

}  // namespace

extern "C" int dlio_bf16_stats_splits(int N, int C, int HW) {
  if (N <= 0 || C <= 0 || HW <= 0) return 0;
  return splits16(N, C, HW);
}

extern "C" size_t dlio_bf16_stats_ws_bytes(int N, int C, int HW) {
  if (N <= 0 || C <= 0 || HW <= 0) return 0;
  return (size_t)C * splits16(N, C, HW) * 2 * sizeof(double);
}

// train: statistics + apply (2 launches); eval_mode != 0: apply only with the given mean / invstd / scale
extern "C" int dlio_bn_bf16_apply(const void* x, int N, int x_ctot, int x_coff, int C, int HW, int post_relu,
                                  const float* gamma, const float* beta, float eps, float momentum,
                                  float* running_mean, float* running_var, float* mean, float* invstd, float* scale,
                                  const void* residual, int r_ctot, int r_coff, void* y, int y_ctot, int y_coff,
                                  float* gap_out, int gap_ctot, int gap_coff, int eval_mode, void* ws, size_t ws_bytes,
                                  int phase, double count_scale, dlio_stream_t stream) {
  if (!x || !y || !mean || !invstd || !scale || N <= 0 || C <= 0 || HW <= 0 || !ws) return DLIO_EINVAL;
  if (phase < 0 || phase > 2 || !(count_scale > 0.0)) return DLIO_EINVAL;
  if (HW & 7) return DLIO_EUNSUP;
  const int splits = splits16(N, C, HW);
  if (ws_bytes < (size_t)C * splits * 2 * sizeof(double)) return DLIO_EWS;
  hipStream_t s = as_stream(stream);
  double* part = reinterpret_cast<double*>(ws);
  const __bf16* xb = reinterpret_cast<const __bf16*>(x);
  const double tensor_bytes = 2.0 * N * (double)C * HW;
  if (!eval_mode && phase != 2) {
    DlioProfScope prof(6, s, 0.0, tensor_bytes);
    hipLaunchKernelGGL(bn16_reduce_kernel<0>, dim3((unsigned)(C * splits)), dim3(RB), 0, s, xb, x_ctot, x_coff,
                       (const __bf16*)nullptr, 0, 0, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr,
                       (const float*)nullptr, N, C, HW, 0, splits, part);
    const int rc = dlio_check_launch();
    if (rc || phase == 1) return rc;
  }
  int chunks, chunk_len;
  plane_chunks16(N * C, HW / 8, gap_out != nullptr, chunks, chunk_len);
  DlioProfScope prof(7, s, 0.0, tensor_bytes * (residual ? 3.0 : 2.0));
  hipLaunchKernelGGL(bn16_plane_apply_kernel, dim3((unsigned)(N * C * chunks)), dim3(256), 0, s, xb, x_ctot, x_coff, part,
                     splits, (double)N * HW * count_scale, gamma, beta, eps, momentum, running_mean, running_var, mean, invstd, scale,
                     reinterpret_cast<const __bf16*>(residual), r_ctot, r_coff, reinterpret_cast<__bf16*>(y), y_ctot,
                     y_coff, N, C, HW, post_relu, chunks, chunk_len, gap_out, gap_ctot, gap_coff, eval_mode);
  return dlio_check_launch();
}

extern "C" int dlio_bn_bf16_bwd(const void* dy, int dy_ctot, int dy_coff, const void* x, int x_ctot, int x_coff,
                                const float* mean, const float* invstd, const float* scale, const float* beta, void* dx,
                                int dx_ctot, int dx_coff, float* dgamma, float* dbeta, int accumulate, int N, int C,
                                int HW, int post_relu, int use_batch_stats, void* ws, size_t ws_bytes, int phase,
                                double count_scale, const void* local_ws, dlio_stream_t stream) {
  if (!dy || !x || !mean || !invstd || !scale || !dx || N <= 0 || C <= 0 || HW <= 0 || !ws) return DLIO_EINVAL;
  if (phase < 0 || phase > 2 || !(count_scale > 0.0)) return DLIO_EINVAL;
  if (HW & 7) return DLIO_EUNSUP;
  const int splits = splits16(N, C, HW);
  if (ws_bytes < (size_t)C * splits * 2 * sizeof(double)) return DLIO_EWS;
  hipStream_t s = as_stream(stream);
  double* part = reinterpret_cast<double*>(ws);
  const __bf16* gb = reinterpret_cast<const __bf16*>(dy);
  const __bf16* xb = reinterpret_cast<const __bf16*>(x);
  const double tensor_bytes = 2.0 * N * (double)C * HW;
  if (phase != 2) {
    DlioProfScope prof(8, s, 0.0, 2.0 * tensor_bytes);
    hipLaunchKernelGGL(bn16_reduce_kernel<1>, dim3((unsigned)(C * splits)), dim3(RB), 0, s, gb, dy_ctot, dy_coff, xb,
                       x_ctot, x_coff, mean, invstd, scale, beta, N, C, HW, post_relu, splits, part);
    const int rc = dlio_check_launch();
    if (rc || phase == 1) return rc;
  }
  int chunks, chunk_len;
  plane_chunks16(N * C, HW / 8, false, chunks, chunk_len);
  DlioProfScope prof(9, s, 0.0, 3.0 * tensor_bytes);
  hipLaunchKernelGGL(bn16_plane_bwd_kernel, dim3((unsigned)(N * C * chunks)), dim3(256), 0, s, gb, dy_ctot, dy_coff, xb,
                     x_ctot, x_coff, mean, invstd, scale, beta, part, reinterpret_cast<const double*>(local_ws),
                     1.0 / ((double)N * HW * count_scale), splits,
                     reinterpret_cast<__bf16*>(dx), dx_ctot, dx_coff, dgamma, dbeta, accumulate, N, C, HW, post_relu,
                     use_batch_stats, chunks, chunk_len);
  return dlio_check_launch();
}

// rolling-window kernels (DLIO_POOL16_STRIP, default 1) vs one output row / one element per thread
static bool pool16_strips() {
  static const int v = 1;
  return v != 0;
}

static bool pool16_ok(int H, int W, int OH, int OW, int K, int SH, int SW, int PH, int PW) {
  return K == 3 && SW == 2 && PH == 1 && PW == 1 && (SH == 1 || SH == 2) && (W & 15) == 0 && OW * 2 == W &&
         OH == (H + 2 - 3) / SH + 1;
}

extern "C" int dlio_maxpool_bf16_fwd(const void* x, const float* x_scale, void* y, uint8_t* idx, int N, int C, int H,
                                     int W, int OH, int OW, int K, int SH, int SW, int PH, int PW, dlio_stream_t stream) {
  if (!x || !y || N <= 0 || C <= 0) return DLIO_EINVAL;
  if (!pool16_ok(H, W, OH, OW, K, SH, SW, PH, PW)) return DLIO_EUNSUP;
  hipStream_t s = as_stream(stream);
  const int64_t planes = (int64_t)N * C, work = planes * OH * (OW / 8);
  DlioProfScope prof(10, s, 0.0, (double)planes * (2.0 * H * W + (idx ? 3.0 : 2.0) * OH * OW));
  if (pool16_strips()) {
    const int64_t wk = planes * cdiv(OH, SH == 1 ? 8 : 4) * (OW / 4);
    if (SH == 1)
      hipLaunchKernelGGL(pool16_fwd_strip<1>, dim3(ew_grid(wk, 256)), dim3(256), 0, s, reinterpret_cast<const __bf16*>(x),
                         x_scale, reinterpret_cast<__bf16*>(y), idx, planes, H, W, OH, OW);
    else
      hipLaunchKernelGGL(pool16_fwd_strip<2>, dim3(ew_grid(wk, 256)), dim3(256), 0, s, reinterpret_cast<const __bf16*>(x),
                         x_scale, reinterpret_cast<__bf16*>(y), idx, planes, H, W, OH, OW);
    return dlio_check_launch();
  }
  if (SH == 1)
    hipLaunchKernelGGL(pool16_fwd_kernel<1>, dim3(ew_grid(work, 256)), dim3(256), 0, s, reinterpret_cast<const __bf16*>(x),
                       x_scale, reinterpret_cast<__bf16*>(y), idx, planes, H, W, OH, OW);
  else
    hipLaunchKernelGGL(pool16_fwd_kernel<2>, dim3(ew_grid(work, 256)), dim3(256), 0, s, reinterpret_cast<const __bf16*>(x),
                       x_scale, reinterpret_cast<__bf16*>(y), idx, planes, H, W, OH, OW);
  return dlio_check_launch();
}

extern "C" int dlio_maxpool_bf16_bwd(const void* dy, const uint8_t* idx, const float* x_scale, const float* x_add,
                                     void* dx, int N, int C, int H, int W, int OH, int OW, int K, int SH, int SW, int PH,
                                     int PW, dlio_stream_t stream) {
  if (!dy || !idx || !dx || N <= 0 || C <= 0) return DLIO_EINVAL;
  if (!pool16_ok(H, W, OH, OW, K, SH, SW, PH, PW)) return DLIO_EUNSUP;
  hipStream_t s = as_stream(stream);
  const int64_t planes = (int64_t)N * C, work = planes * H * (W / 8);
  DlioProfScope prof(10, s, 0.0, (double)planes * (2.0 * H * W + 3.0 * OH * OW));
  if (pool16_strips()) {
    const int64_t wk = planes * cdiv(H, PR16) * (W / 8);
    if (SH == 1)
      hipLaunchKernelGGL(pool16_bwd_strip<1>, dim3(ew_grid(wk, 256)), dim3(256), 0, s, reinterpret_cast<const __bf16*>(dy),
                         idx, x_scale, x_add, reinterpret_cast<__bf16*>(dx), planes, H, W, OH, OW);
    else
      hipLaunchKernelGGL(pool16_bwd_strip<2>, dim3(ew_grid(wk, 256)), dim3(256), 0, s, reinterpret_cast<const __bf16*>(dy),
                         idx, x_scale, x_add, reinterpret_cast<__bf16*>(dx), planes, H, W, OH, OW);
    return dlio_check_launch();
  }
  if (SH == 1)
    hipLaunchKernelGGL(pool16_bwd_kernel<1>, dim3(ew_grid(work, 256)), dim3(256), 0, s, reinterpret_cast<const __bf16*>(dy),
                       idx, x_scale, x_add, reinterpret_cast<__bf16*>(dx), planes, H, W, OH, OW);
  else
    hipLaunchKernelGGL(pool16_bwd_kernel<2>, dim3(ew_grid(work, 256)), dim3(256), 0, s, reinterpret_cast<const __bf16*>(dy),
                       idx, x_scale, x_add, reinterpret_cast<__bf16*>(dx), planes, H, W, OH, OW);
  return dlio_check_launch();
}

extern "C" int dlio_maxpool_bf16_bwd_dot(const void* dy, const uint8_t* idx, const void* x, float* ds, int N, int C,
                                         int H, int W, int OH, int OW, int K, int SH, int SW, int PH, int PW,
                                         dlio_stream_t stream) {
  if (!dy || !idx || !x || !ds || N <= 0 || C <= 0) return DLIO_EINVAL;
  if (!pool16_ok(H, W, OH, OW, K, SH, SW, PH, PW)) return DLIO_EUNSUP;
  hipStream_t s = as_stream(stream);
  int grid = N * C;
  if (grid > 65535) grid = 65535;
  DlioProfScope prof(10, s, 0.0, (double)N * C * (2.0 * H * W + 3.0 * OH * OW));
  if (pool16_strips()) {
    if (SH == 1)
      hipLaunchKernelGGL(pool16_bwd_dot_strip<1>, dim3(grid), dim3(256), 0, s, reinterpret_cast<const __bf16*>(dy), idx,
                         reinterpret_cast<const __bf16*>(x), ds, N * C, H, W, OH, OW);
    else
      hipLaunchKernelGGL(pool16_bwd_dot_strip<2>, dim3(grid), dim3(256), 0, s, reinterpret_cast<const __bf16*>(dy), idx,
                         reinterpret_cast<const __bf16*>(x), ds, N * C, H, W, OH, OW);
    return dlio_check_launch();
  }
  if (SH == 1)
    hipLaunchKernelGGL(pool16_bwd_dot_kernel<1>, dim3(grid), dim3(256), 0, s, reinterpret_cast<const __bf16*>(dy), idx,
                       reinterpret_cast<const __bf16*>(x), ds, N * C, H, W, OH, OW);
  else
    hipLaunchKernelGGL(pool16_bwd_dot_kernel<2>, dim3(grid), dim3(256), 0, s, reinterpret_cast<const __bf16*>(dy), idx,
                       reinterpret_cast<const __bf16*>(x), ds, N * C, H, W, OH, OW);
  return dlio_check_launch();
}

extern "C" int dlio_gap_bf16_fwd(const void* x, int ctot, int coff, float* out, int N, int C, int HW,
                                 dlio_stream_t stream) {
  if (!x || !out || N <= 0 || C <= 0 || HW <= 0) return DLIO_EINVAL;
  if (HW & 7) return DLIO_EUNSUP;
  int grid = N * C;
  if (grid > 65535) grid = 65535;
  hipLaunchKernelGGL(gap16_fwd_kernel, dim3(grid), dim3(256), 0, as_stream(stream), reinterpret_cast<const __bf16*>(x),
                     ctot, coff, out, N, C, HW);
  return dlio_check_launch();
}

extern "C" int dlio_gap_bf16_bwd(const float* dout, void* dx, int N, int C, int HW, dlio_stream_t stream) {
  if (!dout || !dx || N <= 0 || C <= 0 || HW <= 0) return DLIO_EINVAL;
  if (HW & 7) return DLIO_EUNSUP;
  const int64_t total = (int64_t)N * C * (HW / 8);
  hipLaunchKernelGGL(gap16_bwd_kernel, dim3(ew_grid(total, 256)), dim3(256), 0, as_stream(stream), dout,
                     reinterpret_cast<__bf16*>(dx), (int64_t)N * C, HW);
  return dlio_check_launch();
}

// dir 0: fp32 -> bf16 (round to nearest even), dir 1: bf16 -> fp32 (exact); n % 8 == 0
extern "C" int dlio_cast_bf16(const void* src, void* dst, int64_t n, int dir, dlio_stream_t stream) {
  if (!src || !dst || n <= 0 || (dir != 0 && dir != 1)) return DLIO_EINVAL;
  if (n & 7) return DLIO_EUNSUP;
  const int64_t n8 = n >> 3;
  if (dir == 0)
    hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(ew_grid(n8, 256)), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const float*>(src), reinterpret_cast<__bf16*>(dst), n8);
  else
    hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3(ew_grid(n8, 256)), dim3(256), 0, as_stream(stream),
                       reinterpret_cast<const __bf16*>(src), reinterpret_cast<float*>(dst), n8);
  return dlio_check_launch();
}
