// Fire expand pair (pointseg_modules.py:126-133: cat([expand1x1(s), expand3x3(s)], 1)) as ONE launch on the split-bf16
// scheme of conv_bx3.hip, fed by a squeeze tensor that is split into its three bf16 pieces ONCE, where its BatchNorm +
// ReLU is applied:
//
//   bn_split16_kernel      raw squeeze output -> activated fp32 tensor (the weight gradients read it) AND the
//                          position-major planes  sp[n][chunk][piece][H + 2][W + 2][16 channels]  (bf16, zero border:
//                          the 3x3 padding is physical).  6 B per element on the SMALL tensor of a Fire block (S = CE / 8).
//   fire_expand_fwd_kernel a workgroup = 4 waves = 4 output rows x 32 TWN columns x 32 MR channels of expand3x3 AND the
//                          same 32 MR channels of expand1x1.  The 6 x (32 TWN + 2) patch of a 16-channel chunk goes
//                          global -> LDS with buffer_load ... lds (no gather, no split VALU, no staging registers --
//                          conv3x3_bx3_alds_kernel spends ~350 VALU per thread and chunk there); expand1x1 is a tenth
//                          tap on the centre fragments of the same patch (+1/9 MFMA work, no extra LDS reads of the
//                          patch), its weight fragments ride in the weight ring beside the centre tap's.  Both halves of
//                          the concat buffer are written from one epilogue.
#include "common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split3(float x, __bf16& h, __bf16& m, __bf16& l) {
  h = (__bf16)x;
  const float r = x - (float)h;
  m = (__bf16)r;
  l = (__bf16)(r - (float)m);
}

// ---- BatchNorm (+ ReLU) apply of the squeeze output with the split planes as a second output -------------------------
// grid (pixel runs, chunk, n).  A lane owns one pixel and 8 of the chunk's 16 channels (lanes 0..31: 32 consecutive pixels,
// channels 0..7; lanes 32..63: the same pixels, channels 8..15) -- the unit the planes are made of: a wave's store of one
// plane covers 32 positions x 32 B = 1 KB contiguous (a lane with four pixels x 16 channels wrote 16 B at a 128 B stride:
// every store instruction touched 64 lines; 44 -> 28 us at blk1), its loads 128 B per channel row and half.
// Train mode (part != nullptr): the workgroup first sums the split partials of its 16 channels (fixed order: every
// workgroup of a chunk computes bit-identical statistics) and finalises them, workgroup (0, kc, 0) publishes mean / invstd /
// scale and the running statistics -- bn_plane_apply_kernel's scheme.  Eval mode: mean / scale are inputs.
constexpr int SPLIT_U = 4;          // pixels per lane
__global__ __launch_bounds__(256) void bn_split16_kernel(
    const float* __restrict__ x, int x_ctot, int x_coff, const double* __restrict__ part, int splits, double count,
    const float* __restrict__ gamma, const float* __restrict__ beta, float eps, float momentum, float* running_mean,
    float* running_var, float* mean_io, float* invstd_o, float* scale_io, float* y, int y_ctot, int y_coff,
    __bf16* __restrict__ planes, int N, int C, int H, int W, int post_relu, int h2, float* __restrict__ bound_out,
    const float* __restrict__ mm = nullptr) {
  __shared__ float tab[16][4];              // mean, scale, beta
  __shared__ float wmax[4];
  const int kc = blockIdx.y, n = blockIdx.z, KC = gridDim.y;
  const int tid = threadIdx.x;
  // h2 (train mode): two fp16 pieces of x * 2^k instead of three bf16 pieces of x.  2^k from a bound that needs no pass:
  // with batch statistics |x - mean| / sqrt(var + eps) <= sqrt(count) for every element, so |BN(x)| <= |beta| + |gamma|
  // sqrt(count); the bound is mapped to 2^14 (fp16 max 65504).  It overshoots the true maximum by ~2^7: elements down to 1e-3
  // of the maximum keep 2^-22 relative accuracy, smaller ones an absolute error of 2^-32 of the maximum.
  // Round 6: with the statistics partials comes the exact RANGE of every channel (mm: min / max per split, left by the
  // statistics pass that reads the tensor anyway).  BatchNorm is monotone per channel, so the largest |BN(x)| of the tensor is
  // max over channels of max(|BN(min)|, |BN(max)|) (behind the ReLU: max(BN(min), BN(max), 0)) -- the exact maximum instead
  // of a bound 2^7 ... 2^9 above it: elements down to 1e-3 of the TRUE maximum keep 2^-22, channels four to five decades below
  // the largest one keep 1e-4 relative accuracy (tests/test_gpu_ops.py: six-decade test).  Every workgroup finalises every
  // channel's statistics for it (16 threads per channel, chunks of 16 channels; the same summation order as below).
  float hs = 1.f;
  if (h2) {
    float b = 0.f;
    if (mm && part) {
      const int cl = tid >> 4, sub = tid & 15;
      for (int cb = 0; cb < C; cb += 16) {
        const int c = cb + cl;
        if (c < C) {
          double sa = 0.0, sq = 0.0;
          float lo = 3.0e38f, hi = -3.0e38f;
          for (int q = sub; q < splits; q += 16) {
            sa += part[((size_t)c * splits + q) * 2 + 0];
            sq += part[((size_t)c * splits + q) * 2 + 1];
            lo = fminf(lo, mm[((size_t)c * splits + q) * 2 + 0]);
            hi = fmaxf(hi, mm[((size_t)c * splits + q) * 2 + 1]);
          }
#pragma unroll
          for (int o = 8; o > 0; o >>= 1) {
            sa += __shfl_xor(sa, o, 16); sq += __shfl_xor(sq, o, 16);
            lo = fminf(lo, __shfl_xor(lo, o, 16)); hi = fmaxf(hi, __shfl_xor(hi, o, 16));
          }
          const double m = sa / count;
          double var = sq / count - m * m;
          if (var < 0.0) var = 0.0;
          const float is = (float)(1.0 / sqrt(var + (double)eps));
          const float mu = (float)m, sc = (gamma ? gamma[c] : 1.f) * is, be = beta ? beta[c] : 0.f;
          const float v1 = (lo - mu) * sc + be, v2 = (hi - mu) * sc + be;
          b = fmaxf(b, post_relu ? fmaxf(fmaxf(v1, v2), 0.f) : fmaxf(fabsf(v1), fabsf(v2)));
        }
      }
    } else {
      const float rc = sqrtf((float)count);
      for (int c = tid; c < C; c += 256) b = fmaxf(b, fabsf(beta ? beta[c] : 0.f) + fabsf(gamma ? gamma[c] : 1.f) * rc);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) b = fmaxf(b, __shfl_xor(b, o, 64));
    if ((tid & 63) == 0) wmax[tid >> 6] = b;
    __syncthreads();
    b = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
    hs = (b > 0.f && b < 3.0e38f) ? exp2f(floorf(log2f(16384.f / b))) : 1.f;
    if (tid == 0 && blockIdx.x == 0 && kc == 0 && n == 0) {
      // the inverse scale travels behind the planes
      float* tail = reinterpret_cast<float*>(planes + (size_t)N * KC * 2 * ((size_t)(H + 2) * (W + 2) * 16));
      tail[0] = 1.f / hs;
      if (bound_out) bound_out[0] = b;                  // for later consumers of y on the two-piece kernels (weight gradient)
    }
  }
  {
    const int cl = tid >> 4, sub = tid & 15, c = kc * 16 + cl;       // 16 threads per channel
    float mu = 0.f, sc = 0.f, be = 0.f;
    if (c < C) {
      if (part) {
        double a = 0.0, b = 0.0;
        for (int q = sub; q < splits; q += 16) {
          a += part[((size_t)c * splits + q) * 2 + 0];
          b += part[((size_t)c * splits + q) * 2 + 1];
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) { a += __shfl_xor(a, o, 16); b += __shfl_xor(b, o, 16); }
        const double m = a / count;
        double var = b / count - m * m;
        if (var < 0.0) var = 0.0;
        const float is = (float)(1.0 / sqrt(var + (double)eps));
        mu = (float)m; sc = (gamma ? gamma[c] : 1.f) * is;
        if (sub == 0 && blockIdx.x == 0 && n == 0) {
          mean_io[c] = mu; invstd_o[c] = is; scale_io[c] = sc;
          if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mu;
          if (running_var) {
            const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
            running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
          }
        }
      } else {
        mu = mean_io[c]; sc = scale_io[c];
      }
      be = beta ? beta[c] : 0.f;
    }
    if (sub == 0) { tab[cl][0] = mu; tab[cl][1] = sc; tab[cl][2] = be; tab[cl][3] = 0.f; }
  }
  __syncthreads();
  const int HW = H * W, lane = tid & 63, l31 = lane & 31, half = lane >> 5, wave = tid >> 6;
  const int WP = W + 2;
  const size_t pstride = (size_t)(H + 2) * WP * 16;                    // bf16 per plane
  __bf16* pb = planes + ((size_t)n * KC + kc) * (h2 ? 2 : 3) * pstride + 8 * half;
  const float* xb = x + ((size_t)n * x_ctot + x_coff + kc * 16 + 8 * half) * HW;
  float* yb = y ? y + ((size_t)n * y_ctot + y_coff + kc * 16 + 8 * half) * HW : nullptr;
  const int p0 = (blockIdx.x * 4 + wave) * (32 * SPLIT_U) + l31;
  float v[SPLIT_U][8];
#pragma unroll
  for (int u = 0; u < SPLIT_U; ++u) {
    const int p = p0 + 32 * u;
#pragma unroll
    for (int c = 0; c < 8; ++c) v[u][c] = (p < HW && kc * 16 + 8 * half + c < C) ? xb[(size_t)c * HW + p] : 0.f;
  }
  const bf16x8 z8 = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int u = 0; u < SPLIT_U; ++u) {
    const int p = p0 + 32 * u;
    if (p >= HW) continue;
    const int h = p / W, w = p - h * W;
    bf16x8 ph, pm, pl;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      float t = 0.f;
      if (kc * 16 + 8 * half + c < C) {
        const float mu = tab[8 * half + c][0], sc = tab[8 * half + c][1], be = tab[8 * half + c][2];
        t = (v[u][c] - mu) * sc + be;
        if (post_relu) t = fmaxf(t, 0.f);
        if (yb) yb[(size_t)c * HW + p] = t;
      }
      if (h2) {
        const float ts = t * hs;
        const _Float16 hh = (_Float16)ts;
        const _Float16 ll = (_Float16)(ts - (float)hh);
        ph[c] = __builtin_bit_cast(__bf16, hh); pm[c] = __builtin_bit_cast(__bf16, ll);
      } else {
        __bf16 hh, mm, ll;
        split3(t, hh, mm, ll);
        ph[c] = hh; pm[c] = mm; pl[c] = ll;
      }
    }
    __bf16* d = pb + ((size_t)(h + 1) * WP + (w + 1)) * 16;
    *reinterpret_cast<bf16x8*>(d) = ph;
    *reinterpret_cast<bf16x8*>(d + pstride) = pm;
    if (!h2) *reinterpret_cast<bf16x8*>(d + 2 * pstride) = pl;
    // zero border (the 3x3 padding), this lane's channel half: columns 0 / W + 1 beside its row, rows 0 / H + 1 above /
    // below its column, the corners with the edge columns
    auto zero_at = [&](int r, int c) {
      __bf16* z = pb + ((size_t)r * WP + c) * 16;
      *reinterpret_cast<bf16x8*>(z) = z8;
      *reinterpret_cast<bf16x8*>(z + pstride) = z8;
      if (!h2) *reinterpret_cast<bf16x8*>(z + 2 * pstride) = z8;
    };
    const bool left = w == 0, right = w == W - 1;
    if (left) zero_at(h + 1, 0);
    if (right) zero_at(h + 1, W + 1);
    if (h == 0) { zero_at(0, w + 1); if (left) zero_at(0, 0); if (right) zero_at(0, W + 1); }
    if (h == H - 1) { zero_at(H + 1, w + 1); if (left) zero_at(H + 1, 0); if (right) zero_at(H + 1, W + 1); }
  }
}

// ---- fused expand1x1 || expand3x3 forward ---------------------------------------------------------------------------
// weights: w3t [9][KC][3][E][16], w1t [1][KC][3][E][16] (dlio_conv_bx3_prep mode 0 of the two layers)
// ten tap slots in five groups of two; slot 5 (second of group 2, behind the centre tap) is expand1x1.
//
// Measured on the way (tools/bench_fire.py, N = 16; DESIGN has the table): expand1x1 rides along for nothing -- the fused
// launch takes what conv3x3_bx3_alds_kernel takes for expand3x3 alone (blk1 100 us against 89 + 31, blk3 127 / 220 against
// 118 + 33 / 188 + 43).  Timing ablations: the phases of a workgroup ADD UP (blk1: 29 us skeleton + 30 MFMA + 32 stores +
// 20 loads).  Three attempts to overlap them inside the launch were built and measured equal or slower, and are gone:
// a persistent tile loop with a double-buffered patch and a three-slot ring (hides the loads: 51 us without stores, 109 with
// -- a wave is held at store ISSUE for as long as the memory system takes for the bytes, with or without s_waitcnt), the
// same with eight waves per workgroup, and two persistent workgroups per CU started in anti-phase through a per-CU ticket
// (HW_ID / XCC_ID + atomics): 113 us.
// Epilogue without LDS: the four registers r & 3 = 0..3 of a lane quad (four neighbouring pixels) are transposed with two
// DPP butterfly stages, one 16-byte store per four accumulator registers.
// STATS: the workgroup also leaves the sum and the sum of squares of every output channel over its tile in
// stats[workgroup][2][128] (expand1x1 channels co0..co0+63, then expand3x3's): the BatchNorm statistics of an apply-on-load
// block without a pass over the concat buffer (fire_stats_finalize_kernel adds the tiles in index order, in fp64).
// H2: the planes and the weights are TWO fp16 pieces of x 2^k (bn_split16_kernel h2, prep_h2_*): three
// v_mfma_f32_32x32x16_f16 per product (lh, hl, hh) instead of six bf16 ones, two planes through LDS instead of three; the
// epilogue multiplies by the two inverse scales (powers of two: exact).
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <int TWN, bool STATS = false, bool H2 = false>
__global__ __launch_bounds__(256, 2) void fire_expand_fwd_kernel(
    const __bf16* __restrict__ planes, const __bf16* __restrict__ w3t, const __bf16* __restrict__ w1t,
    const float* __restrict__ bias3, const float* __restrict__ bias1, float* __restrict__ y, int N, int KC, int H, int W,
    int E, int y_ctot, int y_coff, int tiles_w, int tiles_h, int co_tiles, float* __restrict__ stats = nullptr,
    const float* __restrict__ inv_a = nullptr, const float* __restrict__ inv_w3 = nullptr,
    const float* __restrict__ inv_w1 = nullptr) {
  constexpr int MR = 2;
  constexpr int NPL = H2 ? 2 : 3;                        // planes (pieces) per operand
  constexpr int TH = 4, TW = 32 * TWN, PR = TH + 2, PC = TW + 2, NPOSP = PR * PC;
  constexpr int PLANE = NPOSP * 16;                      // bf16 per patch plane
  constexpr int PP = NPOSP * 2;                          // 16-byte pieces per plane
  constexpr int PINS = (NPL * PP + 63) / 64;             // wave instructions per patch
  constexpr int PPER = (PINS + 3) / 4;                   // per wave
  constexpr int PATCH_B = PINS * 1024;                   // bytes (padded to whole instructions)
  constexpr int AROWS = 32 * MR;
  constexpr int TG = 2;                                  // tap slots per weight group
  constexpr int AGRP = TG * NPL * AROWS * 16;            // bf16 per ring slot: [slot][plane][row][16 k]
  constexpr int AINS = TG * NPL * MR;                    // wave instructions per group
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __bf16* patch = reinterpret_cast<__bf16*>(smem_raw);   // [3 planes][NPOSP][16]
  __bf16* ring = reinterpret_cast<__bf16*>(smem_raw + PATCH_B);   // [2][AGRP]

  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // (scalar: LDS-DMA destinations and the load dealing are per wave)
  int bid = xcd_block_index();
  const int cot = bid % co_tiles; bid /= co_tiles;
  const int tw = bid % tiles_w; bid /= tiles_w;
  const int th = bid % tiles_h;
  const int n = bid / tiles_h;
  const int co0 = cot * 32 * MR, oh0 = th * TH, ow0 = tw * TW;
  const int WP = W + 2;
  const unsigned pstride_b = (unsigned)(H + 2) * (unsigned)WP * 32u;          // bytes per plane
  const size_t planes_b = (size_t)N * KC * NPL * pstride_b;

  // ---- patch: piece q = plane * PP + r * (2 PC) + cc lands at LDS byte 16 q; its global byte offset within the chunk
  const __amdgpu_buffer_rsrc_t prsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(planes), 0, (int)planes_b, 0x00020000);
  unsigned pvoff[PPER];
#pragma unroll
  for (int i = 0; i < PPER; ++i) {
    const int q = (i * 4 + wave) * 64 + lane;
    const int pl = q / PP, rem = q - pl * PP, r = rem / (2 * PC), cc = rem - r * (2 * PC);
    pvoff[i] = q < NPL * PP ? (unsigned)pl * pstride_b + (unsigned)(r * WP) * 32u + (unsigned)cc * 16u : 0xfffffff0u;
  }
  const unsigned tile_b = (unsigned)(oh0 * WP + ow0) * 32u;
  auto load_patch = [&](int kc) {
    const unsigned so = (unsigned)((n * KC + kc) * NPL) * pstride_b + tile_b;
#pragma unroll
    for (int i = 0; i < PPER; ++i) {
      const int t = i * 4 + wave;
      if (t < PINS) {
#if defined(__HIP_DEVICE_COMPILE__)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(prsrc, (__attribute__((address_space(3))) void*)(smem_raw + (size_t)t * 1024), 16,
                                                 pvoff[i], so, 0, 0);
#else
        (void)so;
#endif
      }
    }
  };

  // ---- weight groups
  const size_t wplane = (size_t)E * 16;
  const __amdgpu_buffer_rsrc_t w3rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<__bf16*>(w3t), 0, (int)((size_t)9 * KC * NPL * wplane * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t w1rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<__bf16*>(w1t), 0, (int)((size_t)KC * NPL * wplane * 2), 0x00020000);
  unsigned awoff[MR];
#pragma unroll
  for (int m = 0; m < MR; ++m) awoff[m] = ((unsigned)min(co0 + m * 32 + (lane >> 1), E - 1) * 16u + 8u * (unsigned)(lane & 1)) * 2u;
  auto load_agroup = [&](int kc, int g, int slot) {
#pragma unroll
    for (int i = 0; i < (AINS + 3) / 4; ++i) {
      const int t = i * 4 + wave;                        // instruction t = (j * 3 + plane) * MR + m
      if (t < AINS) {
        const int m = t % MR, tp = t / MR, j = tp / NPL, pl = tp - NPL * j;
        const int s = g * TG + j;                        // tap slot 0..9; 5 = expand1x1
        __bf16* dst = ring + (size_t)slot * AGRP + (size_t)t * 64 * 8;
#if defined(__HIP_DEVICE_COMPILE__)
        if (s == 5)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(w1rsrc, (__attribute__((address_space(3))) void*)dst, 16, awoff[m],
                                                   (unsigned)(((size_t)kc * NPL + pl) * wplane * 2), 0, 0);
        else
          __builtin_amdgcn_raw_ptr_buffer_load_lds(w3rsrc, (__attribute__((address_space(3))) void*)dst, 16, awoff[m],
                                                   (unsigned)((((size_t)(s < 5 ? s : s - 1) * KC + kc) * NPL + pl) * wplane * 2), 0, 0);
#else
        (void)dst; (void)m; (void)pl; (void)s;
#endif
      }
    }
  };

  f32x16 acc3[MR][TWN], acc1[MR][TWN];
#pragma unroll
  for (int m = 0; m < MR; ++m)
#pragma unroll
    for (int t = 0; t < TWN; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc3[m][t][r] = 0.f; acc1[m][t][r] = 0.f; }
  auto read_a = [&](const __bf16* slot, int j, bf16x8 (&a)[MR][NPL]) {
#pragma unroll
    for (int m = 0; m < MR; ++m)
#pragma unroll
      for (int p = 0; p < NPL; ++p)
        a[m][p] = *reinterpret_cast<const bf16x8*>(slot + ((j * NPL + p) * AROWS + m * 32 + l31) * 16 + 8 * half);
  };
  auto read_b = [&](int tap, bf16x8 (&b)[TWN][NPL]) {
    const int kh = tap / 3, kw = tap - 3 * kh;
#pragma unroll
    for (int t = 0; t < TWN; ++t) {
      const int pos = (wave + kh) * PC + 32 * t + l31 + kw;
#pragma unroll
      for (int p = 0; p < NPL; ++p) b[t][p] = *reinterpret_cast<const bf16x8*>(patch + p * PLANE + pos * 16 + 8 * half);
    }
  };
  constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};
  auto mfma_slot = [&](f32x16 (&acc)[MR][TWN], const bf16x8 (&a)[MR][NPL], const bf16x8 (&b)[TWN][NPL]) {
    if constexpr (H2) {
      constexpr int HA[3] = {1, 0, 0}, HB[3] = {0, 1, 0};              // (lo, hi) (hi, lo) (hi, hi): smallest first
#pragma unroll
      for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int m = 0; m < MR; ++m)
#pragma unroll
          for (int t = 0; t < TWN; ++t)
            acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[m][HA[q]]),
                                                               __builtin_bit_cast(f16x8, b[t][HB[q]]), acc[m][t], 0, 0, 0);
    } else {
#pragma unroll
      for (int q = DLIO_SPLIT_Q0; q < 6; ++q)
#pragma unroll
        for (int m = 0; m < MR; ++m)
#pragma unroll
          for (int t = 0; t < TWN; ++t)
            acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m][PA[q]], b[t][PB[q]], acc[m][t], 0, 0, 0);
    }
  };
  auto compute_group = [&](const __bf16* slot, int g) {
    bf16x8 a[2][MR][NPL], b[TWN][NPL];
    read_a(slot, 0, a[0]);
    read_a(slot, 1, a[1]);
    const int s0 = g * TG;
    read_b(s0 < 5 ? s0 : s0 - 1, b);
    mfma_slot(acc3, a[0], b);
    if (g == 2) {
      mfma_slot(acc1, a[1], b);                          // expand1x1 on the centre fragments already in registers
    } else {
      read_b(s0 + 1 < 5 ? s0 + 1 : s0, b);
      mfma_slot(acc3, a[1], b);
    }
  };

  // ---- pipeline: weight groups double-slotted, patch single-buffered (its loads are asynchronous: the other workgroup
  // of the CU covers them)
  load_patch(0);
  load_agroup(0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int slot = 0;
  for (int kc = 0; kc < KC; ++kc) {
    const bool more = kc + 1 < KC;
#pragma unroll
    for (int g = 0; g < 5; ++g) {
      // (everyone is past the previous group's barrier: the other ring slot is free)
      if (g < 4) load_agroup(kc, g + 1, slot ^ 1);
      else if (more) load_agroup(kc + 1, 0, slot ^ 1);
      compute_group(ring + (size_t)slot * AGRP, g);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      slot ^= 1;
    }
    if (more) {
      load_patch(kc + 1);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  }

  // ---- epilogue: D tile col = pixel (lane & 31), row = (r & 3) + 8 (r >> 2) + 4 half; after the quad transpose lane j of a
  // quad holds channel 8 (r >> 2) + 4 half + j of the quad's four pixels.  expand1x1 -> channels [y_coff, y_coff + E),
  // expand3x3 -> [y_coff + E, y_coff + 2 E)
  const int oh = oh0 + wave;
  if (!STATS && oh >= H) return;
  const bool row_ok = oh < H;
  float st1[2][MR][4], st2[2][MR][4];              // STATS: per (layer, channel tile, register quad) = per channel of this lane
  if constexpr (STATS) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int m = 0; m < MR; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q) st1[a][m][q] = st2[a][m][q] = 0.f;
  }
  const size_t hw = (size_t)H * W;
  float* yb = y + ((size_t)n * y_ctot + y_coff) * hw + (size_t)oh * W + ow0 + 4 * (l31 >> 2);
  const bool o1 = (lane & 1) != 0, o2 = (lane & 2) != 0;
  auto swz = [](float v, auto ctrl) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), decltype(ctrl)::value, 0xf, 0xf, true));
  };
  using X1 = std::integral_constant<int, 0xB1>;         // quad_perm [1, 0, 3, 2]
  using X2 = std::integral_constant<int, 0x4E>;         // quad_perm [2, 3, 0, 1]
  float isc[2] = {1.f, 1.f};
  if constexpr (H2) { const float ia = inv_a[0]; isc[0] = ia * inv_w1[0]; isc[1] = ia * inv_w3[0]; }
#pragma unroll
  for (int set = 0; set < 2; ++set) {
    const float* bias = set == 0 ? bias1 : bias3;
#pragma unroll
    for (int m = 0; m < MR; ++m)
#pragma unroll
      for (int t = 0; t < TWN; ++t)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          const f32x16& A = set == 0 ? acc1[m][t] : acc3[m][t];
          float a0 = A[4 * rq], a1 = A[4 * rq + 1], a2 = A[4 * rq + 2], a3 = A[4 * rq + 3];
          { const float x = o1 ? a0 : a1, yv = swz(x, X1{}); a0 = o1 ? yv : a0; a1 = o1 ? a1 : yv; }
          { const float x = o1 ? a2 : a3, yv = swz(x, X1{}); a2 = o1 ? yv : a2; a3 = o1 ? a3 : yv; }
          { const float x = o2 ? a0 : a2, yv = swz(x, X2{}); a0 = o2 ? yv : a0; a2 = o2 ? a2 : yv; }
          { const float x = o2 ? a1 : a3, yv = swz(x, X2{}); a1 = o2 ? yv : a1; a3 = o2 ? a3 : yv; }
          const int co = co0 + 32 * m + 8 * rq + 4 * half + (lane & 3);
          if (row_ok && co < E && ow0 + 32 * t + 4 * (l31 >> 2) < W) {
            const float bv = bias ? bias[co] : 0.f;
            if constexpr (H2) { a0 *= isc[set]; a1 *= isc[set]; a2 *= isc[set]; a3 *= isc[set]; }
            a0 += bv; a1 += bv; a2 += bv; a3 += bv;
            st4<1>(yb + ((size_t)(set == 0 ? 0 : E) + co) * hw + 32 * t, make_float4(a0, a1, a2, a3));
            if constexpr (STATS) {
              st1[set][m][rq] += (a0 + a1) + (a2 + a3);
              st2[set][m][rq] += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
            }
          }
        }
  }
  if constexpr (STATS) {
    // the eight lanes 4 j + (lane & 3) of a half hold the same channel: add them (xor 4, 8, 16), the four waves (rows)
    // through LDS (the patch is no longer read: every wave is past the last group's barrier), one 1 KB row per workgroup
    float* sl = reinterpret_cast<float*>(smem_raw);          // [4 waves][2][128]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int m = 0; m < MR; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float u = st1[a][m][q], v = st2[a][m][q];
#pragma unroll
          for (int o = 4; o <= 16; o <<= 1) { u += __shfl_xor(u, o, 64); v += __shfl_xor(v, o, 64); }
          if ((l31 >> 2) == 0) {
            const int ch = a * 64 + 32 * m + 8 * q + 4 * half + (lane & 3);
            sl[(wave * 2 + 0) * 128 + ch] = u;
            sl[(wave * 2 + 1) * 128 + ch] = v;
          }
        }
    __syncthreads();
    const int which = tid >> 7, ch = tid & 127;              // 256 threads = 2 x 128
    const float tot = (sl[(0 * 2 + which) * 128 + ch] + sl[(1 * 2 + which) * 128 + ch]) +
                      (sl[(2 * 2 + which) * 128 + ch] + sl[(3 * 2 + which) * 128 + ch]);
    const size_t wg = ((size_t)(n * tiles_h + th) * tiles_w + tw) * co_tiles + cot;
    stats[wg * 256 + tid] = tot;
  }
}

// mean / invstd / scale / shift (+ running statistics) of the 2 E channels of a Fire block's concat buffer from the tile
// sums fire_expand_fwd_kernel<.., true> left: one workgroup per channel, tiles added in index order (fp64)
__global__ __launch_bounds__(256) void fire_stats_finalize_kernel(
    const float* __restrict__ stats, int tiles, int co_tiles, int E, double count, float eps, float momentum,
    const float* __restrict__ gamma1, const float* __restrict__ beta1, float* running_mean1, float* running_var1,
    const float* __restrict__ gamma3, const float* __restrict__ beta3, float* running_mean3, float* running_var3,
    float* __restrict__ mean, float* __restrict__ invstd, float* __restrict__ scale, float* __restrict__ shift) {
  __shared__ double sm[2][16];
  const int c = blockIdx.x, set = c / E, cl = c - set * E, cot = cl >> 6, ch = set * 64 + (cl & 63);
  double a = 0.0, b = 0.0;
  for (int t = threadIdx.x; t < tiles; t += 256) {
    const float* p = stats + ((size_t)t * co_tiles + cot) * 256 + ch;
    a += (double)p[0];
    b += (double)p[128];
  }
  a = block_sum_d(a, sm[0]);
  b = block_sum_d(b, sm[1]);
  if (threadIdx.x != 0) return;
  const double m = a / count;
  double var = b / count - m * m;
  if (var < 0.0) var = 0.0;
  const float is = (float)(1.0 / sqrt(var + (double)eps));
  const float* gamma = set == 0 ? gamma1 : gamma3;
  const float* beta = set == 0 ? beta1 : beta3;
  float* rm = set == 0 ? running_mean1 : running_mean3;
  float* rv = set == 0 ? running_var1 : running_var3;
  mean[c] = (float)m;
  invstd[c] = is;
  scale[c] = (gamma ? gamma[cl] : 1.f) * is;
  if (shift) shift[c] = beta ? beta[cl] : 0.f;
  if (rm) rm[cl] = (1.f - momentum) * rm[cl] + momentum * (float)m;
  if (rv) {
    const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
    rv[cl] = (1.f - momentum) * rv[cl] + momentum * (float)unb;
  }
}

template <int TWN, bool STATS, bool H2>
int launch_fire_fwd_t(const __bf16* planes, const __bf16* w3t, const __bf16* w1t, const float* bias3, const float* bias1,
                      float* y, int N, int KC, int H, int W, int E, int y_ctot, int y_coff, hipStream_t s, float* stats) {
  constexpr int TH = 4, TW = 32 * TWN, NPL = H2 ? 2 : 3;
  const int tiles_w = cdiv(W, TW), tiles_h = cdiv(H, TH), co_tiles = cdiv(E, 64);
  const int64_t blocks = (int64_t)N * tiles_h * tiles_w * co_tiles;
  if (blocks <= 0 || blocks > 0x7fffffff) return DLIO_EINVAL;
  constexpr int PINS = (NPL * (TH + 2) * (TW + 2) * 2 + 63) / 64;
  constexpr size_t lds_k = (size_t)PINS * 1024 + (size_t)2 * 2 * NPL * 64 * 16 * sizeof(__bf16);
  constexpr size_t lds = (STATS && lds_k < 4096) ? 4096 : lds_k;      // (the statistics epilogue uses 4 KB of it)
  // two-piece format: the inverse scales sit behind the planes / the weight layouts
  const float* inv_a = H2 ? reinterpret_cast<const float*>(planes + (size_t)N * KC * 2 * ((size_t)(H + 2) * (W + 2) * 16)) : nullptr;
  const float* inv_w3 = H2 ? reinterpret_cast<const float*>(w3t) + (size_t)9 * KC * E * 16 : nullptr;
  const float* inv_w1 = H2 ? reinterpret_cast<const float*>(w1t) + (size_t)KC * E * 16 : nullptr;
  dlio_set_max_lds(reinterpret_cast<const void*>(&fire_expand_fwd_kernel<TWN, STATS, H2>), (int)lds);
  hipLaunchKernelGGL((fire_expand_fwd_kernel<TWN, STATS, H2>), dim3((unsigned)blocks), dim3(256), lds, s, planes, w3t, w1t, bias3,
                     bias1, y, N, KC, H, W, E, y_ctot, y_coff, tiles_w, tiles_h, co_tiles, stats, inv_a, inv_w3, inv_w1);
  return dlio_check_launch();
}

template <int TWN>
int launch_fire_fwd(const __bf16* planes, const __bf16* w3t, const __bf16* w1t, const float* bias3, const float* bias1,
                    float* y, int N, int KC, int H, int W, int E, int y_ctot, int y_coff, hipStream_t s, float* stats = nullptr,
                    int fmt = 0) {
#define FIRE_GO(ST, HH) launch_fire_fwd_t<TWN, ST, HH>(planes, w3t, w1t, bias3, bias1, y, N, KC, H, W, E, y_ctot, y_coff, s, stats)
  if (fmt) return stats ? FIRE_GO(true, true) : FIRE_GO(false, true);
  return stats ? FIRE_GO(true, false) : FIRE_GO(false, false);
#undef FIRE_GO
}

// ---- weights as two fp16 pieces of w * 2^k: layout [tap][chunk][2][n][16] (the split-bf16 layout with two planes), behind
// it { 2^-k, 2^k } as floats.  2^k maps the tensor's largest magnitude into [2^13, 2^14].
// (1024 threads x 16-byte loads: with 256 threads and scalar loads the largest PointSeg tensor -- 147 k elements, 576 dependent
//  trips -- took 163 us at the head of EVERY step, in front of both encoders' first kernel.  H2_AMAX_WGS workgroups per weight
//  tensor (grid.y): one workgroup reads at ~50 GB/s, FlowNet's conv6 -- 4.7 M elements -- took 373 us.  The partial maxima meet
//  in tail[2] (integer atomicMax on the bits of a non-negative float: order-independent), tail[3] counts arrivals, the last
//  workgroup writes the scales and leaves both words zero for the next launch; the host zeroes them once, at allocation.)
constexpr int H2_AMAX_WGS = 32;
// (round 6: a tensor takes only as many of its grid.y workgroups as it has 16 k-element slices -- the rest leave at once --
//  and the arrival counter is a relaxed agent-scope atomic issued when the maximum's atomic has RETURNED: both words live
//  memory-side, nothing else is exchanged, so no fence.  With all 32 x n_items fat workgroups taking part, each with an
//  acq_rel fence = an L2 write-back + invalidate, the batched launch took 160 us at the head of every step for 6 MB of weights.)
__device__ __forceinline__ int h2_amax_wgs(int64_t n) {
  const int64_t w = (n + 16383) >> 14;
  return (int)(w < 1 ? 1 : (w > H2_AMAX_WGS ? H2_AMAX_WGS : w));
}
__global__ __launch_bounds__(1024) void prep_h2_amax_kernel(const DlioPrepItem* __restrict__ items, DlioPrepItem single) {
  __shared__ float wm[16];
  const DlioPrepItem it = items ? items[blockIdx.x] : single;
  const int64_t n = (int64_t)it.Cout * it.Cin * it.taps;
  const int wgs = h2_amax_wgs(n);
  if ((int)blockIdx.y >= wgs) return;
  const int64_t t0 = (int64_t)blockIdx.y * 1024 + threadIdx.x, stride = (int64_t)wgs * 1024;
  float b = 0.f;
  if ((reinterpret_cast<uintptr_t>(it.w) & 15) == 0) {
    const float4* w4 = reinterpret_cast<const float4*>(it.w);
    const int64_t n4 = n >> 2;
    for (int64_t i = t0; i < n4; i += stride) {
      const float4 v = w4[i];
      b = fmaxf(b, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
    }
    for (int64_t i = (n4 << 2) + t0; i < n; i += stride) b = fmaxf(b, fabsf(it.w[i]));
  } else {
    for (int64_t i = t0; i < n; i += stride) b = fmaxf(b, fabsf(it.w[i]));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) b = fmaxf(b, __shfl_xor(b, o, 64));
  if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = b;
  __syncthreads();
  if (threadIdx.x == 0) {
    b = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) b = fmaxf(b, wm[i]);
    const int K = it.mode == 0 ? it.Cin : it.Cout, Nn = it.mode == 0 ? it.Cout : it.Cin;
    float* tail = it.wt + (size_t)it.taps * ((K + 15) >> 4) * Nn * 16;
    unsigned* tu = reinterpret_cast<unsigned*>(tail);
    if (!(b < 3.0e38f)) b = 3.4e38f;                     // NaN / Inf anywhere: the scale falls back to 1
    unsigned top = __float_as_uint(b);
    if (wgs > 1) {
      const unsigned old = __hip_atomic_fetch_max(tu + 2, top, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::"v"(old) : "memory");     // the maximum is in place before the arrival is counted
      const unsigned arrived = __hip_atomic_fetch_add(tu + 3, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (arrived != (unsigned)wgs - 1) return;
      top = __hip_atomic_load(tu + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(tu + 2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(tu + 3, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    b = __uint_as_float(top);
    const float sc = (b > 0.f && b < 3.0e38f) ? exp2f(floorf(log2f(16384.f / b))) : 1.f;
    tail[0] = 1.f / sc;
    tail[1] = sc;
  }
}

// (a thread builds 8 consecutive k of one (tap, chunk, n): one search of the item table and two 16-byte stores per 8 elements)
typedef _Float16 prep_f16x8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(256) void prep_h2_kernel(const DlioPrepItem* __restrict__ items, int n_items, int64_t total,
                                                      DlioPrepItem single) {
  const int64_t total8 = total >> 3;                // (every item's index space is a multiple of 16)
  for (int64_t i8 = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i8 < total8; i8 += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = i8 << 3;
    int lo = 0, hi = n_items - 1;                 // last item with start <= i
    while (items && lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (items[mid].start <= i) lo = mid; else hi = mid - 1;
    }
    const DlioPrepItem it = items ? items[lo] : single;
    const int64_t e = i - it.start;
    const int K = it.mode == 0 ? it.Cin : it.Cout, Nn = it.mode == 0 ? it.Cout : it.Cin;
    const int KC = (K + 15) >> 4;
    const int kk = (int)(e & 15);                  // 0 or 8
    int64_t t = e >> 4;
    const int n = (int)(t % Nn); t /= Nn;
    const int kc = (int)(t % KC);
    const int tap = (int)(t / KC);
    const int k0 = kc * 16 + kk;
    const float* src = it.mode == 0 ? it.w + ((int64_t)n * it.Cin + k0) * it.taps + tap
                                    : it.w + ((int64_t)k0 * it.Cin + n) * it.taps + (it.taps - 1 - tap);
    const int64_t ks = it.mode == 0 ? (int64_t)it.taps : (int64_t)it.Cin * it.taps;
    const float sc = it.wt[(size_t)it.taps * KC * Nn * 16 + 1];
    prep_f16x8 h, l;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const float vs = (k0 + q < K ? src[q * ks] : 0.f) * sc;
      const _Float16 hh = (_Float16)vs;
      h[q] = hh;
      l[q] = (_Float16)(vs - (float)hh);
    }
    _Float16* wt = reinterpret_cast<_Float16*>(it.wt);
    const int64_t base = (((int64_t)tap * KC + kc) * 2) * Nn * 16 + (int64_t)n * 16 + kk;
    *reinterpret_cast<prep_f16x8*>(wt + base) = h;
    *reinterpret_cast<prep_f16x8*>(wt + base + (int64_t)Nn * 16) = l;
  }
}

}  // namespace

extern "C" size_t dlio_fire_planes_bytes(int N, int S, int H, int W) {
  if (N <= 0 || S <= 0 || H <= 0 || W <= 0) return 0;
  return (size_t)N * ((S + 15) / 16) * 3 * (H + 2) * (W + 2) * 16 * sizeof(__bf16);
}

float* dlio_internal_stats_partials_mm(const float* x, int N, int x_ctot, int x_coff, int C, int HW, void* ws, size_t ws_bytes,
                                       hipStream_t s, int* rc);      // bn.hip

extern "C" int dlio_bn_split16(const float* x, int N, int x_ctot, int x_coff, int C, int H, int W, int post_relu,
                               const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                               float* running_var, float* mean, float* invstd, float* scale, float* y, int y_ctot,
                               int y_coff, void* planes, void* ws, size_t ws_bytes, int mode, double count_scale,
                               float* bound_out, dlio_stream_t stream) {
  // mode 0: train, statistics + apply; 1: train, statistics partials only (-> ws); 2: train, apply from the partials in
  // ws; 3: eval (mean / scale are inputs, ws unused)
  // + 16: the planes as TWO fp16 pieces of x 2^k with 2^-k behind them (train modes only: the bound on |BN(x)| that gives
  // 2^k needs batch statistics)
  const int h2 = (mode & 16) ? 1 : 0;
  mode &= ~16;
  if (!x || !planes || !mean || !scale || N <= 0 || C <= 0 || H <= 0 || W <= 0 || mode < 0 || mode > 3 ||
      !(count_scale >= 1.0) || (h2 && (mode == 3 || mode == 1)))
    return DLIO_EINVAL;
  if ((W & 3) || ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(planes)) & 15))
    return DLIO_EUNSUP;
  hipStream_t s = as_stream(stream);
  const int HW = H * W;
  const double tensor_bytes = 4.0 * N * (double)C * HW;
  int splits = 0;
  const float* mm = nullptr;                 // per-split channel ranges behind the partials (two-piece planes: the exact scale)
  if (mode != 3) {
    if (!invstd || !ws) return DLIO_EINVAL;
    splits = dlio_chan_stats_splits(N, C, HW);
    if (ws_bytes < (size_t)C * splits * 2 * sizeof(double)) return DLIO_EWS;
    static const bool exact = !(getenv("DLIO_SPLIT16_EXACT") && atoi(getenv("DLIO_SPLIT16_EXACT")) == 0);
    if (mode == 0 && h2 && count_scale == 1.0 && exact) {
      int rc = DLIO_OK;
      mm = dlio_internal_stats_partials_mm(x, N, x_ctot, x_coff, C, HW, ws, ws_bytes, s, &rc);
      if (rc) return rc;
    } else if (mode != 2) {
      // statistics partials by the BatchNorm path's own reduction (dlio_bn_train_apply phase 1)
      const int rc = dlio_bn_train_apply(x, N, x_ctot, x_coff, C, HW, 0, post_relu, gamma, beta, eps, momentum, running_mean,
                                         running_var, mean, invstd, scale, nullptr, 0, 0, y ? y : const_cast<float*>(x), y_ctot,
                                         y_coff, nullptr, 0, 0, ws, ws_bytes, 1, count_scale, nullptr, nullptr, nullptr, nullptr, stream);
      if (rc || mode == 1) return rc;
    }
  }
  DlioProfScope prof(7, s, 0.0, tensor_bytes * (y ? 3.5 : 2.5));
  const dim3 grid((unsigned)cdiv(HW, 4 * 32 * SPLIT_U), (unsigned)((C + 15) / 16), (unsigned)N);
  hipLaunchKernelGGL(bn_split16_kernel, grid, dim3(256), 0, s, x, x_ctot, x_coff,
                     mode == 3 ? (const double*)nullptr : reinterpret_cast<const double*>(ws), splits,
                     (double)N * HW * count_scale, gamma, beta, eps, momentum, running_mean, running_var, mean, invstd, scale,
                     y, y_ctot, y_coff, reinterpret_cast<__bf16*>(planes), N, C, H, W, post_relu, h2, h2 ? bound_out : nullptr, mm);
  return dlio_check_launch();
}

extern "C" size_t dlio_conv_h2_prep_floats(int Cout, int Cin, int taps, int mode) {
  if (Cout <= 0 || Cin <= 0 || taps <= 0 || (mode != 0 && mode != 1)) return 0;
  const int K = mode == 0 ? Cin : Cout, Nn = mode == 0 ? Cout : Cin;
  return (size_t)taps * ((K + 15) >> 4) * Nn * 16 + 4;      // two fp16 planes = one float per element, + the scales
}

extern "C" int dlio_conv_h2_prep(const float* w, void* wt, int Cout, int Cin, int taps, int mode, dlio_stream_t stream) {
  if (!w || !wt || Cout <= 0 || Cin <= 0 || taps <= 0 || (mode != 0 && mode != 1)) return DLIO_EINVAL;
  if (reinterpret_cast<uintptr_t>(wt) & 15) return DLIO_EUNSUP;          // (the layout is written in 16-byte pieces)
  const int K = mode == 0 ? Cin : Cout, Nn = mode == 0 ? Cout : Cin;
  const int64_t total = (int64_t)taps * ((K + 15) >> 4) * Nn * 16;
  const DlioPrepItem one{w, reinterpret_cast<float*>(wt), Cout, Cin, taps, mode, 0};
  hipStream_t s = as_stream(stream);
  // (the two scratch words behind the scales must be zero: zeroed here for a layout buffer that comes from anywhere)
  if (hipMemsetAsync(reinterpret_cast<float*>(wt) + total + 2, 0, 2 * sizeof(float), s) != hipSuccess) return DLIO_ELAUNCH;
  hipLaunchKernelGGL(prep_h2_amax_kernel, dim3(1, H2_AMAX_WGS), dim3(1024), 0, s, (const DlioPrepItem*)nullptr, one);
  hipLaunchKernelGGL(prep_h2_kernel, dim3(ew_grid(total >> 3, 256)), dim3(256), 0, s, (const DlioPrepItem*)nullptr, 1, total, one);
  return dlio_check_launch();
}

extern "C" int dlio_conv_h2_prep_batched(const DlioPrepItem* items_dev, int n_items, int64_t total, dlio_stream_t stream) {
  if (!items_dev || n_items <= 0 || total <= 0 || (total & 15)) return DLIO_EINVAL;
  hipStream_t s = as_stream(stream);
  const DlioPrepItem none{nullptr, nullptr, 0, 0, 0, 0, 0};
  hipLaunchKernelGGL(prep_h2_amax_kernel, dim3((unsigned)n_items, H2_AMAX_WGS), dim3(1024), 0, s, items_dev, none);
  hipLaunchKernelGGL(prep_h2_kernel, dim3(ew_grid(total >> 3, 256)), dim3(256), 0, s, items_dev, n_items, total, none);
  return dlio_check_launch();
}

extern "C" int dlio_fire_expand_fwd(const void* planes, const void* w3t, const void* w1t, const float* bias3,
                                    const float* bias1, float* y, int N, int S, int H, int W, int E, int y_ctot,
                                    int y_coff, int planes_fmt, dlio_stream_t stream) {
  if (!planes || !w3t || !w1t || !y || N <= 0 || S <= 0 || H <= 0 || W <= 0 || E <= 0 || y_ctot < y_coff + 2 * E || y_coff < 0 ||
      planes_fmt < 0 || planes_fmt > 1)
    return DLIO_EINVAL;
  const int KC = (S + 15) / 16;
  if ((W & 3) || (reinterpret_cast<uintptr_t>(y) & 15) || (reinterpret_cast<uintptr_t>(planes) & 15)) return DLIO_EUNSUP;
  if (dlio_fire_planes_bytes(N, S, H, W) >= 0x7fffffffull || (size_t)9 * KC * 3 * E * 32 >= 0x7fffffffull) return DLIO_EUNSUP;
  hipStream_t s = as_stream(stream);
  const double P = (double)N * H * W;
  DlioProfScope prof(3, s, 2.0 * P * E * (double)S * 10, 4.0 * P * 2 * E + 6.0 * P * KC * 16);
  const __bf16* pl = reinterpret_cast<const __bf16*>(planes);
  const __bf16* w3 = reinterpret_cast<const __bf16*>(w3t);
  const __bf16* w1 = reinterpret_cast<const __bf16*>(w1t);
  return W > 32 ? launch_fire_fwd<2>(pl, w3, w1, bias3, bias1, y, N, KC, H, W, E, y_ctot, y_coff, s, nullptr, planes_fmt)
                : launch_fire_fwd<1>(pl, w3, w1, bias3, bias1, y, N, KC, H, W, E, y_ctot, y_coff, s, nullptr, planes_fmt);
}

static void fire_tiles(int N, int H, int W, int E, int& tiles, int& co_tiles) {
  const int TW = W > 32 ? 64 : 32;
  tiles = N * cdiv(H, 4) * cdiv(W, TW);
  co_tiles = cdiv(E, 64);
}

extern "C" size_t dlio_fire_expand_stats_ws_bytes(int N, int H, int W, int E) {
  if (N <= 0 || H <= 0 || W <= 0 || E <= 0) return 0;
  int tiles, co_tiles;
  fire_tiles(N, H, W, E, tiles, co_tiles);
  return (size_t)tiles * co_tiles * 256 * sizeof(float);
}

extern "C" int dlio_fire_expand_fwd_stats(const void* planes, const void* w3t, const void* w1t, const float* bias3,
                                          const float* bias1, float* y, int N, int S, int H, int W, int E, int y_ctot,
                                          int y_coff, const float* gamma1, const float* beta1, float* running_mean1,
                                          float* running_var1, const float* gamma3, const float* beta3, float* running_mean3,
                                          float* running_var3, float eps, float momentum, float* mean, float* invstd,
                                          float* scale, float* shift, void* ws, size_t ws_bytes, int planes_fmt,
                                          dlio_stream_t stream) {
  if (!planes || !w3t || !w1t || !y || !mean || !invstd || !scale || !ws || N <= 0 || S <= 0 || H <= 0 || W <= 0 || E <= 0 ||
      y_ctot < y_coff + 2 * E || y_coff < 0 || planes_fmt < 0 || planes_fmt > 1)
    return DLIO_EINVAL;
  const int KC = (S + 15) / 16;
  if ((W & 3) || (reinterpret_cast<uintptr_t>(y) & 15) || (reinterpret_cast<uintptr_t>(planes) & 15)) return DLIO_EUNSUP;
  if (dlio_fire_planes_bytes(N, S, H, W) >= 0x7fffffffull || (size_t)9 * KC * 3 * E * 32 >= 0x7fffffffull) return DLIO_EUNSUP;
  if (ws_bytes < dlio_fire_expand_stats_ws_bytes(N, H, W, E)) return DLIO_EWS;
  hipStream_t s = as_stream(stream);
  const double P = (double)N * H * W;
  const __bf16* pl = reinterpret_cast<const __bf16*>(planes);
  const __bf16* w3 = reinterpret_cast<const __bf16*>(w3t);
  const __bf16* w1 = reinterpret_cast<const __bf16*>(w1t);
  float* st = reinterpret_cast<float*>(ws);
  int rc;
  {
    DlioProfScope prof(3, s, 2.0 * P * E * (double)S * 10, 4.0 * P * 2 * E + 6.0 * P * KC * 16);
    rc = W > 32 ? launch_fire_fwd<2>(pl, w3, w1, bias3, bias1, y, N, KC, H, W, E, y_ctot, y_coff, s, st, planes_fmt)
                : launch_fire_fwd<1>(pl, w3, w1, bias3, bias1, y, N, KC, H, W, E, y_ctot, y_coff, s, st, planes_fmt);
  }
  if (rc) return rc;
  int tiles, co_tiles;
  fire_tiles(N, H, W, E, tiles, co_tiles);
  DlioProfScope prof(6, s, 0.0, (double)tiles * co_tiles * 256 * sizeof(float));       // (a BatchNorm statistics launch)
  hipLaunchKernelGGL(fire_stats_finalize_kernel, dim3((unsigned)(2 * E)), dim3(256), 0, s, st, tiles, co_tiles, E, P, eps, momentum,
                     gamma1, beta1, running_mean1, running_var1, gamma3, beta3, running_mean3, running_var3, mean, invstd,
                     scale, shift);
  return dlio_check_launch();
}

// timing probes compiled into this file (bit 0: DLIO_SPLIT_Q0); 0 in the product build, checked at load (dlio_build_probes)
int dlio_probe_fire() { return ((DLIO_SPLIT_Q0) != 0 ? 1 : 0); }
