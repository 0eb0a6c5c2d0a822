// Weight gradient of the 2-D convolution on v_mfma_f32_32x32x2_f32.
//
// GEMM: dW[co][j] = sum_pixels dY[co][pixel] * X[pixel][j],   j = (ci, tap) flattened with
// tap fastest, i.e. j is exactly the offset inside a row of the standard
// [Cout][Cin][KH][KW] weight layout.
//   A operand: lane l holds dY[co = l&31][pixel + (l>>5)]   (LDS, odd per-channel stride)
//   B operand: lane l holds X'[pixel + (l>>5)][j = l&31]    (LDS patch + per-lane tap offset)
// K (= pixels of the whole batch) is split over blocks; the 4 waves of a block take one
// image row each, their accumulators are summed through LDS and written as one partial
// slab; a second kernel sums the slabs in a fixed order (deterministic, no atomics).
//
// Replaces the weight-gradient half of nn.Conv2d backward for every conv on the path
// (same reference sites as conv_fwd.hip).
#include "common.h"
#include "wgrad3.h"
#include <stdlib.h>

namespace {

template <int KH, int KW, int SH, int SW, int NT, int MRW>
struct WgCfg {
  static constexpr int TH = 4, TW = 32;
  static constexpr int MR = MRW, CO_T = 32 * MRW;
  static constexpr int TAPS = KH * KW;
  // channels per chunk; halved for the 1x1 stride-(2,2) patch so two pipeline stages fit in LDS
  static constexpr int CKMAX = ((NT * 32) / TAPS) / ((TAPS == 1 && SH * SW == 4) ? 2 : 1);
  static constexpr int PR = (TH - 1) * SH + KH;
  static constexpr int PC = (TW - 1) * SW + KW;
  static constexpr int PRPC = PR * PC;
  static constexpr int PLANE = PRPC | 1;          // odd -> lanes (=channels) hit distinct banks
  static constexpr int DYS = TH * TW + 1;         // odd per-channel stride of the dY tile
  static constexpr int XL = CKMAX * PLANE;
  static constexpr int DL = CO_T * DYS;
  static constexpr int BUF = XL + DL;             // one pipeline stage
  static constexpr int RED = MR * NT * 16 * 64;
  static constexpr int SM_FLOATS = 2 * BUF > RED ? 2 * BUF : RED;
  static constexpr size_t LDS_BYTES = (size_t)SM_FLOATS * 4;
  // X staging: LPP threads cover one channel plane, CPAR channels in parallel
  static constexpr int LPP = PRPC <= 128 ? 128 : 256;
  static constexpr int CPAR = 256 / LPP;
  static constexpr int NPOSX = (PRPC + LPP - 1) / LPP;
  static constexpr int NCX = (CKMAX + CPAR - 1) / CPAR;
  static constexpr int NDY = CO_T / 2;            // dY elements per thread
};

typedef __bf16 wg_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 wg_bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int wg_u32x4 __attribute__((ext_vector_type(4)));

// x = hi + mid + lo in bf16 pieces (see conv_bx3.hip): eight values -> three MFMA fragments, pair by pair as in
// conv_wgrad3.hip (v_cvt_pk_bf16_f32 rounds two values at once; a bf16 widens to fp32 by a shift / a mask of the
// packed dword): 11 VALU per pair instead of 7 per value, the same roundings
__device__ __forceinline__ unsigned wg_cvt_pk(float a, float b) {
  return __builtin_bit_cast(unsigned, wg_bf16x2{(__bf16)a, (__bf16)b});
}
__device__ __forceinline__ float wg_as_f(unsigned u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ void wg_split8(const float (&v)[8], wg_bf16x8& h, wg_bf16x8& m, wg_bf16x8& l) {
  unsigned hh[4], mm[4], ll[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float a = v[2 * j], b = v[2 * j + 1];
    hh[j] = wg_cvt_pk(a, b);
    const float ra = a - wg_as_f(hh[j] << 16), rb = b - wg_as_f(hh[j] & 0xffff0000u);
    mm[j] = wg_cvt_pk(ra, rb);
    ll[j] = wg_cvt_pk(ra - wg_as_f(mm[j] << 16), rb - wg_as_f(mm[j] & 0xffff0000u));
  }
  h = __builtin_bit_cast(wg_bf16x8, wg_u32x4{hh[0], hh[1], hh[2], hh[3]});
  m = __builtin_bit_cast(wg_bf16x8, wg_u32x4{mm[0], mm[1], mm[2], mm[3]});
  l = __builtin_bit_cast(wg_bf16x8, wg_u32x4{ll[0], ll[1], ll[2], ll[3]});
}

// Software pipeline over the block's pixel tiles: global loads of tile t+1 (dY tile and X'
// patch) are issued into registers, the MFMAs of tile t run out of LDS stage t&1, the registers
// are written to stage (t+1)&1, one barrier per tile.  The kernel runs one workgroup per CU
// (160 accumulator registers), so this in-block overlap is what hides the HBM latency.
// BX3 (round 6; the PointSeg stem, 3x5 stride (1, 2), 10 input channels: 2 x 258 us at the end of the step, where nothing
// overlaps it): the products on v_mfma_f32_32x32x16_bf16 over the exact three-piece bf16 split of both operands (six MFMAs per
// fp32 product, conv_bx3.hip) instead of eight v_mfma_f32_32x32x2_f32 per 16 pixels: a lane reads its 8 consecutive pixels
// of a dY row / of an X' column out of the same LDS tiles, splits them (11 VALU per pair) and issues 6 MFMAs per (m, t)
// tile -- 120 MFMAs of 32 cycles per tile and wave instead of 160 of 64.
template <int KH, int KW, int SH, int SW, int NT, int MRW, bool VEC, bool BX3 = false>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(
    const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ wsp,
    const float* __restrict__ in_mean, const float* __restrict__ in_scale,
    const float* __restrict__ in_shift, DlioConvDesc d, int co_tiles, int ci_chunks, int splits,
    int tiles_w, int tiles_h) {
  using C = WgCfg<KH, KW, SH, SW, NT, MRW>;
  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int l31 = lane & 31;
  const int half = lane >> 5;

  // XCD-aware order (common.h): the (output-channel tile, input-channel chunk) pairs of one pixel split are
  // neighbours -- they read the same dY / X pixels -- and an XCD works through whole splits
  int bid = xcd_block_index();
  const int npairs = co_tiles * ci_chunks;
  const int split = bid / npairs; bid -= split * npairs;
  const int cic = bid % ci_chunks; bid /= ci_chunks;
  const int cot = bid;
  const int co0 = cot * C::CO_T;
  const int c0 = cic * C::CKMAX;
  const int ck = min(C::CKMAX, d.Cin - c0);   // live channels in this chunk
  const int nj = ck * C::TAPS;                // live columns

  // per-lane LDS offset of column j = t*32 + l31
  int off[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int j = t * 32 + l31;
    int o = 0;
    if (j < nj) {
      const int cl = j / C::TAPS;
      const int tap = j - cl * C::TAPS;
      const int ky = tap / KW, kx = tap - ky * KW;
      o = cl * C::PLANE + ky * C::PC + kx;
    }
    off[t] = o;
  }

  f32x16 acc[C::MR][NT];
#pragma unroll
  for (int m = 0; m < C::MR; ++m)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][t][r] = 0.f;

  const bool has_aff = in_scale != nullptr;
  const int total_tiles = d.N * tiles_h * tiles_w;
  const size_t ohw = (size_t)d.OH * d.OW;
  const size_t HW = (size_t)d.H * d.W;

  // thread-constant staging coordinates
  // VEC (OW % 4 == 0): float4 loads, element f = tid + 256*i -> col4 = f&7, row = (f>>3)&3, ch = f>>5
  constexpr bool XVEC = VEC && KH == 1 && KW == 1 && SH == 1 && SW == 1;
  const int v_c4 = tid & 7, v_r = (tid >> 3) & 3, v_ch = tid >> 5;        // ch = v_ch + 8*i
  const int dy_r = (tid >> 5) & 3, dy_col = tid & 31, dy_co = tid >> 7;   // co = dy_co + 2*i
  const int xp = tid % C::LPP, xcph = tid / C::LPP;                      // c = xcph + CPAR*i
  int xr[C::NPOSX], xc[C::NPOSX];
#pragma unroll
  for (int j = 0; j < C::NPOSX; ++j) {
    const int pos = xp + j * C::LPP;
    xr[j] = pos / C::PC;
    xc[j] = pos - xr[j] * C::PC;
  }

  // All global loads are unconditional buffer loads; an invalid element gets a voffset beyond
  // num_records and the hardware returns 0.  (Conditional loads compile to branch + load +
  // s_waitcnt vmcnt(0) per element, which serialises the prefetch.)
  constexpr unsigned OOB = 0xffffff00u;
  const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(x), 0, (int)((size_t)d.N * d.in_ctot * HW * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(dy), 0, (int)((size_t)d.N * d.out_ctot * ohw * 4), 0x00020000);
  auto ld1 = [&](__amdgpu_buffer_rsrc_t r, unsigned vo) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, vo, 0, 0));
  };
  auto ld4 = [&](__amdgpu_buffer_rsrc_t r, unsigned vo) {
    return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, vo, 0, 0));
  };
  float rdy[C::NDY];
  float rx[C::NCX][C::NPOSX];
  auto load_tile = [&](int tile) {
    int tt = tile;
    const int tw = tt % tiles_w; tt /= tiles_w;
    const int th = tt % tiles_h; tt /= tiles_h;
    const int n = tt;
    const int oh0 = th * C::TH, ow0 = tw * C::TW;
    const int ih0 = oh0 * SH - d.PH, iw0 = ow0 * SW - d.PW;
    const unsigned yimg = (unsigned)(((size_t)n * d.out_ctot + d.out_coff + co0) * ohw * 4);
    const unsigned ximg = (unsigned)(((size_t)n * d.in_ctot + d.in_coff + c0) * HW * 4);
    if constexpr (VEC) {
      const int oh = oh0 + v_r, ow = ow0 + 4 * v_c4;
      const bool pv = oh < d.OH && ow < d.OW;
      const unsigned po = yimg + (unsigned)(oh * d.OW + ow) * 4u;
#pragma unroll
      for (int i = 0; i < C::NDY / 4; ++i) {
        const int co = v_ch + 8 * i;
        const float4 t = ld4(yrsrc, (pv && co0 + co < d.Cout) ? po + (unsigned)co * (unsigned)ohw * 4u : OOB);
        rdy[4 * i] = t.x; rdy[4 * i + 1] = t.y; rdy[4 * i + 2] = t.z; rdy[4 * i + 3] = t.w;
      }
    } else {
      const int oh = oh0 + dy_r, ow = ow0 + dy_col;
      const bool pv = oh < d.OH && ow < d.OW;
      const unsigned po = yimg + (unsigned)(oh * d.OW + ow) * 4u;
#pragma unroll
      for (int i = 0; i < C::NDY; ++i) {
        const int co = dy_co + 2 * i;
        rdy[i] = ld1(yrsrc, (pv && co0 + co < d.Cout) ? po + (unsigned)co * (unsigned)ohw * 4u : OOB);
      }
    }
    if constexpr (XVEC) {
      const int ih = ih0 + v_r, iw = iw0 + 4 * v_c4;
      const bool pvx = ih < d.H && iw < d.W;
      const unsigned po = ximg + (unsigned)(ih * d.W + iw) * 4u;
#pragma unroll
      for (int i = 0; i < C::CKMAX / 8; ++i) {
        const int c = v_ch + 8 * i;
        const bool ok = pvx && c < ck;
        float4 t = ld4(xrsrc, ok ? po + (unsigned)c * (unsigned)HW * 4u : OOB);
        if (has_aff) {
          const int cc = min(c0 + c, d.Cin - 1);
          const float mu = in_mean[cc], sc = in_scale[cc], sh = in_shift[cc];
          t.x = (t.x - mu) * sc + sh; t.y = (t.y - mu) * sc + sh;
          t.z = (t.z - mu) * sc + sh; t.w = (t.w - mu) * sc + sh;
          if (d.in_relu) { t.x = fmaxf(t.x, 0.f); t.y = fmaxf(t.y, 0.f); t.z = fmaxf(t.z, 0.f); t.w = fmaxf(t.w, 0.f); }
          if (!ok) t = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        rx[4 * i][0] = t.x; rx[4 * i + 1][0] = t.y; rx[4 * i + 2][0] = t.z; rx[4 * i + 3][0] = t.w;
      }
      return;
    }
    unsigned po[C::NPOSX];
#pragma unroll
    for (int j = 0; j < C::NPOSX; ++j) {
      const int ih = ih0 + xr[j], iw = iw0 + xc[j];
      const bool pv = (xp + j * C::LPP) < C::PRPC && ih >= 0 && ih < d.H && iw >= 0 && iw < d.W;
      po[j] = pv ? ximg + (unsigned)(ih * d.W + iw) * 4u : OOB;
    }
#pragma unroll
    for (int i = 0; i < C::NCX; ++i) {
      const int c = xcph + C::CPAR * i;
      const bool cv = c < ck;
      const unsigned coff = (unsigned)c * (unsigned)HW * 4u;
#pragma unroll
      for (int j = 0; j < C::NPOSX; ++j) {
        const bool ok = cv && po[j] != OOB;
        float v = ld1(xrsrc, ok ? po[j] + coff : OOB);
        if (has_aff) {
          const int cc = min(c0 + c, d.Cin - 1);
          v = (v - in_mean[cc]) * in_scale[cc] + in_shift[cc];
          if (d.in_relu) v = fmaxf(v, 0.f);
          if (!ok) v = 0.f;
        }
        rx[i][j] = v;
      }
    }
  };
  auto store_tile = [&](float* buf) {
    float* Xl = buf;
    float* Dl = buf + C::XL;
    if constexpr (VEC) {
#pragma unroll
      for (int i = 0; i < C::NDY / 4; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          Dl[(v_ch + 8 * i) * C::DYS + v_r * 32 + v_c4 * 4 + e] = rdy[4 * i + e];
    } else {
#pragma unroll
      for (int i = 0; i < C::NDY; ++i) Dl[(dy_co + 2 * i) * C::DYS + dy_r * 32 + dy_col] = rdy[i];
    }
    if constexpr (XVEC) {
#pragma unroll
      for (int i = 0; i < C::CKMAX / 8; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          Xl[(v_ch + 8 * i) * C::PLANE + v_r * 32 + v_c4 * 4 + e] = rx[4 * i + e][0];
      return;
    }
#pragma unroll
    for (int i = 0; i < C::NCX; ++i) {
      const int c = xcph + C::CPAR * i;
      if (c < C::CKMAX) {
#pragma unroll
        for (int j = 0; j < C::NPOSX; ++j) {
          const int pos = xp + j * C::LPP;
          if (pos < C::PRPC) Xl[c * C::PLANE + pos] = rx[i][j];
        }
      }
    }
  };
  // 8 steps of 2 k-pairs; the LDS operands of step s+1 are read before the MFMAs of step s
  // (register double buffer + sched_group_barrier), so LDS latency overlaps the MFMAs
  auto mfma_tile = [&](const float* buf) {
    if constexpr (BX3) {
      // k block ks = the wave's pixels 16 ks .. 16 ks + 15 of its tile row; lane (row l31, half): pixels 8 half .. + 7 of it
      const float* drow8 = buf + C::XL + l31 * C::DYS + wave * 32 + half * 8;
      const float* xrow8 = buf + (wave * SH) * C::PC + half * 8 * SW;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        wg_bf16x8 ah[C::MR], am[C::MR], al[C::MR];
#pragma unroll
        for (int m = 0; m < C::MR; ++m) {
          float v[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) v[i] = drow8[m * 32 * C::DYS + ks * 16 + i];
          wg_split8(v, ah[m], am[m], al[m]);
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          float v[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) v[i] = xrow8[off[t] + (ks * 16 + i) * SW];
          wg_bf16x8 bh, bm, bl;
          wg_split8(v, bh, bm, bl);
          // smallest products first (conv_bx3.hip): lo hi, mid mid, hi lo, mid hi, hi mid, hi hi
#pragma unroll
          for (int term = 0; term < 6; ++term)
#pragma unroll
            for (int m = 0; m < C::MR; ++m) {
              const wg_bf16x8& av = term == 0 ? al[m] : (term == 1 || term == 3) ? am[m] : ah[m];
              const wg_bf16x8& bw = (term == 0 || term == 3 || term == 5) ? bh : (term == 1 || term == 4) ? bm : bl;
              acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bw, acc[m][t], 0, 0, 0);
            }
        }
      }
      return;
    }
    const float* drow = buf + C::XL + l31 * C::DYS + wave * 32 + half;
    const float* xrow = buf + (wave * SH) * C::PC + half * SW;
    float a[2][C::MR][2], b[2][NT][2];
    auto load_ab = [&](int st, float (&av)[C::MR][2], float (&bv)[NT][2]) {
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int kp = st * 2 + e;
#pragma unroll
        for (int m = 0; m < C::MR; ++m) av[m][e] = drow[m * 32 * C::DYS + kp * 2];
#pragma unroll
        for (int t = 0; t < NT; ++t) bv[t][e] = xrow[off[t] + kp * 2 * SW];
      }
    };
    load_ab(0, a[0], b[0]);
#pragma unroll
    for (int st = 0; st < 8; ++st) {
      if (st + 1 < 8) load_ab(st + 1, a[(st + 1) & 1], b[(st + 1) & 1]);
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int m = 0; m < C::MR; ++m)
#pragma unroll
          for (int t = 0; t < NT; ++t)
            acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[st & 1][m][e], b[st & 1][t][e], acc[m][t], 0, 0, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, C::MR + NT, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 2 * C::MR * NT, 0);
    }
  };

  if (split < total_tiles) {
    load_tile(split);
    store_tile(smem);
  }
  __syncthreads();
  int it = 0;
  for (int tile = split; tile < total_tiles; tile += splits, ++it) {
    float* cur = smem + (it & 1) * C::BUF;
    float* nxt = smem + ((it + 1) & 1) * C::BUF;
    const bool more = tile + splits < total_tiles;
    if (more) load_tile(tile + splits);
    mfma_tile(cur);
    if (more) store_tile(nxt);
    __syncthreads();
  }

  // ---- cross-wave reduction through LDS (fixed order) -------------------------
  float* red = smem;
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int m = 0; m < C::MR; ++m)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int i = ((m * NT + t) * 16 + r) * 64 + lane;
            if (w == 0) red[i] = acc[m][t][r];
            else red[i] += acc[m][t][r];
          }
    }
    __syncthreads();
  }

  // ---- write the partial slab in standard weight layout ------------------------
  const size_t row_len = (size_t)d.Cin * C::TAPS;
  float* out = wsp + (size_t)split * d.Cout * row_len;
  for (int idx = tid; idx < C::CO_T * NT * 32; idx += 256) {
    const int j = idx % (NT * 32);
    const int col = idx / (NT * 32);     // local output channel
    if (j >= nj || co0 + col >= d.Cout) continue;
    const int m = col >> 5, row = col & 31;
    const int hf = (row >> 2) & 1;
    const int r = (row & 3) + 4 * (row >> 3);
    const int t = j >> 5, lj = j & 31;
    out[(size_t)(co0 + col) * row_len + (size_t)c0 * C::TAPS + j] =
        red[((m * NT + t) * 16 + r) * 64 + hf * 32 + lj];
  }
}

__device__ __forceinline__ float f4e(const float4& v, int i) {
  return i == 0 ? v.x : (i == 1 ? v.y : (i == 2 ? v.z : v.w));
}

// ---- 1x1 stride-1 weight gradient straight from global memory -------------------------------
// dW[co][ci] = sum_px dY[co][px] * X[ci][px]: both operands are [channel][pixel] rows, so a lane
// (channel l&31, half l>>5) reads 16 consecutive pixels of its row as 4 float4 (a wave covers 128
// contiguous bytes of 32 rows) and feeds them to 16 MFMA k-steps -- which pixel sits in which
// k slot is irrelevant as long as A and B agree.  No LDS staging and no barriers in the loop: each
// wave strides over 32-pixel segments on its own with a register double buffer, two workgroups
// per CU keep ~128 KB of loads in flight (the LDS-staged kernel above manages ~40 KB for these
// layers, which are HBM-bound).  Partial sums: 4 waves through LDS, then the slab reduction.
// NAT: bf16 operands in memory (mixed-precision path): a lane's 16 pixels are two 16-byte loads that
// ARE the two k-blocks' MFMA fragments -- no conversion, one MFMA per product.
// AFF: the X operand is the BatchNorm+ReLU of what is stored, x' = max(0, (x - mean[ci]) * scale[ci] + shift[ci])
// (apply-on-load: the producer's activated output is never materialised); a lane owns one channel, so
// the three constants are per-lane registers and the transform is 2 VALU per loaded value.
#ifndef W1_COAL_PROBE
#define W1_COAL_PROBE 0
#endif
template <int MR, int NT, bool BX3 = false, bool NAT = false, bool AFF = false>
__global__ __launch_bounds__(256, (BX3 && AFF) ? 1 : 2) void wgrad1x1_direct_kernel(
    const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ wsp,
    DlioConvDesc d, int co_tiles, int ci_chunks, int splits, int segs_per_img,
    const float* __restrict__ in_mean = nullptr, const float* __restrict__ in_scale = nullptr,
    const float* __restrict__ in_shift = nullptr) {
  constexpr int ED = NAT ? 2 : 1;                 // bf16 elements per float slot of the pointer arithmetic
  __shared__ float red[MR * NT * 16 * 64];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int l31 = lane & 31;
  const int half = lane >> 5;

  int bid = xcd_block_index();           // XCD-aware: the pairs of one pixel split are neighbours (see conv_wgrad_kernel)
  const int npairs = co_tiles * ci_chunks;
  const int split = bid / npairs; bid -= split * npairs;
  const int cic = bid % ci_chunks; bid /= ci_chunks;
  const int co0 = bid * 32 * MR;
  const int c0 = cic * 32 * NT;
  const size_t hw = (size_t)d.H * d.W;

  // buffer descriptors over the two tensors + a 32-bit lane offset per row (the 64-bit row pointers of the 64 x 64 tile
  // spilled); the launcher has checked that both tensors are < 4 GB
  constexpr unsigned ES = 4 / ED;                 // bytes per element
  const __amdgpu_buffer_rsrc_t arsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(dy), 0, (int)(unsigned)min((size_t)d.N * d.out_ctot * hw * ES, (size_t)0xffffff00u), 0x00020000);
  const __amdgpu_buffer_rsrc_t brsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(x), 0, (int)(unsigned)min((size_t)d.N * d.in_ctot * hw * ES, (size_t)0xffffff00u), 0x00020000);
  unsigned aoff[MR], boff[NT];
  bool va[MR], vb[NT];
#pragma unroll
  for (int m = 0; m < MR; ++m) {
    const int co = co0 + m * 32 + l31;
    va[m] = co < d.Cout;
    aoff[m] = (unsigned)((((size_t)d.out_coff + (va[m] ? co : 0)) * hw + half * 16) * ES);
  }
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int ci = c0 + t * 32 + l31;
    vb[t] = ci < d.Cin;
    boff[t] = (unsigned)((((size_t)d.in_coff + (vb[t] ? ci : 0)) * hw + half * 16) * ES);
  }
  float amu[NT], asc[NT], ash[NT];
  if constexpr (AFF) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int ci = min(c0 + t * 32 + l31, d.Cin - 1);
      amu[t] = in_mean[ci]; asc[t] = in_scale[ci]; ash[t] = in_shift[ci];
    }
  }
  const unsigned a_img = (unsigned)((size_t)d.out_ctot * hw * ES), b_img = (unsigned)((size_t)d.in_ctot * hw * ES);   // bytes per image

  f32x16 acc[MR][NT];
#pragma unroll
  for (int m = 0; m < MR; ++m)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][t][r] = 0.f;

  float4 a0[MR][4], b0[NT][4], a1[MR][4], b1[NT][4];
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
#pragma unroll
    for (int m = 0; m < MR; ++m) a0[m][q] = a1[m][q] = z4;
#pragma unroll
    for (int t = 0; t < NT; ++t) b0[t][q] = b1[t][q] = z4;
  }
  auto load = [&](float4 (&a)[MR][4], float4 (&b)[NT][4], int g) {
    const int n = g / segs_per_img;
    const unsigned o = (unsigned)(g - n * segs_per_img) * 32u * ES;
#pragma unroll
    for (int m = 0; m < MR; ++m)
      if (va[m]) {
#pragma unroll
        for (int q = 0; q < (NAT ? 2 : 4); ++q) {
#if W1_COAL_PROBE     // timing only (WRONG results): a quad of lanes reads 64 contiguous bytes of one row, 16 rows per instruction
          const unsigned po = (unsigned)(((size_t)d.out_coff + min(co0 + m * 32 + (lane >> 2) + 16 * (q & 1), d.Cout - 1)) * hw * ES) + (lane & 3) * 16u + 64u * (q >> 1);
          a[m][q] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(arsrc, po, (unsigned)n * a_img + o, 0));
#else
          a[m][q] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(arsrc, aoff[m], (unsigned)n * a_img + o + 16u * q, 0));
#endif
        }
      }
#pragma unroll
    for (int t = 0; t < NT; ++t)
      if (vb[t]) {
#pragma unroll
        for (int q = 0; q < (NAT ? 2 : 4); ++q) {
#if W1_COAL_PROBE
          const unsigned po = (unsigned)(((size_t)d.in_coff + min(c0 + t * 32 + (lane >> 2) + 16 * (q & 1), d.Cin - 1)) * hw * ES) + (lane & 3) * 16u + 64u * (q >> 1);
          b[t][q] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(brsrc, po, (unsigned)n * b_img + o, 0));
#else
          b[t][q] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(brsrc, boff[t], (unsigned)n * b_img + o + 16u * q, 0));
#endif
        }
      }
  };
  const bool relu_in = d.in_relu != 0;
  // AFF: the X operand is activated when it is used (not when it is loaded: that would wait for the
  // prefetch the MFMAs of the previous segment are supposed to cover); invalid channels stay 0
  auto xf = [&](float4 v, int t) {
    if constexpr (AFF) {
      v.x = (v.x - amu[t]) * asc[t] + ash[t]; v.y = (v.y - amu[t]) * asc[t] + ash[t];
      v.z = (v.z - amu[t]) * asc[t] + ash[t]; v.w = (v.w - amu[t]) * asc[t] + ash[t];
      if (relu_in) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      if (!vb[t]) v = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    return v;
  };
  auto compute = [&](const float4 (&a)[MR][4], const float4 (&b)[NT][4]) {
    if constexpr (NAT) {
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int m = 0; m < MR; ++m)
#pragma unroll
          for (int t = 0; t < NT; ++t)
            acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(wg_bf16x8, a[m][q]),
                                                                __builtin_bit_cast(wg_bf16x8, b[t][q]), acc[m][t], 0, 0, 0);
      return;
    }
    if constexpr (BX3) {
      // split-bf16 MFMAs (conv_bx3.hip): k-block q = the lane's pixels 8q..8q+7 of both operands
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        wg_bf16x8 ah[MR], am[MR], al[MR], bh[NT], bm[NT], bl[NT];
#pragma unroll
        for (int m = 0; m < MR; ++m) {
          const float v[8] = {a[m][2 * q].x, a[m][2 * q].y, a[m][2 * q].z, a[m][2 * q].w,
                              a[m][2 * q + 1].x, a[m][2 * q + 1].y, a[m][2 * q + 1].z, a[m][2 * q + 1].w};
          wg_split8(v, ah[m], am[m], al[m]);
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const float4 b0 = xf(b[t][2 * q], t), b1 = xf(b[t][2 * q + 1], t);
          const float v[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
          wg_split8(v, bh[t], bm[t], bl[t]);
        }
#pragma unroll
        for (int term = DLIO_SPLIT_Q0; term < 6; ++term)
#pragma unroll
          for (int m = 0; m < MR; ++m)
#pragma unroll
            for (int t = 0; t < NT; ++t) {
              const wg_bf16x8& av = term == 0 ? al[m] : (term == 1 || term == 3) ? am[m] : ah[m];
              const wg_bf16x8& bw = (term == 0 || term == 3 || term == 5) ? bh[t] : (term == 1 || term == 4) ? bm[t] : bl[t];
              acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bw, acc[m][t], 0, 0, 0);
            }
      }
      return;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float4 bq[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) bq[t] = xf(b[t][q], t);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int m = 0; m < MR; ++m)
#pragma unroll
          for (int t = 0; t < NT; ++t)
            acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(f4e(a[m][q], e), f4e(bq[t], e), acc[m][t], 0, 0, 0);
    }
  };

  const int total = d.N * segs_per_img;
  const int stride = splits * 4;
  int g = split * 4 + wave;
  if (g < total) load(a0, b0, g);
  while (g < total) {
    const int g1 = g + stride;
    if (g1 < total) load(a1, b1, g1);
    compute(a0, b0);
    if (g1 >= total) break;
    const int g2 = g1 + stride;
    if (g2 < total) load(a0, b0, g2);
    compute(a1, b1);
    g = g2;
  }

  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int m = 0; m < MR; ++m)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int i = ((m * NT + t) * 16 + r) * 64 + lane;
            if (w == 0) red[i] = acc[m][t][r];
            else red[i] += acc[m][t][r];
          }
    }
    __syncthreads();
  }
  float* out = wsp + (size_t)split * d.Cout * d.Cin;
  for (int idx = tid; idx < 32 * MR * 32 * NT; idx += 256) {
    const int j = idx % (NT * 32);
    const int col = idx / (NT * 32);
    if (c0 + j >= d.Cin || co0 + col >= d.Cout) continue;
    const int m = col >> 5, row = col & 31;
    const int hf = (row >> 2) & 1;
    const int r = (row & 3) + 4 * (row >> 3);
    const int t = j >> 5, lj = j & 31;
    out[(size_t)(co0 + col) * d.Cin + c0 + j] = red[((m * NT + t) * 16 + r) * 64 + hf * 32 + lj];
  }
}

// dw[i] = sum over split slabs, in a fixed order: 4 waves each own every 4th slab of 64
// consecutive elements (8 independent loads in flight), partials combined through LDS.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ wsp,
                                                           float* __restrict__ dw, int64_t n,
                                                           int splits, int accumulate) {
  __shared__ float red[4][64];
  const int e = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int64_t i = (int64_t)blockIdx.x * 64 + e;
  float s = 0.f;
  if (i < n) {
    int k = g;
    for (; k + 28 < splits; k += 32) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = wsp[(size_t)(k + 4 * u) * n + i];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; k < splits; k += 4) s += wsp[(size_t)k * n + i];
  }
  red[g][e] = s;
  __syncthreads();
  if (g == 0 && i < n) {
    const float v = ((red[0][e] + red[1][e]) + red[2][e]) + red[3][e];
    dw[i] = accumulate ? dw[i] + v : v;
  }
}

struct WgPlan {
  int nt, mr, ckmax, co_tiles, ci_chunks, tiles_w, tiles_h, splits, accumulate = 0;
  size_t ws_bytes;
};

int nt_for(const DlioConvDesc& d) {
  if (d.KH == 1 && d.KW == 1) return (d.SH == 1 && d.SW == 1) ? 2 : 1;
  static const int nt33 = 5;   // tuning knob (4 or 5)
  if (d.KH == 3 && d.KW == 3 && d.SH == 1 && d.SW == 1 && nt33 == 4) return 4;
  return 5;
}

// 32-channel tiles (80 accumulator registers -> two workgroups per CU) for the multi-tap kernels;
// 64-channel tiles for 1x1 where the dY tile is the re-read operand.  DLIO_WGRAD_MR overrides.
int mr_for(const DlioConvDesc& d) {
  static const int force = 0;
  const bool has_mr1 = d.SH == 1 && d.SW == 1 && ((d.KH == 1 && d.KW == 1) || (d.KH == 3 && d.KW == 3));
  if (!has_mr1) return 2;
  if (force == 1 || force == 2) return force;
  // micro-bench (tools/bench_conv.py wgrad): 3x3 always faster with 32-channel tiles (2 workgroups
  // per CU); 1x1 only when Cout <= 32 (otherwise the extra dY re-reads cost more)
  if (d.KH == 3) return 1;
  return d.Cout <= 32 ? 1 : 2;
}

bool make_plan(const DlioConvDesc& d, WgPlan& p) {
  p.nt = nt_for(d);
  const int taps = d.KH * d.KW;
  p.ckmax = (p.nt * 32) / taps;
  if (taps == 1 && d.SH * d.SW == 4) p.ckmax /= 2;
  if (p.ckmax < 1) return false;
  p.mr = mr_for(d);
  p.co_tiles = cdiv(d.Cout, 32 * p.mr);
  p.ci_chunks = cdiv(d.Cin, p.ckmax);
  p.tiles_w = cdiv(d.OW, 32);
  p.tiles_h = cdiv(d.OH, 4);
  const int64_t total_tiles = (int64_t)d.N * p.tiles_w * p.tiles_h;
  int64_t pairs = (int64_t)p.co_tiles * p.ci_chunks;
  // one workgroup per CU: the weight-gradient kernels run on companion streams beside the data-gradient chain, and with
  // the XCD-aware order half the workgroups leave the chain more of the chip than they lose (sweep: 256 / 192 -> 26.75,
  // 512 -> 27.0, 128 -> 26.96 ms/step)
  static const int tgt = 256;
  // floor, not ceil: a block more than the slots costs a whole extra round
  int64_t splits = tgt / pairs > 0 ? tgt / pairs : 1;
  if (splits > total_tiles) splits = total_tiles;
  const size_t slab = (size_t)d.Cout * d.Cin * taps * 4;
  const size_t cap = (size_t)96 << 20;
  if (splits * slab > cap) splits = (int64_t)(cap / slab);
  if (splits < 1) splits = 1;
  p.splits = (int)splits;
  p.ws_bytes = (size_t)p.splits * slab;
  return true;
}

template <int KH, int KW, int SH, int SW, int NT, int MRW>
int launch_mr(const float* x, const float* dy, float* dw, const float* in_mean,
              const float* in_scale, const float* in_shift, float* wsp, const DlioConvDesc& d,
              const WgPlan& p, hipStream_t s) {
  using C = WgCfg<KH, KW, SH, SW, NT, MRW>;
  // float4 staging when rows are 16-B aligned (only instantiated for the stride-1 3x3 / 1x1 taps)
  constexpr bool CAN_VEC = SH == 1 && SW == 1 && ((KH == 1 && KW == 1) || (KH == 3 && KW == 3));
  const bool vec = CAN_VEC && (d.OW & 3) == 0 && (d.W & 3) == 0 &&
                   ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy)) & 15) == 0;
  // the multi-tap launches this staged kernel still takes -- the PointSeg stem (3x5 stride (1, 2)), the 5x7 stems and the strided
  // / 3x5 layers of FlowNet, ResNet and Simple-1 whose operands come without a magnitude -- : three-piece bf16 products
  constexpr bool STEM = KH * KW > 1 && !(KH == 3 && KW == 3 && SH == 1 && SW == 1) && MRW == 2;
  static const bool stem_bx3 = !(getenv("DLIO_WGRAD_STEM_BX3") && atoi(getenv("DLIO_WGRAD_STEM_BX3")) == 0);
  auto kern = (CAN_VEC && vec) ? conv_wgrad_kernel<KH, KW, SH, SW, NT, MRW, CAN_VEC>
                               : conv_wgrad_kernel<KH, KW, SH, SW, NT, MRW, false>;
  if constexpr (STEM) {
    // (dY staged with 16-byte loads where its rows allow: 8 instead of 32 load instructions per thread and tile; the X' patch
    //  of a strided layer stays scalar)
    const bool dyvec = (d.OW & 3) == 0 && (reinterpret_cast<uintptr_t>(dy) & 15) == 0;
    if (stem_bx3) kern = dyvec ? conv_wgrad_kernel<KH, KW, SH, SW, NT, MRW, true, true>
                               : conv_wgrad_kernel<KH, KW, SH, SW, NT, MRW, false, true>;
  }
  hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES);
  const int blocks = p.co_tiles * p.ci_chunks * p.splits;
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), C::LDS_BYTES, s, x, dy, wsp, in_mean,
                     in_scale, in_shift, d, p.co_tiles, p.ci_chunks, p.splits, p.tiles_w,
                     p.tiles_h);
  int rc = dlio_check_launch();
  if (rc) return rc;
  const int64_t n = (int64_t)d.Cout * d.Cin * KH * KW;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)cdiv64(n, 64)), dim3(256), 0, s, wsp, dw, n,
                     p.splits, p.accumulate);
  return dlio_check_launch();
}

template <int KH, int KW, int SH, int SW, int NT>
int launch(const float* x, const float* dy, float* dw, const float* in_mean,
           const float* in_scale, const float* in_shift, float* wsp, const DlioConvDesc& d,
           const WgPlan& p, hipStream_t s) {
  // the 32-channel variant is only built for the stride-1 3x3 / 1x1 taps (the headline path)
  constexpr bool HAS_MR1 = SH == 1 && SW == 1 && ((KH == 1 && KW == 1) || (KH == 3 && KW == 3));
  if constexpr (HAS_MR1) {
    if (p.mr == 1)
      return launch_mr<KH, KW, SH, SW, NT, 1>(x, dy, dw, in_mean, in_scale, in_shift, wsp, d, p, s);
  }
  return launch_mr<KH, KW, SH, SW, NT, 2>(x, dy, dw, in_mean, in_scale, in_shift, wsp, d, p, s);
}


// plan of the direct 1x1 kernel (geometry only; pointer alignment is checked at launch and the
// staged kernel's workspace is never smaller, see dlio_conv2d_wgrad_ws_bytes)
struct Wg1Plan { int mr, nt, co_tiles, ci_chunks, splits, segs; size_t ws_bytes; };

bool make_plan_1x1(const DlioConvDesc& d, Wg1Plan& p) {
  static const int off = getenv("DLIO_WGRAD_1X1_DIRECT") ? atoi(getenv("DLIO_WGRAD_1X1_DIRECT")) == 0 : 0;
  if (off || d.KH != 1 || d.KW != 1 || d.SH != 1 || d.SW != 1 || d.PH || d.PW) return false;
  const int64_t hw = (int64_t)d.H * d.W;
  if (hw % 32 || d.OH != d.H || d.OW != d.W) return false;
  p.mr = d.Cout <= 32 ? 1 : 2;
  p.nt = d.Cin <= 32 ? 1 : 2;
  p.co_tiles = cdiv(d.Cout, 32 * p.mr);
  p.ci_chunks = cdiv(d.Cin, 32 * p.nt);
  p.segs = (int)(hw / 32);
  const int64_t total = (int64_t)d.N * p.segs;
  const int64_t pairs = (int64_t)p.co_tiles * p.ci_chunks;
  // measured: in isolation (tools/conv_table.py, blocks 128..768) one workgroup per CU is the
  // sweet spot and a partial second round (320) costs 15-50 %; inside the training step, where
  // the other encoder's kernels share the chip, two per CU is 0.2 ms/step better (tools/sweep_env.sh)
  // (re-measured with the XCD-aware order: one per CU 25.51, 1.5 per CU 25.61, two per CU 25.62 ms/step)
  static const int tgt = dlio_num_cus();
  int64_t splits = tgt / pairs > 0 ? tgt / pairs : 1;
  // every wave should stream at least MINSEG segments, or prologue + slab reduction dominate
  static const int minseg = 2;
  if (splits * 4 * minseg > total) splits = cdiv64(total, 4 * minseg);
  const size_t slab = (size_t)d.Cout * d.Cin * 4;
  const size_t cap = (size_t)96 << 20;
  if (splits * slab > cap) splits = (int64_t)(cap / slab);
  if (splits < 1) splits = 1;
  p.splits = (int)splits;
  p.ws_bytes = (size_t)p.splits * slab;
  return true;
}

template <int MR, int NT>
int launch_1x1(const float* x, const float* dy, float* dw, float* wsp, const DlioConvDesc& d,
               const Wg1Plan& p, int accumulate, hipStream_t s, const float* in_mean = nullptr,
               const float* in_scale = nullptr, const float* in_shift = nullptr) {
  // split-bf16 MFMAs where the fp32 MFMA time shows (64 x 64-channel tiles); narrow layers are HBM-bound
  static const int bx3 = 1;
  static const int aff_bx3 = 1;
  // in-affine + split-bf16 on the 64 x 64 tile needs 26 registers more than two waves per SIMD leave: that instantiation is
  // built for one wave per SIMD (launch bounds (256, 1): 210 VGPR + 64 AGPR, no scratch) -- the kernel runs one workgroup per
  // CU anyway; family 2.30 -> 2.21 ms exclusive
  if (in_scale && aff_bx3 && (bx3 == 2 || (bx3 && MR == 2 && NT == 2)))
    hipLaunchKernelGGL((wgrad1x1_direct_kernel<MR, NT, true, false, true>), dim3(p.co_tiles * p.ci_chunks * p.splits),
                       dim3(256), 0, s, x, dy, wsp, d, p.co_tiles, p.ci_chunks, p.splits, p.segs, in_mean, in_scale,
                       in_shift);
  else if (in_scale)
    hipLaunchKernelGGL((wgrad1x1_direct_kernel<MR, NT, false, false, true>), dim3(p.co_tiles * p.ci_chunks * p.splits),
                       dim3(256), 0, s, x, dy, wsp, d, p.co_tiles, p.ci_chunks, p.splits, p.segs, in_mean, in_scale,
                       in_shift);
  else if (bx3 == 2 || (bx3 && MR == 2 && NT == 2))
    hipLaunchKernelGGL((wgrad1x1_direct_kernel<MR, NT, true>), dim3(p.co_tiles * p.ci_chunks * p.splits),
                       dim3(256), 0, s, x, dy, wsp, d, p.co_tiles, p.ci_chunks, p.splits, p.segs);
  else
    hipLaunchKernelGGL((wgrad1x1_direct_kernel<MR, NT>), dim3(p.co_tiles * p.ci_chunks * p.splits),
                       dim3(256), 0, s, x, dy, wsp, d, p.co_tiles, p.ci_chunks, p.splits, p.segs);
  int rc = dlio_check_launch();
  if (rc) return rc;
  const int64_t n = (int64_t)d.Cout * d.Cin;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)cdiv64(n, 64)), dim3(256), 0, s, wsp, dw, n,
                     p.splits, accumulate);
  return dlio_check_launch();
}

// ---- 3x5 taps, stride (1, 2), padding (1, 2) (the PointSeg stem, pointseg_net.py:18-20; FlowNet conv2 / conv3,
// lidar_feat_nets.py:248-251) through the 3x3 stride-1 kernel: input column 2 ow + kw - 2 is column ow + kw/2 - 1 of the
// EVEN-column image for even kw and column ow + (kw-1)/2 - 1 of the ODD-column image for odd kw, i.e. the five column taps
// are the three taps of a "same" 3x3 correlation with the even image plus the first two of one with the odd image.  The
// input is split into its two column phases once (one pass, 16-byte accesses), wgrad3_kernel runs on each, and the merge
// scatters the two [Cout][Cin][3][3] sums into dW[Cout][Cin][3][5] (the odd phase's third column tap is a tap the layer
// does not have: computed and dropped, 18 taps of MFMA work for 15).
__global__ __launch_bounds__(256) void deinterleave_cols_kernel(const float* __restrict__ x, float* __restrict__ xe,
                                                                float* __restrict__ xo, int64_t rows, int W8, int64_t in_row_stride_c,
                                                                int Cin, int in_ctot, int in_coff, int H) {
  // one thread: 8 consecutive input columns of one row -> 4 even + 4 odd
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= rows * W8) return;
  const int64_t row = i / W8;                 // (n * Cin + c) * H + h over the DENSE phase images
  const int q = (int)(i - row * W8);
  const int64_t nc = row / H;
  const int h = (int)(row - nc * H);
  const int64_t n = nc / Cin;
  const int c = (int)(nc - n * Cin);
  const float* src = x + (((size_t)n * in_ctot + in_coff + c) * H + h) * (size_t)(8 * W8) + 8 * q;
  (void)in_row_stride_c;
  const float4 a = *reinterpret_cast<const float4*>(src), b = *reinterpret_cast<const float4*>(src + 4);
  *reinterpret_cast<float4*>(xe + (size_t)row * (4 * W8) + 4 * q) = make_float4(a.x, a.z, b.x, b.z);
  *reinterpret_cast<float4*>(xo + (size_t)row * (4 * W8) + 4 * q) = make_float4(a.y, a.w, b.y, b.w);
}

__global__ __launch_bounds__(256) void wgrad_merge_3x5_kernel(const float* __restrict__ te, const float* __restrict__ to,
                                                              float* __restrict__ dw, int64_t n15, int accumulate) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n15) return;
  const int64_t cc = i / 15;
  const int t = (int)(i - cc * 15), kh = t / 5, kw = t - 5 * kh;
  const float v = (kw & 1) ? to[cc * 9 + kh * 3 + (kw >> 1)] : te[cc * 9 + kh * 3 + (kw >> 1)];
  dw[i] = accumulate ? dw[i] + v : v;
}

// ---- 3x3 taps, stride (2, 2), padding 1 (FlowNet conv4-6, lidar_feat_nets.py:252-257; ResNet layer3 / layer4) as NINE 1x1
// weight gradients: tap (kh, kw) pairs dY[oh][ow] with X[2 oh + kh - 1][2 ow + kw - 1], i.e. with the (row parity, column
// parity) phase image of X at index (oh + dr, ow + dc), dr = -1 for kh = 0, dc = -1 for kw = 0, else 0.  dY and the four
// phase images are copied once into planes of the SAME padded geometry ((OH + 2) rows x PW columns, PW % 4 == 0, zero
// rows / column in front), so a shift is a flat pointer offset and the zero padding of dY silences whatever a shifted
// read wraps onto; the two odd-column phases also get a copy shifted by one column, so every operand stays 16-byte
// aligned.  Each tap is then the HBM-bound direct 1x1 kernel (wgrad1x1_direct_kernel) on (dY plane, phase plane).
__global__ __launch_bounds__(256) void s2_pad_dy_kernel(const float* __restrict__ dy, float* __restrict__ dyp, int64_t planes_out,
                                                        int Cout, int out_ctot, int out_coff, int OH, int OW, int PW, int S) {
  const int S4 = S >> 2;
  const int64_t total = planes_out * S4;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t pl = i / S4;
    const int f = (int)(i - pl * S4) * 4;
    const int rp = f / PW, cp = f - rp * PW;              // PW % 4 == 0: the four elements share a row
    const int64_t n = pl / Cout;
    const int c = (int)(pl - n * Cout);
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    const int oh = rp - 2;
    if (oh >= 0 && oh < OH) {
      const float* src = dy + ((n * out_ctot + out_coff + c) * (int64_t)OH + oh) * OW;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int ow = cp + e - 1;
        if (ow >= 0 && ow < OW) v[e] = src[ow];
      }
    }
    *reinterpret_cast<float4*>(dyp + pl * S + f) = make_float4(v[0], v[1], v[2], v[3]);
  }
}

// images: 0 = (even rows, even cols), 1 = (even, odd), 2 = (even, odd) shifted one column, 3 = (odd, even), 4 = (odd, odd),
// 5 = (odd, odd) shifted
__global__ __launch_bounds__(256) void s2_phase_split_kernel(const float* __restrict__ x, float* __restrict__ img, int64_t planes,
                                                             int Cin, int in_ctot, int in_coff, int H, int W, int OH, int PW,
                                                             int S, int64_t img_stride) {
  const int S4 = S >> 2;
  const int64_t total = planes * S4;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t pl = i / S4;
    const int f = (int)(i - pl * S4) * 4;
    const int rp = f / PW, cp = f - rp * PW;
    const int64_t n = pl / Cin;
    const int c = (int)(pl - n * Cin);
    const float* xp = x + (n * in_ctot + in_coff + c) * (int64_t)H * W;
    const int r = rp - 2;
#pragma unroll
    for (int im = 0; im < 6; ++im) {
      const int rodd = im >= 3, codd = (im % 3) != 0, sh = (im % 3) == 2;
      float v[4] = {0.f, 0.f, 0.f, 0.f};
      const int ih = 2 * r + rodd;
      if (r >= 0 && r < OH && ih < H) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int cc = cp + e - 1 - sh;                // phase column
          const int iw = 2 * cc + codd;
          if (cc >= 0 && iw < W) v[e] = xp[(int64_t)ih * W + iw];
        }
      }
      *reinterpret_cast<float4*>(img + im * img_stride + pl * S + f) = make_float4(v[0], v[1], v[2], v[3]);
    }
  }
}

__global__ __launch_bounds__(256) void wgrad_scatter_taps_kernel(const float* __restrict__ t9, float* __restrict__ dw,
                                                                 int64_t ncc, int accumulate) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= ncc * 9) return;
  const int64_t cc = i / 9;
  const int tap = (int)(i - cc * 9);
  const float v = t9[(int64_t)tap * ncc + cc];
  dw[i] = accumulate ? dw[i] + v : v;
}

// ---- C[m][n] = sum_k A[m][k] B[n][k] over the planes above ([image][channel][S] rows, K = pixels contiguous) on the split-bf16
// scheme with 128 x 128 tiles: the tap products of wide layers (256 x 512 channels) through the direct 1x1 kernel's
// 64 x 64 tiles re-read both operands 4x / 8x.  Workgroup = 4 waves, each 64 x 64 (2 x 2 MFMA tiles); a stage = 32 pixels of
// 128 + 128 rows, loaded as float4 (eight lanes cover the 128 contiguous bytes of a row), split ONCE into three bf16 planes
// and stored k-contiguous in LDS (row stride 40 bf16 = 80 bytes: 16-byte aligned fragments, banks spread); global loads of
// stage s + 1 fly under the MFMAs of stage s, one barrier per stage; the pixel range of all images is cut into `splits`
// slabs (fixed-order reduction by wgrad_reduce_kernel).  M % 128 == 0, N % 128 == 0, S % 32 == 0.
#ifndef DLIO_GEMM_GK
#define DLIO_GEMM_GK 32
#endif
constexpr int GT = 128, GK = DLIO_GEMM_GK, GLDK = GK + 8;
struct GemmTaps { long long boff[9]; };      // float offset of each tap's B operand (phase image + row shift) from B
__global__ __launch_bounds__(256, GK == 16 ? 4 : 2) void gemm_nt_bx3_kernel(const float* __restrict__ A, const float* __restrict__ B0,
                                                             float* __restrict__ slab0, int M, int N, int S, int nimg,
                                                             int tiles_n, int splits, GemmTaps taps) {
  // blockIdx.y = tap: the nine products of a layer in ONE launch (they share A = dY; 9 x the workgroups, so few pixel slabs)
  const float* B = B0 + taps.boff[blockIdx.y];
  float* slab = slab0 + (size_t)blockIdx.y * splits * M * N;
  extern __shared__ __attribute__((aligned(16))) __bf16 gsm[];      // [A | B][3 planes][128][GLDK]
  constexpr int PLANE = GT * GLDK, OPER = 3 * PLANE, STAGE = 2 * OPER;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
  int bid = blockIdx.x;
  const int split = bid % splits; bid /= splits;
  const int tn = bid % tiles_n, tm = bid / tiles_n;
  const int spi = S / GK;                                            // stages per image
  const int64_t total = (int64_t)nimg * spi;
  const int64_t g0 = total * split / splits, g1 = total * (split + 1) / splits;
  const size_t a_img = (size_t)M * S, b_img = (size_t)N * S;
  const float* Ab = A + (size_t)tm * GT * S;
  const float* Bb = B + (size_t)tn * GT * S;
  constexpr int LPR = GK / 4, RPP = 256 / LPR, NPASS = GT / RPP;     // float4 per row, rows per pass, passes
  const int lr = tid / LPR, lk = (tid % LPR) * 4;                    // row (+RPP per pass), first k of the thread's float4
  float4 ra[NPASS], rb[NPASS];
  auto gload = [&](int64_t g) {
    const int64_t img = g / spi;
    const int ks = (int)(g - img * spi) * GK + lk;
#pragma unroll
    for (int i = 0; i < NPASS; ++i) {
      ra[i] = *reinterpret_cast<const float4*>(Ab + img * a_img + (size_t)(lr + RPP * i) * S + ks);
      rb[i] = *reinterpret_cast<const float4*>(Bb + img * b_img + (size_t)(lr + RPP * i) * S + ks);
    }
  };
  auto sstore = [&](int st) {
    __bf16* base = gsm + (size_t)st * STAGE;
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
      for (int i = 0; i < NPASS; ++i) {
        const float4 v = o == 0 ? ra[i] : rb[i];
        const unsigned h0 = wg_cvt_pk(v.x, v.y), h1 = wg_cvt_pk(v.z, v.w);
        const float r0 = v.x - wg_as_f(h0 << 16), r1 = v.y - wg_as_f(h0 & 0xffff0000u);
        const float r2 = v.z - wg_as_f(h1 << 16), r3 = v.w - wg_as_f(h1 & 0xffff0000u);
        const unsigned m0 = wg_cvt_pk(r0, r1), m1 = wg_cvt_pk(r2, r3);
        const unsigned q0 = wg_cvt_pk(r0 - wg_as_f(m0 << 16), r1 - wg_as_f(m0 & 0xffff0000u));
        const unsigned q1 = wg_cvt_pk(r2 - wg_as_f(m1 << 16), r3 - wg_as_f(m1 & 0xffff0000u));
        __bf16* dst = base + o * OPER + (lr + RPP * i) * GLDK + lk;
        *reinterpret_cast<uint2*>(dst) = make_uint2(h0, h1);
        *reinterpret_cast<uint2*>(dst + PLANE) = make_uint2(m0, m1);
        *reinterpret_cast<uint2*>(dst + 2 * PLANE) = make_uint2(q0, q1);
      }
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][t][r] = 0.f;
  const int wm = wave >> 1, wn = wave & 1;
  auto compute = [&](int st) {
    const __bf16* base = gsm + (size_t)st * STAGE;
#pragma unroll
    for (int kk = 0; kk < GK; kk += 16) {
      wg_bf16x8 a[2][3], b[2][3];
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int p = 0; p < 3; ++p)
          a[m][p] = *reinterpret_cast<const wg_bf16x8*>(base + p * PLANE + (64 * wm + 32 * m + l31) * GLDK + kk + 8 * half);
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int p = 0; p < 3; ++p)
          b[t][p] = *reinterpret_cast<const wg_bf16x8*>(base + OPER + p * PLANE + (64 * wn + 32 * t + l31) * GLDK + kk + 8 * half);
      constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};     // smallest products first
#pragma unroll
      for (int q = DLIO_SPLIT_Q0; q < 6; ++q)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int t = 0; t < 2; ++t)
            acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m][PA[q]], b[t][PB[q]], acc[m][t], 0, 0, 0);
    }
  };
  if (g0 < g1) {
    // ONE LDS stage per workgroup (61 KB) so that two workgroups share a CU: a stage of MFMAs is shorter than a memory
    // round trip, the second workgroup's MFMAs are what covers it (single workgroup with two LDS stages: 35 % MFMA busy)
    gload(g0);
    sstore(0);
    __syncthreads();
    for (int64_t g = g0; g < g1; ++g) {
      const bool more = g + 1 < g1;
      if (more) gload(g + 1);
      compute(0);
      __syncthreads();
      if (more) sstore(0);
      __syncthreads();
    }
  }
  float* out = slab + (size_t)split * M * N + (size_t)(tm * GT + 64 * wm) * N + tn * GT + 64 * wn;
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        out[(size_t)(32 * m + (r & 3) + 8 * (r >> 2) + 4 * half) * N + 32 * t + l31] = acc[m][t][r];
}

// dW[cc][tap] (+)= sum over the tap's slabs, fixed order (slabs [tap][split][ncc])
__global__ __launch_bounds__(256) void wgrad_reduce_taps_kernel(const float* __restrict__ slabs, float* __restrict__ dw,
                                                                int64_t ncc, int splits, int accumulate) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= ncc * 9) return;
  const int tap = (int)(i / ncc);
  const int64_t cc = i - (int64_t)tap * ncc;
  const float* p = slabs + (size_t)tap * splits * ncc + cc;
  float v = 0.f;
  for (int k = 0; k < splits; ++k) v += p[(size_t)k * ncc];
  float* o = dw + cc * 9 + tap;
  *o = accumulate ? *o + v : v;
}

struct WgS2Plan { DlioConvDesc sub; Wg1Plan q; int PW, S, gemm, gsplits; size_t dyp_floats, img_floats, ws_bytes; };

bool make_plan_s2_taps(const DlioConvDesc& d, WgS2Plan& p) {
  static const int on = 1;
  if (!on || d.KH != 3 || d.KW != 3 || d.SH != 2 || d.SW != 2 || d.PH != 1 || d.PW != 1 || d.Cin < 32 || d.Cout < 32 ||
      d.OH != (d.H - 1) / 2 + 1 || d.OW != (d.W - 1) / 2 + 1 || (int64_t)d.N * d.OH * d.OW < 8192)
    return false;
  p.PW = (d.OW + 1 + 3) & ~3;
  p.S = ((d.OH + 2) * p.PW + 31) & ~31;
  p.sub = d;
  p.sub.H = 1; p.sub.W = p.S; p.sub.OH = 1; p.sub.OW = p.S;
  p.sub.KH = p.sub.KW = 1; p.sub.SH = p.sub.SW = 1; p.sub.PH = p.sub.PW = 0;
  p.sub.in_ctot = d.Cin; p.sub.in_coff = 0; p.sub.out_ctot = d.Cout; p.sub.out_coff = 0; p.sub.in_relu = 0;
  if (!make_plan_1x1(p.sub, p.q)) return false;
  p.dyp_floats = (size_t)d.N * d.Cout * p.S;
  p.img_floats = (size_t)d.N * d.Cin * p.S;
  if ((p.dyp_floats + 6 * p.img_floats) * 4 >= 0xffffff00ull) return false;       // 32-bit buffer offsets per operand
  // wide layers: the tap products on the 128 x 128-tile GEMM (DLIO_WGRAD_S2_GEMM, default 1)
  static const int gemm_on = 1;
  p.gemm = gemm_on && d.Cout % GT == 0 && d.Cin % GT == 0;
  p.gsplits = 1;
  size_t slab_bytes = p.q.ws_bytes;
  if (p.gemm) {
    const int tiles = (d.Cout / GT) * (d.Cin / GT);
    const int64_t stages = (int64_t)d.N * (p.S / GK);
    int sp = (GK == 16 ? 4 : 2) * dlio_num_cus() / (tiles * 9);                // nine taps per launch, two (four) workgroups per CU
    if (sp < 1) sp = 1;
    if (sp > stages / 4) sp = (int)(stages / 4 > 0 ? stages / 4 : 1);       // at least four stages per workgroup
    p.gsplits = sp;
    const size_t g = (size_t)9 * sp * d.Cout * d.Cin * sizeof(float);
    if (g > slab_bytes) slab_bytes = g;
  }
  p.ws_bytes = (p.dyp_floats + 6 * p.img_floats + (size_t)9 * d.Cout * d.Cin) * sizeof(float) + slab_bytes + 256;
  return true;
}

struct Wg35Plan { DlioConvDesc sub; DlioWgrad3Plan p3; size_t phase_floats, ws_bytes; };

bool make_plan_3x5s2(const DlioConvDesc& d, Wg35Plan& q) {
  static const int on = 1;
  // (the 5-channel PointSeg stem fills 5 of the kernel's 16-channel slots: 24.09 vs 23.92 ms per step, it stays on the staged kernel)
  if (!on || d.Cin < 16 || d.KH != 3 || d.KW != 5 || d.SH != 1 || d.SW != 2 || d.PH != 1 || d.PW != 2 || (d.W & 7) != 0 ||
      d.OW != d.W / 2 || d.OH != d.H)
    return false;
  q.sub = d;
  q.sub.W = d.W / 2; q.sub.KW = 3; q.sub.SW = 1; q.sub.PW = 1;
  q.sub.in_ctot = d.Cin; q.sub.in_coff = 0; q.sub.in_relu = 0;
  if (!dlio_wgrad3_plan(q.sub, 4, q.p3)) return false;
  q.phase_floats = (size_t)d.N * d.Cin * d.H * (d.W / 2);
  const size_t t9 = (size_t)d.Cout * d.Cin * 9 * sizeof(float);
  q.ws_bytes = 2 * q.phase_floats * sizeof(float) + 2 * q.p3.ws_bytes + 2 * t9 + 64;
  return true;
}

// column phases -> two 3x3 stride-1 launches -> merge (see deinterleave_cols_kernel); amax_x / amax_dy: the two-piece kernel
// (the phases are subsets of x: its bound holds for both)
int run_3x5s2(const float* x, const float* dy, float* dw, float* wsp, const DlioConvDesc& d, const Wg35Plan& q35,
              int accumulate, hipStream_t s, const float* amax_x, const float* amax_dy) {
  const size_t pf = (q35.phase_floats + 3) & ~(size_t)3, sl = (q35.p3.ws_bytes / sizeof(float) + 3) & ~(size_t)3;
  const size_t t9 = (size_t)d.Cout * d.Cin * 9;
  float* xe = wsp; float* xo = xe + pf; float* se = xo + pf; float* so = se + sl; float* te = so + sl; float* to = te + t9;
  const int64_t rows = (int64_t)d.N * d.Cin * d.H;
  const int W8 = d.W / 8;
  hipLaunchKernelGGL(deinterleave_cols_kernel, dim3((unsigned)cdiv64(rows * W8, 256)), dim3(256), 0, s, x, xe, xo, rows, W8,
                     (int64_t)0, d.Cin, d.in_ctot, d.in_coff, d.H);
  int rc = dlio_check_launch();
  if (!rc) rc = dlio_wgrad3_launch(xe, dy, se, q35.sub, q35.p3, 4, s, amax_x, amax_dy);
  if (!rc) rc = dlio_wgrad3_launch(xo, dy, so, q35.sub, q35.p3, 4, s, amax_x, amax_dy);
  if (rc) return rc;
  const int64_t n9 = (int64_t)t9;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)cdiv64(n9, 64)), dim3(256), 0, s, se, te, n9, q35.p3.splits, 0);
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)cdiv64(n9, 64)), dim3(256), 0, s, so, to, n9, q35.p3.splits, 0);
  const int64_t n15 = (int64_t)d.Cout * d.Cin * 15;
  hipLaunchKernelGGL(wgrad_merge_3x5_kernel, dim3((unsigned)cdiv64(n15, 256)), dim3(256), 0, s, te, to, dw, n15, accumulate);
  return dlio_check_launch();
}

}  // namespace

// bf16 x / dy (mixed-precision path): 3x3 stride-1 (dY-direct kernel) and 1x1 stride-1 (direct kernel),
// fp32 slabs + the fixed-order reduction into dw (fp32, optionally accumulated)
extern "C" int dlio_conv2d_wgrad_bf16(const void* x, const void* dy, float* dw, void* ws, size_t ws_bytes,
                                      int accumulate, const DlioConvDesc* dp, dlio_stream_t stream) {
  if (!x || !dy || !dw || !dp || !ws) return DLIO_EINVAL;
  const DlioConvDesc& d = *dp;
  if (d.SH != 1 || d.SW != 1) return DLIO_EUNSUP;
  if (((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy)) & 15) != 0) return DLIO_EUNSUP;
  if ((size_t)d.N * d.in_ctot * d.H * d.W * 2 >= 0xffffff00ull || (size_t)d.N * d.out_ctot * d.OH * d.OW * 2 >= 0xffffff00ull)
    return DLIO_EUNSUP;
  hipStream_t s = as_stream(stream);
  float* wsp = reinterpret_cast<float*>(ws);
  const float* xf = reinterpret_cast<const float*>(x);
  const float* df = reinterpret_cast<const float*>(dy);
  const double flops = 2.0 * d.N * (double)d.OH * d.OW * d.Cout * (double)d.Cin * d.KH * d.KW;
  const double bytes = 2.0 * d.N * ((double)d.Cin * d.H * d.W + (double)d.Cout * d.OH * d.OW);
  DlioProfScope prof(11, s, flops, bytes);
  if (d.KH == 1 && d.KW == 1) {
    Wg1Plan q;
    if (!make_plan_1x1(d, q)) return DLIO_EUNSUP;
    if (ws_bytes < q.ws_bytes) return DLIO_EWS;
    const dim3 grid(q.co_tiles * q.ci_chunks * q.splits);
#define W1(mr, nt) hipLaunchKernelGGL((wgrad1x1_direct_kernel<mr, nt, false, true>), grid, dim3(256), 0, s, xf, df, wsp, d, \
                                      q.co_tiles, q.ci_chunks, q.splits, q.segs)
    if (q.mr == 1 && q.nt == 1) W1(1, 1);
    else if (q.mr == 1) W1(1, 2);
    else if (q.nt == 1) W1(2, 1);
    else W1(2, 2);
#undef W1
    int rc = dlio_check_launch();
    if (rc) return rc;
    const int64_t n = (int64_t)d.Cout * d.Cin;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)cdiv64(n, 64)), dim3(256), 0, s, wsp, dw, n, q.splits, accumulate);
    return dlio_check_launch();
  }
  if (d.KH != 3 || d.KW != 3 || (d.OW & 7) || (d.W & 7)) return DLIO_EUNSUP;
  DlioWgrad3Plan p3;
  if (dlio_wgrad3_plan(d, 2, p3) && ws_bytes >= p3.ws_bytes) {
    int rc = dlio_wgrad3_launch(x, dy, wsp, d, p3, 2, s);
    if (rc) return rc;
    const int64_t n = (int64_t)d.Cout * d.Cin * 9;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)cdiv64(n, 64)), dim3(256), 0, s, wsp, dw, n, p3.splits, accumulate);
    return dlio_check_launch();
  }
  return DLIO_EUNSUP;
}

extern "C" size_t dlio_conv2d_wgrad_ws_bytes(const DlioConvDesc* d) {
  WgPlan p;
  if (!d || !make_plan(*d, p)) return 0;
  size_t need = p.ws_bytes;
  Wg1Plan q;
  if (make_plan_1x1(*d, q) && q.ws_bytes > need) need = q.ws_bytes;
  DlioWgrad3Plan p3;
  if (dlio_wgrad3_plan(*d, 4, p3) && p3.ws_bytes > need) need = p3.ws_bytes;
  Wg35Plan q35;
  if (make_plan_3x5s2(*d, q35) && q35.ws_bytes > need) need = q35.ws_bytes;
  WgS2Plan qs2;
  if (make_plan_s2_taps(*d, qs2) && qs2.ws_bytes > need) need = qs2.ws_bytes;
  return need;
}

/* 3x3 stride-1 pad-1 weight gradient on the two-piece fp16 split (wgrad3_kernel<.., H2>): amax_x / amax_dy = device
 * floats with the operands' largest magnitudes (or bounds on them), left by the kernels that produced the tensors. */
extern "C" int dlio_conv3x3_wgrad_h2_ok(const DlioConvDesc* dp) {
  DlioWgrad3Plan p3;
  Wg35Plan q35;
  return dp && dp->N > 0 && dp->Cin > 0 && dp->Cout > 0 && (dlio_wgrad3_plan(*dp, 4, p3) || make_plan_3x5s2(*dp, q35));
}

extern "C" int dlio_conv3x3_wgrad_h2(const float* x, const float* amax_x, const float* dy, const float* amax_dy, float* dw,
                                     void* ws, size_t ws_bytes, int accumulate, const DlioConvDesc* dp, dlio_stream_t stream) {
  if (!x || !amax_x || !dy || !amax_dy || !dw || !dp || !ws) return DLIO_EINVAL;
  const DlioConvDesc& d = *dp;
  DlioWgrad3Plan p3;
  if (!dlio_conv3x3_wgrad_h2_ok(dp)) return DLIO_EUNSUP;
  hipStream_t s = as_stream(stream);
  if (!dlio_wgrad3_plan(d, 4, p3)) {
    // 3x5 taps, stride (1, 2): the two column phases on the two-piece kernel (run_3x5s2)
    Wg35Plan q35;
    if (!make_plan_3x5s2(d, q35)) return DLIO_EUNSUP;
    if (ws_bytes < q35.ws_bytes) return DLIO_EWS;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(ws)) & 15) return DLIO_EUNSUP;
    DlioProfScope prof35(1, s, 2.0 * d.N * (double)d.OH * d.OW * d.Cout * (double)d.Cin * 15,
                         4.0 * d.N * ((double)d.Cin * d.H * d.W + (double)d.Cout * d.OH * d.OW));
    return run_3x5s2(x, dy, dw, reinterpret_cast<float*>(ws), d, q35, accumulate, s, amax_x, amax_dy);
  }
  if (ws_bytes < p3.ws_bytes) return DLIO_EWS;
  const double flops = 2.0 * d.N * (double)d.OH * d.OW * d.Cout * (double)d.Cin * 9;
  const double bytes = 4.0 * d.N * ((double)d.Cin * d.H * d.W + (double)d.Cout * d.OH * d.OW);
  DlioProfScope prof(4, s, flops, bytes);
  float* wsp = reinterpret_cast<float*>(ws);
  int rc = dlio_wgrad3_launch(x, dy, wsp, d, p3, 4, s, amax_x, amax_dy);
  if (rc) return rc;
  const int64_t n = (int64_t)d.Cout * d.Cin * 9;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)cdiv64(n, 64)), dim3(256), 0, s, wsp, dw, n, p3.splits, accumulate);
  return dlio_check_launch();
}

extern "C" int dlio_conv2d_wgrad(const float* x, const float* dy, float* dw,
                                 const float* in_mean, const float* in_scale,
                                 const float* in_shift, void* ws, size_t ws_bytes, int accumulate,
                                 const DlioConvDesc* dp, dlio_stream_t stream) {
  if (!x || !dy || !dw || !dp || !ws) return DLIO_EINVAL;
  const DlioConvDesc& d = *dp;
  if (in_scale && (!in_mean || !in_shift)) return DLIO_EINVAL;
  WgPlan p;
  if (!make_plan(d, p)) return DLIO_EUNSUP;
  if (ws_bytes < p.ws_bytes) return DLIO_EWS;
  // the kernels address both operands with 32-bit buffer offsets
  if ((size_t)d.N * d.in_ctot * d.H * d.W * 4 >= 0xffffff00ull ||
      (size_t)d.N * d.out_ctot * d.OH * d.OW * 4 >= 0xffffff00ull)
    return DLIO_EUNSUP;
  p.accumulate = accumulate;
  hipStream_t s = as_stream(stream);
  float* wsp = reinterpret_cast<float*>(ws);
  const double flops = 2.0 * d.N * (double)d.OH * d.OW * d.Cout * (double)d.Cin * d.KH * d.KW;
  const double bytes = 4.0 * d.N * ((double)d.Cin * d.H * d.W + (double)d.Cout * d.OH * d.OW);
  // profiler kinds: 4 = 3x3 stride-1 (dY-direct kernel, split-bf16 MFMAs), 5 = 1x1 (HBM-bound direct
  // kernel), 1 = everything else (stems, strided layers: staged fp32-MFMA kernel)
  const int pkind = (d.KH == 3 && d.KW == 3 && d.SH == 1 && d.SW == 1) ? 4
                    : (d.KH == 1 && d.KW == 1 && d.SH == 1 && d.SW == 1) ? 5 : 1;
  dlio_prof_begin(pkind, s, flops, bytes);
  int rc = DLIO_EUNSUP;
  DlioWgrad3Plan p3;
  if (!in_scale && dlio_wgrad3_plan(d, 4, p3) && ws_bytes >= p3.ws_bytes &&
      ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy)) & 15) == 0) {
    rc = dlio_wgrad3_launch(x, dy, wsp, d, p3, 4, s);
    if (!rc) {
      const int64_t n = (int64_t)d.Cout * d.Cin * 9;
      hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)cdiv64(n, 64)), dim3(256), 0, s, wsp, dw, n, p3.splits,
                         accumulate);
      rc = dlio_check_launch();
    }
    dlio_prof_end(pkind, s);
    return rc;
  }
  WgS2Plan qs2;
  if (!in_scale && make_plan_s2_taps(d, qs2) && ws_bytes >= qs2.ws_bytes && (reinterpret_cast<uintptr_t>(ws) & 15) == 0) {
    const size_t ncc = (size_t)d.Cout * d.Cin;
    float* dyp = wsp;                                    // first: shifted reads in front of image 0 land in here
    float* img = dyp + qs2.dyp_floats;
    float* t9 = img + 6 * qs2.img_floats;
    float* slabs = t9 + ((9 * ncc + 3) & ~(size_t)3);
    const int64_t po = (int64_t)d.N * d.Cout, pi = (int64_t)d.N * d.Cin;
    hipLaunchKernelGGL(s2_pad_dy_kernel, dim3(ew_grid(po * (qs2.S / 4), 256)), dim3(256), 0, s, dy, dyp, po, d.Cout, d.out_ctot,
                       d.out_coff, d.OH, d.OW, qs2.PW, qs2.S);
    hipLaunchKernelGGL(s2_phase_split_kernel, dim3(ew_grid(pi * (qs2.S / 4), 256)), dim3(256), 0, s, x, img, pi, d.Cin,
                       d.in_ctot, d.in_coff, d.H, d.W, d.OH, qs2.PW, qs2.S, (int64_t)qs2.img_floats);
    rc = dlio_check_launch();
    if (!rc && qs2.gemm) {
      constexpr int lds = 2 * 3 * GT * GLDK * (int)sizeof(__bf16);
      dlio_set_max_lds(reinterpret_cast<const void*>(&gemm_nt_bx3_kernel), lds);
      GemmTaps taps;
      for (int tap = 0; tap < 9; ++tap) {
        const int kh = tap / 3, kw = tap - 3 * kh;
        const int im = (kh == 1 ? 0 : 3) + (kw == 1 ? 0 : kw == 0 ? 2 : 1);
        taps.boff[tap] = (long long)im * (long long)qs2.img_floats + (kh == 0 ? -qs2.PW : 0);
      }
      const int tiles_n = d.Cin / GT, tiles_m = d.Cout / GT;
      hipLaunchKernelGGL(gemm_nt_bx3_kernel, dim3((unsigned)(tiles_m * tiles_n * qs2.gsplits), 9), dim3(256), lds, s, dyp, img,
                         slabs, d.Cout, d.Cin, qs2.S, d.N, tiles_n, qs2.gsplits, taps);
      hipLaunchKernelGGL(wgrad_reduce_taps_kernel, dim3((unsigned)cdiv64((int64_t)ncc * 9, 256)), dim3(256), 0, s, slabs, dw,
                         (int64_t)ncc, qs2.gsplits, accumulate);
      rc = dlio_check_launch();
      dlio_prof_end(pkind, s);
      return rc;
    }
    for (int tap = 0; tap < 9 && !rc; ++tap) {
      const int kh = tap / 3, kw = tap - 3 * kh;
      const int im = (kh == 1 ? 0 : 3) + (kw == 1 ? 0 : kw == 0 ? 2 : 1);
      const float* xi = img + (size_t)im * qs2.img_floats + (kh == 0 ? -qs2.PW : 0);
      float* out = t9 + (size_t)tap * ncc;
      const Wg1Plan& q = qs2.q;
      if (q.mr == 1 && q.nt == 1) rc = launch_1x1<1, 1>(xi, dyp, out, slabs, qs2.sub, q, 0, s);
      else if (q.mr == 1) rc = launch_1x1<1, 2>(xi, dyp, out, slabs, qs2.sub, q, 0, s);
      else if (q.nt == 1) rc = launch_1x1<2, 1>(xi, dyp, out, slabs, qs2.sub, q, 0, s);
      else rc = launch_1x1<2, 2>(xi, dyp, out, slabs, qs2.sub, q, 0, s);
    }
    if (!rc) {
      hipLaunchKernelGGL(wgrad_scatter_taps_kernel, dim3((unsigned)cdiv64((int64_t)ncc * 9, 256)), dim3(256), 0, s, t9, dw,
                         (int64_t)ncc, accumulate);
      rc = dlio_check_launch();
    }
    dlio_prof_end(pkind, s);
    return rc;
  }
  Wg35Plan q35;
  if (!in_scale && make_plan_3x5s2(d, q35) && ws_bytes >= q35.ws_bytes &&
      ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(ws)) & 15) == 0) {
    rc = run_3x5s2(x, dy, dw, wsp, d, q35, accumulate, s, nullptr, nullptr);
    dlio_prof_end(pkind, s);
    return rc;
  }
  Wg1Plan q;
  if (make_plan_1x1(d, q) && ws_bytes >= q.ws_bytes &&
      ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy)) & 15) == 0) {
    if (q.mr == 1 && q.nt == 1) rc = launch_1x1<1, 1>(x, dy, dw, wsp, d, q, accumulate, s, in_mean, in_scale, in_shift);
    else if (q.mr == 1) rc = launch_1x1<1, 2>(x, dy, dw, wsp, d, q, accumulate, s, in_mean, in_scale, in_shift);
    else if (q.nt == 1) rc = launch_1x1<2, 1>(x, dy, dw, wsp, d, q, accumulate, s, in_mean, in_scale, in_shift);
    else rc = launch_1x1<2, 2>(x, dy, dw, wsp, d, q, accumulate, s, in_mean, in_scale, in_shift);
    dlio_prof_end(pkind, s);
    return rc;
  }
#define WG_CASE(kh, kw, sh, sw, nt)                                                \
  if (d.KH == kh && d.KW == kw && d.SH == sh && d.SW == sw)                        \
    rc = launch<kh, kw, sh, sw, nt>(x, dy, dw, in_mean, in_scale, in_shift, wsp, d, p, s);
  WG_CASE(1, 1, 1, 1, 2)
  else if (d.KH == 3 && d.KW == 3 && d.SH == 1 && d.SW == 1 && p.nt == 4)
    rc = launch<3, 3, 1, 1, 4>(x, dy, dw, in_mean, in_scale, in_shift, wsp, d, p, s);
  else WG_CASE(3, 3, 1, 1, 5)
  else WG_CASE(3, 5, 1, 2, 5)
  else WG_CASE(3, 5, 1, 1, 5)
  else WG_CASE(5, 7, 1, 2, 5)
  else WG_CASE(5, 7, 1, 1, 5)
  else WG_CASE(3, 3, 2, 2, 5)
  else WG_CASE(3, 3, 1, 2, 5)
  else WG_CASE(1, 1, 1, 2, 1)
  else WG_CASE(1, 1, 2, 2, 1)
#undef WG_CASE
  dlio_prof_end(pkind, s);
  return rc;
}

// timing probes compiled into this file (bit 0: DLIO_SPLIT_Q0, bit 2: W1_COAL_PROBE); 0 in the product build, checked at load (dlio_build_probes)
int dlio_probe_wgrad() { return ((DLIO_SPLIT_Q0) != 0 ? 1 : 0) | ((W1_COAL_PROBE) != 0 ? 4 : 0); }
