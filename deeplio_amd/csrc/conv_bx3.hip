// 3x3 stride-1 convolution (forward, and data gradient with the tap-reversed layout) with fp32
// accuracy on the bf16 matrix cores: every fp32 operand is split into three bf16 pieces
// x = hi + mid + lo (8+8+8 mantissa bits) and a*b is formed from the six products
// hi*hi, hi*mid, mid*hi, hi*lo, mid*mid, lo*hi -- each a v_mfma_f32_32x32x16_bf16 with exact
// products and fp32 accumulation (error measured in tools/micro/bf16x3.hip: 4.8e-7 of the tile scale,
// the fp32 MFMA's own 5.5e-7).  Per 16-channel chunk and tap that is 6 MFMAs of 32 cycles instead of
// 8 v_mfma_f32_32x32x2_f32 of 64 cycles.
//
// Workgroup: 4 waves = 4 output rows x (32*TWN) columns x (32*MR) output channels.  The patch of a
// 16-channel chunk is split ONCE while it is staged into LDS (three bf16 planes, [plane][pos][16 k]:
// a lane's B fragment -- 8 consecutive channels of one position -- is one conflict-free
// ds_read_b128 per plane); weights are pre-split by dlio_conv3x3_bx3_prep into
// [tap][chunk][plane][n][16 k], a lane's A fragment is one 16-byte load per plane, fetched one tap
// ahead.  Each wave keeps MR x TWN accumulator tiles, so an A fragment is used for TWN pixel blocks
// and a B fragment for MR channel tiles.
//
// Replaces nn.Conv2d forward / input-gradient for the 3x3 stride-1 layers (pointseg_modules.py:100-106
// expand3x3, resnet.py BasicBlock, base_net.py:55-71 conv3_1 / conv4_1 / conv5_1).
#include "common.h"
#include <type_traits>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// timing ablations of the 3x3 kernel (tools/bx3_ablate.py builds variant libraries; results are WRONG with any
// bit set): 1 no split + LDS store, 2 no patch loads, 4 no weight-fragment loads, 8 no LDS fragment reads (3x3 global-
// fragment kernel); 16 no weight-fragment loads in the 1x1 kernel
#ifndef BX3_ABLATE
#define BX3_ABLATE 0
#endif
#ifndef BX3_STEM_PF
#define BX3_STEM_PF 4      // weight-fragment prefetch distance (taps) of the 15-tap stem: 12 MFMAs per tap cover less latency
#endif
typedef _Float16 pc_f16x8 __attribute__((ext_vector_type(8)));
template <class T>
__device__ __forceinline__ void opaque(T& v) { asm volatile("" : "=v"(v)); }

__device__ __forceinline__ void split3(float x, __bf16& h, __bf16& m, __bf16& l) {
  h = (__bf16)x;
  const float r = x - (float)h;
  m = (__bf16)r;
  l = (__bf16)(r - (float)m);
}

// w [Cout][Cin][taps] -> wt [taps][KC][3][Nn][16] bf16; mode 0: k = ci, n = co; mode 1: k = co, n = ci, taps reversed
__global__ void prep_bx3_kernel(const float* __restrict__ w, __bf16* __restrict__ wt, int Cout, int Cin, int taps,
                                int mode) {
  const int K = mode == 0 ? Cin : Cout, Nn = mode == 0 ? Cout : Cin;
  const int KC = (K + 15) >> 4;
  const int64_t total = (int64_t)taps * KC * Nn * 16;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int kk = (int)(i & 15);
    int64_t t = i >> 4;
    const int n = (int)(t % Nn); t /= Nn;
    const int kc = (int)(t % KC);
    const int tap = (int)(t / KC);
    const int k = kc * 16 + kk;
    float v = 0.f;
    if (k < K) {
      if (mode == 0) v = w[((int64_t)n * Cin + k) * taps + tap];
      else v = w[((int64_t)k * Cin + n) * taps + (taps - 1 - tap)];
    }
    __bf16 h, m, l;
    split3(v, h, m, l);
    const int64_t base = (((int64_t)tap * KC + kc) * 3) * Nn * 16 + (int64_t)n * 16 + kk;
    wt[base] = h;
    wt[base + (int64_t)Nn * 16] = m;
    wt[base + (int64_t)2 * Nn * 16] = l;
  }
}

// all split-bf16 weight layouts of a model in ONE launch; item.start / total count the (tap, chunk, n, k)
// elements of the concatenated index space (one element = three bf16 planes)
// (round 6: a thread builds 8 consecutive k of one (tap, chunk, n) -- one search of the item table and three 16-byte stores
//  per 8 elements instead of a search and three 2-byte stores per element: 33 -> ~10 us at the head of every step)
typedef __bf16 prep_bf16x8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(256) void prep_bx3_batched_kernel(const DlioPrepItem* __restrict__ items, int n_items, int64_t total) {
  const int64_t total8 = total >> 3;                // (every item's index space is a multiple of 16)
  for (int64_t i8 = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i8 < total8; i8 += (int64_t)gridDim.x * blockDim.x) {
    const int64_t i = i8 << 3;
    int lo = 0, hi = n_items - 1;                 // last item with start <= i
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (items[mid].start <= i) lo = mid; else hi = mid - 1;
    }
    const DlioPrepItem it = items[lo];
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
    float v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = k0 + q < K ? src[q * ks] : 0.f;
    prep_bf16x8 h, m, l;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      __bf16 a, b, c;
      split3(v[q], a, b, c);
      h[q] = a; m[q] = b; l[q] = c;
    }
    __bf16* wt = reinterpret_cast<__bf16*>(it.wt);
    const int64_t base = (((int64_t)tap * KC + kc) * 3) * Nn * 16 + (int64_t)n * 16 + kk;
    *reinterpret_cast<prep_bf16x8*>(wt + base) = h;
    *reinterpret_cast<prep_bf16x8*>(wt + base + (int64_t)Nn * 16) = m;
    *reinterpret_cast<prep_bf16x8*>(wt + base + (int64_t)2 * Nn * 16) = l;
  }
}

// KH x KW taps, row stride 1, column stride SW (3x3 / 1: Fire expand3x3 & co; 3x5 / 2: the PointSeg stem,
// pointseg_net.py:18-20): output column c of a tile reads patch columns SW * c + kw.
// H2 (two-piece fp16 split, as conv3x3_bx3_pc_kernel<MR, true>): x as two fp16 pieces of x 2^k (k from *amax_x, left by the
// producer of x), weights from dlio_conv_h2_prep: two planes through LDS, three v_mfma_f32_32x32x16_f16 per product; the
// stores multiply by the two inverse scales.  FlowNet conv2-6 / ResNet stage heads and the phases of their data gradients.
#ifndef BX3_STEM_OCC
#define BX3_STEM_OCC 2      // workgroups per CU the three-piece 15-tap stem is compiled for (3 needs BX3_STEM_PF <= 1: 166 VGPRs)
#endif
template <int MR, int TWN, int KH = 3, int KW = 3, int SW = 1, int SH = 1, bool H2 = false>
__global__ __launch_bounds__(256, (KW == 5 && !H2) ? BX3_STEM_OCC : 2) void conv3x3_bx3_kernel(
    const float* __restrict__ x, const __bf16* __restrict__ wt, const float* __restrict__ bias,
    const float* residual, float* y, DlioConvDesc d, int tiles_w, int tiles_h, int co_tiles, int patch_at,
    int vec_out, const float* __restrict__ amax_x = nullptr) {
  constexpr int NPL = H2 ? 2 : 3;                        // 16-bit planes per operand
  constexpr int TH = 4, TW = 32 * TWN, NT = KH * KW;
  constexpr int PR = SH * (TH - 1) + KH, PC = SW * (TW - 1) + KW, NPOSP = PR * PC;
  constexpr int NPOS = (NPOSP + 255) / 256;              // patch positions per thread
  constexpr int PLANE = NPOSP * 16;                      // bf16 per plane
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __bf16* smem = reinterpret_cast<__bf16*>(smem_raw);    // [2 buffers][NPL planes][NPOSP][16]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
  int bid = xcd_block_index();
  const int cot = bid % co_tiles; bid /= co_tiles;
  const int tw = bid % tiles_w; bid /= tiles_w;
  const int th = bid % tiles_h;
  const int n = bid / tiles_h;
  const int co0 = cot * 32 * MR, oh0 = th * TH, ow0 = tw * TW;
  const int Cin = d.Cin, Cout = d.Cout, HW = d.H * d.W;
  const int KC = (Cin + 15) >> 4;
  float xs = 1.f, isc = 1.f;                             // H2: 2^k of the operand, 2^-k 2^-j
  if constexpr (H2) {
    const float am = amax_x[0];
    xs = (am > 0.f && am < 3.0e38f) ? exp2f(floorf(log2f(16384.f / am))) : 1.f;
    isc = (1.f / xs) * reinterpret_cast<const float*>(wt)[(size_t)NT * KC * Cout * 16];
  }

  // ---- staging: a thread owns NPOS patch positions for all 16 channels of a chunk
  bool pval[NPOS];
  int poff[NPOS];
#pragma unroll
  for (int j = 0; j < NPOS; ++j) {
    const int pos = tid + j * 256;
    const int r = pos / PC, c = pos - r * PC;
    const int ih = oh0 * SH - d.PH + r, iw = ow0 * SW - d.PW + c;
    pval[j] = pos < NPOSP && ih >= 0 && ih < d.H && iw >= 0 && iw < d.W;
    poff[j] = pval[j] ? ih * d.W + iw : 0;
  }
  const float* xn = x + ((size_t)n * d.in_ctot + d.in_coff) * HW;
  float reg[NPOS][16];
  auto load_chunk = [&](int kc) {
    if constexpr (BX3_ABLATE & 2) { if (kc > 0) return; }
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      const int ci = kc * 16 + c;
      const float* xc = xn + (size_t)min(ci, Cin - 1) * HW;
#pragma unroll
      for (int j = 0; j < NPOS; ++j) reg[j][c] = xc[poff[j]];
    }
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      const bool cv = kc * 16 + c < Cin;
#pragma unroll
      for (int j = 0; j < NPOS; ++j) reg[j][c] = (cv && pval[j]) ? reg[j][c] : 0.f;
    }
  };
  // column stride 2: the even and the odd patch columns of a row are stored apart, so that the 32 lanes of a B-fragment
  // read (columns 2 * lane + kw) are 32 consecutive positions again -- conflict-free like the stride-1 layout
  constexpr int PCH = (PC + 1) / 2;
  auto store_chunk = [&](__bf16* buf, int kcs) {
#pragma unroll
    for (int j = 0; j < NPOS; ++j) {
      int pos = tid + j * 256;
      if (pos < NPOSP) {
        if constexpr (SW == 2) {
          const int r = pos / PC, c = pos - r * PC;
          pos = r * PC + (c & 1) * PCH + (c >> 1);
        }
        bf16x8 ph[2], pm[2], pl[2];
#pragma unroll
        for (int c = 0; c < 16; ++c) {
          if constexpr (H2) {
            const float xv = reg[j][c] * xs;
            const _Float16 hh = (_Float16)xv;
            const _Float16 ll = (_Float16)(xv - (float)hh);
            ph[c >> 3][c & 7] = __builtin_bit_cast(__bf16, hh); pm[c >> 3][c & 7] = __builtin_bit_cast(__bf16, ll);
          } else {
            __bf16 h, m, l;
            split3(reg[j][c], h, m, l);
            ph[c >> 3][c & 7] = h; pm[c >> 3][c & 7] = m; pl[c >> 3][c & 7] = l;
          }
        }
        bf16x8* dst = reinterpret_cast<bf16x8*>(buf + pos * 16);
        dst[0] = ph[0]; dst[1] = ph[1];
        dst = reinterpret_cast<bf16x8*>(buf + PLANE + pos * 16);
        dst[0] = pm[0]; dst[1] = pm[1];
        if constexpr (!H2) {
          dst = reinterpret_cast<bf16x8*>(buf + 2 * PLANE + pos * 16);
          dst[0] = pl[0]; dst[1] = pl[1];
        }
      }
    }
  };

  // ---- weight fragments: [tap][kc][plane][Nn][16], lane reads 8 k of row n = co (clamped) per plane
  int nrow[MR];
#pragma unroll
  for (int m = 0; m < MR; ++m) nrow[m] = min(co0 + 32 * m + l31, Cout - 1);
  const size_t wplane = (size_t)Cout * 16;
  auto load_a = [&](int tap, int kc, bf16x8 (&a)[MR][NPL]) {
    const __bf16* base = wt + (((size_t)tap * KC + kc) * NPL) * wplane + 8 * half;
#pragma unroll
    for (int m = 0; m < MR; ++m)
#pragma unroll
      for (int p = 0; p < NPL; ++p) {
        if constexpr (BX3_ABLATE & 4) opaque(a[m][p]);
        else a[m][p] = *reinterpret_cast<const bf16x8*>(base + p * wplane + (size_t)nrow[m] * 16);
      }
  };

  f32x16 acc[MR][TWN];
#pragma unroll
  for (int m = 0; m < MR; ++m)
#pragma unroll
    for (int t = 0; t < TWN; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][t][r] = 0.f;

  // weight fragments run PF taps ahead of their MFMAs (a tap is 6*MR*TWN MFMAs = 0.2-0.6 us of cover
  // per wave; one tap ahead does not hide an L2 miss when only a few waves share the SIMD)
  constexpr int PF = (KW == 5 ? BX3_STEM_PF : 2), RING = PF + 1;
  auto compute = [&](const __bf16* buf, int kc, bool prefetch) {
    bf16x8 a[RING][MR][NPL];
    if (patch_at < 0 && prefetch) load_chunk(kc + 1);
#pragma unroll
    for (int p = 0; p < PF; ++p) load_a(p, kc, a[p]);
#pragma unroll
    for (int tap = 0; tap < NT; ++tap) {
      const int kh = tap / KW, kw = tap - KW * kh;
      if (tap + PF < NT) load_a(tap + PF, kc, a[(tap + PF) % RING]);
      // the next chunk's patch loads go out AFTER the last weight-fragment load of this chunk: loads
      // return in order, and a weight fragment queued behind 16-32 patch loads stalls its MFMAs for
      // a full HBM round trip
      // (keep the test per tap: the uniform branches also keep each tap's loads and MFMAs together
      //  -- with one hoisted test, or sched_barrier(0) fences instead, the compiler's own order is
      //  15-25 % slower on the long-K layers: 64->256 data gradient 193 -> 237-242 us)
      if (tap == patch_at && prefetch) load_chunk(kc + 1);
      bf16x8 b[TWN][NPL];
#pragma unroll
      for (int t = 0; t < TWN; ++t) {
        const int pos = SW == 2 ? (wave * SH + kh) * PC + (kw & 1) * PCH + (32 * t + l31) + (kw >> 1)
                                : (wave * SH + kh) * PC + 32 * t + l31 + kw;
#pragma unroll
        for (int p = 0; p < NPL; ++p) {
          if constexpr (BX3_ABLATE & 8) opaque(b[t][p]);
          else b[t][p] = *reinterpret_cast<const bf16x8*>(buf + p * PLANE + pos * 16 + 8 * half);
        }
      }
      const auto& aa = a[tap % RING];
      if constexpr (H2) {
        constexpr int HA[3] = {1, 0, 0}, HB[3] = {0, 1, 0};              // (lo, hi) (hi, lo) (hi, hi): smallest first
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
          for (int m = 0; m < MR; ++m)
#pragma unroll
            for (int t = 0; t < TWN; ++t)
              acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(pc_f16x8, aa[m][HA[q]]),
                                                                 __builtin_bit_cast(pc_f16x8, b[t][HB[q]]), acc[m][t], 0, 0, 0);
      } else {
        // six products, smallest first; consecutive MFMAs go to different accumulators
        constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
        for (int q = DLIO_SPLIT_Q0; q < 6; ++q)
#pragma unroll
          for (int m = 0; m < MR; ++m)
#pragma unroll
            for (int t = 0; t < TWN; ++t)
              acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aa[m][PA[q] % NPL], b[t][PB[q] % NPL], acc[m][t], 0, 0, 0);
      }
    }
  };

  // ---- chunk loop, LDS double buffered, one barrier per chunk
  load_chunk(0);
  store_chunk(smem, 0);
  __syncthreads();
  for (int kc = 0; kc < KC; ++kc) {
    const __bf16* cur = smem + (size_t)(kc & 1) * NPL * PLANE;
    __bf16* nxt = smem + (size_t)((kc + 1) & 1) * NPL * PLANE;
    compute(cur, kc, kc + 1 < KC);
    if constexpr (!(BX3_ABLATE & 1)) { if (kc + 1 < KC) store_chunk(nxt, kc + 1); }
    __syncthreads();
  }

  // ---- epilogue: D tile col = pixel (lane & 31), row = (r & 3) + 8 (r >> 2) + 4 half
  const int oh = oh0 + wave;
  const bool row_ok = oh < d.OH;
  if (!row_ok) return;
  const size_t ohw = (size_t)d.OH * d.OW;
  // Stores through LDS (vec_out): straight from the accumulators a store instruction writes two 128-byte pieces (32
  // pixels of channel c and of channel c + 4) as dwords; transposed through the wave's own LDS region ([channel][TW + 8
  // floats]: the patch buffers are free, and the row stride puts the two halves on different banks) a lane stores one
  // float4 and an instruction covers 4 / TWN channel rows of TW contiguous pixels -- a quarter of the store instructions,
  // 16-byte accesses, bias and residual applied on the way out.
  if (vec_out) {
    constexpr int TWP = TWN == 1 ? TW + 4 : TW + 8;      // (TW + 4: the 64-channel x 32-pixel tile has to fit its 39 KB of patch buffers)
    float* wbuf = reinterpret_cast<float*>(smem_raw) + wave * (32 * MR * TWP);
#pragma unroll
    for (int m = 0; m < MR; ++m)
#pragma unroll
      for (int t = 0; t < TWN; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          wbuf[(32 * m + (r & 3) + 8 * (r >> 2) + 4 * half) * TWP + 32 * t + l31] = H2 ? acc[m][t][r] * isc : acc[m][t][r];
    // (each wave reads back only what it wrote itself: no barrier)
    constexpr int Q = TW / 4;                   // float4 per channel row
    float* yrow = y + ((size_t)n * d.out_ctot + d.out_coff) * ohw + (size_t)oh * d.OW + ow0;
    const float* rrow = residual ? residual + ((size_t)n * d.res_ctot + d.res_coff) * ohw + (size_t)oh * d.OW + ow0 : nullptr;
#pragma unroll
    for (int i = 0; i < (32 * MR * Q) / 64; ++i) {
      const int idx = i * 64 + lane, cl = idx / Q, q = idx - cl * Q;
      const int co = co0 + cl;
      if (co >= Cout || ow0 + 4 * q >= d.OW) continue;
      float4 v = *reinterpret_cast<const float4*>(wbuf + cl * TWP + 4 * q);
      const float bv = bias ? bias[co] : 0.f;
      v.x += bv; v.y += bv; v.z += bv; v.w += bv;
      if (rrow) {
        const float4 rv = *reinterpret_cast<const float4*>(rrow + (size_t)co * ohw + 4 * q);
        v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
      }
      st4<4>(yrow + (size_t)co * ohw + 4 * q, v);
    }
    return;
  }
#pragma unroll
  for (int m = 0; m < MR; ++m)
#pragma unroll
    for (int t = 0; t < TWN; ++t) {
      const int ow = ow0 + 32 * t + l31;
      if (ow >= d.OW) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + 32 * m + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (co >= Cout) continue;
        float v = H2 ? acc[m][t][r] * isc : acc[m][t][r];
        if (bias) v += bias[co];
        const size_t pix = (size_t)oh * d.OW + ow;
        if (residual) v += residual[((size_t)n * d.res_ctot + d.res_coff + co) * ohw + pix];
        y[((size_t)n * d.out_ctot + d.out_coff + co) * ohw + pix] = v;
      }
    }
}

// ---- 3x3 stride 1, weight fragments through LDS --------------------------------------------------
// conv3x3_bx3_kernel's four waves own four rows of the SAME 32*MR output channels: each of them fetched the same
// 16-byte weight fragments from global memory (6 per tap and wave for MR = 2), and with five streams sharing the chip
// those loads were what the step waited for (timing ablation with the fragment loads removed: 26.2 -> 23.3 ms per
// training step).  Here a workgroup fetches the fragments of three taps ONCE (18 KB, 4-5 sixteen-byte loads per
// thread), stores them into a two-slot LDS ring and every wave reads them with ds_read_b128 -- a quarter of the
// global weight-fragment traffic.  The ring takes the place of the second patch buffer (LDS stays at 75 KB per
// workgroup = two per CU): the patch is single-buffered, its global loads still run under the MFMAs of the
// previous chunk, only the split + LDS store wait for the end-of-chunk barrier.  4 barriers per chunk.
// Forward / data gradient of "same" 3x3 layers without input transform or statistics epilogue.
template <int MR, int TWN>
__global__ __launch_bounds__(256, 2) void conv3x3_bx3_alds_kernel(
    const float* __restrict__ x, const __bf16* __restrict__ wt, const float* __restrict__ bias,
    const float* residual, float* y, DlioConvDesc d, int tiles_w, int tiles_h, int co_tiles, int vec_out, int ksplit,
    float* __restrict__ slab, const float* __restrict__ x1, const __bf16* __restrict__ wt1, int C1) {
  constexpr int TH = 4, TW = 32 * TWN, PR = TH + 2, PC = TW + 2, NPOSP = PR * PC;
  constexpr int NPOS = (NPOSP + 255) / 256;
  constexpr int PLANE = NPOSP * 16;                      // bf16 per patch plane
  constexpr int TG = 3;                                  // taps per weight group
  constexpr int AROWS = 32 * MR;
  constexpr int AGRP = TG * 3 * AROWS * 16;              // bf16 per ring slot: [tap][plane][row][16 k]
  constexpr int APIECES = AGRP / 8;                      // 16-byte pieces per group
  constexpr int ALD = (APIECES + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __bf16* patch = reinterpret_cast<__bf16*>(smem_raw);   // [3 planes][NPOSP][16]
  __bf16* ring = patch + 3 * PLANE;                      // [2 slots][AGRP]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
  int bid = xcd_block_index();
  // K split over workgroups (slab != nullptr: long channel loops on small feature maps -- blk4 / blk5 data gradients fill
  // half of the chip's workgroup slots or less): this workgroup takes chunks [kc_lo, kc_hi) and writes its partial tile
  // to slab[ks] ([ks][n][co][pixel] fp32); slab_sum_kernel adds the slices in a fixed order + bias + residual
  const int ks = slab ? bid % ksplit : 0;
  if (slab) bid /= ksplit;
  const int cot = bid % co_tiles; bid /= co_tiles;
  const int tw = bid % tiles_w; bid /= tiles_w;
  const int th = bid % tiles_h;
  const int n = bid / tiles_h;
  const int co0 = cot * 32 * MR, oh0 = th * TH, ow0 = tw * TW;
  const int Cin = d.Cin, Cout = d.Cout, HW = d.H * d.W;
  const int KC = (Cin + 15) >> 4;
  // second input (x1 [N][C1][H][W], weights wt1 [1][KC1][3][Cout][16]: a 1x1 layer that adds to the same output -- the
  // expand1x1 half of a Fire block's data gradient): KC1 more chunks that use the CENTRE tap only
  const int KC1 = x1 ? (C1 + 15) >> 4 : 0, KCT = KC + KC1;
  const int kc_lo = slab ? (int)((int64_t)KCT * ks / ksplit) : 0, kc_hi = slab ? (int)((int64_t)KCT * (ks + 1) / ksplit) : KCT;

  // ---- patch staging (as conv3x3_bx3_kernel)
  bool pval[NPOS];
  int poff[NPOS];
#pragma unroll
  for (int j = 0; j < NPOS; ++j) {
    const int pos = tid + j * 256;
    const int r = pos / PC, c = pos - r * PC;
    const int ih = oh0 - d.PH + r, iw = ow0 - d.PW + c;
    pval[j] = pos < NPOSP && ih >= 0 && ih < d.H && iw >= 0 && iw < d.W;
    poff[j] = pval[j] ? ih * d.W + iw : 0;
  }
  const float* xn = x + ((size_t)n * d.in_ctot + d.in_coff) * HW;
  const float* xn1 = x1 ? x1 + (size_t)n * C1 * HW : nullptr;
  float reg[NPOS][16];
  auto load_patch = [&](int kc) {
    const bool second = kc >= KC;
    const float* xb = second ? xn1 : xn;
    const int cb = second ? (kc - KC) * 16 : kc * 16, cmax = (second ? C1 : Cin) - 1;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      const float* xc = xb + (size_t)min(cb + c, cmax) * HW;
#pragma unroll
      for (int j = 0; j < NPOS; ++j) reg[j][c] = xc[poff[j]];
    }
  };
  auto store_patch = [&](int kc) {
    const int cend = kc >= KC ? C1 - (kc - KC) * 16 : Cin - kc * 16;       // valid channels of this chunk
#pragma unroll
    for (int j = 0; j < NPOS; ++j) {
      const int pos = tid + j * 256;
      if (pos < NPOSP) {
        bf16x8 ph[2], pm[2], pl[2];
#pragma unroll
        for (int c = 0; c < 16; ++c) {
          const float v = (c < cend && pval[j]) ? reg[j][c] : 0.f;
          __bf16 h, m, l;
          split3(v, h, m, l);
          ph[c >> 3][c & 7] = h; pm[c >> 3][c & 7] = m; pl[c >> 3][c & 7] = l;
        }
        bf16x8* dst = reinterpret_cast<bf16x8*>(patch + pos * 16);
        dst[0] = ph[0]; dst[1] = ph[1];
        dst = reinterpret_cast<bf16x8*>(patch + PLANE + pos * 16);
        dst[0] = pm[0]; dst[1] = pm[1];
        dst = reinterpret_cast<bf16x8*>(patch + 2 * PLANE + pos * 16);
        dst[0] = pl[0]; dst[1] = pl[1];
      }
    }
  };

  // ---- weight groups: piece q = ((tap_l * 3 + plane) * AROWS + row) * 2 + k-half, the same order in LDS
  const size_t wplane = (size_t)Cout * 16;
  const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<__bf16*>(wt), 0, (int)((size_t)9 * KC * 3 * wplane * 2), 0x00020000);
  // straight into LDS (buffer_load_dwordx4 ... lds: the 64 pieces of a wave instruction land at consecutive 16-byte
  // slots): instruction t = (tap_l * 3 + plane) * MR + m covers the 32 rows x 2 k-halves of one (tap, plane, tile), the
  // instructions of a group are dealt round-robin to the four waves
  constexpr int AINS = TG * 3 * MR;
  unsigned awoff[MR];
#pragma unroll
  for (int m = 0; m < MR; ++m) awoff[m] = ((unsigned)min(co0 + m * 32 + (lane >> 1), Cout - 1) * 16u + 8u * (unsigned)(lane & 1)) * 2u;
  const __amdgpu_buffer_rsrc_t wrsrc1 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<__bf16*>(wt1 ? wt1 : wt), 0, (int)((size_t)(KC1 ? KC1 : 1) * 3 * wplane * 2), 0x00020000);
  auto load_agroup = [&](int kc, int g, int slot) {
    const bool second = kc >= KC;            // (one group of ONE tap: the first 3 MR instructions of a group's 9 MR)
#pragma unroll
    for (int i = 0; i < (AINS + 3) / 4; ++i) {
      const int t = i * 4 + wave;
      if (t < AINS && (!second || t < 3 * MR)) {
        const int m = t % MR, tp = t / MR, tap = g * TG + tp / 3, pl = tp - 3 * (tp / 3);
        __bf16* dst = ring + (size_t)slot * AGRP + (size_t)t * 64 * 8;
#if defined(__HIP_DEVICE_COMPILE__)     // (the host pass cannot instantiate the address-space cast)
        if (second)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc1, (__attribute__((address_space(3))) void*)dst, 16, awoff[m],
                                                   (unsigned)(((size_t)(kc - KC) * 3 + pl) * wplane * 2), 0, 0);
        else
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (__attribute__((address_space(3))) void*)dst, 16, awoff[m],
                                                 (unsigned)((((size_t)tap * KC + kc) * 3 + pl) * wplane * 2), 0, 0);
#else
        (void)dst; (void)m; (void)tap; (void)pl;
#endif
      }
    }
  };

  f32x16 acc[MR][TWN];
#pragma unroll
  for (int m = 0; m < MR; ++m)
#pragma unroll
    for (int t = 0; t < TWN; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][t][r] = 0.f;

  auto read_a = [&](const __bf16* slot, int tl, bf16x8 (&a)[MR][3]) {
#pragma unroll
    for (int m = 0; m < MR; ++m)
#pragma unroll
      for (int p = 0; p < 3; ++p)
        a[m][p] = *reinterpret_cast<const bf16x8*>(slot + ((tl * 3 + p) * AROWS + m * 32 + l31) * 16 + 8 * half);
  };
  auto compute_group = [&](const __bf16* slot, int g) {
    bf16x8 a[2][MR][3];
    read_a(slot, 0, a[0]);
#pragma unroll
    for (int tl = 0; tl < TG; ++tl) {
      const int tap = g * TG + tl, kh = tap / 3, kw = tap - 3 * kh;
      if (tl + 1 < TG) read_a(slot, tl + 1, a[(tl + 1) & 1]);
      bf16x8 b[TWN][3];
#pragma unroll
      for (int t = 0; t < TWN; ++t) {
        const int pos = (wave + kh) * PC + 32 * t + l31 + kw;
#pragma unroll
        for (int p = 0; p < 3; ++p)
          b[t][p] = *reinterpret_cast<const bf16x8*>(patch + p * PLANE + pos * 16 + 8 * half);
      }
      const auto& aa = a[tl & 1];
      constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
      for (int q = DLIO_SPLIT_Q0; q < 6; ++q)
#pragma unroll
        for (int m = 0; m < MR; ++m)
#pragma unroll
          for (int t = 0; t < TWN; ++t)
            acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aa[m][PA[q]], b[t][PB[q]], acc[m][t], 0, 0, 0);
    }
  };

  auto compute_centre = [&](const __bf16* slot) {       // the 1x1 chunk: tap (1, 1), fragments at ring position 0
    bf16x8 a[MR][3], b[TWN][3];
    read_a(slot, 0, a);
#pragma unroll
    for (int t = 0; t < TWN; ++t) {
      const int pos = (wave + 1) * PC + 32 * t + l31 + 1;
#pragma unroll
      for (int p = 0; p < 3; ++p) b[t][p] = *reinterpret_cast<const bf16x8*>(patch + p * PLANE + pos * 16 + 8 * half);
    }
    constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
    for (int q = DLIO_SPLIT_Q0; q < 6; ++q)
#pragma unroll
      for (int m = 0; m < MR; ++m)
#pragma unroll
        for (int t = 0; t < TWN; ++t)
          acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m][PA[q]], b[t][PB[q]], acc[m][t], 0, 0, 0);
  };

  // ---- pipeline: weight groups double-slotted, patch single-buffered
  load_agroup(kc_lo, 0, 0);
  load_patch(kc_lo);
  store_patch(kc_lo);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int gi = 0;                                            // running group index -> ring slot
  for (int kc = kc_lo; kc < kc_hi; ++kc) {
    const bool more = kc + 1 < kc_hi;
    const int ngroups = kc >= KC ? 1 : 3;                // (a 1x1 chunk is one group of one tap)
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      if (g >= ngroups) break;
      const bool last = g == ngroups - 1;
      const bool anext = !last || more;
      // the other slot was read last during group gi - 1: every wave is past that barrier
      if (anext) load_agroup(last ? kc + 1 : kc, last ? 0 : g + 1, (gi + 1) & 1);
      if (g == 0 && more) load_patch(kc + 1);
      if (kc >= KC) compute_centre(ring + (size_t)(gi & 1) * AGRP);
      else compute_group(ring + (size_t)(gi & 1) * AGRP, g);
      // the group's pieces have landed; the patch loads issued behind them (group 0) stay in flight
      if (g == 0 && more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPOS * 16) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      ++gi;
    }
    if (more) {
      store_patch(kc + 1);                               // every wave is past the last group's barrier: the patch is free
      __syncthreads();
    }
  }

  // ---- epilogue (conv3x3_bx3_kernel's)
  const int oh = oh0 + wave;
  if (oh >= d.OH) return;
  const size_t ohw = (size_t)d.OH * d.OW;
  if (vec_out) {
    constexpr int TWP = TWN == 1 ? TW + 4 : TW + 8;
    float* wbuf = reinterpret_cast<float*>(smem_raw) + wave * (32 * MR * TWP);
#pragma unroll
    for (int m = 0; m < MR; ++m)
#pragma unroll
      for (int t = 0; t < TWN; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          wbuf[(32 * m + (r & 3) + 8 * (r >> 2) + 4 * half) * TWP + 32 * t + l31] = acc[m][t][r];
    constexpr int Q = TW / 4;
    // (K split: the raw partial tile goes to this slice's slab, bias / residual are added by slab_sum_kernel)
    float* yrow = slab ? slab + (((size_t)ks * d.N + n) * Cout) * ohw + (size_t)oh * d.OW + ow0
                       : y + ((size_t)n * d.out_ctot + d.out_coff) * ohw + (size_t)oh * d.OW + ow0;
    const float* rrow = (residual && !slab) ? residual + ((size_t)n * d.res_ctot + d.res_coff) * ohw + (size_t)oh * d.OW + ow0 : nullptr;
#pragma unroll
    for (int i = 0; i < (32 * MR * Q) / 64; ++i) {
      const int idx = i * 64 + lane, cl = idx / Q, q = idx - cl * Q;
      const int co = co0 + cl;
      if (co >= Cout || ow0 + 4 * q >= d.OW) continue;
      float4 v = *reinterpret_cast<const float4*>(wbuf + cl * TWP + 4 * q);
      const float bv = (bias && !slab) ? bias[co] : 0.f;
      v.x += bv; v.y += bv; v.z += bv; v.w += bv;
      if (rrow) {
        const float4 rv = *reinterpret_cast<const float4*>(rrow + (size_t)co * ohw + 4 * q);
        v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
      }
      st4<4>(yrow + (size_t)co * ohw + 4 * q, v);
    }
    return;
  }
#pragma unroll
  for (int m = 0; m < MR; ++m)
#pragma unroll
    for (int t = 0; t < TWN; ++t) {
      const int ow = ow0 + 32 * t + l31;
      if (ow >= d.OW) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + 32 * m + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (co >= Cout) continue;
        float v = acc[m][t][r];
        if (bias) v += bias[co];
        const size_t pix = (size_t)oh * d.OW + ow;
        if (residual) v += residual[((size_t)n * d.res_ctot + d.res_coff + co) * ohw + pix];
        y[((size_t)n * d.out_ctot + d.out_coff + co) * ohw + pix] = v;
      }
    }
}

// ---- 3x3 stride 1, producer / consumer workgroups ---------------------------------------------------
// conv3x3_bx3_alds_kernel stages a chunk's patch (gather + split: ~350 VALU per thread) BETWEEN the MFMA phases of its four
// waves, and its prologue / epilogue are only covered by the other workgroup of the CU (MFMA busy 50 %).  Here ONE persistent
// workgroup of eight waves per CU splits the roles: waves 0-3 do nothing but LDS reads + MFMAs (a wave = a row of the 4 x 64
// x 32 MR tile), waves 4-7 stage the NEXT chunk -- global loads in group 0, split + LDS store of their two positions in
// groups 1 and 2 -- into the second patch buffer and feed the weight ring (buffer_load ... lds), so VALU staging and MFMA
// issue run side by side on every SIMD, across chunk AND tile boundaries (the producers are one chunk ahead, the first
// chunk of the next tile included).  One barrier per weight group for all eight waves, none extra per chunk.  LDS: 2 patch
// buffers + the two-slot ring = 112 KB.  For the long-K data gradients of fire_blk1-3 (the epilogue is small: S = 16-80
// output channels); tile list as fire_expand's XCD-contiguous order.
// Measured (tools/bench_dgrad3.py, N = 16): 117 / 141 / 153 / 193 us against 127 / 150 / 158 / 203 on the alds kernel, step
// 22.45 vs 22.65 ms.  Timing ablations: the consumers ALONE (no producer work, no memory traffic) take 95 / 102 / 120 / 152
// us = 48-60 % of the bare MFMA time -- one MFMA wave per SIMD, a barrier every three taps, LDS fragment reads behind it:
// the loop that feeds the matrix cores, not the staging, is what is left.  Tried on it, no gain: the barrier moved to the
// middle of the group with a three-slot ring (129 / 147 / 156 / 194 us), a second accumulator set for the 32-channel tile
// (118 / 139); with the LDS fragment reads removed as well the consumers take 81 / 86 / 109 / 139 us -- the MFMA issue
// time itself at the clocks the chip holds under this load.
// H2: the operand as TWO fp16 pieces of x 2^k (2^k from the tensor's largest magnitude, *amax_x, left by the kernel that
// produced x), weights from dlio_conv_h2_prep (two fp16 pieces, { 2^-j, 2^j } behind them): three v_mfma_f32_32x32x16_f16
// per product, two planes through LDS, 7 VALU per value in the split instead of 11; the epilogue multiplies by 2^-k 2^-j.
// Measured and not kept (two-piece format): a whole-chunk weight ring (two slots of nine taps), the ring loads of chunk i + 1
// and the patch loads of chunk i + 2 (second register set) issued at the start of step i, one barrier per chunk -- 90.6 /
// 84.8 / 87.3 / 112.6 us against 92.6-96.1 / 84.9-88.3 / 87.0-87.5 / 112.5-113.4: the chunk time (~3 us) is not a chain of
// exposed load latencies; the step was 0.1 ms slower with it.
template <int MR, bool H2 = false>
__global__ __launch_bounds__(512) void conv3x3_bx3_pc_kernel(
    const float* __restrict__ x, const __bf16* __restrict__ wt, const float* __restrict__ bias, const float* residual, float* y,
    DlioConvDesc d, int tiles_w, int tiles_h, int co_tiles, int total_tiles, const float* __restrict__ amax_x = nullptr) {
  constexpr int NPL = H2 ? 2 : 3;
  constexpr int TWN = 2, TH = 4, TW = 64, PR = TH + 2, PC = TW + 2, NPOSP = PR * PC;
  constexpr int NPOS = (NPOSP + 255) / 256;              // 2 patch positions per producer thread
  constexpr int PLANE = NPOSP * 16;
  constexpr int HPL = NPOSP * 8;                         // one channel half of a plane
  constexpr int PBUF = NPL * PLANE;                      // bf16 per patch buffer
  constexpr int TG = 3;
  constexpr int AROWS = 32 * MR;
  constexpr int AGRP = TG * NPL * AROWS * 16;
  constexpr int AINS = TG * NPL * MR;
  static_assert(NPOS == 2, "two positions per producer thread");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __bf16* patch0 = reinterpret_cast<__bf16*>(smem_raw);  // [2][3 planes][NPOSP][16]
  __bf16* ring = patch0 + 2 * PBUF;                      // [2][AGRP]

  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool producer = wave >= 4;
  const int pw = wave & 3, ptid = tid & 255;
  const int Cin = d.Cin, Cout = d.Cout, HW = d.H * d.W;
  const int KC = (Cin + 15) >> 4;

  // ---- tile list of this workgroup
  const int G = gridDim.x, bx = blockIdx.x & 7, bj = blockIdx.x >> 3, Gx = (G - bx + 7) >> 3;
  const int per = (total_tiles + 7) >> 3;
  const int base = bx * per, cnt = min(per, total_tiles - base);
  struct Cur { int l, n, oh0, ow0, co0; bool ok; };
  auto set_tile = [&](Cur& c) {
    c.ok = c.l < cnt;
    int v = base + (c.ok ? c.l : 0);
    const int cot = v % co_tiles; v /= co_tiles;
    const int tw = v % tiles_w; v /= tiles_w;
    const int th = v % tiles_h;
    c.n = v / tiles_h; c.oh0 = th * TH; c.ow0 = tw * TW; c.co0 = cot * 32 * MR;
  };
  Cur cur;
  cur.l = bj;
  set_tile(cur);
  if (!cur.ok) return;

  const size_t wplane = (size_t)Cout * 16;
  const int nchunks_total = KC;                          // per tile
  float xs = 1.f;                                        // H2: 2^k of the operand
  if constexpr (H2) {
    const float am = amax_x[0];
    xs = (am > 0.f && am < 3.0e38f) ? exp2f(floorf(log2f(16384.f / am))) : 1.f;
  }

  if (producer) {
    // ================================================================ producers
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<__bf16*>(wt), 0, (int)((size_t)9 * KC * NPL * wplane * 2), 0x00020000);
    const int RW = (AINS - pw + 3) / 4;                  // ring instructions of this wave per group
    auto load_agroup = [&](const Cur& c, int kc, int g, int slot) {
#pragma unroll
      for (int i = 0; i < (AINS + 3) / 4; ++i) {
        const int t = i * 4 + pw;
        if (t < AINS) {
          const int m = t % MR, tp = t / MR, tap = g * TG + tp / NPL, pl = tp - NPL * (tp / NPL);
          // LDS image of a (tap, plane, m) block: [half][32 rows][8] -- a 16-byte stride between the lanes of a ds_read_b128
          // lane group (the [row][16] image put lanes l and l + 8 of a group on the same banks: every fragment read 2-way)
          const unsigned awoff = ((unsigned)min(c.co0 + m * 32 + l31, Cout - 1) * 16u + 8u * (unsigned)half) * 2u;
          __bf16* dst = ring + (size_t)slot * AGRP + (size_t)t * 64 * 8;
#if defined(__HIP_DEVICE_COMPILE__)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (__attribute__((address_space(3))) void*)dst, 16, awoff,
                                                   (unsigned)((((size_t)tap * KC + kc) * NPL + pl) * wplane * 2), 0, 0);
#else
          (void)dst; (void)awoff; (void)tap; (void)pl;
#endif
        }
      }
    };
    // The gather: a buffer descriptor over the image's Cin planes, ONE 32-bit lane offset per position (beyond num_records
    // for a position outside the image or the patch: such loads return 0 -- no select per value) and a scalar offset per
    // channel row (64-bit lane addresses + two selects per value were 100 of the producers' ~220 instructions per chunk,
    // and the producers are what the narrow layers wait for: 1100 issue cycles against 430 of MFMAs at blk1)
    float reg[NPOS][16];
    unsigned pvoff[NPOS];
    constexpr unsigned OOB = 0xffffff00u;
    const unsigned HW4 = (unsigned)HW * 4u;               // (Cin HW 4 < 4 GB: bx3_pc_geom_ok)
    __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, 0, 0x00020000);
    auto tile_geometry = [&](const Cur& c) {
#pragma unroll
      for (int j = 0; j < NPOS; ++j) {
        const int pos = ptid + j * 256;
        const int r = pos / PC, cc = pos - r * PC;
        const int ih = c.oh0 - d.PH + r, iw = c.ow0 - d.PW + cc;
        const bool pv = pos < NPOSP && ih >= 0 && ih < d.H && iw >= 0 && iw < d.W;
        pvoff[j] = pv ? (unsigned)(ih * d.W + iw) * 4u : OOB;
      }
      xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x + ((size_t)c.n * d.in_ctot + d.in_coff) * HW), 0,
                                                (int)((unsigned)Cin * HW4), 0x00020000);
    };
    auto load_patch_into = [&](float (&reg)[NPOS][16], int kc) {   // position-major: position 0's sixteen loads are the older ones
      if (kc * 16 + 16 <= Cin) {
#pragma unroll
        for (int j = 0; j < NPOS; ++j)
#pragma unroll
          for (int c = 0; c < 16; ++c)
            reg[j][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrsrc, pvoff[j], (unsigned)(kc * 16 + c) * HW4, 0));
      } else {                                           // the ragged last chunk: rows behind Cin read 0 as well
#pragma unroll
        for (int j = 0; j < NPOS; ++j)
#pragma unroll
          for (int c = 0; c < 16; ++c) {
            const int ch = kc * 16 + c;
            reg[j][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrsrc, ch < Cin ? pvoff[j] : OOB,
                                                                                       (unsigned)min(ch, Cin - 1) * HW4, 0));
          }
      }
    };
    auto load_patch = [&](int kc) { load_patch_into(reg, kc); };
    auto store_pos_from = [&](const float (&reg)[NPOS][16], __bf16* buf, int j) {
      const int pos = ptid + j * 256;
      if (pos < NPOSP) {
        bf16x8 ph[2], pm[2], pl[2];
#pragma unroll
        for (int c = 0; c < 16; ++c) {
          const float v = reg[j][c];
          if constexpr (H2) {
            const float vs = v * xs;
            const _Float16 h = (_Float16)vs;
            const _Float16 l = (_Float16)(vs - (float)h);
            ph[c >> 3][c & 7] = __builtin_bit_cast(__bf16, h); pm[c >> 3][c & 7] = __builtin_bit_cast(__bf16, l);
          } else {
            __bf16 h, m, l;
            split3(v, h, m, l);
            ph[c >> 3][c & 7] = h; pm[c >> 3][c & 7] = m; pl[c >> 3][c & 7] = l;
          }
        }
        // plane image [half][position][8 channels]: 16-byte lane stride for the fragment reads AND for these stores
        *reinterpret_cast<bf16x8*>(buf + pos * 8) = ph[0];
        *reinterpret_cast<bf16x8*>(buf + HPL + pos * 8) = ph[1];
        *reinterpret_cast<bf16x8*>(buf + PLANE + pos * 8) = pm[0];
        *reinterpret_cast<bf16x8*>(buf + PLANE + HPL + pos * 8) = pm[1];
        if constexpr (!H2) {
          *reinterpret_cast<bf16x8*>(buf + 2 * PLANE + pos * 8) = pl[0];
          *reinterpret_cast<bf16x8*>(buf + 2 * PLANE + HPL + pos * 8) = pl[1];
        }
      }
    };
    auto store_pos = [&](__bf16* buf, int, int j) { store_pos_from(reg, buf, j); };
    // prologue: first weight group, first patch
    tile_geometry(cur);
    load_agroup(cur, 0, 0, 0);
    load_patch(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    store_pos(patch0, 0, 0);
    store_pos(patch0, 0, 1);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    int q = 0, gi = 0;
    while (cur.ok) {
      for (int kc = 0; kc < KC; ++kc, ++q) {
        // the chunk the producers stage while the consumers are on (cur, kc)
        Cur nx = cur;
        int nkc = kc + 1;
        if (nkc == KC) { nkc = 0; nx.l += Gx; set_tile(nx); }
        const bool have = nx.ok;
        __bf16* nbuf = patch0 + (size_t)((q + 1) & 1) * PBUF;
        // ---- group 0: ring(g = 1), the next chunk's patch loads
        load_agroup(cur, kc, 1, (gi + 1) & 1);
        if (have) {
          if (nkc == 0) tile_geometry(nx);
          load_patch(nkc);
          asm volatile("s_waitcnt vmcnt(32)" ::: "memory");            // the ring pieces landed, the patch loads fly
        } else {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        asm volatile("s_barrier" ::: "memory");
        ++gi;
        // ---- group 1: ring(g = 2), split + store position 0
        load_agroup(cur, kc, 2, (gi + 1) & 1);
        if (have) {
          if (RW == 5) asm volatile("s_waitcnt vmcnt(21)" ::: "memory");
          else if (RW == 4) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
          else if (RW == 3) asm volatile("s_waitcnt vmcnt(19)" ::: "memory");
          else if (RW == 2) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(17)" ::: "memory");
          store_pos(nbuf, nkc, 0);
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        ++gi;
        // ---- group 2: ring of the next chunk's group 0, split + store position 1
        if (have) {
          load_agroup(nx, nkc, 0, (gi + 1) & 1);
          store_pos(nbuf, nkc, 1);
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        ++gi;
        if (nkc == 0) { cur = nx; }
      }
    }
    return;
  }

  // ================================================================== consumers
  f32x16 acc[MR][TWN];
  auto zero_acc = [&]() {
#pragma unroll
    for (int m = 0; m < MR; ++m)
#pragma unroll
      for (int t = 0; t < TWN; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][t][r] = 0.f;
  };
  zero_acc();
  auto read_a = [&](const __bf16* slot, int tl, bf16x8 (&a)[MR][NPL]) {
#pragma unroll
    for (int m = 0; m < MR; ++m)
#pragma unroll
      for (int p = 0; p < NPL; ++p)
        a[m][p] = *reinterpret_cast<const bf16x8*>(slot + ((tl * NPL + p) * MR + m) * 512 + half * 256 + l31 * 8);
  };
  auto compute_group = [&](const __bf16* patch, const __bf16* slot, int g) {
    bf16x8 a[2][MR][NPL];
    read_a(slot, 0, a[0]);
#pragma unroll
    for (int tl = 0; tl < TG; ++tl) {
      const int tap = g * TG + tl, kh = tap / 3, kw = tap - 3 * kh;
      if (tl + 1 < TG) read_a(slot, tl + 1, a[(tl + 1) & 1]);
      bf16x8 b[TWN][NPL];
#pragma unroll
      for (int t = 0; t < TWN; ++t) {
        const int pos = (wave + kh) * PC + 32 * t + l31 + kw;
#pragma unroll
        for (int p = 0; p < NPL; ++p)
          b[t][p] = *reinterpret_cast<const bf16x8*>(patch + p * PLANE + half * HPL + pos * 8);
      }
      const auto& aa = a[tl & 1];
      if constexpr (H2) {
        constexpr int HA[3] = {1, 0, 0}, HB[3] = {0, 1, 0};            // (lo, hi) (hi, lo) (hi, hi)
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
          for (int m = 0; m < MR; ++m)
#pragma unroll
            for (int t = 0; t < TWN; ++t)
              acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(pc_f16x8, aa[m][HA[q]]),
                                                                 __builtin_bit_cast(pc_f16x8, b[t][HB[q]]), acc[m][t], 0, 0, 0);
      } else {
        constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
        for (int q = DLIO_SPLIT_Q0; q < 6; ++q)
#pragma unroll
          for (int m = 0; m < MR; ++m)
#pragma unroll
            for (int t = 0; t < TWN; ++t)
              acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aa[m][PA[q]], b[t][PB[q]], acc[m][t], 0, 0, 0);
      }
    }
  };
  float isc = 1.f;
  if constexpr (H2) isc = (1.f / xs) * reinterpret_cast<const float*>(wt)[(size_t)9 * KC * Cout * 16];
  auto epilogue = [&](const Cur& c) {
    // D tile col = pixel (lane & 31), row = (r & 3) + 8 (r >> 2) + 4 half; DPP quad transposes -> 16-byte stores (fire_expand)
    const int oh = c.oh0 + wave;
    const size_t ohw = (size_t)d.OH * d.OW;
    if (oh < d.OH) {
      const bool o1 = (lane & 1) != 0, o2 = (lane & 2) != 0;
      auto swz = [](float v, auto ctrl) {
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), decltype(ctrl)::value, 0xf, 0xf, true));
      };
      using X1 = std::integral_constant<int, 0xB1>;
      using X2 = std::integral_constant<int, 0x4E>;
      const size_t pix = (size_t)oh * d.OW + c.ow0 + 4 * (l31 >> 2);
      float* yb = y + ((size_t)c.n * d.out_ctot + d.out_coff) * ohw + pix;
      const float* rb = residual ? residual + ((size_t)c.n * d.res_ctot + d.res_coff) * ohw + pix : nullptr;
#pragma unroll
      for (int m = 0; m < MR; ++m)
#pragma unroll
        for (int t = 0; t < TWN; ++t)
#pragma unroll
          for (int rq = 0; rq < 4; ++rq) {
            float a0 = acc[m][t][4 * rq], a1 = acc[m][t][4 * rq + 1], a2 = acc[m][t][4 * rq + 2], a3 = acc[m][t][4 * rq + 3];
            { const float xx = o1 ? a0 : a1, yv = swz(xx, X1{}); a0 = o1 ? yv : a0; a1 = o1 ? a1 : yv; }
            { const float xx = o1 ? a2 : a3, yv = swz(xx, X1{}); a2 = o1 ? yv : a2; a3 = o1 ? a3 : yv; }
            { const float xx = o2 ? a0 : a2, yv = swz(xx, X2{}); a0 = o2 ? yv : a0; a2 = o2 ? a2 : yv; }
            { const float xx = o2 ? a1 : a3, yv = swz(xx, X2{}); a1 = o2 ? yv : a1; a3 = o2 ? a3 : yv; }
            const int co = c.co0 + 32 * m + 8 * rq + 4 * half + (lane & 3);
            if (co < Cout && c.ow0 + 32 * t + 4 * (l31 >> 2) < d.OW) {
              const float bv = bias ? bias[co] : 0.f;
              if constexpr (H2) { a0 *= isc; a1 *= isc; a2 *= isc; a3 *= isc; }
              float4 o = make_float4(a0 + bv, a1 + bv, a2 + bv, a3 + bv);
              if (rb) {
                const float4 rv = *reinterpret_cast<const float4*>(rb + (size_t)co * ohw + 32 * t);
                o.x += rv.x; o.y += rv.y; o.z += rv.z; o.w += rv.w;
              }
              st4<4>(yb + (size_t)co * ohw + 32 * t, o);
            }
          }
    }
    zero_acc();
  };
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");       // the producers' prologue
  int q = 0, gi = 0;
  while (cur.ok) {
    for (int kc = 0; kc < KC; ++kc, ++q) {
      const __bf16* patch = patch0 + (size_t)(q & 1) * PBUF;
#pragma unroll
      for (int g = 0; g < 3; ++g, ++gi) {
        compute_group(patch, ring + (size_t)(gi & 1) * AGRP, g);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      }
    }
    epilogue(cur);
    cur.l += Gx;
    set_tile(cur);
  }
  (void)nchunks_total;
}

// ---- 1x1 convolution on the same scheme, no LDS (the float4 structure of conv1x1_v4_kernel in
// conv_fwd.hip): a wave owns 128 consecutive pixels, lane l the four pixels 4l..4l+3; per 16-channel
// chunk a lane loads the float4 of 8 channels (8*half + j), component e of those eight is the
// B fragment of pixel tile e (split in registers), weights come pre-split ([chunk][plane][n][16]).
// These layers sit at the fp32-MFMA ridge (20-25 FLOP/B); with 2.7x less MFMA time per product they
// are purely HBM-bound.
// AFF (apply-on-load): the stored input is the producer's RAW output; x' = (x - mean[ci]) * scale[ci] + shift[ci]
// (+ ReLU when d.in_relu) is formed right before the split.  The three constants per input channel sit in LDS
// ([ci][4] floats, zero rows behind Cin so that padded channels stay 0); a lane reads the row of each of its eight
// channels with one ds_read_b128.
// H2: x as two fp16 pieces of x 2^k (k from *amax_x, the operand's largest magnitude or a bound on it), weights from
// dlio_conv_h2_prep (taps 1): three v_mfma_f32_32x32x16_f16 per product, 7 VALU per value in the split instead of 11; the
// stores multiply by 2^-k 2^-j.
template <int MR, bool AFF, bool H2 = false>
__global__ __launch_bounds__(256, 2) void conv1x1_bx3_kernel(
    const float* __restrict__ x, const __bf16* __restrict__ wt, const float* __restrict__ bias,
    const float* __restrict__ in_mean, const float* __restrict__ in_scale, const float* __restrict__ in_shift,
    const float* residual, float* y, DlioConvDesc d, int pix_blocks, int co_tiles, int ksplit, float* __restrict__ slab,
    const float* __restrict__ amax_x = nullptr) {
  static_assert(!(AFF && H2), "the two-piece variant has no apply-on-load input");
  constexpr int NPL = H2 ? 2 : 3;
  extern __shared__ __attribute__((aligned(16))) float aff_tab[];      // AFF: [KC * 16][4]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, half = lane >> 5;
  int bid = xcd_block_index();
  // K split over workgroups (slab != nullptr; narrowing layers on few pixels: a long channel loop and too few
  // waves to hide it): this workgroup takes chunks [kc_lo, kc_hi) and writes its partial tile to slab[ks]
  const int ks = slab ? bid % ksplit : 0;
  if (slab) bid /= ksplit;
  const int cot = bid % co_tiles; bid /= co_tiles;
  const int pb = bid % pix_blocks; bid /= pix_blocks;
  const int n = bid;
  const int P = d.OH * d.OW;
  const int co0 = cot * 32 * MR;
  const int p = (pb * 4 + wave) * 128 + 4 * l31;
  const int Cin = d.Cin, Cout = d.Cout, KC = (Cin + 15) >> 4;
  if constexpr (AFF) {
    for (int c = threadIdx.x; c < KC * 16; c += 256) {
      const bool cv = c < Cin;
      *reinterpret_cast<float4*>(aff_tab + 4 * c) = make_float4(cv ? in_mean[c] : 0.f, cv ? in_scale[c] : 0.f,
                                                                cv ? in_shift[c] : 0.f, 0.f);
    }
    __syncthreads();
  }
  const bool wave_ok = (pb * 4 + wave) * 128 < P;
  if (!wave_ok) return;
  const bool pvalid = p < P;            // P % 4 == 0: a lane's four pixels are all in or all out
  const size_t pc = pvalid ? p : 0;

  f32x16 acc[MR][4];
#pragma unroll
  for (int m = 0; m < MR; ++m)
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][e][r] = 0.f;

  const float* xn = x + ((size_t)n * d.in_ctot + d.in_coff) * (size_t)P;      // wave-uniform
  const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xn), 0, (int)((size_t)Cin * P * 4),
                                                                         0x00020000);
  const unsigned rowb = (unsigned)P * 4u;                                   // bytes per channel plane
  const unsigned voff = ((unsigned)pc + 8u * (unsigned)half * (unsigned)P) * 4u;   // (lanes past P read pixel 0 and store nothing)
  const size_t wplane = (size_t)Cout * 16;
  const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(wt), 0,
                                                                         (int)((size_t)KC * NPL * wplane * 2), 0x00020000);
  float xs = 1.f, isc = 1.f;                                                // H2: 2^k of the operand, 2^-k 2^-j
  if constexpr (H2) {
    const float am = amax_x[0];
    xs = (am > 0.f && am < 3.0e38f) ? exp2f(floorf(log2f(16384.f / am))) : 1.f;
    isc = (1.f / xs) * reinterpret_cast<const float*>(wt)[(size_t)KC * Cout * 16];
  }
  unsigned woff[MR];
#pragma unroll
  for (int m = 0; m < MR; ++m) woff[m] = ((unsigned)min(co0 + m * 32 + l31, Cout - 1) * 16u + 8u * (unsigned)half) * 2u;

  // the activation loads are double buffered; the weight fragments (L1 / L2 hits) have one buffer for two output-channel
  // tiles -- refilled right behind the MFMAs that used them -- so that the kernel fits two waves per SIMD
  constexpr int AB = MR == 2 ? 1 : 2;
  float4 v[2][8];
  bf16x8 a[AB][MR][NPL];
  auto load_a = [&](int kc, int s) {
#pragma unroll
    for (int m = 0; m < MR; ++m)
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl)
        if constexpr (BX3_ABLATE & 16) opaque(a[s % AB][m][pl]);
        else a[s % AB][m][pl] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(
            wrsrc, woff[m], (unsigned)(kc * NPL + pl) * (unsigned)wplane * 2u, 0));
  };
  auto load_chunk = [&](int kc, int s) {
    if (AB == 2) load_a(kc, s);
#pragma unroll
    for (int j = 0; j < 8; ++j)
      // buffer descriptor over this image's Cin planes + ONE per-lane 32-bit offset + a scalar offset per row (eight
      // 64-bit row pointers were what spilled at two waves per SIMD); rows behind Cin are out of range: they read 0
      v[s][j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, voff, (unsigned)(kc * 16 + j) * rowb, 0));
  };
  auto mfma_chunk = [&](int s) {
    // one pixel tile (float4 component e) at a time: its eight channels are split into three bf16x8 fragments and
    // used at once -- 12 fragment registers live instead of 48 (with two output-channel tiles the kernel then
    // fits two waves per SIMD: the load / MFMA / store phases of a wave are serial, a second wave fills them)
    constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};
    if constexpr (AFF) {
      const bool relu_in = d.in_relu != 0;
      const int kc = s >> 1;                 // (the chunk index travels in the upper bits of s)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float4 t = *reinterpret_cast<const float4*>(aff_tab + 4 * (kc * 16 + 8 * half + j));
        float4 w = v[s & 1][j];
        w.x = (w.x - t.x) * t.y + t.z; w.y = (w.y - t.x) * t.y + t.z;
        w.z = (w.z - t.x) * t.y + t.z; w.w = (w.w - t.x) * t.y + t.z;
        if (relu_in) { w.x = fmaxf(w.x, 0.f); w.y = fmaxf(w.y, 0.f); w.z = fmaxf(w.z, 0.f); w.w = fmaxf(w.w, 0.f); }
        v[s & 1][j] = w;
      }
    }
    s &= 1;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if constexpr (H2) {
        pc_f16x8 bh, bl;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float xv = (e == 0 ? v[s][j].x : (e == 1 ? v[s][j].y : (e == 2 ? v[s][j].z : v[s][j].w))) * xs;
          bh[j] = (_Float16)xv;
          bl[j] = (_Float16)(xv - (float)bh[j]);
        }
#pragma unroll
        for (int q = 0; q < 3; ++q)            // (lo, hi) (hi, lo) (hi, hi)
#pragma unroll
          for (int m = 0; m < MR; ++m)
            acc[m][e] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(pc_f16x8, a[s % AB][m][q == 0 ? 1 : 0]),
                                                               q == 1 ? bl : bh, acc[m][e], 0, 0, 0);
        continue;
      }
      bf16x8 bb[3];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float xv = e == 0 ? v[s][j].x : (e == 1 ? v[s][j].y : (e == 2 ? v[s][j].z : v[s][j].w));
        __bf16 h, m, l;
        split3(xv, h, m, l);
        bb[0][j] = h; bb[1][j] = m; bb[2][j] = l;
      }
#pragma unroll
      for (int q = DLIO_SPLIT_Q0; q < 6; ++q)
#pragma unroll
        for (int m = 0; m < MR; ++m)
          acc[m][e] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s % AB][m][PA[q] % NPL], bb[PB[q]], acc[m][e], 0, 0, 0);
    }
  };
  const int kper = slab ? (KC + ksplit - 1) / ksplit : KC;
  const int kc_lo = ks * kper, kc_hi = min(KC, kc_lo + kper);
  if (AB == 1 && wave_ok && kc_lo < kc_hi) load_a(kc_lo, 0);
  if (wave_ok && kc_lo < kc_hi) load_chunk(kc_lo, 0);
  for (int kc = kc_lo; wave_ok && kc < kc_hi; kc += 2) {
    if (kc + 1 < kc_hi) load_chunk(kc + 1, 1);
    mfma_chunk(2 * kc);
    if (AB == 1 && kc + 1 < kc_hi) load_a(kc + 1, 0);
    if (kc + 2 < kc_hi) load_chunk(kc + 2, 0);
    if (kc + 1 < kc_hi) mfma_chunk(2 * (kc + 1) + 1);
    if (AB == 1 && kc + 2 < kc_hi) load_a(kc + 2, 0);
  }
  if (slab) {           // partial tile, [ks][n][co][p] fp32; bias / residual are added by slab_sum_kernel
    if (!pvalid) return;
    float* sb = slab + (((size_t)ks * d.N + n) * Cout) * (size_t)P + pc;
#pragma unroll
    for (int m = 0; m < MR; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (co < Cout)
          *reinterpret_cast<float4*>(sb + (size_t)co * P) =
              make_float4(acc[m][0][r] * isc, acc[m][1][r] * isc, acc[m][2][r] * isc, acc[m][3][r] * isc);
      }
    return;
  }

  if (!pvalid) return;
  const size_t plane = (size_t)P;
  float* yb = y + ((size_t)n * d.out_ctot + d.out_coff) * plane + pc;
  const float* rb = residual ? residual + ((size_t)n * d.res_ctot + d.res_coff) * plane + pc : nullptr;
#pragma unroll
  for (int m = 0; m < MR; ++m) {
    float4 rv[16];
    if (rb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = min(co0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half, Cout - 1);
        rv[r] = *reinterpret_cast<const float4*>(rb + (size_t)co * plane);
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (co < Cout) {
        const float bv = bias ? bias[co] : 0.f;
        float4 o = H2 ? make_float4(acc[m][0][r] * isc + bv, acc[m][1][r] * isc + bv, acc[m][2][r] * isc + bv, acc[m][3][r] * isc + bv)
                      : make_float4(acc[m][0][r] + bv, acc[m][1][r] + bv, acc[m][2][r] + bv, acc[m][3][r] + bv);
        if (rb) { o.x += rv[r].x; o.y += rv[r].y; o.z += rv[r].z; o.w += rv[r].w; }
        st4<2>(yb + (size_t)co * plane, o);
      }
    }
  }
}

// sum of the K-split partial tiles (fixed order) + bias + residual -> the output slice
__global__ __launch_bounds__(256) void slab_sum_kernel(const float* __restrict__ slab, int ksplit, int N, int Cout, int P4,
                                                       const float* __restrict__ bias, const float* residual, int res_ctot,
                                                       int res_coff, float* y, int out_ctot, int out_coff) {
  const int64_t total = (int64_t)N * Cout * P4, stride = total;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int p4 = (int)(i % P4);
    const int64_t t = i / P4;
    const int co = (int)(t % Cout), n = (int)(t / Cout);
    const float4* sp = reinterpret_cast<const float4*>(slab) + i;
    float4 a = sp[0];
    for (int k = 1; k < ksplit; ++k) {
      const float4 b = sp[(int64_t)k * stride];
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    const float bv = bias ? bias[co] : 0.f;
    a.x += bv; a.y += bv; a.z += bv; a.w += bv;
    if (residual) {
      const float4 r = reinterpret_cast<const float4*>(residual + ((size_t)n * res_ctot + res_coff + co) * (size_t)P4 * 4)[p4];
      a.x += r.x; a.y += r.y; a.z += r.z; a.w += r.w;
    }
    reinterpret_cast<float4*>(y + ((size_t)n * out_ctot + out_coff + co) * (size_t)P4 * 4)[p4] = a;
  }
}

template <int MR, int TWN>
int launch_bx3_alds(const float* x, const __bf16* wt, const float* bias, const float* residual, float* y,
                    const DlioConvDesc& d, hipStream_t s, int ksplit = 1, float* slab = nullptr, const float* x1 = nullptr,
                    const __bf16* wt1 = nullptr, int C1 = 0) {
  constexpr int TH = 4, TW = 32 * TWN;
  const int tiles_w = cdiv(d.OW, TW), tiles_h = cdiv(d.OH, TH), co_tiles = cdiv(d.Cout, 32 * MR);
  if (!slab) ksplit = 1;
  const int64_t blocks = (int64_t)d.N * tiles_h * tiles_w * co_tiles * ksplit;
  if (blocks <= 0 || blocks > 0x7fffffff) return DLIO_EINVAL;
  const size_t patch_b = (size_t)3 * (TH + 2) * (TW + 2) * 16 * sizeof(__bf16);
  const size_t ring_b = (size_t)2 * 3 * 3 * 32 * MR * 16 * sizeof(__bf16);
  const size_t epi_b = (size_t)4 * 32 * MR * (TWN == 1 ? TW + 4 : TW + 8) * sizeof(float);
  const size_t lds = patch_b + ring_b;
  dlio_set_max_lds(reinterpret_cast<const void*>(&conv3x3_bx3_alds_kernel<MR, TWN>), (int)lds);
  static const int vec_on = getenv("DLIO_BX3_VEC_OUT") ? atoi(getenv("DLIO_BX3_VEC_OUT")) : 1;
  const int vec_out = vec_on && (d.OW & 3) == 0 && (((size_t)d.OH * d.OW) & 3) == 0 &&
                      ((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(residual)) & 15) == 0 && epi_b <= lds;
  if (slab && !vec_out) return DLIO_EUNSUP;              // (the caller only splits when the float4 store path applies)
  hipLaunchKernelGGL((conv3x3_bx3_alds_kernel<MR, TWN>), dim3((unsigned)blocks), dim3(256), lds, s, x, wt, bias, residual, y,
                     d, tiles_w, tiles_h, co_tiles, vec_out, ksplit, slab, x1, wt1, C1);
  return dlio_check_launch();
}

template <int MR, bool H2 = false>
int launch_bx3_pc(const float* x, const __bf16* wt, const float* bias, const float* residual, float* y, const DlioConvDesc& d,
                  hipStream_t s, const float* amax_x = nullptr) {
  const int tiles_w = cdiv(d.OW, 64), tiles_h = cdiv(d.OH, 4), co_tiles = cdiv(d.Cout, 32 * MR);
  const int64_t tiles = (int64_t)d.N * tiles_h * tiles_w * co_tiles;
  if (tiles <= 0 || tiles > 0x7fffffff) return DLIO_EINVAL;
  constexpr int NPL = H2 ? 2 : 3;
  constexpr size_t lds = (size_t)2 * NPL * 6 * 66 * 16 * sizeof(__bf16) + (size_t)2 * 3 * NPL * 32 * MR * 16 * sizeof(__bf16);
  dlio_set_max_lds(reinterpret_cast<const void*>(&conv3x3_bx3_pc_kernel<MR, H2>), (int)lds);
  const int64_t slots = dlio_num_cus();
  const int grid = (int)(tiles < slots ? tiles : slots);
  hipLaunchKernelGGL((conv3x3_bx3_pc_kernel<MR, H2>), dim3((unsigned)grid), dim3(512), lds, s, x, wt, bias, residual, y, d, tiles_w,
                     tiles_h, co_tiles, (int)tiles, amax_x);
  return dlio_check_launch();
}

template <int MR, int TWN, int KH = 3, int KW = 3, int SW = 1, int SH = 1, bool H2 = false>
int launch_bx3(const float* x, const __bf16* wt, const float* bias, const float* residual, float* y,
               const DlioConvDesc& d, hipStream_t s, const float* amax_x = nullptr) {
  constexpr int TH = 4, TW = 32 * TWN;
  constexpr size_t BUF = (size_t)(H2 ? 2 : 3) * (SH * (TH - 1) + KH) * (SW * (TW - 1) + KW) * 16 * sizeof(__bf16);
  const int tiles_w = cdiv(d.OW, TW), tiles_h = cdiv(d.OH, TH), co_tiles = cdiv(d.Cout, 32 * MR);
  const int64_t blocks = (int64_t)d.N * tiles_h * tiles_w * co_tiles;
  if (blocks <= 0 || blocks > 0x7fffffff) return DLIO_EINVAL;
  // one patch buffer is enough for a single-chunk layer (<= 16 input channels: the stem) -- twice the workgroups per CU
  const size_t lds = (size_t)(d.Cin <= 16 ? 1 : 2) * BUF;
  dlio_set_max_lds(reinterpret_cast<const void*>(&conv3x3_bx3_kernel<MR, TWN, KH, KW, SW, SH, H2>), (int)(2 * BUF));
  // when the next chunk's patch loads are issued: behind the first three taps' weight fragments for
  // long channel loops (loads return in order -- fragments queued behind 16-32 patch loads stall
  // their MFMAs for an HBM round trip: blk4 / blk5 data gradients 100 -> 67 us, 125 -> 60 us), ahead of
  // everything for short loops (<= 5 chunks: 3-6 % faster there)
  static const int force_at = -2;
  const int patch_at = force_at >= -1 ? force_at : ((d.Cin + 15) / 16 > 5 ? 0 : -1);
  // float4 stores through LDS: rows of 4-pixel groups (OW % 4 == 0, 16-byte aligned planes), the transposed tile fits the
  // (possibly single) patch buffer
  static const int vec_on = getenv("DLIO_BX3_VEC_OUT") ? atoi(getenv("DLIO_BX3_VEC_OUT")) : 1;
  const int vec_out = vec_on && (d.OW & 3) == 0 && (((size_t)d.OH * d.OW) & 3) == 0 &&
                      ((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(residual)) & 15) == 0 &&
                      (size_t)4 * 32 * MR * (TWN == 1 ? TW + 4 : TW + 8) * sizeof(float) <= lds;
  hipLaunchKernelGGL((conv3x3_bx3_kernel<MR, TWN, KH, KW, SW, SH, H2>), dim3((unsigned)blocks), dim3(256), lds, s, x, wt, bias, residual, y,
                     d, tiles_w, tiles_h, co_tiles, patch_at, vec_out, amax_x);
  return dlio_check_launch();
}

}  // namespace

extern "C" size_t dlio_conv_bx3_prep_floats(int Cout, int Cin, int taps, int mode) {
  if (Cout <= 0 || Cin <= 0 || taps <= 0 || (mode != 0 && mode != 1)) return 0;
  const int K = mode == 0 ? Cin : Cout, Nn = mode == 0 ? Cout : Cin;
  return (size_t)taps * ((K + 15) >> 4) * 3 * Nn * 16 / 2;      // bf16 elements / 2
}

extern "C" int dlio_conv_bx3_prep(const float* w, void* wt, int Cout, int Cin, int taps, int mode,
                                  dlio_stream_t stream) {
  if (!w || !wt || Cout <= 0 || Cin <= 0 || taps <= 0 || (mode != 0 && mode != 1)) return DLIO_EINVAL;
  const int K = mode == 0 ? Cin : Cout, Nn = mode == 0 ? Cout : Cin;
  const int64_t total = (int64_t)taps * ((K + 15) >> 4) * Nn * 16;
  hipLaunchKernelGGL(prep_bx3_kernel, dim3(ew_grid(total, 256)), dim3(256), 0, as_stream(stream), w,
                     reinterpret_cast<__bf16*>(wt), Cout, Cin, taps, mode);
  return dlio_check_launch();
}

extern "C" size_t dlio_conv3x3_bx3_prep_floats(int Cout, int Cin, int mode) {
  return dlio_conv_bx3_prep_floats(Cout, Cin, 9, mode);
}

extern "C" int dlio_conv3x3_bx3_prep(const float* w, void* wt, int Cout, int Cin, int mode, dlio_stream_t stream) {
  return dlio_conv_bx3_prep(w, wt, Cout, Cin, 9, mode, stream);
}

// which tile shape a launch gets
static void bx3_1x1_shape(const DlioConvDesc& d, int& mr, int& pix_blocks) {
  static const int force_mr = 0;
  mr = force_mr ? force_mr : (d.Cout <= 32 ? 1 : 2);
  pix_blocks = (int)(((int64_t)d.H * d.W + 511) / 512);
}

static void bx3_3x3_shape(const DlioConvDesc& d, int& mr, int& twn) {
  // tile: 64 channels x 64 columns per wave when that still gives every CU a few workgroups,
  // else narrower tiles (small feature maps / few output channels)
  static const int force_mr = 0;     // tuning knobs
  static const int force_twn = 0;
  auto blocks = [&](int m, int t) {
    return (int64_t)d.N * cdiv(d.OH, 4) * cdiv(d.OW, 32 * t) * cdiv(d.Cout, 32 * m);
  };
  const int64_t want = 2 * (int64_t)dlio_num_cus();
  // two pixel blocks per wave halve the weight-fragment traffic (the 16-byte fragment loads run at
  // the L1 bandwidth limit with one block): in isolation that pays from ~192 input channels on
  // (tools/bench_bx3.py), inside the training step from 48 on (full-step sweep: 29.05 -> 28.8 ms)
  static const int twn_cin = 48;
  mr = d.Cout <= 32 ? 1 : 2; twn = (d.OW > 32 && d.Cin >= twn_cin) ? 2 : 1;
  if (blocks(mr, twn) < want && twn == 2) twn = 1;
  if (blocks(mr, twn) < want && mr == 2) mr = 1;
  if (force_mr) mr = force_mr;
  if (force_twn) twn = force_twn;
}

// K split over workgroups for narrowing layers on few pixels: how many slices (1 = none)
static int bx3_1x1_ksplit(const DlioConvDesc& d) {
  static const int on = getenv("DLIO_BX3_1X1_KSPLIT") ? atoi(getenv("DLIO_BX3_1X1_KSPLIT")) : 1;
  if (!on) return 1;
  int mr, pb;
  bx3_1x1_shape(d, mr, pb);
  const int KC = (d.Cin + 15) / 16;
  const int64_t blocks = (int64_t)d.N * pb * cdiv(d.Cout, 32 * mr);
  if ((int64_t)d.N * d.H * d.W >= 65536 || KC < 12 || blocks >= 1024) return 1;
  int ks = (int)((1024 + blocks - 1) / blocks);
  if (ks > KC / 3) ks = KC / 3;                 // at least three chunks per slice
  if (ks > 16) ks = 16;
  static const int force = 0;   // tuning knob
  if (force && (int64_t)d.N * d.H * d.W > 16384) ks = force;
  return ks < 2 ? 1 : ks;
}

extern "C" size_t dlio_conv1x1_bx3_ws_bytes(const DlioConvDesc* dp) {
  if (!dp || dp->KH != 1 || dp->KW != 1) return 0;
  const int ks = bx3_1x1_ksplit(*dp);
  return ks < 2 ? 0 : (size_t)ks * dp->N * dp->Cout * dp->H * dp->W * sizeof(float);
}

static int bx3_1x1_run(const float* x, const void* wt, const float* bias, const float* in_mean, const float* in_scale,
                       const float* in_shift, const float* residual, float* y, void* ws, size_t ws_bytes,
                       const DlioConvDesc* dp, dlio_stream_t stream, const float* amax_x);

extern "C" int dlio_conv1x1_bx3_fwd_ws(const float* x, const void* wt, const float* bias, const float* in_mean,
                                       const float* in_scale, const float* in_shift, const float* residual,
                                       float* y, void* ws, size_t ws_bytes, const DlioConvDesc* dp,
                                       dlio_stream_t stream) {
  return bx3_1x1_run(x, wt, bias, in_mean, in_scale, in_shift, residual, y, ws, ws_bytes, dp, stream, nullptr);
}

/* the same 1x1 convolution on the two-piece fp16 split (conv1x1_bx3_kernel<MR, false, true>): *amax_x = the largest magnitude
 * of x or a bound on it, wt from dlio_conv_h2_prep (taps 1); no apply-on-load input */
extern "C" int dlio_conv1x1_h2_fwd(const float* x, const float* amax_x, const void* wt, const float* bias, const float* residual,
                                   float* y, void* ws, size_t ws_bytes, const DlioConvDesc* dp, dlio_stream_t stream) {
  if (!amax_x) return DLIO_EINVAL;
  return bx3_1x1_run(x, wt, bias, nullptr, nullptr, nullptr, residual, y, ws, ws_bytes, dp, stream, amax_x);
}

static int bx3_1x1_run(const float* x, const void* wt, const float* bias, const float* in_mean, const float* in_scale,
                       const float* in_shift, const float* residual, float* y, void* ws, size_t ws_bytes,
                       const DlioConvDesc* dp, dlio_stream_t stream, const float* amax_x) {
  if (!x || !wt || !y || !dp) return DLIO_EINVAL;
  if (in_scale && (!in_mean || !in_shift)) return DLIO_EINVAL;
  const DlioConvDesc& d = *dp;
  if (d.KH != 1 || d.KW != 1 || d.SH != 1 || d.SW != 1 || d.PH || d.PW) return DLIO_EUNSUP;
  if (d.N <= 0 || d.Cin <= 0 || d.Cout <= 0 || d.H <= 0 || d.W <= 0 || d.OH != d.H || d.OW != d.W) return DLIO_EINVAL;
  const int64_t P = (int64_t)d.H * d.W;
  if ((size_t)d.Cin * P * 4 >= 0x7fffffffull || (size_t)((d.Cin + 15) / 16) * 3 * d.Cout * 32 >= 0x7fffffffull)
    return DLIO_EUNSUP;                      // 32-bit buffer offsets
  if (P % 4 || ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) |
                 reinterpret_cast<uintptr_t>(residual)) & 15))
    return DLIO_EUNSUP;                      // float4 rows: the fp32 kernels of dlio_conv2d_fwd take these
  const size_t lds = in_scale ? (size_t)((d.Cin + 15) / 16) * 16 * 4 * sizeof(float) : 0;
  if (lds > 48 * 1024) return DLIO_EUNSUP;
  hipStream_t s = as_stream(stream);
  const double flops = 2.0 * d.N * (double)P * d.Cout * (double)d.Cin;
  const double bytes = 4.0 * d.N * ((double)d.Cin * P + (double)d.Cout * P * (residual ? 2.0 : 1.0));
  dlio_prof_begin(2, s, flops, bytes);
  int mr, pix_blocks;
  bx3_1x1_shape(d, mr, pix_blocks);
  const int co_tiles = cdiv(d.Cout, 32 * mr);
  int ksplit = bx3_1x1_ksplit(d);
  if (ksplit > 1 && (!ws || ws_bytes < (size_t)ksplit * d.N * d.Cout * P * sizeof(float))) ksplit = 1;
  float* slab = ksplit > 1 ? reinterpret_cast<float*>(ws) : nullptr;
  const int64_t blocks = (int64_t)d.N * pix_blocks * co_tiles * ksplit;
  const __bf16* w = reinterpret_cast<const __bf16*>(wt);
#define BX1(MRV, AFFV, H2V) hipLaunchKernelGGL((conv1x1_bx3_kernel<MRV, AFFV, H2V>), dim3((unsigned)blocks), dim3(256), lds, s, x, w, \
                                               bias, in_mean, in_scale, in_shift, residual, y, d, pix_blocks, co_tiles, ksplit, slab, \
                                               amax_x)
  if (amax_x) { if (mr == 1) BX1(1, false, true); else BX1(2, false, true); }
  else if (mr == 1) { if (in_scale) BX1(1, true, false); else BX1(1, false, false); }
  else { if (in_scale) BX1(2, true, false); else BX1(2, false, false); }
#undef BX1
  int rc = dlio_check_launch();
  if (!rc && slab) {
    const int64_t total = (int64_t)d.N * d.Cout * (P / 4);
    hipLaunchKernelGGL(slab_sum_kernel, dim3(ew_grid(total, 256)), dim3(256), 0, s, slab, ksplit, d.N, d.Cout, (int)(P / 4), bias,
                       residual, d.res_ctot, d.res_coff, y, d.out_ctot, d.out_coff);
    rc = dlio_check_launch();
  }
  dlio_prof_end(2, s);
  return rc;
}

extern "C" int dlio_conv1x1_bx3_fwd_aff(const float* x, const void* wt, const float* bias, const float* in_mean,
                                        const float* in_scale, const float* in_shift, const float* residual,
                                        float* y, const DlioConvDesc* dp, dlio_stream_t stream) {
  return dlio_conv1x1_bx3_fwd_ws(x, wt, bias, in_mean, in_scale, in_shift, residual, y, nullptr, 0, dp, stream);
}

extern "C" int dlio_conv1x1_bx3_fwd(const float* x, const void* wt, const float* bias, const float* residual,
                                    float* y, const DlioConvDesc* dp, dlio_stream_t stream) {
  return dlio_conv1x1_bx3_fwd_ws(x, wt, bias, nullptr, nullptr, nullptr, residual, y, nullptr, 0, dp, stream);
}

extern "C" int dlio_conv3x3_bx3_prep_batched(const DlioPrepItem* items_dev, int n_items, int64_t total,
                                             dlio_stream_t stream) {
  if (!items_dev || n_items <= 0 || total <= 0) return DLIO_EINVAL;
  if (total & 15) return DLIO_EINVAL;
  hipLaunchKernelGGL(prep_bx3_batched_kernel, dim3(ew_grid(total >> 3, 256)), dim3(256), 0, as_stream(stream),
                     items_dev, n_items, total);
  return dlio_check_launch();
}

// the tile a layer gets when workgroup count is no concern
static void bx3_3x3_shape_large(const DlioConvDesc& d, int& mr, int& twn) {
  static const int twn_cin = 48;
  mr = d.Cout <= 32 ? 1 : 2;
  twn = (d.OW > 32 && d.Cin >= twn_cin) ? 2 : 1;
}

// K split over workgroups for long channel loops on small feature maps (1 = none): even the narrowed tile of
// bx3_3x3_shape fills at most 3/4 of the chip's workgroup slots (two per CU); the split launch then runs the LARGE tile
// with the channel loop cut in slices of at least three 16-channel chunks.  (blk5 data gradient 384 -> 80 @16x32: 67 -> 48 us;
// blk4's 256 -> 64 @32x64 fills the slots with 32 x 32 tiles already and gains nothing from 2-8 slices: 62-72 us.)
static int bx3_3x3_ksplit(const DlioConvDesc& d) {
  static const int maxks = getenv("DLIO_BX3_3X3_KSPLIT") ? atoi(getenv("DLIO_BX3_3X3_KSPLIT")) : 8;    // 0 / 1: off
  static const int vec_on = getenv("DLIO_BX3_VEC_OUT") ? atoi(getenv("DLIO_BX3_VEC_OUT")) : 1;
  if (maxks < 2 || !vec_on) return 1;      // (the slabs are written by the float4 store path only)
  const int KC = (d.Cin + 15) / 16;
  auto nblocks = [&](int m, int t) { return (int64_t)d.N * cdiv(d.OH, 4) * cdiv(d.OW, 32 * t) * cdiv(d.Cout, 32 * m); };
  int mr, twn;
  bx3_3x3_shape(d, mr, twn);
  const int64_t slots = 2 * (int64_t)dlio_num_cus();
  if (KC < 6 || nblocks(mr, twn) * 4 > slots * 3 || (d.OW & 3) || (((size_t)d.OH * d.OW) & 3)) return 1;
  bx3_3x3_shape_large(d, mr, twn);
  const int64_t blocks = nblocks(mr, twn);
  int ks = (int)((slots + blocks - 1) / blocks);
  if (ks > KC / 3) ks = KC / 3;
  if (ks > maxks) ks = maxks;
  return ks < 2 ? 1 : ks;
}

extern "C" size_t dlio_conv3x3_bx3_ws_bytes(const DlioConvDesc* dp) {
  if (!dp || dp->KH != 3 || dp->KW != 3 || dp->SH != 1 || dp->SW != 1) return 0;
  if ((size_t)9 * ((dp->Cin + 15) / 16) * 3 * dp->Cout * 32 >= 0x7fffffffull) return 0;
  const int ks = bx3_3x3_ksplit(*dp);
  return ks < 2 ? 0 : (size_t)ks * dp->N * dp->Cout * dp->OH * dp->OW * sizeof(float);
}

static int bx3_3x3_run(const float* x, const void* wt, const float* bias, const float* residual, float* y, void* ws,
                       size_t ws_bytes, const DlioConvDesc* dp, dlio_stream_t stream, const float* x1, const void* wt1, int C1);

extern "C" int dlio_conv3x3_bx3_fwd_ws(const float* x, const void* wt, const float* bias, const float* residual,
                                       float* y, void* ws, size_t ws_bytes, const DlioConvDesc* dp, dlio_stream_t stream) {
  return bx3_3x3_run(x, wt, bias, residual, y, ws, ws_bytes, dp, stream, nullptr, nullptr, 0);
}

/* y = conv3x3(x, wt) + conv1x1(x1, wt1) (+ bias, + residual): the data gradient of a Fire block's expand pair,
 * dS = W3^T * dE3 + W1^T dE1 (autograd's conv2d backward of pointseg_modules.py:126-133), in ONE launch: the 1x1 layer's
 * channels are further K chunks of the 3x3 kernel that use the centre tap only.  x1 contiguous [N][C1][H][W]; wt1 from
 * dlio_conv_bx3_prep(taps = 1, mode 1). */
extern "C" int dlio_fire_expand_dgrad(const float* x, const void* wt, const float* x1, const void* wt1, int C1,
                                      const float* residual, float* y, void* ws, size_t ws_bytes, const DlioConvDesc* dp,
                                      dlio_stream_t stream) {
  if (!x1 || !wt1 || C1 <= 0 || !dp) return DLIO_EINVAL;
  if (dp->PH != 1 || dp->PW != 1 || dp->OH != dp->H || dp->OW != dp->W) return DLIO_EUNSUP;     // "same" 3x3: the centre tap is the pixel itself
  if ((size_t)((C1 + 15) / 16) * 3 * dp->Cout * 32 >= 0x7fffffffull) return DLIO_EUNSUP;
  return bx3_3x3_run(x, wt, nullptr, residual, y, ws, ws_bytes, dp, stream, x1, wt1, C1);
}

// geometry the producer / consumer kernel takes: long channel loops on large maps (the data gradients of fire_blk1-3)
// (min_half_cus: tiles >= this x CUs / 2.  The three-piece kernel needs two tiles per CU to beat the alds kernel; the two-piece
//  one wins from half a tile per CU on -- fire_blk4's 256 -> 64 @32x64: 51.6 us against 73.7 on the alds kernel, three-piece 80.7)
static bool bx3_pc_geom_ok(const DlioConvDesc& d, int pc_min_half_cus = 4) {
  constexpr int pc_kc = 4;              // chunks of 16 input channels from which the producer / consumer split pays
  return d.KH == 3 && d.KW == 3 && d.SH == 1 && d.SW == 1 && d.PH == 1 && d.PW == 1 && d.OH == d.H && d.OW == d.W && d.OW >= 64 &&
         (d.OW & 3) == 0 && (d.Cin + 15) / 16 >= pc_kc && (size_t)9 * ((d.Cin + 15) / 16) * 3 * d.Cout * 32 < 0x7fffffffull &&
         (size_t)d.Cin * d.H * d.W * 4 < 0xffffff00ull &&
         (int64_t)d.N * cdiv(d.OH, 4) * cdiv(d.OW, 64) * cdiv(d.Cout, d.Cout <= 32 ? 32 : 64) * 2 >= pc_min_half_cus * (int64_t)dlio_num_cus();
}

/* the same convolution on the two-piece fp16 split (conv3x3_bx3_pc_kernel<MR, true>): x fp32 with *amax_x = its largest
 * magnitude (left on the device by the kernel that produced x), wt from dlio_conv_h2_prep.  Only the launch sizes of the
 * producer / consumer kernel (dlio_conv3x3_h2_ok), DLIO_EUNSUP otherwise. */
extern "C" int dlio_conv3x3_h2_ok(const DlioConvDesc* dp) {
  return dp && dp->N > 0 && dp->Cin > 0 && dp->Cout > 0 && bx3_pc_geom_ok(*dp, 1);
}

extern "C" int dlio_conv3x3_h2_fwd(const float* x, const float* amax_x, const void* wt, const float* bias, const float* residual,
                                   float* y, const DlioConvDesc* dp, dlio_stream_t stream) {
  if (!x || !amax_x || !wt || !y || !dp) return DLIO_EINVAL;
  const DlioConvDesc& d = *dp;
  if (!dlio_conv3x3_h2_ok(dp) || ((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(residual)) & 15)) return DLIO_EUNSUP;
  hipStream_t s = as_stream(stream);
  const double flops = 2.0 * d.N * (double)d.OH * d.OW * d.Cout * (double)d.Cin * 9;
  const double bytes = 4.0 * d.N * ((double)d.Cin * d.H * d.W + (double)d.Cout * d.OH * d.OW * (residual ? 2.0 : 1.0));
  DlioProfScope prof(3, s, flops, bytes);
  const __bf16* w = reinterpret_cast<const __bf16*>(wt);
  return d.Cout <= 32 ? launch_bx3_pc<1, true>(x, w, bias, residual, y, d, s, amax_x)
                      : launch_bx3_pc<2, true>(x, w, bias, residual, y, d, s, amax_x);
}

static int bx3_3x3_run(const float* x, const void* wt, const float* bias, const float* residual, float* y, void* ws,
                       size_t ws_bytes, const DlioConvDesc* dp, dlio_stream_t stream, const float* x1, const void* wt1, int C1) {
  if (!x || !wt || !y || !dp) return DLIO_EINVAL;
  const DlioConvDesc& d = *dp;
  if (d.KH != 3 || d.KW != 3 || d.SH != 1 || d.SW != 1) return DLIO_EUNSUP;
  if (d.N <= 0 || d.Cin <= 0 || d.Cout <= 0 || d.H <= 0 || d.W <= 0 || d.PH < 0 || d.PW < 0) return DLIO_EINVAL;
  const int oh_lo = d.H + 2 * d.PH - 2, ow_lo = d.W + 2 * d.PW - 2;
  if (d.OH < oh_lo || d.OH > oh_lo + 2 || d.OW < ow_lo || d.OW > ow_lo + 2 || d.OH < 1 || d.OW < 1) return DLIO_EINVAL;
  hipStream_t s = as_stream(stream);
  const double flops = 2.0 * d.N * (double)d.OH * d.OW * d.Cout * ((double)d.Cin * 9 + (x1 ? C1 : 0));
  const double bytes = 4.0 * d.N * ((double)(d.Cin + (x1 ? C1 : 0)) * d.H * d.W + (double)d.Cout * d.OH * d.OW * (residual ? 2.0 : 1.0));
  dlio_prof_begin(3, s, flops, bytes);      // profiler kind 3: split-bf16 3x3 convolutions
  const __bf16* w = reinterpret_cast<const __bf16*>(wt);
  int mr, twn;
  bx3_3x3_shape(d, mr, twn);
  int rc;
  // weight fragments through LDS (conv3x3_bx3_alds_kernel)
  // (DLIO_BX3_ALDS: 0 off, 1 the 64-channel tiles only, 2 = default all tiles: 25.39 / 25.61 / 26.3 ms for 2 / 1 / 0)
  static const int alds = 2;
  bool use_alds = alds && (size_t)9 * ((d.Cin + 15) / 16) * 3 * d.Cout * 32 < 0x7fffffffull && (alds == 2 || mr == 2);
  if (x1 && !use_alds) return DLIO_EUNSUP;
  const __bf16* w1 = reinterpret_cast<const __bf16*>(wt1);
  // K split: the LARGE tile (which bx3_3x3_shape gave up to get more workgroups) with the channel loop cut in slices
  int mrs, twns;
  bx3_3x3_shape_large(d, mrs, twns);
  int ksplit = alds ? bx3_3x3_ksplit(d) : 1;
  const size_t ohw = (size_t)d.OH * d.OW;
  if (ksplit > 1 && (!ws || ws_bytes < (size_t)ksplit * d.N * d.Cout * ohw * sizeof(float) ||
                     ((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(residual) | reinterpret_cast<uintptr_t>(ws)) & 15)))
    ksplit = 1;
  if (ksplit > 1) { mr = mrs; twn = twns; use_alds = true; }
  // producer / consumer kernel: long channel loops on large maps (the data gradients of fire_blk1-3)
  static const int pc_on = getenv("DLIO_BX3_PC") ? atoi(getenv("DLIO_BX3_PC")) : 1;
  if (pc_on && !x1 && ksplit == 1 && use_alds && bx3_pc_geom_ok(d) &&
      ((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(residual)) & 15) == 0) {
    const int rcp = d.Cout <= 32 ? launch_bx3_pc<1>(x, w, bias, residual, y, d, s) : launch_bx3_pc<2>(x, w, bias, residual, y, d, s);
    dlio_prof_end(3, s);
    return rcp;
  }
  float* slab = ksplit > 1 ? reinterpret_cast<float*>(ws) : nullptr;
#define L3(MRV, TWV) (use_alds ? launch_bx3_alds<MRV, TWV>(x, w, bias, residual, y, d, s, ksplit, slab, x1, w1, C1) \
                               : launch_bx3<MRV, TWV>(x, w, bias, residual, y, d, s))
  if (mr == 1) rc = twn == 2 ? L3(1, 2) : L3(1, 1);
  else rc = twn == 2 ? L3(2, 2) : L3(2, 1);
#undef L3
  if (!rc && slab) {
    const int64_t total = (int64_t)d.N * d.Cout * (int64_t)(ohw / 4);
    hipLaunchKernelGGL(slab_sum_kernel, dim3(ew_grid(total, 256)), dim3(256), 0, s, slab, ksplit, d.N, d.Cout, (int)(ohw / 4), bias,
                       residual, d.res_ctot, d.res_coff, y, d.out_ctot, d.out_coff);
    rc = dlio_check_launch();
  }
  dlio_prof_end(3, s);
  return rc;
}

extern "C" int dlio_conv3x3_bx3_fwd(const float* x, const void* wt, const float* bias, const float* residual,
                                    float* y, const DlioConvDesc* dp, dlio_stream_t stream) {
  return dlio_conv3x3_bx3_fwd_ws(x, wt, bias, residual, y, nullptr, 0, dp, stream);
}

// 3x5 taps, stride (1, 2) (the PointSeg stem, pointseg_net.py:18-20: 2C -> 64 channels at 64 x 2048 -> 64 x 1024) on the
// same kernel; weights from dlio_conv_bx3_prep(taps = 15, mode 0).  Forward only (the stem's input needs no gradient).
extern "C" int dlio_conv3x5s2_bx3_fwd(const float* x, const void* wt, const float* bias, const float* residual, float* y,
                                      const DlioConvDesc* dp, dlio_stream_t stream) {
  if (!x || !wt || !y || !dp) return DLIO_EINVAL;
  const DlioConvDesc& d = *dp;
  const bool s35 = d.KH == 3 && d.KW == 5 && d.SH == 1 && d.SW == 2;
  const bool s33 = d.KH == 3 && d.KW == 3 && d.SH == 2 && d.SW == 2;      // FlowNet conv4-6, ResNet layer2-4 (stride 2)
  if (!s35 && !s33) return DLIO_EUNSUP;
  if (d.N <= 0 || d.Cin <= 0 || d.Cout <= 0 || d.H <= 0 || d.W <= 0 || d.PH < 0 || d.PW < 0) return DLIO_EINVAL;
  if (d.OH != (d.H + 2 * d.PH - d.KH) / d.SH + 1 || d.OW != (d.W + 2 * d.PW - d.KW) / d.SW + 1 || d.OH < 1 || d.OW < 1)
    return DLIO_EINVAL;
  hipStream_t s = as_stream(stream);
  if (s33) {
    const double flops33 = 2.0 * d.N * (double)d.OH * d.OW * d.Cout * (double)d.Cin * 9;
    const double bytes33 = 4.0 * d.N * ((double)d.Cin * d.H * d.W + (double)d.Cout * d.OH * d.OW * (residual ? 2.0 : 1.0));
    dlio_prof_begin(3, s, flops33, bytes33);
    const int rc33 = d.Cout <= 32 ? launch_bx3<1, 1, 3, 3, 2, 2>(x, reinterpret_cast<const __bf16*>(wt), bias, residual, y, d, s)
                                  : launch_bx3<2, 1, 3, 3, 2, 2>(x, reinterpret_cast<const __bf16*>(wt), bias, residual, y, d, s);
    dlio_prof_end(3, s);
    return rc33;
  }
  const double flops = 2.0 * d.N * (double)d.OH * d.OW * d.Cout * (double)d.Cin * 15;
  const double bytes = 4.0 * d.N * ((double)d.Cin * d.H * d.W + (double)d.Cout * d.OH * d.OW * (residual ? 2.0 : 1.0));
  dlio_prof_begin(3, s, flops, bytes);
  const __bf16* w = reinterpret_cast<const __bf16*>(wt);
  static const int force_mr = 0;   // tuning knob
  const int mr = force_mr ? force_mr : (d.Cout <= 32 ? 1 : 2);
  const int rc = mr == 1 ? launch_bx3<1, 1, 3, 5, 2>(x, w, bias, residual, y, d, s)
                         : launch_bx3<2, 1, 3, 5, 2>(x, w, bias, residual, y, d, s);
  dlio_prof_end(3, s);
  return rc;
}

// stride-1 convolution with a small tap window and an explicit output extent on the split-bf16 kernel: the phases of a
// strided layer's data gradient (functional._dgrad_phases: dx[a::SH, b::SW] is a stride-1 correlation of dy with the tap
// subset w[:, :, rh::SH, rw::SW], reversed, under a top / left padding of (PH, PW); rows / columns behind the input read
// zero, so OH / OW may exceed the symmetric-padding formula).  FlowNet conv2-6 / ResNet layer2-4 (lidar_feat_nets.py:248-257,
// resnet.py:27-47).  Weights from dlio_conv_bx3_prep(taps = KH * KW, mode).
extern "C" int dlio_conv_bx3_fwd_taps(const float* x, const void* wt, const float* bias, const float* residual, float* y,
                                      const DlioConvDesc* dp, dlio_stream_t stream) {
  if (!x || !wt || !y || !dp) return DLIO_EINVAL;
  const DlioConvDesc& d = *dp;
  if (d.SH != 1 || d.SW != 1) return DLIO_EUNSUP;
  if (d.N <= 0 || d.Cin <= 0 || d.Cout <= 0 || d.H <= 0 || d.W <= 0 || d.PH < 0 || d.PW < 0 || d.OH < 1 || d.OW < 1)
    return DLIO_EINVAL;
  if (d.OH > d.H + 2 * d.PH || d.OW > d.W + 2 * d.PW) return DLIO_EINVAL;
  hipStream_t s = as_stream(stream);
  const double flops = 2.0 * d.N * (double)d.OH * d.OW * d.Cout * (double)d.Cin * d.KH * d.KW;
  const double bytes = 4.0 * d.N * ((double)d.Cin * d.H * d.W + (double)d.Cout * d.OH * d.OW * (residual ? 2.0 : 1.0));
  const __bf16* w = reinterpret_cast<const __bf16*>(wt);
  const bool wide = d.OW >= 48;
  int rc = DLIO_EUNSUP;
  dlio_prof_begin(3, s, flops, bytes);
#define BX3_TAPS(kh, kw)                                                                        \
  if (d.KH == kh && d.KW == kw)                                                                 \
    rc = wide ? launch_bx3<2, 2, kh, kw, 1>(x, w, bias, residual, y, d, s) : launch_bx3<2, 1, kh, kw, 1>(x, w, bias, residual, y, d, s);
  BX3_TAPS(3, 3)
  else BX3_TAPS(3, 2)
  else BX3_TAPS(2, 2)
  else BX3_TAPS(2, 1)
  else BX3_TAPS(1, 2)
  else BX3_TAPS(1, 1)
#undef BX3_TAPS
  dlio_prof_end(3, s);
  return rc;
}

/* The strided / small-tap-window launches above on the two-piece fp16 split (conv3x3_bx3_kernel<.., H2>): *amax_x = the largest
 * magnitude of x (left on the device by its producer), wt from dlio_conv_h2_prep(taps = KH * KW, mode).  Layers with more than
 * 32 output channels (else DLIO_EUNSUP: the three-piece launch takes them). */
extern "C" int dlio_conv_h2_fwd_strided(const float* x, const float* amax_x, const void* wt, const float* bias,
                                        const float* residual, float* y, const DlioConvDesc* dp, dlio_stream_t stream) {
  if (!x || !amax_x || !wt || !y || !dp) return DLIO_EINVAL;
  const DlioConvDesc& d = *dp;
  const bool s35 = d.KH == 3 && d.KW == 5 && d.SH == 1 && d.SW == 2;
  const bool s33 = d.KH == 3 && d.KW == 3 && d.SH == 2 && d.SW == 2;
  if ((!s35 && !s33) || d.Cout <= 32) return DLIO_EUNSUP;
  if (d.N <= 0 || d.Cin <= 0 || d.Cout <= 0 || d.H <= 0 || d.W <= 0 || d.PH < 0 || d.PW < 0) return DLIO_EINVAL;
  if (d.OH != (d.H + 2 * d.PH - d.KH) / d.SH + 1 || d.OW != (d.W + 2 * d.PW - d.KW) / d.SW + 1 || d.OH < 1 || d.OW < 1)
    return DLIO_EINVAL;
  hipStream_t s = as_stream(stream);
  const double flops = 2.0 * d.N * (double)d.OH * d.OW * d.Cout * (double)d.Cin * d.KH * d.KW;
  const double bytes = 4.0 * d.N * ((double)d.Cin * d.H * d.W + (double)d.Cout * d.OH * d.OW * (residual ? 2.0 : 1.0));
  dlio_prof_begin(3, s, flops, bytes);
  const __bf16* w = reinterpret_cast<const __bf16*>(wt);
  const int rc = s33 ? launch_bx3<2, 1, 3, 3, 2, 2, true>(x, w, bias, residual, y, d, s, amax_x)
                     : launch_bx3<2, 1, 3, 5, 2, 1, true>(x, w, bias, residual, y, d, s, amax_x);
  dlio_prof_end(3, s);
  return rc;
}

extern "C" int dlio_conv_h2_fwd_taps(const float* x, const float* amax_x, const void* wt, const float* bias,
                                     const float* residual, float* y, const DlioConvDesc* dp, dlio_stream_t stream) {
  if (!x || !amax_x || !wt || !y || !dp) return DLIO_EINVAL;
  const DlioConvDesc& d = *dp;
  if (d.SH != 1 || d.SW != 1) return DLIO_EUNSUP;
  if (d.N <= 0 || d.Cin <= 0 || d.Cout <= 0 || d.H <= 0 || d.W <= 0 || d.PH < 0 || d.PW < 0 || d.OH < 1 || d.OW < 1)
    return DLIO_EINVAL;
  if (d.OH > d.H + 2 * d.PH || d.OW > d.W + 2 * d.PW) return DLIO_EINVAL;
  hipStream_t s = as_stream(stream);
  const double flops = 2.0 * d.N * (double)d.OH * d.OW * d.Cout * (double)d.Cin * d.KH * d.KW;
  const double bytes = 4.0 * d.N * ((double)d.Cin * d.H * d.W + (double)d.Cout * d.OH * d.OW * (residual ? 2.0 : 1.0));
  const __bf16* w = reinterpret_cast<const __bf16*>(wt);
  const bool wide = d.OW >= 48;
  int rc = DLIO_EUNSUP;
  dlio_prof_begin(3, s, flops, bytes);
#define H2_TAPS(kh, kw)                                                                         \
  if (d.KH == kh && d.KW == kw)                                                                 \
    rc = wide ? launch_bx3<2, 2, kh, kw, 1, 1, true>(x, w, bias, residual, y, d, s, amax_x)     \
              : launch_bx3<2, 1, kh, kw, 1, 1, true>(x, w, bias, residual, y, d, s, amax_x);
  H2_TAPS(3, 3)
  else H2_TAPS(3, 2)
  else H2_TAPS(2, 2)
  else H2_TAPS(2, 1)
  else H2_TAPS(1, 2)
  else H2_TAPS(1, 1)
#undef H2_TAPS
  dlio_prof_end(3, s);
  return rc;
}

// timing probes compiled into this file (bit 0: DLIO_SPLIT_Q0, bit 1: BX3_ABLATE); 0 in the product build, checked at load (dlio_build_probes)
int dlio_probe_bx3() { return ((DLIO_SPLIT_Q0) != 0 ? 1 : 0) | ((BX3_ABLATE) != 0 ? 2 : 0); }
