// Native-bf16 convolutions of the mixed-precision PointSeg path (BASELINE configs[4]: bf16 activation
// and gradient storage, fp32 master weights / accumulation / BatchNorm statistics / loss).  No
// reference counterpart (the reference is fp32 only, SURVEY 2 "new-in-build"); same nn.Conv2d call
// sites as conv_bx3.hip (pointseg_modules.py:96-106 squeeze / expand1x1 / expand3x3).
//
// Same tiling as the split-bf16 kernels of conv_bx3.hip, with ONE bf16 plane per operand and one
// v_mfma_f32_32x32x16_bf16 per product instead of six: activations are bf16 in HBM (half the bytes of
// the fp32 path, which is what these HBM-bound layers are priced in), weights are re-laid-out from the
// fp32 master copy once per optimizer step ([tap][chunk][n][16 k] bf16), accumulation is fp32, the
// epilogue adds the fp32 bias / a bf16 residual and rounds once (RNE) to bf16.
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void prep_one(const float* __restrict__ w, __bf16* __restrict__ wt, int Cout, int Cin,
                                         int taps, int mode, int64_t e) {
  const int K = mode == 0 ? Cin : Cout, Nn = mode == 0 ? Cout : Cin;
  const int KC = (K + 15) >> 4;
  const int kk = (int)(e & 15);
  int64_t t = e >> 4;
  const int n = (int)(t % Nn); t /= Nn;
  const int kc = (int)(t % KC);
  const int tap = (int)(t / KC);
  const int k = kc * 16 + kk;
  float v = 0.f;
  if (k < K) {
    if (mode == 0) v = w[((int64_t)n * Cin + k) * taps + tap];
    else v = w[((int64_t)k * Cin + n) * taps + (taps - 1 - tap)];
  }
  wt[e] = (__bf16)v;
}

// w [Cout][Cin][taps] fp32 -> wt [taps][KC][Nn][16] bf16; mode 0: k = ci, n = co; mode 1: k = co, n = ci, taps reversed
__global__ void prep_bf16_kernel(const float* __restrict__ w, __bf16* __restrict__ wt, int Cout, int Cin, int taps,
                                 int mode) {
  const int K = mode == 0 ? Cin : Cout, Nn = mode == 0 ? Cout : Cin;
  const int64_t total = (int64_t)taps * ((K + 15) >> 4) * Nn * 16;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
    prep_one(w, wt, Cout, Cin, taps, mode, i);
}

__global__ void prep_bf16_batched_kernel(const DlioPrepItem* __restrict__ items, int n_items, int64_t total) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int lo = 0, hi = n_items - 1;                 // last item with start <= i
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (items[mid].start <= i) lo = mid; else hi = mid - 1;
    }
    const DlioPrepItem it = items[lo];
    prep_one(it.w, reinterpret_cast<__bf16*>(it.wt), it.Cout, it.Cin, it.taps, it.mode, i - it.start);
  }
}

// ---- 3x3 stride-1: 4 waves = 4 output rows x 32*TWN columns x 32*MR output channels -------------
template <int MR, int TWN>
__global__ __launch_bounds__(256, 2) void conv3x3_bf16_kernel(
    const __bf16* __restrict__ x, const __bf16* __restrict__ wt, const float* __restrict__ bias,
    const __bf16* residual, __bf16* y, DlioConvDesc d, int tiles_w, int tiles_h, int co_tiles, int patch_at, int vec_out) {
  constexpr int TH = 4, TW = 32 * TWN;
  constexpr int PR = TH + 2, PC = TW + 2, NPOSP = PR * PC;
  constexpr int NPOS = (NPOSP + 255) / 256;              // patch positions per thread
  constexpr int PLANE = NPOSP * 16;                      // bf16 per buffer
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __bf16* smem = reinterpret_cast<__bf16*>(smem_raw);    // [2 buffers][NPOSP][16], then the weight chunks [2][9 taps][32 MR rows][16]
  // Weight fragments through LDS (conv3x3_bx3_alds_kernel, conv_bx3.hip): the four waves own four rows of the SAME output
  // channels and each fetched the same fragments from global memory.  One plane of bf16 weights is small enough to keep
  // both the patch and the weights of a chunk double-buffered (61 KB for the 64 x 64 tile): the workgroup fetches the 9
  // taps of the next chunk straight into LDS (buffer_load_dwordx4 ... lds, 4-5 instructions per wave) while it computes.
  constexpr int ACH = 9 * 32 * MR * 16;                  // bf16 per weight chunk
  constexpr int AINS = 9 * MR;                           // wave instructions per chunk (64 sixteen-byte pieces each)
  __bf16* abuf = smem + 2 * PLANE;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
  int bid = xcd_block_index();
  const int cot = bid % co_tiles; bid /= co_tiles;
  const int tw = bid % tiles_w; bid /= tiles_w;
  const int th = bid % tiles_h;
  const int n = bid / tiles_h;
  const int co0 = cot * 32 * MR, oh0 = th * TH, ow0 = tw * TW;
  const int Cin = d.Cin, Cout = d.Cout, HW = d.H * d.W;
  const int KC = (Cin + 15) >> 4;

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
  const __bf16* xn = x + ((size_t)n * d.in_ctot + d.in_coff) * HW;
  __bf16 reg[NPOS][16];
  auto load_chunk = [&](int kc) {
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      const int ci = kc * 16 + c;
      const __bf16* xc = xn + (size_t)min(ci, Cin - 1) * HW;
#pragma unroll
      for (int j = 0; j < NPOS; ++j) reg[j][c] = xc[poff[j]];
    }
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      const bool cv = kc * 16 + c < Cin;
#pragma unroll
      for (int j = 0; j < NPOS; ++j) reg[j][c] = (cv && pval[j]) ? reg[j][c] : (__bf16)0.f;
    }
  };
  auto store_chunk = [&](__bf16* buf) {
#pragma unroll
    for (int j = 0; j < NPOS; ++j) {
      const int pos = tid + j * 256;
      if (pos < NPOSP) {
        bf16x8 p0, p1;
#pragma unroll
        for (int c = 0; c < 8; ++c) { p0[c] = reg[j][c]; p1[c] = reg[j][8 + c]; }
        bf16x8* dst = reinterpret_cast<bf16x8*>(buf + pos * 16);
        dst[0] = p0; dst[1] = p1;
      }
    }
  };

  const size_t wplane = (size_t)Cout * 16;
  const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(wt), 0,
                                                                         (int)((size_t)9 * KC * wplane * 2), 0x00020000);
  unsigned awoff[MR];            // piece of this lane inside a (tap, chunk) block: row lane / 2 of tile m, k-half lane & 1
#pragma unroll
  for (int m = 0; m < MR; ++m) awoff[m] = ((unsigned)min(co0 + m * 32 + (lane >> 1), Cout - 1) * 16u + 8u * (unsigned)(lane & 1)) * 2u;
  auto load_achunk = [&](int kc, int slot) {              // instruction t = tap * MR + m, dealt round-robin to the waves
#pragma unroll
    for (int i = 0; i < (AINS + 3) / 4; ++i) {
      const int t = i * 4 + wave;
      if (t < AINS) {
        const int tap = t / MR, m = t - tap * MR;
        __bf16* dst = abuf + (size_t)slot * ACH + (size_t)t * 64 * 8;
#if defined(__HIP_DEVICE_COMPILE__)     // (the host pass cannot instantiate the address-space cast)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (__attribute__((address_space(3))) void*)dst, 16, awoff[m],
                                                 (unsigned)(((size_t)tap * KC + kc) * wplane * 2), 0, 0);
#else
        (void)dst; (void)m;
#endif
      }
    }
  };
  // LDS order of a chunk: piece (t, lane) -> row lane / 2, k-half lane & 1 of (tap, m): fragment of lane (l31, half) =
  // row l31, k-half `half` -> piece 2 * l31 + half
  auto read_a = [&](const __bf16* ab, int tap, bf16x8 (&a)[MR]) {
#pragma unroll
    for (int m = 0; m < MR; ++m)
      a[m] = *reinterpret_cast<const bf16x8*>(ab + ((size_t)(tap * MR + m) * 64 + 2 * l31 + half) * 8);
  };

  f32x16 acc[MR][TWN];
#pragma unroll
  for (int m = 0; m < MR; ++m)
#pragma unroll
    for (int t = 0; t < TWN; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][t][r] = 0.f;

  auto compute = [&](const __bf16* buf, const __bf16* ab, int kc, bool prefetch) {
    bf16x8 a[2][MR];
    if (patch_at < 0 && prefetch) load_chunk(kc + 1);
    read_a(ab, 0, a[0]);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int kh = tap / 3, kw = tap - 3 * kh;
      if (tap + 1 < 9) read_a(ab, tap + 1, a[(tap + 1) & 1]);
      if (tap == patch_at && prefetch) load_chunk(kc + 1);
      bf16x8 b[TWN];
#pragma unroll
      for (int t = 0; t < TWN; ++t) {
        const int pos = (wave + kh) * PC + 32 * t + l31 + kw;
        b[t] = *reinterpret_cast<const bf16x8*>(buf + pos * 16 + 8 * half);
      }
      const auto& aa = a[tap & 1];
#pragma unroll
      for (int m = 0; m < MR; ++m)
#pragma unroll
        for (int t = 0; t < TWN; ++t)
          acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aa[m], b[t], acc[m][t], 0, 0, 0);
    }
  };

  load_achunk(0, 0);
  load_chunk(0);
  store_chunk(smem);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int kc = 0; kc < KC; ++kc) {
    const __bf16* cur = smem + (size_t)(kc & 1) * PLANE;
    __bf16* nxt = smem + (size_t)((kc + 1) & 1) * PLANE;
    const bool more = kc + 1 < KC;
    if (more) load_achunk(kc + 1, (kc + 1) & 1);          // that slot was read during chunk kc - 1: every wave is past its barrier
    compute(cur, abuf + (size_t)(kc & 1) * ACH, kc, more);
    if (more) store_chunk(nxt);                         // (waits for the patch loads, which are younger than the weight pieces)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // epilogue: D tile col = pixel (lane & 31), row = (r & 3) + 8 (r >> 2) + 4 half
  const int oh = oh0 + wave;
  if (oh >= d.OH) return;
  const size_t ohw = (size_t)d.OH * d.OW;
  if (vec_out) {
    // Straight from the accumulators a store instruction writes 2-byte elements: two 64-byte pieces.  Transposed through
    // the wave's own LDS region (fp32, 32 channels at a time, [channel][TW + 8]: one rounding, after bias + residual) a
    // lane stores 8 pixels = 16 bytes and an instruction covers 8 / TWN channel rows of TW contiguous pixels
    // (conv3x3_bx3_kernel's epilogue, conv_bx3.hip).
    constexpr int TWP = TWN == 1 ? TW + 4 : TW + 8;
    constexpr int Q = TW / 8;                           // 8-pixel groups per channel row
    float* wbuf = reinterpret_cast<float*>(smem_raw) + wave * (32 * TWP);
    __bf16* yrow = y + ((size_t)n * d.out_ctot + d.out_coff) * ohw + (size_t)oh * d.OW + ow0;
    const __bf16* rrow = residual ? residual + ((size_t)n * d.res_ctot + d.res_coff) * ohw + (size_t)oh * d.OW + ow0 : nullptr;
#pragma unroll
    for (int m = 0; m < MR; ++m) {
#pragma unroll
      for (int t = 0; t < TWN; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          wbuf[((r & 3) + 8 * (r >> 2) + 4 * half) * TWP + 32 * t + l31] = acc[m][t][r];
#pragma unroll
      for (int i = 0; i < (32 * Q) / 64; ++i) {
        const int idx = i * 64 + lane, cl = idx / Q, q = idx - cl * Q;
        const int co = co0 + 32 * m + cl;
        if (co >= Cout || ow0 + 8 * q >= d.OW) continue;
        const float4 v0 = *reinterpret_cast<const float4*>(wbuf + cl * TWP + 8 * q);
        const float4 v1 = *reinterpret_cast<const float4*>(wbuf + cl * TWP + 8 * q + 4);
        float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
        const float bv = bias ? bias[co] : 0.f;
        bf16x8 rv;
        if (rrow) rv = *reinterpret_cast<const bf16x8*>(rrow + (size_t)co * ohw + 8 * q);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (__bf16)(v[e] + bv + (rrow ? (float)rv[e] : 0.f));
        *reinterpret_cast<bf16x8*>(yrow + (size_t)co * ohw + 8 * q) = o;
      }
    }
    return;
  }
#pragma unroll
  for (int m = 0; m < MR; ++m)
#pragma unroll
    for (int t = 0; t < TWN; ++t) {
      const int ow = ow0 + 32 * t + l31;
      if (ow >= d.OW) continue;
      const size_t pix = (size_t)oh * d.OW + ow;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + 32 * m + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (co >= Cout) continue;
        float v = acc[m][t][r];
        if (bias) v += bias[co];
        if (residual) v += (float)residual[((size_t)n * d.res_ctot + d.res_coff + co) * ohw + pix];
        y[((size_t)n * d.out_ctot + d.out_coff + co) * ohw + pix] = (__bf16)v;
      }
    }
}

// ---- 1x1: a wave owns 128 consecutive pixels, lane l the four pixels 4l..4l+3 (one 8-byte load per
// channel), half selects channels 8*half..8*half+7 of the 16-channel chunk; component e of the eight
// loads is the B fragment of pixel tile e.  No LDS.
// Two waves per SIMD (the load / MFMA / store phases of a wave are serial; a second wave fills them): buffer-descriptor
// loads -- one 32-bit lane offset + scalar row offsets instead of eight 64-bit row pointers, rows behind Cin out of
// range = 0 -- keep the kernel under 256 registers (conv1x1_bx3_kernel, conv_bx3.hip).
template <int MR>
__global__ __launch_bounds__(256, 2) void conv1x1_bf16_kernel(
    const __bf16* __restrict__ x, const __bf16* __restrict__ wt, const float* __restrict__ bias,
    const __bf16* residual, __bf16* y, DlioConvDesc d, int pix_blocks, int co_tiles) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, half = lane >> 5;
  int bid = xcd_block_index();
  const int cot = bid % co_tiles; bid /= co_tiles;
  const int pb = bid % pix_blocks; bid /= pix_blocks;
  const int n = bid;
  const int P = d.OH * d.OW;
  const int co0 = cot * 32 * MR;
  const int p = (pb * 4 + wave) * 128 + 4 * l31;
  if ((pb * 4 + wave) * 128 >= P) return;
  const bool pvalid = p < P;            // P % 4 == 0: a lane's four pixels are all in or all out
  const size_t pc = pvalid ? p : 0;
  const int Cin = d.Cin, Cout = d.Cout, KC = (Cin + 15) >> 4;

  f32x16 acc[MR][4];
#pragma unroll
  for (int m = 0; m < MR; ++m)
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][e][r] = 0.f;

  const __bf16* xn = x + ((size_t)n * d.in_ctot + d.in_coff) * (size_t)P;       // wave-uniform
  const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(xn), 0, (int)((size_t)Cin * P * 2),
                                                                         0x00020000);
  const unsigned rowb = (unsigned)P * 2u;                                      // bytes per channel plane
  const unsigned voff = ((unsigned)pc + 8u * (unsigned)half * (unsigned)P) * 2u;
  const size_t wplane = (size_t)Cout * 16;
  const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(wt), 0, (int)((size_t)KC * wplane * 2),
                                                                         0x00020000);
  unsigned woff[MR];
#pragma unroll
  for (int m = 0; m < MR; ++m) woff[m] = ((unsigned)min(co0 + m * 32 + l31, Cout - 1) * 16u + 8u * (unsigned)half) * 2u;

  bf16x4 v[2][8];
  bf16x8 a[2][MR];
  auto load_chunk = [&](int kc, int s) {
#pragma unroll
    for (int m = 0; m < MR; ++m)
      a[s][m] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, woff[m], (unsigned)kc * (unsigned)wplane * 2u, 0));
#pragma unroll
    for (int j = 0; j < 8; ++j)
      v[s][j] = __builtin_bit_cast(bf16x4, __builtin_amdgcn_raw_buffer_load_b64(xrsrc, voff, (unsigned)(kc * 16 + j) * rowb, 0));
  };
  auto mfma_chunk = [&](int s) {
    bf16x8 b[4];
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) b[e][j] = v[s][j][e];
#pragma unroll
    for (int m = 0; m < MR; ++m)
#pragma unroll
      for (int e = 0; e < 4; ++e)
        acc[m][e] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s][m], b[e], acc[m][e], 0, 0, 0);
  };
  load_chunk(0, 0);
  for (int kc = 0; kc < KC; kc += 2) {
    if (kc + 1 < KC) load_chunk(kc + 1, 1);
    mfma_chunk(0);
    if (kc + 2 < KC) load_chunk(kc + 2, 0);
    if (kc + 1 < KC) mfma_chunk(1);
  }

  if (!pvalid) return;
  const size_t plane = (size_t)P;
  __bf16* yb = y + ((size_t)n * d.out_ctot + d.out_coff) * plane + pc;
  const __bf16* rb = residual ? residual + ((size_t)n * d.res_ctot + d.res_coff) * plane + pc : nullptr;
#pragma unroll
  for (int m = 0; m < MR; ++m) {
    bf16x4 rv[16];
    if (rb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = min(co0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half, Cout - 1);
        rv[r] = *reinterpret_cast<const bf16x4*>(rb + (size_t)co * plane);
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (co >= Cout) continue;
      const float bv = bias ? bias[co] : 0.f;
      bf16x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (__bf16)(acc[m][e][r] + bv + (rb ? (float)rv[r][e] : 0.f));
      *reinterpret_cast<bf16x4*>(yb + (size_t)co * plane) = o;
    }
  }
}

template <int MR, int TWN>
int launch_3x3(const __bf16* x, const __bf16* wt, const float* bias, const __bf16* residual, __bf16* y,
               const DlioConvDesc& d, hipStream_t s) {
  constexpr int TH = 4, TW = 32 * TWN;
  const int tiles_w = cdiv(d.OW, TW), tiles_h = cdiv(d.OH, TH), co_tiles = cdiv(d.Cout, 32 * MR);
  const int64_t blocks = (int64_t)d.N * tiles_h * tiles_w * co_tiles;
  if (blocks <= 0 || blocks > 0x7fffffff) return DLIO_EINVAL;
  size_t lds = (size_t)2 * (TH + 2) * (TW + 2) * 16 * sizeof(__bf16) + (size_t)2 * 9 * 32 * MR * 16 * sizeof(__bf16);   // patch + weight chunks
  const int patch_at = (d.Cin + 15) / 16 > 5 ? 0 : -1;
  // 16-byte stores through LDS: rows of 8-pixel groups, 16-byte aligned planes; the fp32 transposed tiles of the four
  // waves (32 channels at a time) need more LDS than the bf16 patch buffers
  static const int vec_on = 1;
  const int vec_out = vec_on && (d.OW & 7) == 0 && (((size_t)d.OH * d.OW) & 7) == 0 &&
                      ((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(residual)) & 15) == 0;
  if (vec_out) {
    const size_t need = (size_t)4 * 32 * (TWN == 1 ? TW + 4 : TW + 8) * sizeof(float);
    if (need > lds) lds = need;
  }
  {
    dlio_set_max_lds(reinterpret_cast<const void*>(&conv3x3_bf16_kernel<MR, TWN>), 64 * 1024);
  }
  hipLaunchKernelGGL((conv3x3_bf16_kernel<MR, TWN>), dim3((unsigned)blocks), dim3(256), lds, s, x, wt, bias, residual,
                     y, d, tiles_w, tiles_h, co_tiles, patch_at, vec_out);
  return dlio_check_launch();
}

}  // namespace

extern "C" size_t dlio_conv_bf16_prep_elems(int Cout, int Cin, int taps, int mode) {
  if (Cout <= 0 || Cin <= 0 || taps <= 0 || (mode != 0 && mode != 1)) return 0;
  const int K = mode == 0 ? Cin : Cout, Nn = mode == 0 ? Cout : Cin;
  return (size_t)taps * ((K + 15) >> 4) * Nn * 16;
}

extern "C" int dlio_conv_bf16_prep(const float* w, void* wt, int Cout, int Cin, int taps, int mode,
                                   dlio_stream_t stream) {
  if (!w || !wt || Cout <= 0 || Cin <= 0 || taps <= 0 || (mode != 0 && mode != 1)) return DLIO_EINVAL;
  const int64_t total = (int64_t)dlio_conv_bf16_prep_elems(Cout, Cin, taps, mode);
  hipLaunchKernelGGL(prep_bf16_kernel, dim3(ew_grid(total, 256)), dim3(256), 0, as_stream(stream), w,
                     reinterpret_cast<__bf16*>(wt), Cout, Cin, taps, mode);
  return dlio_check_launch();
}

extern "C" int dlio_conv_bf16_prep_batched(const DlioPrepItem* items_dev, int n_items, int64_t total,
                                           dlio_stream_t stream) {
  if (!items_dev || n_items <= 0 || total <= 0) return DLIO_EINVAL;
  hipLaunchKernelGGL(prep_bf16_batched_kernel, dim3(ew_grid(total, 256)), dim3(256), 0, as_stream(stream), items_dev,
                     n_items, total);
  return dlio_check_launch();
}

extern "C" int dlio_conv3x3_bf16_fwd(const void* x, const void* wt, const float* bias, const void* residual, void* y,
                                     const DlioConvDesc* dp, dlio_stream_t stream) {
  if (!x || !wt || !y || !dp) return DLIO_EINVAL;
  const DlioConvDesc& d = *dp;
  if (d.KH != 3 || d.KW != 3 || d.SH != 1 || d.SW != 1) return DLIO_EUNSUP;
  if (d.N <= 0 || d.Cin <= 0 || d.Cout <= 0 || d.H <= 0 || d.W <= 0 || d.PH < 0 || d.PW < 0) return DLIO_EINVAL;
  const int oh_lo = d.H + 2 * d.PH - 2, ow_lo = d.W + 2 * d.PW - 2;
  if (d.OH < oh_lo || d.OH > oh_lo + 2 || d.OW < ow_lo || d.OW > ow_lo + 2 || d.OH < 1 || d.OW < 1) return DLIO_EINVAL;
  hipStream_t s = as_stream(stream);
  const double flops = 2.0 * d.N * (double)d.OH * d.OW * d.Cout * (double)d.Cin * 9;
  const double bytes = 2.0 * d.N * ((double)d.Cin * d.H * d.W + (double)d.Cout * d.OH * d.OW * (residual ? 2.0 : 1.0));
  DlioProfScope prof(11, s, flops, bytes);
  const __bf16* xb = reinterpret_cast<const __bf16*>(x);
  const __bf16* w = reinterpret_cast<const __bf16*>(wt);
  const __bf16* rb = reinterpret_cast<const __bf16*>(residual);
  __bf16* yb = reinterpret_cast<__bf16*>(y);
  auto blocks = [&](int mr, int twn) {
    return (int64_t)d.N * cdiv(d.OH, 4) * cdiv(d.OW, 32 * twn) * cdiv(d.Cout, 32 * mr);
  };
  const int64_t want = 2 * (int64_t)dlio_num_cus();
  int mr = d.Cout <= 32 ? 1 : 2, twn = d.OW > 32 ? 2 : 1;
  if (blocks(mr, twn) < want && twn == 2) twn = 1;
  if (blocks(mr, twn) < want && mr == 2) mr = 1;
  {
    const int lds = 2 * 6 * 66 * 16 * 2;
    dlio_set_max_lds(reinterpret_cast<const void*>(&conv3x3_bf16_kernel<1, 2>), lds);
    dlio_set_max_lds(reinterpret_cast<const void*>(&conv3x3_bf16_kernel<2, 2>), lds);
  }
  if (mr == 1) return twn == 2 ? launch_3x3<1, 2>(xb, w, bias, rb, yb, d, s) : launch_3x3<1, 1>(xb, w, bias, rb, yb, d, s);
  return twn == 2 ? launch_3x3<2, 2>(xb, w, bias, rb, yb, d, s) : launch_3x3<2, 1>(xb, w, bias, rb, yb, d, s);
}

extern "C" int dlio_conv1x1_bf16_fwd(const void* x, const void* wt, const float* bias, const void* residual, void* y,
                                     const DlioConvDesc* dp, dlio_stream_t stream) {
  if (!x || !wt || !y || !dp) return DLIO_EINVAL;
  const DlioConvDesc& d = *dp;
  if (d.KH != 1 || d.KW != 1 || d.SH != 1 || d.SW != 1 || d.PH || d.PW) return DLIO_EUNSUP;
  if (d.N <= 0 || d.Cin <= 0 || d.Cout <= 0 || d.H <= 0 || d.W <= 0 || d.OH != d.H || d.OW != d.W) return DLIO_EINVAL;
  const int64_t P = (int64_t)d.H * d.W;
  if (P % 4 || ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(residual)) & 7))
    return DLIO_EUNSUP;
  if ((size_t)d.Cin * P * 2 >= 0x7fffffffull || (size_t)((d.Cin + 15) / 16) * d.Cout * 32 >= 0x7fffffffull)
    return DLIO_EUNSUP;                      // 32-bit buffer offsets
  hipStream_t s = as_stream(stream);
  const double flops = 2.0 * d.N * (double)P * d.Cout * (double)d.Cin;
  const double bytes = 2.0 * d.N * ((double)d.Cin * P + (double)d.Cout * P * (residual ? 2.0 : 1.0));
  DlioProfScope prof(11, s, flops, bytes);
  const int mr = d.Cout <= 32 ? 1 : 2;
  const int pix_blocks = (int)((P + 511) / 512), co_tiles = cdiv(d.Cout, 32 * mr);
  const int64_t blocks = (int64_t)d.N * pix_blocks * co_tiles;
  const __bf16* xb = reinterpret_cast<const __bf16*>(x);
  const __bf16* w = reinterpret_cast<const __bf16*>(wt);
  const __bf16* rb = reinterpret_cast<const __bf16*>(residual);
  __bf16* yb = reinterpret_cast<__bf16*>(y);
  if (mr == 1)
    hipLaunchKernelGGL((conv1x1_bf16_kernel<1>), dim3((unsigned)blocks), dim3(256), 0, s, xb, w, bias, rb, yb, d,
                       pix_blocks, co_tiles);
  else
    hipLaunchKernelGGL((conv1x1_bf16_kernel<2>), dim3((unsigned)blocks), dim3(256), 0, s, xb, w, bias, rb, yb, d,
                       pix_blocks, co_tiles);
  return dlio_check_launch();
}
