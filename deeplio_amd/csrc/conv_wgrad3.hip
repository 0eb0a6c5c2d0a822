// Weight gradient of the 3x3 stride-1 pad-1 convolution on the bf16 matrix cores with fp32 accuracy
// (three-way operand splits, six v_mfma_f32_32x32x16_bf16 per product -- conv_bx3.hip), rebuilt around ONE
// idea: the tap's column shift is paid on the dY side, in registers, once per tile.
//
//   dW[co][ci][ky][kx] = sum_{n,oh,iw} dY[n][co][oh][iw - kx + 1] * X[n][ci][oh + ky - 1][iw]
//
// GEMM view per wave: K = 32 consecutive input columns iw of one row,
//   A (32 output channels x K): lane (co = l & 31, half = l >> 5) owns dY[co][oh][iw0 + 16 half - 1 .. + 16]
//      (4 aligned 16-byte loads + 2 halo dwords straight from global memory).  The 18 values are split
//      into bf16 (hi, mid, lo) ONCE and packed in pairs: E[m] = (v[2m], v[2m+1]) serves kx = 2 (E[4q..4q+3])
//      and kx = 0 (E[4q+1..4q+4]); O[m] = v_alignbit(E[m+1], E[m], 16) serves kx = 1.  ~125 VALU per tile
//      instead of a 3-way split per MFMA operand use.
//   B (K x 32 lanes): X is staged through LDS already split (three bf16 planes, pairs of columns per dword,
//      written once per element), WITHOUT any column halo: the fragment of a lane is the same 16 aligned
//      bytes for all three kx.  The 32 B lanes are two SLOTS of 16 channels; slot s = (channel group g, ky),
//      i.e. the row shift is a per-lane LDS address.  One ds_read_b128 per plane feeds 3 (kx) x 6 x MR MFMAs.
//   C: acc[m][t][kx] -- 9 taps = 3 accumulator sets (kx) x 3 slot rows (ky).
// 16-channel slots fit every PointSeg squeeze width (16/32/48/64/80); a chunk of CKC = 16 * G channels uses
// 3 G slots = NTB tiles of two (NTB = 3: 32 channels, no waste; NTB = 2: 16 channels, 3 of 4 slots).
//
// Workgroup = 4 waves = 4 consecutive rows of a 4 x 32 pixel tile, all on the same (channel tile, chunk);
// pixel tiles are split over workgroups, each workgroup's partial sum is one slab, a second kernel sums the slabs
// in a fixed order (deterministic, no atomics: conv_wgrad.hip).  One workgroup per CU, software pipeline:
// during the MFMAs of tile n the patch of tile n+1 is split into the other LDS buffer and its dY registers
// are prepared, and the global loads of tile n+2 are issued.
//
// NAT (mixed-precision path, BASELINE configs[4]): x / dy are bf16 in memory -- one plane, one MFMA per product,
// no split at all; the aligned pairs a lane loads ARE O, E is the funnel shift.
//
// H2 (two-piece fp16 split): both operands as TWO fp16 pieces of x 2^k (2^k from the tensor's largest magnitude or
// a bound on it, left on the device by the kernel that produced the tensor): three v_mfma_f32_32x32x16_f16 per product, two
// planes through LDS, 8 VALU per pair in the split instead of 11; the slab write multiplies by the two inverse scales.
//
// Replaces the weight-gradient half of nn.Conv2d backward for pointseg_modules.py:103 (expand3x3),
// resnet.py BasicBlock conv3x3, base_net.py:55-71 conv3_1 / conv4_1 / conv5_1.
#include "common.h"
#include "wgrad3.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pk(bf16x2 v) { return __builtin_bit_cast(unsigned, v); }

// compile-time loop: f(integral_constant<int, i>) for i in [B, E) -- the instruction stream below is laid out by index
// arithmetic that must fold (a `#pragma unroll` the optimiser declines leaves register arrays indexed at run time = scratch)
// timing ablations (tools/w3_ablate.sh builds variant libraries; results are WRONG with any bit set): 1 no patch split / LDS
// store, 2 no dY operand preparation, 4 no global loads, 8 no MFMAs, 16 no LDS fragment reads, 32 no barrier -- all in the loop
#ifndef W3_ABL
#define W3_ABL 0
#endif
#ifndef W3_SGB
#define W3_SGB 5
#endif
template <int I> struct IC { static constexpr int value = I; };
template <int B, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (B < E) {
    f(IC<B>{});
    static_for<B + 1, E>(f);
  }
}

// voffset: per-lane, range-checked against num_records (an invalid lane passes an offset beyond it and reads zeros).
// The scalar tile base is added to it on the VALU (32-bit, wraps): the hardware adds soffset zero-extended and
// unchecked, so a base that is "negative" before the lane's row offset is added (row -1 of the first tile row, the
// left halo of column 0) cannot travel in it.
__device__ __forceinline__ u32x4 load_b128(__amdgpu_buffer_rsrc_t rsrc, unsigned voffset, unsigned soffset) {
  return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voffset, soffset, 0));
}
// NEVER write __builtin_bit_cast(float, vec[i]): hipcc 7.2 miscompiles a bit_cast whose operand is a vector-element
// lvalue -- it reads element 0 whatever i is (a 16-byte load then shrinks to one dword, replicated).  Passing the
// element by value first is what works.
__device__ __forceinline__ float as_f(unsigned u) { return __builtin_bit_cast(float, u); }

// (a, b) -> packed bf16 pairs (low half = a) of the three pieces a = h + m + l: 11 VALU (v_cvt_pk_bf16_f32 rounds to
// nearest even; a bf16 widens to fp32 by a shift / a mask of the packed dword)
__device__ __forceinline__ unsigned cvt_pk(float a, float b) {
  return pk(bf16x2{(__bf16)a, (__bf16)b});
}
__device__ __forceinline__ void split_pair(float a, float b, unsigned& h, unsigned& m, unsigned& l) {
  h = cvt_pk(a, b);
  const float ra = a - as_f(h << 16), rb = b - as_f(h & 0xffff0000u);
  m = cvt_pk(ra, rb);
  l = cvt_pk(ra - as_f(m << 16), rb - as_f(m & 0xffff0000u));
}

// two fp16 pieces of (a s, b s), packed pairs: a s = h + l up to 2^-22 of the piece's magnitude (s a power of two: exact)
__device__ __forceinline__ void split_pair_h2(float a, float b, float s, unsigned& h, unsigned& l) {
  const float as = a * s, bs = b * s;
  const f16x2 hv = {(_Float16)as, (_Float16)bs};
  h = __builtin_bit_cast(unsigned, hv);
  const f16x2 lv = {(_Float16)(as - (float)hv[0]), (_Float16)(bs - (float)hv[1])};
  l = __builtin_bit_cast(unsigned, lv);
}
// 2^k that maps a tensor's largest magnitude (or a bound on it) to 2^14 (fp16 max 65504); 1 for an all-zero / non-finite one
__device__ __forceinline__ float h2_scale(float am) {
  return (am > 0.f && am < 3.0e38f) ? exp2f(floorf(log2f(16384.f / am))) : 1.f;
}
// two-piece products, smallest first: (lo,hi) (hi,lo) (hi,hi)
__device__ constexpr int HA_[3] = {1, 0, 0}, HB_[3] = {0, 1, 0};

// six products, smallest first: (lo,hi) (mid,mid) (hi,lo) (mid,hi) (hi,mid) (hi,hi)   [A plane, B plane]
__device__ constexpr int TA_[9] = {2, 1, 0, 1, 0, 0, 0, 0, 0}, TB_[9] = {0, 1, 2, 0, 1, 0, 0, 0, 0};

template <int MR, int NTB, bool NAT, bool H2 = false>
struct W3Cfg {
  static_assert(!(NAT && H2), "one format");
  static constexpr int TH = 4, TW = 32;
  static constexpr int NP = NAT ? 1 : H2 ? 2 : 3;        // 16-bit planes
  static constexpr int G = (2 * NTB) / 3;                // 16-channel groups per chunk
  static constexpr int CKC = 16 * G;
  static constexpr int CS = 20;                          // dwords per (row, channel): 16 + 4 pad -> odd number of 16-B slots
  static constexpr int RS = CKC * CS;                    // row stride (RS / 4 = 0 mod 16: the two slots of a tile never collide)
  static constexpr int PS = (TH + 2) * RS;               // plane stride
  static constexpr int BUF = NP * PS;                    // dwords of one patch buffer
  static constexpr int RED = MR * NTB * 3 * 16 * 64;     // accumulator exchange at the end
  static constexpr int SM_DWORDS = 2 * BUF > 2 * RED ? 2 * BUF : 2 * RED;   // epilogue: two accumulator regions
  static constexpr size_t LDS_BYTES = (size_t)SM_DWORDS * 4;
  static constexpr int PPR = NAT ? 4 : 8;                // 16-byte pieces per (row, channel): 4 fp32 / 8 bf16 columns each
  static constexpr int XITEMS = (TH + 2) * CKC * PPR;
  static constexpr int NXI = (XITEMS + 255) / 256;       // per thread
};

template <int MR, int NTB, bool NAT, bool H2 = false>
__global__ __launch_bounds__(256, 1) void wgrad3_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                        float* __restrict__ wsp, DlioConvDesc d, int co_tiles,
                                                        int ci_chunks, int splits, int tiles_w, int tiles_h,
                                                        const float* __restrict__ amax_x = nullptr,
                                                        const float* __restrict__ amax_dy = nullptr) {
  using C = W3Cfg<MR, NTB, NAT, H2>;
  float xs = 1.f, ys = 1.f;                              // H2: the operands' 2^k
  if constexpr (H2) { xs = h2_scale(amax_x[0]); ys = h2_scale(amax_dy[0]); }
  constexpr int NP = C::NP;
  constexpr unsigned EB = NAT ? 2u : 4u;
  extern __shared__ __attribute__((aligned(16))) unsigned smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;

  int bid = xcd_block_index();
  const int npairs = co_tiles * ci_chunks;
  const int split = bid / npairs; bid -= split * npairs;
  const int cic = bid % ci_chunks; bid /= ci_chunks;
  const int co0 = bid * 32 * MR;
  const int c0 = cic * C::CKC;

  // B fragment byte offsets inside a plane (q = 0): slot (t, lane bit 4) -> (group, ky); empty slots read slot 0
  unsigned boff[NTB];
#pragma unroll
  for (int t = 0; t < NTB; ++t) {
    const int s = 2 * t + ((lane >> 4) & 1);
    const int g = s / 3, ky = s - 3 * g;
    const bool ok = g < C::G;
    boff[t] = (unsigned)(((wave + (ok ? ky : 0)) * C::RS + ((ok ? 16 * g : 0) + (lane & 15)) * C::CS + 8 * half) * 4);
  }

  f32x16 acc[MR][NTB][3];
#pragma unroll
  for (int m = 0; m < MR; ++m)
#pragma unroll
    for (int t = 0; t < NTB; ++t)
#pragma unroll
      for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][t][k][r] = 0.f;

  const int total_tiles = d.N * tiles_h * tiles_w;
  const unsigned ohw = (unsigned)(d.OH * d.OW), HW = (unsigned)(d.H * d.W);     // byte sizes < 4 GB (dlio_wgrad3_plan)
  constexpr unsigned OOB = 0xffffff00u;
  const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(x), 0, (int)((size_t)d.N * d.in_ctot * HW * EB), 0x00020000);
  const __amdgpu_buffer_rsrc_t arsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(dy), 0, (int)((size_t)d.N * d.out_ctot * ohw * EB), 0x00020000);

  // ---- X patch: rows oh0 - 1 .. oh0 + 4, columns iw0 .. iw0 + 31 (no column halo), CKC channels, as 16-byte pieces
  // (fp32: 4 columns, 8 per row; bf16: 8 columns, 4 per row).  Everything that does not depend on the tile is computed
  // once; per tile a piece costs one scalar base + a select (unconditional loads: an invalid piece reads beyond
  // num_records and gets zeros, no branches).
  constexpr int PPR = C::PPR, NXI = C::NXI;
  constexpr int RPI = 256 / (PPR * C::CKC);              // patch rows covered by one piece index (256 threads)
  static_assert(RPI * PPR * C::CKC == 256, "a piece index must cover whole patch rows");
  const int xc4 = tid % PPR, xch = (tid / PPR) % C::CKC, xr0 = RPI == 1 ? 0 : tid / (PPR * C::CKC);
  const int xcol = xc4 * (32 / PPR);
  // per-thread constants: byte offset inside (image, tile) and the largest tile column base for which the piece is inside
  // the image (-1: never -- channel beyond Cin)
  const unsigned xoff0 = ((unsigned)xch * HW + (unsigned)(xr0 * d.W + xcol)) * EB;
  const int xlim = c0 + xch < d.Cin ? d.W - xcol : -1;
  const unsigned xlds0 = (unsigned)(xr0 * C::RS + xch * C::CS + (NAT ? 4 : 2) * xc4);
  struct XRaw { u32x4 v[NXI]; };
  // tile coordinates: decoded once per tile (scalar unit), shared by its three load groups
  struct Tile { int n, th, tw; bool ok; };
  auto decode = [&](int tile) {
    Tile t;
    int tt = tile;
    t.tw = tt % tiles_w; tt /= tiles_w;
    t.th = tt % tiles_h; tt /= tiles_h;
    t.n = tt;
    t.ok = tile < total_tiles;
    return t;
  };
  auto load_x = [&](const Tile& t, XRaw& rx, int i0, int i1) {
    // (row -1 of the first tile row wraps below the image base: such pieces are invalid and never use it)
    const unsigned base = ((unsigned)(t.n * d.in_ctot + d.in_coff + c0) * HW + (unsigned)((t.th * C::TH - 1) * d.W + t.tw * C::TW)) * EB;
    const int col0 = t.ok ? t.tw * C::TW : 0x7fffffff;
#pragma unroll
    for (int i = 0; i < NXI; ++i) {
      if (i < i0 || i >= i1) continue;
      const int r = RPI * i + xr0;
      const int ih = t.th * C::TH - 1 + r;
      bool v;
      if constexpr (RPI == 1) v = (((unsigned)ih < (unsigned)d.H) ? col0 : 0x7fffffff) < xlim;      // row test on the scalar unit
      else v = (col0 < xlim) & (r < C::TH + 2) & ((unsigned)ih < (unsigned)d.H);
      rx.v[i] = load_b128(xrsrc, v ? base + (unsigned)(RPI * i * d.W) * EB + xoff0 : OOB, 0);
    }
  };
  // one half of a piece (fp32: a pair of columns -> one dword per plane, kept in `xt` until the piece's second half
  // completes the 8-byte stores); bf16: the whole piece at hp == 0
  unsigned xt[3];
  auto store_half = [&](const XRaw& rx, int i, int hp, unsigned* buf) {
    unsigned* p = buf + xlds0 + RPI * i * C::RS;
    const bool ok = !(C::XITEMS % 256 != 0 && RPI * i + xr0 >= C::TH + 2);
    if constexpr (NAT) {
      if (hp == 0 && ok) *reinterpret_cast<u32x4*>(p) = rx.v[i];
    } else if constexpr (H2) {
      unsigned h, l;
      split_pair_h2(as_f(rx.v[i][2 * hp]), as_f(rx.v[i][2 * hp + 1]), xs, h, l);
      if (hp == 0) { xt[0] = h; xt[1] = l; }
      else if (ok) {
        *reinterpret_cast<u32x2*>(p) = u32x2{xt[0], h};
        *reinterpret_cast<u32x2*>(p + C::PS) = u32x2{xt[1], l};
      }
    } else {
      unsigned h, m, l;
      split_pair(as_f(rx.v[i][2 * hp]), as_f(rx.v[i][2 * hp + 1]), h, m, l);
      if (hp == 0) { xt[0] = h; xt[1] = m; xt[2] = l; }
      else if (ok) {
        *reinterpret_cast<u32x2*>(p) = u32x2{xt[0], h};
        *reinterpret_cast<u32x2*>(p + C::PS) = u32x2{xt[1], m};
        *reinterpret_cast<u32x2*>(p + 2 * C::PS) = u32x2{xt[2], l};
      }
    }
  };
  auto store_piece = [&](const XRaw& rx, int i, unsigned* buf) {
    if (C::XITEMS % 256 != 0 && RPI * i + xr0 >= C::TH + 2) return;
    unsigned* p = buf + xlds0 + RPI * i * C::RS;
    if constexpr (NAT) {
      *reinterpret_cast<u32x4*>(p) = rx.v[i];
    } else if constexpr (H2) {
      unsigned h0, l0, h1, l1;
      split_pair_h2(as_f(rx.v[i][0]), as_f(rx.v[i][1]), xs, h0, l0);
      split_pair_h2(as_f(rx.v[i][2]), as_f(rx.v[i][3]), xs, h1, l1);
      *reinterpret_cast<u32x2*>(p) = u32x2{h0, h1};
      *reinterpret_cast<u32x2*>(p + C::PS) = u32x2{l0, l1};
    } else {
      const float v0 = as_f(rx.v[i][0]), v1 = as_f(rx.v[i][1]), v2 = as_f(rx.v[i][2]), v3 = as_f(rx.v[i][3]);
      unsigned h0, m0, l0, h1, m1, l1;
      split_pair(v0, v1, h0, m0, l0);
      split_pair(v2, v3, h1, m1, l1);
      *reinterpret_cast<u32x2*>(p) = u32x2{h0, h1};
      *reinterpret_cast<u32x2*>(p + C::PS) = u32x2{m0, m1};
      *reinterpret_cast<u32x2*>(p + 2 * C::PS) = u32x2{l0, l1};
    }
  };

  // ---- dY: lane (co, half) of wave w: row oh0 + w, columns iw0 + 16 half - 1 .. + 16
  constexpr int NAL = NAT ? 2 : 4;                       // aligned 16-byte loads of the 16 columns
  struct ARaw { u32x4 v[MR][NAL]; unsigned hl[MR], hr[MR]; };
  unsigned aoff[MR];
  int alim[MR];                                          // largest tile column base for which the lane's first 4 columns exist
#pragma unroll
  for (int m = 0; m < MR; ++m) {
    aoff[m] = ((unsigned)(32 * m + l31) * ohw + (unsigned)(wave * d.OW + 16 * half)) * EB;
    alim[m] = co0 + 32 * m + l31 < d.Cout ? d.OW - 16 * half : -1;
  }
  auto load_a = [&](const Tile& t, ARaw& a) {
    const int oh = t.th * C::TH + wave;
    const unsigned base = ((unsigned)(t.n * d.out_ctot + d.out_coff + co0) * ohw + (unsigned)(t.th * C::TH * d.OW + t.tw * C::TW)) * EB;
    const int col0 = (t.ok & (oh < d.OH)) ? t.tw * C::TW : 0x7fffffff;          // wave-uniform
#pragma unroll
    for (int m = 0; m < MR; ++m) {
#pragma unroll
      for (int q = 0; q < NAL; ++q)
        a.v[m][q] = load_b128(arsrc, (col0 < alim[m] - (16 / NAL) * q) ? base + aoff[m] + 16u * q : OOB, 0);
      // halos: column 16 half - 1 (not for the image's first column) and 16 half + 16
      const bool vl = (col0 < alim[m] + 1) & !((col0 == 0) & (half == 0)), vr = col0 < alim[m] - 16;
      if constexpr (NAT) {
        a.hl[m] = (unsigned)__builtin_amdgcn_raw_buffer_load_b16(arsrc, vl ? base + aoff[m] - 2u : OOB, 0, 0);
        a.hr[m] = (unsigned)__builtin_amdgcn_raw_buffer_load_b16(arsrc, vr ? base + aoff[m] + 32u : OOB, 0, 0);
      } else {
        a.hl[m] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(arsrc, vl ? base + aoff[m] - 4u : OOB, 0, 0);
        a.hr[m] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(arsrc, vr ? base + aoff[m] + 64u : OOB, 0, 0);
      }
    }
  };
  // ---- A operands of one k-block q (8 columns per lane): from the ten values v[8q .. 8q+9] (v[0] = left halo,
  // v[1..16] = the lane's columns, v[17] = right halo) the even pairs e[j] = (v[8q+2j], v[8q+2j+1]), j = 0..4; then
  // kx = 2 reads e[0..3], kx = 0 reads c[0..3] = e[1..4] (a copy: MFMA operands are aligned register tuples), kx = 1
  // reads o[j] = funnel shift of (e[j+1], e[j]) by 16 bits.  bf16 operands: the loaded dwords are the odd pairs, e is the shift.
  struct AQ { unsigned e[NP][MR][5], c[NP][MR][4], o[NP][MR][4]; };
  auto a_pair = [&](const ARaw& a, int q, int j, AQ& f) {
#pragma unroll
    for (int m = 0; m < MR; ++m) {
      const int k = 8 * q + 2 * j;                       // v index of the pair's low half, 0 .. 16
      if constexpr (NAT) {
        // raw dword i = (v[2i+1], v[2i+2]);  (v[k], v[k+1]) with k even = (raw[k/2 - 1].hi, raw[k/2].lo)
        const int i = k / 2;
        if (i == 0) f.e[0][m][j] = (a.hl[m] & 0xffffu) | (a.v[m][0][0] << 16);
        else if (i == 8) f.e[0][m][j] = (a.v[m][1][3] >> 16) | (a.hr[m] << 16);
        else f.e[0][m][j] = __builtin_amdgcn_alignbit(a.v[m][i >> 2][i & 3], a.v[m][(i - 1) >> 2][(i - 1) & 3], 16);
        if (j < 4) f.o[0][m][j] = a.v[m][(4 * q + j) >> 2][(4 * q + j) & 3];
      } else {
        const float lo = k == 0 ? as_f(a.hl[m]) : as_f(a.v[m][(k - 1) >> 2][(k - 1) & 3]);
        const float hi = k == 16 ? as_f(a.hr[m]) : as_f(a.v[m][k >> 2][k & 3]);
        if constexpr (H2) split_pair_h2(lo, hi, ys, f.e[0][m][j], f.e[1][m][j]);
        else split_pair(lo, hi, f.e[0][m][j], f.e[1][m][j], f.e[2][m][j]);
      }
    }
  };
  auto a_finish = [&](int p, AQ& f) {
#pragma unroll
    for (int m = 0; m < MR; ++m)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f.c[p][m][j] = f.e[p][m][j + 1];
        if constexpr (!NAT) f.o[p][m][j] = __builtin_amdgcn_alignbit(f.e[p][m][j + 1], f.e[p][m][j], 16);
      }
  };

  // six products, smallest first: (lo,hi) (mid,mid) (hi,lo) (mid,hi) (hi,mid) (hi,hi)   [A plane, B plane]
  constexpr int NTERM = NAT ? 1 : H2 ? 3 : 6 - DLIO_SPLIT_Q0;      // (DLIO_SPLIT_Q0: common.h, 0 in the product build)
  constexpr int NS = 2 * NTB;                            // steps of a tile: (k-block q, B tile t)
  constexpr int UH = NTB * NTERM;                        // work units (= groups of 3 MR MFMAs) per k-block half

  struct BFrag { bf16x8 v[NP]; };
  auto read_b = [&](const unsigned* buf, int step, BFrag& b) {
    const int q = step / NTB, t = step - q * NTB;
    const char* bp = reinterpret_cast<const char*>(buf) + 16 * q;
#pragma unroll
    for (int p = 0; p < NP; ++p)
      b.v[p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(bp + boff[t] + (size_t)p * C::PS * 4));
  };

  // ---- pipeline.  One wave per SIMD issues everything in order, so the kernel is a hand-laid instruction stream: a tile
  // is 2 NTB steps of NTERM groups of 3 MR MFMAs, and behind every group sits ONE item of the staging work, in this order:
  //   first half  (MFMAs of k-block 0): the k-block-1 operands of THIS tile (5 pair splits, NP finishes), the dY loads
  //                of tile n+2 into the registers that just became free, the first half of tile n+1's patch pieces
  //                (split + LDS store into the other buffer), the loads of those pieces for tile n+2;
  //   second half (MFMAs of k-block 1): the k-block-0 operands of tile n+1, the second half of tile n+1's patch, its reload.
  // Every load has at least half a tile of MFMAs between issue and first use; no branches (a tile behind the last one
  // loads zeros).
  constexpr int NLO = NXI / 2;                           // patch pieces staged in the first half
  constexpr int XH = NAT ? 1 : 2;                        // items per piece
  constexpr int L0 = 5 + NP + 1 + XH * NLO + 1;          // items of the first half
  constexpr int L1 = 5 + NP + XH * (NXI - NLO) + 1;      // items of the second half
  XRaw xa;
  ARaw ra, rb;
  AQ a0, a1;
  unsigned* buf0 = smem;
  unsigned* buf1 = smem + C::BUF;
  const int ntiles = split < total_tiles ? (total_tiles - split + splits - 1) / splits : 0;
  {
    const Tile t0 = decode(split);
    load_x(t0, xa, 0, NXI);
    load_a(t0, ra);
  }
#pragma unroll
  for (int i = 0; i < NXI; ++i)
#pragma unroll
    for (int hp = 0; hp < XH; ++hp) store_half(xa, i, hp, buf0);
#pragma unroll
  for (int j = 0; j < 5; ++j) a_pair(ra, 0, j, a0);
#pragma unroll
  for (int p = 0; p < NP; ++p) a_finish(p, a0);
  {
    const Tile t1 = decode(split + splits);
    load_x(t1, xa, 0, NXI);
    load_a(t1, rb);
  }
  __syncthreads();

  // coordinates of the tile two ahead, advanced by `splits` tiles per body with carries (a division-free update on the
  // scalar unit)
  Tile tq = decode(split + 2 * splits);
  const int adv_w = splits % tiles_w, adv_h = (splits / tiles_w) % tiles_h, adv_n = splits / tiles_w / tiles_h;
  auto advance = [&](Tile& t) {
    t.tw += adv_w;
    const int cw = t.tw >= tiles_w;
    t.tw -= cw ? tiles_w : 0;
    t.th += adv_h + cw;
    const int ch = t.th >= tiles_h;
    t.th -= ch ? tiles_h : 0;
    t.n += adv_n + ch;
    t.ok = t.n < d.N;
  };
  auto body = [&](const unsigned* bufc, unsigned* bufn, ARaw& rc, const ARaw& rn) {
    const Tile tile2 = tq;
    advance(tq);
    // item `it` of half `h`
    auto item = [&](auto h_, auto it_) {
      constexpr int h = decltype(h_)::value, it = decltype(it_)::value;
      if constexpr (h == 0) {
        if constexpr (it < 5) { if constexpr (!(W3_ABL & 2)) a_pair(rc, 1, it, a1); }
        else if constexpr (it < 5 + NP) { if constexpr (!(W3_ABL & 2)) a_finish(it - 5, a1); }
        else if constexpr (it == 5 + NP) { if constexpr (!(W3_ABL & 4)) load_a(tile2, rc); }
        else if constexpr (it < L0 - 1) { if constexpr (!(W3_ABL & 1)) store_half(xa, (it - 6 - NP) / XH, (it - 6 - NP) % XH, bufn); }
        else { if constexpr (!(W3_ABL & 4)) load_x(tile2, xa, 0, NLO); }
      } else {
        if constexpr (it < 5) { if constexpr (!(W3_ABL & 2)) a_pair(rn, 0, it, a0); }
        else if constexpr (it < 5 + NP) { if constexpr (!(W3_ABL & 2)) a_finish(it - 5, a0); }
        else if constexpr (it < L1 - 1) { if constexpr (!(W3_ABL & 1)) store_half(xa, NLO + (it - 5 - NP) / XH, (it - 5 - NP) % XH, bufn); }
        else { if constexpr (!(W3_ABL & 4)) load_x(tile2, xa, NLO, NXI); }
      }
    };
    BFrag bq[2];
    read_b(bufc, 0, bq[0]);
    static_for<0, NS>([&](auto st_) {
      constexpr int st = decltype(st_)::value;
      constexpr int q = st / NTB, t = st - q * NTB;
      if constexpr (st + 1 < NS && !(W3_ABL & 16)) read_b(bufc, st + 1, bq[(st + 1) & 1]);
      const BFrag& b = bq[st & 1];
      static_for<0, NTERM>([&](auto term_) {
        constexpr int term = decltype(term_)::value;
        const AQ& f = q == 0 ? a0 : a1;
        constexpr int pa = NAT ? 0 : H2 ? HA_[term] : TA_[term + DLIO_SPLIT_Q0];
        constexpr int pb = NAT ? 0 : H2 ? HB_[term] : TB_[term + DLIO_SPLIT_Q0];
        static_for<0, 3>([&](auto kx_) {
          constexpr int kx = decltype(kx_)::value;
#pragma unroll
          for (int m = 0; m < MR; ++m) {
            const unsigned* src = kx == 2 ? f.e[pa][m] : kx == 0 ? f.c[pa][m] : f.o[pa][m];
            const u32x4 av = {src[0], src[1], src[2], src[3]};
            if constexpr (W3_ABL & 8) {
            } else if constexpr (H2) {
              acc[m][t][kx] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, av), __builtin_bit_cast(f16x8, b.v[pb]),
                                                                   acc[m][t][kx], 0, 0, 0);
            } else {
              acc[m][t][kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), b.v[pb], acc[m][t][kx], 0, 0, 0);
            }
          }
        });
        // the items of this unit (normally one; the bf16 variant has fewer units than items)
        constexpr int u = t * NTERM + term;              // unit inside the half
        constexpr int L = q == 0 ? L0 : L1;
        static_for<0, L>([&](auto it_) {
          constexpr int it = decltype(it_)::value;
          if constexpr ((L <= UH ? it : it * UH / L) == u) item(IC<q>{}, it_);
        });
        // an MFMA occupies its pipe for 32 cycles = 8 issue slots: deal the item's instructions evenly behind the
        // unit's MFMAs (three in a row followed by 15 VALU leave the pipe idle for half the time)
#if W3_SGB
#pragma unroll
        for (int k = 0; k < 3 * MR; ++k) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, W3_SGB, 0);
          __builtin_amdgcn_sched_group_barrier(0x320, 1, 0);
        }
        // nothing crosses a unit: the groups above would otherwise pull later items' instructions forward -- and with them
        // the s_waitcnt on loads that were meant to have half a tile of MFMAs to land
        __builtin_amdgcn_sched_barrier(0);
#endif
      });
    });
    if constexpr (!(W3_ABL & 32)) __syncthreads();
  };
  // two tiles per trip and no exit in the middle (with an exit between the bodies the accumulators get different registers
  // on the two paths and 144 v_accvgpr_mov per trip): an odd tile count runs one all-zero tile
  for (int it = 0; it < ntiles; it += 2) {
    body(buf0, buf1, ra, rb);
    body(buf1, buf0, rb, ra);
  }

  // ---- the four waves' accumulators meet in LDS: waves 2, 3 park theirs in two regions (16-byte stores), waves 0, 1 add
  // them to their own and write the sums back, every thread then adds the two regions while it writes the slab --
  // a fixed order ((w0 + w2) + (w1 + w3)), two barriers, and no more LDS than the pipeline's two patch buffers
  float* red = reinterpret_cast<float*>(smem);
  {
    float* mine = red + (wave & 1) * C::RED;
    auto slot = [&](int m, int t, int k, int r4) { return mine + ((((m * NTB + t) * 3 + k) * 4 + r4) * 64 + lane) * 4; };
    if (wave >= 2) {
#pragma unroll
      for (int m = 0; m < MR; ++m)
#pragma unroll
        for (int t = 0; t < NTB; ++t)
#pragma unroll
          for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4)
              *reinterpret_cast<f32x4*>(slot(m, t, k, r4)) =
                  f32x4{acc[m][t][k][4 * r4], acc[m][t][k][4 * r4 + 1], acc[m][t][k][4 * r4 + 2], acc[m][t][k][4 * r4 + 3]};
    }
    __syncthreads();
    if (wave < 2) {
#pragma unroll
      for (int m = 0; m < MR; ++m)
#pragma unroll
        for (int t = 0; t < NTB; ++t)
#pragma unroll
          for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
              f32x4* p = reinterpret_cast<f32x4*>(slot(m, t, k, r4));
              const f32x4 o = *p;
              *p = f32x4{acc[m][t][k][4 * r4] + o[0], acc[m][t][k][4 * r4 + 1] + o[1], acc[m][t][k][4 * r4 + 2] + o[2],
                         acc[m][t][k][4 * r4 + 3] + o[3]};
            }
    }
  }
  __syncthreads();
  const size_t row_len = (size_t)d.Cin * 9;
  float* out = wsp + (size_t)split * d.Cout * row_len;
  const int nch = min(C::CKC, d.Cin - c0);
  const float isc = (1.f / xs) * (1.f / ys);             // H2: powers of two (1 otherwise)
  for (int idx = tid; idx < 32 * MR * nch * 9; idx += 256) {
    const int tap = idx % 9;
    const int ch = (idx / 9) % nch;
    const int col = idx / (9 * nch);
    if (co0 + col >= d.Cout) continue;
    const int ky = tap / 3, kx = tap - 3 * ky;
    const int s = 3 * (ch >> 4) + ky, t = s >> 1;
    const int bl = 16 * (s & 1) + (ch & 15);             // B lane
    const int m = col >> 5, row = col & 31;
    const int hf = (row >> 2) & 1, r = (row & 3) + 4 * (row >> 3);
    const int i = ((((m * NTB + t) * 3 + kx) * 4 + (r >> 2)) * 64 + hf * 32 + bl) * 4 + (r & 3);
    out[(size_t)(co0 + col) * row_len + (size_t)(c0 + ch) * 9 + tap] =
        H2 ? (red[i] + red[C::RED + i]) * isc : red[i] + red[C::RED + i];
  }
}

template <int MR, int NTB, bool NAT, bool H2 = false>
int launch_w3(const void* x, const void* dy, float* wsp, const DlioConvDesc& d, const DlioWgrad3Plan& p, hipStream_t s,
              const float* amax_x = nullptr, const float* amax_dy = nullptr) {
  using C = W3Cfg<MR, NTB, NAT, H2>;
  auto k = wgrad3_kernel<MR, NTB, NAT, H2>;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES);
  hipLaunchKernelGGL(k, dim3(p.co_tiles * p.ci_chunks * p.splits), dim3(256), C::LDS_BYTES, s,
                     reinterpret_cast<const float*>(x), reinterpret_cast<const float*>(dy), wsp, d, p.co_tiles,
                     p.ci_chunks, p.splits, p.tiles_w, p.tiles_h, amax_x, amax_dy);
  return dlio_check_launch();
}

}  // namespace

bool dlio_wgrad3_plan(const DlioConvDesc& d, int elem_bytes, DlioWgrad3Plan& p) {
  static const int off = getenv("DLIO_WGRAD3") ? atoi(getenv("DLIO_WGRAD3")) == 0 : 0;
  if (off) return false;
  if (d.KH != 3 || d.KW != 3 || d.SH != 1 || d.SW != 1 || d.PH != 1 || d.PW != 1) return false;
  if (d.OH != d.H || d.OW != d.W) return false;
  const int al = 16 / elem_bytes;                          // columns per aligned 16-byte load
  if (d.W % al) return false;
  if ((size_t)d.N * d.in_ctot * d.H * d.W * elem_bytes >= 0xffffff00ull ||
      (size_t)d.N * d.out_ctot * d.OH * d.OW * elem_bytes >= 0xffffff00ull)
    return false;
  // tile shapes: <= 16 input channels: one 16-channel chunk (NTB = 2), 64-channel output tiles when there are that many;
  // otherwise 32-channel chunks x 32-channel tiles
  if (d.Cin <= 16) { p.ntb = 2; p.mr = 1; }
  else { p.ntb = 3; p.mr = 1; }
  const int ckc = 16 * ((2 * p.ntb) / 3);
  p.co_tiles = cdiv(d.Cout, 32 * p.mr);
  p.ci_chunks = cdiv(d.Cin, ckc);
  p.tiles_w = cdiv(d.OW, 32);
  p.tiles_h = cdiv(d.OH, 4);
  const int64_t total_tiles = (int64_t)d.N * p.tiles_w * p.tiles_h;
  const int64_t pairs = (int64_t)p.co_tiles * p.ci_chunks;
  static const int tgt = dlio_num_cus();
  int64_t splits = tgt / pairs > 0 ? tgt / pairs : 1;
  if (splits > total_tiles) splits = total_tiles;
  const size_t slab = (size_t)d.Cout * d.Cin * 9 * 4;
  const size_t cap = (size_t)96 << 20;
  if (splits * slab > cap) splits = (int64_t)(cap / slab);
  if (splits < 1) splits = 1;
  p.splits = (int)splits;
  p.ws_bytes = (size_t)p.splits * slab;
  return true;
}

int dlio_wgrad3_launch(const void* x, const void* dy, float* wsp, const DlioConvDesc& d, const DlioWgrad3Plan& p,
                       int elem_bytes, hipStream_t s, const float* amax_x, const float* amax_dy) {
  if (((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy)) & 15) != 0) return DLIO_EUNSUP;
  if (amax_x || amax_dy) {                               // two-piece fp16 split: fp32 operands with their largest magnitudes
    if (elem_bytes != 4 || !amax_x || !amax_dy) return DLIO_EINVAL;
    if (p.ntb == 2) return launch_w3<1, 2, false, true>(x, dy, wsp, d, p, s, amax_x, amax_dy);
    return launch_w3<1, 3, false, true>(x, dy, wsp, d, p, s, amax_x, amax_dy);
  }
  if (elem_bytes == 4) {
    if (p.ntb == 2) return launch_w3<1, 2, false>(x, dy, wsp, d, p, s);
    return launch_w3<1, 3, false>(x, dy, wsp, d, p, s);
  }
  if (p.ntb == 2) return launch_w3<1, 2, true>(x, dy, wsp, d, p, s);
  return launch_w3<1, 3, true>(x, dy, wsp, d, p, s);
}

// timing probes compiled into this file (bit 0: DLIO_SPLIT_Q0); 0 in the product build, checked at load (dlio_build_probes)
int dlio_probe_wgrad3() { return ((DLIO_SPLIT_Q0) != 0 ? 1 : 0); }
