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
// Replaces the weight-gradient half of nn.Conv2d backward for pointseg_modules.py:103 (expand3x3),
// resnet.py BasicBlock conv3x3, base_net.py:55-71 conv3_1 / conv4_1 / conv5_1.
#include "common.h"
#include "wgrad3.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pk(bf16x2 v) { return __builtin_bit_cast(unsigned, v); }

__device__ __forceinline__ u32x4 load_b128(__amdgpu_buffer_rsrc_t rsrc, unsigned voffset) {
  return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voffset, 0, 0));
}
// NEVER write __builtin_bit_cast(float, vec[i]): hipcc 7.2 miscompiles a bit_cast whose operand is a vector-element
// lvalue -- it reads element 0 whatever i is (a 16-byte load then shrinks to one dword, replicated).  Passing the
// element by value first is what works.
__device__ __forceinline__ float as_f(unsigned u) { return __builtin_bit_cast(float, u); }

// (a, b) -> packed bf16 pairs (low half = a) of the three pieces a = h + m + l
__device__ __forceinline__ void split_pair(float a, float b, unsigned& h, unsigned& m, unsigned& l) {
  const bf16x2 hh = {(__bf16)a, (__bf16)b};
  const float ra = a - (float)hh[0], rb = b - (float)hh[1];
  const bf16x2 mm = {(__bf16)ra, (__bf16)rb};
  const float sa = ra - (float)mm[0], sb = rb - (float)mm[1];
  const bf16x2 ll = {(__bf16)sa, (__bf16)sb};
  h = pk(hh); m = pk(mm); l = pk(ll);
}

template <int MR, int NTB, bool NAT>
struct W3Cfg {
  static constexpr int TH = 4, TW = 32;
  static constexpr int NP = NAT ? 1 : 3;                 // bf16 planes
  static constexpr int G = (2 * NTB) / 3;                // 16-channel groups per chunk
  static constexpr int CKC = 16 * G;
  static constexpr int CS = 20;                          // dwords per (row, channel): 16 + 4 pad -> odd number of 16-B slots
  static constexpr int RS = CKC * CS;                    // row stride (RS / 4 = 0 mod 16: the two slots of a tile never collide)
  static constexpr int PS = (TH + 2) * RS;               // plane stride
  static constexpr int BUF = NP * PS;                    // dwords of one patch buffer
  static constexpr int RED = MR * NTB * 3 * 16 * 64;     // accumulator exchange at the end
  static constexpr int SM_DWORDS = 2 * BUF > RED ? 2 * BUF : RED;
  static constexpr size_t LDS_BYTES = (size_t)SM_DWORDS * 4;
  static constexpr int PPR = NAT ? 4 : 8;                // 16-byte pieces per (row, channel): 4 fp32 / 8 bf16 columns each
  static constexpr int XITEMS = (TH + 2) * CKC * PPR;
  static constexpr int NXI = (XITEMS + 255) / 256;       // per thread
};

template <int MR, int NTB, bool NAT>
__global__ __launch_bounds__(256, 1) void wgrad3_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                        float* __restrict__ wsp, DlioConvDesc d, int co_tiles,
                                                        int ci_chunks, int splits, int tiles_w, int tiles_h) {
  using C = W3Cfg<MR, NTB, NAT>;
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
  const size_t ohw = (size_t)d.OH * d.OW, HW = (size_t)d.H * d.W;
  constexpr unsigned OOB = 0xffffff00u;
  const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(x), 0, (int)((size_t)d.N * d.in_ctot * HW * EB), 0x00020000);
  const __amdgpu_buffer_rsrc_t arsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(dy), 0, (int)((size_t)d.N * d.out_ctot * ohw * EB), 0x00020000);

  // ---- X patch: rows oh0 - 1 .. oh0 + 4, columns iw0 .. iw0 + 31 (no column halo), CKC channels.
  // fp32: piece = 4 columns; thread -> (c4 = tid & 7, then channel, then row).  NAT: piece = 8 columns (c4 < 4).
  constexpr int PPR = C::PPR, NXI = C::NXI;
  u32x4 rx[NXI];
  auto load_x = [&](int tile) {
    int tt = tile;
    const int tw = tt % tiles_w; tt /= tiles_w;
    const int th = tt % tiles_h; tt /= tiles_h;
    const int n = tt;
    const unsigned img = (unsigned)(((size_t)n * d.in_ctot + d.in_coff + c0) * HW * EB);     // uniform
#pragma unroll
    for (int i = 0; i < NXI; ++i) {
      const int it = tid + 256 * i;
      const int c4 = it % PPR, ch = (it / PPR) % C::CKC, r = it / (PPR * C::CKC);
      const int ih = th * C::TH - 1 + r, iw = tw * C::TW + c4 * (32 / PPR);
      const bool v = it < C::XITEMS && c0 + ch < d.Cin && ih >= 0 && ih < d.H && iw < d.W;
      const unsigned vo = v ? img + (unsigned)(((size_t)ch * HW + (size_t)ih * d.W + iw) * EB) : OOB;
      rx[i] = load_b128(xrsrc, vo);
    }
  };
  auto store_x = [&](unsigned* buf) {
#pragma unroll
    for (int i = 0; i < NXI; ++i) {
      const int it = tid + 256 * i;
      const int c4 = it % PPR, ch = (it / PPR) % C::CKC, r = it / (PPR * C::CKC);
      if (C::XITEMS % 256 != 0 && it >= C::XITEMS) continue;
      if constexpr (NAT) {
        *reinterpret_cast<u32x4*>(buf + r * C::RS + ch * C::CS + 4 * c4) = rx[i];
      } else {
        const float v0 = as_f(rx[i][0]), v1 = as_f(rx[i][1]);
        const float v2 = as_f(rx[i][2]), v3 = as_f(rx[i][3]);
        unsigned h0, m0, l0, h1, m1, l1;
        split_pair(v0, v1, h0, m0, l0);
        split_pair(v2, v3, h1, m1, l1);
        const u32x2 h = {h0, h1}, m = {m0, m1}, l = {l0, l1};
        unsigned* p = buf + r * C::RS + ch * C::CS + 2 * c4;
        *reinterpret_cast<u32x2*>(p) = h;
        *reinterpret_cast<u32x2*>(p + C::PS) = m;
        *reinterpret_cast<u32x2*>(p + 2 * C::PS) = l;
      }
    }
  };

  // ---- dY: lane (co, half) of wave w: row oh0 + w, columns iw0 + 16 half - 1 .. + 16
  constexpr int NAL = NAT ? 2 : 4;                       // aligned 16-byte loads of the 16 columns
  struct ARaw { u32x4 v[MR][NAL]; unsigned hl[MR], hr[MR]; };
  auto load_a = [&](int tile, ARaw& a) {
    int tt = tile;
    const int tw = tt % tiles_w; tt /= tiles_w;
    const int th = tt % tiles_h; tt /= tiles_h;
    const int n = tt;
    const int oh = th * C::TH + wave, ow = tw * C::TW + half * 16;
    const unsigned img = (unsigned)(((size_t)n * d.out_ctot + d.out_coff + co0) * ohw * EB);   // uniform
    const unsigned row = (unsigned)(((size_t)l31 * ohw + (size_t)oh * d.OW + ow) * EB);
#pragma unroll
    for (int m = 0; m < MR; ++m) {
      const bool vm = co0 + m * 32 + l31 < d.Cout && oh < d.OH;
      const unsigned base = img + row + (unsigned)((size_t)m * 32 * ohw * EB);
#pragma unroll
      for (int q = 0; q < NAL; ++q) {
        const bool v = vm && ow + (16 / NAL) * q < d.OW;
        a.v[m][q] = load_b128(arsrc, v ? base + 16u * q : OOB);
      }
      const bool vl = vm && ow - 1 >= 0 && ow - 1 < d.OW, vr = vm && ow + 16 < d.OW;
      if constexpr (NAT) {
        a.hl[m] = (unsigned)__builtin_amdgcn_raw_buffer_load_b16(arsrc, vl ? base - 2u : OOB, 0, 0);
        a.hr[m] = (unsigned)__builtin_amdgcn_raw_buffer_load_b16(arsrc, vr ? base + 32u : OOB, 0, 0);
      } else {
        a.hl[m] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(arsrc, vl ? base - 4u : OOB, 0, 0);
        a.hr[m] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(arsrc, vr ? base + 64u : OOB, 0, 0);
      }
    }
  };
  // packed fragments: E[p][m][0..8] (pairs (v0,v1) .. (v16,v17)), O[p][m][0..7] (pairs (v1,v2) .. (v15,v16))
  struct AFrag { unsigned E[NP][MR][9], O[NP][MR][8]; };
  auto prep_a = [&](const ARaw& a, AFrag& f) {
#pragma unroll
    for (int m = 0; m < MR; ++m) {
      if constexpr (NAT) {
        // the loaded dwords are the aligned pairs (v1,v2) ... (v15,v16) = O; E is the funnel shift
#pragma unroll
        for (int j = 0; j < 8; ++j) f.O[0][m][j] = a.v[m][j >> 2][j & 3];
        f.E[0][m][0] = (a.hl[m] & 0xffffu) | (f.O[0][m][0] << 16);
#pragma unroll
        for (int j = 1; j < 8; ++j) f.E[0][m][j] = __builtin_amdgcn_alignbit(f.O[0][m][j], f.O[0][m][j - 1], 16);
        f.E[0][m][8] = (f.O[0][m][7] >> 16) | (a.hr[m] << 16);
      } else {
        float v[18];
        v[0] = as_f(a.hl[m]);
#pragma unroll
        for (int j = 0; j < 16; ++j) v[1 + j] = as_f(a.v[m][j >> 2][j & 3]);
        v[17] = as_f(a.hr[m]);
#pragma unroll
        for (int j = 0; j < 9; ++j) split_pair(v[2 * j], v[2 * j + 1], f.E[0][m][j], f.E[1][m][j], f.E[2][m][j]);
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
          for (int j = 0; j < 8; ++j) f.O[p][m][j] = __builtin_amdgcn_alignbit(f.E[p][m][j + 1], f.E[p][m][j], 16);
      }
    }
  };
  auto a_op = [&](const AFrag& f, int p, int m, int kx, int q) -> bf16x8 {
    u32x4 r;
    if (kx == 1) r = u32x4{f.O[p][m][4 * q], f.O[p][m][4 * q + 1], f.O[p][m][4 * q + 2], f.O[p][m][4 * q + 3]};
    else {
      const int b = 4 * q + (kx == 0 ? 1 : 0);
      r = u32x4{f.E[p][m][b], f.E[p][m][b + 1], f.E[p][m][b + 2], f.E[p][m][b + 3]};
    }
    return __builtin_bit_cast(bf16x8, r);
  };

  // six products, smallest first: (lo,hi) (mid,mid) (hi,lo) (mid,hi) (hi,mid) (hi,hi)   [A plane, B plane]
  constexpr int TA[6] = {2, 1, 0, 1, 0, 0}, TB[6] = {0, 1, 2, 0, 1, 0};
  constexpr int NTERM = NAT ? 1 : 6;

  auto mfma_half = [&](const unsigned* buf, const AFrag& f, int q) {
    const char* bp = reinterpret_cast<const char*>(buf) + 16 * q;
    bf16x8 b[NTB][NP];
#pragma unroll
    for (int t = 0; t < NTB; ++t)
#pragma unroll
      for (int p = 0; p < NP; ++p)
        b[t][p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(bp + boff[t] + (size_t)p * C::PS * 4));
#ifdef W3_DEBUG
    if (blockIdx.x == 0 && wave == 0 && (lane == W3_DEBUG || lane == 20) && !NAT) {
      for (int t = 0; t < NTB; ++t)
        for (int p = 0; p < NP; ++p) {
          const u32x4 v = __builtin_bit_cast(u32x4, b[t][p]);
          printf("lane %d q %d B[t=%d][p=%d] = %08x %08x %08x %08x (boff %u)\n", lane, q, t, p, v[0], v[1], v[2], v[3], boff[t]);
        }
      for (int p = 0; p < NP; ++p) {
        printf("lane %d q %d E[p=%d] =", lane, q, p);
        for (int j = 0; j < 9; ++j) printf(" %08x", f.E[p][0][j]);
        printf(" | O =");
        for (int j = 0; j < 8; ++j) printf(" %08x", f.O[p][0][j]);
        printf("\n");
      }
    }
#endif
#pragma unroll
    for (int t = 0; t < NTB; ++t)
#pragma unroll
      for (int term = 0; term < NTERM; ++term)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
          for (int m = 0; m < MR; ++m)
            acc[m][t][kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_op(f, NAT ? 0 : TA[term], m, kx, q),
                                                                  b[t][NAT ? 0 : TB[term]], acc[m][t][kx], 0, 0, 0);
  };

  // ---- pipeline
  ARaw ra;
  AFrag fa, fb;
  unsigned* buf0 = smem;
  unsigned* buf1 = smem + C::BUF;
  int tile = split;
  if (tile < total_tiles) {
    load_x(tile);
    load_a(tile, ra);
    store_x(buf0);
    prep_a(ra, fa);
    if (tile + splits < total_tiles) { load_x(tile + splits); load_a(tile + splits, ra); }
  }
  __syncthreads();
  // two tiles per trip: (buf0, fa) then (buf1, fb)
  while (tile < total_tiles) {
    {
      const bool more = tile + splits < total_tiles;
      mfma_half(buf0, fa, 0);
      if (more) { store_x(buf1); prep_a(ra, fb); }
      if (tile + 2 * splits < total_tiles) { load_x(tile + 2 * splits); load_a(tile + 2 * splits, ra); }
      mfma_half(buf0, fa, 1);
      __syncthreads();
      if (!more) break;
      tile += splits;
    }
    {
      const bool more = tile + splits < total_tiles;
      mfma_half(buf1, fb, 0);
      if (more) { store_x(buf0); prep_a(ra, fa); }
      if (tile + 2 * splits < total_tiles) { load_x(tile + 2 * splits); load_a(tile + 2 * splits, ra); }
      mfma_half(buf1, fb, 1);
      __syncthreads();
      if (!more) break;
      tile += splits;
    }
  }

  // ---- sum the four waves' accumulators through LDS (fixed order), write the slab
  float* red = reinterpret_cast<float*>(smem);
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int m = 0; m < MR; ++m)
#pragma unroll
        for (int t = 0; t < NTB; ++t)
#pragma unroll
          for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int i = (((m * NTB + t) * 3 + k) * 16 + r) * 64 + lane;
              if (w == 0) red[i] = acc[m][t][k][r];
              else red[i] += acc[m][t][k][r];
            }
    }
    __syncthreads();
  }
  const size_t row_len = (size_t)d.Cin * 9;
  float* out = wsp + (size_t)split * d.Cout * row_len;
  const int nch = min(C::CKC, d.Cin - c0);
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
    out[(size_t)(co0 + col) * row_len + (size_t)(c0 + ch) * 9 + tap] =
        red[(((m * NTB + t) * 3 + kx) * 16 + r) * 64 + hf * 32 + bl];
  }
}

template <int MR, int NTB, bool NAT>
int launch_w3(const void* x, const void* dy, float* wsp, const DlioConvDesc& d, const DlioWgrad3Plan& p, hipStream_t s) {
  using C = W3Cfg<MR, NTB, NAT>;
  auto k = wgrad3_kernel<MR, NTB, NAT>;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES);
  hipLaunchKernelGGL(k, dim3(p.co_tiles * p.ci_chunks * p.splits), dim3(256), C::LDS_BYTES, s,
                     reinterpret_cast<const float*>(x), reinterpret_cast<const float*>(dy), wsp, d, p.co_tiles,
                     p.ci_chunks, p.splits, p.tiles_w, p.tiles_h);
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
  static const int tgt = getenv("DLIO_WGRAD3_BLOCKS") ? atoi(getenv("DLIO_WGRAD3_BLOCKS")) : dlio_num_cus();
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
                       int elem_bytes, hipStream_t s) {
  if (((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy)) & 15) != 0) return DLIO_EUNSUP;
  if (elem_bytes == 4) {
    if (p.ntb == 2) return launch_w3<1, 2, false>(x, dy, wsp, d, p, s);
    return launch_w3<1, 3, false>(x, dy, wsp, d, p, s);
  }
  if (p.ntb == 2) return launch_w3<1, 2, true>(x, dy, wsp, d, p, s);
  return launch_w3<1, 3, true>(x, dy, wsp, d, p, s);
}
