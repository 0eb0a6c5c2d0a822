// Implicit-GEMM 2-D convolution, fp32 in / fp32 accumulate, on v_mfma_f32_32x32x2_f32.
//
// GEMM orientation (chosen for NCHW): D[co][pixel] = sum_k W[co][k] * X[k][pixel]
//   A operand = weights, lane l holds W[co = l&31][k + (l>>5)]
//   B operand = im2col,  lane l holds X[k + (l>>5)][pixel = l&31]  -> 32 consecutive
//               pixels of one image row
//   D: lane = pixel column, 16 regs = 16 output channels -> every store instruction
//      writes 2 x 128 B contiguous row segments of the NCHW output.
// Two kernels:
//   conv_fwd_kernel        general taps/strides: input patch (halo, zero padding, optional fused
//                          producer BN-apply+ReLU) double-buffered in LDS with register prefetch
//                          of the next channel chunk; weights straight from L1/L2.
//   conv1x1_direct_kernel  1x1 stride 1: no LDS at all, both operands loaded from global with an
//                          explicit two-stage register pipeline.
//
// Replaces nn.Conv2d forward (and, with mode-1 prepped weights, the stride-1 data
// gradient) on the reference path: pointseg_net.py:18, pointseg_modules.py:96-106,
// base_net.py:55-71, resnet.py:36, lidar_feat_nets.py:279-304.
#include "common.h"
#include <stdlib.h>

namespace {

// bias + residual + NCHW store of one wave's MR x NR accumulator tiles.  Residual values are
// fetched for the whole tile BEFORE the first store so the loads are in flight together
// (a load->add->store chain per element serialises on memory latency).
template <int MR, int NR>
__device__ __forceinline__ void store_tiles(f32x16 (&acc)[MR][NR], const float* __restrict__ bias,
                                            const float* residual, float* y, const DlioConvDesc& d,
                                            int n, int co0, int half, bool (&pv)[NR],
                                            size_t (&pix)[NR], size_t plane) {
  if (residual) {
    const float* rb = residual + ((size_t)n * d.res_ctot + d.res_coff) * plane;
#pragma unroll
    for (int m = 0; m < MR; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        const int cc = co < d.Cout ? co : d.Cout - 1;
#pragma unroll
        for (int q = 0; q < NR; ++q) acc[m][q][r] += pv[q] ? rb[(size_t)cc * plane + pix[q]] : 0.f;
      }
  }
  float* yb = y + ((size_t)n * d.out_ctot + d.out_coff) * plane;
#pragma unroll
  for (int m = 0; m < MR; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (co >= d.Cout) continue;
      const float bv = bias ? bias[co] : 0.f;
#pragma unroll
      for (int q = 0; q < NR; ++q)
        if (pv[q]) yb[(size_t)co * plane + pix[q]] = acc[m][q][r] + bv;
    }
}

// Prepped-weight layout (both modes): wt[tap][k/8][k%2][n][(k/2)%4], k padded with zero rows to a
// multiple of 16.  A lane of the MFMA A operand (n = output channel, half = k%2) finds the weights
// of four consecutive k-steps (k = 8g + 2u + half, u = 0..3) in ONE 16-byte load.
__host__ __device__ __forceinline__ size_t wt_index(int tap, int k, int n, int KPAD, int Nn) {
  return ((((size_t)tap * (KPAD >> 3) + (k >> 3)) * 2 + (k & 1)) * Nn + n) * 4 + ((k >> 1) & 3);
}

__device__ __forceinline__ float buf_load(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}

template <int KH, int KW, int SH, int SW, int CK, int TWN, int MR>
struct ConvCfg {
  static constexpr int TH = 4;                  // output rows per block (one per wave)
  static constexpr int TW = 32 * TWN;           // output cols per block
  static constexpr int CO_T = 32 * MR;          // output channels per block
  static constexpr int NR = TWN;                // pixel tiles per wave
  static constexpr int PR = (TH - 1) * SH + KH; // patch rows
  static constexpr int PC = (TW - 1) * SW + KW; // patch cols
  static constexpr int PLANE = PR * PC;
  static constexpr int NPOS = (PLANE + 255) / 256;   // patch positions per thread
  // LDS layout of the input patch.  CK >= 8: position-major [pos][half][k] (channel c = 2k + half)
  // with a row stride of CK + 4 floats, so that a lane of the B operand finds the values of
  // four consecutive k-steps in ONE ds_read_b128 (conflict-free: 20-float stride = 8 distinct
  // 4-bank groups) and the stores are ds_write_b128 as well.  CK < 8: channel-major planes.
  static constexpr bool PMAJOR = CK >= 8;
  static constexpr int PS = CK + 4;             // floats per position (position-major)
  static constexpr int XL = PMAJOR ? PLANE * PS : CK * PLANE;   // floats per LDS buffer
  static constexpr size_t LDS_BYTES = (size_t)2 * XL * 4;   // double buffered
};

// Pipeline per CK-channel chunk: the global loads of chunk i+1 are issued into registers, the
// MFMAs of chunk i run out of LDS buffer i&1 with the weight operand read straight from L1/L2
// (all four waves of a block read the same [k][co] rows), then the registers are written to
// buffer (i+1)&1 and ONE barrier closes the iteration.  A thread owns the same NPOS patch
// positions for every channel, so the per-element index arithmetic is done once per block.
template <int KH, int KW, int SH, int SW, int CK, int TWN, int MR, bool AFF>
__global__ __launch_bounds__(256, 2) void conv_fwd_kernel(
    const float* __restrict__ x, const float* __restrict__ wt, const float* __restrict__ bias,
    const float* __restrict__ in_mean, const float* __restrict__ in_scale,
    const float* __restrict__ in_shift, const float* residual, float* y, DlioConvDesc d,
    int tiles_w, int tiles_h, int co_tiles) {
  using C = ConvCfg<KH, KW, SH, SW, CK, TWN, MR>;
  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;       // = output row inside the tile
  const int l31 = lane & 31;
  const int half = lane >> 5;

  int bid = xcd_block_index();
  const int cot = bid % co_tiles; bid /= co_tiles;
  const int tw = bid % tiles_w;   bid /= tiles_w;
  const int th = bid % tiles_h;   bid /= tiles_h;
  const int n = bid;

  const int co0 = cot * C::CO_T;
  const int oh0 = th * C::TH;
  const int ow0 = tw * C::TW;
  const int ih0 = oh0 * SH - d.PH;
  const int iw0 = ow0 * SW - d.PW;
  const int Cin = d.Cin, Cout = d.Cout;
  const size_t HW = (size_t)d.H * d.W;

  f32x16 acc[MR][C::NR];
#pragma unroll
  for (int m = 0; m < MR; ++m)
#pragma unroll
    for (int q = 0; q < C::NR; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][q][r] = 0.f;

  // patch positions owned by this thread (same for every channel and chunk)
  int poff[C::NPOS];
  bool pval[C::NPOS];
#pragma unroll
  for (int j = 0; j < C::NPOS; ++j) {
    const int pos = tid + j * 256;
    const int r = pos / C::PC;
    const int col = pos - r * C::PC;
    const int ih = ih0 + r, iw = iw0 + col;
    pval[j] = pos < C::PLANE && ih >= 0 && ih < d.H && iw >= 0 && iw < d.W;
    poff[j] = pval[j] ? ih * d.W + iw : 0;
  }
  const float* xn = x + ((size_t)n * d.in_ctot + d.in_coff) * HW;
  const size_t xbytes = (size_t)Cin * HW * 4;
  const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(xn), 0, xbytes > 0xffffffffull ? -1 : (int)xbytes, 0x00020000);
  // clamped weight columns (rows >= Cout are dropped at the store)
  unsigned loff[MR];
#pragma unroll
  for (int m = 0; m < MR; ++m) loff[m] = 16u * (unsigned)(half * Cout + min(co0 + m * 32 + l31, Cout - 1));
  const int KP = (Cin + 15) & ~15;
  // buffer descriptors: per-lane 32-bit voffset + scalar soffset, no 64-bit vector addresses
  const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(wt), 0, (int)((size_t)KH * KW * KP * Cout * 4), 0x00020000);

  float reg[CK][C::NPOS];
  // branch-free: invalid positions read element 0 of a valid plane and are zeroed by a select
  auto load_chunk = [&](int c0) {
#pragma unroll
    for (int c = 0; c < CK; ++c) {
      const int ci = c0 + c;
      const bool cv = ci < Cin;
      const unsigned coff = (unsigned)(cv ? ci : 0) * (unsigned)HW * 4u;   // uniform
#pragma unroll
      for (int j = 0; j < C::NPOS; ++j) reg[c][j] = buf_load(xrsrc, (unsigned)poff[j] * 4u, coff);
    }
    if constexpr (AFF) {
#pragma unroll
      for (int c = 0; c < CK; ++c) {
        const int ci = min(c0 + c, Cin - 1);
        const float mu = in_mean[ci], sc = in_scale[ci], sh = in_shift[ci];
#pragma unroll
        for (int j = 0; j < C::NPOS; ++j) {
          float v = (reg[c][j] - mu) * sc + sh;
          reg[c][j] = d.in_relu ? fmaxf(v, 0.f) : v;
        }
      }
    }
#pragma unroll
    for (int c = 0; c < CK; ++c) {
      const bool cv = c0 + c < Cin;
#pragma unroll
      for (int j = 0; j < C::NPOS; ++j) reg[c][j] = (cv && pval[j]) ? reg[c][j] : 0.f;
    }
  };
  auto store_chunk = [&](float* Xl) {
    if constexpr (C::PMAJOR) {
#pragma unroll
      for (int j = 0; j < C::NPOS; ++j) {
        const int pos = tid + j * 256;
        if (pos < C::PLANE) {
          float* dst = Xl + pos * C::PS;
#pragma unroll
          for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int g = 0; g < CK / 8; ++g)
              *reinterpret_cast<float4*>(dst + h * (CK / 2) + 4 * g) =
                  make_float4(reg[2 * (4 * g + 0) + h][j], reg[2 * (4 * g + 1) + h][j],
                              reg[2 * (4 * g + 2) + h][j], reg[2 * (4 * g + 3) + h][j]);
        }
      }
    } else {
#pragma unroll
      for (int c = 0; c < CK; ++c)
#pragma unroll
        for (int j = 0; j < C::NPOS; ++j) {
          const int pos = tid + j * 256;
          if (pos < C::PLANE) Xl[c * C::PLANE + pos] = reg[c][j];
        }
    }
  };
  // weight operands of one tap (CK/2 k-steps x MR tiles), fetched one tap ahead of their MFMAs.
  // Prepped weights are [tap][KP][Cout] with KP = Cin rounded up to 16 and zero rows behind Cin,
  // so no clamping: address = uniform (SGPR) row base + per-lane 32-bit byte offset.
  float aw[2][CK / 2][MR];
  auto load_tap = [&](int tap, int c0, int s) {
    // one 16-byte (CK >= 8), 8-byte (CK == 4) or 4-byte (CK == 2) load per 4 / 2 / 1 k-steps
    constexpr int VW = CK / 2 < 4 ? CK / 2 : 4;
    const unsigned base = (unsigned)((((tap * (KP >> 3) + (c0 >> 3)) * 2) * Cout) * 4 + ((c0 >> 1) & 3)) * 4u;
#pragma unroll
    for (int g = 0; g < (CK / 2 + 3) / 4; ++g)
#pragma unroll
      for (int m = 0; m < MR; ++m) {
        const unsigned so = base + (unsigned)(g * 2 * Cout * 4) * 4u;
        if constexpr (VW == 4) {
          const float4 v = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, loff[m], so, 0));
          aw[s][4 * g][m] = v.x; aw[s][4 * g + 1][m] = v.y; aw[s][4 * g + 2][m] = v.z; aw[s][4 * g + 3][m] = v.w;
        } else if constexpr (VW == 2) {
          const float2 v = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(wrsrc, loff[m], so, 0));
          aw[s][0][m] = v.x; aw[s][1][m] = v.y;
        } else {
          aw[s][0][m] = buf_load(wrsrc, loff[m], so);
        }
      }
  };
  // k-steps of a chunk in issue order: step = tap * CK/2 + c2/2.  The LDS operand of step s+1
  // is read before the MFMAs of step s (register double buffer), so the ~100-cycle LDS latency
  // overlaps 4 MFMAs instead of stalling in front of them.
  constexpr int KS = CK / 2, NSTEP = KH * KW * KS;
  auto load_b = [&](const float* Xl, int step, float (&b)[C::NR]) {
    const int tap = step / KS, c = (step % KS) * 2 + half;
    const int dy = tap / KW, dx = tap % KW;
    const float* xrow = Xl + (wave * SH + dy) * C::PC + dx + l31 * SW;
#pragma unroll
    for (int q = 0; q < C::NR; ++q) b[q] = xrow[c * C::PLANE + q * 32 * SW];
  };
  // position-major: all KS values of a tap for this lane in KS/4 reads of 16 bytes
  auto load_b_tap = [&](const float* Xl, int tap, float (&b)[C::NR][KS]) {
    const int dy = tap / KW, dx = tap % KW;
    const float* xrow = Xl + ((wave * SH + dy) * C::PC + dx + l31 * SW) * C::PS + half * KS;
#pragma unroll
    for (int q = 0; q < C::NR; ++q)
#pragma unroll
      for (int g = 0; g < KS / 4; ++g) {
        const float4 v = *reinterpret_cast<const float4*>(xrow + q * 32 * SW * C::PS + 4 * g);
        b[q][4 * g] = v.x; b[q][4 * g + 1] = v.y; b[q][4 * g + 2] = v.z; b[q][4 * g + 3] = v.w;
      }
  };
  // `prefetch` issues the global loads of the NEXT chunk.  It is called after the weight loads of
  // taps 0 and 1 are in flight: loads return in order, so weights queued BEHIND the 2*CK patch
  // loads would make the first MFMAs of every chunk wait for a full HBM round trip.
  auto mfma_chunk = [&](const float* Xl, int c0, auto&& prefetch) {
    if constexpr (C::PMAJOR) {
      // per tap: KS*MR*NR MFMAs against (KS/4)*NR LDS reads and (KS/4)*MR weight loads, all of
      // them for the NEXT tap (register double buffers)
      float b[2][C::NR][KS];
      load_tap(0, c0, 0);
      load_b_tap(Xl, 0, b[0]);
#pragma unroll
      for (int tap = 0; tap < KH * KW; ++tap) {
        if (tap + 1 < KH * KW) {
          load_b_tap(Xl, tap + 1, b[(tap + 1) & 1]);
          load_tap(tap + 1, c0, (tap + 1) & 1);
        }
        if (tap == 0) prefetch();
#pragma unroll
        for (int k = 0; k < KS; ++k) {
#pragma unroll
          for (int m = 0; m < MR; ++m)
#pragma unroll
            for (int q = 0; q < C::NR; ++q)
              acc[m][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(aw[tap & 1][k][m], b[tap & 1][q][k], acc[m][q], 0, 0, 0);
          if (k < (KS / 4) * C::NR) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          else if (k < (KS / 4) * (C::NR + MR)) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, MR * C::NR, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      return;
    }
    float b[2][C::NR];
    load_tap(0, c0, 0);
    load_b(Xl, 0, b[0]);
#pragma unroll
    for (int tap = 0; tap < KH * KW; ++tap) {
      if (tap + 1 < KH * KW) load_tap(tap + 1, c0, (tap + 1) & 1);
      if (tap == 0) prefetch();
#pragma unroll
      for (int k = 0; k < KS; ++k) {
        const int step = tap * KS + k;
        if (step + 1 < NSTEP) load_b(Xl, step + 1, b[(step + 1) & 1]);
#pragma unroll
        for (int m = 0; m < MR; ++m)
#pragma unroll
          for (int q = 0; q < C::NR; ++q)
            acc[m][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(aw[tap & 1][k][m], b[step & 1][q], acc[m][q], 0, 0, 0);
        // issue order inside the tap: next step's LDS operand, this step's share of the next
        // tap's weight loads, then this step's MFMAs
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        if (k < ((CK / 2 + 3) / 4) * MR) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, MR * C::NR, 0);
      }
      // keep the scheduler from hoisting every tap's weight loads to the top of the chunk
      // (245+ VGPRs, one wave per SIMD): exactly one tap of prefetch stays in flight
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  const int nch = (Cin + CK - 1) / CK;
  load_chunk(0);
  store_chunk(smem);
  __syncthreads();
  for (int i = 0; i < nch; ++i) {
    float* cur = smem + (i & 1) * C::XL;
    float* nxt = smem + ((i + 1) & 1) * C::XL;
    mfma_chunk(cur, i * CK, [&]() { if (i + 1 < nch) load_chunk((i + 1) * CK); });
    if (i + 1 < nch) store_chunk(nxt);
    __syncthreads();
  }

  const int oh = oh0 + wave;
  bool pv[C::NR];
  size_t pix[C::NR];
#pragma unroll
  for (int q = 0; q < C::NR; ++q) {
    const int ow = ow0 + q * 32 + l31;
    pv[q] = oh < d.OH && ow < d.OW;
    pix[q] = pv[q] ? (size_t)oh * d.OW + ow : 0;
  }
  store_tiles<MR, C::NR>(acc, bias, residual, y, d, n, co0, half, pv, pix, (size_t)d.OH * d.OW);
}

template <int KH, int KW, int SH, int SW, int CK, int TWN, int MR, bool AFF = false>
int launch(const float* x, const float* wt, const float* bias, const float* in_mean,
           const float* in_scale, const float* in_shift, const float* residual, float* y,
           const DlioConvDesc& d, hipStream_t s) {
  using C = ConvCfg<KH, KW, SH, SW, CK, TWN, MR>;
  const int tiles_w = cdiv(d.OW, C::TW), tiles_h = cdiv(d.OH, C::TH);
  const int co_tiles = cdiv(d.Cout, C::CO_T);
  const int64_t blocks = (int64_t)tiles_w * tiles_h * co_tiles * d.N;
  if (blocks <= 0 || blocks > 0x7fffffff) return DLIO_EINVAL;
  auto kern = conv_fwd_kernel<KH, KW, SH, SW, CK, TWN, MR, AFF>;
  dlio_set_max_lds(reinterpret_cast<const void*>(kern), (int)C::LDS_BYTES);
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), C::LDS_BYTES, s, x, wt, bias,
                     in_mean, in_scale, in_shift, residual, y, d, tiles_w, tiles_h, co_tiles);
  return dlio_check_launch();
}

// tile shape by problem size: 64-wide pixel tiles when the row is long enough, and 32-channel
// (MR=1) tiles when 64-channel tiles would leave the 256 CUs with less than ~2 workgroups each
template <int KH, int KW, int SH, int SW, int CK>
int launch_tw(const float* x, const float* wt, const float* bias, const float* in_mean,
              const float* in_scale, const float* in_shift, const float* residual, float* y,
              const DlioConvDesc& d, hipStream_t s) {
  // fused producer-affine on load (not on the headline path): one small-tile instantiation
  if (in_scale)
    return launch<KH, KW, SH, SW, CK, 1, 1, true>(x, wt, bias, in_mean, in_scale, in_shift, residual, y, d, s);
  static const int force_twn = 0;   // tuning knob
  // sweep (tools/conv_table.py, DLIO_CONV_TWN x DLIO_CONV_MR): 32-pixel tiles win everywhere
  // (fewer registers, more resident workgroups) except for the <=32-channel outputs of blk1
  const int twn = force_twn ? force_twn : ((d.OW > 32 && d.Cout <= 32) ? 2 : 1);
  const int64_t blocks2 = (int64_t)cdiv(d.OW, 32 * twn) * cdiv(d.OH, 4) * cdiv(d.Cout, 64) * d.N;
  static const int force_mr = 0;   // tuning knob
  // micro-bench (tools/bench_conv.py): 32-channel tiles win for Cout <= 32 (234 vs 414 us on the
  // blk1 expand3x3 data gradient) and whenever 64-channel tiles give < 2 workgroups per CU
  const bool small = force_mr ? force_mr == 1 : (blocks2 < 512 || d.Cout <= 32);
  if (twn == 2) {
    if (small) return launch<KH, KW, SH, SW, CK, 2, 1>(x, wt, bias, in_mean, in_scale, in_shift, residual, y, d, s);
    return launch<KH, KW, SH, SW, CK, 2, 2>(x, wt, bias, in_mean, in_scale, in_shift, residual, y, d, s);
  }
  if (small) return launch<KH, KW, SH, SW, CK, 1, 1>(x, wt, bias, in_mean, in_scale, in_shift, residual, y, d, s);
  return launch<KH, KW, SH, SW, CK, 1, 2>(x, wt, bias, in_mean, in_scale, in_shift, residual, y, d, s);
}

// ---------------------------------------------------------------------------------------
// 1x1, stride 1: pointwise GEMM over the flattened pixel axis.  No LDS and no barriers: the
// input is used exactly once per output-channel tile, so staging it would only add a round
// trip.  The MFMA B operand (2 channels x 32 consecutive pixels) is loaded straight from HBM
// as two coalesced 128-B rows per wave instruction, the A operand (weights, [Cin][Cout]) from
// L1/L2.  Loads run one group of U k-steps ahead of the MFMAs in a second register set.
// Squeeze layers (Cout 16..80) are HBM-bound here, expand1x1 / squeeze-dgrad write-bound.
template <int MR, int NR, bool AFF>
__global__ __launch_bounds__(256) void conv1x1_direct_kernel(
    const float* __restrict__ x, const float* __restrict__ wt, const float* __restrict__ bias,
    const float* __restrict__ in_mean, const float* __restrict__ in_scale,
    const float* __restrict__ in_shift, const float* residual, float* y, DlioConvDesc d,
    int pix_blocks, int co_tiles) {
  constexpr int U = 4;   // k-steps (= 8 channels) per pipeline group
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int l31 = lane & 31;
  const int half = lane >> 5;
  int bid = xcd_block_index();
  const int cot = bid % co_tiles; bid /= co_tiles;
  const int pb = bid % pix_blocks; bid /= pix_blocks;
  const int n = bid;
  const int P = d.OH * d.OW;
  const int co0 = cot * 32 * MR;
  const int p0 = (pb * 4 + wave) * (32 * NR);
  if (p0 >= P) return;

  f32x16 acc[MR][NR];
#pragma unroll
  for (int m = 0; m < MR; ++m)
#pragma unroll
    for (int q = 0; q < NR; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][q][r] = 0.f;

  bool pv[NR];
  size_t pix[NR];
  const float* xq[NR];
  const float* xn = x + ((size_t)n * d.in_ctot + d.in_coff) * (size_t)P;
#pragma unroll
  for (int q = 0; q < NR; ++q) {
    const int p = p0 + q * 32 + l31;
    pv[q] = p < P;
    pix[q] = pv[q] ? p : P - 1;       // clamped address; the column is dropped at the store
    xq[q] = xn + pix[q];
  }
  const float* wq[MR];
#pragma unroll
  for (int m = 0; m < MR; ++m) wq[m] = wt + (size_t)min(co0 + m * 32 + l31, d.Cout - 1) * 4;
  const int Cin = d.Cin, Cout = d.Cout;

  float a[2][U][MR], b[2][U][NR];
  auto load_group = [&](int k0, int s) {
    // the U k-steps of a group are contiguous in the prepped layout: one 16-byte (U = 4) or
    // 8-byte (U = 2) load per output-channel tile; rows behind Cin are zero (k0 + 2U <= KP)
#pragma unroll
    for (int m = 0; m < MR; ++m) {
      const float* wp = wq[m] + wt_index(0, k0 + half, 0, 0, Cout);
      if constexpr (U == 4) {
        const float4 v = *reinterpret_cast<const float4*>(wp);
        a[s][0][m] = v.x; a[s][1][m] = v.y; a[s][2][m] = v.z; a[s][3][m] = v.w;
      } else {
        const float2 v = *reinterpret_cast<const float2*>(wp);
        a[s][0][m] = v.x; a[s][1][m] = v.y;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int kk = k0 + 2 * u + half;
      const bool kv = kk < Cin;
      const int kc = kv ? kk : Cin - 1;
#pragma unroll
      for (int q = 0; q < NR; ++q) {
        float v = xq[q][(size_t)kc * P];
        if (AFF) {
          v = (v - in_mean[kc]) * in_scale[kc] + in_shift[kc];
          if (d.in_relu) v = fmaxf(v, 0.f);
        }
        b[s][u][q] = kv ? v : 0.f;    // channel past Cin contributes nothing
      }
    }
  };
  auto mfma_group = [&](int s) {
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int m = 0; m < MR; ++m)
#pragma unroll
        for (int q = 0; q < NR; ++q)
          acc[m][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s][u][m], b[s][u][q], acc[m][q], 0, 0, 0);
  };

  const int ngroups = (Cin + 2 * U - 1) / (2 * U);
  load_group(0, 0);
  for (int g = 0; g < ngroups; g += 2) {
    if (g + 1 < ngroups) load_group((g + 1) * 2 * U, 1);
    mfma_group(0);
    if (g + 2 < ngroups) load_group((g + 2) * 2 * U, 0);
    if (g + 1 < ngroups) mfma_group(1);
  }
  store_tiles<MR, NR>(acc, bias, residual, y, d, n, co0, half, pv, pix, (size_t)P);
}

// float4 variant for P % 4 == 0: a wave owns 128 consecutive pixels and lane l loads/stores the
// four pixels 4l..4l+3 as one 16-byte access.  Component e of the float4 is the B operand of
// MFMA e, i.e. output tile e holds pixels {4j+e}; at the store the four tiles of a lane are
// again one float4 -> 512 contiguous bytes per half-wave, a quarter of the load/store
// instructions of the dword kernel.
template <int MR, bool AFF>
__global__ __launch_bounds__(256, MR == 2 ? 2 : 1) void conv1x1_v4_kernel(
    const float* __restrict__ x, const float* __restrict__ wt, const float* __restrict__ bias,
    const float* __restrict__ in_mean, const float* __restrict__ in_scale,
    const float* __restrict__ in_shift, const float* residual, float* y, DlioConvDesc d,
    int pix_blocks, int co_tiles) {
  constexpr int U = MR == 3 ? 2 : 4;   // k-steps per pipeline group
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int l31 = lane & 31;
  const int half = lane >> 5;
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

  f32x16 acc[MR][4];
#pragma unroll
  for (int m = 0; m < MR; ++m)
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][e][r] = 0.f;

  const float* xq = x + ((size_t)n * d.in_ctot + d.in_coff) * (size_t)P + pc;
  const float* wq[MR];
#pragma unroll
  for (int m = 0; m < MR; ++m) wq[m] = wt + (size_t)min(co0 + m * 32 + l31, d.Cout - 1) * 4;
  const int Cin = d.Cin, Cout = d.Cout;

  float a[2][U][MR];
  float4 b[2][U];
  float af[2][U][3];
  const bool relu_in = d.in_relu != 0;
  auto load_group = [&](int k0, int s) {
    // the U k-steps of a group are contiguous in the prepped layout: one 16-byte (U = 4) or
    // 8-byte (U = 2) load per output-channel tile; rows behind Cin are zero (k0 + 2U <= KP)
#pragma unroll
    for (int m = 0; m < MR; ++m) {
      const float* wp = wq[m] + wt_index(0, k0 + half, 0, 0, Cout);
      if constexpr (U == 4) {
        const float4 v = *reinterpret_cast<const float4*>(wp);
        a[s][0][m] = v.x; a[s][1][m] = v.y; a[s][2][m] = v.z; a[s][3][m] = v.w;
      } else {
        const float2 v = *reinterpret_cast<const float2*>(wp);
        a[s][0][m] = v.x; a[s][1][m] = v.y;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int kk = k0 + 2 * u + half;
      const bool kv = kk < Cin;
      const int kc = kv ? kk : Cin - 1;
      float4 v = *reinterpret_cast<const float4*>(xq + (size_t)kc * P);
      if (AFF) {
        // the constants travel with the operand and are applied when it is USED (mfma_group): applying
        // them here would wait for the load and serialise the two-deep prefetch
        af[s][u][0] = kv ? in_mean[kc] : 0.f; af[s][u][1] = kv ? in_scale[kc] : 0.f; af[s][u][2] = kv ? in_shift[kc] : 0.f;
      }
      if (!kv) v = make_float4(0.f, 0.f, 0.f, 0.f);
      b[s][u] = v;
    }
  };
  auto mfma_group = [&](int s) {
    if (AFF) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const float mu = af[s][u][0], sc = af[s][u][1], sh = af[s][u][2];
        float4 v = b[s][u];
        v.x = (v.x - mu) * sc + sh; v.y = (v.y - mu) * sc + sh;
        v.z = (v.z - mu) * sc + sh; v.w = (v.w - mu) * sc + sh;
        if (relu_in) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        b[s][u] = v;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int m = 0; m < MR; ++m) {
        acc[m][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s][u][m], b[s][u].x, acc[m][0], 0, 0, 0);
        acc[m][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s][u][m], b[s][u].y, acc[m][1], 0, 0, 0);
        acc[m][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s][u][m], b[s][u].z, acc[m][2], 0, 0, 0);
        acc[m][3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s][u][m], b[s][u].w, acc[m][3], 0, 0, 0);
      }
  };
  const int ngroups = (Cin + 2 * U - 1) / (2 * U);
  load_group(0, 0);
  for (int g = 0; g < ngroups; g += 2) {
    if (g + 1 < ngroups) load_group((g + 1) * 2 * U, 1);
    mfma_group(0);
    if (g + 2 < ngroups) load_group((g + 2) * 2 * U, 0);
    if (g + 1 < ngroups) mfma_group(1);
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
      if (co >= Cout) continue;
      const float bv = bias ? bias[co] : 0.f;
      float4 o = make_float4(acc[m][0][r] + bv, acc[m][1][r] + bv, acc[m][2][r] + bv, acc[m][3][r] + bv);
      if (rb) { o.x += rv[r].x; o.y += rv[r].y; o.z += rv[r].z; o.w += rv[r].w; }
      *reinterpret_cast<float4*>(yb + (size_t)co * plane) = o;
    }
  }
}

template <int MR>
int launch_1x1_v4(const float* x, const float* wt, const float* bias, const float* in_mean,
                  const float* in_scale, const float* in_shift, const float* residual, float* y,
                  const DlioConvDesc& d, hipStream_t s) {
  const int P = d.OH * d.OW;
  const int pix_blocks = cdiv(P, 512);
  const int co_tiles = cdiv(d.Cout, 32 * MR);
  const int64_t blocks = (int64_t)pix_blocks * co_tiles * d.N;
  if (blocks <= 0 || blocks > 0x7fffffff) return DLIO_EINVAL;
  if (in_scale)
    hipLaunchKernelGGL((conv1x1_v4_kernel<MR, true>), dim3((unsigned)blocks), dim3(256), 0, s, x, wt,
                       bias, in_mean, in_scale, in_shift, residual, y, d, pix_blocks, co_tiles);
  else
    hipLaunchKernelGGL((conv1x1_v4_kernel<MR, false>), dim3((unsigned)blocks), dim3(256), 0, s, x, wt,
                       bias, in_mean, in_scale, in_shift, residual, y, d, pix_blocks, co_tiles);
  return dlio_check_launch();
}

template <int MR, int NR>
int launch_1x1(const float* x, const float* wt, const float* bias, const float* in_mean,
               const float* in_scale, const float* in_shift, const float* residual, float* y,
               const DlioConvDesc& d, hipStream_t s) {
  const int P = d.OH * d.OW;
  const int pix_blocks = cdiv(P, 128 * NR);
  const int co_tiles = cdiv(d.Cout, 32 * MR);
  const int64_t blocks = (int64_t)pix_blocks * co_tiles * d.N;
  if (blocks <= 0 || blocks > 0x7fffffff) return DLIO_EINVAL;
  if (in_scale)
    hipLaunchKernelGGL((conv1x1_direct_kernel<MR, NR, true>), dim3((unsigned)blocks), dim3(256), 0, s,
                       x, wt, bias, in_mean, in_scale, in_shift, residual, y, d, pix_blocks, co_tiles);
  else
    hipLaunchKernelGGL((conv1x1_direct_kernel<MR, NR, false>), dim3((unsigned)blocks), dim3(256), 0, s,
                       x, wt, bias, in_mean, in_scale, in_shift, residual, y, d, pix_blocks, co_tiles);
  return dlio_check_launch();
}

// ---- 1x1, float4 pixels, K split over the four waves of a workgroup ---------------------------
// For squeeze-type layers (many input channels, few output channels, few pixels: blk3-5) the
// float4 kernel above has only one wave per SIMD in flight and the scalar kernel too few bytes per
// load.  Here all four waves of a workgroup share the same 128 pixels x 32*MR channels and each
// takes a quarter of the input channels (4x the waves, 16-byte loads); the partial accumulators
// are combined through LDS in a fixed order and every wave stores a quarter of the rows.
template <int MR, bool AFF>
__global__ __launch_bounds__(256, 2) void conv1x1_v4_splitk_kernel(
    const float* __restrict__ x, const float* __restrict__ wt, const float* __restrict__ bias,
    const float* __restrict__ in_mean, const float* __restrict__ in_scale,
    const float* __restrict__ in_shift, const float* residual, float* y, DlioConvDesc d,
    int pix_blocks, int co_tiles) {
  constexpr int U = 4;
  __shared__ float red[MR * 4 * 16 * 64];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int l31 = lane & 31;
  const int half = lane >> 5;
  int bid = xcd_block_index();
  const int cot = bid % co_tiles; bid /= co_tiles;
  const int pb = bid % pix_blocks; bid /= pix_blocks;
  const int n = bid;
  const int P = d.OH * d.OW;
  const int co0 = cot * 32 * MR;
  const int p = pb * 128 + 4 * l31;
  const bool pvalid = p < P;
  const size_t pc = pvalid ? p : 0;
  const int Cin = d.Cin, Cout = d.Cout;

  f32x16 acc[MR][4];
#pragma unroll
  for (int m = 0; m < MR; ++m)
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][e][r] = 0.f;

  const float* xq = x + ((size_t)n * d.in_ctot + d.in_coff) * (size_t)P + pc;
  const float* wq[MR];
#pragma unroll
  for (int m = 0; m < MR; ++m) wq[m] = wt + (size_t)min(co0 + m * 32 + l31, Cout - 1) * 4;

  float a[2][U][MR];
  float4 b[2][U];
  float af[2][U][3];
  const bool relu_in = d.in_relu != 0;
  auto load_group = [&](int k0, int s) {
#pragma unroll
    for (int m = 0; m < MR; ++m) {
      const float4 v = *reinterpret_cast<const float4*>(wq[m] + wt_index(0, k0 + half, 0, 0, Cout));
      a[s][0][m] = v.x; a[s][1][m] = v.y; a[s][2][m] = v.z; a[s][3][m] = v.w;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int kk = k0 + 2 * u + half;
      const bool kv = kk < Cin;
      const int kc = kv ? kk : Cin - 1;
      float4 v = *reinterpret_cast<const float4*>(xq + (size_t)kc * P);
      if (AFF) {   // applied at use time, see conv1x1_v4_kernel
        af[s][u][0] = kv ? in_mean[kc] : 0.f; af[s][u][1] = kv ? in_scale[kc] : 0.f; af[s][u][2] = kv ? in_shift[kc] : 0.f;
      }
      if (!kv) v = make_float4(0.f, 0.f, 0.f, 0.f);
      b[s][u] = v;
    }
  };
  auto mfma_group = [&](int s) {
    if (AFF) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const float mu = af[s][u][0], sc = af[s][u][1], sh = af[s][u][2];
        float4 v = b[s][u];
        v.x = (v.x - mu) * sc + sh; v.y = (v.y - mu) * sc + sh;
        v.z = (v.z - mu) * sc + sh; v.w = (v.w - mu) * sc + sh;
        if (relu_in) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        b[s][u] = v;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int m = 0; m < MR; ++m) {
        acc[m][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s][u][m], b[s][u].x, acc[m][0], 0, 0, 0);
        acc[m][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s][u][m], b[s][u].y, acc[m][1], 0, 0, 0);
        acc[m][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s][u][m], b[s][u].z, acc[m][2], 0, 0, 0);
        acc[m][3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s][u][m], b[s][u].w, acc[m][3], 0, 0, 0);
      }
  };
  // this wave's share of the channel groups (8 channels each)
  const int ngroups = (Cin + 2 * U - 1) / (2 * U);
  const int per = (ngroups + 3) / 4;
  const int g0 = wave * per, g1 = min(ngroups, g0 + per);
  if (g0 < g1) {
    load_group(g0 * 2 * U, 0);
    for (int g = g0; g < g1; g += 2) {
      if (g + 1 < g1) load_group((g + 1) * 2 * U, 1);
      mfma_group(0);
      if (g + 2 < g1) load_group((g + 2) * 2 * U, 0);
      if (g + 1 < g1) mfma_group(1);
    }
  }
  // fixed-order sum of the four partial tiles
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int m = 0; m < MR; ++m)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int i = ((m * 4 + e) * 16 + r) * 64 + lane;
            if (w == 0) red[i] = acc[m][e][r];
            else red[i] += acc[m][e][r];
          }
    }
    __syncthreads();
  }
  if (!pvalid) return;
  // wave w stores rows r = 4w .. 4w+3 of every 16-row group
  const size_t plane = (size_t)P;
  float* yb = y + ((size_t)n * d.out_ctot + d.out_coff) * plane + pc;
  const float* rb = residual ? residual + ((size_t)n * d.res_ctot + d.res_coff) * plane + pc : nullptr;
#pragma unroll
  for (int m = 0; m < MR; ++m)
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int r = 4 * wave + rr;
      const int co = co0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (co >= Cout) continue;
      const float bv = bias ? bias[co] : 0.f;
      float4 o = make_float4(red[((m * 4 + 0) * 16 + r) * 64 + lane] + bv, red[((m * 4 + 1) * 16 + r) * 64 + lane] + bv,
                             red[((m * 4 + 2) * 16 + r) * 64 + lane] + bv, red[((m * 4 + 3) * 16 + r) * 64 + lane] + bv);
      if (rb) {
        const float4 rv = *reinterpret_cast<const float4*>(rb + (size_t)co * plane);
        o.x += rv.x; o.y += rv.y; o.z += rv.z; o.w += rv.w;
      }
      *reinterpret_cast<float4*>(yb + (size_t)co * plane) = o;
    }
}

template <int MR>
int launch_1x1_splitk(const float* x, const float* wt, const float* bias, const float* in_mean,
                      const float* in_scale, const float* in_shift, const float* residual, float* y,
                      const DlioConvDesc& d, hipStream_t s) {
  const int P = d.OH * d.OW;
  const int pix_blocks = cdiv(P, 128);
  const int co_tiles = cdiv(d.Cout, 32 * MR);
  const int64_t blocks = (int64_t)pix_blocks * co_tiles * d.N;
  if (blocks <= 0 || blocks > 0x7fffffff) return DLIO_EINVAL;
  if (in_scale)
    hipLaunchKernelGGL((conv1x1_v4_splitk_kernel<MR, true>), dim3((unsigned)blocks), dim3(256), 0, s, x, wt,
                       bias, in_mean, in_scale, in_shift, residual, y, d, pix_blocks, co_tiles);
  else
    hipLaunchKernelGGL((conv1x1_v4_splitk_kernel<MR, false>), dim3((unsigned)blocks), dim3(256), 0, s, x, wt,
                       bias, in_mean, in_scale, in_shift, residual, y, d, pix_blocks, co_tiles);
  return dlio_check_launch();
}

template <int MR>
int launch_1x1_nr(const float* x, const float* wt, const float* bias, const float* in_mean,
                  const float* in_scale, const float* in_shift, const float* residual, float* y,
                  const DlioConvDesc& d, hipStream_t s) {
  // 64-pixel waves only when that still gives >= 4 waves per SIMD over the chip
  const int P = d.OH * d.OW;
  static const int use_v4 = 1;    // tuning knob
  const uintptr_t al = reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) |
                       reinterpret_cast<uintptr_t>(residual);
  const int64_t waves4 = (int64_t)cdiv(P, 128) * cdiv(d.Cout, 32 * MR) * d.N;
  if (use_v4 && (P & 3) == 0 && (al & 15) == 0 && waves4 >= 2048)
    return launch_1x1_v4<MR>(x, wt, bias, in_mean, in_scale, in_shift, residual, y, d, s);
  // under-filled float4 launch with a long K: split K over the workgroup's waves
  static const int splitk = 1;   // tuning knob
  if constexpr (MR <= 2) {
    if (splitk && use_v4 && (P & 3) == 0 && (al & 15) == 0 && d.Cin >= 256)   // sweep: 192-channel layers lose
      return launch_1x1_splitk<MR>(x, wt, bias, in_mean, in_scale, in_shift, residual, y, d, s);
    // an in-affine layer would otherwise fall to the dword kernel: the under-filled float4 launch is faster
    if (use_v4 && in_scale && (P & 3) == 0 && (al & 15) == 0 && waves4 >= 512)
      return launch_1x1_v4<MR>(x, wt, bias, in_mean, in_scale, in_shift, residual, y, d, s);
  }
  const int64_t waves2 = (int64_t)cdiv(d.OH * d.OW, 64) * cdiv(d.Cout, 32 * MR) * d.N;
  static const int force_nr = 0;   // tuning knob
  if (MR < 3 && (force_nr == 2 || (force_nr == 0 && waves2 >= 4096)))
    return launch_1x1<MR, 2>(x, wt, bias, in_mean, in_scale, in_shift, residual, y, d, s);
  return launch_1x1<MR, 1>(x, wt, bias, in_mean, in_scale, in_shift, residual, y, d, s);
}

// wt[tap][k][n], k padded with zero rows to KP = roundup(K, 16)
__global__ void prep_weight_kernel(const float* __restrict__ w, float* __restrict__ wt, int Cout,
                                   int Cin, int taps, int mode) {
  const int K = mode == 0 ? Cin : Cout, Nn = mode == 0 ? Cout : Cin;
  const int KPAD = (K + 15) & ~15;
  const int64_t total = (int64_t)taps * KPAD * Nn;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int nn = i % Nn;
    const int k = (i / Nn) % KPAD;
    const int tap = i / ((int64_t)Nn * KPAD);
    float v = 0.f;
    if (k < K) {
      if (mode == 0) v = w[((int64_t)nn * Cin + k) * taps + tap];              // k = ci, n = co
      else v = w[((int64_t)k * Cin + nn) * taps + (taps - 1 - tap)];           // k = co, n = ci
    }
    wt[wt_index(tap, k, nn, KPAD, Nn)] = v;
  }
}

// all convolution weights of a model in ONE launch (146 prep launches per training step before)
__global__ void prep_weights_batched_kernel(const DlioPrepItem* __restrict__ items, int n_items,
                                            int64_t total) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    int lo = 0, hi = n_items - 1;                 // last item with start <= i
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (items[mid].start <= i) lo = mid; else hi = mid - 1;
    }
    const DlioPrepItem it = items[lo];
    const int64_t e = i - it.start;
    const int K = it.mode == 0 ? it.Cin : it.Cout, Nn = it.mode == 0 ? it.Cout : it.Cin;
    const int KPAD = (K + 15) & ~15;
    const int nn = (int)(e % Nn);
    const int k = (int)((e / Nn) % KPAD);
    const int tap = (int)(e / ((int64_t)Nn * KPAD));
    float v = 0.f;
    if (k < K) {
      if (it.mode == 0) v = it.w[((int64_t)nn * it.Cin + k) * it.taps + tap];
      else v = it.w[((int64_t)k * it.Cin + nn) * it.taps + (it.taps - 1 - tap)];
    }
    it.wt[wt_index(tap, k, nn, KPAD, Nn)] = v;
  }
}

}  // namespace

extern "C" int dlio_conv2d_prep_weights_batched(const DlioPrepItem* items_dev, int n_items,
                                                int64_t total_floats, dlio_stream_t stream) {
  if (!items_dev || n_items <= 0 || total_floats <= 0) return DLIO_EINVAL;
  hipLaunchKernelGGL(prep_weights_batched_kernel, dim3(ew_grid(total_floats, 256)), dim3(256), 0,
                     as_stream(stream), items_dev, n_items, total_floats);
  return dlio_check_launch();
}

extern "C" size_t dlio_conv2d_prep_weight_floats(int Cout, int Cin, int KH, int KW, int mode) {
  if (Cout <= 0 || Cin <= 0 || KH <= 0 || KW <= 0 || (mode != 0 && mode != 1)) return 0;
  const int K = mode == 0 ? Cin : Cout, Nn = mode == 0 ? Cout : Cin;
  return (size_t)KH * KW * ((K + 15) & ~15) * Nn;
}

extern "C" int dlio_conv2d_prep_weight(const float* w, float* wt, int Cout, int Cin, int KH,
                                       int KW, int mode, dlio_stream_t stream) {
  if (!w || !wt || Cout <= 0 || Cin <= 0 || KH <= 0 || KW <= 0 || (mode != 0 && mode != 1))
    return DLIO_EINVAL;
  const int64_t total = (int64_t)dlio_conv2d_prep_weight_floats(Cout, Cin, KH, KW, mode);
  hipLaunchKernelGGL(prep_weight_kernel, dim3(ew_grid(total, 256)), dim3(256), 0,
                     as_stream(stream), w, wt, Cout, Cin, KH * KW, mode);
  return dlio_check_launch();
}

extern "C" int dlio_conv2d_fwd(const float* x, const float* wt, const float* bias,
                               const float* in_mean, const float* in_scale,
                               const float* in_shift, const float* residual, float* y,
                               const DlioConvDesc* dp, dlio_stream_t stream) {
  if (!x || !wt || !y || !dp) return DLIO_EINVAL;
  const DlioConvDesc& d = *dp;
  if (d.N <= 0 || d.Cin <= 0 || d.Cout <= 0 || d.H <= 0 || d.W <= 0) return DLIO_EINVAL;
  // output extent: the floor or the ceil form of the usual formula; a stride-1 convolution may also
  // ask for up to K-1 further rows / columns, which read the zero padding behind the input
  // (asymmetric padding: the phases of a strided data gradient, functional.conv_dgrad)
  const int oh_lo = (d.H + 2 * d.PH - d.KH) / d.SH + 1, ow_lo = (d.W + 2 * d.PW - d.KW) / d.SW + 1;
  const int oh_hi = d.SH == 1 ? oh_lo + d.KH - 1 : (d.H + 2 * d.PH - d.KH + d.SH - 1) / d.SH + 1;
  const int ow_hi = d.SW == 1 ? ow_lo + d.KW - 1 : (d.W + 2 * d.PW - d.KW + d.SW - 1) / d.SW + 1;
  if (d.OH < oh_lo || d.OH > oh_hi || d.OW < ow_lo || d.OW > ow_hi || d.OH < 1 || d.OW < 1) return DLIO_EINVAL;
  if (in_scale && (!in_mean || !in_shift)) return DLIO_EINVAL;
  hipStream_t s = as_stream(stream);
  const double flops = 2.0 * d.N * (double)d.OH * d.OW * d.Cout * (double)d.Cin * d.KH * d.KW;
  const double bytes = 4.0 * d.N * ((double)d.Cin * d.H * d.W + (double)d.Cout * d.OH * d.OW *
                                   (residual ? 2.0 : 1.0));
  // profiler kinds: 0 = multi-tap convolutions (MFMA-bound), 2 = 1x1 convolutions (HBM-bound)
  const int pkind = (d.KH == 1 && d.KW == 1) ? 2 : 0;
  dlio_prof_begin(pkind, s, flops, bytes);
  int rc = DLIO_EUNSUP;
  static const int ck8 = 1;   // tuning knob
#define CONV_CASE(kh, kw, sh, sw, ck)                                                        \
  if (d.KH == kh && d.KW == kw && d.SH == sh && d.SW == sw)                                  \
    rc = launch_tw<kh, kw, sh, sw, ck>(x, wt, bias, in_mean, in_scale, in_shift, residual, y, d, s);
  if (d.KH == 1 && d.KW == 1 && d.SH == 1 && d.SW == 1 && d.PH == 0 && d.PW == 0 && d.OH == d.H && d.OW == d.W) {
    static const int shortk_mr1 = 80;
    // short K (squeeze data gradient, expand1x1 forward: <= 80 input channels) is store-bound:
    // 32-channel tiles (twice the waves, half the registers) are 8-12 % faster there (sweep)
    if (d.Cout <= 32 || (shortk_mr1 && d.Cin <= shortk_mr1))
      rc = launch_1x1_nr<1>(x, wt, bias, in_mean, in_scale, in_shift, residual, y, d, s);
    else if (d.Cout > 64 && d.Cout <= 96)
      rc = launch_1x1_nr<3>(x, wt, bias, in_mean, in_scale, in_shift, residual, y, d, s);
    else rc = launch_1x1_nr<2>(x, wt, bias, in_mean, in_scale, in_shift, residual, y, d, s);
  }
  else CONV_CASE(1, 1, 1, 1, 16)
  else if (d.KH == 3 && d.KW == 3 && d.SH == 1 && d.SW == 1 && ck8 && d.Cin <= 16)
    rc = launch_tw<3, 3, 1, 1, 8>(x, wt, bias, in_mean, in_scale, in_shift, residual, y, d, s);
  else CONV_CASE(3, 3, 1, 1, 16)
  else CONV_CASE(3, 5, 1, 2, 4)
  else CONV_CASE(3, 5, 1, 1, 8)
  else CONV_CASE(5, 7, 1, 2, 2)
  else CONV_CASE(5, 7, 1, 1, 4)
  else CONV_CASE(3, 3, 2, 2, 8)
  else CONV_CASE(3, 3, 1, 2, 8)
  // tap subsets of the strided layers above (phases of their data gradients)
  else CONV_CASE(3, 2, 1, 1, 16)
  else CONV_CASE(3, 1, 1, 1, 16)
  else CONV_CASE(2, 2, 1, 1, 16)
  else CONV_CASE(2, 1, 1, 1, 16)
  else CONV_CASE(1, 2, 1, 1, 16)
  else CONV_CASE(1, 1, 1, 2, 16)
  else CONV_CASE(1, 1, 2, 2, 16)
#undef CONV_CASE
  dlio_prof_end(pkind, s);
  return rc;
}
