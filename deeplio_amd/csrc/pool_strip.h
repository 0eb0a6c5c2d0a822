// internal: the routed gradient of a strip of a 3x3 / stride (SH, 2) / padding 1 max-pool (shared by pool.hip's rolling-window
// backward kernels and bn.hip's BatchNorm + pool backward)
#pragma once
#include "common.h"

namespace {

constexpr int PR = 8;

template <int SH>
__device__ __forceinline__ void pool3_strip(const float* __restrict__ dyp,
                                            const uint8_t* __restrict__ ip, int ih0, int b, int OH,
                                            int OW, float (&G)[PR][4]) {
#pragma unroll
  for (int r = 0; r < PR; ++r) G[r][0] = G[r][1] = G[r][2] = G[r][3] = 0.f;
  const bool has2 = 2 * b + 2 < OW;
  // output rows whose 3-row window (input rows oh*SH-1 .. oh*SH+1) meets input rows ih0 .. ih0+PR-1
  constexpr int J0 = SH == 1 ? -1 : 0, J1 = SH == 1 ? PR : PR / 2;
  // loads first and unconditional (row clamped, halo column clamped to a valid address): under a
  // branch every load is its own HBM round trip
  constexpr int NJ = J1 - J0 + 1;
  float2 v01s[NJ];
  unsigned short k01s[NJ];
  float v2s[NJ];
  uint8_t k2s[NJ];
  const int c2 = has2 ? 2 : 1;
#pragma unroll
  for (int j = J0; j <= J1; ++j) {
    const int oh = min(max(ih0 / SH + j, 0), OH - 1);
    const size_t ro = (size_t)oh * OW + 2 * b;
    v01s[j - J0] = *reinterpret_cast<const float2*>(dyp + ro);
    k01s[j - J0] = *reinterpret_cast<const unsigned short*>(ip + ro);
    v2s[j - J0] = dyp[ro + c2];
    k2s[j - J0] = ip[ro + c2];
  }
#pragma unroll
  for (int j = J0; j <= J1; ++j) {
    const int oh = ih0 / SH + j;
    if (oh < 0 || oh >= OH) continue;
    const float2 v01 = v01s[j - J0];
    const unsigned short k01 = k01s[j - J0];
    const float v[3] = {v01.x, v01.y, has2 ? v2s[j - J0] : 0.f};
    const int k[3] = {k01 & 0xff, k01 >> 8, has2 ? (int)k2s[j - J0] : 4};   // 4 = (ky 1, kx 1): col 2b+2 -> none
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int ky = k[c] / 3, kx = k[c] - ky * 3;
#pragma unroll
      for (int kyv = 0; kyv < 3; ++kyv) {
        const int r = j * SH - 1 + kyv;            // input row inside the strip (compile-time)
        if (r < 0 || r >= PR) continue;
        const float val = ky == kyv ? v[c] : 0.f;
        if (c == 0) { G[r][0] += kx == 1 ? val : 0.f; G[r][1] += kx == 2 ? val : 0.f; }
        else if (c == 1) { G[r][1] += kx == 0 ? val : 0.f; G[r][2] += kx == 1 ? val : 0.f; G[r][3] += kx == 2 ? val : 0.f; }
        else if (has2) { G[r][3] += kx == 0 ? val : 0.f; }
      }
    }
  }
}

}  // namespace
