// pair_fuse_fc: the head of the siamese lidar feature nets (lidar_feat_nets.py:84-94 PointSeg, :131-141 FlowNet):
//   x = adaptive_avg_pool2d(enc1(xyz)) (+|-) adaptive_avg_pool2d(enc2(normals)); y = act(fc1(x))
// in ONE launch instead of four (two plane averages, the add / sub, the dense layer).  Workgroup (n, block of 32
// channels): its four waves average eight planes of each encoder output (16-byte loads, wave reductions), combine them,
// and every thread j < F adds the block's 32 channels into its partial dot product with row j of the weight; the partials
// go to a scratch buffer and the workgroup that arrives LAST at the image's counter (release / acquire at agent scope, once
// per workgroup) adds the blocks in index order, applies bias + activation and restores the counter -- a fixed summation
// order, so results are reproducible from run to run.
#include "common.h"

namespace {

constexpr int PF_CB = 32;      // channels per workgroup
constexpr int PF_T = 256;

__device__ __forceinline__ float pf_act(float v, int act) {
  switch (act) {
    case 1: return fmaxf(v, 0.f);
    case 2: return v > 0.f ? v : 0.01f * v;
    case 3: return 1.0f / (1.0f + expf(-v));
    case 4: return tanhf(v);
    default: return v;
  }
}

__global__ __launch_bounds__(PF_T) void pair_fuse_fc_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                                int C, int HW, int mode, const float* __restrict__ w,
                                                                const float* __restrict__ bias, int F, int act,
                                                                float* __restrict__ feat, float* __restrict__ y,
                                                                float* part, int* cnt) {
  __shared__ float fs[PF_CB];
  __shared__ int last;
  const int blocks = (C + PF_CB - 1) / PF_CB;
  const int n = blockIdx.x / blocks, cb = blockIdx.x - n * blocks;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c0 = cb * PF_CB;
  const float inv = 1.f / (float)HW;
  for (int k = wave; k < PF_CB; k += PF_T / 64) {
    const int c = c0 + k;
    float sa = 0.f, sb = 0.f;
    if (c < C) {
      const float* pa = a + ((size_t)n * C + c) * HW;
      const float* pb = b + ((size_t)n * C + c) * HW;
      if ((HW & 3) == 0) {
        for (int i = lane * 4; i < HW; i += 256) {
          const float4 va = *reinterpret_cast<const float4*>(pa + i), vb = *reinterpret_cast<const float4*>(pb + i);
          sa += (va.x + va.y) + (va.z + va.w);
          sb += (vb.x + vb.y) + (vb.z + vb.w);
        }
      } else {
        for (int i = lane; i < HW; i += 64) { sa += pa[i]; sb += pb[i]; }
      }
    }
    sa = wave_sum(sa); sb = wave_sum(sb);
    if (lane == 0) {
      const float ga = sa * inv, gb = sb * inv;
      const float f = c < C ? (mode == 0 ? ga + gb : ga - gb) : 0.f;
      fs[k] = f;
      if (c < C) feat[(size_t)n * C + c] = f;
    }
  }
  __syncthreads();
  for (int j = threadIdx.x; j < F; j += PF_T) {
    const float* wr = w + (size_t)j * C + c0;
    float acc = 0.f;
    const int kn = min(PF_CB, C - c0);
    for (int k = 0; k < kn; ++k) acc += wr[k] * fs[k];
    part[((size_t)n * blocks + cb) * F + j] = acc;
  }
  __syncthreads();
  if (threadIdx.x == 0)
    last = __hip_atomic_fetch_add(cnt + n, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == blocks - 1;
  __syncthreads();
  if (!last) return;
  for (int j = threadIdx.x; j < F; j += PF_T) {
    float acc = bias ? bias[j] : 0.f;
    const float* pp = part + (size_t)n * blocks * F + j;
    for (int k = 0; k < blocks; ++k) acc += __hip_atomic_load(pp + (size_t)k * F, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    y[(size_t)n * F + j] = pf_act(acc, act);
  }
  if (threadIdx.x == 0) __hip_atomic_store(cnt + n, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// da[n][c][:] = df[n][c] / HW, db[n][c][:] = (+|-) df[n][c] / HW  (backward of the two plane averages and the add / sub)
__global__ __launch_bounds__(256) void pair_fuse_bwd_kernel(const float* __restrict__ df, float* __restrict__ da,
                                                            float* __restrict__ db, int64_t planes, int HW, int mode) {
  const float inv = 1.f / (float)HW;
  if ((HW & 3) == 0) {
    const int hw4 = HW >> 2;
    const int64_t total = planes * hw4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
      const float v = df[i / hw4] * inv, u = mode == 0 ? v : -v;
      reinterpret_cast<float4*>(da)[i] = make_float4(v, v, v, v);
      reinterpret_cast<float4*>(db)[i] = make_float4(u, u, u, u);
    }
  } else {
    const int64_t total = planes * HW;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
      const float v = df[i / HW] * inv;
      da[i] = v;
      db[i] = mode == 0 ? v : -v;
    }
  }
}

}  // namespace

extern "C" size_t dlio_pair_fuse_fc_ws_bytes(int N, int C, int F) {
  if (N <= 0 || C <= 0 || F <= 0) return 0;
  const size_t blocks = (size_t)(C + PF_CB - 1) / PF_CB;
  return ((size_t)N * blocks * F + 16) * sizeof(float);
}

extern "C" int dlio_pair_fuse_fc_fwd(const float* a, const float* b, int N, int C, int HW, int mode, const float* w,
                                     const float* bias, int F, int act, float* feat, float* y, void* ws, size_t ws_bytes,
                                     int* counters, dlio_stream_t stream) {
  if (!a || !b || !w || !feat || !y || !ws || !counters || N <= 0 || C <= 0 || HW <= 0 || F <= 0 || mode < 0 || mode > 1 ||
      act < 0 || act > 4)
    return DLIO_EINVAL;
  if (ws_bytes < dlio_pair_fuse_fc_ws_bytes(N, C, F)) return DLIO_EWS;
  if ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) & 15) return DLIO_EUNSUP;
  const int blocks = (C + PF_CB - 1) / PF_CB;
  if ((int64_t)N * blocks > 0x7fffffff) return DLIO_EUNSUP;
  hipLaunchKernelGGL(pair_fuse_fc_fwd_kernel, dim3((unsigned)(N * blocks)), dim3(PF_T), 0, as_stream(stream), a, b, C, HW, mode,
                     w, bias, F, act, feat, y, reinterpret_cast<float*>(ws), counters);
  return dlio_check_launch();
}

extern "C" int dlio_pair_fuse_bwd(const float* df, float* da, float* db, int N, int C, int HW, int mode,
                                  dlio_stream_t stream) {
  if (!df || !da || !db || N <= 0 || C <= 0 || HW <= 0 || mode < 0 || mode > 1) return DLIO_EINVAL;
  if (((reinterpret_cast<uintptr_t>(da) | reinterpret_cast<uintptr_t>(db)) & 15) && (HW & 3) == 0) return DLIO_EUNSUP;
  const int64_t planes = (int64_t)N * C, total = planes * ((HW & 3) == 0 ? HW / 4 : HW);
  int64_t grid = (total + 255) / 256;
  if (grid > 8192) grid = 8192;
  hipLaunchKernelGGL(pair_fuse_bwd_kernel, dim3((unsigned)grid), dim3(256), 0, as_stream(stream), df, da, db, planes, HW, mode);
  return dlio_check_launch();
}
