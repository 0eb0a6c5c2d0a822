// Shared helpers for the gfx950 kernels of libdeeplio_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/deeplio_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define DLIO_WAVE 64

// Every launcher converts its stream argument first; use that point to drop any stale
// (non-sticky) error left in this thread by an earlier runtime call of the host framework,
// so that dlio_check_launch() reports only OUR launch.
static inline hipStream_t as_stream(dlio_stream_t s) {
  (void)hipGetLastError();
  return reinterpret_cast<hipStream_t>(s);
}

extern thread_local int dlio_last_hip_error;   // runtime.hip
static inline int dlio_check_launch() {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) dlio_last_hip_error = (int)e;
  return e == hipSuccess ? DLIO_OK : DLIO_ELAUNCH;
}

static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// grid size for grid-stride memory-bound kernels: 256 CUs x 8 blocks
static inline int ew_grid(int64_t work_items, int block) {
  int64_t g = cdiv64(work_items, block);
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  return (int)g;
}

// XCD-aware workgroup order.  The dispatcher deals workgroups round-robin over the 8 XCDs (workgroup b runs on XCD
// b % 8), each with its own L2: neighbours in blockIdx -- which the kernels make neighbours in the tensor (the
// output-channel tiles of one pixel tile, adjacent halo tiles) so that they share input -- land on eight different
// L2s and every one of them fetches the shared input from HBM again.  With v = (b % 8) * (n / 8) + b / 8 an XCD
// works through a contiguous range of virtual indices, in order.
// Timing experiment only (tools/variant_lib.py build ... -DDLIO_SPLIT_Q0=3): start the six-product sequence of the
// split-bf16 kernels at product 3, i.e. run (mid,hi) (hi,mid) (hi,hi) = a TWO-piece split with three MFMAs per product and no
// fragment reads of the third plane -- the instruction mix a two-piece fp16 split would have (open items).
// The product build uses 0.
#ifndef DLIO_SPLIT_Q0
#define DLIO_SPLIT_Q0 0
#endif
#ifndef DLIO_XCD_SWIZZLE
#define DLIO_XCD_SWIZZLE 1
#endif
__device__ __forceinline__ int xcd_block_index() {
  const int b = blockIdx.x;
  if constexpr (!DLIO_XCD_SWIZZLE) return b;
  const int per = gridDim.x >> 3;
  return b < (per << 3) ? (b & 7) * per + (b >> 3) : b;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// block-wide sum of a double (blockDim.x multiple of 64, <= 1024); result valid in thread 0
__device__ __forceinline__ double block_sum_d(double v, double* smem /* >= 16 doubles */) {
  v = wave_sum_d(v);
  int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) smem[w] = v;
  __syncthreads();
  double r = 0.0;
  if (threadIdx.x == 0) {
    int nw = (blockDim.x + 63) >> 6;
    for (int i = 0; i < nw; ++i) r += smem[i];
  }
  return r;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }

// largest magnitude a workgroup produced -> *amax_out (one device float, zero before the launch: the operand scale of the
// two-piece fp16 kernels).  Every thread of the workgroup calls this (barriers inside); at most ONE global atomic per workgroup,
// and only when it would raise the value (same-address atomics serialise at ~13 ns each: thousands of workgroups).
__device__ __forceinline__ void block_amax_commit(float amax, float* amax_out, unsigned* s_wg /* one unsigned in LDS */) {
  if (threadIdx.x == 0) *s_wg = 0u;
  __syncthreads();
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
  if ((threadIdx.x & 63) == 0) atomicMax(s_wg, __float_as_uint(amax));
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned w = *s_wg;
    if (w > __hip_atomic_load(reinterpret_cast<unsigned*>(amax_out), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
      atomicMax(reinterpret_cast<unsigned*>(amax_out), w);
  }
}
// Streaming (non-temporal) 16-byte store for outputs no later launch finds in L2 anyway (activations of 100+ MB):
// DLIO_NT_SITES is a bit set of the kernels that use it (A/B by tools/variant_lib.py), 0 = plain stores everywhere.
#ifndef DLIO_NT_SITES
#define DLIO_NT_SITES 59
#endif
template <int SITE>
__device__ __forceinline__ void st4(float* p, float4 v) {
  if constexpr ((DLIO_NT_SITES & SITE) != 0) {
    f32x4 o = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(p));
  } else {
    *reinterpret_cast<float4*>(p) = v;
  }
}
__device__ __forceinline__ float amax4(float m, float a, float b, float c, float d) {
  return fmaxf(m, fmaxf(fmaxf(fabsf(a), fabsf(b)), fmaxf(fabsf(c), fabsf(d))));
}

// compute units of the current device (cached; 256 on MI355X, also the answer without a device
// so that workspace queries work on a build host)
int dlio_num_cus();

// hipFuncAttributeMaxDynamicSharedMemorySize for a kernel, once per (device, kernel) -- the attribute is per DEVICE: a
// process-wide `static bool done` left a second GPU of the same process without it (runtime.hip; thread-safe)
void dlio_set_max_lds(const void* kernel, int bytes);

// profiling hooks (runtime.hip); kinds: include/deeplio_hip.h
void dlio_prof_begin(int kind, hipStream_t s, double flops, double bytes);
void dlio_prof_end(int kind, hipStream_t s);
struct DlioProfScope {          // brackets the launches issued during its lifetime
  int kind; hipStream_t s;
  DlioProfScope(int k, hipStream_t st, double flops, double bytes) : kind(k), s(st) { dlio_prof_begin(k, st, flops, bytes); }
  ~DlioProfScope() { dlio_prof_end(kind, s); }
};
