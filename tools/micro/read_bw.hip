// What read bandwidth does a streaming reduction reach on this part, and which launch shape gets
// it?  Variants: loads in flight per thread (U), grid-stride vs one contiguous chunk per
// workgroup, nontemporal loads, workgroups per CU.  Buffer is 1 GiB (4x the Infinity Cache).
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { if ((x) != hipSuccess) { printf("hip error line %d\n", __LINE__); exit(1); } } while (0)

template <int U, int MODE>   // MODE 0 grid-stride, 1 contiguous chunk per WG, 2 grid-stride nontemporal
__global__ __launch_bounds__(256) void rd(const float4* __restrict__ x, size_t n4, float* out) {
  float s = 0.f;
  if (MODE == 1) {
    const size_t per = n4 / gridDim.x;
    const float4* p = x + (size_t)blockIdx.x * per;
    for (size_t i = threadIdx.x; i + (U - 1) * 256 < per; i += U * 256) {
      float4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = p[i + u * 256];
#pragma unroll
      for (int u = 0; u < U; ++u) s += (v[u].x + v[u].y) + (v[u].z + v[u].w);
    }
  } else {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i + (U - 1) * stride < n4; i += U * stride) {
      float4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (MODE == 2) {
          const float* q = reinterpret_cast<const float*>(x + i + u * stride);
          v[u].x = __builtin_nontemporal_load(q); v[u].y = __builtin_nontemporal_load(q + 1);
          v[u].z = __builtin_nontemporal_load(q + 2); v[u].w = __builtin_nontemporal_load(q + 3);
        } else v[u] = x[i + u * stride];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) s += (v[u].x + v[u].y) + (v[u].z + v[u].w);
    }
  }
  if (s == 12345.678f) out[0] = s;
}

// copy: read + write
template <int U>
__global__ __launch_bounds__(256) void cp(const float4* __restrict__ x, float4* __restrict__ y, size_t n4) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i + (U - 1) * stride < n4; i += U * stride) {
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = x[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; ++u) y[i + u * stride] = v[u];
  }
}

template <typename F>
double timeit(F f) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  f(); f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0)); for (int i = 0; i < 5; ++i) f(); CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / 5 * 1e-3;
}

int main() {
  const size_t bytes = (size_t)1 << 30, n4 = bytes / 16;
  float4 *x, *y; float* out;
  CK(hipMalloc(&x, bytes)); CK(hipMalloc(&y, bytes)); CK(hipMalloc(&out, 4));
  CK(hipMemset(x, 0, bytes)); CK(hipMemset(y, 0, bytes));
  const int grids[] = {256, 512, 1024, 2048, 4096, 16384, 65536};
  for (int g : grids) {
    printf("grid %6d:", g);
#define R(U, M) printf("  U%d/m%d %.2f", U, M, bytes / timeit([&] { hipLaunchKernelGGL((rd<U, M>), dim3(g), dim3(256), 0, 0, x, n4, out); }) / 1e12)
    R(1, 0); R(2, 0); R(4, 0); R(8, 0); R(4, 1); R(8, 1); R(4, 2);
    printf("  | copy U1 %.2f U4 %.2f TB/s(r+w)\n",
           2.0 * bytes / timeit([&] { hipLaunchKernelGGL((cp<1>), dim3(g), dim3(256), 0, 0, x, y, n4); }) / 1e12,
           2.0 * bytes / timeit([&] { hipLaunchKernelGGL((cp<4>), dim3(g), dim3(256), 0, 0, x, y, n4); }) / 1e12);
  }
  return 0;
}
