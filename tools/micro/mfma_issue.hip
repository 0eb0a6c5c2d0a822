// Does the instruction mix of the conv inner loop (per 2 MFMAs: 1 ds_read2_b32, 2 L1-resident
// buffer loads, a few SALU ops) cap the MFMA rate below what bare MFMAs sustain?
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { if ((x) != hipSuccess) { printf("hip error line %d\n", __LINE__); return; } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int MODE, int NMF>
__global__ __launch_bounds__(256) void k(const float* __restrict__ w, float* out, int iters) {
  __shared__ float lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = 1.f + i * 1e-6f;
  __syncthreads();
  f32x16 acc[NMF];
  for (int i = 0; i < NMF; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  const int lane = threadIdx.x & 63;
  float a[2] = {1.f, 1.f}, b[2] = {1.f, 1.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (MODE >= 1) { b[0] = lds[(it * 8 + u) * 64 % 3968 + lane]; b[1] = lds[(it * 8 + u) * 64 % 3968 + lane + 32]; }
      if (MODE >= 2) { a[0] = w[((it * 8 + u) * 128 + lane) & 8191]; a[1] = w[((it * 8 + u) * 128 + 64 + lane) & 8191]; }
#pragma unroll
      for (int i = 0; i < NMF; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i & 1], b[i & 1], acc[i], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int i = 0; i < NMF; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE, int NMF>
void run(int blocks, const char* tag) {
  float *out, *w; CK(hipMalloc(&out, (size_t)blocks * 256 * 4)); CK(hipMalloc(&w, 8192 * 4)); CK(hipMemset(w, 0, 8192 * 4));
  const int iters = 4000;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  k<MODE, NMF><<<blocks, 256>>>(w, out, 50); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0)); k<MODE, NMF><<<blocks, 256>>>(w, out, iters); CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("%-46s blocks=%d mfma/step=%d: %.1f TFLOP/s\n", tag, blocks, NMF, (double)blocks * 4 * iters * 8 * NMF * 4096.0 / ms / 1e9);
  CK(hipFree(out)); CK(hipFree(w));
}
int main() {
  run<0, 2>(1024, "bare MFMA, 4 waves/SIMD");
  run<1, 2>(1024, "+ ds_read x2 per step");
  run<2, 2>(1024, "+ ds_read x2 + 2 global loads per step");
  run<2, 4>(1024, "same mix, 4 MFMA per step");
  run<2, 2>(512, "same mix, 2 waves/SIMD");
  return 0;
}
