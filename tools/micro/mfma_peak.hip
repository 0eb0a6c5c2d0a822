// Sustained v_mfma_f32_32x32x2_f32 rate on this box (no memory traffic): calibrates the 157.3 TF/s
// spec peak used in bench.py against what the clocks actually sustain.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { if ((x) != hipSuccess) { printf("hip error line %d\n", __LINE__); return; } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = a0 + threadIdx.x * 1e-6f, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
void run(int blocks, const char* tag) {
  float* out; CK(hipMalloc(&out, (size_t)blocks * 256 * 4));
  int iters = 20000;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  k<NACC><<<blocks, 256>>>(out, 100, 1.f, 1.f);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  k<NACC><<<blocks, 256>>>(out, iters, 1.f, 1.f);
  CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  double fl = (double)blocks * 4 * iters * 8 * NACC * 2.0 * 32 * 32 * 2;
  printf("%s blocks=%d nacc=%d: %.2f ms  %.1f TFLOP/s\n", tag, blocks, NACC, ms, fl / ms / 1e9);
  CK(hipFree(out));
}
int main() {
  run<4>(256, "1 wave/SIMD");
  run<4>(512, "2 waves/SIMD");
  run<2>(1024, "4 waves/SIMD");
  run<1>(512, "dependent chain, 2 waves/SIMD");
  run<4>(512, "2 waves/SIMD (repeat, warm clocks)");
  return 0;
}
