// Feasibility probe for round 2: fp32 convolutions on the bf16 matrix cores by operand splitting.
// x = hi + mid + lo with three bf16 pieces (8+8+8 mantissa bits); a*b is approximated by
//   3 terms: hi*hi + hi*mid + mid*hi                     (drops ~2^-16 |a||b|)
//   6 terms: + hi*lo + mid*mid + lo*hi                   (drops ~2^-24 |a||b|)
// each term one v_mfma_f32_32x32x16_bf16 (16x the rate of v_mfma_f32_32x32x2_f32, products exact in
// fp32, fp32 accumulation).  Part 1: error of a 32x32x576 tile (the K of a 64-channel 3x3 layer)
// against fp64, beside the fp32 MFMA.  Part 2: sustained fp32-EQUIVALENT TFLOP/s of the MFMA stream
// including the per-use split of one operand (the other is pre-split, as conv weights would be).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
#define CK(x) do { if ((x) != hipSuccess) { printf("hip error line %d\n", __LINE__); exit(1); } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split3(float x, __bf16& h, __bf16& m, __bf16& l) {
  h = (__bf16)x;
  const float r = x - (float)h;
  m = (__bf16)r;
  l = (__bf16)(r - (float)m);
}

// one wave, one 32x32 tile: A [32][K] row-major, B [K][32] row-major
template <int TERMS>   // 0 = fp32 MFMA, 3 / 6 = bf16 split
__global__ void tile_kernel(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int K) {
  const int lane = threadIdx.x, l31 = lane & 31, half = lane >> 5;
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  if (TERMS == 0) {
    for (int k = 0; k < K; k += 2)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[l31 * K + k + half], B[(k + half) * 32 + l31], acc, 0, 0, 0);
  } else {
    for (int k0 = 0; k0 < K; k0 += 16) {
      bf16x8 ah, am, al, bh, bm, bl;
      for (int j = 0; j < 8; ++j) {
        const int k = k0 + 8 * half + j;
        __bf16 h, m, l;
        split3(A[l31 * K + k], h, m, l); ah[j] = h; am[j] = m; al[j] = l;
        split3(B[k * 32 + l31], h, m, l); bh[j] = h; bm[j] = m; bl[j] = l;
      }
      if (TERMS == 6) {      // smallest terms first
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
      }
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
    }
  }
  for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * half) * 32 + l31] = acc[r];
}

// rate: MR accumulators share one B fragment that is split per K-step; A fragments are pre-split
template <int TERMS, int MR>
__global__ __launch_bounds__(256) void rate_kernel(const float* __restrict__ src, float* out, int iters) {
  __shared__ float lds[2048];
  for (int i = threadIdx.x; i < 2048; i += 256) lds[i] = src[i];
  __syncthreads();
  f32x16 acc[MR];
  for (int i = 0; i < MR; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bf16x8 ah[MR], am[MR], al[MR];
  for (int i = 0; i < MR; ++i) for (int j = 0; j < 8; ++j) {
    __bf16 h, m, l; split3(src[(threadIdx.x + 64 * i + j) & 2047], h, m, l); ah[i][j] = h; am[i][j] = m; al[i][j] = l;
  }
  const int lane = threadIdx.x & 63;
  for (int it = 0; it < iters; ++it) {
    bf16x8 bh, bm, bl;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      __bf16 h, m, l;
      split3(lds[(it * 64 + lane * 8 + j) & 2047], h, m, l); bh[j] = h; bm[j] = m; bl[j] = l;
    }
#pragma unroll
    for (int i = 0; i < MR; ++i) {
      if (TERMS == 6) {
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh, acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[i], bm, acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl, acc[i], 0, 0, 0);
      }
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[i], bh, acc[i], 0, 0, 0);
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bm, acc[i], 0, 0, 0);
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh, acc[i], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int i = 0; i < MR; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int TERMS>
void accuracy(const std::vector<float>& A, const std::vector<float>& B, const std::vector<double>& ref, double scale,
              int K, const char* tag) {
  float *dA, *dB, *dC;
  CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, B.size() * 4)); CK(hipMalloc(&dC, 1024 * 4));
  CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
  tile_kernel<TERMS><<<1, 64>>>(dA, dB, dC, K);
  std::vector<float> C(1024);
  CK(hipMemcpy(C.data(), dC, 1024 * 4, hipMemcpyDeviceToHost));
  double emax = 0, esum = 0;
  for (int i = 0; i < 1024; ++i) { const double e = fabs(C[i] - ref[i]); emax = fmax(emax, e); esum += e * e; }
  printf("%-28s max |err| / max|C| = %.2e   rms = %.2e\n", tag, emax / scale, sqrt(esum / 1024) / scale);
  CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dC));
}

template <int TERMS, int MR>
void rate(int blocks, const float* src) {
  float* out; CK(hipMalloc(&out, (size_t)blocks * 256 * 4));
  const int iters = 4000;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  rate_kernel<TERMS, MR><<<blocks, 256>>>(src, out, 50); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0)); rate_kernel<TERMS, MR><<<blocks, 256>>>(src, out, iters); CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double fl = (double)blocks * 4 * iters * MR * 2.0 * 32 * 32 * 16;     // fp32-equivalent FLOPs
  printf("%d-term split, %d tiles per B fragment, %d waves/SIMD: %.1f fp32-equivalent TFLOP/s (fp32 MFMA peak 157.3)\n",
         TERMS, MR, blocks / 256, fl / ms / 1e9);
  CK(hipFree(out));
}

int main() {
  const int K = 576;
  std::vector<float> A(32 * K), B(K * 32);
  srand(7);
  auto rnd = [] { return (float)((rand() / (double)RAND_MAX) * 2.0 - 1.0); };
  for (auto& v : A) v = rnd() * 0.1f;           // weights ~ N(0, small)
  for (auto& v : B) v = fmaxf(rnd() + 0.3f, 0.f);   // post-ReLU activations: non-negative, many zeros
  std::vector<double> ref(1024, 0.0);
  double scale = 0;
  for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
    double s = 0; for (int k = 0; k < K; ++k) s += (double)A[i * K + k] * (double)B[k * 32 + j];
    ref[i * 32 + j] = s; scale = fmax(scale, fabs(s));
  }
  accuracy<0>(A, B, ref, scale, K, "fp32 MFMA (32x32x2)");
  accuracy<3>(A, B, ref, scale, K, "bf16 split, 3 terms");
  accuracy<6>(A, B, ref, scale, K, "bf16 split, 6 terms");
  float* src; CK(hipMalloc(&src, 2048 * 4));
  std::vector<float> h(2048); for (auto& v : h) v = rnd();
  CK(hipMemcpy(src, h.data(), 2048 * 4, hipMemcpyHostToDevice));
  rate<6, 1>(512, src); rate<6, 2>(512, src); rate<6, 4>(512, src); rate<6, 4>(256, src);
  rate<3, 2>(512, src); rate<3, 4>(512, src);
  return 0;
}
