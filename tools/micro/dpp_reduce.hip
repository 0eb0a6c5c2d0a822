// Half-wave (32-lane) sums with DPP row operations: the reduction a conv epilogue needs to fold
// BatchNorm statistics (per-channel sums over the 32 pixels a lane group holds) without ds_bpermute.
// v += row_shr:1, :2, :4, :8 (inside rows of 16 lanes, zero fill) then row_bcast:15 into rows 1 and 3:
// lane 31 holds the sum of lanes 0..31, lane 63 the sum of lanes 32..63.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { if ((x) != hipSuccess) { printf("hip error line %d\n", __LINE__); exit(1); } } while (0)

__device__ __forceinline__ float dpp_add(float v, int ctrl_dummy);   // (documentation only)

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, true));
}

__device__ __forceinline__ float half_wave_sum(float v) {
  v += dpp_mov<0x111, 0xf>(v);      // row_shr:1
  v += dpp_mov<0x112, 0xf>(v);      // row_shr:2
  v += dpp_mov<0x114, 0xf>(v);      // row_shr:4
  v += dpp_mov<0x118, 0xf>(v);      // row_shr:8  -> lane 15 of every row holds the row sum
  v += dpp_mov<0x142, 0xa>(v);      // row_bcast:15 into rows 1 and 3 -> lanes 31 / 63 hold the half sums
  return v;
}

__global__ void k(const float* in, float* out) {
  const float v = in[threadIdx.x];
  const float s = half_wave_sum(v);
  out[threadIdx.x] = s;
}

int main() {
  float h[64], o[64], *di, *dout_;
  double ref[2] = {0, 0};
  srand(3);
  for (int i = 0; i < 64; ++i) { h[i] = (float)(rand() % 1000) / 37.f - 10.f; ref[i >> 5] += h[i]; }
  CK(hipMalloc(&di, 256)); CK(hipMalloc(&dout_, 256));
  CK(hipMemcpy(di, h, 256, hipMemcpyHostToDevice));
  k<<<1, 64>>>(di, dout_);
  CK(hipMemcpy(o, dout_, 256, hipMemcpyDeviceToHost));
  printf("lane 31: %.5f (ref %.5f)   lane 63: %.5f (ref %.5f)\n", o[31], ref[0], o[63], ref[1]);
  printf("lane 15: %.5f  lane 47: %.5f (row sums of rows 0 / 2)\n", o[15], o[47]);
  return 0;
}
