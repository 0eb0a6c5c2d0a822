// Device-side batch preparation: what DataCombiCreater.process (deeplio/models/misc.py:24-125)
// does on the host with advanced-index gathers, .contiguous() copies and per-sample Python
// loops over 4x4 matrices.
//   pair_stack : images [B][F][Ctot][H][W] + combinations [S][2] -> xyz [B][S][2][c0][H][W] and
//                normals [B][S][2][Ctot-c0][H][W] in ONE float4 streaming pass (misc.py:65-69).
//   gt_relative: ground-truth rows [x(3), R(9), v(3)] -> frame-to-frame [dx, log(R)] and
//                frame-to-first [p, quat wxyz] targets (misc.py:83-125), one thread per (b, s),
//                fp32 arithmetic in the reference's operation order (inv_SE3: R^T, -(R^T t);
//                4x4 product; liegroups SO3.log / to_quaternion).
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void pair_stack_kernel(const float* __restrict__ img,
                                                         const int32_t* __restrict__ comb,
                                                         float* __restrict__ xyz,
                                                         float* __restrict__ nrm, int B, int F,
                                                         int Ctot, int c0, int HW4, int S) {
  // one float4 per thread; output index space: [B][S][2][Ctot][HW4] (channel routed to xyz / nrm)
  const int64_t total = (int64_t)B * S * 2 * Ctot * HW4;
  const int c1 = Ctot - c0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int p = (int)(i % HW4);
    int64_t t = i / HW4;
    const int c = (int)(t % Ctot); t /= Ctot;
    const int k = (int)(t % 2); t /= 2;
    const int s = (int)(t % S);
    const int b = (int)(t / S);
    const int f = comb[s * 2 + k];
    const float4 v = reinterpret_cast<const float4*>(img)[(((int64_t)b * F + f) * Ctot + c) * HW4 + p];
    if (c < c0)
      reinterpret_cast<float4*>(xyz)[((((int64_t)b * S + s) * 2 + k) * c0 + c) * HW4 + p] = v;
    else
      reinterpret_cast<float4*>(nrm)[((((int64_t)b * S + s) * 2 + k) * c1 + (c - c0)) * HW4 + p] = v;
  }
}

struct T34 { float R[9]; float t[3]; };

__device__ __forceinline__ T34 load_pose(const float* g) {   // [x(3), R(9) row-major, v(3)]
  T34 T;
  T.t[0] = g[0]; T.t[1] = g[1]; T.t[2] = g[2];
#pragma unroll
  for (int i = 0; i < 9; ++i) T.R[i] = g[3 + i];
  return T;
}
// inv_SE3 (common/spatial.py:904-923): R^T and -(R^T t)
__device__ __forceinline__ T34 inv_pose(const T34& T) {
  T34 I;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) I.R[i * 3 + j] = T.R[j * 3 + i];
#pragma unroll
  for (int i = 0; i < 3; ++i)
    I.t[i] = -(I.R[i * 3] * T.t[0] + I.R[i * 3 + 1] * T.t[1] + I.R[i * 3 + 2] * T.t[2]);
  return I;
}
// A @ B for homogeneous [R t; 0 1]
__device__ __forceinline__ T34 mul_pose(const T34& A, const T34& B) {
  T34 C;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
#pragma unroll
    for (int j = 0; j < 3; ++j)
      C.R[i * 3 + j] = A.R[i * 3] * B.R[j] + A.R[i * 3 + 1] * B.R[3 + j] + A.R[i * 3 + 2] * B.R[6 + j];
    C.t[i] = A.R[i * 3] * B.t[0] + A.R[i * 3 + 1] * B.t[1] + A.R[i * 3 + 2] * B.t[2] + A.t[i];
  }
  return C;
}

__global__ void gt_relative_kernel(const float* __restrict__ gts, const int32_t* __restrict__ comb,
                                   float* __restrict__ f2f, float* __restrict__ f2g,
                                   int32_t* __restrict__ flag, int B, int F, int S) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * S) return;
  const int b = i / S, s = i - b * S;
  const float* gb = gts + (size_t)b * F * 15;
  const T34 Ti = load_pose(gb + (size_t)comb[s * 2] * 15);
  const T34 Tn = load_pose(gb + (size_t)comb[s * 2 + 1] * 15);
  const T34 T0 = load_pose(gb);
  // frame to frame: [dx, SO3.log(R)] (misc.py:98-110)
  const T34 rel = mul_pose(inv_pose(Ti), Tn);
  float* o = f2f + (size_t)i * 6;
  o[0] = rel.t[0]; o[1] = rel.t[1]; o[2] = rel.t[2];
  {
    float ca = 0.5f * (rel.R[0] + rel.R[4] + rel.R[8]) - 0.5f;
    ca = fminf(fmaxf(ca, -1.f), 1.f);
    const float ang = acosf(ca);
    float m21, m02, m10;
    if (fabsf(ang) < 1e-6f) {
      m21 = rel.R[7]; m02 = rel.R[2]; m10 = rel.R[3];          // vee(R - I)
    } else {
      const float k = 0.5f * ang / sinf(ang);
      m21 = k * (rel.R[7] - rel.R[5]); m02 = k * (rel.R[2] - rel.R[6]); m10 = k * (rel.R[3] - rel.R[1]);
    }
    o[3] = m21; o[4] = m02; o[5] = m10;
    bool bad = false;
#pragma unroll
    for (int j = 0; j < 6; ++j) bad |= !(fabsf(o[j]) <= 3.402823466e38f);
    if (bad && flag) atomicOr(flag, 1);                         // misc.py:106-107 raises ValueError
  }
  // frame to first frame: [p, quaternion wxyz] (misc.py:112-121)
  const T34 glo = mul_pose(inv_pose(T0), Tn);
  float* g = f2g + (size_t)i * 7;
  g[0] = glo.t[0]; g[1] = glo.t[1]; g[2] = glo.t[2];
  const float* R = glo.R;
  float qw = 0.5f * sqrtf(1.f + R[0] + R[4] + R[8]), qx, qy, qz;
  if (!(fabsf(qw) < 1e-6f)) {
    const float d = 4.f * qw;
    qx = (R[7] - R[5]) / d; qy = (R[2] - R[6]) / d; qz = (R[3] - R[1]) / d;
  } else if (R[0] > R[4] && R[0] > R[8]) {
    const float d = 2.f * sqrtf(1.f + R[0] - R[4] - R[8]);
    qw = (R[7] - R[5]) / d; qx = 0.25f * d; qy = (R[3] + R[1]) / d; qz = (R[2] + R[6]) / d;
  } else if (R[4] > R[8]) {
    const float d = 2.f * sqrtf(1.f + R[4] - R[0] - R[8]);
    qw = (R[2] - R[6]) / d; qx = (R[3] + R[1]) / d; qy = 0.25f * d; qz = (R[7] + R[5]) / d;
  } else {
    const float d = 2.f * sqrtf(1.f + R[8] - R[0] - R[4]);
    qw = (R[3] - R[1]) / d; qx = (R[2] + R[6]) / d; qy = (R[7] + R[5]) / d; qz = 0.25f * d;
  }
  g[3] = qw; g[4] = qx; g[5] = qy; g[6] = qz;
}

}  // namespace

extern "C" int dlio_pair_stack(const float* images, const int32_t* combinations, float* xyz,
                               float* normals, int B, int F, int Ctot, int c_split, int H, int W,
                               int S, dlio_stream_t stream) {
  if (!images || !combinations || !xyz || !normals || B <= 0 || F <= 0 || S <= 0 || Ctot <= 0 ||
      c_split <= 0 || c_split >= Ctot)
    return DLIO_EINVAL;
  if (((int64_t)H * W) % 4 != 0) return DLIO_EUNSUP;
  const int HW4 = H * W / 4;
  const int64_t total = (int64_t)B * S * 2 * Ctot * HW4;
  hipLaunchKernelGGL(pair_stack_kernel, dim3(ew_grid(total, 256)), dim3(256), 0, as_stream(stream),
                     images, combinations, xyz, normals, B, F, Ctot, c_split, HW4, S);
  return dlio_check_launch();
}

extern "C" int dlio_gt_relative(const float* gts, const int32_t* combinations, float* f2f, float* f2g,
                                int32_t* flag, int B, int F, int S, dlio_stream_t stream) {
  if (!gts || !combinations || !f2f || !f2g || B <= 0 || F <= 0 || S <= 0) return DLIO_EINVAL;
  hipLaunchKernelGGL(gt_relative_kernel, dim3(cdiv(B * S, 64)), dim3(64), 0, as_stream(stream), gts,
                     combinations, f2f, f2g, flag, B, F, S);
  return dlio_check_launch();
}
