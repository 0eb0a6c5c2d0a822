// SE(3) pose chain (f2f increments -> frame-to-global poses) and the pose losses, forward
// and hand-derived backward, one thread per batch element: no host round trips, no
// per-(b,s) launches.
//
// Replaces Trainer.se3_to_SE3 (trainer.py:324-351; tester.py:223-251 for xyzw order) with its
// liegroups.torch.SO3 calls (exp: Rodrigues, first-order below 1e-6 rad; to_quaternion:
// qw = sqrt(1+tr)/2 with the three largest-diagonal fallbacks when |qw| < 1e-6), and
// HWSLoss / LWSLoss (losses/losses.py:21-39, 68-86).
#include "common.h"

namespace {

struct M3 { float m[9]; };

__device__ __forceinline__ M3 mat_mul(const M3& a, const M3& b) {
  M3 r;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
      r.m[i * 3 + j] = a.m[i * 3] * b.m[j] + a.m[i * 3 + 1] * b.m[3 + j] + a.m[i * 3 + 2] * b.m[6 + j];
  return r;
}
__device__ __forceinline__ float det3(const M3& a) {
  return a.m[0] * (a.m[4] * a.m[8] - a.m[5] * a.m[7]) - a.m[1] * (a.m[3] * a.m[8] - a.m[5] * a.m[6]) +
         a.m[2] * (a.m[3] * a.m[7] - a.m[4] * a.m[6]);
}
__device__ __forceinline__ bool close1(float v) {  // torch.isclose(v, 1) defaults
  return fabsf(v - 1.f) <= 1e-8f + 1e-5f;
}

__device__ __forceinline__ M3 so3_exp(float wx, float wy, float wz) {
  M3 R;
  const float ang = sqrtf(wx * wx + wy * wy + wz * wz);
  if (ang < 1e-6f) {
    R.m[0] = 1.f; R.m[1] = -wz; R.m[2] = wy;
    R.m[3] = wz;  R.m[4] = 1.f; R.m[5] = -wx;
    R.m[6] = -wy; R.m[7] = wx;  R.m[8] = 1.f;
    return R;
  }
  const float ax = wx / ang, ay = wy / ang, az = wz / ang;
  const float s = sinf(ang), c = cosf(ang), k = 1.f - c;
  R.m[0] = c + k * ax * ax;      R.m[1] = k * ax * ay - s * az; R.m[2] = k * ax * az + s * ay;
  R.m[3] = k * ay * ax + s * az; R.m[4] = c + k * ay * ay;      R.m[5] = k * ay * az - s * ax;
  R.m[6] = k * az * ax - s * ay; R.m[7] = k * az * ay + s * ax; R.m[8] = c + k * az * az;
  return R;
}

// G = dL/dR  ->  dL/dw
__device__ __forceinline__ void so3_exp_bwd(float wx, float wy, float wz, const M3& G, float* dw) {
  const float v0 = G.m[7] - G.m[5], v1 = G.m[2] - G.m[6], v2 = G.m[3] - G.m[1];
  const float ang = sqrtf(wx * wx + wy * wy + wz * wz);
  if (ang < 1e-6f) { dw[0] = v0; dw[1] = v1; dw[2] = v2; return; }
  const float a[3] = {wx / ang, wy / ang, wz / ang};
  const float s = sinf(ang), c = cosf(ang), k = 1.f - c;
  const float tr = G.m[0] + G.m[4] + G.m[8];
  float Ga[3], Gta[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    Ga[i] = G.m[i * 3] * a[0] + G.m[i * 3 + 1] * a[1] + G.m[i * 3 + 2] * a[2];
    Gta[i] = G.m[i] * a[0] + G.m[3 + i] * a[1] + G.m[6 + i] * a[2];
  }
  const float aGa = a[0] * Ga[0] + a[1] * Ga[1] + a[2] * Ga[2];
  const float av = a[0] * v0 + a[1] * v1 + a[2] * v2;
  const float dth = -s * tr + s * aGa + c * av;
  float da[3] = {k * (Ga[0] + Gta[0]) + s * v0, k * (Ga[1] + Gta[1]) + s * v1,
                 k * (Ga[2] + Gta[2]) + s * v2};
  const float ada = a[0] * da[0] + a[1] * da[1] + a[2] * da[2];
#pragma unroll
  for (int i = 0; i < 3; ++i) dw[i] = dth * a[i] + (da[i] - a[i] * ada) / ang;
}

// quaternion as (w, x, y, z); returns branch id
__device__ __forceinline__ int rot_to_quat(const M3& R, float* q) {
  float qw = 0.5f * sqrtf(1.f + R.m[0] + R.m[4] + R.m[8]);
  if (!(fabsf(qw) < 1e-6f)) {
    const float d = 4.f * qw;
    q[0] = qw; q[1] = (R.m[7] - R.m[5]) / d; q[2] = (R.m[2] - R.m[6]) / d; q[3] = (R.m[3] - R.m[1]) / d;
    return 0;
  }
  if (R.m[0] > R.m[4] && R.m[0] > R.m[8]) {
    const float d = 2.f * sqrtf(1.f + R.m[0] - R.m[4] - R.m[8]);
    q[0] = (R.m[7] - R.m[5]) / d; q[1] = 0.25f * d; q[2] = (R.m[3] + R.m[1]) / d; q[3] = (R.m[2] + R.m[6]) / d;
    return 1;
  }
  if (R.m[4] > R.m[8]) {
    const float d = 2.f * sqrtf(1.f + R.m[4] - R.m[0] - R.m[8]);
    q[0] = (R.m[2] - R.m[6]) / d; q[1] = (R.m[3] + R.m[1]) / d; q[2] = 0.25f * d; q[3] = (R.m[7] + R.m[5]) / d;
    return 2;
  }
  const float d = 2.f * sqrtf(1.f + R.m[8] - R.m[0] - R.m[4]);
  q[0] = (R.m[3] - R.m[1]) / d; q[1] = (R.m[2] + R.m[6]) / d; q[2] = (R.m[7] + R.m[5]) / d; q[3] = 0.25f * d;
  return 3;
}

// g = dL/dq (w,x,y,z)  ->  accumulate dL/dR into G
__device__ __forceinline__ void rot_to_quat_bwd(const M3& R, const float* g, M3& G) {
  float q[4];
  const int br = rot_to_quat(R, q);
  if (br == 0) {
    const float qw = q[0];
    const float Gw = g[0] - (g[1] * q[1] + g[2] * q[2] + g[3] * q[3]) / qw;
    const float dd = Gw / (8.f * qw), e = 1.f / (4.f * qw);
    G.m[0] += dd; G.m[4] += dd; G.m[8] += dd;
    G.m[7] += g[1] * e; G.m[5] -= g[1] * e;
    G.m[2] += g[2] * e; G.m[6] -= g[2] * e;
    G.m[3] += g[3] * e; G.m[1] -= g[3] * e;
    return;
  }
  // fallback branches: component `br` equals d/4, d = 2*sqrt(u)
  const int sgn[4][3] = {{0, 0, 0}, {1, -1, -1}, {-1, 1, -1}, {-1, -1, 1}};
  const float u = 1.f + sgn[br][0] * R.m[0] + sgn[br][1] * R.m[4] + sgn[br][2] * R.m[8];
  const float d = 2.f * sqrtf(u);
  float rest = 0.f;
  for (int c = 0; c < 4; ++c) if (c != br) rest += g[c] * q[c];
  const float gd = g[br] * 0.25f - rest / d;
  const float du = gd * 2.f / d;
  G.m[0] += sgn[br][0] * du; G.m[4] += sgn[br][1] * du; G.m[8] += sgn[br][2] * du;
  const float e = 1.f / d;
  if (br == 1) {
    G.m[7] += g[0] * e; G.m[5] -= g[0] * e;
    G.m[3] += g[2] * e; G.m[1] += g[2] * e;
    G.m[2] += g[3] * e; G.m[6] += g[3] * e;
  } else if (br == 2) {
    G.m[2] += g[0] * e; G.m[6] -= g[0] * e;
    G.m[3] += g[1] * e; G.m[1] += g[1] * e;
    G.m[7] += g[3] * e; G.m[5] += g[3] * e;
  } else {
    G.m[3] += g[0] * e; G.m[1] -= g[0] * e;
    G.m[2] += g[1] * e; G.m[6] += g[1] * e;
    G.m[7] += g[2] * e; G.m[5] += g[2] * e;
  }
}

__global__ void se3_chain_fwd_kernel(const float* __restrict__ t, const float* __restrict__ w,
                                     float* __restrict__ p, float* __restrict__ q,
                                     float* __restrict__ R_all, int32_t* status, int B, int S,
                                     int order) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  M3 P;
#pragma unroll
  for (int i = 0; i < 9; ++i) P.m[i] = (i % 4 == 0) ? 1.f : 0.f;
  float pp[3] = {0.f, 0.f, 0.f};
  int bad = 0;
  for (int s = 0; s < S; ++s) {
    const float* ts = t + ((size_t)b * S + s) * 3;
    const float* wv = w + ((size_t)b * S + s) * 3;
    const M3 R = so3_exp(wv[0], wv[1], wv[2]);
    if (!close1(det3(R))) bad |= 1;
    float np[3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
      np[i] = P.m[i * 3] * ts[0] + P.m[i * 3 + 1] * ts[1] + P.m[i * 3 + 2] * ts[2] + pp[i];
    pp[0] = np[0]; pp[1] = np[1]; pp[2] = np[2];
    P = mat_mul(P, R);
    if (!close1(det3(P))) bad |= 1;
    // liegroups from_matrix(normalize=True) validity test (tol 1e-6): flagged, not repaired
    {
      float worst = fabsf(det3(P) - 1.f);
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          const float d = P.m[i] * P.m[j] + P.m[3 + i] * P.m[3 + j] + P.m[6 + i] * P.m[6 + j] -
                          (i == j ? 1.f : 0.f);
          worst = fmaxf(worst, fabsf(d));
        }
      if (!(worst < 1e-6f)) bad |= 2;
    }
    float qq[4];
    rot_to_quat(P, qq);
    float* qo = q + ((size_t)b * S + s) * 4;
    if (order == 0) { qo[0] = qq[0]; qo[1] = qq[1]; qo[2] = qq[2]; qo[3] = qq[3]; }
    else { qo[0] = qq[1]; qo[1] = qq[2]; qo[2] = qq[3]; qo[3] = qq[0]; }
    float* po = p + ((size_t)b * S + s) * 3;
    po[0] = pp[0]; po[1] = pp[1]; po[2] = pp[2];
    float* Ro = R_all + ((size_t)b * S + s) * 9;
#pragma unroll
    for (int i = 0; i < 9; ++i) Ro[i] = P.m[i];
  }
  if (bad && status) atomicOr(status, bad);
}

__global__ void se3_chain_bwd_kernel(const float* __restrict__ t, const float* __restrict__ w,
                                     const float* __restrict__ R_all, const float* __restrict__ dp,
                                     const float* __restrict__ dq, float* __restrict__ dt,
                                     float* __restrict__ dw, int B, int S, int order) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  M3 dP;
#pragma unroll
  for (int i = 0; i < 9; ++i) dP.m[i] = 0.f;
  float dpa[3] = {0.f, 0.f, 0.f};
  for (int s = S - 1; s >= 0; --s) {
    const size_t o = (size_t)b * S + s;
    M3 Ps, Pm;
#pragma unroll
    for (int i = 0; i < 9; ++i) Ps.m[i] = R_all[o * 9 + i];
    if (s > 0) {
#pragma unroll
      for (int i = 0; i < 9; ++i) Pm.m[i] = R_all[(o - 1) * 9 + i];
    } else {
#pragma unroll
      for (int i = 0; i < 9; ++i) Pm.m[i] = (i % 4 == 0) ? 1.f : 0.f;
    }
    // quaternion gradient into dP
    float g[4] = {0.f, 0.f, 0.f, 0.f};
    if (dq) {
      const float* gq = dq + o * 4;
      if (order == 0) { g[0] = gq[0]; g[1] = gq[1]; g[2] = gq[2]; g[3] = gq[3]; }
      else { g[1] = gq[0]; g[2] = gq[1]; g[3] = gq[2]; g[0] = gq[3]; }
    }
    rot_to_quat_bwd(Ps, g, dP);
    float gp[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) gp[i] = dpa[i] + (dp ? dp[o * 3 + i] : 0.f);
    const float* ts = t + o * 3;
    const float* wv = w + o * 3;
    // p_s = Pm t_s + p_{s-1}
    float* dto = dt + o * 3;
#pragma unroll
    for (int i = 0; i < 3; ++i) dto[i] = Pm.m[i] * gp[0] + Pm.m[3 + i] * gp[1] + Pm.m[6 + i] * gp[2];
    const M3 R = so3_exp(wv[0], wv[1], wv[2]);
    // P_s = Pm R : dPm = dP R^T + gp t^T ; dR = Pm^T dP
    M3 dPm, dR;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        dPm.m[i * 3 + j] = dP.m[i * 3] * R.m[j * 3] + dP.m[i * 3 + 1] * R.m[j * 3 + 1] +
                           dP.m[i * 3 + 2] * R.m[j * 3 + 2] + gp[i] * ts[j];
        dR.m[i * 3 + j] = Pm.m[i] * dP.m[j] + Pm.m[3 + i] * dP.m[3 + j] + Pm.m[6 + i] * dP.m[6 + j];
      }
    so3_exp_bwd(wv[0], wv[1], wv[2], dR, dw + o * 3);
    dP = dPm;
    dpa[0] = gp[0]; dpa[1] = gp[1]; dpa[2] = gp[2];
  }
}

// ---- pose loss: single block ------------------------------------------------------
struct LossArgs {
  const float* pred[4];
  const float* gt[4];
  float* dpred[4];
  int32_t n[4];
};

__global__ __launch_bounds__(256) void pose_loss_fwd_kernel(LossArgs a, const float* sx,
                                                            const float* sq, float beta, int mode,
                                                            float* out) {
  __shared__ double sm[16];
  __shared__ float mse[4];
  for (int i = 0; i < 4; ++i) {
    double s = 0.0;
    for (int e = threadIdx.x; e < a.n[i]; e += 256) {
      const float d = a.pred[i][e] - a.gt[i][e];
      s += (double)d * d;
    }
    const double r = block_sum_d(s, sm);
    if (threadIdx.x == 0) mse[i] = a.n[i] > 0 ? (float)(r / a.n[i]) : 0.f;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float Lt = mse[0], Lw = mse[1], Lp = mse[2], Lq = mse[3];
    float loss;
    if (mode == 0) loss = (Lp + Lt) * expf(-sx[0]) + sx[0] + (Lq + Lw) * expf(-sq[0]) + sq[0];
    else loss = (Lp + Lt) + beta * (Lq + Lw);
    out[0] = loss; out[1] = Lt; out[2] = Lw; out[3] = Lp; out[4] = Lq;
  }
}

__global__ __launch_bounds__(256) void pose_loss_bwd_kernel(LossArgs a, const float* sx,
                                                            const float* sq, float beta, int mode,
                                                            const float* out, const float* gscale,
                                                            float* dsx, float* dsq) {
  const float gs = gscale ? gscale[0] : 1.f;
  float cx, cq;
  if (mode == 0) { cx = expf(-sx[0]); cq = expf(-sq[0]); }
  else { cx = 1.f; cq = beta; }
  for (int i = 0; i < 4; ++i) {
    if (a.n[i] <= 0 || !a.dpred[i]) continue;
    const float coef = gs * ((i == 0 || i == 2) ? cx : cq) * 2.f / (float)a.n[i];
    for (int e = threadIdx.x; e < a.n[i]; e += 256) a.dpred[i][e] = coef * (a.pred[i][e] - a.gt[i][e]);
  }
  if (threadIdx.x == 0 && mode == 0) {
    const float Lt = out[1], Lw = out[2], Lp = out[3], Lq = out[4];
    if (dsx) dsx[0] = gs * (1.f - (Lp + Lt) * cx);
    if (dsq) dsq[0] = gs * (1.f - (Lq + Lw) * cq);
  }
}

}  // namespace

extern "C" int dlio_se3_chain_fwd(const float* t, const float* w, float* p, float* q, float* R_all,
                                  int32_t* status, int B, int S, int order, dlio_stream_t stream) {
  if (!t || !w || !p || !q || !R_all || B <= 0 || S <= 0 || (order != 0 && order != 1))
    return DLIO_EINVAL;
  hipLaunchKernelGGL(se3_chain_fwd_kernel, dim3(cdiv(B, 64)), dim3(64), 0, as_stream(stream), t, w,
                     p, q, R_all, status, B, S, order);
  return dlio_check_launch();
}

extern "C" int dlio_se3_chain_bwd(const float* t, const float* w, const float* R_all,
                                  const float* dp, const float* dq, float* dt, float* dw, int B,
                                  int S, int order, dlio_stream_t stream) {
  if (!t || !w || !R_all || !dt || !dw || B <= 0 || S <= 0) return DLIO_EINVAL;
  hipLaunchKernelGGL(se3_chain_bwd_kernel, dim3(cdiv(B, 64)), dim3(64), 0, as_stream(stream), t, w,
                     R_all, dp, dq, dt, dw, B, S, order);
  return dlio_check_launch();
}

static int fill_loss_args(LossArgs& a, const float* const* pred, const float* const* gt,
                          const int32_t* n, float* const* dpred) {
  for (int i = 0; i < 4; ++i) {
    a.n[i] = n[i];
    a.pred[i] = pred[i];
    a.gt[i] = gt[i];
    a.dpred[i] = dpred ? dpred[i] : nullptr;
    if (n[i] < 0 || (n[i] > 0 && (!pred[i] || !gt[i]))) return DLIO_EINVAL;
  }
  return DLIO_OK;
}

extern "C" int dlio_pose_loss_fwd(const float* const* pred, const float* const* gt,
                                  const int32_t* n, const float* sx, const float* sq, float beta,
                                  int mode, float* out, dlio_stream_t stream) {
  if (!pred || !gt || !n || !out || (mode == 0 && (!sx || !sq)) || (mode != 0 && mode != 1))
    return DLIO_EINVAL;
  LossArgs a;
  int rc = fill_loss_args(a, pred, gt, n, nullptr);
  if (rc) return rc;
  hipLaunchKernelGGL(pose_loss_fwd_kernel, dim3(1), dim3(256), 0, as_stream(stream), a, sx, sq,
                     beta, mode, out);
  return dlio_check_launch();
}

extern "C" int dlio_pose_loss_bwd(const float* const* pred, const float* const* gt,
                                  const int32_t* n, const float* sx, const float* sq, float beta,
                                  int mode, const float* out, const float* gscale,
                                  float* const* dpred, float* dsx, float* dsq,
                                  dlio_stream_t stream) {
  if (!pred || !gt || !n || !out || !dpred || (mode == 0 && (!sx || !sq))) return DLIO_EINVAL;
  LossArgs a;
  int rc = fill_loss_args(a, pred, gt, n, dpred);
  if (rc) return rc;
  hipLaunchKernelGGL(pose_loss_bwd_kernel, dim3(1), dim3(256), 0, as_stream(stream), a, sx, sq,
                     beta, mode, out, gscale, dsx, dsq);
  return dlio_check_launch();
}
