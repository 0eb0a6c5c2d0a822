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

// liegroups from_matrix(normalize=True) validity test (SO3.is_valid_matrix, tol 1e-6)
__device__ __forceinline__ bool rot_is_valid(const M3& P) {
  float worst = fabsf(det3(P) - 1.f);
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const float d = P.m[i] * P.m[j] + P.m[3 + i] * P.m[3 + j] + P.m[6 + i] * P.m[6 + j] -
                      (i == j ? 1.f : 0.f);
      worst = fmaxf(worst, fabsf(d));
    }
  return worst < 1e-6f;
}

// SO3.normalize (liegroups: U diag(1,1,det U det V) V^T of the SVD) without an SVD: the orthogonal
// polar factor by Newton's iteration X <- (X + X^-T)/2, quadratically convergent -- from the 1e-6..1e-3
// defects a product of fp32 rotations can have, four steps reach fp32 round-off.  For det > 0 (always
// the case for a chain of exponentials) the polar factor IS U V^T.
__device__ __forceinline__ M3 rot_project(const M3& P) {
  M3 X = P;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const float c00 = X.m[4] * X.m[8] - X.m[5] * X.m[7], c01 = X.m[5] * X.m[6] - X.m[3] * X.m[8],
                c02 = X.m[3] * X.m[7] - X.m[4] * X.m[6];
    const float c10 = X.m[2] * X.m[7] - X.m[1] * X.m[8], c11 = X.m[0] * X.m[8] - X.m[2] * X.m[6],
                c12 = X.m[1] * X.m[6] - X.m[0] * X.m[7];
    const float c20 = X.m[1] * X.m[5] - X.m[2] * X.m[4], c21 = X.m[2] * X.m[3] - X.m[0] * X.m[5],
                c22 = X.m[0] * X.m[4] - X.m[1] * X.m[3];
    const float inv = 1.f / (X.m[0] * c00 + X.m[1] * c01 + X.m[2] * c02);
    // X^-T = cofactor(X) / det
    const float cof[9] = {c00, c01, c02, c10, c11, c12, c20, c21, c22};
#pragma unroll
    for (int i = 0; i < 9; ++i) X.m[i] = 0.5f * (X.m[i] + cof[i] * inv);
  }
  return X;
}

// backward of the projection at a (nearly) orthogonal point: dQ = Q skew(Q^T dP), hence
// dL/dP = Q skew(Q^T G), skew(A) = (A - A^T)/2 (exact on SO(3); the chain's defects are <= 1e-5)
__device__ __forceinline__ void rot_project_bwd(const M3& Q, const M3& G, M3& dP) {
  float A[9];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
      A[i * 3 + j] = Q.m[i] * G.m[j] + Q.m[3 + i] * G.m[3 + j] + Q.m[6 + i] * G.m[6 + j];
  float K[9];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) K[i * 3 + j] = 0.5f * (A[i * 3 + j] - A[j * 3 + i]);
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
      dP.m[i * 3 + j] += Q.m[i * 3] * K[j] + Q.m[i * 3 + 1] * K[3 + j] + Q.m[i * 3 + 2] * K[6 + j];
}

__global__ void so3_project_kernel(const float* __restrict__ R, float* __restrict__ Q, int32_t* valid, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  M3 P;
#pragma unroll
  for (int k = 0; k < 9; ++k) P.m[k] = R[(size_t)i * 9 + k];
  const bool ok = rot_is_valid(P);
  const M3 X = ok ? P : rot_project(P);
#pragma unroll
  for (int k = 0; k < 9; ++k) Q[(size_t)i * 9 + k] = X.m[k];
  if (valid) valid[i] = ok ? 1 : 0;
}

__global__ void so3_project_bwd_kernel(const float* __restrict__ R, const float* __restrict__ G,
                                       float* __restrict__ dR, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  M3 P, g, d;
#pragma unroll
  for (int k = 0; k < 9; ++k) { P.m[k] = R[(size_t)i * 9 + k]; g.m[k] = G[(size_t)i * 9 + k]; d.m[k] = 0.f; }
  if (rot_is_valid(P)) d = g;
  else rot_project_bwd(rot_project(P), g, d);
#pragma unroll
  for (int k = 0; k < 9; ++k) dR[(size_t)i * 9 + k] = d.m[k];
}

__global__ void se3_chain_fwd_kernel(const float* __restrict__ t, const float* __restrict__ w,
                                     float* __restrict__ p, float* __restrict__ q,
                                     float* __restrict__ R_all, int32_t* status, int B, int S,
                                     int order, int32_t* nonfinite = nullptr) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  M3 P;
#pragma unroll
  for (int i = 0; i < 9; ++i) P.m[i] = (i % 4 == 0) ? 1.f : 0.f;
  float pp[3] = {0.f, 0.f, 0.f};
  int bad = 0, nf = 0;
  for (int s = 0; s < S; ++s) {
    const float* ts = t + ((size_t)b * S + s) * 3;
    const float* wv = w + ((size_t)b * S + s) * 3;
    if (nonfinite)        // trainer.py:240-243 (torch.isnan / isinf of the model output), without a launch of its own
      for (int i = 0; i < 3; ++i) nf |= !(fabsf(ts[i]) <= 3.402823466e38f) || !(fabsf(wv[i]) <= 3.402823466e38f);
    const M3 R = so3_exp(wv[0], wv[1], wv[2]);
    if (!close1(det3(R))) bad |= 1;
    float np[3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
      np[i] = P.m[i * 3] * ts[0] + P.m[i * 3 + 1] * ts[1] + P.m[i * 3 + 2] * ts[2] + pp[i];
    pp[0] = np[0]; pp[1] = np[1]; pp[2] = np[2];
    P = mat_mul(P, R);
    if (!close1(det3(P))) bad |= 1;
    // SO3.from_matrix(P, normalize=True) (trainer.py:349): an invalid product is re-orthonormalised
    // before the quaternion is taken (status bit 1 records that it happened); the chain itself goes
    // on with the raw product, as in the reference
    float qq[4];
    if (rot_is_valid(P)) {
      rot_to_quat(P, qq);
    } else {
      bad |= 2;
      rot_to_quat(rot_project(P), qq);
    }
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
  if (nf) atomicOr(nonfinite, 1);
}

__global__ void se3_chain_bwd_kernel(const float* __restrict__ t, const float* __restrict__ w,
                                     const float* __restrict__ R_all, const float* __restrict__ dp,
                                     const float* __restrict__ dq, float* __restrict__ dt,
                                     float* __restrict__ dw, int B, int S, int order,
                                     int g0 = 0, int g1 = 1 << 30, const float* __restrict__ addt = nullptr,
                                     const float* __restrict__ addw = nullptr) {
  // g0, g1: dp / dq hold the rows g0 <= s < g1 of every batch item only ([B][g1 - g0][3 | 4]); the other rows' gradient is zero.
  // addt, addw: added to dt, dw (the gradient the increments get from the criterion's local terms)
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  g1 = min(g1, S);
  const int G = g1 - g0;
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
    const bool in_g = s >= g0 && s < g1;
    const size_t og = (size_t)b * G + (s - g0);
    if (dq && in_g) {
      const float* gq = dq + og * 4;
      if (order == 0) { g[0] = gq[0]; g[1] = gq[1]; g[2] = gq[2]; g[3] = gq[3]; }
      else { g[1] = gq[0]; g[2] = gq[1]; g[3] = gq[2]; g[0] = gq[3]; }
    }
    if (rot_is_valid(Ps)) {
      rot_to_quat_bwd(Ps, g, dP);
    } else {
      const M3 Qn = rot_project(Ps);
      M3 Gq;
#pragma unroll
      for (int i = 0; i < 9; ++i) Gq.m[i] = 0.f;
      rot_to_quat_bwd(Qn, g, Gq);
      rot_project_bwd(Qn, Gq, dP);
    }
    float gp[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) gp[i] = dpa[i] + ((dp && in_g) ? dp[og * 3 + i] : 0.f);
    const float* ts = t + o * 3;
    const float* wv = w + o * 3;
    // p_s = Pm t_s + p_{s-1}
    float* dto = dt + o * 3;
#pragma unroll
    for (int i = 0; i < 3; ++i)
      dto[i] = Pm.m[i] * gp[0] + Pm.m[3 + i] * gp[1] + Pm.m[6 + i] * gp[2] + (addt ? addt[o * 3 + i] : 0.f);
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
    if (addw)
      for (int i = 0; i < 3; ++i) dw[o * 3 + i] += addw[o * 3 + i];
    dP = dPm;
    dpa[0] = gp[0]; dpa[1] = gp[1]; dpa[2] = gp[2];
  }
}

// ---- geodesic rotation terms (BASELINE configs[4]; no reference counterpart: the reference has
// MSE on so(3) vectors / quaternion components only, losses/losses.py:71-85) -----------------------
// theta(a, b) = 2 atan2(|a ^ b|, |<a, b>|) for two 4-vectors: the rotation angle between the
// rotations two (unit) quaternions stand for, = 2 acos|<a,b>| = |log(Ra^T Rb)|, but conditioned at
// every angle (acos near 1 loses half the digits), invariant to the sign and the scale of either
// quaternion and to the component order (wxyz / xyzw).  |a ^ b|^2 = sum_{i<j} (a_i b_j - a_j b_i)^2.
// Loss term = mean over rows of theta^2 (the squared geodesic distance; MSE's analogue).
// Returns theta^2; g[i] = d(theta^2)/d a_i when g != nullptr.
__device__ __forceinline__ float geo_theta2(const float* a, const float* b, float* g) {
  float d = 0.f, n2 = 0.f, wb[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 4; ++i) d += a[i] * b[i];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (j == i) continue;
      const float w = a[i] * b[j] - a[j] * b[i];
      if (j > i) n2 += w * w;
      wb[i] += w * b[j];                 // (a ^ b) . b : gradient of |a ^ b|^2 / 2 w.r.t. a
    }
  const float n = sqrtf(n2), c = fabsf(d);
  const float half = atan2f(n, c);
  if (g) {
    const float den = n2 + c * c;
    if (!(den > 0.f)) {
      g[0] = g[1] = g[2] = g[3] = 0.f;
    } else {
      // d half = (c dn - n dc) / den ; dn = wb / n ; dc = sign(d) b ; half / n -> 1 / c as n -> 0
      const float hn = n > 1e-6f * c ? half / n : 1.f / c;
      const float A = 8.f * hn * c / den;
      const float Bc = 8.f * half * n / den * (d < 0.f ? -1.f : 1.f);
#pragma unroll
      for (int i = 0; i < 4; ++i) g[i] = A * wb[i] - Bc * b[i];
    }
  }
  return 4.f * half * half;
}

// unit quaternion (w, x, y, z) of exp(phi^): (cos(|phi|/2), sin(|phi|/2) phi/|phi|)
__device__ __forceinline__ void so3_to_quat(const float* phi, float* q) {
  const float t2 = phi[0] * phi[0] + phi[1] * phi[1] + phi[2] * phi[2];
  const float t = sqrtf(t2);
  float k;                               // sin(t/2)/t
  if (t < 1e-3f) k = 0.5f - t2 * (1.f / 48.f);
  else k = sinf(0.5f * t) / t;
  q[0] = cosf(0.5f * t); q[1] = k * phi[0]; q[2] = k * phi[1]; q[3] = k * phi[2];
}
// g = dL/dq -> dphi
__device__ __forceinline__ void so3_to_quat_bwd(const float* phi, const float* g, float* dphi) {
  const float t2 = phi[0] * phi[0] + phi[1] * phi[1] + phi[2] * phi[2];
  const float t = sqrtf(t2);
  float k, kp;                           // k = sin(t/2)/t ; kp = (dk/dt)/t
  if (t < 1e-3f) { k = 0.5f - t2 * (1.f / 48.f); kp = -1.f / 24.f + t2 * (1.f / 960.f); }
  else { const float sh = sinf(0.5f * t), ch = cosf(0.5f * t); k = sh / t; kp = (0.5f * ch * t - sh) / (t2 * t); }
  const float gv = g[1] * phi[0] + g[2] * phi[1] + g[3] * phi[2];
  // q_w = cos(t/2): dq_w/dphi = -k/2 phi ; q_v = k phi: dq_v/dphi = k I + kp phi phi^T
#pragma unroll
  for (int i = 0; i < 3; ++i) dphi[i] = -0.5f * k * g[0] * phi[i] + k * g[1 + i] + kp * gv * phi[i];
}

// mean over rows of theta^2; kind 1: rows of 3 (so(3) vectors), kind 3: rows of 4 (quaternions)
__device__ __forceinline__ float geo_row(int kind, const float* pr, const float* gt, float* grad) {
  if (kind == 3) return geo_theta2(pr, gt, grad);
  float qa[4], qb[4];
  so3_to_quat(pr, qa);
  so3_to_quat(gt, qb);
  if (!grad) return geo_theta2(qa, qb, nullptr);
  float gq[4];
  const float v = geo_theta2(qa, qb, gq);
  so3_to_quat_bwd(pr, gq, grad);
  return v;
}

// ---- pose loss: single block ------------------------------------------------------
// term i compares n[i] elements = rows of D[i] columns; row j of the term is row (j % R[i]) of batch item j / R[i]:
// pred[i] + (j / R) pbs + (j % R) prs, gt[i] + (j / R) gbs + (j % R) grs (element strides).  The plain entry points pass
// contiguous arrays (R = all rows, prs = grs = D); dlio_pose_tail_* read the predicted poses' rows g0..g1 and the columns of
// gt_f2f [B][S][6] / gt_f2g [B][S][7] in place.  dpred is always contiguous [n / D][D].
struct LossArgs {
  const float* pred[4];
  const float* gt[4];
  float* dpred[4];
  int32_t n[4];
  int32_t D[4], R[4], pbs[4], prs[4], gbs[4], grs[4];
};
__device__ __forceinline__ const float* loss_row(const float* base, int j, int R, int bs, int rs) {
  const int b = j / R;
  return base + (size_t)b * bs + (size_t)(j - b * R) * rs;
}

__global__ __launch_bounds__(256) void pose_loss_fwd_kernel(LossArgs a, const float* sx,
                                                            const float* sq, float beta, int mode,
                                                            float* out) {
  __shared__ double sm[16];
  __shared__ float mse[4];
  const bool geo = (mode & 2) != 0;
  mode &= 1;
  for (int i = 0; i < 4; ++i) {
    double s = 0.0;
    int cnt = a.n[i];
    if (geo && (i == 1 || i == 3)) {
      cnt = a.n[i] / a.D[i];
      for (int r = threadIdx.x; r < cnt; r += 256)
        s += (double)geo_row(i, loss_row(a.pred[i], r, a.R[i], a.pbs[i], a.prs[i]),
                             loss_row(a.gt[i], r, a.R[i], a.gbs[i], a.grs[i]), nullptr);
    } else {
      for (int e = threadIdx.x; e < a.n[i]; e += 256) {
        const int r = e / a.D[i], c = e - r * a.D[i];
        const float d = loss_row(a.pred[i], r, a.R[i], a.pbs[i], a.prs[i])[c] - loss_row(a.gt[i], r, a.R[i], a.gbs[i], a.grs[i])[c];
        s += (double)d * d;
      }
    }
    const double r = block_sum_d(s, sm);
    if (threadIdx.x == 0) mse[i] = cnt > 0 ? (float)(r / cnt) : 0.f;
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
                                                            float* dsx, float* dsq, int acc_hyper = 0) {
  const float gs = gscale ? gscale[0] : 1.f;
  const bool geo = (mode & 2) != 0;
  mode &= 1;
  float cx, cq;
  if (mode == 0) { cx = expf(-sx[0]); cq = expf(-sq[0]); }
  else { cx = 1.f; cq = beta; }
  for (int i = 0; i < 4; ++i) {
    if (a.n[i] <= 0 || !a.dpred[i]) continue;
    if (geo && (i == 1 || i == 3)) {
      const int w = a.D[i], rows = a.n[i] / w;
      const float coef = gs * cq / (float)rows;
      for (int r = threadIdx.x; r < rows; r += 256) {
        float g[4];
        geo_row(i, loss_row(a.pred[i], r, a.R[i], a.pbs[i], a.prs[i]), loss_row(a.gt[i], r, a.R[i], a.gbs[i], a.grs[i]), g);
        for (int k = 0; k < w; ++k) a.dpred[i][(size_t)r * w + k] = coef * g[k];
      }
      continue;
    }
    const float coef = gs * ((i == 0 || i == 2) ? cx : cq) * 2.f / (float)a.n[i];
    for (int e = threadIdx.x; e < a.n[i]; e += 256) {
      const int r = e / a.D[i], c = e - r * a.D[i];
      a.dpred[i][e] = coef * (loss_row(a.pred[i], r, a.R[i], a.pbs[i], a.prs[i])[c] - loss_row(a.gt[i], r, a.R[i], a.gbs[i], a.grs[i])[c]);
    }
  }
  if (threadIdx.x == 0 && mode == 0) {
    const float Lt = out[1], Lw = out[2], Lp = out[3], Lq = out[4];
    if (dsx) dsx[0] = (acc_hyper ? dsx[0] : 0.f) + gs * (1.f - (Lp + Lt) * cx);
    if (dsq) dsq[0] = (acc_hyper ? dsq[0] : 0.f) + gs * (1.f - (Lq + Lw) * cq);
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

extern "C" int dlio_so3_project(const float* R, float* Q, int32_t* valid, int n, dlio_stream_t stream) {
  if (!R || !Q || n <= 0) return DLIO_EINVAL;
  hipLaunchKernelGGL(so3_project_kernel, dim3(cdiv(n, 64)), dim3(64), 0, as_stream(stream), R, Q, valid, n);
  return dlio_check_launch();
}

extern "C" int dlio_so3_project_bwd(const float* R, const float* G, float* dR, int n, dlio_stream_t stream) {
  if (!R || !G || !dR || n <= 0) return DLIO_EINVAL;
  hipLaunchKernelGGL(so3_project_bwd_kernel, dim3(cdiv(n, 64)), dim3(64), 0, as_stream(stream), R, G, dR, n);
  return dlio_check_launch();
}

static int fill_loss_args(LossArgs& a, const float* const* pred, const float* const* gt,
                          const int32_t* n, float* const* dpred) {
  for (int i = 0; i < 4; ++i) {
    a.n[i] = n[i];
    a.pred[i] = pred[i];
    a.gt[i] = gt[i];
    a.dpred[i] = dpred ? dpred[i] : nullptr;
    a.D[i] = i == 3 ? 4 : 3;
    a.R[i] = n[i] / a.D[i] > 0 ? n[i] / a.D[i] : 1;
    a.pbs[i] = a.gbs[i] = 0;
    a.prs[i] = a.grs[i] = a.D[i];
    if (n[i] < 0 || n[i] % a.D[i] || (n[i] > 0 && (!pred[i] || !gt[i]))) return DLIO_EINVAL;
  }
  return DLIO_OK;
}

// the criterion's operands as views (see LossArgs): increments t, w [B][S][3], chained poses p [B][S][3], q [B][S][4], rows g0..g1
static int fill_tail_args(LossArgs& a, const float* t, const float* w, const float* p, const float* q, const float* gt_f2f,
                          const float* gt_f2g, int B, int S, int g0, int g1, int terms, float* const* dpred) {
  const int G = g1 - g0;
  const float* pred[4] = {t, w, p + (size_t)g0 * 3, q + (size_t)g0 * 4};
  const float* gt[4] = {gt_f2f, gt_f2f + 3, gt_f2g + (size_t)g0 * 7, gt_f2g + (size_t)g0 * 7 + 3};
  for (int i = 0; i < 4; ++i) {
    const bool local = i < 2;
    const bool on = (terms & (local ? 1 : 2)) != 0;
    a.D[i] = i == 3 ? 4 : 3;
    a.R[i] = local ? S : G;
    a.n[i] = on ? B * a.R[i] * a.D[i] : 0;
    a.pred[i] = pred[i];
    a.gt[i] = gt[i];
    a.dpred[i] = (dpred && on) ? dpred[i] : nullptr;
    a.pbs[i] = S * a.D[i];
    a.prs[i] = a.D[i];
    a.gbs[i] = S * (local ? 6 : 7);
    a.grs[i] = local ? 6 : 7;
  }
  return DLIO_OK;
}

extern "C" int dlio_pose_loss_fwd(const float* const* pred, const float* const* gt,
                                  const int32_t* n, const float* sx, const float* sq, float beta,
                                  int mode, float* out, dlio_stream_t stream) {
  if (!pred || !gt || !n || !out || ((mode & 1) == 0 && (!sx || !sq)) || mode < 0 || mode > 3)
    return DLIO_EINVAL;
  if ((mode & 2) && (n[1] % 3 || n[3] % 4)) return DLIO_EINVAL;
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
  if (!pred || !gt || !n || !out || !dpred || ((mode & 1) == 0 && (!sx || !sq)) || mode < 0 || mode > 3)
    return DLIO_EINVAL;
  if ((mode & 2) && (n[1] % 3 || n[3] % 4)) return DLIO_EINVAL;
  LossArgs a;
  int rc = fill_loss_args(a, pred, gt, n, dpred);
  if (rc) return rc;
  hipLaunchKernelGGL(pose_loss_bwd_kernel, dim3(1), dim3(256), 0, as_stream(stream), a, sx, sq,
                     beta, mode, out, gscale, dsx, dsq);
  return dlio_check_launch();
}

// ---- Trainer's tail: SE(3) chain + criterion, two launches each way ------------------------------------------------------------
extern "C" int dlio_pose_tail_fwd(const float* t, const float* w, const float* gt_f2f, const float* gt_f2g, int B, int S,
                                  int g0, int g1, int terms, const float* sx, const float* sq, float beta, int mode,
                                  int order, float* p, float* q, float* R_all, int32_t* status, int32_t* nonfinite,
                                  float* out, dlio_stream_t stream) {
  if (!t || !w || !gt_f2f || !gt_f2g || !p || !q || !R_all || !out || B <= 0 || S <= 0 || g0 < 0 || g1 > S || g0 >= g1 ||
      terms < 1 || terms > 3 || ((mode & 1) == 0 && (!sx || !sq)) || mode < 0 || mode > 3 || (order != 0 && order != 1))
    return DLIO_EINVAL;
  hipLaunchKernelGGL(se3_chain_fwd_kernel, dim3(cdiv(B, 64)), dim3(64), 0, as_stream(stream), t, w, p, q, R_all, status, B, S,
                     order, nonfinite);
  int rc = dlio_check_launch();
  if (rc) return rc;
  LossArgs a;
  fill_tail_args(a, t, w, p, q, gt_f2f, gt_f2g, B, S, g0, g1, terms, nullptr);
  hipLaunchKernelGGL(pose_loss_fwd_kernel, dim3(1), dim3(256), 0, as_stream(stream), a, sx, sq, beta, mode, out);
  return dlio_check_launch();
}

extern "C" int dlio_pose_tail_bwd(const float* t, const float* w, const float* gt_f2f, const float* gt_f2g, int B, int S,
                                  int g0, int g1, int terms, const float* sx, const float* sq, float beta, int mode,
                                  int order, const float* p, const float* q, const float* R_all, const float* out,
                                  const float* gscale, float* ws, float* dt, float* dw, float* dsx, float* dsq,
                                  int acc_hyper, dlio_stream_t stream) {
  if (!t || !w || !gt_f2f || !gt_f2g || !p || !q || !R_all || !out || !ws || !dt || !dw || B <= 0 || S <= 0 || g0 < 0 ||
      g1 > S || g0 >= g1 || terms < 1 || terms > 3 || ((mode & 1) == 0 && (!sx || !sq)) || mode < 0 || mode > 3 ||
      (order != 0 && order != 1))
    return DLIO_EINVAL;
  // ws: [B][S][3] x 2 (the local terms' gradients) + [B][G][3] + [B][G][4] (the global terms'); see dlio_pose_tail_ws_floats
  const int G = g1 - g0;
  float* dpred[4] = {ws, ws + (size_t)B * S * 3, ws + (size_t)B * S * 6, ws + (size_t)B * S * 6 + (size_t)B * G * 3};
  LossArgs a;
  fill_tail_args(a, t, w, p, q, gt_f2f, gt_f2g, B, S, g0, g1, terms, dpred);
  hipLaunchKernelGGL(pose_loss_bwd_kernel, dim3(1), dim3(256), 0, as_stream(stream), a, sx, sq, beta, mode, out, gscale, dsx,
                     dsq, acc_hyper);
  int rc = dlio_check_launch();
  if (rc) return rc;
  const bool loc = (terms & 1) != 0, glob = (terms & 2) != 0;
  hipLaunchKernelGGL(se3_chain_bwd_kernel, dim3(cdiv(B, 64)), dim3(64), 0, as_stream(stream), t, w, R_all,
                     glob ? dpred[2] : nullptr, glob ? dpred[3] : nullptr, dt, dw, B, S, order, g0, g1,
                     loc ? dpred[0] : nullptr, loc ? dpred[1] : nullptr);
  return dlio_check_launch();
}

extern "C" size_t dlio_pose_tail_ws_floats(int B, int S, int g0, int g1) {
  return (size_t)B * S * 6 + (size_t)B * (g1 - g0) * 7;
}
