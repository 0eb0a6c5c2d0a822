// Spherical (range-image) projection of a lidar scan and image-space normal estimation:
// the step immediately in front of the training path.
//
// Replaces deeplio/common/laserscan.py:
//   LaserScan.do_range_projection  :122-185   -> dlio_scan_project
//   LaserScan.do_normal_projection :215-248   -> dlio_scan_normals
// and deeplio/datasets/kitti.py:
//   get_velo_image :83-97 + transform_images :345-364 (normalised branch) -> dlio_velo_image
//
// Index arithmetic follows the reference's float32 numpy expression operation by operation
// (no FMA contraction, IEEE division and sqrt).  atan2/asin are evaluated in fp64 and rounded to
// fp32, i.e. correctly rounded; numpy's own float32 arctan2/arcsin are NOT (SVML on AVX512
// hosts, glibc elsewhere: up to a few ulp apart from each other), so the reference's indices
// differ between hosts for ~2e-5 of the points; on the committed golden the HIP indices are
// bit-identical (tests/test_projection.py).
//
// "closest point wins" (the reference scatters in order of decreasing depth) is a 64-bit
// atomicMin on (depth bits << 32 | point index): deterministic, equal depths resolved towards
// the smaller index (the reference's argsort leaves ties undefined).
#include "common.h"
#include <math.h>

// float32 arithmetic exactly as numpy evaluates it: no FMA contraction anywhere in this file,
// and IEEE-correct sqrt / divide independent of compiler flags (through fp64: for float operands
// the double result rounded to float is the correctly rounded float result, 53 >= 2*24+2).
#pragma clang fp contract(off)

namespace {

__device__ __forceinline__ float mul_rn(float a, float b) { return a * b; }
__device__ __forceinline__ float add_rn(float a, float b) { return a + b; }
__device__ __forceinline__ float sub_rn(float a, float b) { return a - b; }
__device__ __forceinline__ float div_rn(float a, float b) { return (float)((double)a / (double)b); }
__device__ __forceinline__ float sqrt_rn(float a) { return (float)sqrt((double)a); }

constexpr float PI_F = 3.14159274101257324219f;   // float32(np.pi)

__global__ void proj_init_kernel(unsigned long long* __restrict__ keys, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) keys[i] = ~0ull;
}

__global__ void proj_points_kernel(const float* __restrict__ pts, int N, int H, int W,
                                   float abs_fov_down, float fov, int32_t* __restrict__ proj_x,
                                   int32_t* __restrict__ proj_y, float* __restrict__ unproj_range,
                                   unsigned long long* __restrict__ keys) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const float x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
  // np.linalg.norm(points, 2, axis=1): sqrt(add.reduce(x*x)) in float32
  const float depth = sqrt_rn(add_rn(add_rn(mul_rn(x, x), mul_rn(y, y)), mul_rn(z, z)));
  const float yaw = -(float)atan2((double)y, (double)x);
  const float pitch = (float)asin((double)div_rn(z, depth));
  float px = mul_rn(0.5f, add_rn(div_rn(yaw, PI_F), 1.0f));
  float py = sub_rn(1.0f, div_rn(add_rn(pitch, abs_fov_down), fov));
  px = floorf(mul_rn(px, (float)W));
  py = floorf(mul_rn(py, (float)H));
  px = fmaxf(0.f, fminf((float)(W - 1), px));      // NaN (zero-depth point) -> 0
  py = fmaxf(0.f, fminf((float)(H - 1), py));
  const int ix = (int)px, iy = (int)py;
  proj_x[i] = ix;
  proj_y[i] = iy;
  unproj_range[i] = depth;
  const unsigned long long key = ((unsigned long long)__float_as_uint(depth) << 32) | (unsigned)i;
  atomicMin(&keys[(size_t)iy * W + ix], key);
}

__global__ void proj_gather_kernel(const unsigned long long* __restrict__ keys,
                                   const float* __restrict__ pts, const float* __restrict__ rem,
                                   int HW, float* __restrict__ proj_range,
                                   float* __restrict__ proj_xyz, float* __restrict__ proj_rem,
                                   int32_t* __restrict__ proj_idx, int32_t* __restrict__ proj_mask) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= HW) return;
  const unsigned long long k = keys[p];
  // laserscan.py:27-60 reset(): empty pixels are 0 everywhere (the "-1" comments are stale)
  float r = 0.f, x = 0.f, y = 0.f, z = 0.f, rm = 0.f;
  int idx = 0;
  if (k != ~0ull) {
    idx = (int)(unsigned)(k & 0xffffffffu);
    r = __uint_as_float((unsigned)(k >> 32));
    x = pts[3 * idx]; y = pts[3 * idx + 1]; z = pts[3 * idx + 2];
    rm = rem ? rem[idx] : 0.f;
  }
  proj_range[p] = r;
  proj_xyz[3 * p] = x; proj_xyz[3 * p + 1] = y; proj_xyz[3 * p + 2] = z;
  proj_rem[p] = rm;
  proj_idx[p] = idx;
  if (proj_mask) proj_mask[p] = idx > 0;      // :185 (point 0 counts as empty, as in the reference)
}

struct V4 { float x, y, z, r; };
__device__ __forceinline__ V4 ld4(const float* xyz, const float* rng, int p) {
  return V4{xyz[3 * p], xyz[3 * p + 1], xyz[3 * p + 2], rng[p]};
}
__device__ __forceinline__ V4 sub4(V4 a, V4 b) {
  return V4{sub_rn(a.x, b.x), sub_rn(a.y, b.y), sub_rn(a.z, b.z), sub_rn(a.r, b.r)};
}
// np.cross(a, b) of two float32 vectors: a1*b2 - a2*b1, ... without contraction
__device__ __forceinline__ void cross3(const float (&a)[3], const float (&b)[3], float (&c)[3]) {
  c[0] = sub_rn(mul_rn(a[1], b[2]), mul_rn(a[2], b[1]));
  c[1] = sub_rn(mul_rn(a[2], b[0]), mul_rn(a[0], b[2]));
  c[2] = sub_rn(mul_rn(a[0], b[1]), mul_rn(a[1], b[0]));
}

__global__ void normals_kernel(const float* __restrict__ xyz, const float* __restrict__ rng,
                               float* __restrict__ out, int H, int W) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= H * W) return;
  const int r = p / W, c = p - r * W;
  float n[3] = {0.f, 0.f, 0.f};
  if (r > 0 && r < H - 1 && c > 0 && c < W - 1) {      // border = np.pad zeros (:247)
    const V4 m = ld4(xyz, rng, p);
    // :226-233  top = img[r-1]-img[r], bottom = -(img[r]-img[r+1]), left = img[c-1]-img[c],
    //           right = -(img[c]-img[c+1])
    const V4 t = sub4(ld4(xyz, rng, p - W), m);
    V4 b = sub4(m, ld4(xyz, rng, p + W));
    const V4 l = sub4(ld4(xyz, rng, p - 1), m);
    V4 rr = sub4(m, ld4(xyz, rng, p + 1));
    b = V4{-b.x, -b.y, -b.z, -b.r};
    rr = V4{-rr.x, -rr.y, -rr.z, -rr.r};
    // :221-222,236  w = exp(-0.8 * |range diff|)
    const float wt = expf(mul_rn(-0.8f, fabsf(t.r))), wl = expf(mul_rn(-0.8f, fabsf(l.r)));
    const float wb = expf(mul_rn(-0.8f, fabsf(b.r))), wr = expf(mul_rn(-0.8f, fabsf(rr.r)));
    const float vt[3] = {mul_rn(wt, t.x), mul_rn(wt, t.y), mul_rn(wt, t.z)};
    const float vl[3] = {mul_rn(wl, l.x), mul_rn(wl, l.y), mul_rn(wl, l.z)};
    const float vb[3] = {mul_rn(wb, b.x), mul_rn(wb, b.y), mul_rn(wb, b.z)};
    const float vr[3] = {mul_rn(wr, rr.x), mul_rn(wr, rr.y), mul_rn(wr, rr.z)};
    float c0[3], c1[3], c2[3], c3[3];
    cross3(vt, vl, c0);      // :238-241
    cross3(vl, vb, c1);
    cross3(vb, vr, c2);
    cross3(vr, vt, c3);
#pragma unroll
    for (int k = 0; k < 3; ++k) n[k] = add_rn(add_rn(add_rn(c0[k], c1[k]), c2[k]), c3[k]);
    const float nn = add_rn(sqrt_rn(add_rn(add_rn(mul_rn(n[0], n[0]), mul_rn(n[1], n[1])),
                                                    mul_rn(n[2], n[2]))), 1e-8f);
#pragma unroll
    for (int k = 0; k < 3; ++k) n[k] = div_rn(n[k], nn);
  }
  out[3 * p] = n[0]; out[3 * p + 1] = n[1]; out[3 * p + 2] = n[2];
}

struct VeloSel { int ch[8]; float mean[8]; };

// image = dstack(xyz / max_depth, remission, normal, range)  (kitti.py:91-96), then
// crop, HWC -> CHW, subtract the per-channel mean, select channels (kitti.py:345-364)
__global__ void velo_image_kernel(const float* __restrict__ xyz, const float* __restrict__ rem,
                                  const float* __restrict__ nrm, const float* __restrict__ rng,
                                  float max_depth, VeloSel sel, int nch, int H, int W, int ct, int cl,
                                  float* __restrict__ out) {
  const int OH = H - 2 * ct, OW = W - 2 * cl;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= OH * OW) return;
  const int oh = i / OW, ow = i - oh * OW;
  const int p = (oh + ct) * W + ow + cl;
  float v[8];
  v[0] = div_rn(xyz[3 * p], max_depth); v[1] = div_rn(xyz[3 * p + 1], max_depth);
  v[2] = div_rn(xyz[3 * p + 2], max_depth);
  v[3] = rem[p];
  v[4] = nrm[3 * p]; v[5] = nrm[3 * p + 1]; v[6] = nrm[3 * p + 2];
  v[7] = rng[p];
  for (int k = 0; k < nch; ++k) {
    const int c = sel.ch[k];
    float val = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) val = c == q ? v[q] : val;
    out[(size_t)k * OH * OW + i] = sub_rn(val, sel.mean[k]);
  }
}

}  // namespace

extern "C" size_t dlio_scan_project_ws_bytes(int H, int W) {
  if (H <= 0 || W <= 0) return 0;
  return (size_t)H * W * sizeof(unsigned long long);
}

extern "C" int dlio_scan_project(const float* points, const float* remissions, int N, int H, int W,
                                 double fov_up_deg, double fov_down_deg, int32_t* proj_x,
                                 int32_t* proj_y, float* unproj_range, float* proj_range,
                                 float* proj_xyz, float* proj_remission, int32_t* proj_idx,
                                 int32_t* proj_mask, void* ws, size_t ws_bytes,
                                 dlio_stream_t stream) {
  if (!proj_range || !proj_xyz || !proj_remission || !proj_idx || !ws || N < 0 || H <= 0 || W <= 0)
    return DLIO_EINVAL;
  if (N > 0 && (!points || !proj_x || !proj_y || !unproj_range)) return DLIO_EINVAL;
  if (ws_bytes < dlio_scan_project_ws_bytes(H, W)) return DLIO_EWS;
  // laserscan.py:129-131, evaluated in double like the python floats, used as float32 arrays' peers
  const double fu = fov_up_deg / 180.0 * M_PI, fd = fov_down_deg / 180.0 * M_PI;
  const double fov = fabs(fd) + fabs(fu);
  if (!(fov > 0.0)) return DLIO_EINVAL;
  hipStream_t s = as_stream(stream);
  unsigned long long* keys = reinterpret_cast<unsigned long long*>(ws);
  const int HW = H * W;
  hipLaunchKernelGGL(proj_init_kernel, dim3(cdiv(HW, 256)), dim3(256), 0, s, keys, HW);
  int rc = dlio_check_launch();
  if (rc) return rc;
  if (N > 0) {
    hipLaunchKernelGGL(proj_points_kernel, dim3(cdiv(N, 256)), dim3(256), 0, s, points, N, H, W,
                       (float)fabs(fd), (float)fov, proj_x, proj_y, unproj_range, keys);
    rc = dlio_check_launch();
    if (rc) return rc;
  }
  hipLaunchKernelGGL(proj_gather_kernel, dim3(cdiv(HW, 256)), dim3(256), 0, s, keys, points,
                     remissions, HW, proj_range, proj_xyz, proj_remission, proj_idx, proj_mask);
  return dlio_check_launch();
}

extern "C" int dlio_scan_normals(const float* proj_xyz, const float* proj_range, float* normals,
                                 int H, int W, dlio_stream_t stream) {
  if (!proj_xyz || !proj_range || !normals || H <= 0 || W <= 0) return DLIO_EINVAL;
  hipLaunchKernelGGL(normals_kernel, dim3(cdiv(H * W, 256)), dim3(256), 0, as_stream(stream),
                     proj_xyz, proj_range, normals, H, W);
  return dlio_check_launch();
}

extern "C" int dlio_velo_image(const float* proj_xyz, const float* proj_remission,
                               const float* normals, const float* proj_range, float max_depth,
                               const int32_t* channels, const float* mean, int n_channels, int H,
                               int W, int crop_top, int crop_left, float* out,
                               dlio_stream_t stream) {
  if (!proj_xyz || !proj_remission || !normals || !proj_range || !channels || !out ||
      n_channels <= 0 || n_channels > 8 || H <= 0 || W <= 0 || crop_top < 0 || crop_left < 0 ||
      2 * crop_top >= H || 2 * crop_left >= W || !(max_depth > 0.f))
    return DLIO_EINVAL;
  VeloSel sel;
  for (int k = 0; k < 8; ++k) { sel.ch[k] = 0; sel.mean[k] = 0.f; }
  for (int k = 0; k < n_channels; ++k) {
    if (channels[k] < 0 || channels[k] > 7) return DLIO_EINVAL;
    sel.ch[k] = channels[k];
    sel.mean[k] = mean ? mean[channels[k]] : 0.f;     // mean is indexed by ORIGINAL channel (:359-360)
  }
  const int n = (H - 2 * crop_top) * (W - 2 * crop_left);
  hipLaunchKernelGGL(velo_image_kernel, dim3(cdiv(n, 256)), dim3(256), 0, as_stream(stream), proj_xyz,
                     proj_remission, normals, proj_range, max_depth, sel, n_channels, H, W, crop_top,
                     crop_left, out);
  return dlio_check_launch();
}
