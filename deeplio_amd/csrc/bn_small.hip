// Train-mode BatchNorm2d (+ ReLU, + residual) of SMALL feature maps in one launch, forward and backward
// (pointseg_modules.py:98-106 in fire_blk4 / fire_blk5: 32 x 64 and 16 x 32 maps).  bn.hip's two-launch form
// (statistics partials, then an apply kernel that finalises them) reads the tensor twice and costs two dependent
// launches of 5-15 us; these blocks are the serial valley of the step (DESIGN: 14 % of the time for 5 % of the work),
// where launch count is what matters.  Here ONE workgroup of 16 waves owns a channel: wave w holds image n = w of the
// channel in registers (V float4 per lane: N <= 16, H * W = 256 V, V in {1, 2, 4, 8}), the statistics are a wave
// reduction + 16 partials through LDS, the apply runs on the registers.  Each element is read ONCE.
// A launch may cover two layers' channels at once (the expand1x1 / expand3x3 halves of a Fire block's concat buffer:
// channels [0, C1) take the first parameter set, [C1, C) the second), so a Fire block's two expand BatchNorms are one launch.
// Accumulation: per float4 in fp32, across float4 / lanes / waves in fp64 (as bn.hip), fixed order.
#include "common.h"
#include <map>
#include <mutex>
#include <utility>

namespace {

struct BnSet {            // one layer's parameters (device pointers, nullable where noted)
  const float* gamma;     // nullable
  const float* beta;      // nullable
  float* running_mean;    // nullable
  float* running_var;     // nullable
  float* dgamma;          // backward, nullable
  float* dbeta;           // backward, nullable
};

__device__ __forceinline__ double block16_sum(double v, double* sm, int wave, int lane) {
  v = wave_sum_d(v);
  __syncthreads();
  if (lane == 0) sm[wave] = v;
  __syncthreads();
  double r = 0.0;
#pragma unroll
  for (int i = 0; i < 16; ++i) r += sm[i];
  return r;                                         // every thread has the total
}

template <int V>
__global__ __launch_bounds__(1024) void bn_small_fwd_kernel(
    const float* __restrict__ x, int x_ctot, int x_coff, int N, int C1, BnSet s1, BnSet s2, float eps, float momentum,
    float* __restrict__ mean_o, float* __restrict__ invstd_o, float* __restrict__ scale_o, float* __restrict__ shift_o,
    const float* residual, int r_ctot, int r_coff, const float* __restrict__ r_mean, const float* __restrict__ r_scale,
    const float* __restrict__ r_shift, float* y, int y_ctot, int y_coff, float* __restrict__ gap_out, int gap_ctot,
    int gap_coff, int post_relu, float* __restrict__ amax_out = nullptr) {
  constexpr int HW = 256 * V;
  __shared__ double sm[2][16];
  __shared__ unsigned s_amax;
  const int c = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const BnSet& ps = c < C1 ? s1 : s2;
  const int cl = c < C1 ? c : c - C1;
  const bool have = wave < N;
  float4 v[V];
  const float* xp = x + ((size_t)(have ? wave : 0) * x_ctot + x_coff + c) * HW;
  double a = 0.0, b = 0.0;
#pragma unroll
  for (int j = 0; j < V; ++j) {
    v[j] = have ? *reinterpret_cast<const float4*>(xp + 4 * (lane + 64 * j)) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float f0 = (v[j].x + v[j].y) + (v[j].z + v[j].w);
    const float f1 = (v[j].x * v[j].x + v[j].y * v[j].y) + (v[j].z * v[j].z + v[j].w * v[j].w);
    a += f0; b += f1;
  }
  a = block16_sum(a, sm[0], wave, lane);
  b = block16_sum(b, sm[1], wave, lane);
  const double count = (double)N * HW;
  const double m = a / count;
  double var = b / count - m * m;
  if (var < 0.0) var = 0.0;
  const float is = (float)(1.0 / sqrt(var + (double)eps));
  const float mu = (float)m, sc = (ps.gamma ? ps.gamma[cl] : 1.f) * is, be = ps.beta ? ps.beta[cl] : 0.f;
  if (threadIdx.x == 0) {
    mean_o[c] = mu; invstd_o[c] = is; scale_o[c] = sc;
    if (shift_o) shift_o[c] = be;
    if (ps.running_mean) ps.running_mean[cl] = (1.f - momentum) * ps.running_mean[cl] + momentum * mu;
    if (ps.running_var) {
      const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
      ps.running_var[cl] = (1.f - momentum) * ps.running_var[cl] + momentum * (float)unb;
    }
  }
  if (!y) return;                                   // statistics only (apply-on-load consumers)
  float amax = 0.f;
  if (have) {
  const float* rp = residual ? residual + ((size_t)wave * r_ctot + r_coff + c) * HW : nullptr;
  const bool raff = rp && r_scale;
  const float rmu = raff ? r_mean[r_coff + c] : 0.f, rsc = raff ? r_scale[r_coff + c] : 1.f, rsh = raff ? r_shift[r_coff + c] : 0.f;
  float* yp = y + ((size_t)wave * y_ctot + y_coff + c) * HW;
  double gs = 0.0;
#pragma unroll
  for (int j = 0; j < V; ++j) {
    float e[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
    float4 rv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (rp) rv = *reinterpret_cast<const float4*>(rp + 4 * (lane + 64 * j));
    float re[4] = {rv.x, rv.y, rv.z, rv.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float o = (e[k] - mu) * sc + be;
      if (post_relu) o = fmaxf(o, 0.f);
      if (raff) re[k] = fmaxf((re[k] - rmu) * rsc + rsh, 0.f);
      e[k] = o + re[k];
    }
    *reinterpret_cast<float4*>(yp + 4 * (lane + 64 * j)) = make_float4(e[0], e[1], e[2], e[3]);
    gs += (double)((e[0] + e[1]) + (e[2] + e[3]));
    amax = amax4(amax, e[0], e[1], e[2], e[3]);
  }
  if (gap_out) {                                     // plane average of the OUTPUT (the SELayer behind the block): a wave = a plane
    gs = wave_sum_d(gs);
    if (lane == 0) gap_out[(size_t)wave * gap_ctot + gap_coff + c] = (float)(gs / (double)HW);
  }
  }
  if (amax_out) block_amax_commit(amax, amax_out, &s_amax);      // (every wave: the images a workgroup does not have add 0)
}

// dx = scale * (g - mean(g) - xhat * mean(g * xhat)), g = dy where the activated output was > 0 (post_relu);
// channels [0, C1) -> dx1 [N][C1][HW], [C1, C) -> dx2 [N][C - C1][HW]; dgamma / dbeta per set (accumulated when asked)
template <int V>
__global__ __launch_bounds__(1024) void bn_small_bwd_kernel(
    const float* __restrict__ dy, int dy_ctot, int dy_coff, const float* __restrict__ x, int x_ctot, int x_coff, int N, int C,
    int C1, BnSet s1, BnSet s2, const float* __restrict__ mean, const float* __restrict__ invstd,
    const float* __restrict__ scale, float* __restrict__ dx1, float* __restrict__ dx2, int accumulate, int post_relu,
    float* __restrict__ amax_out) {
  constexpr int HW = 256 * V;
  __shared__ double sm[2][16];
  __shared__ unsigned wg_amax;
  if (threadIdx.x == 0) wg_amax = 0u;                 // (block16_sum's barriers order this before the atomics below)
  const int c = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const BnSet& ps = c < C1 ? s1 : s2;
  const int cl = c < C1 ? c : c - C1;
  const bool have = wave < N;
  const float mu = mean[c], is = invstd[c], sc = scale[c], be = ps.beta ? ps.beta[cl] : 0.f;
  const float* gp = dy + ((size_t)(have ? wave : 0) * dy_ctot + dy_coff + c) * HW;
  const float* xp = x + ((size_t)(have ? wave : 0) * x_ctot + x_coff + c) * HW;
  float4 g[V], xh[V];
  double sg = 0.0, sgx = 0.0;
#pragma unroll
  for (int j = 0; j < V; ++j) {
    g[j] = have ? *reinterpret_cast<const float4*>(gp + 4 * (lane + 64 * j)) : make_float4(0.f, 0.f, 0.f, 0.f);
    xh[j] = have ? *reinterpret_cast<const float4*>(xp + 4 * (lane + 64 * j)) : make_float4(mu, mu, mu, mu);
  }
#pragma unroll
  for (int j = 0; j < V; ++j) {
    float ge[4] = {g[j].x, g[j].y, g[j].z, g[j].w};
    float xe[4] = {xh[j].x, xh[j].y, xh[j].z, xh[j].w};
    float f0 = 0.f, f1 = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (post_relu && !((xe[k] - mu) * sc + be > 0.f)) ge[k] = 0.f;
      xe[k] = (xe[k] - mu) * is;
      f0 += ge[k]; f1 += ge[k] * xe[k];
    }
    g[j] = make_float4(ge[0], ge[1], ge[2], ge[3]);
    xh[j] = make_float4(xe[0], xe[1], xe[2], xe[3]);
    sg += f0; sgx += f1;
  }
  sg = block16_sum(sg, sm[0], wave, lane);
  sgx = block16_sum(sgx, sm[1], wave, lane);
  if (threadIdx.x == 0) {
    if (ps.dbeta) ps.dbeta[cl] = accumulate ? ps.dbeta[cl] + (float)sg : (float)sg;
    if (ps.dgamma) ps.dgamma[cl] = accumulate ? ps.dgamma[cl] + (float)sgx : (float)sgx;
  }
  const double inv_cnt = 1.0 / ((double)N * HW);
  const float mg = (float)(sg * inv_cnt), mgx = (float)(sgx * inv_cnt);
  float* op = c < C1 ? dx1 + ((size_t)wave * C1 + c) * HW : dx2 + ((size_t)wave * (C - C1) + cl) * HW;
  float amax = 0.f;
#pragma unroll
  for (int j = 0; j < V; ++j) {
    if (!have) break;
    float4 o;
    o.x = sc * (g[j].x - mg - xh[j].x * mgx); o.y = sc * (g[j].y - mg - xh[j].y * mgx);
    o.z = sc * (g[j].z - mg - xh[j].z * mgx); o.w = sc * (g[j].w - mg - xh[j].w * mgx);
    *reinterpret_cast<float4*>(op + 4 * (lane + 64 * j)) = o;
    amax = fmaxf(amax, fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fmaxf(fabsf(o.z), fabsf(o.w))));
  }
  if (amax_out) {                                     // largest |dx| (for the two-piece split kernels, as bn_coop_bwd_kernel)
    // ONE atomic per workgroup: the 16 waves x C workgroups of a launch on one address (12 k serialised atomics at fire_blk5)
    // cost 0.7 ms per step
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
    if (lane == 0) atomicMax(&wg_amax, __float_as_uint(amax));
  }
  __syncthreads();
  if (amax_out && threadIdx.x == 0 && wg_amax) atomicMax(reinterpret_cast<unsigned*>(amax_out), wg_amax);
}

// ---- the same for LARGE feature maps: a channel spread over N cooperating workgroups ---------------------------------
// fire_blk1-3 hold 0.5-2 MB per channel -- more than a CU's register file.  Here a workgroup owns 1 / P of an (n, c) plane (T
// threads x V float4, 32 KB per operand) and the N P workgroups of a channel exchange their partial sums through global
// memory: publish (sum, sum of squares) into the item's own slot pair -> poll the channel's slots -> every workgroup adds
// the N P partials in the same fixed order -> apply from registers.  The tensor is read ONCE (bn.hip's two launches read it
// twice: statistics, then apply; three times against five in backward).
//
// Work distribution (round 5): items (c, part) are handed out IN ORDER by a ticket counter, so the only thing progress needs
// is that N P workgroups of the launch are resident at the same time -- not, as with the static item -> workgroup map of round
// 4, that the WHOLE grid is (which capped a launch at 104 of the 256 CUs so that two of them always fit the chip).  A
// workgroup that holds ticket t knows every ticket below t is held by a workgroup that is running; it only ever waits for
// partners of its own channel, publishes everything it holds of that channel before it waits, and tickets of one workgroup
// increase -- so waits only point at equal or lower channels and the lowest unfinished channel always completes.  The grid
// is whatever the occupancy query says fits the chip; workgroups that are not resident yet simply have not drawn a ticket.
//
// Software pipeline: a workgroup draws its NEXT ticket at the top of an item and issues that item's loads right after it has
// published its own sums, so the exchange (a few microseconds of polling) and the stores of item i run under the loads of
// item i + 1 (round 4: load-all -> publish -> spin -> apply -> store strictly in sequence, no loads in flight during the
// exchange).  Should the next item belong to the SAME channel (few resident workgroups), its sums are published before the
// wait as well -- otherwise the workgroup would wait for itself.
// A bounded spin (COOP_SPIN_LIMIT) turns a broken assumption into wrong numbers and an error flag (sync[0]) instead of a
// hung GPU; ops.bn_coop_check() reads the flag, re-initialises the scratch and falls back to the two-launch kernels.
// Slots and counters are restored by the last workgroup that leaves a channel / the launch, so the scratch is initialised
// only once.  sync: [0] error flag, [1] next ticket, [2] workgroups that have left, [4 + c] departures of channel c.
constexpr int COOP_SPIN_LIMIT = 1 << 20;
constexpr int COOP_HDR = 4;
// A slot of `part` is its own arrival flag: it holds COOP_EMPTY (a NaN pattern no sum can produce) until its owner stores
// the partial sum -- one relaxed 64-bit atomic store at agent scope -- and the partners poll the slots themselves (relaxed
// atomic loads at agent scope: they bypass the non-coherent per-XCD L2).  No release / acquire: on gfx950 an agent-scope
// acquire is a buffer_inv of the L2 and a release a write-back of it, per spin iteration (measured: 60 us per exchange).
constexpr unsigned long long COOP_EMPTY = 0x7ff8dead0badbeefull;

__device__ __forceinline__ unsigned long long coop_poll(unsigned long long* p, int* errflag, int code = 1) {
  unsigned long long v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  int spins = 0;
  while (v == COOP_EMPTY) {
    __builtin_amdgcn_s_sleep(2);
    v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // (the flag says who gave up: 1 the exchange of the sums, 2 the plane sums of a cut plane; sync[3] = the slot's index)
    if (++spins > COOP_SPIN_LIMIT) {
      __hip_atomic_store(errflag, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return 0ull;
    }
  }
  return v;
}

constexpr int COOP_MAX_NP = 256;          // cooperating workgroups per channel (N * parts); the slot regions are sized for it

// thread 0 publishes this item's pair of sums in the item's slots
__device__ __forceinline__ void coop_publish(double a, double b, double* part, int c, int np, int NP) {
  unsigned long long* slots = reinterpret_cast<unsigned long long*>(part) + ((size_t)c * NP + np) * 2;
  __hip_atomic_store(slots + 0, __builtin_bit_cast(unsigned long long, a), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(slots + 1, __builtin_bit_cast(unsigned long long, b), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// the lanes of wave 0 poll the channel's 2 NP slots side by side (one after the other it was 2 NP dependent memory round
// trips: 30 us per exchange at N = 16); every thread adds them in slot order: same order, same result everywhere
__device__ __forceinline__ void coop_gather(double& a, double& b, double* part, int* sync, int c, int NP,
                                            double* vals /* LDS, 2 NP doubles */) {
  unsigned long long* slots = reinterpret_cast<unsigned long long*>(part) + (size_t)c * NP * 2;
  if (threadIdx.x < 64)
    for (int k = threadIdx.x; k < 2 * NP; k += 64) vals[k] = __builtin_bit_cast(double, coop_poll(slots + k, sync));
  __syncthreads();
  double ta = 0.0, tb = 0.0;
  for (int k = 0; k < NP; ++k) { ta += vals[2 * k]; tb += vals[2 * k + 1]; }
  a = ta; b = tb;
}

// publish + gather of one item (kernels that hold one item at a time)
__device__ __forceinline__ void coop_exchange(double& a, double& b, double* part, int* sync, int c, int np, int NP, double* vals) {
  if (threadIdx.x == 0) coop_publish(a, b, part, c, np, NP);
  coop_gather(a, b, part, sync, c, NP, vals);
}

__device__ __forceinline__ void coop_depart(double* part, int* sync, int c, int NP, int gapC = 0, int C = 0, bool oneshot = false) {
  // every partner has gathered before it departs (its loads returned before the barrier behind the gather): the last
  // one out empties the channel's slots and clears the counter for the next launch
  // (gapC = C when the parts of a plane exchanged their plane sums through the third slot region: emptied as well)
  if (threadIdx.x == 0) {
    const int d = __hip_atomic_fetch_add(sync + COOP_HDR + c, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (d == NP - 1) {
      unsigned long long* slots = reinterpret_cast<unsigned long long*>(part) + (size_t)c * NP * 2;
      if (gapC) {
        unsigned long long* gsl = reinterpret_cast<unsigned long long*>(part) + (size_t)gapC * COOP_MAX_NP * 2 + (size_t)c * NP;
        for (int k = 0; k < NP; ++k) __hip_atomic_store(gsl + k, COOP_EMPTY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      for (int k = 0; k < 2 * NP; ++k) __hip_atomic_store(slots + k, COOP_EMPTY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(sync + COOP_HDR + c, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // one item per workgroup, grid = items: when the LAST channel is complete every ticket of the launch has been drawn
      if (oneshot && c == C - 1) __hip_atomic_store(sync + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// thread 0 draws the next ticket into *s_tk (the value is read behind a later barrier)
__device__ __forceinline__ void coop_draw(int* sync, int* s_tk) {
  if (threadIdx.x == 0) *s_tk = __hip_atomic_fetch_add(sync + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int coop_draw_sync(int* sync, int* s_tk) {
  __syncthreads();
  coop_draw(sync, s_tk);
  __syncthreads();
  return *s_tk;
}
// a workgroup leaves after it has drawn ONE ticket beyond the last item: when all gridDim.x have left, every draw of this
// launch has happened and the last one out restores the dispenser
__device__ __forceinline__ void coop_leave(int* sync) {
  if (threadIdx.x == 0) {
    const int e = __hip_atomic_fetch_add(sync + 2, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (e == (int)gridDim.x - 1) {
      __hip_atomic_store(sync + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(sync + 2, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

template <int T>
__device__ __forceinline__ double block_sum_t(double v, double* sm) {      // total in thread 0 (others: partial garbage)
  v = wave_sum_d(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sm[w] = v;
  __syncthreads();
  double r = 0.0;
  if (threadIdx.x == 0) {
#pragma unroll
    for (int i = 0; i < T / 64; ++i) r += sm[i];
  }
  return r;
}

template <int T>
__device__ __forceinline__ void block_sum2_t(double& a, double& b, double (*sm)[16]) {   // totals in thread 0; one barrier pair
  a = wave_sum_d(a);
  b = wave_sum_d(b);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) { sm[0][w] = a; sm[1][w] = b; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double ra = 0.0, rb = 0.0;
#pragma unroll
    for (int i = 0; i < T / 64; ++i) { ra += sm[0][i]; rb += sm[1][i]; }
    a = ra; b = rb;
  }
}

template <int V, int T, bool PIPE = true>
__global__ __launch_bounds__(T) __attribute__((amdgpu_waves_per_eu((T == 256 && !PIPE) ? 3 : 2, (T == 256 && !PIPE) ? 3 : 4))) void bn_coop_fwd_kernel(
    const float* __restrict__ x, int x_ctot, int x_coff, int N, int C, int C1, BnSet s1, BnSet s2, float eps, float momentum,
    float* __restrict__ mean_o, float* __restrict__ invstd_o, float* __restrict__ scale_o, const float* residual, int r_ctot,
    int r_coff, const float* __restrict__ r_mean, const float* __restrict__ r_scale, const float* __restrict__ r_shift,
    float* y, int y_ctot, int y_coff, float* __restrict__ gap_out, int gap_ctot, int gap_coff, int post_relu,
    double* part, int* sync, int P, int loop = 1, float* __restrict__ amax_out = nullptr) {
  // PIPE: persistent workgroups, the next item's loads in flight under the exchange.  !PIPE: one item at a time -- loop = 0:
  // one item per workgroup (grid = items), loop = 1: persistent, the next ticket drawn when the item is done
  constexpr int CH = 4 * V * T;                       // floats per workgroup: 1 / P of a plane
  const int HW = CH * P, NP = N * P;
  __shared__ double sm[2][16];
  __shared__ double bc[512];
  __shared__ int s_tk;
  const int items = C * NP;
  auto issue = [&](int it, float4 (&v)[V]) {
    const int c = it / NP, np = it - c * NP, n = np / P, po = (np - n * P) * CH;
    const float* xp = x + ((size_t)n * x_ctot + x_coff + c) * HW + po;
#pragma unroll
    for (int j = 0; j < V; ++j) v[j] = *reinterpret_cast<const float4*>(xp + 4 * (threadIdx.x + T * j));
  };
  auto sums = [&](const float4 (&v)[V], double& a, double& b) {
    a = 0.0; b = 0.0;
#pragma unroll
    for (int j = 0; j < V; ++j) {
      const float f0 = (v[j].x + v[j].y) + (v[j].z + v[j].w);
      const float f1 = (v[j].x * v[j].x + v[j].y * v[j].y) + (v[j].z * v[j].z + v[j].w * v[j].w);
      a += f0; b += f1;
    }
    block_sum2_t<T>(a, b, sm);
  };
  int it = 0, tk_next = 0;
  bool prepub = false, early = false;
  float amax = 0.f;                                   // largest |y| this thread wrote (amax_out)
  float4 v[V], vn[V];
  if constexpr (PIPE) {
    it = coop_draw_sync(sync, &s_tk);
    if (it < items) issue(it, v);
  }
  for (;;) {
    if constexpr (PIPE) {
      if (it >= items) break;
      coop_draw(sync, &s_tk);                         // (read behind the barriers of the reduction)
    } else {
      // (loop mode: the next ticket was drawn while the previous item was applied and stored -- behind its exchange, where
      //  holding it cannot make anybody wait -- and published to the workgroup by the barrier that ended that item)
      it = early ? s_tk : coop_draw_sync(sync, &s_tk);
      if (it >= items) break;
      issue(it, v);
    }
    const int c = it / NP, np = it - c * NP, n = np / P, po = (np - n * P) * CH;
    const BnSet& ps = c < C1 ? s1 : s2;
    const int cl = c < C1 ? c : c - C1;
    double a, b;
    sums(v, a, b);
    if (threadIdx.x == 0 && !prepub) coop_publish(a, b, part, c, np, NP);
    // the residual of this item and the next item's loads travel under the exchange
    const float* rp = residual ? residual + ((size_t)n * r_ctot + r_coff + c) * HW + po : nullptr;
    float4 rv[V];
    if (rp) {
#pragma unroll
      for (int j = 0; j < V; ++j) rv[j] = *reinterpret_cast<const float4*>(rp + 4 * (threadIdx.x + T * j));
    }
    const int nxt = PIPE ? s_tk : items;              // (!PIPE: one item per workgroup, the grid covers the items)
    const bool have = PIPE && nxt < items;
    if (have) issue(nxt, vn);
    const bool same = have && nxt / NP == c;
    if (same) {                                       // (uniform) this workgroup holds two items of the channel: both published before it waits
      double a2, b2;
      sums(vn, a2, b2);
      if (threadIdx.x == 0) coop_publish(a2, b2, part, c, nxt - c * NP, NP);
    }
    coop_gather(a, b, part, sync, c, NP, bc);
    if (!PIPE && loop && threadIdx.x == 0) tk_next = __hip_atomic_fetch_add(sync + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const double count = (double)N * HW;
    const double m = a / count;
    double var = b / count - m * m;
    if (var < 0.0) var = 0.0;
    const float is = (float)(1.0 / sqrt(var + (double)eps));
    const float mu = (float)m, sc = (ps.gamma ? ps.gamma[cl] : 1.f) * is, be = ps.beta ? ps.beta[cl] : 0.f;
    if (threadIdx.x == 0 && np == 0) {
      mean_o[c] = mu; invstd_o[c] = is; scale_o[c] = sc;
      if (ps.running_mean) ps.running_mean[cl] = (1.f - momentum) * ps.running_mean[cl] + momentum * mu;
      if (ps.running_var) {
        const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
        ps.running_var[cl] = (1.f - momentum) * ps.running_var[cl] + momentum * (float)unb;
      }
    }
    const bool raff = rp && r_scale;
    const float rmu = raff ? r_mean[r_coff + c] : 0.f, rsc = raff ? r_scale[r_coff + c] : 1.f, rsh = raff ? r_shift[r_coff + c] : 0.f;
    float* yp = y + ((size_t)n * y_ctot + y_coff + c) * HW + po;
    double gs = 0.0;
#pragma unroll
    for (int j = 0; j < V; ++j) {
      float e[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
      float re[4] = {0.f, 0.f, 0.f, 0.f};
      if (rp) { re[0] = rv[j].x; re[1] = rv[j].y; re[2] = rv[j].z; re[3] = rv[j].w; }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float o = (e[k] - mu) * sc + be;
        if (post_relu) o = fmaxf(o, 0.f);
        if (raff) re[k] = fmaxf((re[k] - rmu) * rsc + rsh, 0.f);
        e[k] = o + re[k];
      }
      st4<16>(yp + 4 * (threadIdx.x + T * j), make_float4(e[0], e[1], e[2], e[3]));
      gs += (double)((e[0] + e[1]) + (e[2] + e[3]));
      amax = amax4(amax, e[0], e[1], e[2], e[3]);
    }
    if (gap_out) {                                   // (uniform: the barriers inside are taken by everyone)
      gs = block_sum_t<T>(gs, sm[0]);
      if (P == 1) {
        if (threadIdx.x == 0) gap_out[(size_t)n * gap_ctot + gap_coff + c] = (float)(gs / (double)HW);
      } else if (threadIdx.x < 64) {
        // the P parts of a plane: each publishes its sum in its own slot (third slot region); the LAST part of the image --
        // the highest ticket: everything it waits for was drawn before it -- gathers them, lanes side by side, added in
        // part order, and writes the plane average
        unsigned long long* gsl = reinterpret_cast<unsigned long long*>(part) + (size_t)C * COOP_MAX_NP * 2 + (size_t)c * NP + n * P;
        if (threadIdx.x == 0)
          __hip_atomic_store(gsl + (np - n * P), __builtin_bit_cast(unsigned long long, gs), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (np == n * P + P - 1) {
          double pv = 0.0;
          if ((int)threadIdx.x < P) pv = __builtin_bit_cast(double, coop_poll(gsl + threadIdx.x, sync, 2));
          double t = 0.0;
          for (int k = 0; k < P; ++k) t += __shfl(pv, k);
          if (threadIdx.x == 0) gap_out[(size_t)n * gap_ctot + gap_coff + c] = (float)(t / (double)HW);
        }
      }
    }
    coop_depart(part, sync, c, NP, (gap_out && P > 1) ? C : 0, C, !PIPE && !loop);
    if (!PIPE && loop) { if (threadIdx.x == 0) s_tk = tk_next; early = true; }
    __syncthreads();                                 // bc / sm / s_tk are reused by the next item
    if (have) {
#pragma unroll
      for (int j = 0; j < V; ++j) v[j] = vn[j];
    }
    it = nxt;
    prepub = same;
    if (!PIPE && !loop) break;                       // one item per workgroup
  }
  if (PIPE || loop) coop_leave(sync);
  if (amax_out) {
    __shared__ unsigned s_amax;
    block_amax_commit(amax, amax_out, &s_amax);
  }
}

// POOL (1 / 2 = the pool's row stride): the gradient of the BatchNorm output is not stored -- the block ends in an SELayer + 3x3 /
// stride (SH, 2) / padding 1 max-pool (pointseg_net.py:27-46) and dy = xs[n, c] * route(dyp, idx) + xadd[n, c] (+ dy when given: the
// part of the gradient that IS stored) is formed while loading: pool3_strip8 routes the POOLED gradient through the arg-max map
// for a strip of 4 rows x 8 columns, so a thread holds such a strip (T / (W / 8) strips of W / 8 threads per part of
// a plane) instead of 8 float4 at a T float4 stride.  Saves writing the full-resolution gradient and reading it back (twice: the
// bypass residual of the squeeze data gradient goes the same way into the previous block's launch).
struct CoopPool { const float* dyp; const uint8_t* idx; const float* xs; const float* xadd; int W, OH, OW; };

// routed gradient of a strip of 4 input rows x 8 input columns (columns 8 b .. 8 b + 7 = pooled columns 4 b .. 4 b + 3, + the
// kx = 0 tap of pooled column 4 b + 4, which the NEXT lane holds: one shuffle instead of two more loads; the threads of a row
// are W / 8 <= 64 consecutive lanes, so the last thread of a row -- no such column -- is also the only one whose neighbour sits
// elsewhere).  Per pooled row ONE 16-byte load + ONE 4-byte load (pool3_strip: 4 loads of 8 / 4 / 2 / 1 bytes per row for half
// the columns).  Forward: out(oh, ow) = max over (ky, kx) of in(oh SH - 1 + ky, 2 ow - 1 + kx), idx = 3 ky + kx: an even input
// column 2 m receives from (ow = m, kx = 1), an odd one 2 m + 1 from (m, kx = 2) and (m + 1, kx = 0).
// Loads and routing are separate so that the loads of the NEXT item can be issued ahead (software pipeline above).
template <int SH> struct Pool3Rows { static constexpr int NJ = SH == 1 ? 6 : 3; };   // pooled rows whose windows meet input rows r0 .. r0 + 3

template <int SH>
__device__ __forceinline__ void pool3_strip8_load(const float* __restrict__ dyp, const uint8_t* __restrict__ ip, int r0, int b,
                                                  int OH, int OW, float4 (&v)[Pool3Rows<SH>::NJ], unsigned (&k)[Pool3Rows<SH>::NJ]) {
  constexpr int NJ = Pool3Rows<SH>::NJ;
  const int oh0 = SH == 1 ? r0 - 1 : r0 / 2;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {                        // unconditional (row clamped)
    const int oh = min(max(oh0 + j, 0), OH - 1);
    const size_t ro = (size_t)oh * OW + 4 * b;
    v[j] = *reinterpret_cast<const float4*>(dyp + ro);
    k[j] = *reinterpret_cast<const unsigned*>(ip + ro);
  }
}

template <int SH>
__device__ __forceinline__ void pool3_strip8_route(const float4 (&v)[Pool3Rows<SH>::NJ], const unsigned (&k)[Pool3Rows<SH>::NJ],
                                                   int r0, bool row_end, int OH, float (&G)[4][8]) {
  constexpr int NJ = Pool3Rows<SH>::NJ;
  const int oh0 = SH == 1 ? r0 - 1 : r0 / 2;
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 8; ++c) G[r][c] = 0.f;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const bool rv = (unsigned)(oh0 + j) < (unsigned)OH;
    const float hv = __shfl_down(v[j].x, 1, 64);       // pooled column 4 b + 4
    const unsigned hk = __shfl_down(k[j], 1, 64) & 0xffu;
    const float val[5] = {rv ? v[j].x : 0.f, rv ? v[j].y : 0.f, rv ? v[j].z : 0.f, rv ? v[j].w : 0.f, (rv && !row_end) ? hv : 0.f};
    const unsigned kk[5] = {k[j] & 0xffu, (k[j] >> 8) & 0xffu, (k[j] >> 16) & 0xffu, k[j] >> 24, hk};
#pragma unroll
    for (int c = 0; c < 5; ++c) {
      const unsigned ky = kk[c] >= 6u ? 2u : (kk[c] >= 3u ? 1u : 0u), kx = kk[c] - 3u * ky;
#pragma unroll
      for (int kyv = 0; kyv < 3; ++kyv) {
        const int rr = SH == 1 ? j + kyv - 2 : 2 * j - 1 + kyv;      // input row inside the strip (compile time)
        if (rr < 0 || rr >= 4) continue;
        const float t = ky == (unsigned)kyv ? val[c] : 0.f;
        if (c < 4) {
          G[rr][2 * c] += kx == 1u ? t : 0.f;
          G[rr][2 * c + 1] += kx == 2u ? t : 0.f;
          if (c > 0) G[rr][2 * c - 1] += kx == 0u ? t : 0.f;
        } else {
          G[rr][7] += kx == 0u ? t : 0.f;
        }
      }
    }
  }
}

// what a thread loads for one item: the (stored) gradient, the raw BatchNorm input, the pooled rows
template <int V, int POOL> struct CoopBwdRaw {
  float4 g[V], x[V];
  float4 pv[Pool3Rows<POOL == 0 ? 1 : POOL>::NJ];
  unsigned pk[Pool3Rows<POOL == 0 ? 1 : POOL>::NJ];
};

template <int V, int T, int POOL = 0, bool PIPE = true>
__global__ __launch_bounds__(T) __attribute__((amdgpu_waves_per_eu((T == 256 && !PIPE) ? 3 : 2, (T == 256 && !PIPE) ? 3 : 4))) void bn_coop_bwd_kernel(
    const float* __restrict__ dy, int dy_ctot, int dy_coff, const float* __restrict__ x, int x_ctot, int x_coff, int N, int C,
    int C1, BnSet s1, BnSet s2, const float* __restrict__ mean, const float* __restrict__ invstd,
    const float* __restrict__ scale, float* __restrict__ dx1, float* __restrict__ dx2, int accumulate, int post_relu,
    double* part, int* sync, int P, float* amax_out, CoopPool pl = CoopPool{}, int loop = 1) {
  static_assert(!POOL || V == 8, "a thread holds one 4 x 8 pool strip");
  constexpr int CH = 4 * V * T;                       // floats per workgroup: 1 / P of a plane
  const int HW = CH * P, NP = N * P;
  __shared__ double sm[2][16];
  __shared__ double bc[512];
  __shared__ int s_tk;
  const int items = C * NP;
  float amax = 0.f;                                   // largest |dx| this thread wrote (amax_out: for the two-piece split kernels)
  // element j of this thread inside a part: float offset EO(j)
  int e0 = 4 * threadIdx.x, pb = 0, prow = 0, pw = 0;
  bool prow_end = false;
  if constexpr (POOL != 0) {                          // a strip of 4 rows x 8 columns: j = 2 row + column half
    const int W8 = pl.W >> 3;
    pb = threadIdx.x % W8; prow = (threadIdx.x / W8) * 4; pw = pl.W;
    prow_end = pb == W8 - 1;
    e0 = prow * pl.W + 8 * pb;
  }
  auto EO = [&](int j) { return POOL != 0 ? e0 + (j >> 1) * pw + 4 * (j & 1) : e0 + 4 * T * j; };
  using Raw = CoopBwdRaw<V, POOL>;
  auto issue = [&](int it, Raw& r) {
    const int c = it / NP, np = it - c * NP, n = np / P, po = (np - n * P) * CH;
    const float* xp = x + ((size_t)n * x_ctot + x_coff + c) * HW + po;
    if (dy) {
      const float* gp = dy + ((size_t)n * dy_ctot + dy_coff + c) * HW + po;
#pragma unroll
      for (int j = 0; j < V; ++j) r.g[j] = *reinterpret_cast<const float4*>(gp + EO(j));
    }
#pragma unroll
    for (int j = 0; j < V; ++j) r.x[j] = *reinterpret_cast<const float4*>(xp + EO(j));
    if constexpr (POOL != 0) {
      const size_t plane = ((size_t)n * C + c) * pl.OH * pl.OW;       // (the pooled tensors hold exactly these C channels)
      pool3_strip8_load<POOL>(pl.dyp + plane, pl.idx + plane, po / pl.W + prow, pb, pl.OH, pl.OW, r.pv, r.pk);
    }
  };
  // raw -> (masked gradient, normalised input) + this thread's part of the two sums
  auto process = [&](int it, const Raw& r, float4 (&g)[V], float4 (&xh)[V], double& sg, double& sgx) {
    const int c = it / NP, np = it - c * NP, n = np / P, po = (np - n * P) * CH;
    const BnSet& ps = c < C1 ? s1 : s2;
    const int cl = c < C1 ? c : c - C1;
    const float mu = mean[c], is = invstd[c], sc = scale[c], be = ps.beta ? ps.beta[cl] : 0.f;
    if constexpr (POOL != 0) {
      float G[4][8];
      pool3_strip8_route<POOL>(r.pv, r.pk, po / pl.W + prow, prow_end, pl.OH, G);
      const size_t plane = (size_t)n * C + c;
      const float ps_ = pl.xs ? pl.xs[plane] : 1.f, pa = pl.xadd ? pl.xadd[plane] : 0.f;
#pragma unroll
      for (int j = 0; j < V; ++j) {
        const int rr = j >> 1, cc = 4 * (j & 1);
        g[j] = make_float4(G[rr][cc] * ps_ + pa, G[rr][cc + 1] * ps_ + pa, G[rr][cc + 2] * ps_ + pa, G[rr][cc + 3] * ps_ + pa);
        if (dy) { g[j].x += r.g[j].x; g[j].y += r.g[j].y; g[j].z += r.g[j].z; g[j].w += r.g[j].w; }
      }
    } else {
#pragma unroll
      for (int j = 0; j < V; ++j) g[j] = r.g[j];
    }
    sg = 0.0; sgx = 0.0;
#pragma unroll
    for (int j = 0; j < V; ++j) {
      float ge[4] = {g[j].x, g[j].y, g[j].z, g[j].w};
      float xe[4] = {r.x[j].x, r.x[j].y, r.x[j].z, r.x[j].w};
      float f0 = 0.f, f1 = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (post_relu && !((xe[k] - mu) * sc + be > 0.f)) ge[k] = 0.f;
        xe[k] = (xe[k] - mu) * is;
        f0 += ge[k]; f1 += ge[k] * xe[k];
      }
      g[j] = make_float4(ge[0], ge[1], ge[2], ge[3]);
      xh[j] = make_float4(xe[0], xe[1], xe[2], xe[3]);
      sg += f0; sgx += f1;
    }
  };
  int it = 0, tk_next = 0;
  bool prepub = false, early = false;
  Raw raw;
  if constexpr (PIPE) {
    it = coop_draw_sync(sync, &s_tk);
    if (it < items) issue(it, raw);
  }
  for (;;) {
    if constexpr (PIPE) {
      if (it >= items) break;
      coop_draw(sync, &s_tk);                         // (read behind the barriers of the reduction)
    } else {
      it = early ? s_tk : coop_draw_sync(sync, &s_tk);      // (see bn_coop_fwd_kernel)
      if (it >= items) break;
      issue(it, raw);
    }
    const int c = it / NP, np = it - c * NP, n = np / P, po = (np - n * P) * CH;
    const BnSet& ps = c < C1 ? s1 : s2;
    const int cl = c < C1 ? c : c - C1;
    float4 g[V], xh[V];
    double sg, sgx;
    process(it, raw, g, xh, sg, sgx);
    block_sum2_t<T>(sg, sgx, sm);
    if (threadIdx.x == 0 && !prepub) coop_publish(sg, sgx, part, c, np, NP);
    const int nxt = PIPE ? s_tk : items;              // (!PIPE: one item per workgroup, the grid covers the items)
    const bool have = PIPE && nxt < items;
    if (have) issue(nxt, raw);                        // the next item's loads travel under the exchange and the stores
    const bool same = have && nxt / NP == c;
    if (same) {                                       // (uniform) two items of one channel in this workgroup: both published before it waits
      float4 g2[V], xh2[V];
      double s2a, s2b;
      process(nxt, raw, g2, xh2, s2a, s2b);
      block_sum2_t<T>(s2a, s2b, sm);
      if (threadIdx.x == 0) coop_publish(s2a, s2b, part, c, nxt - c * NP, NP);
    }
    coop_gather(sg, sgx, part, sync, c, NP, bc);
    if (!PIPE && loop && threadIdx.x == 0) tk_next = __hip_atomic_fetch_add(sync + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (threadIdx.x == 0 && np == 0) {
      if (ps.dbeta) ps.dbeta[cl] = accumulate ? ps.dbeta[cl] + (float)sg : (float)sg;
      if (ps.dgamma) ps.dgamma[cl] = accumulate ? ps.dgamma[cl] + (float)sgx : (float)sgx;
    }
    const float sc = scale[c];
    const double inv_cnt = 1.0 / ((double)N * HW);
    const float mg = (float)(sg * inv_cnt), mgx = (float)(sgx * inv_cnt);
    float* op = (c < C1 ? dx1 + ((size_t)n * C1 + c) * HW : dx2 + ((size_t)n * (C - C1) + cl) * HW) + po;
#pragma unroll
    for (int j = 0; j < V; ++j) {
      float4 o;
      o.x = sc * (g[j].x - mg - xh[j].x * mgx); o.y = sc * (g[j].y - mg - xh[j].y * mgx);
      o.z = sc * (g[j].z - mg - xh[j].z * mgx); o.w = sc * (g[j].w - mg - xh[j].w * mgx);
      st4<8>(op + EO(j), o);
      amax = fmaxf(amax, fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fmaxf(fabsf(o.z), fabsf(o.w))));
    }
    coop_depart(part, sync, c, NP, 0, C, !PIPE && !loop);
    if (!PIPE && loop) { if (threadIdx.x == 0) s_tk = tk_next; early = true; }
    __syncthreads();                                  // bc / sm / s_tk are reused by the next item
    it = nxt;
    prepub = same;
    if (!PIPE && !loop) break;                        // one item per workgroup
  }
  if (PIPE || loop) coop_leave(sync);
  if (amax_out) {                                     // at most one atomic per workgroup (same-address atomics serialise: ~13 ns each)
    __shared__ unsigned wg_amax;
    if (threadIdx.x == 0) wg_amax = 0u;
    __syncthreads();
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(&wg_amax, __float_as_uint(amax));
    __syncthreads();
    // ... and only when it would raise the value (a relaxed read first: thousands of one-item workgroups, a handful of raises)
    if (threadIdx.x == 0 && wg_amax > __hip_atomic_load(reinterpret_cast<unsigned*>(amax_out), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
      atomicMax(reinterpret_cast<unsigned*>(amax_out), wg_amax);
  }
}

// ---- the same over bf16 storage (BASELINE configs[4], deeplio_amd/mixed.py; arithmetic as mixed_bf16.hip's two-launch
// kernels: fp32 per element, fp64 sums, one rounding to bf16 on the way out, plane averages of the STORED values):
// a thread holds 4 x 8 bf16 = the same 32 elements per operand
typedef __bf16 cbf16x8 __attribute__((ext_vector_type(8)));

template <int T>
__global__ __launch_bounds__(T) void bn16_coop_fwd_kernel(
    const __bf16* __restrict__ x, int x_ctot, int x_coff, int N, int C, BnSet ps, float eps, float momentum,
    float* __restrict__ mean_o, float* __restrict__ invstd_o, float* __restrict__ scale_o, const __bf16* residual, int r_ctot,
    int r_coff, __bf16* y, int y_ctot, int y_coff, float* __restrict__ gap_out, int gap_ctot, int gap_coff, int post_relu,
    double* part, int* sync, int P) {
  constexpr int CH = 32 * T;                          // elements per workgroup: 1 / P of a plane
  const int HW = CH * P, NP = N * P;
  __shared__ double sm[2][16];
  __shared__ double bc[512];
  __shared__ int s_tk;
  const int items = C * NP;
  for (int it = coop_draw_sync(sync, &s_tk); it < items; it = coop_draw_sync(sync, &s_tk)) {
    const int c = it / NP, np = it - c * NP, n = np / P, po = (np - n * P) * CH;
    const __bf16* xp = x + ((size_t)n * x_ctot + x_coff + c) * HW + po;
    cbf16x8 v[4];
    double a = 0.0, b = 0.0;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = *reinterpret_cast<const cbf16x8*>(xp + 8 * (threadIdx.x + T * j));
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float f0 = 0.f, f1 = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) { const float e = (float)v[j][k]; f0 += e; f1 += e * e; }
      a += f0; b += f1;
    }
    block_sum2_t<T>(a, b, sm);
    coop_exchange(a, b, part, sync, c, np, NP, bc);
    const double count = (double)N * HW;
    const double m = a / count;
    double var = b / count - m * m;
    if (var < 0.0) var = 0.0;
    const float is = (float)(1.0 / sqrt(var + (double)eps));
    const float mu = (float)m, sc = (ps.gamma ? ps.gamma[c] : 1.f) * is, be = ps.beta ? ps.beta[c] : 0.f;
    if (threadIdx.x == 0 && np == 0) {
      mean_o[c] = mu; invstd_o[c] = is; scale_o[c] = sc;
      if (ps.running_mean) ps.running_mean[c] = (1.f - momentum) * ps.running_mean[c] + momentum * mu;
      if (ps.running_var) {
        const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
        ps.running_var[c] = (1.f - momentum) * ps.running_var[c] + momentum * (float)unb;
      }
    }
    const __bf16* rp = residual ? residual + ((size_t)n * r_ctot + r_coff + c) * HW + po : nullptr;
    __bf16* yp = y + ((size_t)n * y_ctot + y_coff + c) * HW + po;
    double gs = 0.0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      cbf16x8 rv = {0, 0, 0, 0, 0, 0, 0, 0};
      if (rp) rv = *reinterpret_cast<const cbf16x8*>(rp + 8 * (threadIdx.x + T * j));
      cbf16x8 o;
      float fs = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        float t = ((float)v[j][k] - mu) * sc + be;
        if (post_relu) t = fmaxf(t, 0.f);
        if (rp) t += (float)rv[k];
        o[k] = (__bf16)t;
        fs += (float)o[k];
      }
      *reinterpret_cast<cbf16x8*>(yp + 8 * (threadIdx.x + T * j)) = o;
      gs += fs;
    }
    if (gap_out) {
      gs = block_sum_t<T>(gs, sm[0]);
      if (threadIdx.x == 0) gap_out[(size_t)n * gap_ctot + gap_coff + c] = (float)(gs / (double)HW);
    }
    coop_depart(part, sync, c, NP);
    __syncthreads();
  }
  coop_leave(sync);
}

template <int T>
__global__ __launch_bounds__(T) void bn16_coop_bwd_kernel(
    const __bf16* __restrict__ dy, int dy_ctot, int dy_coff, const __bf16* __restrict__ x, int x_ctot, int x_coff, int N, int C,
    BnSet ps, const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ scale,
    __bf16* __restrict__ dx, int dx_ctot, int dx_coff, int accumulate, int post_relu, double* part, int* sync, int P) {
  constexpr int CH = 32 * T;
  const int HW = CH * P, NP = N * P;
  __shared__ double sm[2][16];
  __shared__ double bc[512];
  __shared__ int s_tk;
  const int items = C * NP;
  for (int it = coop_draw_sync(sync, &s_tk); it < items; it = coop_draw_sync(sync, &s_tk)) {
    const int c = it / NP, np = it - c * NP, n = np / P, po = (np - n * P) * CH;
    const float mu = mean[c], is = invstd[c], sc = scale[c], be = ps.beta ? ps.beta[c] : 0.f;
    const __bf16* gp = dy + ((size_t)n * dy_ctot + dy_coff + c) * HW + po;
    const __bf16* xp = x + ((size_t)n * x_ctot + x_coff + c) * HW + po;
    cbf16x8 gv[4], xv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      gv[j] = *reinterpret_cast<const cbf16x8*>(gp + 8 * (threadIdx.x + T * j));
      xv[j] = *reinterpret_cast<const cbf16x8*>(xp + 8 * (threadIdx.x + T * j));
    }
    float g[4][8];
    double sg = 0.0, sgx = 0.0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float f0 = 0.f, f1 = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float xe = (float)xv[j][k];
        float ge = (float)gv[j][k];
        if (post_relu && !((xe - mu) * sc + be > 0.f)) ge = 0.f;
        g[j][k] = ge;
        f0 += ge; f1 += ge * ((xe - mu) * is);
      }
      sg += f0; sgx += f1;
    }
    block_sum2_t<T>(sg, sgx, sm);
    coop_exchange(sg, sgx, part, sync, c, np, NP, bc);
    if (threadIdx.x == 0 && np == 0) {
      if (ps.dbeta) ps.dbeta[c] = accumulate ? ps.dbeta[c] + (float)sg : (float)sg;
      if (ps.dgamma) ps.dgamma[c] = accumulate ? ps.dgamma[c] + (float)sgx : (float)sgx;
    }
    const double inv_cnt = 1.0 / ((double)N * HW);
    const float mg = (float)(sg * inv_cnt), mgx = (float)(sgx * inv_cnt);
    __bf16* op = dx + ((size_t)n * dx_ctot + dx_coff + c) * HW + po;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      cbf16x8 o;
#pragma unroll
      for (int k = 0; k < 8; ++k) o[k] = (__bf16)(sc * (g[j][k] - mg - ((float)xv[j][k] - mu) * is * mgx));
      *reinterpret_cast<cbf16x8*>(op + 8 * (threadIdx.x + T * j)) = o;
    }
    coop_depart(part, sync, c, NP);
    __syncthreads();
  }
  coop_leave(sync);
}

// planes the cooperative kernels take: P parts of T x 8 float4 each
int coop_t(int N, int HW, int& P, bool whole_plane = false) {
  static const int tt = getenv("DLIO_BN_COOP_T") ? atoi(getenv("DLIO_BN_COOP_T")) : 256;      // tuning knob: 256 / 512 / 1024
  P = 0;
  if (N < 2 || N > 64 || HW < 8192 || HW > 65536) return 0;
  if (whole_plane) {                                   // plane averages wanted: the plane in ONE workgroup
    if (HW != 8192 && HW != 16384 && HW != 32768) return 0;
    P = 1;
    return HW / 32;
  }
  int T = tt > 512 ? 512 : tt;                         // (the fp32 kernels hold two register sets: 1024 threads would spill)
  while (T > 256 && HW % (32 * T)) T >>= 1;
  if (HW % (32 * T)) return 0;
  P = HW / (32 * T);
  return N * P <= COOP_MAX_NP ? T : 0;
}

// Grid of a cooperative launch: what the occupancy query says is resident at once on the CUs the launch may use (all of
// them by default; dlio_bn_coop_set_cus / DLIO_BN_COOP_CUS cap it).  The ticket dispenser makes any grid CORRECT as long as
// N * parts workgroups of the launch can be resident together; a grid beyond the chip's capacity would only queue workgroups
// that find no ticket left.
int g_coop_cus = 0;
// 1: one item per workgroup, the grid covers the items (no software pipeline: the hardware's dispatcher interleaves the
// launch with whatever else runs, like any ordinary kernel); 0: persistent workgroups with the next item's loads in flight
int g_coop_mode = -1;
int coop_mode() {
  static const int m = getenv("DLIO_BN_COOP_MODE") ? atoi(getenv("DLIO_BN_COOP_MODE")) : 3;
  return g_coop_mode >= 0 ? g_coop_mode : m;
}
int coop_oneshot() { return coop_mode() != 0; }      // (the one-item-at-a-time kernel variants)
int coop_occupancy(const void* kernel, int T) {
  static std::mutex mu;
  static std::map<std::pair<int, const void*>, int> cache;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = -1; }
  std::lock_guard<std::mutex> lock(mu);
  auto f = cache.find({dev, kernel});
  if (f != cache.end()) return f->second;
  int occ = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kernel, T, 0) != hipSuccess || occ < 1) { (void)hipGetLastError(); occ = 1; }
  if (occ > 2048 / T) occ = 2048 / T;
  cache[{dev, kernel}] = occ;
  return occ;
}
// 0: the chip cannot hold 2 NP workgroups of this kernel at once (two launches -- the two encoder streams -- must each find
// NP resident workgroups): the caller refuses the path
// Persistent (1) or one item per workgroup (0) for THIS launch of a one-item-at-a-time kernel.  Mode 3: one item per workgroup
// where that cannot deadlock, persistent otherwise.  In the one-item mode the partners of a channel are whichever workgroups
// the dispatcher starts next; a workgroup that has drawn its ticket holds its slot until the channel is complete.  The eight
// XCDs dispatch their shares of a grid independently, so the tickets still to be drawn may all belong to workgroups of ONE XCD,
// and they start only if that XCD has a free slot: it has S = occupancy x CUs / 8 of them, a launch waits with at most NP - 1
// workgroups, so up to three concurrent launches are safe when 3 (NP - 1) < S -- at N = 16 the layers of fire_blk2 / blk3
// (NP = 32 / 16, S = 96); the two encoders' fire_blk1 launches (NP = 64 each) are not: measured in the training step with
// mode 1, one step in ~150 stalls until the spin limit, both launches at once, and never with one launch at a time.
int coop_loop(const void* kernel, int NP, int T) {
  const int m = coop_mode();
  if (m != 3) return m == 1 ? 0 : 1;
  const int S = coop_occupancy(kernel, T) * (dlio_num_cus() / 8);
  return 3 * (NP - 1) < S ? 0 : 1;
}
int coop_grid(const void* kernel, int NP, int C, int T, bool oneshot = false) {
  static const int env = getenv("DLIO_BN_COOP_CUS") ? atoi(getenv("DLIO_BN_COOP_CUS")) : 0;
  const int occ = coop_occupancy(kernel, T), cus = dlio_num_cus();
  if ((int64_t)cus * occ < 2 * (int64_t)NP) return 0;
  if (oneshot && !coop_loop(kernel, NP, T)) return (int64_t)C * NP > 0x7fffffff ? 0 : C * NP;      // one item per workgroup
  if (g_coop_cus < 0) return -g_coop_cus;             // (test hook: an exact grid, also one too small to make progress)
  // default 160 of 256 CUs' worth: in the five-stream step a grid that fills the chip leaves the neighbours' kernels nothing
  // (sweep, mode 2: 48 -> 19.7, 64 -> 19.3, 96 -> 19.5, 128 -> 18.8, 256 -> 19.0 ms per step; alone the launch is fastest at 256)
  int use = env > 0 ? env : (g_coop_cus > 0 ? g_coop_cus : (cus * 5) / 8);
  if (use > cus) use = cus;
  int64_t g = (int64_t)use * occ;
  if (g < NP) g = NP;
  const int64_t items = (int64_t)C * NP;
  return (int)(g < items ? g : items);
}

int small_v(int N, int HW) {
  if (N < 1 || N > 16) return 0;
  return HW == 256 ? 1 : HW == 512 ? 2 : HW == 1024 ? 4 : HW == 2048 ? 8 : 0;
}

}  // namespace

extern "C" int dlio_bn_small_ok(int N, int HW) { return small_v(N, HW) != 0; }

extern "C" int dlio_bn_small_fwd(const float* x, int N, int x_ctot, int x_coff, int C, int C1, int HW, int post_relu,
                                 const float* gamma1, const float* beta1, float* running_mean1, float* running_var1,
                                 const float* gamma2, const float* beta2, float* running_mean2, float* running_var2,
                                 float eps, float momentum, float* mean, float* invstd, float* scale, float* shift_out,
                                 const float* residual, int r_ctot, int r_coff, const float* r_mean, const float* r_scale,
                                 const float* r_shift, float* y, int y_ctot, int y_coff, float* gap_out, int gap_ctot,
                                 int gap_coff, float* amax_out, dlio_stream_t stream) {
  if (!x || !mean || !invstd || !scale || C <= 0 || C1 < 0 || C1 > C || N <= 0 || HW <= 0) return DLIO_EINVAL;
  const int v = small_v(N, HW);
  if (!v) return DLIO_EUNSUP;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(residual)) & 15) return DLIO_EUNSUP;
  hipStream_t s = as_stream(stream);
  const BnSet s1{gamma1, beta1, running_mean1, running_var1, nullptr, nullptr};
  const BnSet s2{gamma2, beta2, running_mean2, running_var2, nullptr, nullptr};
  DlioProfScope prof(7, s, 0.0, 4.0 * N * (double)C * HW * (y ? (residual ? 3.0 : 2.0) : 1.0));
#define BNS(VV) hipLaunchKernelGGL(bn_small_fwd_kernel<VV>, dim3((unsigned)C), dim3(1024), 0, s, x, x_ctot, x_coff, N, C1, s1, s2, eps, \
                                   momentum, mean, invstd, scale, shift_out, residual, r_ctot, r_coff, r_mean, r_scale, r_shift, y,     \
                                   y_ctot, y_coff, gap_out, gap_ctot, gap_coff, post_relu, amax_out)
  if (v == 1) BNS(1); else if (v == 2) BNS(2); else if (v == 4) BNS(4); else BNS(8);
#undef BNS
  return dlio_check_launch();
}

extern "C" int dlio_bn_small_bwd(const float* dy, int dy_ctot, int dy_coff, const float* x, int x_ctot, int x_coff,
                                 const float* mean, const float* invstd, const float* scale, const float* beta1,
                                 const float* beta2, float* dx1, float* dx2, float* dgamma1, float* dbeta1, float* dgamma2,
                                 float* dbeta2, int accumulate, int N, int C, int C1, int HW, int post_relu,
                                 float* amax_out, dlio_stream_t stream) {
  if (!dy || !x || !mean || !invstd || !scale || C <= 0 || C1 < 0 || C1 > C || N <= 0 || HW <= 0) return DLIO_EINVAL;
  if ((C1 > 0 && !dx1) || (C1 < C && !dx2)) return DLIO_EINVAL;
  const int v = small_v(N, HW);
  if (!v) return DLIO_EUNSUP;
  if ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dx1) |
       reinterpret_cast<uintptr_t>(dx2)) & 15)
    return DLIO_EUNSUP;
  hipStream_t s = as_stream(stream);
  const BnSet s1{nullptr, beta1, nullptr, nullptr, dgamma1, dbeta1};
  const BnSet s2{nullptr, beta2, nullptr, nullptr, dgamma2, dbeta2};
  DlioProfScope prof(9, s, 0.0, 3.0 * 4.0 * N * (double)C * HW);
#define BNS(VV) hipLaunchKernelGGL(bn_small_bwd_kernel<VV>, dim3((unsigned)C), dim3(1024), 0, s, dy, dy_ctot, dy_coff, x, x_ctot, x_coff, \
                                   N, C, C1, s1, s2, mean, invstd, scale, dx1, dx2, accumulate, post_relu, amax_out)
  if (v == 1) BNS(1); else if (v == 2) BNS(2); else if (v == 4) BNS(4); else BNS(8);
#undef BNS
  return dlio_check_launch();
}

extern "C" int dlio_bn_coop_set_mode(int oneshot) {
  if (oneshot < -1 || oneshot > 3) return DLIO_EINVAL;
  g_coop_mode = oneshot;            // -1: the default (DLIO_BN_COOP_MODE, else 3)
  return DLIO_OK;
}

extern "C" int dlio_bn_coop_get_mode(void) { return coop_mode(); }
// 1: a cooperative launch over N images of H * W = HW runs one item per workgroup under the mode in force (it then takes part in
// the "at most three in flight" rule of mode 3 / must not overlap another such launch in mode 1), 0: persistent / no such launch
extern "C" int dlio_bn_coop_one_item(int N, int HW) {
  int P = 0;
  const int T = coop_t(N, HW, P);
  if (!T || !coop_oneshot()) return 0;
  const void* k = T == 512 ? reinterpret_cast<const void*>(&bn_coop_bwd_kernel<8, 512, 0, false>)
                           : reinterpret_cast<const void*>(&bn_coop_bwd_kernel<8, 256, 0, false>);
  return coop_loop(k, N * P, T) ? 0 : 1;
}

extern "C" int dlio_bn_coop_set_cus(int cus) {
  g_coop_cus = cus;                 // 0 = all CUs; < 0: exactly -cus workgroups (tests of the ticket protocol's corner cases)
  return DLIO_OK;
}

extern "C" int dlio_bn_coop_ok(int N, int HW) { int P; return coop_t(N, HW, P) != 0; }
extern "C" int dlio_bn_coop_parts(int N, int HW) { int P; coop_t(N, HW, P); return P; }
extern "C" int dlio_bn_coop_gap_ok(int N, int HW) { int P; return coop_t(N, HW, P, true) != 0; }

extern "C" size_t dlio_bn_coop_ws_bytes(int N, int C) {
  if (N <= 0 || C <= 0) return 0;
  return (size_t)C * COOP_MAX_NP * 3 * sizeof(double);   // per channel: N * parts <= 256 slot pairs + as many plane-sum slots
}

extern "C" unsigned long long dlio_bn_coop_empty(void) { return COOP_EMPTY; }

extern "C" int dlio_bn_coop_fwd(const float* x, int N, int x_ctot, int x_coff, int C, int C1, int HW, int post_relu,
                                const float* gamma1, const float* beta1, float* running_mean1, float* running_var1,
                                const float* gamma2, const float* beta2, float* running_mean2, float* running_var2,
                                float eps, float momentum, float* mean, float* invstd, float* scale,
                                const float* residual, int r_ctot, int r_coff, const float* r_mean, const float* r_scale,
                                const float* r_shift, float* y, int y_ctot, int y_coff, float* gap_out, int gap_ctot,
                                int gap_coff, void* part, void* sync, float* amax_out, dlio_stream_t stream) {
  if (!x || !y || !mean || !invstd || !scale || !part || !sync || C <= 0 || C1 < 0 || C1 > C || N <= 0 || HW <= 0)
    return DLIO_EINVAL;
  int P;
  const int T = coop_t(N, HW, P);
  if (!T) return DLIO_EUNSUP;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(residual)) & 15) return DLIO_EUNSUP;
  hipStream_t s = as_stream(stream);
  const BnSet s1{gamma1, beta1, running_mean1, running_var1, nullptr, nullptr};
  const BnSet s2{gamma2, beta2, running_mean2, running_var2, nullptr, nullptr};
  DlioProfScope prof(7, s, 0.0, 4.0 * N * (double)C * HW * (residual ? 3.0 : 2.0));
  int grid = 0;
#define BNC(TT, PP) do { grid = coop_grid(reinterpret_cast<const void*>(&bn_coop_fwd_kernel<8, TT, PP>), N * P, C, TT, !PP); if (grid > 0) hipLaunchKernelGGL((bn_coop_fwd_kernel<8, TT, PP>), dim3((unsigned)grid), dim3(TT), 0, s, x, x_ctot, x_coff, N, C, C1, s1, s2, \
                                   eps, momentum, mean, invstd, scale, residual, r_ctot, r_coff, r_mean, r_scale, r_shift, y, y_ctot,     \
                                   y_coff, gap_out, gap_ctot, gap_coff, post_relu, reinterpret_cast<double*>(part),                      \
                                   reinterpret_cast<int*>(sync), P, coop_loop(reinterpret_cast<const void*>(&bn_coop_fwd_kernel<8, TT, PP>), N * P, TT), amax_out); } while (0)
  if (coop_oneshot()) { if (T == 512) BNC(512, false); else BNC(256, false); }
  else { if (T == 512) BNC(512, true); else BNC(256, true); }
#undef BNC
  if (grid <= 0) return DLIO_EUNSUP;      // (the chip cannot hold the cooperating workgroups of two such launches)
  return dlio_check_launch();
}

extern "C" int dlio_bn_coop_bwd(const float* dy, int dy_ctot, int dy_coff, const float* x, int x_ctot, int x_coff,
                                const float* mean, const float* invstd, const float* scale, const float* beta1,
                                const float* beta2, float* dx1, float* dx2, float* dgamma1, float* dbeta1, float* dgamma2,
                                float* dbeta2, int accumulate, int N, int C, int C1, int HW, int post_relu, void* part,
                                void* sync, float* amax_out, dlio_stream_t stream) {
  if (!dy || !x || !mean || !invstd || !scale || !part || !sync || C <= 0 || C1 < 0 || C1 > C || N <= 0 || HW <= 0)
    return DLIO_EINVAL;
  if ((C1 > 0 && !dx1) || (C1 < C && !dx2)) return DLIO_EINVAL;
  int P;
  const int T = coop_t(N, HW, P);
  if (!T) return DLIO_EUNSUP;
  if ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dx1) |
       reinterpret_cast<uintptr_t>(dx2)) & 15)
    return DLIO_EUNSUP;
  hipStream_t s = as_stream(stream);
  const BnSet s1{nullptr, beta1, nullptr, nullptr, dgamma1, dbeta1};
  const BnSet s2{nullptr, beta2, nullptr, nullptr, dgamma2, dbeta2};
  DlioProfScope prof(9, s, 0.0, 3.0 * 4.0 * N * (double)C * HW);
  int grid = 0;
#define BNC(TT, PP) do { grid = coop_grid(reinterpret_cast<const void*>(&bn_coop_bwd_kernel<8, TT, 0, PP>), N * P, C, TT, !PP); if (grid > 0) hipLaunchKernelGGL((bn_coop_bwd_kernel<8, TT, 0, PP>), dim3((unsigned)grid), dim3(TT), 0, s, dy, dy_ctot, dy_coff, x, x_ctot,  \
                                   x_coff, N, C, C1, s1, s2, mean, invstd, scale, dx1, dx2, accumulate, post_relu,                    \
                                   reinterpret_cast<double*>(part), reinterpret_cast<int*>(sync), P, amax_out, CoopPool{},          \
                                   coop_loop(reinterpret_cast<const void*>(&bn_coop_bwd_kernel<8, TT, 0, PP>), N * P, TT)); } while (0)
  if (coop_oneshot()) { if (T == 512) BNC(512, false); else BNC(256, false); }
  else if (T == 512) BNC(512, true); else BNC(256, true);
#undef BNC
  if (grid <= 0) return DLIO_EUNSUP;      // (the chip cannot hold the cooperating workgroups of two such launches)
  return dlio_check_launch();
}

/* dlio_bn_coop_bwd with the gradient of the BatchNorm output routed out of a max-pool's POOLED gradient while it loads
 * (bn_coop_bwd_kernel<.., POOL>): dy nullable (the stored part of the gradient, added), dy_pooled / idx [N][C][OH][OW] (idx: the
 * tap 0-8 of dlio_maxpool2d_fwd), x_scale / x_add [N][C] nullable. */
extern "C" int dlio_bn_coop_pool_ok(int N, int H, int W, int SH) {
  int P;
  const int T = coop_t(N, H * W, P);
  if (!T || (SH != 1 && SH != 2) || W < 16 || (W & 15) || (H & 3)) return 0;
  const int W8 = W >> 3;
  // a part of a plane = T / W8 strips of 4 rows x W columns; the W8 threads of a row inside one wave (the halo shuffle)
  return T % W8 == 0 && W8 <= 64 && 64 % W8 == 0;
}

extern "C" int dlio_bn_coop_bwd_pool(const float* dy, int dy_ctot, int dy_coff, const float* dy_pooled, const unsigned char* idx,
                                     const float* x_scale, const float* x_add, int H, int W, int SH, const float* x, int x_ctot,
                                     int x_coff, const float* mean, const float* invstd, const float* scale, const float* beta1,
                                     const float* beta2, float* dx1, float* dx2, float* dgamma1, float* dbeta1, float* dgamma2,
                                     float* dbeta2, int accumulate, int N, int C, int C1, int post_relu, void* part, void* sync,
                                     float* amax_out, dlio_stream_t stream) {
  if (!dy_pooled || !idx || !x || !mean || !invstd || !scale || !part || !sync || C <= 0 || C1 < 0 || C1 > C || N <= 0 || H <= 0 ||
      W <= 0)
    return DLIO_EINVAL;
  if ((C1 > 0 && !dx1) || (C1 < C && !dx2)) return DLIO_EINVAL;
  if (!dlio_bn_coop_pool_ok(N, H, W, SH)) return DLIO_EUNSUP;
  const int HW = H * W;
  int P;
  const int T = coop_t(N, HW, P);
  if ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dx1) |
       reinterpret_cast<uintptr_t>(dx2)) & 15)
    return DLIO_EUNSUP;
  const int OH = (H + 2 - 3) / SH + 1, OW = W / 2;
  if ((reinterpret_cast<uintptr_t>(dy_pooled) & 15) || (reinterpret_cast<uintptr_t>(idx) & 3) || (OW & 3))
    return DLIO_EUNSUP;                                 // 16-byte / 4-byte reads of the pooled rows
  hipStream_t s = as_stream(stream);
  const BnSet s1{nullptr, beta1, nullptr, nullptr, dgamma1, dbeta1};
  const BnSet s2{nullptr, beta2, nullptr, nullptr, dgamma2, dbeta2};
  const CoopPool pl{dy_pooled, idx, x_scale, x_add, W, OH, OW};
  DlioProfScope prof(9, s, 0.0, 4.0 * N * (double)C * HW * (dy ? 3.0 : 2.0) + 5.0 * N * (double)C * OH * OW);
  int grid = 0;
#define BNC(TT, PL, PP) do { grid = coop_grid(reinterpret_cast<const void*>(&bn_coop_bwd_kernel<8, TT, PL, PP>), N * P, C, TT, !PP); if (grid > 0) hipLaunchKernelGGL((bn_coop_bwd_kernel<8, TT, PL, PP>), dim3((unsigned)grid), dim3(TT), 0, s, dy, dy_ctot, dy_coff, x,                                         x_ctot, x_coff, N, C, C1, s1, s2, mean, invstd, scale, dx1, dx2, accumulate, post_relu,                                               reinterpret_cast<double*>(part), reinterpret_cast<int*>(sync), P, amax_out, pl, coop_loop(reinterpret_cast<const void*>(&bn_coop_bwd_kernel<8, TT, PL, PP>), N * P, TT)); } while (0)
  if (coop_oneshot()) {
    if (SH == 1) { if (T == 512) BNC(512, 1, false); else BNC(256, 1, false); }
    else { if (T == 512) BNC(512, 2, false); else BNC(256, 2, false); }
  } else {
    if (SH == 1) { if (T == 512) BNC(512, 1, true); else BNC(256, 1, true); }
    else { if (T == 512) BNC(512, 2, true); else BNC(256, 2, true); }
  }
#undef BNC
  if (grid <= 0) return DLIO_EUNSUP;      // (the chip cannot hold the cooperating workgroups of two such launches)
  return dlio_check_launch();
}

/* bf16 storage (csrc/mixed_bf16.hip's dlio_bn_bf16_apply / dlio_bn_bf16_bwd, train mode, one layer per launch): same
 * geometry rule in ELEMENTS (H * W a multiple of 8192 up to 65536, 2 <= N <= 64, N * parts <= 256) */
extern "C" int dlio_bn_bf16_coop_fwd(const void* x, int N, int x_ctot, int x_coff, int C, int HW, int post_relu,
                                     const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                                     float* running_var, float* mean, float* invstd, float* scale, const void* residual,
                                     int r_ctot, int r_coff, void* y, int y_ctot, int y_coff, float* gap_out, int gap_ctot,
                                     int gap_coff, void* part, void* sync, dlio_stream_t stream) {
  if (!x || !y || !mean || !invstd || !scale || !part || !sync || C <= 0 || N <= 0 || HW <= 0) return DLIO_EINVAL;
  int P;
  const int T = coop_t(N, HW, P, gap_out != nullptr);
  if (!T) return DLIO_EUNSUP;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(residual)) & 15) return DLIO_EUNSUP;
  hipStream_t s = as_stream(stream);
  const BnSet ps{gamma, beta, running_mean, running_var, nullptr, nullptr};
  DlioProfScope prof(7, s, 0.0, 2.0 * N * (double)C * HW * (residual ? 3.0 : 2.0));
  int grid = 0;
#define BNC(TT) do { grid = coop_grid(reinterpret_cast<const void*>(&bn16_coop_fwd_kernel<TT>), N * P, C, TT); if (grid > 0) hipLaunchKernelGGL((bn16_coop_fwd_kernel<TT>), dim3((unsigned)grid), dim3(TT), 0, s, reinterpret_cast<const __bf16*>(x),  \
                                   x_ctot, x_coff, N, C, ps, eps, momentum, mean, invstd, scale,                                     \
                                   reinterpret_cast<const __bf16*>(residual), r_ctot, r_coff, reinterpret_cast<__bf16*>(y), y_ctot,  \
                                   y_coff, gap_out, gap_ctot, gap_coff, post_relu, reinterpret_cast<double*>(part),                 \
                                   reinterpret_cast<int*>(sync), P); } while (0)
  if (T == 1024) BNC(1024); else if (T == 512) BNC(512); else BNC(256);
#undef BNC
  if (grid <= 0) return DLIO_EUNSUP;      // (the chip cannot hold the cooperating workgroups of two such launches)
  return dlio_check_launch();
}

extern "C" int dlio_bn_bf16_coop_bwd(const void* dy, int dy_ctot, int dy_coff, const void* x, int x_ctot, int x_coff,
                                     const float* mean, const float* invstd, const float* scale, const float* beta, void* dx,
                                     int dx_ctot, int dx_coff, float* dgamma, float* dbeta, int accumulate, int N, int C,
                                     int HW, int post_relu, void* part, void* sync, dlio_stream_t stream) {
  if (!dy || !x || !mean || !invstd || !scale || !dx || !part || !sync || C <= 0 || N <= 0 || HW <= 0) return DLIO_EINVAL;
  int P;
  const int T = coop_t(N, HW, P);
  if (!T) return DLIO_EUNSUP;
  if ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dx)) & 15) return DLIO_EUNSUP;
  hipStream_t s = as_stream(stream);
  const BnSet ps{nullptr, beta, nullptr, nullptr, dgamma, dbeta};
  DlioProfScope prof(9, s, 0.0, 3.0 * 2.0 * N * (double)C * HW);
  int grid = 0;
#define BNC(TT) do { grid = coop_grid(reinterpret_cast<const void*>(&bn16_coop_bwd_kernel<TT>), N * P, C, TT); if (grid > 0) hipLaunchKernelGGL((bn16_coop_bwd_kernel<TT>), dim3((unsigned)grid), dim3(TT), 0, s, reinterpret_cast<const __bf16*>(dy), \
                                   dy_ctot, dy_coff, reinterpret_cast<const __bf16*>(x), x_ctot, x_coff, N, C, ps, mean, invstd,    \
                                   scale, reinterpret_cast<__bf16*>(dx), dx_ctot, dx_coff, accumulate, post_relu,                   \
                                   reinterpret_cast<double*>(part), reinterpret_cast<int*>(sync), P); } while (0)
  if (T == 1024) BNC(1024); else if (T == 512) BNC(512); else BNC(256);
#undef BNC
  if (grid <= 0) return DLIO_EUNSUP;      // (the chip cannot hold the cooperating workgroups of two such launches)
  return dlio_check_launch();
}
