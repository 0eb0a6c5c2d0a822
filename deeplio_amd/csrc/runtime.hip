// Library info, error strings, and the optional per-launch hipEvent profiler used by
// bench.py's roofline leg (events are recorded on the launch stream, so the measured
// interval is the kernel's own duration).
#include "common.h"
#include <mutex>
#include <set>
#include <utility>
#include <mutex>
#include <vector>

namespace {
struct ProfKind {
  std::vector<hipEvent_t> start, stop;
  size_t used = 0, seen = 0;
  double flops = 0.0, bytes = 0.0;
};
int g_prof_stride = 1;      // time every stride-th launch of a kind (sampling keeps the timed region unperturbed)
thread_local bool t_armed[DLIO_PROF_KINDS] = {};
ProfKind g_prof[DLIO_PROF_KINDS];
int g_prof_mask = 0;        // bit k: kind k is timed
std::mutex g_mu;
}  // namespace

void dlio_prof_begin(int kind, hipStream_t s, double flops, double bytes) {
  if (kind < 0 || kind >= DLIO_PROF_KINDS || !(g_prof_mask >> kind & 1)) return;
  std::lock_guard<std::mutex> lk(g_mu);
  ProfKind& k = g_prof[kind];
  t_armed[kind] = (k.seen++ % (size_t)g_prof_stride) == 0;
  if (!t_armed[kind]) return;
  if (k.used == k.start.size()) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    k.start.push_back(a);
    k.stop.push_back(b);
  }
  hipEventRecord(k.start[k.used], s);
  k.flops += flops;
  k.bytes += bytes;
}

void dlio_prof_end(int kind, hipStream_t s) {
  if (kind < 0 || kind >= DLIO_PROF_KINDS || !(g_prof_mask >> kind & 1)) return;
  if (!t_armed[kind]) return;
  t_armed[kind] = false;
  std::lock_guard<std::mutex> lk(g_mu);
  ProfKind& k = g_prof[kind];
  hipEventRecord(k.stop[k.used], s);
  k.used++;
}

extern "C" int dlio_prof_enable(int kinds_mask) {
  g_prof_mask = kinds_mask & ((1 << DLIO_PROF_KINDS) - 1);
  return DLIO_OK;
}

extern "C" int dlio_prof_reset(void) {
  std::lock_guard<std::mutex> lk(g_mu);
  for (auto& k : g_prof) { k.used = 0; k.seen = 0; k.flops = 0.0; k.bytes = 0.0; }
  return DLIO_OK;
}

extern "C" int dlio_prof_release(void) {
  std::lock_guard<std::mutex> lk(g_mu);
  for (auto& k : g_prof) {
    for (auto e : k.start) (void)hipEventDestroy(e);
    for (auto e : k.stop) (void)hipEventDestroy(e);
    k.start.clear(); k.stop.clear();
    k.used = 0; k.seen = 0; k.flops = 0.0; k.bytes = 0.0;
  }
  return DLIO_OK;
}

extern "C" int dlio_prof_sample(int stride) {
  if (stride < 1) return DLIO_EINVAL;
  g_prof_stride = stride;
  return DLIO_OK;
}

extern "C" int dlio_prof_collect(int kind, double* ms, double* flops, double* bytes,
                                 int64_t* launches) {
  if (kind < 0 || kind >= DLIO_PROF_KINDS || !ms || !flops || !bytes || !launches) return DLIO_EINVAL;
  std::lock_guard<std::mutex> lk(g_mu);
  ProfKind& k = g_prof[kind];
  double total = 0.0;
  for (size_t i = 0; i < k.used; ++i) {
    if (hipEventSynchronize(k.stop[i]) != hipSuccess) return DLIO_ELAUNCH;
    float t = 0.f;
    if (hipEventElapsedTime(&t, k.start[i], k.stop[i]) != hipSuccess) return DLIO_ELAUNCH;
    total += t;
  }
  *ms = total; *flops = k.flops; *bytes = k.bytes; *launches = (int64_t)k.used;
  return DLIO_OK;
}

thread_local int dlio_last_hip_error = 0;

extern "C" const char* dlio_last_hip_error_string(void) {
  return hipGetErrorString((hipError_t)dlio_last_hip_error);
}

extern "C" int dlio_version(void) { return DLIO_ABI_VERSION; }
#ifndef DLIO_HEADER_CRC
#define DLIO_HEADER_CRC 0u
#endif
extern "C" uint32_t dlio_abi_hash(void) { return (uint32_t)DLIO_HEADER_CRC; }
int dlio_num_cus() {
  static int cached = 0;
  if (cached > 0) return cached;
  int dev = 0, n = 0;
  if (hipGetDevice(&dev) == hipSuccess &&
      hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0)
    cached = n;
  else
    (void)hipGetLastError();
  return cached > 0 ? cached : 256;
}

void dlio_set_max_lds(const void* kernel, int bytes) {
  static std::mutex mu;
  static std::set<std::pair<int, const void*>> done;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = -1; }
  std::lock_guard<std::mutex> lock(mu);
  if (dev >= 0 && done.count({dev, kernel})) return;
  (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (dev >= 0) done.insert({dev, kernel});
}

// Timing probes (-DDLIO_SPLIT_Q0, -DBX3_ABLATE, -DW1_COAL_PROBE: tools/variant_lib.py, tools/bx3_ablate.py) build kernels that
// compute WRONG results on purpose; the product build sets none of them and the Python side refuses a library that has any.
int dlio_probe_bx3();
int dlio_probe_wgrad();
int dlio_probe_wgrad3();
int dlio_probe_fire();
extern "C" int dlio_build_probes(void) {
  return ((DLIO_SPLIT_Q0) != 0 ? 1 : 0) | dlio_probe_bx3() | dlio_probe_wgrad() | dlio_probe_wgrad3() | dlio_probe_fire();
}

// ---- which HIP streams share a hardware queue -----------------------------------------------------------------------
// The runtime serves all streams of a process from a small pool of hardware queues (4 by default); two streams on one queue run
// their launches strictly one after the other.  The training step keeps four heavy streams (two encoders, their
// weight-gradient companions) that must not share: +2 ms per step when any two of them do.  Which queue a stream gets
// depends on what the process created before, so the host side probes it: a short launch on b, issued behind a spinning one on
// a, finishes early exactly when the two streams have different queues.
__global__ void spin_kernel(long long ticks) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}

extern "C" int dlio_streams_share_queue(dlio_stream_t a, dlio_stream_t b, int* shared) {
  if (!shared) return DLIO_EINVAL;
  hipStream_t sa = as_stream(a), sb = as_stream(b);
  if (sa == sb) { *shared = 1; return DLIO_OK; }
  int dev = 0, khz = 0;
  if (hipGetDevice(&dev) != hipSuccess ||
      hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || khz <= 0) {
    (void)hipGetLastError();
    khz = 100000;
  }
  const long long long_ticks = (long long)khz * 150 / 1000, short_ticks = (long long)khz / 1000;     // 150 us, 1 us
  hipEvent_t e0, ea, eb;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&ea) != hipSuccess || hipEventCreate(&eb) != hipSuccess)
    return DLIO_ELAUNCH;
  int votes = 0, rc = DLIO_OK;
  for (int rep = 0; rep < 3 && rc == DLIO_OK; ++rep) {      // (rep 0 = warm-up: the first launch on a stream is slow)
    (void)hipStreamSynchronize(sa);
    (void)hipStreamSynchronize(sb);
    (void)hipEventRecord(e0, sa);
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(1), 0, sa, long_ticks);
    (void)hipEventRecord(ea, sa);
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(1), 0, sb, short_ticks);
    (void)hipEventRecord(eb, sb);
    rc = dlio_check_launch();
    if (hipStreamSynchronize(sa) != hipSuccess || hipStreamSynchronize(sb) != hipSuccess) rc = DLIO_ELAUNCH;
    float ta = 0.f, tb = 0.f;
    if (rc == DLIO_OK && (hipEventElapsedTime(&ta, e0, ea) != hipSuccess || hipEventElapsedTime(&tb, e0, eb) != hipSuccess))
      rc = DLIO_ELAUNCH;
    if (rep > 0 && tb > 0.5f * ta) ++votes;
  }
  (void)hipEventDestroy(e0); (void)hipEventDestroy(ea); (void)hipEventDestroy(eb);
  *shared = votes == 2;
  return rc;
}

extern "C" const char* dlio_arch(void) { return "gfx950"; }
extern "C" const char* dlio_strerror(int code) {
  switch (code) {
    case DLIO_OK: return "ok";
    case DLIO_EINVAL: return "invalid argument (shape / null pointer)";
    case DLIO_EUNSUP: return "unsupported configuration";
    case DLIO_ELAUNCH: return "HIP launch/runtime error";
    case DLIO_EWS: return "workspace too small";
    default: return "unknown error";
  }
}
