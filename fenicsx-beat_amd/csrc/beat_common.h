// Shared host/device helpers of libbeat_hip (gfx950 only; wave = 64 lanes).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <string>

#include "../../include/beat_hip.h"

#define BEAT_WAVE 64
#define BEAT_BLOCK 256  // every kernel that reduces uses 256-thread (4-wave) workgroups

struct beat_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  double* d_partials = nullptr;  // scratch for block partial sums: BEAT_NRED * BEAT_MAX_PARTIALS
  double* h_pinned = nullptr;    // small pinned staging buffer (64 doubles)
  double* d_small = nullptr;     // small device staging buffer (64 doubles)
};
#define BEAT_MAX_PARTIALS 16384
#define BEAT_NRED 3

void beat_set_error(const char* fmt, ...);

#define BEAT_HIP_CHECK(expr)                                                              \
  do {                                                                                    \
    hipError_t _e = (expr);                                                               \
    if (_e != hipSuccess) {                                                               \
      beat_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__,     \
                     __LINE__);                                                           \
      return BEAT_EHIP;                                                                   \
    }                                                                                     \
  } while (0)

#define BEAT_REQUIRE(cond, ...)    \
  do {                             \
    if (!(cond)) {                 \
      beat_set_error(__VA_ARGS__); \
      return BEAT_EINVAL;          \
    }                              \
  } while (0)

// A launch is checked with hipGetLastError, which (HIP 7) returns the last error of ANY runtime call this host thread has
// made, however long ago -- another library's harmless probe included (seen: "invalid device ordinal" surfacing at the
// check behind a launch helper that had nothing to launch, in a thread that had torn down an RCCL communicator).
// BEAT_KERNEL drops whatever is pending before the launch, and BEAT_LAUNCH_CHECK only looks when a launch has been
// made since the last check, so that it reads the launches' own verdict and nothing else.
extern thread_local bool beat_tls_launched;
#define BEAT_KERNEL(...)                         \
  do {                                           \
    if (!beat_tls_launched) (void)hipGetLastError(); \
    beat_tls_launched = true;                    \
    hipLaunchKernelGGL(__VA_ARGS__);             \
  } while (0)
#define BEAT_LAUNCH_CHECK()               \
  do {                                    \
    if (beat_tls_launched) {              \
      beat_tls_launched = false;          \
      BEAT_HIP_CHECK(hipGetLastError());  \
    }                                     \
  } while (0)

// ---- wave / block reductions (fixed summation order => run-to-run deterministic) -------------
__device__ __forceinline__ double beat_wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;  // valid in lane 0
}
__device__ __forceinline__ double beat_wave_min(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmin(v, __shfl_down(v, off, 64));
  return v;
}
__device__ __forceinline__ double beat_wave_max(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_down(v, off, 64));
  return v;
}

// Sum over a 256-thread block; result valid in thread 0.  `smem` holds >= 4 doubles and may be
// reused immediately after return by thread 0 only (a trailing barrier is included).
__device__ __forceinline__ double beat_block_sum(double v, double* smem) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  v = beat_wave_sum(v);
  if (lane == 0) smem[wave] = v;
  __syncthreads();
  double s = 0.0;
  if (threadIdx.x == 0) s = (smem[0] + smem[1]) + (smem[2] + smem[3]);
  __syncthreads();
  return s;
}

namespace beat_pde_detail {
// Extrapolated initial guess (beat_pde_set_guess_order): the solve starts from x0 = v_ + e, where e was prepared by
// the previous solve's x update from the increments d = x - v_ of the last solves: polynomial extrapolation in time
// of degree m - 1 through the last m increments, e = sum_{i=1..m} (-1)^(i+1) C(m, i) d_i  (m = 1: d1; 2: 2 d1 - d2;
// 3: 3 d1 - 3 d2 + d3; 4: 4 d1 - 6 d2 + 4 d3 - d4).  e is never added to x by a pass of its own: it rides with the
// deferred update  x += inc, inc = e + sum alpha_j p_j,  which also records d <- inc (over the oldest increment kept)
// and prepares the next guess  e <- a inc + cd d_old + sum cp_j dp_j  in place.
constexpr int BEAT_GUESS_MAX_ORDER = 4;
struct GuessTerms {
  double* d = nullptr;   // in: the oldest increment kept, out: this solve's; nullptr: no guess in use
  const double* dp[BEAT_GUESS_MAX_ORDER - 2] = {nullptr, nullptr};  // the newer increments (d1, d2): read only
  double* e = nullptr;   // in: this solve's guess increment (if use_e), out: the next solve's
  double a = 1.0, cd = 0.0, cp[BEAT_GUESS_MAX_ORDER - 2] = {0.0, 0.0};
  int use_e = 0;
  // an x update of a later ring cycle of the same solve: e went to x with the first cycle, this one adds its
  // directions to x and to what the first cycle recorded (d += inc, e += a inc)
  int accumulate = 0;
};

// what an x update does to (d, e) once its increment is known -- one expression shared by the flush kernels and the
// ionic kernel's pending path, so that both leave the same bits behind.  d_old / dp_old / e_old: the values the
// fields held (read by the caller up front, together with its other loads; unused ones may be anything)
__device__ __forceinline__ bool beat_guess_needs_d(const GuessTerms& gt) { return gt.accumulate || gt.cd != 0.0; }
__device__ __forceinline__ bool beat_guess_needs_dp(const GuessTerms& gt, int j) { return !gt.accumulate && gt.cp[j] != 0.0; }
__device__ __forceinline__ bool beat_guess_needs_e(const GuessTerms& gt) { return gt.accumulate || gt.use_e; }
__device__ __forceinline__ void beat_guess_record(const GuessTerms& gt, double* d, double* e, double inc, double d_old,
                                                  double dp0_old, double dp1_old, double e_old) {
  if (gt.accumulate) {
    *d = d_old + inc;
    *e = fma(gt.a, inc, e_old);
  } else {
    double en = gt.a * inc;
    if (gt.cp[0] != 0.0) en = fma(gt.cp[0], dp0_old, en);
    if (gt.cp[1] != 0.0) en = fma(gt.cp[1], dp1_old, en);
    if (gt.cd != 0.0) en = fma(gt.cd, d_old, en);
    *d = inc;
    *e = en;
  }
}
}  // namespace beat_pde_detail
