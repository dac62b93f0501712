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

#define BEAT_LAUNCH_CHECK() BEAT_HIP_CHECK(hipGetLastError())

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
// the previous solve's x update from the increments d = x - v_ of the last solves (e = d1, or 2 d1 - d2).  e is never
// added to x by a pass of its own: it rides with the deferred update  x += inc, inc = e + sum alpha_j p_j,  which
// also records d <- inc and prepares the next guess  e <- a inc + b d_old  in place.
struct GuessTerms {
  double* d = nullptr;  // most recent increment (in: d_old, out: this solve's); nullptr: no guess in use
  double* e = nullptr;  // in: this solve's guess increment (if use_e), out: the next solve's
  double a = 1.0, b = 0.0;
  int use_e = 0;
  // an x update of a later ring cycle of the same solve: e went to x with the first cycle, this one adds its
  // directions to x and to what the first cycle recorded (d += inc, e += a inc)
  int accumulate = 0;
};

// what an x update does to (d, e) once its increment is known -- one expression shared by the flush kernels and the
// ionic kernel's pending path, so that both leave the same bits behind.  d_old / e_old: the values the two fields
// held (read by the caller up front, together with its other loads; unused ones may be anything)
__device__ __forceinline__ bool beat_guess_needs_d(const GuessTerms& gt) { return gt.accumulate || gt.b != 0.0; }
__device__ __forceinline__ void beat_guess_record(const GuessTerms& gt, double* d, double* e, double inc, double d_old,
                                                  double e_old) {
  if (gt.accumulate) {
    *d = d_old + inc;
    *e = fma(gt.a, inc, e_old);
  } else {
    *d = inc;
    *e = gt.b != 0.0 ? fma(gt.b, d_old, gt.a * inc) : gt.a * inc;
  }
}
}  // namespace beat_pde_detail
