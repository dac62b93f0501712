// The whole diffusion solve of a SMALL constant-coefficient grid in one launch of one workgroup.
//
// On a grid of a few thousand nodes (the Niederer slab at dx = 0.5 mm: 41 x 15 x 7 = 4305) every kernel of the
// multi-launch PCG finishes in a couple of microseconds and a solve is nothing but launch latency: 5 dependent
// launches per iteration, 0.30 ms per solve at 15-20 iterations.  Here one 512-thread workgroup does the right-hand
// side, all iterations and the update of x: the search direction lives in LDS (the only vector a node's neighbours
// read), r, the accumulated increment and 1/diag of a thread's own nodes in registers, the dot products are block
// reductions (wave shuffle + 8 partials in LDS, summed in a fixed order by every thread: deterministic), the
// convergence test is the library's (||r|| <= max(rtol ||b||, atol)), an iteration costs three barriers.
//
// Same arithmetic as beat_pde_solve's loop (Jacobi-PCG from x0 = v_ + e, x = v_ + e + sum alpha_j p_j), different
// summation order in the dot products.  Replaces, like beat_pde.hip, dolfinx assemble_vector + PETSc KSP.solve of
// src/beat/base_model.py:196-236 -- for the reference's own CPU-sized configurations (demos/niederer_benchmark.py).
#include "beat_pde_internal.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>

namespace {
using namespace beat_pde_detail;

// 512 threads, up to 16 nodes each: Niederer dx = 0.5 mm measured 0.17 / 0.15 / 0.16 ms per split step with 1024 / 512 /
// 256 threads (fewer waves: cheaper barriers and reductions, same LDS traffic; fewer still: too few waves to hide it)
constexpr int SMALL_THREADS = 512;
constexpr int SMALL_MAX_PER_THREAD = 16;
constexpr int SMALL_WAVES = SMALL_THREADS / 64;

struct SmallArgs {
  int nx, ny, nz, n;
  int margin;            // zero-filled doubles before and after the vector in LDS (>= plane + nx + 1)
  int doff[15];          // linear offsets of the 15 stencil points
  const double* tabA;    // (27, TABW) rows of A
  const double* tabB;    // rows of B
  const double* dinv;    // 27
  const double* v;       // v_
  double* x;             // solution (may alias v)
  const double* w[BEAT_MAX_STIM];
  double amp[BEAT_MAX_STIM];
  int nstim;
  double dt, rtol, atol;
  int max_it;
  GuessTerms gt;         // gt.d == nullptr: no guess
  double* st;            // PCG scalar state out (BB, RR, ITERS, REASON, STOP, NUPD)
};

__device__ __forceinline__ int axis_type_s(int i, int n) {
  if (n == 1) return 1;  // collapsed axis: no coupling along it
  if (i == 0) return 0;
  if (i == n - 1) return 2;
  return 1;
}

// sum over the workgroup, the same value in every thread; `slot` alternates between two partial buffers so that one
// barrier per reduction suffices
__device__ __forceinline__ double block_sum_all(double v, double* part) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
  __syncthreads();
  double s = 0.0;
#pragma unroll
  for (int k = 0; k < SMALL_WAVES; ++k) s += part[k];
  return s;
}

// two sums behind one barrier
__device__ __forceinline__ void block_sum_all2(double& u, double& v, double* part) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    u += __shfl_xor(u, off, 64);
    v += __shfl_xor(v, off, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    part[threadIdx.x >> 6] = u;
    part[SMALL_WAVES + (threadIdx.x >> 6)] = v;
  }
  __syncthreads();
  double su = 0.0, sv = 0.0;
#pragma unroll
  for (int k = 0; k < SMALL_WAVES; ++k) {
    su += part[k];
    sv += part[SMALL_WAVES + k];
  }
  u = su;
  v = sv;
}

template <int M>
__global__ __launch_bounds__(SMALL_THREADS) void pcg_small_kernel(SmallArgs a) {
  extern __shared__ double lds[];
  double* tabA = lds;                       // 27 * TABW
  double* tabB = tabA + 27 * TABW;          // 27 * TABW
  double* part = tabB + 27 * TABW;          // 6 * SMALL_WAVES (alternating partial buffers)
  double* dinv = part + 6 * SMALL_WAVES;    // 32
  double* pbuf = dinv + 32;                 // margin + n + margin
  double* p = pbuf + a.margin;
  const int tid = threadIdx.x;
  for (int i = tid; i < 27 * TABW; i += SMALL_THREADS) {
    tabA[i] = a.tabA[i];
    tabB[i] = a.tabB[i];
  }
  if (tid < 27) dinv[tid] = a.dinv[tid];
  const bool use_e = a.gt.d != nullptr && a.gt.use_e;
  for (int i = tid; i < a.margin; i += SMALL_THREADS) {
    pbuf[i] = 0.0;
    pbuf[a.margin + a.n + i] = 0.0;
  }
  // the nodes this thread owns: tid, tid + 512, ...
  // (r, the accumulated increment and the node type stay in registers; v_, e and 1/diag are re-read where needed)
  int type[M];
  double r[M], xinc[M];
  const int plane = a.nx * a.ny;
#pragma unroll
  for (int j = 0; j < M; ++j) {
    const int i = tid + j * SMALL_THREADS;
    type[j] = -1;
    r[j] = xinc[j] = 0.0;
    if (i < a.n) {
      const int iz = i / plane, rem = i - iz * plane, iy = rem / a.nx, ix = rem - iy * a.nx;
      type[j] = axis_type_s(ix, a.nx) + 3 * axis_type_s(iy, a.ny) + 9 * axis_type_s(iz, a.nz);
      p[i] = a.v[i];
    }
  }
  __syncthreads();
  // b = B v_ + dt sum amp_k w_k (kept in r until A x0 is subtracted; only its norm survives), then r = b - A (v_ + e)
  double acc = 0.0;
#pragma unroll
  for (int j = 0; j < M; ++j) {
    if (type[j] >= 0) {
      const int i = tid + j * SMALL_THREADS;
      const double* c = tabB + type[j] * TABW;
      double s = 0.0;
#pragma unroll
      for (int k = 0; k < 15; ++k) s = fma(c[k], p[i + a.doff[k]], s);
      double stim = 0.0;
      for (int k = 0; k < a.nstim; ++k) stim = fma(a.amp[k], a.w[k][i], stim);
      r[j] = fma(a.dt, stim, s);
      acc = fma(r[j], r[j], acc);
    }
  }
  const double bb = block_sum_all(acc, part);  // (its barrier also ends the reads of p = v_)
  if (use_e) {
#pragma unroll
    for (int j = 0; j < M; ++j)
      if (type[j] >= 0) p[tid + j * SMALL_THREADS] += a.gt.e[tid + j * SMALL_THREADS];
  }
  __syncthreads();
  double acc_rz = 0.0, acc_rr = 0.0;
#pragma unroll
  for (int j = 0; j < M; ++j) {
    if (type[j] >= 0) {
      const int i = tid + j * SMALL_THREADS;
      const double* c = tabA + type[j] * TABW;
      double s = 0.0;
#pragma unroll
      for (int k = 0; k < 15; ++k) s = fma(c[k], p[i + a.doff[k]], s);
      r[j] -= s;
      acc_rz = fma(r[j] * dinv[type[j]], r[j], acc_rz);
      acc_rr = fma(r[j], r[j], acc_rr);
    }
  }
  block_sum_all2(acc_rz, acc_rr, part + SMALL_WAVES);
  double rz = acc_rz, rr = acc_rr;
  const double rr0 = rr;
  const double tr = a.rtol * a.rtol * bb, ta = a.atol * a.atol;
  const double tol2 = tr > ta ? tr : ta;
  int iters = 0, reason = 0;
  if (rr <= tol2) reason = rr <= tr ? 2 : 3;
  // first direction p = D^-1 r (the reductions' barriers have ended the reads of p = x0)
  if (reason == 0) {
#pragma unroll
    for (int j = 0; j < M; ++j)
      if (type[j] >= 0) p[tid + j * SMALL_THREADS] = dinv[type[j]] * r[j];
  }
  __syncthreads();
  while (reason == 0) {
    if (iters >= a.max_it) {
      reason = -3;
      break;
    }
    double q[M], acc_pq = 0.0;
#pragma unroll
    for (int j = 0; j < M; ++j) {
      q[j] = 0.0;
      if (type[j] >= 0) {
        const int i = tid + j * SMALL_THREADS;
        const double* c = tabA + type[j] * TABW;
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < 15; ++k) s = fma(c[k], p[i + a.doff[k]], s);
        q[j] = s;
        acc_pq = fma(p[i], s, acc_pq);
      }
    }
    const double pq = block_sum_all(acc_pq, part + ((iters & 1) ? 0 : 3 * SMALL_WAVES));  // barrier: every read of p is done
    const double alpha = rz / pq;
    acc_rz = acc_rr = 0.0;
#pragma unroll
    for (int j = 0; j < M; ++j) {
      if (type[j] >= 0) {
        const int i = tid + j * SMALL_THREADS;
        xinc[j] = fma(alpha, p[i], xinc[j]);
        r[j] = fma(-alpha, q[j], r[j]);
        acc_rz = fma(r[j] * dinv[type[j]], r[j], acc_rz);
        acc_rr = fma(r[j], r[j], acc_rr);
      }
    }
    block_sum_all2(acc_rz, acc_rr, part + ((iters & 1) ? 4 * SMALL_WAVES : SMALL_WAVES));
    const double rzn = acc_rz;
    rr = acc_rr;
    const double beta = rzn / rz;
    rz = rzn;
    ++iters;
    if (rr <= tol2) {
      reason = rr <= tr ? 2 : 3;
      break;
    }
#pragma unroll
    for (int j = 0; j < M; ++j) {
      if (type[j] >= 0) {
        const int i = tid + j * SMALL_THREADS;
        p[i] = fma(beta, p[i], dinv[type[j]] * r[j]);  // own node: no other thread writes it, the reads ended at the barriers above
      }
    }
    __syncthreads();
  }
  // x = v_ + e + sum alpha_j p_j; the increment is recorded and the next guess prepared as the big kernels' x update does
#pragma unroll
  for (int j = 0; j < M; ++j) {
    if (type[j] >= 0) {
      const int i = tid + j * SMALL_THREADS;
      const double e_old = use_e ? a.gt.e[i] : 0.0;
      const double inc = e_old + xinc[j];
      a.x[i] = a.v[i] + inc;
      if (a.gt.d != nullptr) {
        const double d_old = beat_guess_needs_d(a.gt) ? a.gt.d[i] : 0.0;
        const double dp0 = beat_guess_needs_dp(a.gt, 0) ? a.gt.dp[0][i] : 0.0;
        const double dp1 = beat_guess_needs_dp(a.gt, 1) ? a.gt.dp[1][i] : 0.0;
        beat_guess_record(a.gt, a.gt.d + i, a.gt.e + i, inc, d_old, dp0, dp1, e_old);
      }
    }
  }
  if (tid == 0) {
    a.st[BB] = bb;
    a.st[RR0] = rr0;
    a.st[RR] = rr;
    a.st[RZ] = rz;
    a.st[ITERS] = (double)iters;
    a.st[NUPD] = (double)iters;
    a.st[REASON] = (double)reason;
    a.st[STOP] = 1.0;
  }
}

size_t small_lds_bytes(const beat_pde* pde, int margin) {
  return sizeof(double) * ((size_t)2 * 27 * TABW + 6 * SMALL_WAVES + 32 + (size_t)pde->n + 2 * (size_t)margin);
}

int small_margin(const beat_pde* pde) { return (int)(pde->g.plane + pde->g.nx + 1 + 7) / 8 * 8; }
}  // namespace

// Grids the one-workgroup solve takes: constant coefficients, Jacobi, both z faces physical, at most 16 nodes per
// thread (their r, increment and A p in registers) and the search direction + tables within the 160 KB of LDS.  BEAT_SMALL=0 switches it off for
// the process, beat_pde_set_small_grid_solve for one operator.
bool beat_small_available(const beat_pde* pde) {
  static const bool enabled = [] {
    const char* e = std::getenv("BEAT_SMALL");
    return !(e && e[0] == '0');
  }();
  if (!enabled || !pde->small_enabled || pde->var || pde->pc_ncoef != 1) return false;
  if (!pde->g.z_lo_phys || !pde->g.z_hi_phys) return false;
  if (pde->n > (int64_t)SMALL_MAX_PER_THREAD * SMALL_THREADS) return false;
  return small_lds_bytes(pde, small_margin(pde)) <= (size_t)150 * 1024;
}

// Enqueue one solve (initial-guess terms taken with beat_guess_begin, scalar results to dev_st[0..16)); no
// synchronisation: the caller reads dev_st back when it wants to and calls beat_guess_advance (after
// beat_guess_observe, if it has the iteration count by then).
int beat_small_launch(beat_pde* pde, const double* dev_v_prev, const double* const* host_dev_stim_w,
                      const double* host_stim_amp, int n_stim, double* dev_x, double rtol, double atol, int max_it,
                      double* dev_st) {
  BEAT_REQUIRE(pde->have_dt, "beat_pde_set_timestep has not been called");
  BEAT_REQUIRE(n_stim >= 0 && n_stim <= BEAT_MAX_STIM, "at most %d stimuli", BEAT_MAX_STIM);
  BEAT_REQUIRE(!pde->guess_pending, "the previous solve's deferred update has not been applied");
  const Geom& f = pde->g;
  SmallArgs a{};
  a.nx = f.nx;
  a.ny = f.ny;
  a.nz = f.nz;
  a.n = (int)pde->n;
  a.margin = small_margin(pde);
  for (int k = 0; k < 15; ++k)
    a.doff[k] = kOffsets[3 * k] + f.nx * kOffsets[3 * k + 1] + (int)f.plane * kOffsets[3 * k + 2];
  a.tabA = pde->d_tab(0);
  a.tabB = pde->d_tab(1);
  a.dinv = pde->d_dinv();
  a.v = dev_v_prev;
  a.x = dev_x;
  for (int k = 0; k < n_stim; ++k) {
    if (host_dev_stim_w[k] == nullptr || host_stim_amp[k] == 0.0) continue;
    a.w[a.nstim] = host_dev_stim_w[k];
    a.amp[a.nstim] = host_stim_amp[k];
    ++a.nstim;
  }
  a.dt = pde->dt;
  a.rtol = rtol;
  a.atol = atol;
  a.max_it = max_it;
  beat_guess_begin(pde);
  a.gt = pde->guess;
  a.st = dev_st;
  const size_t lds = small_lds_bytes(pde, a.margin);
  const int per_thread = (a.n + SMALL_THREADS - 1) / SMALL_THREADS;
  hipStream_t s = pde->ctx->stream;
#define BEAT_SMALL_LAUNCH(MV)                                                                                           \
  do {                                                                                                                  \
    static bool attr_set[64] = {}; /* per device: the attribute belongs to the function on ONE device */              \
    const int dev_ = pde->ctx->device & 63;                                                                             \
    if (!attr_set[dev_]) {                                                                                              \
      BEAT_HIP_CHECK(hipFuncSetAttribute((const void*)pcg_small_kernel<MV>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                         160 * 1024));                                                                  \
      attr_set[dev_] = true;                                                                                            \
    }                                                                                                                   \
    BEAT_KERNEL((pcg_small_kernel<MV>), dim3(1), dim3(SMALL_THREADS), lds, s, a);                                \
  } while (0)
  if (per_thread <= 2)
    BEAT_SMALL_LAUNCH(2);
  else if (per_thread <= 4)
    BEAT_SMALL_LAUNCH(4);
  else if (per_thread <= 6)
    BEAT_SMALL_LAUNCH(6);
  else if (per_thread <= 8)
    BEAT_SMALL_LAUNCH(8);
  else if (per_thread <= 10)
    BEAT_SMALL_LAUNCH(10);
  else if (per_thread <= 12)
    BEAT_SMALL_LAUNCH(12);
  else
    BEAT_SMALL_LAUNCH(16);
#undef BEAT_SMALL_LAUNCH
  BEAT_LAUNCH_CHECK();
  return BEAT_OK;
}

int beat_small_solve(beat_pde* pde, const double* dev_v_prev, const double* const* host_dev_stim_w,
                     const double* host_stim_amp, int n_stim, double* dev_x, double rtol, double atol, int max_it,
                     beat_ksp_info* info) {
  if (int rc = beat_small_launch(pde, dev_v_prev, host_dev_stim_w, host_stim_amp, n_stim, dev_x, rtol, atol, max_it, pde->d_st))
    return rc;
  hipStream_t s = pde->ctx->stream;
  double* h = pde->ctx->h_pinned;
  BEAT_HIP_CHECK(hipMemcpyAsync(h, pde->d_st, sizeof(double) * 16, hipMemcpyDeviceToHost, s));
  BEAT_HIP_CHECK(hipStreamSynchronize(s));
  if (pde->guess.d != nullptr) {  // the kernel recorded this solve's increment
    beat_guess_observe(pde, (int)h[ITERS]);
    beat_guess_advance(pde);
  }
  const int iters = (int)h[ITERS], reason = (int)h[REASON];
  pde->last_iters = iters;
  if (info) {
    info->iterations = iters;
    info->converged_reason = reason;
    info->residual_norm = std::sqrt(h[RR]);
    info->rhs_norm = std::sqrt(h[BB]);
  }
  if (reason < 0) {
    beat_set_error("PCG did not converge in %d iterations (||r|| = %.3e, ||b|| = %.3e)", iters, std::sqrt(h[RR]),
                   std::sqrt(h[BB]));
    return BEAT_ENOTCONV;
  }
  return BEAT_OK;
}
