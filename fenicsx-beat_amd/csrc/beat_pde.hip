// Diffusion step on a structured z-slab: matrix-free P1 operators as a 15-point stencil,
// right-hand-side build and Jacobi-PCG.  Replaces dolfinx assemble_vector + PETSc KSP.solve of
// src/beat/base_model.py:196-236 (forms: src/beat/monodomain_model.py:68-98).
// This file: the constant-coefficient (27 node types) kernels, the PCG vector kernels and the C ABI;
// beat_pde_var.hip holds the per-node-coefficient form of the same operators.
//
// Data layout: a field is nx*ny*nz_local doubles, x fastest, with one ghost xy-plane addressable
// on either side.  The 27x15 coefficient tables (one row per boundary type of a node) come from
// the caller, derived by element assembly; the interior row (type 13) is passed by value so it
// lives in SGPRs, the 26 boundary rows are looked up from a small device table by the few lanes
// that need them.
//
// Stencil kernel structure (gfx950): a 256-thread workgroup owns a TX x TY tile of 1024 nodes (256 x 4 when
// the rows are long enough, else 128 x 8 or 64 x 16) and marches along z through a chunk of planes.  Three
// (TY+2)x(TX+2) planes live in a ring of LDS slots; the next two planes are prefetched into registers while the
// current one is computed (global loads are row-contiguous, 8 B/lane).  Every output needs 15 LDS reads
// (ds_read_b64, conflict-free: lanes read consecutive doubles); a thread computes 4 rows, unrolled by 2, so the
// compiler shares the in-plane reads.  Algorithmic HBM traffic: 8 B read + 8 B written per node per operator
// application; the tile halo and the chunk's two extra planes are re-reads that the XCD-local L2 mostly absorbs
// (measured: reads 1.09x algorithmic; tiles are dealt to XCDs in contiguous runs, see tile_of_block()).
// PMC picture of the SpMV at 512^3 (0.56 ms): LDS-limited to 4 workgroups per CU (37 KB each), no bank
// conflicts, LDS pipe 37 % busy, waves parked on s_waitcnt / barriers 64 % of their lifetime -- latency bound.
#include "beat_pde_internal.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

const int beat_pde_detail::kOffsets[45] = {0, 0, 0,  1, 0, 0,  -1, 0, 0,  0, 1, 0,  0, -1, 0,  0, 0, 1,  0, 0, -1,
                          1, 1, 0,  -1, -1, 0,  0, 1, 1,  0, -1, -1,  1, 0, 1,  -1, 0, -1,
                          1, 1, 1,  -1, -1, -1};

namespace {
using namespace beat_pde_detail;

// Tile shape of the stencil kernels: TX x TY nodes per workgroup and plane (TX*TY = 1024, four
// rows per thread).  Wider tiles read longer contiguous row segments (DRAM page locality) at the
// price of more y-halo rows (served by L2).
template <int TX_, int TY_>
struct Tile {
  static constexpr int TX = TX_, TY = TY_;
  static constexpr int PITCH = TX + 2;
  static constexpr int SLOT = (TY + 2) * PITCH;                       // doubles per staged plane
  static constexpr int NLOAD = (SLOT + BEAT_BLOCK - 1) / BEAT_BLOCK;  // staged values per thread
  static constexpr int ROWS_PER_THREAD = TY / (BEAT_BLOCK / TX);      // 4
  static_assert(TX * TY == 1024 && ROWS_PER_THREAD == 4, "tile must hold 1024 nodes");
};
constexpr int TARGET_BLOCKS = 1024;



struct Coef15 {
  double c[15];
};


struct StencilArgs {
  const double* x;     // input field
  const double* x2;    // PC: residual r (centre values)
  double* y;           // APPLY: y | SPMV: q | RHS: r | PC: output of the pass
  double* y2;          // RHS: p
  double* y3;          // RHS: x (copy of v_) or nullptr
  const double* tab;   // APPLY/SPMV: operator table | RHS: mass table
  const double* tab2;  // RHS: stiffness table
  const double* dinv;  // RHS: 1/diag(A) per type
  Coef15 ci, ci2;
  double dinv_i;
  double cm, omt_dt, dt;
  const double* w[BEAT_MAX_STIM];
  double amp[BEAT_MAX_STIM];
  int nstim;
  double* partials;
  const double* st;
  double c_in, c_r;    // PC: staged input is scaled by c_in * D^-1 (first pass only); weight of the D^-1 r term
  int pc_first, pc_last;
};

// APPLY: y = T x.  SPMV_DOT: q = A p, partial p.q.  RHS: see beat_pde_rhs.
// (Forming p = D^-1 r + beta p_old while staging was tried and rejected: the halo makes it re-read two
//  fields 1.5x, 1.8 ms against 0.58 + 0.66 ms for SpMV + a streaming p-update at 512^3.)
// PC: one Horner pass of the polynomial preconditioner z = sum_j c_j (D^-1 A)^j D^-1 r:
//     out = c_r D^-1 r + D^-1 A in,   in = c_in D^-1 r on the first pass, the previous output afterwards;
//     the last pass also reduces r.z.

__device__ __forceinline__ int axis_type(int i, int n, int lo_phys, int hi_phys) {
  if (n == 1 && lo_phys && hi_phys) return 1;  // collapsed axis: no coupling along it
  if (i == 0 && lo_phys) return 0;
  if (i == n - 1 && hi_phys) return 2;
  return 1;
}

// Blocks are dealt round-robin to the 8 XCDs; give each XCD a contiguous run of tiles so that
// tiles sharing a halo share an L2.  Pure performance heuristic (placement is not relied upon).
__device__ __forceinline__ int tile_of_block(int b, int total) {
  const int per = (total + 7) >> 3;
  return (b & 7) * per + (b >> 3);
}

// Per-thread staging descriptors, computed once per tile: which elements of the (TY+2)x(TX+2)
// staged plane this thread moves, their offset inside an xy-plane (-1: outside the box or idle),
template <class T>
struct StageDesc {
  int off[T::NLOAD];
  int txy[T::NLOAD];  // tx + 3 ty of the staged element (only read by the PC mode)
};

template <class T>
__device__ __forceinline__ void stage_setup(StageDesc<T>& d, const Geom& g, int x0, int y0) {
  constexpr int NLOAD = T::NLOAD, PITCH = T::PITCH, SLOT = T::SLOT;
#pragma unroll
  for (int l = 0; l < NLOAD; ++l) {
    const int idx = threadIdx.x + l * BEAT_BLOCK;
    const int row = idx / PITCH, col = idx - row * PITCH;
    const int gx = x0 + col - 1, gy = y0 + row - 1;
    const bool ok = idx < SLOT && gx >= 0 && gx < g.nx && gy >= 0 && gy < g.ny;
    d.off[l] = ok ? gy * g.nx + gx : -1;
    d.txy[l] = axis_type(gx, g.nx, 1, 1) + 3 * axis_type(gy, g.ny, 1, 1);
  }
}

template <int MODE, class T>
__device__ __forceinline__ void stage_load(double (&reg)[T::NLOAD], const StageDesc<T>& d, const StencilArgs& a,
                                           const Geom& g, int gz) {
  constexpr int NLOAD = T::NLOAD;
  const bool zvalid = (gz >= 0 || !g.z_lo_phys) && (gz < g.nz || !g.z_hi_phys);
  const double* __restrict__ base = a.x + (int64_t)gz * g.plane;
#pragma unroll
  for (int l = 0; l < NLOAD; ++l) reg[l] = (zvalid && d.off[l] >= 0) ? base[d.off[l]] : 0.0;
  if (MODE == MODE_PC) {
    if (a.pc_first) {  // stage c_in * D^-1 r instead of r
      const int tz9 = 9 * axis_type(gz, g.nz, g.z_lo_phys, g.z_hi_phys);
#pragma unroll
      for (int l = 0; l < NLOAD; ++l) {
        const int type = d.txy[l] + tz9;
        reg[l] *= a.c_in * ((type == 13) ? a.dinv_i : a.dinv[type]);
      }
    }
  }
}

template <class T>
__device__ __forceinline__ void stage_store(const double (&reg)[T::NLOAD], double* __restrict__ slot) {
  constexpr int NLOAD = T::NLOAD, SLOT = T::SLOT;
#pragma unroll
  for (int l = 0; l < NLOAD; ++l) {
    const int idx = threadIdx.x + l * BEAT_BLOCK;
    if (idx < SLOT) slot[idx] = reg[l];
  }
}

// One output plane of the tile from the three staged planes.
template <int MODE, class T>
__device__ __forceinline__ void compute_plane(const Geom& g, const StencilArgs& a, const double* __restrict__ lds,
                                              int z, int x0, int y0, int lx, int wave, int tx, double& acc0,
                                              double& acc1, double& acc2) {
  constexpr int PITCH = T::PITCH, SLOT = T::SLOT, ROWS_PER_THREAD = T::ROWS_PER_THREAD;
  const double* __restrict__ Pm = lds + ((z + 2) % 3) * SLOT;  // plane z-1
  const double* __restrict__ P0 = lds + (z % 3) * SLOT;
  const double* __restrict__ Pp = lds + ((z + 1) % 3) * SLOT;
  const int tz = axis_type(z, g.nz, g.z_lo_phys, g.z_hi_phys);
  const int gx = x0 + lx;
#ifndef BEAT_ROW_UNROLL
#define BEAT_ROW_UNROLL 2  // 4 costs the RHS kernel half its occupancy (173 vs 128 VGPRs) for no gain
#endif
#pragma unroll BEAT_ROW_UNROLL
  for (int rr = 0; rr < ROWS_PER_THREAD; ++rr) {
    const int ly = wave * ROWS_PER_THREAD + rr;
    const int gy = y0 + ly;
    const int c = (ly + 1) * PITCH + (lx + 1);
    double v[15];
    v[0] = P0[c];
    v[1] = P0[c + 1];
    v[2] = P0[c - 1];
    v[3] = P0[c + PITCH];
    v[4] = P0[c - PITCH];
    v[5] = Pp[c];
    v[6] = Pm[c];
    v[7] = P0[c + PITCH + 1];
    v[8] = P0[c - PITCH - 1];
    v[9] = Pp[c + PITCH];
    v[10] = Pm[c - PITCH];
    v[11] = Pp[c + 1];
    v[12] = Pm[c - 1];
    v[13] = Pp[c + PITCH + 1];
    v[14] = Pm[c - PITCH - 1];
    if (gx < g.nx && gy < g.ny) {
      const int ty = axis_type(gy, g.ny, 1, 1);
      const int type = tx + 3 * ty + 9 * tz;
      const int64_t gi = (int64_t)z * g.plane + (int64_t)gy * g.nx + gx;
      double s = 0.0;
#pragma unroll
      for (int k = 0; k < 15; ++k) s = fma(a.ci.c[k], v[k], s);
      if (MODE == MODE_RHS) {
        double s2 = 0.0;
#pragma unroll
        for (int k = 0; k < 15; ++k) s2 = fma(a.ci2.c[k], v[k], s2);
        double di = a.dinv_i;
        if (type != 13) {
          s = 0.0;
          s2 = 0.0;
          const double* __restrict__ r1 = a.tab + type * TABW;
          const double* __restrict__ r2 = a.tab2 + type * TABW;
#pragma unroll
          for (int k = 0; k < 15; ++k) {
            s = fma(r1[k], v[k], s);
            s2 = fma(r2[k], v[k], s2);
          }
          di = a.dinv[type];
        }
        // s = (Mass v)_i, s2 = (K v)_i
        double stim = 0.0;
        for (int k = 0; k < a.nstim; ++k) stim = fma(a.amp[k], a.w[k][gi], stim);
        const double b = a.cm * s - a.omt_dt * s2 + a.dt * stim;
        const double r = a.dt * (stim - s2);
        const double zz = di * r;
        a.y[gi] = r;
        a.y2[gi] = zz;
        if (a.y3 != nullptr) a.y3[gi] = v[0];
        acc0 = fma(b, b, acc0);
        acc1 = fma(r, zz, acc1);
        acc2 = fma(r, r, acc2);
      } else {
        double di = a.dinv_i;
        if (type != 13) {
          s = 0.0;
          const double* __restrict__ r1 = a.tab + type * TABW;
#pragma unroll
          for (int k = 0; k < 15; ++k) s = fma(r1[k], v[k], s);
          if (MODE == MODE_PC) di = a.dinv[type];
        }
        if (MODE == MODE_PC) {
          const double ri = a.x2[gi];
          const double out = di * fma(a.c_r, ri, s);
          a.y[gi] = out;
          if (a.pc_last) acc0 = fma(ri, out, acc0);
        } else {
          a.y[gi] = s;
          if (MODE == MODE_SPMV_DOT) acc0 = fma(v[0], s, acc0);
        }
      }
    }
  }
}

template <int MODE, class T>
__global__ __launch_bounds__(BEAT_BLOCK) void stencil_kernel(Geom g, StencilArgs a) {
  constexpr int NLOAD = T::NLOAD, SLOT = T::SLOT, TX = T::TX, TY = T::TY;
  __shared__ double lds[3 * SLOT];
  __shared__ double red[4];
  if (MODE == MODE_SPMV_DOT || MODE == MODE_PC) {
    if (a.st[STOP] != 0.0) return;  // convergence latch: nothing left to do in this solve
  }
  const int t = tile_of_block(blockIdx.x, g.total);
  if (t >= g.total) return;
  const int tile_x = t % g.tiles_x;
  const int tile_y = (t / g.tiles_x) % g.tiles_y;
  const int chunk = t / (g.tiles_x * g.tiles_y);
  const int x0 = tile_x * TX, y0 = tile_y * TY;
  const int z_begin = g.z_lo + chunk * g.zc;
  const int z_end = min(z_begin + g.zc, g.z_hi);

  const int lx = threadIdx.x & (TX - 1);
  const int wave = threadIdx.x / TX;  // thread row group: rows 4*group .. 4*group+3 of the tile
  const int tx = axis_type(x0 + lx, g.nx, 1, 1);

  double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0;  // block partial sums (mode dependent)

  StageDesc<T> d;
  stage_setup<T>(d, g, x0, y0);
  // planes z_begin-1 and z_begin go straight to LDS; the next two are held in registers so that
  // two planes of global loads are always in flight behind the plane being computed
  double ra[NLOAD], rb[NLOAD];
  stage_load<MODE, T>(ra, d, a, g, z_begin - 1);
  stage_load<MODE, T>(rb, d, a, g, z_begin);
  stage_store<T>(ra, lds + ((z_begin + 2) % 3) * SLOT);
  stage_store<T>(rb, lds + (z_begin % 3) * SLOT);
  stage_load<MODE, T>(ra, d, a, g, z_begin + 1);
  if (z_begin + 2 <= z_end) stage_load<MODE, T>(rb, d, a, g, z_begin + 2);

  for (int z = z_begin; z < z_end; z += 2) {
    stage_store<T>(ra, lds + ((z + 1) % 3) * SLOT);
    if (z + 3 <= z_end) stage_load<MODE, T>(ra, d, a, g, z + 3);
    __syncthreads();
    compute_plane<MODE, T>(g, a, lds, z, x0, y0, lx, wave, tx, acc0, acc1, acc2);
    __syncthreads();
    if (z + 1 >= z_end) break;
    stage_store<T>(rb, lds + ((z + 2) % 3) * SLOT);
    if (z + 4 <= z_end) stage_load<MODE, T>(rb, d, a, g, z + 4);
    __syncthreads();
    compute_plane<MODE, T>(g, a, lds, z + 1, x0, y0, lx, wave, tx, acc0, acc1, acc2);
    __syncthreads();
  }

  if (MODE == MODE_SPMV_DOT || MODE == MODE_PC) {
    const double s0 = beat_block_sum(acc0, red);
    if (threadIdx.x == 0) a.partials[g.part_off + t] = s0;
  } else if (MODE == MODE_RHS) {
    const double s0 = beat_block_sum(acc0, red);
    const double s1 = beat_block_sum(acc1, red);
    const double s2 = beat_block_sum(acc2, red);
    if (threadIdx.x == 0) {
      a.partials[g.part_off + t] = s0;
      a.partials[BEAT_MAX_PARTIALS + g.part_off + t] = s1;
      a.partials[2 * BEAT_MAX_PARTIALS + g.part_off + t] = s2;
    }
  }
}

// Sum `count` block partials of `nsum` quantities in a fixed order and store them at out[0..nsum).
// `then` (round 5: one launch instead of two behind a residual update / a right-hand side on a single slab; the same arithmetic in the
// same order as the kernels it replaces): 1 = the scalar roll of the iteration (pcg_next_kernel: beta, iteration count, latch) on the
// state `roll_st` the sums were just written into; 2 = the start of a solve (pcg_begin_kernel) with rtol / atol / max_it.
__device__ __forceinline__ void beat_pcg_roll(double* st) {
  st[BETA] = st[RZN] / st[RZ];
  st[RZ] = st[RZN];
  st[RR] = st[RRN];
  st[ITERS] += 1.0;
  const double tr = st[RTOL] * st[RTOL] * st[BB];
  if (st[RR] <= st[TOL2]) {
    st[STOP] = 1.0;
    st[REASON] = st[RR] <= tr ? 2.0 : 3.0;
  } else if (st[ITERS] >= st[MAXIT]) {
    st[STOP] = 1.0;
    st[REASON] = -3.0;
  }
}
__device__ __forceinline__ void beat_pcg_begin(double* st, double rtol, double atol, double max_it) {
  const double bb = st[BB], rr = st[RR];
  const double tr = rtol * rtol * bb, ta = atol * atol;
  const double tol2 = tr > ta ? tr : ta;
  st[TOL2] = tol2;
  st[ITERS] = 0.0;
  st[NUPD] = 0.0;
  st[RTOL] = rtol;
  st[ATOL] = atol;
  st[MAXIT] = max_it;
  st[BETA] = 0.0;
  st[RR0] = rr;
  const bool done = rr <= tol2;
  st[STOP] = done ? 1.0 : 0.0;
  st[REASON] = done ? (rr <= tr ? 2.0 : 3.0) : 0.0;
}
__global__ __launch_bounds__(BEAT_BLOCK) void reduce_partials_kernel(const double* __restrict__ partials, int count, int nsum, double* out,
                                                                     const double* st, double* counter, int then, double* roll_st,
                                                                     double rtol, double atol, double max_it) {
  __shared__ double red[4];
  if (st != nullptr && st[STOP] != 0.0) return;
  if (counter != nullptr && threadIdx.x == 0) counter[0] += 1.0;  // one more executed residual update
  for (int k = 0; k < nsum; ++k) {
    double s = 0.0;
    for (int i = threadIdx.x; i < count; i += BEAT_BLOCK) s += partials[(int64_t)k * BEAT_MAX_PARTIALS + i];
    s = beat_block_sum(s, red);
    if (threadIdx.x == 0) out[k] = s;
  }
  if (then != 0 && threadIdx.x == 0) {  // (the thread that wrote the sums: its own stores are ahead of these loads)
    if (then == 1) beat_pcg_roll(roll_st);
    else beat_pcg_begin(roll_st, rtol, atol, max_it);
  }
}

__global__ void pcg_begin_kernel(double* st, double rtol, double atol, double max_it) { beat_pcg_begin(st, rtol, atol, max_it); }

__global__ void pcg_next_kernel(double* st) {
  if (st[STOP] != 0.0) return;
  beat_pcg_roll(st);
}

// x += alpha p ; r -= alpha q ; partial sums of r.z (z = D^-1 r) and r.r.  Row-per-wave so the node
// type (hence 1/diag) needs no integer division per element.
template <bool VAR>
__global__ __launch_bounds__(BEAT_BLOCK) void cg_update_kernel(Geom g, const double* __restrict__ st,
                                                               double* __restrict__ x,
                                                               double* __restrict__ r,
                                                               const double* __restrict__ p,
                                                               const double* __restrict__ q,
                                                               const double* __restrict__ dinv,
                                                               double dinv_i, double* __restrict__ partials) {
  __shared__ double red[4];
  if (st[STOP] != 0.0) return;
  const double alpha = st[RZ] / st[PQ];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nrows = g.ny * g.nz;
  double s_rz = 0.0, s_rr = 0.0;
  for (int row = blockIdx.x * 4 + wave; row < nrows; row += gridDim.x * 4) {
    const int iz = row / g.ny, iy = row - iz * g.ny;
    const int tyz = 3 * axis_type(iy, g.ny, 1, 1) + 9 * axis_type(iz, g.nz, g.z_lo_phys, g.z_hi_phys);
    const int64_t base = (int64_t)row * g.nx;
    for (int ix = lane; ix < g.nx; ix += 64) {
      const int type = axis_type(ix, g.nx, 1, 1) + tyz;
      const int64_t i = base + ix;
      const double di = VAR ? dinv[i] : (type == 13) ? dinv_i : dinv[type];
      const double pi = p[i], qi = q[i];
      const double xi = fma(alpha, pi, x[i]);
      const double ri = fma(-alpha, qi, r[i]);
      x[i] = xi;
      r[i] = ri;
      s_rz = fma(ri * di, ri, s_rz);
      s_rr = fma(ri, ri, s_rr);
    }
  }
  const double a0 = beat_block_sum(s_rz, red);
  const double a1 = beat_block_sum(s_rr, red);
  if (threadIdx.x == 0) {
    partials[blockIdx.x] = a0;
    partials[BEAT_MAX_PARTIALS + blockIdx.x] = a1;
  }
}

// p = D^-1 r + beta p
template <bool VAR>
__global__ __launch_bounds__(BEAT_BLOCK) void cg_pupdate_kernel(Geom g, const double* __restrict__ st,
                                                                const double* __restrict__ r,
                                                                double* __restrict__ p,
                                                                const double* __restrict__ dinv,
                                                                double dinv_i) {
  if (st[STOP] != 0.0) return;
  const double beta = st[BETA];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nrows = g.ny * g.nz;
  for (int row = blockIdx.x * 4 + wave; row < nrows; row += gridDim.x * 4) {
    const int iz = row / g.ny, iy = row - iz * g.ny;
    const int tyz = 3 * axis_type(iy, g.ny, 1, 1) + 9 * axis_type(iz, g.nz, g.z_lo_phys, g.z_hi_phys);
    const int64_t base = (int64_t)row * g.nx;
    for (int ix = lane; ix < g.nx; ix += 64) {
      const int type = axis_type(ix, g.nx, 1, 1) + tyz;
      const int64_t i = base + ix;
      const double di = VAR ? dinv[i] : (type == 13) ? dinv_i : dinv[type];
      p[i] = fma(beta, p[i], di * r[i]);
    }
  }
}

// ---- deferred-x PCG (single-slab solve) -----------------------------------------------------------
// x only matters when the solve ends, so the per-iteration x += alpha p (24 B/node) is replaced by
// keeping the search directions in a ring (the p-update writes out of place at the same traffic) and
// adding sum_i alpha_i p_i to x once the ring is full or the solve is over (8 (k+2) B/node in total).

// r -= alpha q ; alpha = rz/pq is also stored in alphas[slot]; partial sums of r.D^-1 r and r.r.
template <bool VAR>
__global__ __launch_bounds__(BEAT_BLOCK) void cg_update_r_kernel(Geom g, const double* __restrict__ st,
                                                                 double* __restrict__ r,
                                                                 const double* __restrict__ q,
                                                                 const double* __restrict__ dinv, double dinv_i,
                                                                 double* __restrict__ partials,
                                                                 double* __restrict__ alphas, int slot) {
  __shared__ double red[4];
  if (st[STOP] != 0.0) return;
  const double alpha = st[RZ] / st[PQ];
  if (blockIdx.x == 0 && threadIdx.x == 0) alphas[slot] = alpha;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nrows = g.ny * g.nz;
  double s_rz = 0.0, s_rr = 0.0;
  for (int row = blockIdx.x * 4 + wave; row < nrows; row += gridDim.x * 4) {
    const int iz = row / g.ny, iy = row - iz * g.ny;
    const int tyz = 3 * axis_type(iy, g.ny, 1, 1) + 9 * axis_type(iz, g.nz, g.z_lo_phys, g.z_hi_phys);
    const int64_t base = (int64_t)row * g.nx;
    for (int ix = lane; ix < g.nx; ix += 64) {
      const int type = axis_type(ix, g.nx, 1, 1) + tyz;
      const int64_t i = base + ix;
      const double di = VAR ? dinv[i] : (type == 13) ? dinv_i : dinv[type];
      const double ri = fma(-alpha, q[i], r[i]);
      r[i] = ri;
      s_rz = fma(ri * di, ri, s_rz);
      s_rr = fma(ri, ri, s_rr);
    }
  }
  const double a0 = beat_block_sum(s_rz, red);
  const double a1 = beat_block_sum(s_rr, red);
  if (threadIdx.x == 0) {
    partials[blockIdx.x] = a0;
    partials[BEAT_MAX_PARTIALS + blockIdx.x] = a1;
  }
}

// p_new = D^-1 r + beta p_old (out of place)
template <bool VAR>
__global__ __launch_bounds__(BEAT_BLOCK) void cg_pupdate_oop_kernel(Geom g, const double* __restrict__ st,
                                                                    const double* __restrict__ r,
                                                                    const double* __restrict__ p_old,
                                                                    double* __restrict__ p_new,
                                                                    const double* __restrict__ dinv, double dinv_i) {
  if (st[STOP] != 0.0) return;
  const double beta = st[BETA];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nrows = g.ny * g.nz;
  for (int row = blockIdx.x * 4 + wave; row < nrows; row += gridDim.x * 4) {
    const int iz = row / g.ny, iy = row - iz * g.ny;
    const int tyz = 3 * axis_type(iy, g.ny, 1, 1) + 9 * axis_type(iz, g.nz, g.z_lo_phys, g.z_hi_phys);
    const int64_t base = (int64_t)row * g.nx;
    for (int ix = lane; ix < g.nx; ix += 64) {
      const int type = axis_type(ix, g.nx, 1, 1) + tyz;
      const int64_t i = base + ix;
      const double di = VAR ? dinv[i] : (type == 13) ? dinv_i : dinv[type];
      p_new[i] = fma(beta, p_old[i], di * r[i]);
    }
  }
}

// x += sum_{j < nvalid} alphas[j] * P_j, nvalid = clamp(executed updates - ring_base, 0, PRING).
// Runs regardless of the latch: it is what brings x up to date after convergence.
// With guess terms (gt.d != nullptr): inc = e + sum alpha_j P_j;  x += inc;  then beat_guess_record (d, e updated).
__global__ __launch_bounds__(BEAT_BLOCK) void x_flush_kernel(int64_t n, const double* __restrict__ st,
                                                             double* __restrict__ x,
                                                             const double* __restrict__ ring, int64_t fld,
                                                             const double* __restrict__ alphas, int ring_base,
                                                             int only_if_full, GuessTerms gt, int R) {
  int nvalid = (int)st[NUPD] - ring_base;
  nvalid = nvalid < 0 ? 0 : (nvalid > R ? R : nvalid);
  // in-loop flushes are enqueued ahead of time: they must do nothing unless their ring cycle really
  // filled up (a partially filled last cycle is flushed once, after the host has seen the latch)
  if ((only_if_full && nvalid < R) || (nvalid == 0 && gt.d == nullptr)) return;
  double a[PRING_MAX];
#pragma unroll
  for (int j = 0; j < PRING_MAX; ++j) a[j] = (j < nvalid) ? alphas[j] : 0.0;
  const int64_t stride = (int64_t)gridDim.x * BEAT_BLOCK;
  if (gt.d != nullptr) {
    double* d = gt.d;
    double* e = gt.e;
    for (int64_t i = (int64_t)blockIdx.x * BEAT_BLOCK + threadIdx.x; i < n; i += stride) {
      const double e_old = beat_guess_needs_e(gt) ? e[i] : 0.0;
      const double d_old = beat_guess_needs_d(gt) ? d[i] : 0.0;
      const double dp0 = beat_guess_needs_dp(gt, 0) ? gt.dp[0][i] : 0.0;
      const double dp1 = beat_guess_needs_dp(gt, 1) ? gt.dp[1][i] : 0.0;
      double inc = gt.accumulate ? 0.0 : e_old;
#pragma unroll
      for (int j = 0; j < PRING_MAX; ++j)
        if (j < nvalid) inc = fma(a[j], ring[(int64_t)j * fld + i], inc);
      x[i] += inc;
      beat_guess_record(gt, d + i, e + i, inc, d_old, dp0, dp1, e_old);
    }
    return;
  }
  for (int64_t i = (int64_t)blockIdx.x * BEAT_BLOCK + threadIdx.x; i < n; i += stride) {
    double xi = x[i];
#pragma unroll
    for (int j = 0; j < PRING_MAX; ++j)
      if (j < nvalid) xi = fma(a[j], ring[(int64_t)j * fld + i], xi);
    x[i] = xi;
  }
}

// p = z + beta p  (z = M^-1 r already formed by the polynomial preconditioner)
__global__ __launch_bounds__(BEAT_BLOCK) void cg_pupdate_z_kernel(int64_t n, const double* __restrict__ st,
                                                                  const double* __restrict__ z,
                                                                  double* __restrict__ p) {
  if (st[STOP] != 0.0) return;
  const double beta = st[BETA];
  const int64_t stride = (int64_t)gridDim.x * BEAT_BLOCK;
  if (beta == 0.0) {  // first direction: p may hold anything
    for (int64_t i = (int64_t)blockIdx.x * BEAT_BLOCK + threadIdx.x; i < n; i += stride) p[i] = z[i];
  } else {
    for (int64_t i = (int64_t)blockIdx.x * BEAT_BLOCK + threadIdx.x; i < n; i += stride) p[i] = fma(beta, p[i], z[i]);
  }
}


}  // namespace

using namespace beat_pde_detail;

// vector kernels are compiled for both the typed (27 node types) and the per-node 1/diag
#define BEAT_LAUNCH_VEC(pde, kernel, ...)                                   \
  do {                                                                      \
    if ((pde)->var)                                                         \
      BEAT_KERNEL((kernel<true>), __VA_ARGS__);                      \
    else                                                                    \
      BEAT_KERNEL((kernel<false>), __VA_ARGS__);                     \
  } while (0)

static Coef15 interior(const double* tab) {
  Coef15 c;
  for (int k = 0; k < 15; ++k) c.c[k] = tab[13 * 15 + k];
  return c;
}

static int upload_tables(beat_pde* pde);
extern "C" int beat_pde_destroy(beat_pde* pde);

extern "C" const int* beat_stencil_offsets(void) { return kOffsets; }

extern "C" int beat_pde_create(beat_ctx* ctx, const int64_t n[3], int z_lo_phys, int z_hi_phys,
                               const double* host_mass_tab, const double* host_stiff_tab,
                               beat_pde** out) {
  BEAT_REQUIRE(ctx != nullptr && n != nullptr && host_mass_tab && host_stiff_tab && out, "null argument");
  BEAT_REQUIRE(n[0] >= 1 && n[1] >= 1 && n[2] >= 1, "node counts must be >= 1");
  BEAT_REQUIRE(n[0] * n[1] * n[2] < ((int64_t)1 << 40) && n[0] < (1 << 30) && n[1] * n[2] < (1LL << 31),
               "grid too large");
  beat_pde* p = new beat_pde();
  p->ctx = ctx;
  Geom& g = p->g;
  g.nx = (int)n[0];
  g.ny = (int)n[1];
  g.nz = (int)n[2];
  g.plane = n[0] * n[1];
  g.z_lo_phys = z_lo_phys ? 1 : 0;
  g.z_hi_phys = z_hi_phys ? 1 : 0;
  // tile width: as wide as the rows allow (long contiguous reads), overridable for experiments
  int tile_tx = g.nx >= 256 ? 256 : g.nx >= 128 ? 128 : 64;
  if (const char* e = std::getenv("BEAT_TILE_TX")) {
    const int v = std::atoi(e);
    if (v == 64 || v == 128 || v == 256) tile_tx = v;
  }
  g.tile_tx = tile_tx;
  const int TX = tile_tx, TY = 1024 / tile_tx;
  g.tiles_x = (g.nx + TX - 1) / TX;
  g.tiles_y = (g.ny + TY - 1) / TY;
  const int tiles = g.tiles_x * g.tiles_y;
  int nchunks = std::max(1, (TARGET_BLOCKS + tiles - 1) / tiles);
  nchunks = std::min(nchunks, g.nz);
  g.zc = (g.nz + nchunks - 1) / nchunks;
  g.nchunks = (g.nz + g.zc - 1) / g.zc;
  const int64_t total = (int64_t)tiles * g.nchunks;
  if (total > BEAT_MAX_PARTIALS) {
    // fall back to fewer, longer chunks
    g.nchunks = std::max(1, (int)(BEAT_MAX_PARTIALS / tiles));
    g.zc = (g.nz + g.nchunks - 1) / g.nchunks;
    g.nchunks = (g.nz + g.zc - 1) / g.zc;
  }
  BEAT_REQUIRE((int64_t)tiles * g.nchunks <= BEAT_MAX_PARTIALS, "xy plane too large: %d tiles", tiles);
  g.total = tiles * g.nchunks;
  g.z_lo = 0;
  g.z_hi = g.nz;
  g.part_off = 0;
  p->n = n[0] * n[1] * n[2];
  const int64_t rows = (int64_t)g.ny * g.nz;
  p->vec_grid = (unsigned)std::min<int64_t>(2048, std::max<int64_t>(1, (rows + 3) / 4));
  std::memcpy(p->h_mass, host_mass_tab, sizeof(p->h_mass));
  std::memcpy(p->h_stiff, host_stiff_tab, sizeof(p->h_stiff));
  BEAT_HIP_CHECK(hipSetDevice(ctx->device));
  BEAT_HIP_CHECK(hipMalloc(&p->d_tabs, sizeof(double) * (4 * 27 * TABW + 32)));
  BEAT_HIP_CHECK(hipMalloc(&p->d_st, sizeof(double) * BEAT_ST_DOUBLES));
  BEAT_HIP_CHECK(hipMalloc(&p->d_alphas, sizeof(double) * PRING_MAX));
  BEAT_HIP_CHECK(hipMemsetAsync(p->d_st, 0, sizeof(double) * BEAT_ST_DOUBLES, ctx->stream));
  const int rc = upload_tables(p);  // Mass / K usable before the first set_timestep
  if (rc) return rc;
  *out = p;
  return BEAT_OK;
}


extern "C" int beat_pde_set_ghost_types(beat_pde* pde, int ghost_lo_type, int ghost_hi_type) {
  BEAT_REQUIRE(pde != nullptr, "null pde");
  BEAT_REQUIRE(ghost_lo_type >= 0 && ghost_lo_type <= 2 && ghost_hi_type >= 0 && ghost_hi_type <= 2, "node types are 0, 1 or 2");
  pde->ghost_lo_tz = ghost_lo_type;
  pde->ghost_hi_tz = ghost_hi_type;
  return BEAT_OK;
}

extern "C" int beat_pde_destroy(beat_pde* pde) {
  if (pde == nullptr) return BEAT_OK;
  // (an open solve -- beat_pde_solve_begin without its end -- has kernels and the copy of its scalar state in the queue: they read and
  // write what is freed below)
  if (pde->open.on && pde->ctx != nullptr) (void)hipStreamSynchronize(pde->ctx->stream);
  (void)hipFree(pde->d_tabs);
  (void)hipFree(pde->d_st);
  (void)hipFree(pde->d_alphas);
  (void)hipFree(pde->d_batch_st);
  (void)hipFree(pde->d_hist_alloc);
  if (pde->h_st) (void)hipHostFree(pde->h_st);
  if (pde->ev_st) (void)hipEventDestroy(pde->ev_st);
  (void)hipFree(pde->v_A);
  (void)hipFree(pde->v_dinv);
  (void)hipFree(pde->v_B);
  (void)hipFree(pde->v_gc0);
  (void)hipFree(pde->v_seg);
  (void)hipFree(pde->v_segmask);
  (void)hipFree(pde->v_seg_tiled);
  (void)hipFree(pde->v_segmask_tiled);
  beat_vrr_destroy(pde);
  beat_vtl_destroy(pde);
  delete pde;
  return BEAT_OK;
}

static int upload_tables(beat_pde* pde) {
  const double C_m = pde->C_m, theta = pde->theta, dt = pde->dt;
  std::vector<double> h(4 * 27 * TABW + 32, 0.0);
  for (int t = 0; t < 27; ++t) {
    for (int k = 0; k < 15; ++k) {
      const double m = pde->h_mass[t * 15 + k], s = pde->h_stiff[t * 15 + k];
      pde->h_A[t * 15 + k] = C_m * m + theta * dt * s;
      pde->h_B[t * 15 + k] = C_m * m - (1.0 - theta) * dt * s;
      h[(0 * 27 + t) * TABW + k] = pde->h_A[t * 15 + k];
      h[(1 * 27 + t) * TABW + k] = pde->h_B[t * 15 + k];
      h[(2 * 27 + t) * TABW + k] = m;
      h[(3 * 27 + t) * TABW + k] = s;
    }
    const double diag = pde->h_A[t * 15];
    pde->h_dinv[t] = diag != 0.0 ? 1.0 / diag : 0.0;
    h[4 * 27 * TABW + t] = pde->h_dinv[t];
  }
  // stream-ordered: later kernels on ctx->stream see the new tables
  BEAT_HIP_CHECK(hipMemcpyAsync(pde->d_tabs, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice,
                                pde->ctx->stream));
  BEAT_HIP_CHECK(hipStreamSynchronize(pde->ctx->stream));  // h goes out of scope
  return BEAT_OK;
}

extern "C" int beat_pde_set_timestep(beat_pde* pde, double C_m, double theta, double dt) {
  BEAT_REQUIRE(pde != nullptr, "null pde");
  BEAT_REQUIRE(!pde->open.on, "the operator has an open solve: finish it first (beat_pde_solve_end)");
  pde->C_m = C_m;
  pde->theta = theta;
  pde->dt = dt;
  pde->have_dt = true;
  pde->last_iters = -1;
  pde->hist_n = 0;  // increments of another time step say nothing about this one
  pde->auto_e_order = 0;
  if (pde->var) return beat_var_form_A(pde);
  return upload_tables(pde);
}

static inline unsigned stencil_grid(const Geom& g) { return (unsigned)(((g.total + 7) / 8) * 8); }

// Geometry of a launch restricted to planes [z_lo, z_hi) of the slab, writing block partials from
// slot part_off on (used to overlap the interior SpMV with the halo exchange).
static Geom range_geom(const Geom& full, int z_lo, int z_hi, int part_off) {
  Geom g = full;
  g.z_lo = z_lo;
  g.z_hi = z_hi;
  const int tiles = g.tiles_x * g.tiles_y;
  const int nzr = std::max(0, z_hi - z_lo);
  int nchunks = std::max(1, (TARGET_BLOCKS + tiles - 1) / tiles);
  nchunks = std::max(1, std::min(nchunks, nzr));
  g.zc = std::max(1, (nzr + nchunks - 1) / nchunks);
  g.nchunks = (nzr + g.zc - 1) / g.zc;
  g.total = tiles * g.nchunks;
  g.part_off = part_off;
  return g;
}

template <int MODE>
static void launch_stencil(const beat_pde* pde, const StencilArgs& a, const Geom* range = nullptr) {
  const Geom& g = range ? *range : pde->g;
  if (g.total <= 0) return;
  const dim3 grid(stencil_grid(g)), block(BEAT_BLOCK);
  hipStream_t s = pde->ctx->stream;
  switch (g.tile_tx) {
    case 256: BEAT_KERNEL((stencil_kernel<MODE, Tile<256, 4>>), grid, block, 0, s, g, a); break;
    case 128: BEAT_KERNEL((stencil_kernel<MODE, Tile<128, 8>>), grid, block, 0, s, g, a); break;
    default: BEAT_KERNEL((stencil_kernel<MODE, Tile<64, 16>>), grid, block, 0, s, g, a); break;
  }
}

extern "C" int beat_pde_apply(beat_pde* pde, int which, const double* dev_x, double* dev_y) {
  BEAT_REQUIRE(pde != nullptr && dev_x && dev_y, "null argument");
  BEAT_REQUIRE(which >= 0 && which < 4, "which must be 0..3");
  BEAT_REQUIRE(which >= 2 || pde->have_dt, "beat_pde_set_timestep has not been called");
  BEAT_REQUIRE(dev_x != dev_y, "in-place apply is not supported");
  if (pde->var) return beat_var_apply(pde, which, dev_x, dev_y);
  const double* host_tab = which == 0 ? pde->h_A : which == 1 ? pde->h_B : which == 2 ? pde->h_mass : pde->h_stiff;
  StencilArgs a{};
  a.x = dev_x;
  a.y = dev_y;
  a.tab = pde->d_tab(which);
  a.ci = interior(host_tab);
  launch_stencil<MODE_APPLY>(pde, a);
  BEAT_LAUNCH_CHECK();
  return BEAT_OK;
}

int beat_pde_launch_reduce(beat_pde* pde, int count, int nsum, double* out, const double* st, double* counter, int then, double* roll_st,
                           double rtol, double atol, int max_it) {
  static const bool fuse = [] {  // BEAT_PCG_FUSE=0: the scalar step in a launch of its own, as before round 5 (A/B runs)
    const char* e = std::getenv("BEAT_PCG_FUSE");
    return !(e && e[0] == '0');
  }();
  BEAT_KERNEL(reduce_partials_kernel, dim3(1), dim3(BEAT_BLOCK), 0, pde->ctx->stream, (const double*)pde->ctx->d_partials, count, nsum, out, st,
              counter, fuse ? then : 0, roll_st, rtol, atol, (double)max_it);
  BEAT_LAUNCH_CHECK();
  if (then != 0 && !fuse) {
    if (then == 1) BEAT_KERNEL(pcg_next_kernel, dim3(1), dim3(1), 0, pde->ctx->stream, roll_st);
    else BEAT_KERNEL(pcg_begin_kernel, dim3(1), dim3(1), 0, pde->ctx->stream, roll_st, rtol, atol, (double)max_it);
    BEAT_LAUNCH_CHECK();
  }
  return BEAT_OK;
}

extern "C" int beat_pde_rhs(beat_pde* pde, const double* dev_v_prev, const double* const* host_dev_stim_w,
                            const double* host_stim_amp, int n_stim, double* dev_x, double* dev_r,
                            double* dev_p, double* dev_red) {
  BEAT_REQUIRE(pde != nullptr && dev_v_prev && dev_x && dev_r && dev_p && dev_red, "null argument");
  BEAT_REQUIRE(pde->have_dt, "beat_pde_set_timestep has not been called");
  BEAT_REQUIRE(n_stim >= 0 && n_stim <= BEAT_MAX_STIM, "at most %d stimuli", BEAT_MAX_STIM);
  BEAT_REQUIRE(dev_r != dev_v_prev && dev_p != dev_v_prev && dev_r != dev_p, "r, p must be distinct work fields");
  BEAT_REQUIRE(!pde->guess_pending, "the previous solve's deferred update has not been applied");
  beat_guess_skip(pde);  // the stage-driven loops start from x0 = v_
  if (pde->var)
    return beat_var_rhs(pde, dev_v_prev, host_dev_stim_w, host_stim_amp, n_stim, dev_x, dev_r, dev_p, dev_red);
  StencilArgs a{};
  a.x = dev_v_prev;
  a.y = dev_r;
  a.y2 = dev_p;
  a.y3 = (dev_x == dev_v_prev) ? nullptr : dev_x;
  a.tab = pde->d_tab(2);
  a.tab2 = pde->d_tab(3);
  a.dinv = pde->d_dinv();
  a.ci = interior(pde->h_mass);
  a.ci2 = interior(pde->h_stiff);
  a.dinv_i = pde->h_dinv[13];
  a.cm = pde->C_m;
  a.omt_dt = (1.0 - pde->theta) * pde->dt;
  a.dt = pde->dt;
  a.nstim = 0;
  for (int k = 0; k < n_stim; ++k) {
    if (host_dev_stim_w[k] == nullptr || host_stim_amp[k] == 0.0) continue;
    a.w[a.nstim] = host_dev_stim_w[k];
    a.amp[a.nstim] = host_stim_amp[k];
    ++a.nstim;
  }
  a.partials = pde->ctx->d_partials;
  launch_stencil<MODE_RHS>(pde, a);
  BEAT_LAUNCH_CHECK();
  return beat_pde_launch_reduce(pde, pde->g.total, 3, dev_red, nullptr);
}

extern "C" int beat_pde_cg_begin(beat_pde* pde, double* dev_st, double rtol, double atol, int max_it) {
  BEAT_REQUIRE(pde != nullptr && dev_st, "null argument");
  BEAT_KERNEL(pcg_begin_kernel, dim3(1), dim3(1), 0, pde->ctx->stream, dev_st, rtol, atol,
                     (double)max_it);
  BEAT_LAUNCH_CHECK();
  return BEAT_OK;
}

extern "C" int beat_pde_spmv_dot(beat_pde* pde, const double* dev_p, double* dev_q, double* dev_st) {
  BEAT_REQUIRE(pde != nullptr && dev_p && dev_q && dev_st, "null argument");
  BEAT_REQUIRE(pde->have_dt, "beat_pde_set_timestep has not been called");
  if (pde->var) return beat_var_spmv_dot(pde, dev_p, dev_q, dev_st);
  StencilArgs a{};
  a.x = dev_p;
  a.y = dev_q;
  a.tab = pde->d_tab(0);
  a.ci = interior(pde->h_A);
  a.partials = pde->ctx->d_partials;
  a.st = dev_st;
  launch_stencil<MODE_SPMV_DOT>(pde, a);
  BEAT_LAUNCH_CHECK();
  return beat_pde_launch_reduce(pde, pde->g.total, 1, dev_st + PQ, dev_st);
}

extern "C" int beat_pde_spmv_dot_part(beat_pde* pde, const double* dev_p, double* dev_q, double* dev_st, int part) {
  BEAT_REQUIRE(pde != nullptr && dev_p && dev_q && dev_st, "null argument");
  BEAT_REQUIRE(pde->have_dt, "beat_pde_set_timestep has not been called");
  BEAT_REQUIRE(part == 0 || part == 1, "part must be 0 (interior) or 1 (boundary planes + reduce)");
  if (pde->var) return beat_var_spmv_dot_part(pde, dev_p, dev_q, dev_st, part);
  const Geom& f = pde->g;
  const int lo = f.z_lo_phys ? 0 : 1, hi = f.nz - (f.z_hi_phys ? 0 : 1);  // planes that need no ghost data
  StencilArgs a{};
  a.x = dev_p;
  a.y = dev_q;
  a.tab = pde->d_tab(0);
  a.ci = interior(pde->h_A);
  a.partials = pde->ctx->d_partials;
  a.st = dev_st;
  const Geom gi = range_geom(f, lo, std::max(lo, hi), 0);
  if (part == 0) {
    launch_stencil<MODE_SPMV_DOT>(pde, a, &gi);
    BEAT_LAUNCH_CHECK();
    return BEAT_OK;
  }
  int off = gi.total;
  if (!f.z_lo_phys) {
    const Geom gb = range_geom(f, 0, 1, off);
    launch_stencil<MODE_SPMV_DOT>(pde, a, &gb);
    off += gb.total;
  }
  if (!f.z_hi_phys && (f.nz > 1 || f.z_lo_phys)) {
    const Geom gb = range_geom(f, f.nz - 1, f.nz, off);
    launch_stencil<MODE_SPMV_DOT>(pde, a, &gb);
    off += gb.total;
  }
  BEAT_LAUNCH_CHECK();
  BEAT_REQUIRE(off <= BEAT_MAX_PARTIALS, "too many block partials");
  return beat_pde_launch_reduce(pde, off, 1, dev_st + PQ, dev_st);
}

extern "C" int beat_pde_cg_update(beat_pde* pde, double* dev_st, double* dev_x, double* dev_r,
                                  const double* dev_p, const double* dev_q) {
  BEAT_REQUIRE(pde != nullptr && dev_st && dev_x && dev_r && dev_p && dev_q, "null argument");
  BEAT_LAUNCH_VEC(pde, cg_update_kernel, dim3(pde->vec_grid), dim3(BEAT_BLOCK), 0, pde->ctx->stream, pde->g,
                     (const double*)dev_st, dev_x, dev_r, dev_p, dev_q, pde->dinv_arg(), pde->h_dinv[13],
                     pde->ctx->d_partials);
  BEAT_LAUNCH_CHECK();
  return beat_pde_launch_reduce(pde, (int)pde->vec_grid, 2, dev_st + RZN, dev_st);
}

extern "C" int beat_pde_set_preconditioner(beat_pde* pde, int ncoef, const double* host_coef) {
  BEAT_REQUIRE(pde != nullptr, "null pde");
  BEAT_REQUIRE(ncoef >= 1 && ncoef <= 8, "polynomial preconditioner needs 1..8 coefficients, got %d", ncoef);
  BEAT_REQUIRE(ncoef == 1 || host_coef != nullptr, "null coefficients");
  BEAT_REQUIRE(ncoef == 1 || !pde->var, "the polynomial preconditioner is not available with per-node coefficients");
  pde->pc_ncoef = ncoef;
  for (int k = 0; k < ncoef; ++k) pde->pc_coef[k] = host_coef ? host_coef[k] : 1.0;
  pde->last_iters = -1;
  return BEAT_OK;
}

extern "C" int beat_pde_pc_num_passes(beat_pde* pde) { return pde ? pde->pc_ncoef - 1 : BEAT_EINVAL; }

// Horner pass j (0 .. ncoef-2) of z = sum_k c_k (D^-1 A)^k D^-1 r.  Pass j reads r (first pass) or the
// previous output and writes q or z alternately such that the LAST pass writes z; the last pass also
// reduces the local r.z into *dev_red.
extern "C" int beat_pde_pc_pass(beat_pde* pde, int j, const double* dev_r, double* dev_z, double* dev_q,
                                double* dev_st, double* dev_red) {
  BEAT_REQUIRE(pde != nullptr && dev_r && dev_z && dev_q && dev_st && dev_red, "null argument");
  const int npass = pde->pc_ncoef - 1;
  BEAT_REQUIRE(j >= 0 && j < npass, "pass %d out of range (have %d)", j, npass);
  BEAT_REQUIRE(pde->have_dt, "beat_pde_set_timestep has not been called");
  // outputs alternate and end in z: pass j writes z if (npass - 1 - j) is even, else q
  auto out_of = [&](int jj) { return ((npass - 1 - jj) % 2 == 0) ? dev_z : dev_q; };
  StencilArgs a{};
  a.x = (j == 0) ? dev_r : out_of(j - 1);
  a.x2 = dev_r;
  a.y = out_of(j);
  a.tab = pde->d_tab(0);
  a.dinv = pde->d_dinv();
  a.dinv_i = pde->h_dinv[13];
  a.ci = interior(pde->h_A);
  a.partials = pde->ctx->d_partials;
  a.st = dev_st;
  a.pc_first = (j == 0);
  a.pc_last = (j == npass - 1);
  a.c_in = pde->pc_coef[npass];          // highest coefficient scales the first staged input
  a.c_r = pde->pc_coef[npass - 1 - j];   // Horner: s_k = c_k D^-1 r + (D^-1 A) s_{k+1}
  launch_stencil<MODE_PC>(pde, a);
  BEAT_LAUNCH_CHECK();
  if (a.pc_last) return beat_pde_launch_reduce(pde, pde->g.total, 1, dev_red, dev_st);
  return BEAT_OK;
}

// p = z + beta p after the scalar roll (polynomial-preconditioned variant of beat_pde_cg_next)
extern "C" int beat_pde_cg_next_z(beat_pde* pde, double* dev_st, const double* dev_z, double* dev_p) {
  BEAT_REQUIRE(pde != nullptr && dev_st && dev_z && dev_p, "null argument");
  BEAT_KERNEL(pcg_next_kernel, dim3(1), dim3(1), 0, pde->ctx->stream, dev_st);
  BEAT_LAUNCH_CHECK();
  const unsigned grid = (unsigned)std::min<int64_t>(2048, (pde->n + BEAT_BLOCK - 1) / BEAT_BLOCK);
  BEAT_KERNEL(cg_pupdate_z_kernel, dim3(grid), dim3(BEAT_BLOCK), 0, pde->ctx->stream, pde->n,
                     (const double*)dev_st, dev_z, dev_p);
  BEAT_LAUNCH_CHECK();
  return BEAT_OK;
}

// p = z (first search direction) without touching the scalars
extern "C" int beat_pde_cg_first_z(beat_pde* pde, double* dev_st, const double* dev_z, double* dev_p) {
  BEAT_REQUIRE(pde != nullptr && dev_st && dev_z && dev_p, "null argument");
  const unsigned grid = (unsigned)std::min<int64_t>(2048, (pde->n + BEAT_BLOCK - 1) / BEAT_BLOCK);
  BEAT_KERNEL(cg_pupdate_z_kernel, dim3(grid), dim3(BEAT_BLOCK), 0, pde->ctx->stream, pde->n,
                     (const double*)dev_st, dev_z, dev_p);
  BEAT_LAUNCH_CHECK();
  return BEAT_OK;
}

extern "C" int beat_pde_cg_next(beat_pde* pde, double* dev_st, const double* dev_r, double* dev_p) {
  BEAT_REQUIRE(pde != nullptr && dev_st && dev_r && dev_p, "null argument");
  BEAT_KERNEL(pcg_next_kernel, dim3(1), dim3(1), 0, pde->ctx->stream, dev_st);
  BEAT_LAUNCH_CHECK();
  BEAT_LAUNCH_VEC(pde, cg_pupdate_kernel, dim3(pde->vec_grid), dim3(BEAT_BLOCK), 0, pde->ctx->stream, pde->g,
                     (const double*)dev_st, dev_r, dev_p, pde->dinv_arg(), pde->h_dinv[13]);
  BEAT_LAUNCH_CHECK();
  return BEAT_OK;
}

// Distance (doubles) between consecutive fields of the work area and of the guess's history: a field with its two ghost planes.
// BEAT_FIELD_SKEW=<doubles> adds a padding (rounded down to 256 B) -- an experiment of round 4: at 512^3 the fields lie a
// multiple of 4 KiB apart like the state rows did before round 3's StateArray padding, and the kernels that stream several of
// them at once might queue on the same memory channels.  Two A/B runs looked like 1 % for 33 x 256 B; a third with eight
// alternating runs on one box showed none (13.67-13.89 ms/step with, 13.60-13.79 without), so the default is no padding.
// Callers size and address the work area through this function either way.
extern "C" int64_t beat_pde_field_stride(const beat_pde* pde) {
  if (pde == nullptr) return 0;
  const int64_t fld = pde->n + 2 * pde->g.plane;
  static const int64_t forced = [] {
    const char* e = std::getenv("BEAT_FIELD_SKEW");
    return e ? (int64_t)std::max(0, std::atoi(e)) / 32 * 32 : (int64_t)0;
  }();
  return fld + forced;
}

extern "C" int beat_pde_work_fields(beat_pde* pde) {
  return pde ? 3 + pde->ring : BEAT_EINVAL;  // r, q, z + the ring of search directions
}

// ---- deferred-x stages for callers that drive the iteration themselves (slab-decomposed solve) ----------
extern "C" int beat_pde_ring_size(void) { return PRING; }

// r -= alpha q, alpha = st[1]/st[3] (also kept for the later x update, slot = iteration % ring size);
// LOCAL r.D^-1 r, r.r -> dev_st[4..5]; counts the executed update in dev_st[14].
extern "C" int beat_pde_cg_update_r(beat_pde* pde, double* dev_st, double* dev_r, const double* dev_q, int slot) {
  BEAT_REQUIRE(pde != nullptr && dev_st && dev_r && dev_q, "null argument");
  BEAT_REQUIRE(slot >= 0 && slot < pde->ring, "slot %d out of range", slot);
  if (pde->var) return beat_var_update_r(pde, dev_st, dev_r, dev_q, slot);
  BEAT_LAUNCH_VEC(pde, cg_update_r_kernel, dim3(pde->vec_grid), dim3(BEAT_BLOCK), 0, pde->ctx->stream, pde->g,
                     (const double*)dev_st, dev_r, dev_q, pde->dinv_arg(), pde->h_dinv[13], pde->ctx->d_partials,
                     pde->d_alphas, slot);
  BEAT_LAUNCH_CHECK();
  return beat_pde_launch_reduce(pde, (int)pde->vec_grid, 2, dev_st + RZN, dev_st, dev_st + NUPD);
}

// scalar roll (beta, latch, iteration count) then p_next = D^-1 r + beta p_cur
extern "C" int beat_pde_cg_next_oop(beat_pde* pde, double* dev_st, const double* dev_r, const double* dev_p_cur,
                                    double* dev_p_next) {
  BEAT_REQUIRE(pde != nullptr && dev_st && dev_r && dev_p_cur && dev_p_next, "null argument");
  BEAT_REQUIRE(dev_p_cur != dev_p_next, "the p-update is out of place");
  BEAT_KERNEL(pcg_next_kernel, dim3(1), dim3(1), 0, pde->ctx->stream, dev_st);
  BEAT_LAUNCH_CHECK();
  if (pde->var) return beat_var_pupdate_oop(pde, dev_st, dev_r, dev_p_cur, dev_p_next);
  BEAT_LAUNCH_VEC(pde, cg_pupdate_oop_kernel, dim3(pde->vec_grid), dim3(BEAT_BLOCK), 0, pde->ctx->stream, pde->g,
                     (const double*)dev_st, dev_r, dev_p_cur, dev_p_next, pde->dinv_arg(), pde->h_dinv[13]);
  BEAT_LAUNCH_CHECK();
  return BEAT_OK;
}

// x += sum_j alpha_j ring_j over the valid directions of the ring cycle starting at iteration ring_base
// (ring_j = dev_ring0 + j*field_stride); with only_if_full it acts only when that cycle filled up.
int beat_pde_x_flush_terms(beat_pde* pde, const double* dev_st, double* dev_x, const double* dev_ring0, int64_t field_stride,
                           int ring_base, int only_if_full, const GuessTerms& gt) {
  if (dev_st == nullptr) dev_st = pde->d_st;  // the scalar state of beat_pde_solve[_ex]
  if (pde->var) return beat_var_flush(pde, dev_st, dev_x, dev_ring0, field_stride, ring_base, only_if_full, gt);
  const unsigned grid = (unsigned)std::min<int64_t>(2048, (pde->n + BEAT_BLOCK - 1) / BEAT_BLOCK);
  BEAT_KERNEL(x_flush_kernel, dim3(grid), dim3(BEAT_BLOCK), 0, pde->ctx->stream, pde->n, dev_st, dev_x,
                     dev_ring0, field_stride, (const double*)pde->d_alphas, ring_base, only_if_full, gt, pde->ring);
  BEAT_LAUNCH_CHECK();
  return BEAT_OK;
}

extern "C" int beat_pde_x_flush(beat_pde* pde, const double* dev_st, double* dev_x, const double* dev_ring0,
                                int64_t field_stride, int ring_base, int only_if_full) {
  BEAT_REQUIRE(pde != nullptr && dev_x && dev_ring0, "null argument");
  BEAT_REQUIRE(!pde->open.on, "the operator has an open solve: finish it first (beat_pde_solve_end)");
  // the application a deferring solve left to its caller carries that solve's guess terms
  GuessTerms gt{};
  if (pde->guess_pending && !only_if_full) {
    gt = pde->guess_final;
    pde->guess_pending = false;
  }
  return beat_pde_x_flush_terms(pde, dev_st, dev_x, dev_ring0, field_stride, ring_base, only_if_full, gt);
}

// 1 when beat_pde_solve[_ex] takes the one-launch path for this operator (and beat_split_steps accepts it)
extern "C" int beat_pde_small_grid_solve_active(const beat_pde* pde) {
  return pde != nullptr && beat_small_available(pde) ? 1 : 0;
}

extern "C" int beat_pde_set_small_grid_solve(beat_pde* pde, int enable) {
  BEAT_REQUIRE(pde != nullptr, "null pde");
  pde->small_enabled = enable != 0;
  return BEAT_OK;
}

// ---- extrapolated initial guess --------------------------------------------------------------------------------
extern "C" int beat_pde_tile_route(const beat_pde* pde) {
  if (pde == nullptr || !pde->var) return 0;
  return (beat_vtl_available(pde) ? 1 : 0) | (beat_vtl_pdot_available(pde) ? 2 : 0) | (beat_vtl_rhs_available(pde) ? 4 : 0) |
         (pde->ring == beat_pde_detail::PRING_MAX ? 8 : 0);
}
extern "C" int beat_pde_fused_dist_pass(const beat_pde* pde) { return (pde != nullptr && pde->v_gc0_valid && pde->v_pdot_dist) ? 1 : 0; }

extern "C" int beat_pde_set_single_reduction(beat_pde* pde, int on) {
  BEAT_REQUIRE(pde != nullptr && on >= -1 && on <= 1, "bad argument");
  pde->single_reduction = on;
  return BEAT_OK;
}

extern "C" int beat_pde_set_guess_order(beat_pde* pde, int order) {
  BEAT_REQUIRE(pde != nullptr, "null pde");
  BEAT_REQUIRE(order >= -1 && order <= BEAT_GUESS_MAX_ORDER, "guess order must be -1 (adaptive) or 0..%d, got %d",
               BEAT_GUESS_MAX_ORDER, order);
  BEAT_REQUIRE(!pde->guess_pending, "a deferred update is pending: apply it before changing the guess order");
  pde->guess_order = order;
  pde->hist_n = 0;
  pde->guess = GuessTerms{};
  pde->auto_cur = pde->auto_next = 3;
  pde->auto_e_order = 0;
  for (int k = 0; k < 4; ++k) pde->auto_seen[k] = 0;
  pde->auto_since_probe = 0;
  if (order < 0) order = BEAT_GUESS_MAX_ORDER;  // adaptive: the fields the cubic needs
  // fields: the max(order - 1, 1) increments kept + the guess (1024^3: 8.6 GB each -- only what the order needs)
  const int need = order > 0 ? std::max(1, order - 1) + 1 : 0;
  if (need > pde->hist_fields) {
    const int64_t fld = beat_pde_field_stride(pde);
    BEAT_HIP_CHECK(hipStreamSynchronize(pde->ctx->stream));
    (void)hipFree(pde->d_hist_alloc);
    pde->d_hist_alloc = nullptr;
    pde->hist_fields = 0;
    pde->d_hist[0] = pde->d_hist[1] = pde->d_hist[2] = pde->d_guess = nullptr;
    if (hipMalloc(&pde->d_hist_alloc, sizeof(double) * need * fld) != hipSuccess) {
      (void)hipGetLastError();
      pde->d_hist_alloc = nullptr;
      pde->guess_order = 0;  // no room for the history: the solves keep starting from x0 = v_
      beat_set_error("no device memory for the %d fields of the initial guess (%.1f GB)", need, 8e-9 * need * fld);
      return BEAT_EHIP;
    }
    BEAT_HIP_CHECK(hipMemsetAsync(pde->d_hist_alloc, 0, sizeof(double) * need * fld, pde->ctx->stream));
    pde->hist_fields = need;
    for (int j = 0; j < need - 1; ++j) pde->d_hist[j] = pde->d_hist_alloc + pde->g.plane + (int64_t)j * fld;
    pde->d_guess = pde->d_hist_alloc + pde->g.plane + (int64_t)(need - 1) * fld;
  }
  return BEAT_OK;
}

extern "C" int beat_pde_guess_reset(beat_pde* pde) {
  BEAT_REQUIRE(pde != nullptr, "null pde");
  BEAT_REQUIRE(!pde->open.on, "the operator has an open solve: finish it first (beat_pde_solve_end)");
  BEAT_REQUIRE(!pde->guess_pending, "a deferred update is pending");
  pde->hist_n = 0;
  pde->auto_e_order = 0;
  return BEAT_OK;
}

// 1 when the last deferring solve left an application to its caller that carries guess terms (it is due even when the
// count of pending search directions is 0)
extern "C" int beat_pde_guess_pending(const beat_pde* pde) { return pde != nullptr && pde->guess_pending ? 1 : 0; }

// device pointers of the last recorded increment d = x - v_ and of the guess increment e prepared for the next solve,
// and the number of solves recorded since the history was dropped (0: the next solve starts from x0 = v_)
extern "C" int beat_pde_guess_history(const beat_pde* pde, double** dev_d, double** dev_e, int* count) {
  BEAT_REQUIRE(pde != nullptr, "null pde");
  if (dev_d) *dev_d = pde->d_hist[0];
  if (dev_e) *dev_e = pde->d_guess;
  if (count) *count = pde->hist_n;
  return BEAT_OK;
}

extern "C" int beat_pde_guess_traffic(const beat_pde* pde, int* host_out) {
  BEAT_REQUIRE(pde != nullptr && host_out != nullptr, "null argument");
  // (an update applied by a launch enqueued behind an open solve: its terms were kept when that solve was finished)
  const GuessTerms& g = pde->guess_pending ? pde->guess_final : (pde->applied_behind ? pde->applied_terms : pde->guess);
  int reads = 0, writes = 0;
  if (g.d != nullptr) {
    reads += (g.accumulate || g.use_e) ? 1 : 0;          // e
    reads += (g.accumulate || g.cd != 0.0) ? 1 : 0;      // the oldest increment kept
    for (int j = 0; j < BEAT_GUESS_MAX_ORDER - 2; ++j) reads += (!g.accumulate && g.cp[j] != 0.0) ? 1 : 0;
    writes = 2;                                          // d, e
  }
  host_out[0] = reads;
  host_out[1] = writes;
  host_out[2] = pde->guess_order < 0 ? pde->auto_cur : pde->guess_order;
  host_out[3] = pde->guess_pending ? 1 : (pde->applied_behind ? 2 : 0);  // 2: applied by the launch behind the open solve
  return BEAT_OK;
}

void beat_guess_skip(beat_pde* pde) {
  pde->hist_n = 0;
  pde->auto_e_order = 0;
  pde->guess = GuessTerms{};
}

static inline int guess_max_order(const beat_pde* pde) { return pde->guess_order < 0 ? BEAT_GUESS_MAX_ORDER : pde->guess_order; }

void beat_guess_begin(beat_pde* pde) {
  pde->guess = GuessTerms{};
  if (pde->guess_order == 0 || pde->d_hist[0] == nullptr) return;
  GuessTerms& g = pde->guess;
  // increments kept: order - 1 (at least one), newest first in d_hist; the oldest one's storage takes this solve's
  const int nb = std::max(1, guess_max_order(pde) - 1);
  g.d = pde->d_hist[nb - 1];
  for (int j = 0; j + 1 < nb; ++j) g.dp[j] = pde->d_hist[j];
  g.e = pde->d_guess;
  g.use_e = pde->hist_n >= 1;
  // the guess after this solve extrapolates through the m increments then on record (this one included):
  // e = sum_{i=0}^{m-1} (-1)^i C(m, i+1) D_i,  D_0 = this solve's, D_i = d_hist[i-1] as it is now
  const int want = pde->guess_order < 0 ? pde->auto_next : pde->guess_order;
  const int m = std::min(want, pde->hist_n + 1);
  static const double binom[5][5] = {{1, 0, 0, 0, 0}, {1, 1, 0, 0, 0}, {1, 2, 1, 0, 0}, {1, 3, 3, 1, 0}, {1, 4, 6, 4, 1}};
  g.a = binom[m][1];
  for (int i = 1; i < m; ++i) {
    const double c = ((i & 1) ? -1.0 : 1.0) * binom[m][i + 1];
    if (i - 1 == nb - 1)
      g.cd = c;
    else
      g.cp[i - 1] = c;
  }
}

// Terms of an x update for the ring cycle starting at iteration ring_base: the first cycle carries e and records the
// increment, later ones add to it.
GuessTerms beat_guess_terms(const beat_pde* pde, int ring_base) {
  GuessTerms g = pde->guess;
  if (g.d != nullptr && ring_base > 0) g.accumulate = 1;
  return g;
}

void beat_guess_advance(beat_pde* pde) {
  const int nb = std::max(1, guess_max_order(pde) - 1);
  double* newest = pde->d_hist[nb - 1];
  for (int j = nb - 1; j > 0; --j) pde->d_hist[j] = pde->d_hist[j - 1];
  pde->d_hist[0] = newest;
  pde->hist_n = std::min(BEAT_GUESS_MAX_ORDER, pde->hist_n + 1);
}

// Adaptive order: called by the solve paths once the host has the scalar state of the solve that just ended, before
// beat_guess_end / beat_guess_advance.  Scores the order the guess was built with by what it is for -- the
// iterations the solve took (the norm of the initial residual is a poor judge: the cubic's is smaller even where it
// costs more iterations, because what is left is the amplified noise of the recorded increments, rough, and Jacobi-PCG
// takes longer over it than over the smooth truncation error of the quadratic) -- and picks the order of the guess
// after next (the next one is being prepared by this solve's x update, whose coefficients were fixed when it began).
// beat_guess_policy is the move itself (hill climbing over the orders 1..4, see beat_pde_internal.h).
int beat_guess_policy(beat_pde* pde) {
  constexpr int LO = 1, HI = BEAT_GUESS_MAX_ORDER;
  int& cur = pde->auto_cur;
  // a neighbour that has been looked at and costs fewer iterations takes over (ties stay)
  for (int nb = cur - 1; nb <= cur + 1; nb += 2) {
    if (nb < LO || nb > HI || !pde->auto_seen[nb - 1] || !pde->auto_seen[cur - 1]) continue;
    if (pde->auto_score[nb - 1] < pde->auto_score[cur - 1] - 0.05) {
      cur = nb;
      pde->auto_since_probe = 0;  // look around from the new position soon
      break;
    }
  }
  int next = cur;
  if (pde->auto_seen[cur - 1] && ++pde->auto_since_probe >= 12) {
    pde->auto_since_probe = 0;
    int nb = cur + (pde->auto_probe_up ? 1 : -1);
    if (nb < LO || nb > HI) nb = cur - (pde->auto_probe_up ? 1 : -1);
    pde->auto_probe_up = !pde->auto_probe_up;
    if (nb >= LO && nb <= HI) next = nb;
  }
  return next;
}

void beat_guess_observe(beat_pde* pde, int iterations) {
  if (pde->guess_order >= 0) return;
  const int used = pde->auto_e_order;  // order behind the e this solve started from (0: none yet, or fewer increments)
  if (used >= 1) {
    const int k = used - 1;
    pde->auto_score[k] = pde->auto_seen[k] ? 0.5 * pde->auto_score[k] + 0.5 * iterations : (double)iterations;
    pde->auto_seen[k] = 1;
  }
  // the e the x update of THIS solve prepares has order min(auto_next, increments on record): remember it for the
  // next observation, then choose for the one after
  const int prepared = std::min(pde->auto_next, pde->hist_n + 1);
  pde->auto_e_order = prepared == pde->auto_next ? prepared : 0;
  pde->auto_next = beat_guess_policy(pde);
}

bool beat_guess_end(beat_pde* pde, int nupd, bool deferred) {
  const int PR = pde->ring;
  const bool partial = nupd % PR != 0;
  if (pde->guess.d == nullptr) return partial;
  const bool e_due = nupd == 0 && pde->guess.use_e;  // no ring cycle carried e to x yet
  if (nupd == 0 && !e_due) {  // x = v_ is the answer and nothing was recorded: the history ends here
    beat_guess_skip(pde);
    return false;
  }
  const bool due = partial || e_due;
  if (due && deferred) {
    pde->guess_final = beat_guess_terms(pde, (nupd / PR) * PR);
    pde->guess_pending = true;
  }
  beat_guess_advance(pde);
  return due;
}

extern "C" int beat_pde_solve(beat_pde* pde, const double* dev_v_prev,
                              const double* const* host_dev_stim_w, const double* host_stim_amp,
                              int n_stim, double* dev_x, double* dev_work, double rtol, double atol,
                              int max_it, beat_ksp_info* info) {
  return beat_pde_solve_ex(pde, dev_v_prev, host_dev_stim_w, host_stim_amp, n_stim, dev_x, dev_work, rtol, atol, max_it,
                           0, info, nullptr);
}

bool beat_solve_lazy_available(const beat_pde* pde) {
  // Jacobi on one slab, not the one-launch path of small grids: the two loops that keep every scalar on the device
  return pde != nullptr && pde->g.z_lo_phys && pde->g.z_hi_phys && !beat_small_available(pde) && pde->pc_ncoef == 1 &&
         (beat_rr_available(pde) || pde->var);
}

// `count` more iterations of the open solve, enqueued
static int solve_enqueue_iterations(beat_pde* pde, int count) {
  beat_pde::OpenSolve& o = pde->open;
  const int64_t fld = beat_pde_field_stride(pde);
  double* r = o.work + pde->g.plane;
  double* q = r + fld;
  double* ring = q + 2 * fld;
  double* st = pde->d_st;
  const int PR = pde->ring;
  int rc;
  for (int it = 0; it < count; ++it) {
    const int i = o.launched + it, slot = i % PR;
    double* p_cur = ring + (int64_t)slot * fld;
    const double* p_old = ring + (int64_t)((i + PR - 1) % PR) * fld;
    if (o.kind == 0) {
      // iteration i: p_i = D^-1 r + beta p_{i-1} and p_i . A p_i in one pass (ring slot i % PR), then r_new = r - alpha A p_i
      // with A p_i recomputed (written to the other of the two residual buffers: the kernel then needs no store-before-load
      // ordering), then the scalar roll
      double* rbuf[2] = {r, q};
      if ((rc = beat_rr_pdot(pde, st, rbuf[i & 1], p_old, p_cur))) return rc;
      if ((rc = beat_rr_rupd(pde, st, rbuf[i & 1], rbuf[(i + 1) & 1], p_cur, slot))) return rc;
      if (slot == PR - 1) {  // ring full: bring x up to date before slot 0 is overwritten
        if ((rc = beat_pde_x_flush_terms(pde, st, o.x, ring, fld, i + 1 - PR, 1, beat_guess_terms(pde, i + 1 - PR)))) return rc;
      }
    } else {
      // per-node rows (round 4): the tile kernel forms p_i = D^-1 r + beta p_{i-1} while loading, stores it and q = A p_i in the
      // same pass (beat_vtl_pdot); same expressions, same bits as the three-kernel iteration (BEAT_VTL_PDOT=0)
      double* p_next = ring + (int64_t)((i + 1) % PR) * fld;
      if (o.pdot) {
        if ((rc = beat_vtl_pdot(pde, st, r, p_old, p_cur, q, i == 0))) return rc;
      } else if ((rc = beat_pde_spmv_dot(pde, p_cur, q, st))) {
        return rc;
      }
      // (with the fused pass the scalar roll -- beta for the next pass, the latch, the iteration count -- runs in the launch that sums
      // the residual update's partials; the x update, when the ring is full, reads the update count only and may follow it)
      if ((rc = o.pdot ? beat_var_update_r(pde, st, r, q, slot, true) : beat_pde_cg_update_r(pde, st, r, q, slot))) return rc;
      if (slot == PR - 1) {
        if ((rc = beat_pde_x_flush_terms(pde, st, o.x, ring, fld, i + 1 - PR, 1, beat_guess_terms(pde, i + 1 - PR)))) return rc;
      }
      if (!o.pdot && (rc = beat_pde_cg_next_oop(pde, st, r, p_cur, p_next))) return rc;
    }
  }
  o.launched += count;
  return BEAT_OK;
}

int beat_solve_begin(beat_pde* pde, const double* dev_v_prev, const double* const* host_dev_stim_w, const double* host_stim_amp, int n_stim,
                     double* dev_x, double* dev_work, double rtol, double atol, int max_it) {
  BEAT_REQUIRE(pde != nullptr && dev_work != nullptr && dev_v_prev && dev_x, "null argument");
  BEAT_REQUIRE(beat_solve_lazy_available(pde), "this operator's solves are not of the kind that can be left open");
  BEAT_REQUIRE(!pde->open.on, "the previous solve has not been finished (beat_pde_solve_end)");
  BEAT_REQUIRE(max_it >= 0, "max_it must be >= 0");
  BEAT_REQUIRE(pde->have_dt, "beat_pde_set_timestep has not been called");
  BEAT_REQUIRE(n_stim >= 0 && n_stim <= BEAT_MAX_STIM, "at most %d stimuli", BEAT_MAX_STIM);
  BEAT_REQUIRE(!pde->guess_pending, "the previous solve's deferred update has not been applied");
  beat_ctx* ctx = pde->ctx;
  if (pde->h_st == nullptr) {
    BEAT_HIP_CHECK(hipHostMalloc((void**)&pde->h_st, sizeof(double) * 16, hipHostMallocDefault));
    BEAT_HIP_CHECK(hipEventCreateWithFlags(&pde->ev_st, hipEventDisableTiming));
  }
  beat_pde::OpenSolve& o = pde->open;
  o = beat_pde::OpenSolve{};
  o.kind = beat_rr_available(pde) ? 0 : 1;
  o.pdot = o.kind == 1 && beat_vtl_pdot_available(pde);
  o.v_prev = dev_v_prev;
  o.x = dev_x;
  o.work = dev_work;
  o.rtol = rtol;
  o.atol = atol;
  o.max_it = max_it;
  const int64_t fld = beat_pde_field_stride(pde);
  double* r = dev_work + pde->g.plane;
  double* q = r + fld;
  double* ring = q + 2 * fld;
  double* st = pde->d_st;
  int rc;
  beat_guess_begin(pde);
  pde->fuse_begin = beat_pde::FuseBegin{true, false, rtol, atol, max_it};
  if (o.kind == 0) {
    rc = beat_rr_rhs(pde, dev_v_prev, host_dev_stim_w, host_stim_amp, n_stim, dev_x, r, st);
  } else if (beat_vtl_rhs_available(pde)) {  // two tile passes (b = B v_ + dt stim, r = b - A (v_ + e)); q is free until iteration 0
    rc = beat_vtl_rhs(pde, dev_v_prev, host_dev_stim_w, host_stim_amp, n_stim, dev_x, r, ring, q, st, pde->guess.use_e ? pde->guess.e : nullptr);
  } else {  // the gather kernel: the guess increment e next to v_
    rc = beat_var_rhs(pde, dev_v_prev, host_dev_stim_w, host_stim_amp, n_stim, dev_x, r, ring, st, pde->guess.use_e ? pde->guess.e : nullptr);
  }
  const bool begun = pde->fuse_begin.done;  // (the right-hand side's reduction has run the start of the solve in its own launch)
  pde->fuse_begin.on = false;
  if (rc) return rc;
  if (!begun && (rc = beat_pde_cg_begin(pde, st, rtol, atol, max_it))) return rc;
  if ((rc = solve_enqueue_iterations(pde, std::min(beat_pde_first_chunk(pde), max_it)))) return rc;
  BEAT_HIP_CHECK(hipMemcpyAsync(pde->h_st, st, sizeof(double) * 16, hipMemcpyDeviceToHost, ctx->stream));
  BEAT_HIP_CHECK(hipEventRecord(pde->ev_st, ctx->stream));
  o.on = true;
  return BEAT_OK;
}

int beat_solve_end(beat_pde* pde, int defer_flush, beat_ksp_info* info, int* host_pending, bool* needed_more) {
  BEAT_REQUIRE(pde != nullptr, "null pde");
  if (host_pending) host_pending[0] = host_pending[1] = 0;
  if (needed_more) *needed_more = false;
  if (!pde->open.on) {  // nothing open: the record of the last solve that was finished
    if (info) *info = pde->last_info;
    return pde->last_rc;
  }
  BEAT_REQUIRE(!defer_flush || host_pending != nullptr, "defer_flush needs host_pending[2]");
  if (pde->open.comm != nullptr) return beat_dist_solve_end(pde, defer_flush, info, host_pending, needed_more);  // a decomposed solve
  beat_pde::OpenSolve& o = pde->open;
  beat_ctx* ctx = pde->ctx;
  double* h = pde->h_st;
  double* st = pde->d_st;
  const int64_t fld = beat_pde_field_stride(pde);
  double* ring = o.work + pde->g.plane + 3 * fld;
  const int PR = pde->ring;
  int rc;
  BEAT_HIP_CHECK(hipEventSynchronize(pde->ev_st));
  // (a launch enqueued behind this solve saw the state the host sees at this first look: unlatched, it did nothing -- also when no
  // further iteration may be enqueued (max_it = 0: pcg_begin does not latch and the loop below is skipped))
  if (needed_more && h[STOP] == 0.0) *needed_more = true;
  while (!(h[STOP] != 0.0 || o.launched >= o.max_it)) {
    if ((rc = solve_enqueue_iterations(pde, std::min(2, o.max_it - o.launched)))) {
      o.on = false;
      return rc;
    }
    BEAT_HIP_CHECK(hipMemcpyAsync(h, st, sizeof(double) * 16, hipMemcpyDeviceToHost, ctx->stream));
    BEAT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  }
  o.on = false;
  pde->applied_behind = false;  // (step_behind_open_solve sets it again when its launch has applied this solve's update)
  // directions of the last, partially filled ring cycle (stream-ordered before anything that reads x) and / or the guess increment
  const int nupd = (int)h[NUPD], base = (nupd / PR) * PR;
  const GuessTerms last = beat_guess_terms(pde, base);
  beat_guess_observe(pde, (int)h[ITERS]);
  if (beat_guess_end(pde, nupd, defer_flush != 0)) {
    if (defer_flush) {  // the caller adds these directions itself (beat_ode_step_pending / beat_pde_x_flush)
      host_pending[0] = pde->last_base = base;
      host_pending[1] = nupd % PR;
    } else if ((rc = beat_pde_x_flush_terms(pde, st, o.x, ring, fld, base, 0, last))) {
      return rc;
    }
  }
  const int iters = (int)h[ITERS];
  pde->last_iters = iters;
  int reason = (int)h[REASON];
  if (h[STOP] == 0.0) reason = -3;  // max_it == launched without the latch (max_it = 0)
  pde->last_info.iterations = iters;
  pde->last_info.converged_reason = reason;
  pde->last_info.residual_norm = std::sqrt(h[RR]);
  pde->last_info.rhs_norm = std::sqrt(h[BB]);
  if (info) *info = pde->last_info;
  pde->last_rc = BEAT_OK;
  if (reason < 0) {
    beat_set_error("PCG did not converge in %d iterations (||r|| = %.3e, ||b|| = %.3e)", iters, std::sqrt(h[RR]), std::sqrt(h[BB]));
    pde->last_rc = BEAT_ENOTCONV;
  }
  return pde->last_rc;
}

extern "C" int beat_pde_solve_begin(beat_pde* pde, const double* dev_v_prev, const double* const* host_dev_stim_w,
                                    const double* host_stim_amp, int n_stim, double* dev_x, double* dev_work, double rtol, double atol,
                                    int max_it) {
  return beat_solve_begin(pde, dev_v_prev, host_dev_stim_w, host_stim_amp, n_stim, dev_x, dev_work, rtol, atol, max_it);
}
extern "C" int beat_pde_solve_end(beat_pde* pde, beat_ksp_info* info, int* host_pending) {
  BEAT_REQUIRE(host_pending != nullptr, "host_pending[2] expected");
  return beat_solve_end(pde, 1, info, host_pending, nullptr);
}
extern "C" int beat_pde_solve_is_open(const beat_pde* pde) { return pde != nullptr && pde->open.on ? 1 : 0; }
extern "C" int beat_pde_solve_can_open(const beat_pde* pde) { return beat_solve_lazy_available(pde) ? 1 : 0; }

extern "C" int beat_pde_solve_ex(beat_pde* pde, const double* dev_v_prev,
                                 const double* const* host_dev_stim_w, const double* host_stim_amp,
                                 int n_stim, double* dev_x, double* dev_work, double rtol, double atol,
                                 int max_it, int defer_flush, beat_ksp_info* info, int* host_pending) {
  BEAT_REQUIRE(!defer_flush || host_pending != nullptr, "defer_flush needs host_pending[2]");
  if (host_pending) host_pending[0] = host_pending[1] = 0;
  BEAT_REQUIRE(pde != nullptr && dev_work != nullptr, "null argument");
  BEAT_REQUIRE(pde->g.z_lo_phys && pde->g.z_hi_phys, "beat_pde_solve is the single-slab path");
  BEAT_REQUIRE(max_it >= 0, "max_it must be >= 0");
  BEAT_REQUIRE(!pde->open.on, "the previous solve has not been finished (beat_pde_solve_end)");
  if (beat_small_available(pde)) {  // a few thousand nodes: the whole solve in one launch of one workgroup
    BEAT_REQUIRE(dev_v_prev && dev_x, "null argument");
    return beat_small_solve(pde, dev_v_prev, host_dev_stim_w, host_stim_amp, n_stim, dev_x, rtol, atol, max_it, info);
  }
  if (beat_solve_lazy_available(pde)) {
    // Jacobi with the register-row kernels (constant coefficients: the loop that never stores q = A p) or on per-node rows:
    // enqueue, then wait -- the two halves a caller may also drive apart (beat_pde_solve_begin / _end)
    if (int rc = beat_solve_begin(pde, dev_v_prev, host_dev_stim_w, host_stim_amp, n_stim, dev_x, dev_work, rtol, atol, max_it)) return rc;
    return beat_solve_end(pde, defer_flush, info, host_pending, nullptr);
  }
  const int64_t fld = beat_pde_field_stride(pde);
  double* r = dev_work + pde->g.plane;
  double* q = r + fld;
  double* z = q + fld;     // only touched by the polynomial preconditioner
  double* ring = z + fld;  // pde->ring search directions, ring[j] = ring + j*fld
  const int PR = pde->ring;
  double* st = pde->d_st;
  beat_ctx* ctx = pde->ctx;
  double* h = ctx->h_pinned;
  const int npass = pde->pc_ncoef - 1;
  int rc;
  if ((rc = beat_pde_rhs(pde, dev_v_prev, host_dev_stim_w, host_stim_amp, n_stim, dev_x, r, ring, st))) return rc;
  if ((rc = beat_pde_cg_begin(pde, st, rtol, atol, max_it))) return rc;
  int launched = 0;
  int chunk = beat_pde_first_chunk(pde);
  if (npass > 0) {
    // polynomial preconditioner: classic in-place recurrences with p = ring[0]
    double* p = ring;
    for (int j = 0; j < npass; ++j)
      if ((rc = beat_pde_pc_pass(pde, j, r, z, q, st, st + RZ))) return rc;
    if ((rc = beat_pde_cg_first_z(pde, st, z, p))) return rc;
    while (true) {
      chunk = std::min(chunk, max_it - launched);
      for (int it = 0; it < chunk; ++it) {
        if ((rc = beat_pde_spmv_dot(pde, p, q, st))) return rc;
        if ((rc = beat_pde_cg_update(pde, st, dev_x, r, p, q))) return rc;
        for (int j = 0; j < npass; ++j)
          if ((rc = beat_pde_pc_pass(pde, j, r, z, q, st, st + RZN))) return rc;
        if ((rc = beat_pde_cg_next_z(pde, st, z, p))) return rc;
      }
      launched += chunk;
      BEAT_HIP_CHECK(hipMemcpyAsync(h, st, sizeof(double) * 16, hipMemcpyDeviceToHost, ctx->stream));
      BEAT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
      if (h[STOP] != 0.0 || launched >= max_it) break;
      chunk = 2;
    }
  } else {
    // Jacobi on the LDS-tiled constant-coefficient kernels (BEAT_RR=0), deferred x: iteration i uses p_i = ring[i % PR]
    while (true) {
      chunk = std::min(chunk, max_it - launched);
      for (int it = 0; it < chunk; ++it) {
        const int i = launched + it, slot = i % PR;
        double* p_cur = ring + (int64_t)slot * fld;
        double* p_next = ring + (int64_t)((i + 1) % PR) * fld;
        if ((rc = beat_pde_spmv_dot(pde, p_cur, q, st))) return rc;
        if ((rc = beat_pde_cg_update_r(pde, st, r, q, slot))) return rc;
        if (slot == PR - 1) {  // ring full: bring x up to date before slot 0 is overwritten
          if ((rc = beat_pde_x_flush_terms(pde, st, dev_x, ring, fld, i + 1 - PR, 1, beat_guess_terms(pde, i + 1 - PR))))
            return rc;
        }
        if ((rc = beat_pde_cg_next_oop(pde, st, r, p_cur, p_next))) return rc;
      }
      launched += chunk;
      BEAT_HIP_CHECK(hipMemcpyAsync(h, st, sizeof(double) * 16, hipMemcpyDeviceToHost, ctx->stream));
      BEAT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
      if (h[STOP] != 0.0 || launched >= max_it) break;
      chunk = 2;
    }
    // directions of the last, partially filled ring cycle (stream-ordered before anything that reads x)
    const int nupd = (int)h[NUPD], base = (nupd / PR) * PR;
    const GuessTerms last = beat_guess_terms(pde, base);
    beat_guess_observe(pde, (int)h[ITERS]);
    if (beat_guess_end(pde, nupd, defer_flush != 0)) {
      if (defer_flush) {  // the caller adds these directions itself (beat_ode_step_pending / beat_pde_x_flush)
        host_pending[0] = pde->last_base = base;
        host_pending[1] = nupd % PR;
      } else if ((rc = beat_pde_x_flush_terms(pde, st, dev_x, ring, fld, base, 0, last))) {
        return rc;
      }
    }
  }
  const int iters = (int)h[ITERS];
  pde->last_iters = iters;
  int reason = (int)h[REASON];
  if (h[STOP] == 0.0) reason = -3;  // max_it == launched without the latch (max_it = 0)
  pde->last_info.iterations = iters;
  pde->last_info.converged_reason = reason;
  pde->last_info.residual_norm = std::sqrt(h[RR]);
  pde->last_info.rhs_norm = std::sqrt(h[BB]);
  if (info) *info = pde->last_info;
  pde->last_rc = BEAT_OK;
  if (reason < 0) {
    beat_set_error("PCG did not converge in %d iterations (||r|| = %.3e, ||b|| = %.3e)", iters,
                   std::sqrt(h[RR]), std::sqrt(h[BB]));
    pde->last_rc = BEAT_ENOTCONV;
  }
  return pde->last_rc;
}
