// Diffusion step on a structured z-slab: matrix-free P1 operators as a 15-point stencil,
// right-hand-side build and Jacobi-PCG.  Replaces dolfinx assemble_vector + PETSc KSP.solve of
// src/beat/base_model.py:196-236 (forms: src/beat/monodomain_model.py:68-98).
//
// Data layout: a field is nx*ny*nz_local doubles, x fastest, with one ghost xy-plane addressable
// on either side.  The 27x15 coefficient tables (one row per boundary type of a node) come from
// the caller, derived by element assembly; the interior row (type 13) is passed by value so it
// lives in SGPRs, the 26 boundary rows are looked up from a small device table by the few lanes
// that need them.
//
// Stencil kernel structure (gfx950): a 256-thread workgroup owns a 64(x) x 16(y) tile and marches
// along z through a chunk of planes.  Three (TY+2)x(TX+2) planes live in a ring of LDS slots; the
// next plane is prefetched into registers while the current one is computed (global loads are
// row-contiguous, 8 B/lane).  Every output needs 15 LDS reads (ds_read_b64, conflict-free: lanes
// read consecutive doubles); a thread computes 4 rows so the compiler shares the in-plane reads.
// Algorithmic HBM traffic: 8 B read + 8 B written per node per operator application; the tile
// halo (+16%) and the chunk's two extra planes are re-reads that the XCD-local L2 mostly absorbs
// (tiles are dealt to XCDs in contiguous runs, see tile_of_block()).
#include "beat_common.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {

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
constexpr int TABW = 16;  // padded row width of the device coefficient tables

// slots of the PCG scalar state `st` (device, caller-owned, >= 16 doubles)
enum St { BB = 0, RZ, RR, PQ, RZN, RRN, TOL2, BETA, STOP, ITERS, REASON, RTOL, ATOL, MAXIT, NUPD };
constexpr int PRING = 6;  // search directions kept by the deferred-x PCG before x must be brought up to date

const int kOffsets[45] = {0, 0, 0,  1, 0, 0,  -1, 0, 0,  0, 1, 0,  0, -1, 0,  0, 0, 1,  0, 0, -1,
                          1, 1, 0,  -1, -1, 0,  0, 1, 1,  0, -1, -1,  1, 0, 1,  -1, 0, -1,
                          1, 1, 1,  -1, -1, -1};

struct Coef15 {
  double c[15];
};

struct Geom {
  int nx, ny, nz;
  int64_t plane;
  int tiles_x, tiles_y, nchunks, zc, total;
  int z_lo_phys, z_hi_phys;
  int tile_tx;  // 64, 128 or 256: which Tile<> instantiation the grid was sized for
  int z_lo, z_hi;    // planes [z_lo, z_hi) computed by this launch (whole slab: 0, nz)
  int part_off;      // first block-partial slot this launch writes
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
enum { MODE_APPLY = 0, MODE_SPMV_DOT = 1, MODE_RHS = 2, MODE_PC = 3 };

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
__global__ __launch_bounds__(BEAT_BLOCK) void reduce_partials_kernel(const double* __restrict__ partials,
                                                                     int count, int nsum,
                                                                     double* __restrict__ out,
                                                                     const double* __restrict__ st,
                                                                     double* __restrict__ counter) {
  __shared__ double red[4];
  if (st != nullptr && st[STOP] != 0.0) return;
  if (counter != nullptr && threadIdx.x == 0) counter[0] += 1.0;  // one more executed residual update
  for (int k = 0; k < nsum; ++k) {
    double s = 0.0;
    for (int i = threadIdx.x; i < count; i += BEAT_BLOCK) s += partials[(int64_t)k * BEAT_MAX_PARTIALS + i];
    s = beat_block_sum(s, red);
    if (threadIdx.x == 0) out[k] = s;
  }
}

__global__ void pcg_begin_kernel(double* st, double rtol, double atol, double max_it) {
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
  const bool done = rr <= tol2;
  st[STOP] = done ? 1.0 : 0.0;
  st[REASON] = done ? (rr <= tr ? 2.0 : 3.0) : 0.0;
}

__global__ void pcg_next_kernel(double* st) {
  if (st[STOP] != 0.0) return;
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
__global__ __launch_bounds__(BEAT_BLOCK) void x_flush_kernel(int64_t n, const double* __restrict__ st,
                                                             double* __restrict__ x,
                                                             const double* __restrict__ ring, int64_t fld,
                                                             const double* __restrict__ alphas, int ring_base,
                                                             int only_if_full) {
  int nvalid = (int)st[NUPD] - ring_base;
  nvalid = nvalid < 0 ? 0 : (nvalid > PRING ? PRING : nvalid);
  // in-loop flushes are enqueued ahead of time: they must do nothing unless their ring cycle really
  // filled up (a partially filled last cycle is flushed once, after the host has seen the latch)
  if (nvalid == 0 || (only_if_full && nvalid < PRING)) return;
  double a[PRING];
#pragma unroll
  for (int j = 0; j < PRING; ++j) a[j] = (j < nvalid) ? alphas[j] : 0.0;
  const int64_t stride = (int64_t)gridDim.x * BEAT_BLOCK;
  for (int64_t i = (int64_t)blockIdx.x * BEAT_BLOCK + threadIdx.x; i < n; i += stride) {
    double xi = x[i];
#pragma unroll
    for (int j = 0; j < PRING; ++j)
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


// ---- variable-coefficient operators (beat_pde_create_var) -----------------------------------------------
// Voxel-masked domains and spatially varying conductivity: every node carries its own 15 coefficients per
// operator, stored coefficient-major ((15, ld) arrays, so a wave reads 15 contiguous 512 B segments).  The
// coefficient streams are 120 of the 136 B/node an application moves, so the neighbour values are simply
// gathered through L1/L2 (rows of x are contiguous across the wave) instead of being staged through LDS.
// A neighbour is only read where its coefficient is non-zero: rows never reach outside the box (or into an
// inactive voxel), so no out-of-range address is formed and stale ghost planes cannot leak NaNs.
struct VarArgs {
  const double* T1;  // (15, ld) coefficients
  const double* T2;  // second operator (RHS: stiffness; APPLY: optional) or nullptr
  double c1, c2;     // APPLY: y = (c1 T1 + c2 T2) x
  int64_t ld;
  const double* x;
  double* y;         // APPLY: y | SPMV: q | RHS: r
  double* y2;        // RHS: p
  double* y3;        // RHS: x (copy of v_) or nullptr
  const double* dinv;
  double dt;
  const double* w[BEAT_MAX_STIM];
  double amp[BEAT_MAX_STIM];
  int nstim;
  double* partials;
  int part_off;
  const double* st;
  int64_t i_lo, i_hi;  // node range of this launch
  int doff[15];        // linear offsets of the 15 stencil points
  const int* seg;      // active 256-node segments covering [i_lo, i_hi) (nullptr: every node of the range)
  int nseg;
  const double* mdiag; // RHS: mass diagonal (0 = node outside the tissue)
};

template <int MODE>
__global__ __launch_bounds__(BEAT_BLOCK) void var_stencil_kernel(VarArgs a) {
  __shared__ double red[4];
  if (MODE == MODE_SPMV_DOT) {
    if (a.st[STOP] != 0.0) return;
  }
  double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0;
  // Work items: the 256-node segments that hold at least one tissue node (list built at create time); segments
  // entirely outside the tissue are never read or written (their r, p, q stay zero, x keeps its value).
  const int64_t nwork = a.seg ? a.nseg : (a.i_hi - a.i_lo + BEAT_BLOCK - 1) / BEAT_BLOCK;
  for (int64_t w = blockIdx.x; w < nwork; w += gridDim.x) {
    const int64_t i = (a.seg ? (int64_t)a.seg[w] * BEAT_BLOCK : a.i_lo + w * BEAT_BLOCK) + threadIdx.x;
    if (i < a.i_lo || i >= a.i_hi) continue;
    double s1 = 0.0, s2 = 0.0;
    double xc = 0.0;
#pragma unroll
    for (int k = 0; k < 15; ++k) {
      const double c1 = a.T1[(int64_t)k * a.ld + i];
      double c2 = 0.0;
      if (MODE == MODE_RHS || (MODE == MODE_APPLY && a.T2 != nullptr)) c2 = a.T2[(int64_t)k * a.ld + i];
      const bool need = (k == 0) || c1 != 0.0 || c2 != 0.0;
      const double xv = need ? a.x[i + a.doff[k]] : 0.0;
      if (k == 0) xc = xv;
      s1 = fma(c1, xv, s1);
      s2 = fma(c2, xv, s2);
    }
    if (MODE == MODE_APPLY) {
      a.y[i] = a.c1 * s1 + a.c2 * s2;
    } else if (MODE == MODE_SPMV_DOT) {
      a.y[i] = s1;
      acc0 = fma(xc, s1, acc0);
    } else {  // RHS: T1 = A, T2 = K; b = A v + r, r = dt (stim - K v)
      double stim = 0.0;
      for (int k = 0; k < a.nstim; ++k) stim = fma(a.amp[k], a.w[k][i], stim);
      const double r = a.dt * (stim - s2);
      const double b = a.mdiag[i] != 0.0 ? s1 + r : 0.0;  // nodes outside the tissue are not part of the system
      const double zz = a.dinv[i] * r;
      a.y[i] = r;
      a.y2[i] = zz;
      acc0 = fma(b, b, acc0);
      acc1 = fma(r, zz, acc1);
      acc2 = fma(r, r, acc2);
    }
  }
  if (MODE == MODE_SPMV_DOT) {
    const double s0 = beat_block_sum(acc0, red);
    if (threadIdx.x == 0) a.partials[a.part_off + blockIdx.x] = s0;
  } else if (MODE == MODE_RHS) {
    const double s0 = beat_block_sum(acc0, red);
    const double s1 = beat_block_sum(acc1, red);
    const double s2 = beat_block_sum(acc2, red);
    if (threadIdx.x == 0) {
      a.partials[a.part_off + blockIdx.x] = s0;
      a.partials[BEAT_MAX_PARTIALS + a.part_off + blockIdx.x] = s1;
      a.partials[2 * BEAT_MAX_PARTIALS + a.part_off + blockIdx.x] = s2;
    }
  }
}

// PCG vector updates over the active segments (per-node 1/diag)
__global__ __launch_bounds__(BEAT_BLOCK) void var_update_r_kernel(const int* __restrict__ seg, int nseg, int64_t n,
                                                                  const double* __restrict__ st,
                                                                  double* __restrict__ r, const double* __restrict__ q,
                                                                  const double* __restrict__ dinv,
                                                                  double* __restrict__ partials,
                                                                  double* __restrict__ alphas, int slot) {
  __shared__ double red[4];
  if (st[STOP] != 0.0) return;
  const double alpha = st[RZ] / st[PQ];
  if (blockIdx.x == 0 && threadIdx.x == 0) alphas[slot] = alpha;
  double s_rz = 0.0, s_rr = 0.0;
  for (int w = blockIdx.x; w < nseg; w += gridDim.x) {
    const int64_t i = (int64_t)seg[w] * BEAT_BLOCK + threadIdx.x;
    if (i >= n) continue;
    const double ri = fma(-alpha, q[i], r[i]);
    r[i] = ri;
    s_rz = fma(ri * dinv[i], ri, s_rz);
    s_rr = fma(ri, ri, s_rr);
  }
  const double a0 = beat_block_sum(s_rz, red);
  const double a1 = beat_block_sum(s_rr, red);
  if (threadIdx.x == 0) {
    partials[blockIdx.x] = a0;
    partials[BEAT_MAX_PARTIALS + blockIdx.x] = a1;
  }
}

__global__ __launch_bounds__(BEAT_BLOCK) void var_pupdate_oop_kernel(const int* __restrict__ seg, int nseg, int64_t n,
                                                                     const double* __restrict__ st,
                                                                     const double* __restrict__ r,
                                                                     const double* __restrict__ p_old,
                                                                     double* __restrict__ p_new,
                                                                     const double* __restrict__ dinv) {
  if (st[STOP] != 0.0) return;
  const double beta = st[BETA];
  for (int w = blockIdx.x; w < nseg; w += gridDim.x) {
    const int64_t i = (int64_t)seg[w] * BEAT_BLOCK + threadIdx.x;
    if (i >= n) continue;
    p_new[i] = fma(beta, p_old[i], dinv[i] * r[i]);
  }
}

__global__ __launch_bounds__(BEAT_BLOCK) void var_flush_kernel(const int* __restrict__ seg, int nseg, int64_t n,
                                                               const double* __restrict__ st, double* __restrict__ x,
                                                               const double* __restrict__ ring, int64_t fld,
                                                               const double* __restrict__ alphas, int ring_base,
                                                               int only_if_full) {
  int nvalid = (int)st[NUPD] - ring_base;
  nvalid = nvalid < 0 ? 0 : (nvalid > PRING ? PRING : nvalid);
  if (nvalid == 0 || (only_if_full && nvalid < PRING)) return;
  double a[PRING];
#pragma unroll
  for (int j = 0; j < PRING; ++j) a[j] = (j < nvalid) ? alphas[j] : 0.0;
  for (int w = blockIdx.x; w < nseg; w += gridDim.x) {
    const int64_t i = (int64_t)seg[w] * BEAT_BLOCK + threadIdx.x;
    if (i >= n) continue;
    double xi = x[i];
#pragma unroll
    for (int j = 0; j < PRING; ++j)
      if (j < nvalid) xi = fma(a[j], ring[(int64_t)j * fld + i], xi);
    x[i] = xi;
  }
}

// flags[s] = 1 if segment s holds a node touched by an element (mass diagonal > 0)
__global__ __launch_bounds__(BEAT_BLOCK) void var_segment_flags_kernel(int64_t n, const double* __restrict__ mass_diag,
                                                                       unsigned char* __restrict__ flags) {
  __shared__ int any;
  const int64_t nsegs = (n + BEAT_BLOCK - 1) / BEAT_BLOCK;
  for (int64_t s = blockIdx.x; s < nsegs; s += gridDim.x) {
    if (threadIdx.x == 0) any = 0;
    __syncthreads();
    const int64_t i = s * BEAT_BLOCK + threadIdx.x;
    if (i < n && mass_diag[i] != 0.0) any = 1;
    __syncthreads();
    if (threadIdx.x == 0) flags[s] = (unsigned char)any;
    __syncthreads();
  }
}

// A = C_m Mass + theta dt K per node, 1/diag(A); rows without any element (inactive voxels) become identity
__global__ __launch_bounds__(BEAT_BLOCK) void var_form_A_kernel(int64_t n, int64_t ld, const double* __restrict__ M,
                                                                const double* __restrict__ K, double cm, double tdt,
                                                                double* __restrict__ A, double* __restrict__ dinv) {
  const int64_t stride = (int64_t)gridDim.x * BEAT_BLOCK;
  for (int64_t i = (int64_t)blockIdx.x * BEAT_BLOCK + threadIdx.x; i < n; i += stride) {
    double d = cm * M[i] + tdt * K[i];
    const bool inactive = (M[i] == 0.0);
    if (inactive) d = 1.0;
    A[i] = d;
    dinv[i] = 1.0 / d;
#pragma unroll
    for (int k = 1; k < 15; ++k) {
      const int64_t j = (int64_t)k * ld + i;
      A[j] = inactive ? 0.0 : cm * M[j] + tdt * K[j];
    }
  }
}


// Assembly of the per-node rows on the device from per-voxel data (replaces dolfinx assemble_matrix of
// base_model.py:114-124 for voxelised geometries).  A voxel's 8x8 element stiffness matrix is linear in its
// conductivity tensor, K_e[a][b] = sum_ij T[a][b][i][j] M_ij with T fixed by the cell size and the 6-tet
// subdivision; node i gathers, from the <= 8 active voxels around it, the entries K_e[a][b] (a = its corner in
// that voxel) into the stencil slot of corner b - corner a.  One thread per node, coalesced row writes.
struct AsmArgs {
  int nx, ny, nz;       // local nodes
  int cx, cy, cz;       // global voxels per axis (1 for unused axes)
  int z0;               // global plane index of local plane 0
  const double* T;      // device, [8][8][9]
  const double* Me;     // device, [8][8] element mass
  const double* M;      // device (nvox, 9) or nullptr
  double Mc[9];         // constant tensor when M == nullptr
  const unsigned char* active;  // device (nvox) or nullptr
  int64_t ld;
  double* mass;
  double* stiff;
  signed char slot[64];  // stencil slot of (a, b), -1 if corner b - corner a is not a stencil offset
};

__global__ __launch_bounds__(BEAT_BLOCK) void assemble_rows_kernel(AsmArgs a) {
  __shared__ double sT[8 * 8 * 9];
  __shared__ double sMe[64];
  for (int k = threadIdx.x; k < 8 * 8 * 9; k += BEAT_BLOCK) sT[k] = a.T[k];
  if (threadIdx.x < 64) sMe[threadIdx.x] = a.Me[threadIdx.x];
  __syncthreads();
  const int64_t n = (int64_t)a.nx * a.ny * a.nz;
  const int64_t stride = (int64_t)gridDim.x * BEAT_BLOCK;
  for (int64_t i = (int64_t)blockIdx.x * BEAT_BLOCK + threadIdx.x; i < n; i += stride) {
    const int ix = (int)(i % a.nx);
    const int iy = (int)((i / a.nx) % a.ny);
    const int iz = (int)(i / ((int64_t)a.nx * a.ny)) + a.z0;
    double km[15], kk[15];
#pragma unroll
    for (int s = 0; s < 15; ++s) km[s] = kk[s] = 0.0;
    for (int c = 0; c < 8; ++c) {
      // voxel whose corner `c` is this node
      const int vx = ix - (c & 1), vy = iy - ((c >> 1) & 1), vz = iz - ((c >> 2) & 1);
      if (vx < 0 || vx >= a.cx || vy < 0 || vy >= a.cy || vz < 0 || vz >= a.cz) continue;
      const int64_t v = vx + (int64_t)a.cx * (vy + (int64_t)a.cy * vz);
      if (a.active != nullptr && a.active[v] == 0) continue;
      double m[9];
#pragma unroll
      for (int q = 0; q < 9; ++q) m[q] = a.M != nullptr ? a.M[v * 9 + q] : a.Mc[q];
      for (int b = 0; b < 8; ++b) {
        const int s = a.slot[c * 8 + b];
        if (s < 0) continue;
        const double* t = sT + (c * 8 + b) * 9;
        double acc = 0.0;
#pragma unroll
        for (int q = 0; q < 9; ++q) acc = fma(t[q], m[q], acc);
        // runtime slot index: select into the register arrays without dynamic indexing
#pragma unroll
        for (int u = 0; u < 15; ++u) {
          if (u == s) {
            kk[u] += acc;
            km[u] += sMe[c * 8 + b];
          }
        }
      }
    }
#pragma unroll
    for (int s = 0; s < 15; ++s) {
      a.mass[(int64_t)s * a.ld + i] = km[s];
      a.stiff[(int64_t)s * a.ld + i] = kk[s];
    }
  }
}


// Dirichlet conditions on per-node rows (symmetric elimination): flagged rows become identity, the couplings of
// free rows to flagged nodes move to the right-hand side f.
__global__ __launch_bounds__(BEAT_BLOCK) void rows_dirichlet_kernel(int64_t n, int64_t ld, double* __restrict__ rows,
                                                                    const unsigned char* __restrict__ flag,
                                                                    const double* __restrict__ g,
                                                                    double* __restrict__ f, VarArgs offs) {
  const int64_t stride = (int64_t)gridDim.x * BEAT_BLOCK;
  for (int64_t i = (int64_t)blockIdx.x * BEAT_BLOCK + threadIdx.x; i < n; i += stride) {
    if (flag[i]) {
      f[i] = g[i];
      rows[i] = 1.0;
#pragma unroll
      for (int k = 1; k < 15; ++k) rows[(int64_t)k * ld + i] = 0.0;
      continue;
    }
    double acc = 0.0;
#pragma unroll
    for (int k = 1; k < 15; ++k) {
      const double c = rows[(int64_t)k * ld + i];
      if (c == 0.0) continue;  // rows never couple outside the box, so i + doff is a valid node here
      const int64_t j = i + offs.doff[k];
      if (flag[j]) {
        acc = fma(-c, g[j], acc);
        rows[(int64_t)k * ld + i] = 0.0;
      }
    }
    f[i] = acc;
  }
}

}  // namespace

struct beat_pde {
  beat_ctx* ctx = nullptr;
  Geom g{};
  int64_t n = 0;
  double h_mass[27 * 15], h_stiff[27 * 15];
  double h_A[27 * 15], h_B[27 * 15], h_dinv[27];
  bool have_dt = false;
  double C_m = 1.0, theta = 0.5, dt = 0.0;
  // device: 4 padded tables (A, B, Mass, K), then dinv[32]
  double* d_tabs = nullptr;
  double* d_st = nullptr;  // 16 doubles, PCG scalar state of beat_pde_solve
  int last_iters = -1;
  unsigned vec_grid = 1;
  double* d_alphas = nullptr;  // PRING step lengths of the deferred-x PCG
  int pc_ncoef = 1;       // 1: Jacobi; m >= 2: Chebyshev polynomial of degree m-1 in D^-1 A (m-1 stencil passes)
  double pc_coef[8] = {1.0};
  // variable-coefficient mode (beat_pde_create_var): caller-owned Mass / K rows, A and 1/diag owned here
  bool var = false;
  const double* v_mass = nullptr;
  const double* v_stiff = nullptr;
  double* v_A = nullptr;
  double* v_dinv = nullptr;
  int64_t v_ld = 0;
  int* v_seg = nullptr;        // device: indices of the 256-node segments that hold tissue nodes (ascending)
  std::vector<int> h_seg;      // host copy (sub-ranges are located by binary search)
  const double* d_tab(int which) const { return d_tabs + (size_t)which * 27 * TABW; }
  const double* d_dinv() const { return d_tabs + (size_t)4 * 27 * TABW; }
  const double* dinv_arg() const { return var ? v_dinv : d_dinv(); }
};

// vector kernels are compiled for both the typed (27 node types) and the per-node 1/diag
#define BEAT_LAUNCH_VEC(pde, kernel, ...)                                   \
  do {                                                                      \
    if ((pde)->var)                                                         \
      hipLaunchKernelGGL((kernel<true>), __VA_ARGS__);                      \
    else                                                                    \
      hipLaunchKernelGGL((kernel<false>), __VA_ARGS__);                     \
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
  BEAT_HIP_CHECK(hipMalloc(&p->d_st, sizeof(double) * 16));
  BEAT_HIP_CHECK(hipMalloc(&p->d_alphas, sizeof(double) * PRING));
  BEAT_HIP_CHECK(hipMemsetAsync(p->d_st, 0, sizeof(double) * 16, ctx->stream));
  const int rc = upload_tables(p);  // Mass / K usable before the first set_timestep
  if (rc) return rc;
  *out = p;
  return BEAT_OK;
}


// ---- variable-coefficient mode ------------------------------------------------------------------------
static void var_offsets(const beat_pde* pde, VarArgs& a) {
  const Geom& g = pde->g;
  for (int k = 0; k < 15; ++k)
    a.doff[k] = kOffsets[3 * k] + g.nx * kOffsets[3 * k + 1] + (int)g.plane * kOffsets[3 * k + 2];
  a.ld = pde->v_ld;
}

// Work of a launch over planes [z_lo, z_hi): the sub-list of active segments intersecting the range and the grid
// (= number of block partials the launch writes).
struct VarRange {
  const int* seg;
  int nseg;
  unsigned grid;
};
static VarRange var_range(const beat_pde* pde, int z_lo, int z_hi, bool dense) {
  VarRange r{nullptr, 0, 0};
  if (z_hi <= z_lo) return r;
  const int64_t i_lo = (int64_t)z_lo * pde->g.plane, i_hi = (int64_t)z_hi * pde->g.plane;
  int64_t nwork = (i_hi - i_lo + BEAT_BLOCK - 1) / BEAT_BLOCK;
  if (!dense) {
    const int s_lo = (int)(i_lo / BEAT_BLOCK), s_hi = (int)((i_hi + BEAT_BLOCK - 1) / BEAT_BLOCK);
    const auto lo = std::lower_bound(pde->h_seg.begin(), pde->h_seg.end(), s_lo);
    const auto hi = std::lower_bound(pde->h_seg.begin(), pde->h_seg.end(), s_hi);
    r.seg = pde->v_seg + (lo - pde->h_seg.begin());
    r.nseg = (int)(hi - lo);
    nwork = r.nseg;
  }
  r.grid = (unsigned)std::min<int64_t>(4096, std::max<int64_t>(1, nwork));
  return r;
}

// launches over planes [z_lo, z_hi); returns the number of block partials written from part_off on.
// `dense` ignores the segment list (APPLY must write every node of y).
template <int MODE>
static int launch_var(const beat_pde* pde, VarArgs& a, int z_lo, int z_hi, int part_off, bool dense = false) {
  if (z_hi <= z_lo) return 0;
  const VarRange r = var_range(pde, z_lo, z_hi, dense);
  a.i_lo = (int64_t)z_lo * pde->g.plane;
  a.i_hi = (int64_t)z_hi * pde->g.plane;
  a.part_off = part_off;
  a.seg = r.seg;
  a.nseg = r.nseg;
  hipLaunchKernelGGL((var_stencil_kernel<MODE>), dim3(r.grid), dim3(BEAT_BLOCK), 0, pde->ctx->stream, a);
  return (int)r.grid;
}

static unsigned var_vec_grid(const beat_pde* pde) {
  return (unsigned)std::min<size_t>(4096, std::max<size_t>(1, pde->h_seg.size()));
}

static int var_form_A(beat_pde* pde) {
  const unsigned grid = (unsigned)std::min<int64_t>(4096, (pde->n + BEAT_BLOCK - 1) / BEAT_BLOCK);
  hipLaunchKernelGGL(var_form_A_kernel, dim3(grid), dim3(BEAT_BLOCK), 0, pde->ctx->stream, pde->n, pde->v_ld,
                     pde->v_mass, pde->v_stiff, pde->C_m, pde->theta * pde->dt, pde->v_A, pde->v_dinv);
  BEAT_LAUNCH_CHECK();
  return BEAT_OK;
}

extern "C" int beat_pde_create_var(beat_ctx* ctx, const int64_t n[3], int z_lo_phys, int z_hi_phys,
                                   const double* dev_mass, const double* dev_stiff, int64_t ld, beat_pde** out) {
  BEAT_REQUIRE(ctx != nullptr && n != nullptr && dev_mass && dev_stiff && out, "null argument");
  BEAT_REQUIRE(ld >= n[0] * n[1] * n[2], "leading dimension %lld smaller than the node count", (long long)ld);
  BEAT_REQUIRE(n[0] * n[1] * (n[2] + 2) < ((int64_t)1 << 31), "slab too large for 32-bit stencil offsets");
  std::vector<double> zeros(27 * 15, 0.0);
  beat_pde* p = nullptr;
  int rc = beat_pde_create(ctx, n, z_lo_phys, z_hi_phys, zeros.data(), zeros.data(), &p);
  if (rc) return rc;
  p->var = true;
  p->v_mass = dev_mass;
  p->v_stiff = dev_stiff;
  p->v_ld = ld;
  if (hipMalloc(&p->v_A, sizeof(double) * 15 * (size_t)ld) != hipSuccess ||
      hipMalloc(&p->v_dinv, sizeof(double) * (size_t)ld) != hipSuccess) {
    beat_pde_destroy(p);
    beat_set_error("out of device memory for the %lld-node coefficient rows", (long long)ld);
    return BEAT_EHIP;
  }
  // list of the 256-node segments that hold tissue nodes
  const int64_t nsegs = (p->n + BEAT_BLOCK - 1) / BEAT_BLOCK;
  unsigned char* d_flags = nullptr;
  std::vector<unsigned char> flags((size_t)nsegs);
  hipError_t e = hipMalloc(&d_flags, (size_t)nsegs);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(var_segment_flags_kernel, dim3((unsigned)std::min<int64_t>(4096, nsegs)), dim3(BEAT_BLOCK), 0,
                       ctx->stream, p->n, dev_mass, d_flags);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(flags.data(), d_flags, (size_t)nsegs, hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  (void)hipFree(d_flags);
  if (e == hipSuccess) {
    for (int64_t sidx = 0; sidx < nsegs; ++sidx)
      if (flags[(size_t)sidx]) p->h_seg.push_back((int)sidx);
    e = hipMalloc(&p->v_seg, sizeof(int) * std::max<size_t>(1, p->h_seg.size()));
  }
  if (e == hipSuccess && !p->h_seg.empty())
    e = hipMemcpy(p->v_seg, p->h_seg.data(), sizeof(int) * p->h_seg.size(), hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    beat_pde_destroy(p);
    beat_set_error("beat_pde_create_var: %s", hipGetErrorString(e));
    return BEAT_EHIP;
  }
  *out = p;
  return BEAT_OK;
}

extern "C" int beat_rows_apply_dirichlet(beat_ctx* ctx, const int64_t n[3], double* dev_rows, int64_t ld,
                                         const unsigned char* dev_flag, const double* dev_g, double* dev_f) {
  BEAT_REQUIRE(ctx && n && dev_rows && dev_flag && dev_g && dev_f, "null argument");
  const int64_t nn = n[0] * n[1] * n[2];
  BEAT_REQUIRE(nn >= 1 && ld >= nn && nn < ((int64_t)1 << 31), "bad sizes");
  VarArgs offs{};
  for (int k = 0; k < 15; ++k)
    offs.doff[k] = kOffsets[3 * k] + (int)n[0] * kOffsets[3 * k + 1] + (int)(n[0] * n[1]) * kOffsets[3 * k + 2];
  const unsigned grid = (unsigned)std::min<int64_t>(4096, (nn + BEAT_BLOCK - 1) / BEAT_BLOCK);
  hipLaunchKernelGGL(rows_dirichlet_kernel, dim3(grid), dim3(BEAT_BLOCK), 0, ctx->stream, nn, ld, dev_rows, dev_flag,
                     dev_g, dev_f, offs);
  BEAT_LAUNCH_CHECK();
  return BEAT_OK;
}

extern "C" int beat_pde_assemble_rows(beat_ctx* ctx, const int64_t n[3], const int64_t cells[3], int64_t z0,
                                      const double* host_T, const double* host_Me, const double* dev_M,
                                      const double* host_M_const, const unsigned char* dev_active,
                                      double* dev_mass, double* dev_stiff, int64_t ld) {
  BEAT_REQUIRE(ctx && n && cells && host_T && host_Me && dev_mass && dev_stiff, "null argument");
  BEAT_REQUIRE(dev_M != nullptr || host_M_const != nullptr, "no conductivity given");
  BEAT_REQUIRE(n[0] >= 1 && n[1] >= 1 && n[2] >= 1 && ld >= n[0] * n[1] * n[2], "bad sizes");
  BEAT_REQUIRE(n[0] * n[1] * n[2] < ((int64_t)1 << 40) && cells[0] * cells[1] * cells[2] < ((int64_t)1 << 40),
               "grid too large");
  AsmArgs a{};
  a.nx = (int)n[0];
  a.ny = (int)n[1];
  a.nz = (int)n[2];
  a.cx = (int)cells[0];
  a.cy = (int)cells[1];
  a.cz = (int)cells[2];
  a.z0 = (int)z0;
  a.M = dev_M;
  if (host_M_const)
    for (int q = 0; q < 9; ++q) a.Mc[q] = host_M_const[q];
  a.active = dev_active;
  a.ld = ld;
  a.mass = dev_mass;
  a.stiff = dev_stiff;
  for (int ca = 0; ca < 8; ++ca)
    for (int cb = 0; cb < 8; ++cb) {
      const int d[3] = {(cb & 1) - (ca & 1), ((cb >> 1) & 1) - ((ca >> 1) & 1), ((cb >> 2) & 1) - ((ca >> 2) & 1)};
      int slot = -1;
      for (int k = 0; k < 15; ++k)
        if (kOffsets[3 * k] == d[0] && kOffsets[3 * k + 1] == d[1] && kOffsets[3 * k + 2] == d[2]) slot = k;
      a.slot[ca * 8 + cb] = (signed char)slot;
    }
  BEAT_HIP_CHECK(hipSetDevice(ctx->device));
  double* d_t = nullptr;
  BEAT_HIP_CHECK(hipMalloc(&d_t, sizeof(double) * (8 * 8 * 9 + 64)));
  hipError_t e = hipMemcpyAsync(d_t, host_T, sizeof(double) * 8 * 8 * 9, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess)
    e = hipMemcpyAsync(d_t + 8 * 8 * 9, host_Me, sizeof(double) * 64, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) {
    a.T = d_t;
    a.Me = d_t + 8 * 8 * 9;
    const int64_t nn = n[0] * n[1] * n[2];
    const unsigned grid = (unsigned)std::min<int64_t>(8192, (nn + BEAT_BLOCK - 1) / BEAT_BLOCK);
    hipLaunchKernelGGL(assemble_rows_kernel, dim3(grid), dim3(BEAT_BLOCK), 0, ctx->stream, a);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);  // host tensors and d_t go out of scope
  }
  (void)hipFree(d_t);
  if (e != hipSuccess) {
    beat_set_error("beat_pde_assemble_rows: %s", hipGetErrorString(e));
    return BEAT_EHIP;
  }
  return BEAT_OK;
}

extern "C" int beat_pde_destroy(beat_pde* pde) {
  if (pde == nullptr) return BEAT_OK;
  (void)hipFree(pde->d_tabs);
  (void)hipFree(pde->d_st);
  (void)hipFree(pde->d_alphas);
  (void)hipFree(pde->v_A);
  (void)hipFree(pde->v_dinv);
  (void)hipFree(pde->v_seg);
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
  pde->C_m = C_m;
  pde->theta = theta;
  pde->dt = dt;
  pde->have_dt = true;
  pde->last_iters = -1;
  if (pde->var) return var_form_A(pde);
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
    case 256: hipLaunchKernelGGL((stencil_kernel<MODE, Tile<256, 4>>), grid, block, 0, s, g, a); break;
    case 128: hipLaunchKernelGGL((stencil_kernel<MODE, Tile<128, 8>>), grid, block, 0, s, g, a); break;
    default: hipLaunchKernelGGL((stencil_kernel<MODE, Tile<64, 16>>), grid, block, 0, s, g, a); break;
  }
}

extern "C" int beat_pde_apply(beat_pde* pde, int which, const double* dev_x, double* dev_y) {
  BEAT_REQUIRE(pde != nullptr && dev_x && dev_y, "null argument");
  BEAT_REQUIRE(which >= 0 && which < 4, "which must be 0..3");
  BEAT_REQUIRE(which >= 2 || pde->have_dt, "beat_pde_set_timestep has not been called");
  BEAT_REQUIRE(dev_x != dev_y, "in-place apply is not supported");
  if (pde->var) {
    VarArgs a{};
    var_offsets(pde, a);
    a.x = dev_x;
    a.y = dev_y;
    a.c1 = 1.0;
    a.c2 = 0.0;
    if (which == 0) {
      a.T1 = pde->v_A;
    } else if (which == 1) {  // B = C_m Mass - (1 - theta) dt K
      a.T1 = pde->v_mass;
      a.c1 = pde->C_m;
      a.T2 = pde->v_stiff;
      a.c2 = -(1.0 - pde->theta) * pde->dt;
    } else {
      a.T1 = which == 2 ? pde->v_mass : pde->v_stiff;
    }
    launch_var<MODE_APPLY>(pde, a, 0, pde->g.nz, 0, /*dense=*/true);
    BEAT_LAUNCH_CHECK();
    return BEAT_OK;
  }
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

static int launch_reduce(beat_pde* pde, int count, int nsum, double* out, const double* st,
                         double* counter = nullptr) {
  hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(BEAT_BLOCK), 0, pde->ctx->stream,
                     (const double*)pde->ctx->d_partials, count, nsum, out, st, counter);
  BEAT_LAUNCH_CHECK();
  return BEAT_OK;
}

extern "C" int beat_pde_rhs(beat_pde* pde, const double* dev_v_prev, const double* const* host_dev_stim_w,
                            const double* host_stim_amp, int n_stim, double* dev_x, double* dev_r,
                            double* dev_p, double* dev_red) {
  BEAT_REQUIRE(pde != nullptr && dev_v_prev && dev_x && dev_r && dev_p && dev_red, "null argument");
  BEAT_REQUIRE(pde->have_dt, "beat_pde_set_timestep has not been called");
  BEAT_REQUIRE(n_stim >= 0 && n_stim <= BEAT_MAX_STIM, "at most %d stimuli", BEAT_MAX_STIM);
  BEAT_REQUIRE(dev_r != dev_v_prev && dev_p != dev_v_prev && dev_r != dev_p, "r, p must be distinct work fields");
  if (pde->var) {
    VarArgs a{};
    var_offsets(pde, a);
    a.T1 = pde->v_A;
    a.T2 = pde->v_stiff;
    a.x = dev_v_prev;
    a.y = dev_r;
    a.y2 = dev_p;
    a.y3 = nullptr;
    if (dev_x != dev_v_prev)  // nodes outside the tissue keep their value: copy everything first
      BEAT_HIP_CHECK(hipMemcpyAsync(dev_x, dev_v_prev, sizeof(double) * (size_t)pde->n, hipMemcpyDeviceToDevice,
                                    pde->ctx->stream));
    a.dinv = pde->v_dinv;
    a.mdiag = pde->v_mass;
    a.dt = pde->dt;
    for (int k = 0; k < n_stim; ++k) {
      if (host_dev_stim_w[k] == nullptr || host_stim_amp[k] == 0.0) continue;
      a.w[a.nstim] = host_dev_stim_w[k];
      a.amp[a.nstim] = host_stim_amp[k];
      ++a.nstim;
    }
    a.partials = pde->ctx->d_partials;
    const int nb = launch_var<MODE_RHS>(pde, a, 0, pde->g.nz, 0);
    BEAT_LAUNCH_CHECK();
    return launch_reduce(pde, nb, 3, dev_red, nullptr);
  }
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
  return launch_reduce(pde, pde->g.total, 3, dev_red, nullptr);
}

extern "C" int beat_pde_cg_begin(beat_pde* pde, double* dev_st, double rtol, double atol, int max_it) {
  BEAT_REQUIRE(pde != nullptr && dev_st, "null argument");
  hipLaunchKernelGGL(pcg_begin_kernel, dim3(1), dim3(1), 0, pde->ctx->stream, dev_st, rtol, atol,
                     (double)max_it);
  BEAT_LAUNCH_CHECK();
  return BEAT_OK;
}

extern "C" int beat_pde_spmv_dot(beat_pde* pde, const double* dev_p, double* dev_q, double* dev_st) {
  BEAT_REQUIRE(pde != nullptr && dev_p && dev_q && dev_st, "null argument");
  BEAT_REQUIRE(pde->have_dt, "beat_pde_set_timestep has not been called");
  if (pde->var) {
    VarArgs a{};
    var_offsets(pde, a);
    a.T1 = pde->v_A;
    a.x = dev_p;
    a.y = dev_q;
    a.partials = pde->ctx->d_partials;
    a.st = dev_st;
    const int nb = launch_var<MODE_SPMV_DOT>(pde, a, 0, pde->g.nz, 0);
    BEAT_LAUNCH_CHECK();
    return launch_reduce(pde, nb, 1, dev_st + PQ, dev_st);
  }
  StencilArgs a{};
  a.x = dev_p;
  a.y = dev_q;
  a.tab = pde->d_tab(0);
  a.ci = interior(pde->h_A);
  a.partials = pde->ctx->d_partials;
  a.st = dev_st;
  launch_stencil<MODE_SPMV_DOT>(pde, a);
  BEAT_LAUNCH_CHECK();
  return launch_reduce(pde, pde->g.total, 1, dev_st + PQ, dev_st);
}

extern "C" int beat_pde_spmv_dot_part(beat_pde* pde, const double* dev_p, double* dev_q, double* dev_st, int part) {
  BEAT_REQUIRE(pde != nullptr && dev_p && dev_q && dev_st, "null argument");
  BEAT_REQUIRE(pde->have_dt, "beat_pde_set_timestep has not been called");
  BEAT_REQUIRE(part == 0 || part == 1, "part must be 0 (interior) or 1 (boundary planes + reduce)");
  const Geom& f = pde->g;
  const int lo = f.z_lo_phys ? 0 : 1, hi = f.nz - (f.z_hi_phys ? 0 : 1);  // planes that need no ghost data
  if (pde->var) {
    VarArgs a{};
    var_offsets(pde, a);
    a.T1 = pde->v_A;
    a.x = dev_p;
    a.y = dev_q;
    a.partials = pde->ctx->d_partials;
    a.st = dev_st;
    // block-partial slots: the interior launch owns [0, 4096), the boundary planes follow
    if (part == 0) {
      launch_var<MODE_SPMV_DOT>(pde, a, lo, std::max(lo, hi), 0);
      BEAT_LAUNCH_CHECK();
      return BEAT_OK;
    }
    int off = (int)var_range(pde, lo, std::max(lo, hi), false).grid;  // partial slots of the interior launch (part 0)
    if (!f.z_lo_phys) off += launch_var<MODE_SPMV_DOT>(pde, a, 0, 1, off);
    if (!f.z_hi_phys && (f.nz > 1 || f.z_lo_phys)) off += launch_var<MODE_SPMV_DOT>(pde, a, f.nz - 1, f.nz, off);
    BEAT_LAUNCH_CHECK();
    BEAT_REQUIRE(off <= BEAT_MAX_PARTIALS, "too many block partials");
    return launch_reduce(pde, off, 1, dev_st + PQ, dev_st);
  }
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
  return launch_reduce(pde, off, 1, dev_st + PQ, dev_st);
}

extern "C" int beat_pde_cg_update(beat_pde* pde, double* dev_st, double* dev_x, double* dev_r,
                                  const double* dev_p, const double* dev_q) {
  BEAT_REQUIRE(pde != nullptr && dev_st && dev_x && dev_r && dev_p && dev_q, "null argument");
  BEAT_LAUNCH_VEC(pde, cg_update_kernel, dim3(pde->vec_grid), dim3(BEAT_BLOCK), 0, pde->ctx->stream, pde->g,
                     (const double*)dev_st, dev_x, dev_r, dev_p, dev_q, pde->dinv_arg(), pde->h_dinv[13],
                     pde->ctx->d_partials);
  BEAT_LAUNCH_CHECK();
  return launch_reduce(pde, (int)pde->vec_grid, 2, dev_st + RZN, dev_st);
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
  if (a.pc_last) return launch_reduce(pde, pde->g.total, 1, dev_red, dev_st);
  return BEAT_OK;
}

// p = z + beta p after the scalar roll (polynomial-preconditioned variant of beat_pde_cg_next)
extern "C" int beat_pde_cg_next_z(beat_pde* pde, double* dev_st, const double* dev_z, double* dev_p) {
  BEAT_REQUIRE(pde != nullptr && dev_st && dev_z && dev_p, "null argument");
  hipLaunchKernelGGL(pcg_next_kernel, dim3(1), dim3(1), 0, pde->ctx->stream, dev_st);
  BEAT_LAUNCH_CHECK();
  const unsigned grid = (unsigned)std::min<int64_t>(2048, (pde->n + BEAT_BLOCK - 1) / BEAT_BLOCK);
  hipLaunchKernelGGL(cg_pupdate_z_kernel, dim3(grid), dim3(BEAT_BLOCK), 0, pde->ctx->stream, pde->n,
                     (const double*)dev_st, dev_z, dev_p);
  BEAT_LAUNCH_CHECK();
  return BEAT_OK;
}

// p = z (first search direction) without touching the scalars
extern "C" int beat_pde_cg_first_z(beat_pde* pde, double* dev_st, const double* dev_z, double* dev_p) {
  BEAT_REQUIRE(pde != nullptr && dev_st && dev_z && dev_p, "null argument");
  const unsigned grid = (unsigned)std::min<int64_t>(2048, (pde->n + BEAT_BLOCK - 1) / BEAT_BLOCK);
  hipLaunchKernelGGL(cg_pupdate_z_kernel, dim3(grid), dim3(BEAT_BLOCK), 0, pde->ctx->stream, pde->n,
                     (const double*)dev_st, dev_z, dev_p);
  BEAT_LAUNCH_CHECK();
  return BEAT_OK;
}

extern "C" int beat_pde_cg_next(beat_pde* pde, double* dev_st, const double* dev_r, double* dev_p) {
  BEAT_REQUIRE(pde != nullptr && dev_st && dev_r && dev_p, "null argument");
  hipLaunchKernelGGL(pcg_next_kernel, dim3(1), dim3(1), 0, pde->ctx->stream, dev_st);
  BEAT_LAUNCH_CHECK();
  BEAT_LAUNCH_VEC(pde, cg_pupdate_kernel, dim3(pde->vec_grid), dim3(BEAT_BLOCK), 0, pde->ctx->stream, pde->g,
                     (const double*)dev_st, dev_r, dev_p, pde->dinv_arg(), pde->h_dinv[13]);
  BEAT_LAUNCH_CHECK();
  return BEAT_OK;
}

extern "C" int beat_pde_work_fields(beat_pde* pde) {
  return pde ? 3 + PRING : BEAT_EINVAL;  // r, q, z + the ring of search directions
}

// ---- deferred-x stages for callers that drive the iteration themselves (slab-decomposed solve) ----------
extern "C" int beat_pde_ring_size(void) { return PRING; }

// r -= alpha q, alpha = st[1]/st[3] (also kept for the later x update, slot = iteration % ring size);
// LOCAL r.D^-1 r, r.r -> dev_st[4..5]; counts the executed update in dev_st[14].
extern "C" int beat_pde_cg_update_r(beat_pde* pde, double* dev_st, double* dev_r, const double* dev_q, int slot) {
  BEAT_REQUIRE(pde != nullptr && dev_st && dev_r && dev_q, "null argument");
  BEAT_REQUIRE(slot >= 0 && slot < PRING, "slot %d out of range", slot);
  if (pde->var) {
    const unsigned grid = var_vec_grid(pde);
    hipLaunchKernelGGL(var_update_r_kernel, dim3(grid), dim3(BEAT_BLOCK), 0, pde->ctx->stream, (const int*)pde->v_seg,
                       (int)pde->h_seg.size(), pde->n, (const double*)dev_st, dev_r, dev_q, (const double*)pde->v_dinv,
                       pde->ctx->d_partials, pde->d_alphas, slot);
    BEAT_LAUNCH_CHECK();
    return launch_reduce(pde, (int)grid, 2, dev_st + RZN, dev_st, dev_st + NUPD);
  }
  BEAT_LAUNCH_VEC(pde, cg_update_r_kernel, dim3(pde->vec_grid), dim3(BEAT_BLOCK), 0, pde->ctx->stream, pde->g,
                     (const double*)dev_st, dev_r, dev_q, pde->dinv_arg(), pde->h_dinv[13], pde->ctx->d_partials,
                     pde->d_alphas, slot);
  BEAT_LAUNCH_CHECK();
  return launch_reduce(pde, (int)pde->vec_grid, 2, dev_st + RZN, dev_st, dev_st + NUPD);
}

// scalar roll (beta, latch, iteration count) then p_next = D^-1 r + beta p_cur
extern "C" int beat_pde_cg_next_oop(beat_pde* pde, double* dev_st, const double* dev_r, const double* dev_p_cur,
                                    double* dev_p_next) {
  BEAT_REQUIRE(pde != nullptr && dev_st && dev_r && dev_p_cur && dev_p_next, "null argument");
  BEAT_REQUIRE(dev_p_cur != dev_p_next, "the p-update is out of place");
  hipLaunchKernelGGL(pcg_next_kernel, dim3(1), dim3(1), 0, pde->ctx->stream, dev_st);
  BEAT_LAUNCH_CHECK();
  if (pde->var) {
    hipLaunchKernelGGL(var_pupdate_oop_kernel, dim3(var_vec_grid(pde)), dim3(BEAT_BLOCK), 0, pde->ctx->stream,
                       (const int*)pde->v_seg, (int)pde->h_seg.size(), pde->n, (const double*)dev_st, dev_r, dev_p_cur,
                       dev_p_next, (const double*)pde->v_dinv);
    BEAT_LAUNCH_CHECK();
    return BEAT_OK;
  }
  BEAT_LAUNCH_VEC(pde, cg_pupdate_oop_kernel, dim3(pde->vec_grid), dim3(BEAT_BLOCK), 0, pde->ctx->stream, pde->g,
                     (const double*)dev_st, dev_r, dev_p_cur, dev_p_next, pde->dinv_arg(), pde->h_dinv[13]);
  BEAT_LAUNCH_CHECK();
  return BEAT_OK;
}

// x += sum_j alpha_j ring_j over the valid directions of the ring cycle starting at iteration ring_base
// (ring_j = dev_ring0 + j*field_stride); with only_if_full it acts only when that cycle filled up.
extern "C" int beat_pde_x_flush(beat_pde* pde, const double* dev_st, double* dev_x, const double* dev_ring0,
                                int64_t field_stride, int ring_base, int only_if_full) {
  BEAT_REQUIRE(pde != nullptr && dev_st && dev_x && dev_ring0, "null argument");
  if (pde->var) {
    hipLaunchKernelGGL(var_flush_kernel, dim3(var_vec_grid(pde)), dim3(BEAT_BLOCK), 0, pde->ctx->stream,
                       (const int*)pde->v_seg, (int)pde->h_seg.size(), pde->n, dev_st, dev_x, dev_ring0, field_stride,
                       (const double*)pde->d_alphas, ring_base, only_if_full);
    BEAT_LAUNCH_CHECK();
    return BEAT_OK;
  }
  const unsigned grid = (unsigned)std::min<int64_t>(2048, (pde->n + BEAT_BLOCK - 1) / BEAT_BLOCK);
  hipLaunchKernelGGL(x_flush_kernel, dim3(grid), dim3(BEAT_BLOCK), 0, pde->ctx->stream, pde->n, dev_st, dev_x,
                     dev_ring0, field_stride, (const double*)pde->d_alphas, ring_base, only_if_full);
  BEAT_LAUNCH_CHECK();
  return BEAT_OK;
}

extern "C" int beat_pde_solve(beat_pde* pde, const double* dev_v_prev,
                              const double* const* host_dev_stim_w, const double* host_stim_amp,
                              int n_stim, double* dev_x, double* dev_work, double rtol, double atol,
                              int max_it, beat_ksp_info* info) {
  BEAT_REQUIRE(pde != nullptr && dev_work != nullptr, "null argument");
  BEAT_REQUIRE(pde->g.z_lo_phys && pde->g.z_hi_phys, "beat_pde_solve is the single-slab path");
  BEAT_REQUIRE(max_it >= 0, "max_it must be >= 0");
  const int64_t fld = pde->n + 2 * pde->g.plane;
  double* r = dev_work + pde->g.plane;
  double* q = r + fld;
  double* z = q + fld;     // only touched by the polynomial preconditioner
  double* ring = z + fld;  // PRING search directions, ring[j] = ring + j*fld
  double* st = pde->d_st;
  beat_ctx* ctx = pde->ctx;
  double* h = ctx->h_pinned;
  const int npass = pde->pc_ncoef - 1;
  int rc = beat_pde_rhs(pde, dev_v_prev, host_dev_stim_w, host_stim_amp, n_stim, dev_x, r, ring, st);
  if (rc) return rc;
  if ((rc = beat_pde_cg_begin(pde, st, rtol, atol, max_it))) return rc;
  int launched = 0;
  int chunk = pde->last_iters > 0 ? pde->last_iters : 8;
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
    // Jacobi, deferred x: iteration i uses p_i = ring[i % PRING]
    while (true) {
      chunk = std::min(chunk, max_it - launched);
      for (int it = 0; it < chunk; ++it) {
        const int i = launched + it, slot = i % PRING;
        double* p_cur = ring + (int64_t)slot * fld;
        double* p_next = ring + (int64_t)((i + 1) % PRING) * fld;
        if ((rc = beat_pde_spmv_dot(pde, p_cur, q, st))) return rc;
        if ((rc = beat_pde_cg_update_r(pde, st, r, q, slot))) return rc;
        if (slot == PRING - 1) {  // ring full: bring x up to date before slot 0 is overwritten
          if ((rc = beat_pde_x_flush(pde, st, dev_x, ring, fld, i + 1 - PRING, 1))) return rc;
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
    const int nupd = (int)h[NUPD];
    if (nupd % PRING != 0)
      if ((rc = beat_pde_x_flush(pde, st, dev_x, ring, fld, (nupd / PRING) * PRING, 0))) return rc;
  }
  const int iters = (int)h[ITERS];
  pde->last_iters = iters;
  int reason = (int)h[REASON];
  if (h[STOP] == 0.0) reason = -3;  // max_it == launched without the latch (max_it = 0)
  if (info) {
    info->iterations = iters;
    info->converged_reason = reason;
    info->residual_norm = std::sqrt(h[RR]);
    info->rhs_norm = std::sqrt(h[BB]);
  }
  if (reason < 0) {
    beat_set_error("PCG did not converge in %d iterations (||r|| = %.3e, ||b|| = %.3e)", iters,
                   std::sqrt(h[RR]), std::sqrt(h[BB]));
    return BEAT_ENOTCONV;
  }
  return BEAT_OK;
}
