// Per-node-coefficient form of the diffusion operators (voxel-masked domains, spatially varying conductivity):
// kernels, device-side assembly of the rows from per-voxel tensors, Dirichlet elimination for the Laplace
// problems of utils.expand_layer, and the beat_var_* stage functions the C ABI in beat_pde.hip dispatches to.
#include "beat_pde_internal.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

namespace {
using namespace beat_pde_detail;

// Granularity of the tissue bookkeeping: one wavefront's worth of consecutive nodes.  A workgroup takes
// VAR_SEGS_PER_BLOCK list entries at a time, one per wave.
constexpr int VAR_SEG = 64;
constexpr int VAR_SEGS_PER_BLOCK = BEAT_BLOCK / VAR_SEG;

// ---- variable-coefficient operators (beat_pde_create_var) -----------------------------------------------
// Voxel-masked domains and spatially varying conductivity: every node carries its own 15 coefficients per
// operator, stored coefficient-major ((15, ld) arrays: a wave reads contiguous 512 B segments).  Work is organised
// by a list of the 64-node segments that hold tissue, one list entry per wavefront, with a 64-bit mask of the
// tissue nodes in it: lanes on other nodes issue no loads or stores, segments without tissue are never visited.
// The PCG's matrix-vector product is var_spmv_kernel (rows of x loaded once and shifted across the wave, backward
// coefficients read as the neighbours' forward ones); the right-hand side and the plain application y = (c1 T1 +
// c2 T2) x gather their neighbour values through L1/L2 (var_stencil_kernel).  A neighbour value is only used where
// its coefficient is non-zero: rows never reach outside the box (or into an inactive voxel), so no out-of-range
// address is formed and stale ghost planes cannot leak NaNs.
struct VarArgs {
  const double* T1;  // (15, ld) coefficients
  const double* T2;  // second operator (RHS: stiffness; APPLY: optional) or nullptr
  double c1, c2;     // APPLY: y = (c1 T1 + c2 T2) x
  int64_t ld;
  const double* x;
  const double* x2;  // RHS: the initial-guess increment e (x0 = v_ + e, r = b - A x0), or nullptr
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
  const int* seg;      // active segments (VAR_SEG nodes each) covering [i_lo, i_hi) (nullptr: every node of the range)
  const unsigned long long* segmask;  // per list entry: which of its nodes are tissue nodes
  int nseg;
  int tiled;  // the list is in tile order (runs of 8 x 8 rows): every XCD takes one contiguous eighth of it, in order
};

// The list entries a wave visits.  Plain: entry 4 b + wave of block b, stride 4 gridDim -- consecutive entries land on
// consecutive blocks, i.e. on all eight XCDs.  Tiled (whole-slab launches of the solver's kernels): the list is ordered
// by tiles of 8 rows x 8 planes (all x segments of a row together), XCD x (blocks b with b % 8 == x) walks the x-th eighth
// of it front to back, so that the ~640 entries an XCD has in flight are one or two tiles: the rows of p around a node and
// the neighbours' forward coefficients it reads as its backward ones are lines that XCD's L2 has just fetched.
// The wave's index in its block as a SCALAR: everything derived from it -- the list entry w, seg[w], segmask[w] -- is
// then wave-uniform for the compiler too and read with scalar loads.  Derived from threadIdx.x alone those two reads were
// vector loads, each followed by s_waitcnt vmcnt(0): two more dependent trips to memory per segment ahead of the
// coefficients and the rows of p, and a wait that any load issued earlier had to share.
__device__ __forceinline__ int var_wave() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x / VAR_SEG)); }

struct VarWalk {
  int64_t w, end, stride;
};
__device__ __forceinline__ VarWalk var_walk(const VarArgs& a, int64_t nwork, int wave) {
  if (a.tiled) {
    const int nbx = gridDim.x >> 3;
    const int64_t per = (nwork + 7) >> 3;
    const int xcd = blockIdx.x & 7, lb = blockIdx.x >> 3;
    const int64_t start = xcd * per;
    const int64_t end = lb < nbx ? (start + per < nwork ? start + per : nwork) : 0;
    return VarWalk{start + (int64_t)lb * VAR_SEGS_PER_BLOCK + wave, end, (int64_t)nbx * VAR_SEGS_PER_BLOCK};
  }
  return VarWalk{(int64_t)blockIdx.x * VAR_SEGS_PER_BLOCK + wave, nwork, (int64_t)gridDim.x * VAR_SEGS_PER_BLOCK};
}

template <int MODE>
__global__ __launch_bounds__(BEAT_BLOCK) void var_stencil_kernel(VarArgs a) {
  static_assert(MODE == MODE_APPLY || MODE == MODE_RHS, "the PCG's SpMV is var_spmv_kernel");
  __shared__ double red[4];
  double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0;
  // Work items: the wavefront-sized segments that hold at least one tissue node (list built at create time); segments
  // entirely outside the tissue are never read or written (their r, p, q stay zero, x keeps its value).
  const int64_t nwork = a.seg ? a.nseg : (a.i_hi - a.i_lo + VAR_SEG - 1) / VAR_SEG;
  const int wave = var_wave(), lane = threadIdx.x % VAR_SEG;
  const VarWalk walk = var_walk(a, nwork, wave);
  for (int64_t w = walk.w; w < walk.end; w += walk.stride) {
    const int64_t i = (a.seg ? (int64_t)a.seg[w] * VAR_SEG : a.i_lo + w * VAR_SEG) + lane;
    if (i < a.i_lo || i >= a.i_hi) continue;
    // lanes on nodes outside the tissue issue no loads or stores: lines without a tissue node are never fetched
    if (a.seg && !((a.segmask[w] >> lane) & 1ull)) continue;
    double s1 = 0.0, s2 = 0.0, se = 0.0;
#pragma unroll
    for (int k = 0; k < 15; ++k) {
      // The operators are symmetric: the coefficient towards a "backward" neighbour (even slot k, offset
      // -o) equals that neighbour's "forward" coefficient (slot k-1) -- a line some other wave streams in anyway,
      // so it comes from L2 / Infinity Cache instead of being a 15th..9th HBM stream (SpMV -26 % on a 64 M-node
      // box).  Only rows this slab stores can be used that way; the first planes of a slab with a live ghost
      // plane read their own backward slots.
      int64_t src = (int64_t)k * a.ld + i;
      if (MODE != MODE_APPLY && k >= 2 && (k & 1) == 0) {
        const int64_t jn = i + a.doff[k];
        if (jn >= 0) src = (int64_t)(k - 1) * a.ld + jn;
      }
      const double c1 = a.T1[src];
      double c2 = 0.0;
      if (MODE == MODE_RHS) c2 = a.T2[src];
      if (MODE == MODE_APPLY && a.T2 != nullptr) c2 = a.T2[src];
      const bool need = (k == 0) || c1 != 0.0 || c2 != 0.0;
      const double xv = need ? a.x[i + a.doff[k]] : 0.0;
      s1 = fma(c1, xv, s1);
      s2 = fma(c2, xv, s2);
      if (MODE == MODE_RHS && a.x2 != nullptr) {  // A e, same gather pattern
        const double ev = (k == 0 || c1 != 0.0) ? a.x2[i + a.doff[k]] : 0.0;
        se = fma(c1, ev, se);
      }
    }
    if (MODE == MODE_APPLY) {
      a.y[i] = a.c1 * s1 + a.c2 * s2;
    } else {  // RHS: T1 = A, T2 = K; b = A v + r, r = dt (stim - K v)
      double stim = 0.0;
      for (int k = 0; k < a.nstim; ++k) stim = fma(a.amp[k], a.w[k][i], stim);
      const double r0 = a.dt * (stim - s2);
      const double b = s1 + r0;  // (nodes outside the tissue are masked out above: not part of the system)
      const double r = r0 - se;  // residual at x0 = v_ + e (se = 0 without a guess)
      const double zz = a.dinv[i] * r;
      a.y[i] = r;
      a.y2[i] = zz;
      acc0 = fma(b, b, acc0);
      acc1 = fma(r, zz, acc1);
      acc2 = fma(r, r, acc2);
    }
  }
  if (MODE == MODE_RHS) {
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

// q = A p and the block partials of p.q: the PCG's matrix-vector product, one wavefront per 64-node list entry.
//
// The 15 stencil points lie in 7 rows of x ((dy, dz) pairs); a row's dx = 0 value of lane l is the dx = +1 value of
// lane l-1 and the dx = -1 value of lane l+1, so each row is loaded once and shifted across the wave (DPP / bpermute)
// instead of being gathered two or three times through L1: 7 row loads plus one 8-lane load for the values just
// outside the segment (lane 63's +1 taps, lane 0's -1 taps) replace 15 gathers (SpMV 0.71 -> 0.59 ms on a 401^3
// shell: the kernel is bound by the rate of its 512-B gathers, not by HBM bytes; 0.49 ms would be the time without
// those loads at all).  Measured and not adopted: taking the slot-2 coefficient from the neighbouring lane's slot 1 as
// well (0.64 ms) and capping the kernel at 80 VGPRs for six waves per SIMD (0.63 ms, 12 spills).  A lane loads a row value if it
// needs it itself or a neighbouring lane does (ballots of the coefficient tests); values are selected, never
// multiplied by a zero coefficient, so stale ghost planes cannot leak NaNs.  The sums are accumulated in slot order
// 0..14 exactly as var_stencil_kernel does.
__device__ __forceinline__ double var_readlane(double v, int l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}

// (helpers of var_rhs_kernel.  The SpMV below keeps its own, branchy form: rebuilt on these helpers it ran 0.65 instead of
// 0.58 ms on a 401^3 shell -- 14.1 against 13.3 ms per step, alternated on one box.)
// x at the 15 stencil points of node seg0 + lane, for all lanes of a 64-node segment at once (every lane of the wave
// must call this).  nz[k]: this lane uses slot k; own0: the lane's node is an active node (slot 0).  A row value is
// loaded from where it lives if the lane itself or the lane it is shifted to uses it, and from `safe` (an index inside
// the slab) otherwise: every lane loads, so the loads of a segment are issued back to back instead of one branch and one
// wait per row -- and what an unused lane loaded is never selected, so no address outside the box is formed and a
// stale ghost plane cannot leak a NaN.
__device__ __forceinline__ void var_row_values(const double* __restrict__ x, int64_t seg0, int lane, bool own0,
                                               int64_t safe, int edge_off, const bool (&nz)[15], const int (&doff)[15],
                                               double (&xk)[15]) {
  // rows of the stencil: slot of the dx = 0 point, of the dx = +1 point and of the dx = -1 point (-1: none)
  constexpr int kBase[7] = {0, 3, 4, 5, 6, 9, 10};
  constexpr int kPlus[7] = {1, 7, -1, 11, -1, 13, -1};
  constexpr int kMinus[7] = {2, -1, 8, -1, 12, -1, 14};
  const int64_t i = seg0 + lane;
  unsigned long long bp[7], bm[7];
#pragma unroll
  for (int r = 0; r < 7; ++r) {
    bp[r] = kPlus[r] >= 0 ? __ballot(nz[kPlus[r] >= 0 ? kPlus[r] : 0]) : 0ull;
    bm[r] = kMinus[r] >= 0 ? __ballot(nz[kMinus[r] >= 0 ? kMinus[r] : 0]) : 0ull;
  }
  // values just outside the segment: lanes 0..3 fetch lane 63's +1 taps (rows 0, 1, 3, 5), lanes 4..7 lane 0's
  // -1 taps (rows 0, 2, 4, 6); bit l of `edges`: lane l's value is wanted
  const unsigned edges = (unsigned)(bp[0] >> 63) | (unsigned)(bp[1] >> 63) << 1 | (unsigned)(bp[3] >> 63) << 2 |
                         (unsigned)(bp[5] >> 63) << 3 | (unsigned)(bm[0] & 1ull) << 4 | (unsigned)(bm[2] & 1ull) << 5 |
                         (unsigned)(bm[4] & 1ull) << 6 | (unsigned)(bm[6] & 1ull) << 7;
  const bool want_edge = lane < 8 && ((edges >> (lane & 7)) & 1u);
  const double edge = x[want_edge ? seg0 + edge_off : safe];
  // the seven row loads, issued together
  double X[7];
#pragma unroll
  for (int r = 0; r < 7; ++r) {
    const int kb = kBase[r];
    const bool own = kb == 0 ? own0 : nz[kb];
    const unsigned long long wanted = (bp[r] << 1) | (bm[r] >> 1);  // lane l-1 wants its +1, lane l+1 its -1 tap
    const bool take = own | (((wanted >> lane) & 1ull) != 0ull);
    X[r] = x[take ? i + doff[kb] : safe];
    xk[kb] = own ? X[r] : 0.0;
  }
#pragma unroll
  for (int r = 0; r < 7; ++r) {
    if (kPlus[r] >= 0) {
      const int kp = kPlus[r] >= 0 ? kPlus[r] : 0;
      const int e = r == 0 ? 0 : r == 1 ? 1 : r == 3 ? 2 : 3;
      const double shifted = __shfl_down(X[r], 1, VAR_SEG);
      const double v = lane == VAR_SEG - 1 ? var_readlane(edge, e) : shifted;
      xk[kp] = nz[kp] ? v : 0.0;
    }
    if (kMinus[r] >= 0) {
      const int km = kMinus[r] >= 0 ? kMinus[r] : 0;
      const int e = r == 0 ? 4 : r == 2 ? 5 : r == 4 ? 6 : 7;
      const double shifted = __shfl_up(X[r], 1, VAR_SEG);
      const double v = lane == 0 ? var_readlane(edge, e) : shifted;
      xk[km] = nz[km] ? v : 0.0;
    }
  }
}

// offset (from the first node of a segment) of the value lane `lane` < 8 fetches for var_row_values' segment edges
__device__ __forceinline__ int var_edge_offset(int lane, const int (&doff)[15]) {
  const int e = lane & 7;
  const int off = e == 0 || e == 4 ? doff[0] : e == 1 ? doff[3] : e == 2 ? doff[5] : e == 3 ? doff[9]
                  : e == 5 ? doff[4] : e == 6 ? doff[6] : doff[10];
  return (e < 4 ? VAR_SEG : -1) + off;
}

// coefficient row of node i (0.0 on inactive lanes, which load the row of node `safe` instead of branching around the
// loads); symmetric operator: the coefficient towards a backward neighbour is that neighbour's forward coefficient (see
// var_stencil_kernel)
__device__ __forceinline__ void var_row_coefficients(const double* __restrict__ T, int64_t ld, int64_t i, bool active,
                                                     int64_t safe, const int (&doff)[15], double (&c)[15]) {
  const int64_t ii = active ? i : safe;
  // (the row stride opaque per segment: the 15 products k ld of each operator are loop invariants otherwise, hoisted into
  // SGPR pairs the kernel does not have -- 111 spilled SGPRs in var_rhs_kernel)
  asm volatile("" : "+s"(ld));
#pragma unroll
  for (int k = 0; k < 15; ++k) {
    int64_t src = (int64_t)k * ld + ii;
    if (k >= 2 && (k & 1) == 0) {
      const int64_t jn = ii + doff[k];
      if (jn >= 0) src = (int64_t)(k - 1) * ld + jn;
    }
    const double v = T[src];
    c[k] = active ? v : 0.0;
  }
}

__global__ __launch_bounds__(BEAT_BLOCK) void var_spmv_kernel(VarArgs a) {
  __shared__ double red[4];
  if (a.st[STOP] != 0.0) return;
  // rows of the stencil: slot of the dx = 0 point, of the dx = +1 point and of the dx = -1 point (-1: none)
  constexpr int kBase[7] = {0, 3, 4, 5, 6, 9, 10};
  constexpr int kPlus[7] = {1, 7, -1, 11, -1, 13, -1};
  constexpr int kMinus[7] = {2, -1, 8, -1, 12, -1, 14};
  double acc0 = 0.0;
  const int64_t nwork = a.seg ? a.nseg : (a.i_hi - a.i_lo + VAR_SEG - 1) / VAR_SEG;
  const int wave = var_wave(), lane = threadIdx.x % VAR_SEG;
  const VarWalk walk = var_walk(a, nwork, wave);
  for (int64_t w = walk.w; w < walk.end; w += walk.stride) {
    const int64_t seg0 = a.seg ? (int64_t)a.seg[w] * VAR_SEG : a.i_lo + w * VAR_SEG;  // wave-uniform
    const int64_t i = seg0 + lane;
    const bool active = i >= a.i_lo && i < a.i_hi && (!a.seg || ((a.segmask[w] >> lane) & 1ull));
    double c[15];
#pragma unroll
    for (int k = 0; k < 15; ++k) {
      // symmetric operator: the coefficient towards a backward neighbour is that neighbour's forward coefficient
      // (see var_stencil_kernel)
      int64_t src = (int64_t)k * a.ld + i;
      if (k >= 2 && (k & 1) == 0) {
        const int64_t jn = i + a.doff[k];
        if (jn >= 0) src = (int64_t)(k - 1) * a.ld + jn;
      }
      c[k] = active ? a.T1[src] : 0.0;
    }
    unsigned long long bp[7], bm[7];
#pragma unroll
    for (int r = 0; r < 7; ++r) {
      bp[r] = kPlus[r] >= 0 ? __ballot(c[kPlus[r] >= 0 ? kPlus[r] : 0] != 0.0) : 0ull;
      bm[r] = kMinus[r] >= 0 ? __ballot(c[kMinus[r] >= 0 ? kMinus[r] : 0] != 0.0) : 0ull;
    }
    // values just outside the segment: lanes 0..3 fetch lane 63's +1 taps (rows 0, 1, 3, 5), lanes 4..7 lane 0's
    // -1 taps (rows 0, 2, 4, 6)
    double edge = 0.0;
    {
      const int e = lane & 7;
      const bool plus = e < 4;
      const int off = e == 0 || e == 4 ? a.doff[0] : e == 1 ? a.doff[3] : e == 2 ? a.doff[5] : e == 3 ? a.doff[9]
                      : e == 5 ? a.doff[4] : e == 6 ? a.doff[6] : a.doff[10];
      const unsigned long long wanted = e == 0 ? bp[0] >> 63 : e == 1 ? bp[1] >> 63 : e == 2 ? bp[3] >> 63
                                        : e == 3 ? bp[5] >> 63 : e == 4 ? bm[0] : e == 5 ? bm[2] : e == 6 ? bm[4] : bm[6];
      if (lane < 8 && (wanted & 1ull)) edge = a.x[seg0 + (plus ? VAR_SEG : -1) + off];
    }
    // all seven rows of p requested before the first of them is used: written as one loop -- load a row, shift it across
    // the wave, load the next -- every load sat in its own predicated block behind the shuffles of the row before it and
    // was waited for (s_waitcnt vmcnt(0)) before the next one was issued: seven trips to memory in a row per segment
    double xk[15], Xr[7];
#pragma unroll
    for (int r = 0; r < 7; ++r) {
      const int kb = kBase[r];
      const bool own = kb == 0 ? active : c[kb] != 0.0;
      const unsigned long long wanted = (bp[r] << 1) | (bm[r] >> 1);  // lane l-1 wants its +1, lane l+1 its -1 tap
      Xr[r] = (own || ((wanted >> lane) & 1ull)) ? a.x[i + a.doff[kb]] : 0.0;
    }
#pragma unroll
    for (int r = 0; r < 7; ++r) {
      const int kb = kBase[r];
      const bool own = kb == 0 ? active : c[kb] != 0.0;
      const double X = Xr[r];
      xk[kb] = own ? X : 0.0;
      if (kPlus[r] >= 0) {
        const int kp = kPlus[r] >= 0 ? kPlus[r] : 0;
        const int e = r == 0 ? 0 : r == 1 ? 1 : r == 3 ? 2 : 3;
        const double shifted = __shfl_down(X, 1, VAR_SEG);
        const double v = lane == VAR_SEG - 1 ? var_readlane(edge, e) : shifted;
        xk[kp] = c[kp] != 0.0 ? v : 0.0;
      }
      if (kMinus[r] >= 0) {
        const int km = kMinus[r] >= 0 ? kMinus[r] : 0;
        const int e = r == 0 ? 4 : r == 2 ? 5 : r == 4 ? 6 : 7;
        const double shifted = __shfl_up(X, 1, VAR_SEG);
        const double v = lane == 0 ? var_readlane(edge, e) : shifted;
        xk[km] = c[km] != 0.0 ? v : 0.0;
      }
    }
    double s1 = 0.0;
#pragma unroll
    for (int k = 0; k < 15; ++k) s1 = fma(c[k], xk[k], s1);
    if (active) {
      a.y[i] = s1;
      acc0 = fma(xk[0], s1, acc0);
    }
  }
  const double s0 = beat_block_sum(acc0, red);
  if (threadIdx.x == 0) a.partials[a.part_off + blockIdx.x] = s0;
}

// The right-hand side of a step (what var_stencil_kernel<MODE_RHS> computes, same sums in the same order, so the
// same bits) with the rows of v_ -- and of the guess increment e -- loaded once and shifted across the wave like the
// SpMV's: T1 = A, T2 = K;  r0 = dt (stim - K v_),  b = A v_ + r0,  r = r0 - A e,  z = D^-1 r,  partials of b.b, r.z, r.r.
// The 2 x 15 gathers per node of the plain kernel made it cost three SpMVs (1.82 against 0.58 ms on a 401^3 shell).
__global__ __launch_bounds__(BEAT_BLOCK) void var_rhs_kernel(VarArgs a_) {
  __shared__ double red[4];
  double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0;
  const int64_t nwork = a_.seg ? a_.nseg : (a_.i_hi - a_.i_lo + VAR_SEG - 1) / VAR_SEG;
  const int wave = var_wave(), lane = threadIdx.x % VAR_SEG;
  const int edge_off = var_edge_offset(lane, a_.doff);
  const VarWalk walk = var_walk(a_, nwork, wave);
  for (int64_t w = walk.w; w < walk.end; w += walk.stride) {
    // the arguments through a pointer the optimiser cannot see through, per segment: the 15 offsets, the operator and
    // stimulus pointers are read where a segment needs them instead of being held in SGPRs across the loop (67 spilled)
    typedef const __attribute__((address_space(4))) char* KArgPtr;
    KArgPtr ka = (KArgPtr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(ka));
    const VarArgs& a = *(const VarArgs*)ka;
    const int64_t seg0 = a.seg ? (int64_t)a.seg[w] * VAR_SEG : a.i_lo + w * VAR_SEG;  // wave-uniform
    const int64_t i = seg0 + lane;
    const bool active = i >= a.i_lo && i < a.i_hi && (!a.seg || ((a.segmask[w] >> lane) & 1ull));
    const int64_t safe = i < a.i_hi ? (i < a.i_lo ? a.i_lo : i) : a.i_hi - 1;
    double cA[15], cK[15], xk[15];
    bool nz[15];
    var_row_coefficients(a.T1, a.ld, i, active, safe, a.doff, cA);
    var_row_coefficients(a.T2, a.ld, i, active, safe, a.doff, cK);
#pragma unroll
    for (int k = 0; k < 15; ++k) nz[k] = (cA[k] != 0.0) | (cK[k] != 0.0);
    var_row_values(a.x, seg0, lane, active, safe, edge_off, nz, a.doff, xk);
    double s1 = 0.0, s2 = 0.0, se = 0.0;
#pragma unroll
    for (int k = 0; k < 15; ++k) {
      s1 = fma(cA[k], xk[k], s1);
      s2 = fma(cK[k], xk[k], s2);
    }
    if (a.x2 != nullptr) {  // A e: only where A itself has an entry
#pragma unroll
      for (int k = 0; k < 15; ++k) nz[k] = cA[k] != 0.0;
      var_row_values(a.x2, seg0, lane, active, safe, edge_off, nz, a.doff, xk);
#pragma unroll
      for (int k = 0; k < 15; ++k) se = fma(cA[k], xk[k], se);
    }
    if (active) {
      double stim = 0.0;
      for (int k = 0; k < a.nstim; ++k) stim = fma(a.amp[k], a.w[k][i], stim);
      const double r0 = a.dt * (stim - s2);
      const double b = s1 + r0;
      const double r = r0 - se;
      const double zz = a.dinv[i] * r;
      a.y[i] = r;
      a.y2[i] = zz;
      acc0 = fma(b, b, acc0);
      acc1 = fma(r, zz, acc1);
      acc2 = fma(r, r, acc2);
    }
  }
  const double s0 = beat_block_sum(acc0, red);
  const double s1 = beat_block_sum(acc1, red);
  const double s2 = beat_block_sum(acc2, red);
  if (threadIdx.x == 0) {
    a_.partials[a_.part_off + blockIdx.x] = s0;
    a_.partials[BEAT_MAX_PARTIALS + a_.part_off + blockIdx.x] = s1;
    a_.partials[2 * BEAT_MAX_PARTIALS + a_.part_off + blockIdx.x] = s2;
  }
}

// PCG vector updates over the active segments (per-node 1/diag)
__global__ __launch_bounds__(BEAT_BLOCK) void var_update_r_kernel(const int* __restrict__ seg,
    const unsigned long long* __restrict__ segmask, int nseg, int64_t n,
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
  for (int w = blockIdx.x * VAR_SEGS_PER_BLOCK + var_wave(); w < nseg; w += gridDim.x * VAR_SEGS_PER_BLOCK) {
    const int64_t i = (int64_t)seg[w] * VAR_SEG + threadIdx.x % VAR_SEG;
    if (i >= n || !((segmask[w] >> (threadIdx.x % VAR_SEG)) & 1ull)) continue;
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

__global__ __launch_bounds__(BEAT_BLOCK) void var_pupdate_oop_kernel(const int* __restrict__ seg,
    const unsigned long long* __restrict__ segmask, int nseg, int64_t n,
                                                                     const double* __restrict__ st,
                                                                     const double* __restrict__ r,
                                                                     const double* __restrict__ p_old,
                                                                     double* __restrict__ p_new,
                                                                     const double* __restrict__ dinv) {
  if (st[STOP] != 0.0) return;
  const double beta = st[BETA];
  for (int w = blockIdx.x * VAR_SEGS_PER_BLOCK + var_wave(); w < nseg; w += gridDim.x * VAR_SEGS_PER_BLOCK) {
    const int64_t i = (int64_t)seg[w] * VAR_SEG + threadIdx.x % VAR_SEG;
    if (i >= n || !((segmask[w] >> (threadIdx.x % VAR_SEG)) & 1ull)) continue;
    p_new[i] = fma(beta, p_old[i], dinv[i] * r[i]);
  }
}

__global__ __launch_bounds__(BEAT_BLOCK) void var_flush_kernel(const int* __restrict__ seg,
    const unsigned long long* __restrict__ segmask, int nseg, int64_t n,
                                                               const double* __restrict__ st, double* __restrict__ x,
                                                               const double* __restrict__ ring, int64_t fld,
                                                               const double* __restrict__ alphas, int ring_base,
                                                               int only_if_full, GuessTerms gt, int R) {
  int nvalid = (int)st[NUPD] - ring_base;
  nvalid = nvalid < 0 ? 0 : (nvalid > R ? R : nvalid);
  if ((only_if_full && nvalid < R) || (nvalid == 0 && gt.d == nullptr)) return;
  double a[PRING_MAX];
#pragma unroll
  for (int j = 0; j < PRING_MAX; ++j) a[j] = (j < nvalid) ? alphas[j] : 0.0;
  for (int w = blockIdx.x * VAR_SEGS_PER_BLOCK + var_wave(); w < nseg; w += gridDim.x * VAR_SEGS_PER_BLOCK) {
    const int64_t i = (int64_t)seg[w] * VAR_SEG + threadIdx.x % VAR_SEG;
    if (i >= n || !((segmask[w] >> (threadIdx.x % VAR_SEG)) & 1ull)) continue;
    if (gt.d != nullptr) {  // inc = e + sum alpha_j P_j;  x += inc;  (d, e) updated  (see x_flush_kernel)
      const double e_old = beat_guess_needs_e(gt) ? gt.e[i] : 0.0;
      const double d_old = beat_guess_needs_d(gt) ? gt.d[i] : 0.0;
      const double dp0 = beat_guess_needs_dp(gt, 0) ? gt.dp[0][i] : 0.0;
      const double dp1 = beat_guess_needs_dp(gt, 1) ? gt.dp[1][i] : 0.0;
      double inc = gt.accumulate ? 0.0 : e_old;
#pragma unroll
      for (int j = 0; j < PRING_MAX; ++j)
        if (j < nvalid) inc = fma(a[j], ring[(int64_t)j * fld + i], inc);
      x[i] += inc;
      beat_guess_record(gt, gt.d + i, gt.e + i, inc, d_old, dp0, dp1, e_old);
      continue;
    }
    double xi = x[i];
#pragma unroll
    for (int j = 0; j < PRING_MAX; ++j)
      if (j < nvalid) xi = fma(a[j], ring[(int64_t)j * fld + i], xi);
    x[i] = xi;
  }
}

// flags[s]: bit l set if node VAR_SEG s + l is touched by an element (mass diagonal > 0)
__global__ __launch_bounds__(BEAT_BLOCK) void var_segment_flags_kernel(int64_t n, const double* __restrict__ mass_diag,
                                                                       unsigned long long* __restrict__ flags) {
  const int64_t nsegs = (n + VAR_SEG - 1) / VAR_SEG;
  const int wave = var_wave(), lane = threadIdx.x % VAR_SEG;
  for (int64_t s = (int64_t)blockIdx.x * VAR_SEGS_PER_BLOCK + wave; s < nsegs; s += (int64_t)gridDim.x * VAR_SEGS_PER_BLOCK) {
    const int64_t i = s * VAR_SEG + lane;
    const bool mine = i < n && mass_diag[i] != 0.0;
    const unsigned long long any = __ballot(mine);
    if (lane == 0) flags[s] = any;
  }
}

// A = C_m Mass + theta dt K per node, 1/diag(A); rows without any element (inactive voxels) become identity.  B (or nullptr):
// the rows of C_m Mass - (1 - theta) dt K, the operator of the right-hand side b = B v_ + dt stim (beat_vtl_rhs)
__global__ __launch_bounds__(BEAT_BLOCK) void var_form_A_kernel(int64_t n, int64_t ld, const double* __restrict__ M,
                                                                const double* __restrict__ K, double cm, double tdt, double omt_dt,
                                                                double* __restrict__ A, double* __restrict__ dinv,
                                                                double* __restrict__ B) {
  const int64_t stride = (int64_t)gridDim.x * BEAT_BLOCK;
  for (int64_t i = (int64_t)blockIdx.x * BEAT_BLOCK + threadIdx.x; i < n; i += stride) {
    double d = cm * M[i] + tdt * K[i];
    const bool inactive = (M[i] == 0.0);
    if (inactive) d = 1.0;
    A[i] = d;
    dinv[i] = 1.0 / d;
    if (B != nullptr) B[i] = inactive ? 1.0 : cm * M[i] - omt_dt * K[i];
#pragma unroll
    for (int k = 1; k < 15; ++k) {
      const int64_t j = (int64_t)k * ld + i;
      A[j] = inactive ? 0.0 : cm * M[j] + tdt * K[j];
      if (B != nullptr) B[j] = inactive ? 0.0 : cm * M[j] - omt_dt * K[j];
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

using namespace beat_pde_detail;

// ---- variable-coefficient mode ------------------------------------------------------------------------
static void var_offsets(const beat_pde* pde, VarArgs& a) {
  const Geom& g = pde->g;
  for (int k = 0; k < 15; ++k)
    a.doff[k] = kOffsets[3 * k] + g.nx * kOffsets[3 * k + 1] + (int)g.plane * kOffsets[3 * k + 2];
  a.ld = pde->v_ld;
}

// Work of a launch over planes [z_lo, z_hi): the sub-list of active segments intersecting the range and the grid
// (= number of block partials the launch writes).
// Grid-stride kernels are launched with exactly the number of workgroups the chip holds at once (occupancy x CUs):
// every wave then sweeps the segment list in lockstep with its neighbours.  Measured on the 401^3 shell: SpMV 676 us
// with 4096 workgroups, 610 us with the 1536 resident ones, 687 / 918 us with 1280 / 1792.
template <class Kernel>
static unsigned resident_blocks(Kernel kernel) {
  int per_cu = 0, cus = 0, dev = 0;
  if (hipGetDevice(&dev) != hipSuccess ||
      hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, BEAT_BLOCK, 0) != hipSuccess ||
      hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || per_cu < 1 || cus < 1)
    return 4096;
  return (unsigned)std::min(per_cu * cus, BEAT_MAX_PARTIALS / 4);
}

struct VarRange {
  const int* seg;
  const unsigned long long* segmask;
  int nseg;
  unsigned grid;
  int tiled;
};
// BEAT_VAR_TILE = rows and planes per tile of the tile-ordered list (0: node order everywhere).  On since the kernels
// stopped serialising their loads (round 3): either change alone bought 0 - 2 % on the 17 M-node shell, both together 5 %
// of the step (tools/bench_biv.py --n 400: 13.2 -> 12.6 ms) -- the trips to memory and the L2 misses bounded the SpMV at
// the same level.  (A software pipeline over the list -- the next entry's coefficients requested behind the current
// entry's rows of p -- on top of both: 12.09 against 12.11 ms, not kept.)
static int var_tile_edge() {
  static const int t = [] {
    const char* e = getenv("BEAT_VAR_TILE");
    return e != nullptr ? std::max(0, atoi(e)) : 8;
  }();
  return t;
}
static VarRange var_range(const beat_pde* pde, int z_lo, int z_hi, bool dense) {
  VarRange r{nullptr, nullptr, 0, 0, 0};
  if (z_hi <= z_lo) return r;
  if (!dense && z_lo == 0 && z_hi == pde->g.nz && pde->v_seg_tiled != nullptr && pde->h_seg.size() >= 4096) {
    r.seg = pde->v_seg_tiled;
    r.segmask = pde->v_segmask_tiled;
    r.nseg = (int)pde->h_seg.size();
    r.grid = (unsigned)std::max<int64_t>(8, (r.nseg + VAR_SEGS_PER_BLOCK - 1) / VAR_SEGS_PER_BLOCK);
    r.tiled = 1;
    return r;
  }
  const int64_t i_lo = (int64_t)z_lo * pde->g.plane, i_hi = (int64_t)z_hi * pde->g.plane;
  int64_t nwork = (i_hi - i_lo + VAR_SEG - 1) / VAR_SEG;
  if (!dense) {
    const int s_lo = (int)(i_lo / VAR_SEG), s_hi = (int)((i_hi + VAR_SEG - 1) / VAR_SEG);
    const auto lo = std::lower_bound(pde->h_seg.begin(), pde->h_seg.end(), s_lo);
    const auto hi = std::lower_bound(pde->h_seg.begin(), pde->h_seg.end(), s_hi);
    r.seg = pde->v_seg + (lo - pde->h_seg.begin());
    r.segmask = pde->v_segmask + (lo - pde->h_seg.begin());
    r.nseg = (int)(hi - lo);
    nwork = r.nseg;
  }
  r.grid = (unsigned)std::max<int64_t>(1, (nwork + VAR_SEGS_PER_BLOCK - 1) / VAR_SEGS_PER_BLOCK);  // capped by the launcher
  return r;
}

template <int MODE>
static unsigned var_stencil_grid(unsigned wanted) {
  static const unsigned resident = [] {
    if constexpr (MODE == MODE_SPMV_DOT)
      return resident_blocks(var_spmv_kernel);
    else
      return resident_blocks(var_stencil_kernel<MODE>);
  }();
  return std::min(wanted, resident);
}

// BEAT_VAR_RHS_GATHER=1: the right-hand side with one gather per stencil point (var_stencil_kernel<MODE_RHS>, the kernel
// the row-shifting var_rhs_kernel is checked against bit for bit)
static bool var_rhs_by_gathers() {
  static const bool on = [] {
    const char* e = getenv("BEAT_VAR_RHS_GATHER");
    return e != nullptr && atoi(e) != 0;
  }();
  return on;
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
  a.segmask = r.segmask;
  a.nseg = r.nseg;
  a.tiled = r.tiled;
  auto whole_xcds = [&](unsigned grid) { return r.tiled ? std::max(8u, grid & ~7u) : grid; };  // tiled: blocks in eights
  if constexpr (MODE == MODE_SPMV_DOT) {
    const unsigned grid = whole_xcds(var_stencil_grid<MODE>(r.grid));
    BEAT_KERNEL(var_spmv_kernel, dim3(grid), dim3(BEAT_BLOCK), 0, pde->ctx->stream, a);
    return (int)grid;
  } else if (MODE == MODE_RHS && !var_rhs_by_gathers()) {
    static const unsigned resident = resident_blocks(var_rhs_kernel);
    const unsigned grid = whole_xcds(std::min(r.grid, resident));
    BEAT_KERNEL(var_rhs_kernel, dim3(grid), dim3(BEAT_BLOCK), 0, pde->ctx->stream, a);
    return (int)grid;
  }
  if constexpr (MODE != MODE_SPMV_DOT) {
    const unsigned grid = whole_xcds(var_stencil_grid<MODE>(r.grid));
    BEAT_KERNEL((var_stencil_kernel<MODE>), dim3(grid), dim3(BEAT_BLOCK), 0, pde->ctx->stream, a);
    return (int)grid;
  }
  return 0;
}

static unsigned var_vec_grid(const beat_pde* pde) {
  // the vector kernels use 12-20 VGPRs: eight waves per SIMD, i.e. eight workgroups per CU
  static const unsigned resident = resident_blocks(var_update_r_kernel);
  return (unsigned)std::min<size_t>(resident, std::max<size_t>(1, (pde->h_seg.size() + VAR_SEGS_PER_BLOCK - 1) / VAR_SEGS_PER_BLOCK));
}

int beat_var_form_A(beat_pde* pde) {
  pde->v_gc0_valid = false;  // the neighbours' centre coefficients change with theta dt as well
  const unsigned grid = (unsigned)std::min<int64_t>(4096, (pde->n + BEAT_BLOCK - 1) / BEAT_BLOCK);
  if (pde->v_B == nullptr && beat_vtl_rhs_wanted(pde)) {
    // the rows of B for the right-hand side on the tiles: 120 B/node more (401^3 box: 7.7 GB); without the memory the gather
    // kernel keeps building the right-hand side from the rows of K
    // Only with room to spare: the rows of B nearly double the operator's memory, and what the caller allocates
    // AFTER this point -- the cell model's state array (up to 52 rows: 416 B/node), the PCG's work fields (r, q, z + a ring of 12:
    // 120 B/node), the guess history (32 B/node) -- must still fit, or a grid that ran with the gather kernel would fail later for
    // the sake of 0.3 ms per step.  Taken when, after the 120 B/node of B, BEAT_VTL_RHS_RESERVE_B_PER_NODE (default 700) bytes per
    // node + 1 GiB stay free; BEAT_VTL_RHS_RESERVE_B_PER_NODE=0 restores "whenever the allocation itself succeeds".
    size_t free_b = 0, total_b = 0;
    const char* rs = std::getenv("BEAT_VTL_RHS_RESERVE_B_PER_NODE");
    const double reserve_per_node = rs != nullptr ? std::atof(rs) : 700.0;
    const size_t rows_b = sizeof(double) * 15 * (size_t)pde->v_ld;
    const size_t reserve_b = reserve_per_node > 0.0 ? (size_t)(reserve_per_node * (double)pde->v_ld) + ((size_t)1 << 30) : 0;
    const bool roomy = reserve_b == 0 || (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b >= rows_b + reserve_b);
    if (!roomy) (void)hipGetLastError();
    if (!roomy || hipMalloc(&pde->v_B, rows_b) != hipSuccess) {
      (void)hipGetLastError();
      pde->v_B = nullptr;
    }
  }
  BEAT_KERNEL(var_form_A_kernel, dim3(grid), dim3(BEAT_BLOCK), 0, pde->ctx->stream, pde->n, pde->v_ld,
                     pde->v_mass, pde->v_stiff, pde->C_m, pde->theta * pde->dt, (1.0 - pde->theta) * pde->dt, pde->v_A, pde->v_dinv, pde->v_B);
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
  // a single slab keeps 12 search directions before x is brought up to date (see PRING_MAX; BEAT_VAR_RING=6: as the other paths)
  if (z_lo_phys && z_hi_phys) {
    const char* e = std::getenv("BEAT_VAR_RING");
    p->ring = (e && std::atoi(e) == PRING) ? PRING : PRING_MAX;
  }
  if (hipMalloc(&p->v_A, sizeof(double) * 15 * (size_t)ld) != hipSuccess ||
      hipMalloc(&p->v_dinv, sizeof(double) * (size_t)ld) != hipSuccess ||
      (!(z_lo_phys && z_hi_phys) && hipMalloc(&p->v_gc0, sizeof(double) * 2 * (size_t)(n[0] * n[1])) != hipSuccess)) {
    beat_pde_destroy(p);
    beat_set_error("out of device memory for the %lld-node coefficient rows", (long long)ld);
    return BEAT_EHIP;
  }
  // list of the segments (VAR_SEG consecutive nodes) that hold tissue nodes
  const int64_t nsegs = (p->n + VAR_SEG - 1) / VAR_SEG;
  unsigned long long* d_flags = nullptr;
  std::vector<unsigned long long> flags((size_t)nsegs), masks;
  hipError_t e = hipMalloc(&d_flags, sizeof(unsigned long long) * (size_t)nsegs);
  if (e == hipSuccess) {
    BEAT_KERNEL(var_segment_flags_kernel, dim3((unsigned)std::min<int64_t>(4096, (nsegs + VAR_SEGS_PER_BLOCK - 1) / VAR_SEGS_PER_BLOCK)),
                       dim3(BEAT_BLOCK), 0, ctx->stream, p->n, dev_mass, d_flags);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(flags.data(), d_flags, sizeof(unsigned long long) * (size_t)nsegs, hipMemcpyDeviceToHost,
                                          ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  (void)hipFree(d_flags);
  if (e == hipSuccess) {
    for (int64_t sidx = 0; sidx < nsegs; ++sidx)
      if (flags[(size_t)sidx]) {
        p->h_seg.push_back((int)sidx);
        masks.push_back(flags[(size_t)sidx]);
      }
    e = hipMalloc(&p->v_seg, sizeof(int) * std::max<size_t>(1, p->h_seg.size()));
  }
  if (e == hipSuccess) e = hipMalloc(&p->v_segmask, sizeof(unsigned long long) * std::max<size_t>(1, masks.size()));
  if (e == hipSuccess && !p->h_seg.empty())
    e = hipMemcpy(p->v_seg, p->h_seg.data(), sizeof(int) * p->h_seg.size(), hipMemcpyHostToDevice);
  if (e == hipSuccess && !masks.empty())
    e = hipMemcpy(p->v_segmask, masks.data(), sizeof(unsigned long long) * masks.size(), hipMemcpyHostToDevice);
  if (e == hipSuccess && p->g.nz > 1 && !p->h_seg.empty() && var_tile_edge() > 0) {
    // the same list in tile order: tiles of T rows x T planes (keyed by the segment's first node), node order inside
    const int TY = var_tile_edge(), TZ = var_tile_edge();
    const int64_t nyb = (p->g.ny + TY - 1) / TY;
    std::vector<int64_t> key(p->h_seg.size());
    std::vector<int> order(p->h_seg.size());
    for (size_t k = 0; k < p->h_seg.size(); ++k) {
      const int64_t i0 = (int64_t)p->h_seg[k] * VAR_SEG;
      const int64_t z = i0 / p->g.plane, y = (i0 % p->g.plane) / p->g.nx;
      key[k] = (z / TZ) * nyb + y / TY;
      order[k] = (int)k;
    }
    std::stable_sort(order.begin(), order.end(), [&](int u, int v) { return key[(size_t)u] < key[(size_t)v]; });
    std::vector<int> seg_t(order.size());
    std::vector<unsigned long long> mask_t(order.size());
    for (size_t k = 0; k < order.size(); ++k) {
      seg_t[k] = p->h_seg[(size_t)order[k]];
      mask_t[k] = masks[(size_t)order[k]];
    }
    e = hipMalloc(&p->v_seg_tiled, sizeof(int) * seg_t.size());
    if (e == hipSuccess) e = hipMalloc(&p->v_segmask_tiled, sizeof(unsigned long long) * mask_t.size());
    if (e == hipSuccess) e = hipMemcpy(p->v_seg_tiled, seg_t.data(), sizeof(int) * seg_t.size(), hipMemcpyHostToDevice);
    if (e == hipSuccess)
      e = hipMemcpy(p->v_segmask_tiled, mask_t.data(), sizeof(unsigned long long) * mask_t.size(), hipMemcpyHostToDevice);
  }
  if (e != hipSuccess) {
    beat_pde_destroy(p);
    beat_set_error("beat_pde_create_var: %s", hipGetErrorString(e));
    return BEAT_EHIP;
  }
  if ((rc = beat_vrr_setup(p, flags)) || (rc = beat_vtl_setup(p, flags))) {
    beat_pde_destroy(p);
    return rc;
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
  BEAT_KERNEL(rows_dirichlet_kernel, dim3(grid), dim3(BEAT_BLOCK), 0, ctx->stream, nn, ld, dev_rows, dev_flag,
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
    BEAT_KERNEL(assemble_rows_kernel, dim3(grid), dim3(BEAT_BLOCK), 0, ctx->stream, a);
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

int beat_var_apply(beat_pde* pde, int which, const double* dev_x, double* dev_y) {
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

int beat_var_rhs(beat_pde* pde, const double* dev_v_prev, const double* const* host_dev_stim_w,
                 const double* host_stim_amp, int n_stim, double* dev_x, double* dev_r, double* dev_p, double* dev_red,
                 const double* dev_e, int part) {
  VarArgs a{};
  var_offsets(pde, a);
  a.T1 = pde->v_A;
  a.T2 = pde->v_stiff;
  a.x = dev_v_prev;
  a.x2 = dev_e;
  a.y = dev_r;
  a.y2 = dev_p;
  a.y3 = nullptr;
  if (dev_x != dev_v_prev && part <= 0)  // nodes outside the tissue keep their value: copy everything first
    BEAT_HIP_CHECK(hipMemcpyAsync(dev_x, dev_v_prev, sizeof(double) * (size_t)pde->n, hipMemcpyDeviceToDevice,
                                  pde->ctx->stream));
  a.dinv = pde->v_dinv;
  a.dt = pde->dt;
  for (int k = 0; k < n_stim; ++k) {
    if (host_dev_stim_w[k] == nullptr || host_stim_amp[k] == 0.0) continue;
    a.w[a.nstim] = host_dev_stim_w[k];
    a.amp[a.nstim] = host_stim_amp[k];
    ++a.nstim;
  }
  a.partials = pde->ctx->d_partials;
  if (part < 0) {
    const int nb = launch_var<MODE_RHS>(pde, a, 0, pde->g.nz, 0);
    BEAT_LAUNCH_CHECK();
    return beat_pde_launch_reduce(pde, nb, 3, dev_red, nullptr);
  }
  // in two parts on a decomposed grid (as beat_var_spmv_dot_part): the planes that need no ghost plane of v_ / e while
  // those travel, then the slab-boundary planes and the reduction over all block partials
  const Geom& f = pde->g;
  const int lo = f.z_lo_phys ? 0 : 1, hi = f.nz - (f.z_hi_phys ? 0 : 1);
  if (part == 0) {
    pde->rhs_part_blocks = launch_var<MODE_RHS>(pde, a, lo, std::max(lo, hi), 0);
    BEAT_LAUNCH_CHECK();
    return BEAT_OK;
  }
  int off = pde->rhs_part_blocks;
  if (!f.z_lo_phys) off += launch_var<MODE_RHS>(pde, a, 0, 1, off);
  if (!f.z_hi_phys && (f.nz > 1 || f.z_lo_phys)) off += launch_var<MODE_RHS>(pde, a, f.nz - 1, f.nz, off);
  BEAT_LAUNCH_CHECK();
  BEAT_REQUIRE(off <= BEAT_MAX_PARTIALS, "too many block partials");
  return beat_pde_launch_reduce(pde, off, 3, dev_red, nullptr);
}

int beat_var_spmv_dot(beat_pde* pde, const double* dev_p, double* dev_q, double* dev_st) {
  if (beat_vrr_available(pde)) return beat_vrr_spmv_dot(pde, dev_p, dev_q, dev_st, -1);
  if (beat_vtl_available(pde)) return beat_vtl_spmv_dot(pde, dev_p, dev_q, dev_st);
  VarArgs a{};
  var_offsets(pde, a);
  a.T1 = pde->v_A;
  a.x = dev_p;
  a.y = dev_q;
  a.partials = pde->ctx->d_partials;
  a.st = dev_st;
  const int nb = launch_var<MODE_SPMV_DOT>(pde, a, 0, pde->g.nz, 0);
  BEAT_LAUNCH_CHECK();
  return beat_pde_launch_reduce(pde, nb, 1, dev_st + PQ, dev_st);
}

int beat_var_spmv_dot_part(beat_pde* pde, const double* dev_p, double* dev_q, double* dev_st, int part) {
  if (beat_vrr_available(pde)) return beat_vrr_spmv_dot(pde, dev_p, dev_q, dev_st, part);
  if (beat_vtl_parts_available(pde)) return beat_vtl_spmv_dot_part(pde, dev_p, dev_q, dev_st, part);
  const Geom& f = pde->g;
  const int lo = f.z_lo_phys ? 0 : 1, hi = f.nz - (f.z_hi_phys ? 0 : 1);  // planes that need no ghost data
  VarArgs a{};
  var_offsets(pde, a);
  a.T1 = pde->v_A;
  a.x = dev_p;
  a.y = dev_q;
  a.partials = pde->ctx->d_partials;
  a.st = dev_st;
  // block-partial slots: the interior launch's come first, the boundary planes follow
  if (part == 0) {
    launch_var<MODE_SPMV_DOT>(pde, a, lo, std::max(lo, hi), 0);
    BEAT_LAUNCH_CHECK();
    return BEAT_OK;
  }
  int off = (int)var_stencil_grid<MODE_SPMV_DOT>(var_range(pde, lo, std::max(lo, hi), false).grid);  // partial slots of the interior launch (part 0)
  if (!f.z_lo_phys) off += launch_var<MODE_SPMV_DOT>(pde, a, 0, 1, off);
  if (!f.z_hi_phys && (f.nz > 1 || f.z_lo_phys)) off += launch_var<MODE_SPMV_DOT>(pde, a, f.nz - 1, f.nz, off);
  BEAT_LAUNCH_CHECK();
  BEAT_REQUIRE(off <= BEAT_MAX_PARTIALS, "too many block partials");
  return beat_pde_launch_reduce(pde, off, 1, dev_st + PQ, dev_st);
}

int beat_var_update_r(beat_pde* pde, double* dev_st, double* dev_r, const double* dev_q, int slot, bool roll) {
  const unsigned grid = var_vec_grid(pde);
  BEAT_KERNEL(var_update_r_kernel, dim3(grid), dim3(BEAT_BLOCK), 0, pde->ctx->stream, (const int*)pde->v_seg, (const unsigned long long*)pde->v_segmask,
                     (int)pde->h_seg.size(), pde->n, (const double*)dev_st, dev_r, dev_q, (const double*)pde->v_dinv,
                     pde->ctx->d_partials, pde->d_alphas, slot);
  BEAT_LAUNCH_CHECK();
  return beat_pde_launch_reduce(pde, (int)grid, 2, dev_st + RZN, dev_st, dev_st + NUPD, roll ? 1 : 0, dev_st);  // (roll: the scalar step in the same launch)
}

int beat_var_pupdate_oop(beat_pde* pde, double* dev_st, const double* dev_r, const double* dev_p_cur, double* dev_p_next) {
  BEAT_KERNEL(var_pupdate_oop_kernel, dim3(var_vec_grid(pde)), dim3(BEAT_BLOCK), 0, pde->ctx->stream,
                     (const int*)pde->v_seg, (const unsigned long long*)pde->v_segmask, (int)pde->h_seg.size(), pde->n, (const double*)dev_st, dev_r, dev_p_cur,
                     dev_p_next, (const double*)pde->v_dinv);
  BEAT_LAUNCH_CHECK();
  return BEAT_OK;
}

int beat_var_flush(beat_pde* pde, const double* dev_st, double* dev_x, const double* dev_ring0, int64_t field_stride,
                   int ring_base, int only_if_full, const GuessTerms& gt) {
  BEAT_KERNEL(var_flush_kernel, dim3(var_vec_grid(pde)), dim3(BEAT_BLOCK), 0, pde->ctx->stream,
                     (const int*)pde->v_seg, (const unsigned long long*)pde->v_segmask, (int)pde->h_seg.size(), pde->n, dev_st, dev_x, dev_ring0, field_stride,
                     (const double*)pde->d_alphas, ring_base, only_if_full, gt, pde->ring);
  BEAT_LAUNCH_CHECK();
  return BEAT_OK;
}
