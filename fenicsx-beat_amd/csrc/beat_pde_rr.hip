// Constant-coefficient diffusion solve without a stored q = A p: "register-row" stencil kernels.
//
// The 15-point P1 stencil touches 7 rows of x (3 of the plane z, 2 of z+1, 2 of z-1).  A wave owns 64 consecutive x-nodes from
// a multiple of 64 (the iteration's passes since round 5: the x-halo comes with one extra load per plane, see BEAT_RR_ALIGN; the
// right-hand side keeps the original form: 62 nodes, lanes 0 and 63 carrying the x-halo) of RY = 4 consecutive rows and marches along z with the
// 3 x 6 row values it needs held in registers: every row is loaded ONCE per wave (one coalesced 512 B request),
// the x +- 1 neighbours come from the adjacent lanes by DPP wave shifts (v_mov_b32_dpp wave_shr:1 / wave_shl:1, no
// LDS, no barriers), and the plane z+2 is in flight while plane z is computed.  The y-halo rows (6 rows loaded
// per 4 rows computed) are re-reads that the XCD's L2 serves.
//
// With the operator this cheap to re-apply, the PCG never stores q:
//   K_A  (pdot):  p_new = D^-1 r + beta p_old  formed while loading, stored once;  partial  p_new . (A p_new)
//   K_B  (rupd):  alpha = r.z / p.q;  r -= alpha (A p)  with A p recomputed from p;  partials  r.D^-1 r, r.r
// i.e. 24 + 24 B/node per iteration instead of 16 (SpMV) + 24 (r update) + 24 (p update) = 64, and the right-hand
// side writes r only (16 B instead of 24; the first K_A forms p_0 = D^-1 r itself).  Values are those of the
// classic three-kernel loop up to the order in which the block partial sums are added.
// A decomposed solve may ask for the single-reduction form instead (K_U / K_P: beat_rr_udot_part below).
//
// Replaces, like beat_pde.hip, dolfinx assemble_vector + PETSc KSP.solve of src/beat/base_model.py:196-236.
#include "beat_pde_internal.h"

#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdlib>

namespace {
using namespace beat_pde_detail;

constexpr int SEG = 62;  // x-nodes computed per wave and row where lanes 0 and 63 carry the x-halo (the right-hand side; BEAT_RR_ALIGN=0)
// rows computed per wave: template parameter RY (2 or 4; BEAT_RR_RY, default 4), NR = RY + 2 rows held per plane;
// PD: planes fetched ahead of their use (BEAT_RR_PD)

struct Coef {
  double c[15];
};

struct RGeom {
  int nx, ny, nz;
  int64_t plane;
  int ry;
  int nsegx, nrb, nchunks, zc;
  int total_waves, total_blocks;
  int z_lo_phys, z_hi_phys;
  int z_lo, z_hi;                  // planes [z_lo, z_hi) computed by this launch (whole slab: 0, nz)
  int part_off;                    // first block-partial slot this launch writes
  int ghost_lo_tz, ghost_hi_tz;    // z node type (0 low face, 1 interior, 2 high face) of the ghost planes -1 / nz
  int nrg;                         // > 0: the four waves of a block own four ADJACENT row blocks of one x segment (nrg groups of 4 row
                                   // blocks per plane): the y-halo rows they share are fetched within one workgroup, in step (see make_geom)
};

// RR_UDOT / RR_PRUPD: the single-reduction iteration of a decomposed solve (BEAT_DIST_MERGED, see beat_rr_udot_part)
enum { RR_PDOT = 0, RR_RUPD = 1, RR_RHS = 2, RR_UDOT = 3, RR_PRUPD = 4 };

struct RArgs {
  const double* x;      // PDOT, UDOT, PRUPD: r | RUPD: p | RHS: v_
  const double* x2;     // PDOT, PRUPD: p_old | RUPD: r
  double* y;            // PDOT, PRUPD: p_new | RUPD: r_new (another buffer than r) | RHS: r
  double* y2;           // RHS: x (copy of v_) or nullptr | PRUPD: r_new (another buffer than r)
  const double* tab;    // A table (PDOT, RUPD) | Mass table (RHS)
  const double* tab2;   // RHS: K table
  const double* dinv;   // 1/diag(A) per node type
  Coef ci, ci2;         // interior rows of tab / tab2
  double dinv_i;
  double cm, omt_dt, dt;
  const double* w[BEAT_MAX_STIM];
  double amp[BEAT_MAX_STIM];
  int nstim;
  double* partials;
  double* st;           // PCG scalar state
  double* alphas;       // RUPD: step lengths of the deferred-x ring
  int slot;
  // RHS with an extrapolated initial guess x0 = v_ + e:  r = b - A x0 = dt (stim - K v_) - A e;  taba / cia: the A table
  const double* e;
  const double* taba;
  Coef cia;
};

// the kernel-argument segment of rr_kernel as the hardware lays it out (every member naturally aligned): the right-hand-side
// instances read their coefficient rows and scalars through an opaque pointer to it, per plane (see the kernel)
struct RKernArgs {
  RGeom g;
  RArgs a;
  const double* X;
  const double* X2;
  double* Y;
  double* Y2;
};

// lane i <- lane i-1 (lane 0 <- 0) / lane i <- lane i+1 (lane 63 <- 0); all 64 lanes must be active
__device__ __forceinline__ double from_left(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x138, 0xf, 0xf, true);  // wave_shr:1
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x138, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double from_right(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x130, 0xf, 0xf, true);  // wave_shl:1
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x130, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}

// BEAT_RR_BUF (round 5; 0 = the clamped global loads of rounds 2 - 4): the rows of a plane through a raw buffer over that plane -- a lane
// that is to hold 0 (outside the box, a plane beyond a physical face) passes an offset beyond the buffer's end (or the plane gets no
// records): the load returns 0 and fetches nothing, and neither the clamped address nor the select that zeroed the value afterwards
// is needed.  1 = every mode, 2 = every mode but the right-hand side (the build).  Measured on one box, alternating
// (profiles/r05_rr_resources.md): PDOT 776 - 779 -> 742 us (162 -> 156 VGPRs), RUPD 715 - 718 -> 719 (174 -> 162 VGPRs, 2 -> 3 waves per
// SIMD: no effect), the right-hand side 951 - 985 -> 1002 - 1018 with them (hence 2); the diffusion part of the 512^3 step 3.65 - 3.68 ->
// 3.62 ms, of the developed front's 5.56 - 5.58 -> 5.48 - 5.49.
#ifndef BEAT_RR_BUF
#define BEAT_RR_BUF 2
#endif
typedef int rr_v2i __attribute__((ext_vector_type(2)));
[[maybe_unused]] constexpr unsigned RR_OOB = 0x80000000u;
// BEAT_RR_NT (round 6): bit 0 non-temporal row loads, bit 1 non-temporal stores of the pass's output field.  2 is the build: the
// output field is written once and read by the NEXT launch (1 GB at 512^3: from HBM either way), and not keeping it in the L2 leaves
// the L2 to the y-halo rows neighbouring waves re-read -- diffusion part of the 512^3 step 3.38 - 3.42 -> 3.31 - 3.36 ms, of the developed
// front's 5.06 - 5.12 -> 4.93 - 5.01; non-temporal LOADS throw exactly those rows away: 3.84 - 3.88 / 6.12 - 6.16 (profiles/r06_ab_rr_nt.txt)
#ifndef BEAT_RR_NT
#define BEAT_RR_NT 2
#endif
__device__ __forceinline__ void rr_store(double* p, double v) {
#if BEAT_RR_NT & 2
  __builtin_nontemporal_store(v, p);
#else
  *p = v;
#endif
}
__device__ __forceinline__ double rr_buf_load(const double* base, unsigned bytes, unsigned off) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000);
  const rr_v2i v = __builtin_amdgcn_raw_buffer_load_b64(r, (int)off, 0, (BEAT_RR_NT & 1) ? 2 : 0);
  return __hiloint2double(v.y, v.x);
}

// BEAT_RR_ALIGN = 1 (round 5): in PDOT and RUPD (and the two passes of the single-reduction iteration) a wave owns 64 x-nodes that start on a multiple of 64 (whole aligned 512-byte pieces of
// a row) instead of 62 that start 8 bytes before a multiple of 496.  The access pattern alone -- six rows of a plane read, four written,
// no arithmetic -- streams at 4.2 - 4.3 TB/s in the 62-node form and at 5.0 - 5.1 TB/s in the aligned form INCLUDING what it needs
// instead of the two halo lanes (tools/march_probe.py, profiles/r05_march_probe.md): ONE more load per plane in which lanes 0 .. NR-1
// fetch the element left of the segment in rows 0 .. NR-1 and lanes 8 .. 8+NR-1 the element right of it (every other lane passes an
// out-of-range offset), kept in one register pair per plane of the window; the lane shifts put that element into lane 0 / lane 63
// (v_readlane, then the shift keeps it where a lane has no source).  Needs the raw-buffer loads (BEAT_RR_BUF).
#ifndef BEAT_RR_ALIGN
#define BEAT_RR_ALIGN 1
#endif
// lane i <- lane i-1, lane 0 <- lane `src_lane` of `h` / lane i <- lane i+1, lane 63 <- lane `src_lane` of `h`: the lane without a
// source keeps the shift's `old` operand, which is that element broadcast (v_readlane, v_mov, v_mov_dpp per half)
__device__ __forceinline__ double from_left_h(double v, double h, int src_lane) {
  const int lo = __builtin_amdgcn_update_dpp(__builtin_amdgcn_readlane(__double2loint(h), src_lane), __double2loint(v), 0x138, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(__builtin_amdgcn_readlane(__double2hiint(h), src_lane), __double2hiint(v), 0x138, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double from_right_h(double v, double h, int src_lane) {
  const int lo = __builtin_amdgcn_update_dpp(__builtin_amdgcn_readlane(__double2loint(h), src_lane), __double2loint(v), 0x130, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(__builtin_amdgcn_readlane(__double2hiint(h), src_lane), __double2hiint(v), 0x130, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
// (the same condition on the host, where the geometry of a launch is made: make_geom)
#ifndef BEAT_RR_ALIGN_RHS
#define BEAT_RR_ALIGN_RHS 0  // 1: the right-hand side on aligned segments too (with raw-buffer loads and a halo element per window); measured, see profiles/r05_rr_resources.md
#endif
constexpr bool rr_aligned_mode(int mode) {
  return BEAT_RR_ALIGN == 1 && BEAT_RR_BUF != 0 && (mode == 0 || mode == 1 || mode == 3 || mode == 4 || (mode == 2 && BEAT_RR_ALIGN_RHS == 1));
}

__device__ __forceinline__ int axis_type3(int i, int n, int lo_phys, int hi_phys) {
  if (n == 1 && lo_phys && hi_phys) return 1;  // collapsed axis: no coupling along it
  if (i == 0 && lo_phys) return 0;
  if (i == n - 1 && hi_phys) return 2;
  return 1;
}

__device__ __forceinline__ int xcd_block(int b, int total) {
  const int per = (total + 7) >> 3;
  return (b & 7) * per + (b >> 3);
}

// waves per SIMD asked of the register allocator: the right-hand side with a guess (two register windows) / the iteration's two passes
#ifndef BEAT_RR_WAVES_RHS
#define BEAT_RR_WAVES_RHS 3
#endif
#ifndef BEAT_RR_WAVES_IT
#define BEAT_RR_WAVES_IT 1
#endif
// X / X2 / Y / Y2: the fields of RArgs::x, x2 (RHS with a guess: e), y, y2 as separate restrict-qualified kernel parameters -- the launch
// passes distinct buffers (the residual update writes r out of place), and without the no-alias guarantee every
// store would have to complete (s_waitcnt vmcnt(0)) before the next plane's loads may issue.
template <int MODE, int RY, int PD, bool GUESS = false>
__global__ __launch_bounds__(BEAT_BLOCK, (GUESS && RY == 2 && PD == 1) ? BEAT_RR_WAVES_RHS : ((MODE == RR_PDOT || MODE == RR_RUPD) && PD == 1 ? BEAT_RR_WAVES_IT : 1)) void rr_kernel(RGeom g, RArgs a, const double* __restrict__ X,
                                                        const double* __restrict__ X2, double* __restrict__ Y,
                                                        double* __restrict__ Y2) {
  constexpr int NR = RY + 2;
  constexpr int NE = GUESS ? NR : 1;  // rows of the second register window (the guess increment e)
  constexpr bool FORM = MODE == RR_PDOT || MODE == RR_UDOT || MODE == RR_PRUPD;  // the staged value is formed from r
  constexpr bool OLD = MODE == RR_PDOT || MODE == RR_PRUPD;                      // ... with beta p_old, and stored
  constexpr bool RAW = MODE == RR_UDOT || MODE == RR_PRUPD;                      // r itself of the owned rows is kept too
  constexpr int NW = RAW ? RY : 1;
  constexpr bool BUF = BEAT_RR_BUF == 1 || (BEAT_RR_BUF == 2 && (MODE != RR_RHS || BEAT_RR_ALIGN_RHS == 1));  // raw-buffer loads of the rows (see BEAT_RR_BUF)
  constexpr bool ALIGNED = rr_aligned_mode(MODE);  // 64 aligned x-nodes per wave, the x-halo from one more load per plane (see BEAT_RR_ALIGN)
  static_assert(!ALIGNED || BUF, "aligned segments come with the raw-buffer loads");
  constexpr int SEGW = ALIGNED ? 64 : SEG;
  static_assert(!GUESS || MODE == RR_RHS, "the initial guess enters the right-hand side only");
  __shared__ double red[4];
  // boundary rows of the coefficient tables and 1/diag per node type, staged in LDS: the lanes on a face of the box
  // look them up every step, and a global load there would put a full vmcnt(0) drain -- outstanding stores and the
  // prefetched plane included -- into the march
  __shared__ double s_dinv[32];
  __shared__ double s_tab[27 * TABW];
  __shared__ double s_tab2[MODE == RR_RHS ? 27 * TABW : 1];
  __shared__ double s_taba[GUESS ? 27 * TABW : 1];
  if (MODE != RR_RHS) {
    if (a.st[STOP] != 0.0) return;  // convergence latch (uniform over the grid)
  }
  for (int i = threadIdx.x; i < 27 * TABW; i += BEAT_BLOCK) {
    s_tab[i] = a.tab[i];
    if (MODE == RR_RHS) s_tab2[i] = a.tab2[i];
    if (GUESS) s_taba[i] = a.taba[i];
  }
  if (threadIdx.x < 27) s_dinv[threadIdx.x] = a.dinv[threadIdx.x];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int blk = xcd_block(blockIdx.x, g.total_blocks);
  const int w = blk * 4 + (threadIdx.x >> 6);
  int seg = w % g.nsegx;
  int rb = (w / g.nsegx) % g.nrb;
  int chunk = w / (g.nsegx * g.nrb);
  bool wave_ok = blk < g.total_blocks && w < g.total_waves;
  if (g.nrg > 0) {
    seg = blk % g.nsegx;
    rb = ((blk / g.nsegx) % g.nrg) * 4 + (int)(threadIdx.x >> 6);
    chunk = blk / (g.nsegx * g.nrg);
    wave_ok = blk < g.total_blocks && rb < g.nrb;
  }
  const int gx = seg * SEGW + (ALIGNED ? 0 : -1) + lane;
  const int y0 = rb * RY - 1;  // global row of register row 0
  const int zb = g.z_lo + chunk * g.zc;
  const int ze = wave_ok ? min(zb + g.zc, g.z_hi) : zb;
  const bool x_in = wave_ok && gx >= 0 && gx < g.nx;
  const bool x_out = x_in && (ALIGNED || (lane >= 1 && lane <= SEG));  // this lane produces outputs
  const int tx = axis_type3(gx, g.nx, 1, 1);
  bool row_in[NR];
  int txy[NR];  // tx + 3 ty of the register rows
  int off[NR];  // in-plane offset of the (clamped, always addressable) element this lane loads of each row
  const int cx = min(max(gx, 0), g.nx - 1);
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const int gy = y0 + r;
    row_in[r] = x_in && gy >= 0 && gy < g.ny;
    txy[r] = tx + 3 * axis_type3(gy, g.ny, 1, 1);
    off[r] = min(max(gy, 0), g.ny - 1) * g.nx + cx;
    if constexpr (BUF) off[r] = row_in[r] ? (int)((unsigned)(gy * g.nx + gx) * 8u) : (int)RR_OOB;  // byte offset within the plane, or out of range
  }
  [[maybe_unused]] const unsigned pbytes = (unsigned)(g.plane * 8);
  // ALIGNED: what this lane fetches in the x-halo load of a plane (lanes 0 .. NR-1: left of the segment in rows 0 .. NR-1, lanes
  // 8 .. 8+NR-1: right of it), and the node type of that element (PDOT forms p_new there as everywhere)
  [[maybe_unused]] unsigned hoff = RR_OOB;
  [[maybe_unused]] int htxy = 13 % 9;
  if constexpr (ALIGNED) {
    const int hr = lane < 8 ? lane : lane - 8;
    const int hx = lane < 8 ? seg * SEGW - 1 : seg * SEGW + SEGW;
    const int hy = y0 + hr;
    const bool hv = wave_ok && (lane < NR || (lane >= 8 && lane < 8 + NR)) && hx >= 0 && hx < g.nx && hy >= 0 && hy < g.ny;
    hoff = hv ? (unsigned)(hy * g.nx + hx) * 8u : RR_OOB;
    htxy = axis_type3(min(max(hx, 0), g.nx - 1), g.nx, 1, 1) + 3 * axis_type3(min(max(hy, 0), g.ny - 1), g.ny, 1, 1);
  }
  double beta = 0.0, alpha = 0.0;
  if (OLD) beta = a.st[BETA];
  if (MODE == RR_PRUPD) alpha = a.st[ALPHA];
  if (MODE == RR_RUPD) {
    alpha = a.st[RZ] / a.st[PQ];
    if (blockIdx.x == 0 && threadIdx.x == 0) a.alphas[a.slot] = alpha;
  }
  const bool have_old = beta != 0.0;  // first direction: p_old may hold anything and is not read

  // Loads are unconditional (no divergent branches around them): out-of-box lanes read a clamped, valid element
  // and the value is replaced by 0 afterwards; the planes -1 and nz are the ghost planes every field carries.
  // Plane k is fetched PD steps before the step that first needs it (as plane z+1 of step z = k-1) into raw slot
  // (step index) % PD; the loop is unrolled by PD so that the slots are static registers.
  double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0;
  double Cm[NR], C0[NR], Cp[NR], ra[PD][NR], rb2[PD][NR];
  double rvn[PD][RY];
  [[maybe_unused]] double Hm = 0.0, H0 = 0.0, Hp = 0.0, rha[PD], rhb[PD];  // ALIGNED: the x-halo elements of the three planes, raw values in flight
  [[maybe_unused]] double HEm = 0.0, HE0 = 0.0, HEp = 0.0;              // ... and of the second window (GUESS: x0 = v_ + e; rhb holds e's)
#pragma unroll
  for (int u = 0; u < PD; ++u) rha[u] = rhb[u] = 0.0;
  double Rw0[NW], Rwp[NW];  // RAW: r of the owned rows on the planes z and z+1
#pragma unroll
  for (int j = 0; j < NW; ++j) Rw0[j] = Rwp[j] = 0.0;
  // GUESS: the same window of e (fetched with the plane of v_, staged with it)
  double Em[NE], E0[NE], Ep[NE], re1[PD][NE];
#pragma unroll
  for (int r = 0; r < NE; ++r) {
    Em[r] = E0[r] = Ep[r] = 0.0;
#pragma unroll
    for (int u = 0; u < PD; ++u) re1[u][r] = 0.0;
  }
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    Cm[r] = C0[r] = Cp[r] = 0.0;
#pragma unroll
    for (int u = 0; u < PD; ++u) ra[u][r] = rb2[u][r] = 0.0;
  }
#pragma unroll
  for (int u = 0; u < PD; ++u)
#pragma unroll
    for (int j = 0; j < RY; ++j) rvn[u][j] = 0.0;
  // The march starts PD + 2 steps early: those steps only fill the pipeline (planes z_b-1, z_b, ...).
  for (int zz = zb - 2 - PD; zz < ze; zz += PD) {
#pragma unroll
  for (int u = 0; u < PD; ++u) {
    const int z = zz + u;
    if (z < ze) {
    const int k = z + 1;  // plane whose raw values (slot u) become the staged plane z+1 now
    if (k >= zb - 1 && k <= ze) {  // staged values of plane k; PDOT forms p_new here and stores the rows this wave owns
      [[maybe_unused]] const bool zok = (k >= 0 || !g.z_lo_phys) && (k < g.nz || !g.z_hi_phys);
      // ghost planes (another rank's boundary planes): their z type comes with the geometry; PDOT keeps p_new there
      // too -- both neighbours form it from the same exchanged r and the same p_old, so it needs no exchange of its own
      const int tz9 = 9 * (k < 0 ? g.ghost_lo_tz : k >= g.nz ? g.ghost_hi_tz : axis_type3(k, g.nz, g.z_lo_phys, g.z_hi_phys));
      const bool own_plane = (k >= zb && k < ze) || (k == -1 && zb == 0 && !g.z_lo_phys) ||
                             (k == g.nz && ze == g.nz && !g.z_hi_phys);
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        double c = ra[u][r];
        if (RAW && r >= 1 && r <= RY) Rwp[RAW ? r - 1 : 0] = c;
        if (FORM) {
          const int type = txy[r] + tz9;
          const double di = s_dinv[type];  // (a select against the kernel argument a.dinv_i trips an LDS-pointer cast bug in hipcc 7.2)
          // p_new = D^-1 r + beta p_old (same expression as cg_pupdate_oop_kernel; p_old unread while beta = 0)
          c = have_old ? fma(beta, rb2[u][r], di * ra[u][r]) : di * ra[u][r];
        }
        if constexpr (!BUF) c = (zok && row_in[r]) ? c : 0.0;
        Cp[r] = c;
        if (GUESS) {
          constexpr int q = GUESS ? 1 : 0;
          if constexpr (BUF)
            Ep[r * q] = c + re1[u][r * q];  // x0 = v_ + e (both 0 where nothing was loaded)
          else
            Ep[r * q] = (zok && row_in[r]) ? c + re1[u][r * q] : 0.0;  // x0 = v_ + e
        }
        if (OLD) {
          if (own_plane && r >= 1 && r <= RY && x_out && row_in[r])
            rr_store(&Y[(int64_t)k * g.plane + (int64_t)(y0 + r) * g.nx + gx], c);
        }
      }
      if constexpr (ALIGNED) {  // the x-halo elements of plane k, formed like the rows (a lane that fetched nothing holds 0)
        double hc = rha[u];
        if (FORM) {
          const double di = s_dinv[htxy + tz9];
          hc = have_old ? fma(beta, rhb[u], di * rha[u]) : di * rha[u];
        }
        Hp = hc;
        if (GUESS) HEp = hc + rhb[u];  // x0 = v_ + e on the halo elements
      }
    }
    double rv[RY];
    if (MODE == RR_RUPD) {  // residual values of the owned rows of plane z (fetched PD steps ago)
#pragma unroll
      for (int j = 0; j < RY; ++j) rv[j] = rvn[u][j];
      if (z + PD >= zb && z + PD < ze) {
        const double* __restrict__ br = X2 + (int64_t)(z + PD) * g.plane;
#pragma unroll
        for (int j = 0; j < RY; ++j) rvn[u][j] = BUF ? rr_buf_load(br, pbytes, (unsigned)off[j + 1]) : br[off[j + 1]];
      }
    }
    {
      const int kf = k + PD;  // fetched now into the slot just consumed
      if (kf >= zb - 1 && kf <= ze) {
        const int cz = min(max(kf, -1), g.nz);
        const double* __restrict__ bx = X + (int64_t)cz * g.plane;
        const double* __restrict__ bx2 = (OLD && have_old) ? X2 + (int64_t)cz * g.plane : bx;
        if constexpr (BUF) {
          // (a plane beyond a physical face holds nothing: no records, every load 0)
          const unsigned pb = ((kf >= 0 || !g.z_lo_phys) && (kf < g.nz || !g.z_hi_phys)) ? pbytes : 0u;
#pragma unroll
          for (int r = 0; r < NR; ++r) {
            ra[u][r] = rr_buf_load(bx, pb, (unsigned)off[r]);
            if (OLD) rb2[u][r] = rr_buf_load(bx2, pb, (unsigned)off[r]);
          }
          if constexpr (ALIGNED) {
            rha[u] = rr_buf_load(bx, pb, hoff);
            if (OLD) rhb[u] = rr_buf_load(bx2, pb, hoff);
            if (GUESS) rhb[u] = rr_buf_load(X2 + (int64_t)cz * g.plane, pb, hoff);
          }
          if (GUESS) {
            const double* __restrict__ b1 = X2 + (int64_t)cz * g.plane;  // (e comes as X2: see the note on aliasing)
#pragma unroll
            for (int r = 0; r < NR; ++r) re1[u][r < NE ? r : 0] = rr_buf_load(b1, pb, (unsigned)off[r]);
          }
        } else {
#pragma unroll
          for (int r = 0; r < NR; ++r) {
            ra[u][r] = bx[off[r]];
            if (OLD) rb2[u][r] = bx2[off[r]];
          }
          if (GUESS) {
            const double* __restrict__ b1 = X2 + (int64_t)cz * g.plane;  // (e comes as X2: see the note on aliasing)
#pragma unroll
            for (int r = 0; r < NR; ++r) re1[u][r < NE ? r : 0] = b1[off[r]];
          }
        }
      }
    }
    if (z >= zb) {
    // RHS: the interior coefficient rows (B and A with a guess, Mass and K without: 2 x 15 doubles = 60 SGPRs), the scalars and
    // the stimulus table are read from the kernel-argument segment through a pointer the optimiser cannot see through, HERE, once
    // per plane -- as kernel parameters they are loop invariants, held in SGPRs from the top of the kernel on and, there being 106,
    // spilled to VGPR lanes: 32 - 45 SGPRs, 49 v_readlane + 32 v_writelane in the 604 VALU instructions of <RHS, 2, 1, guess>
    // (round 5; the recipe of beat_pde_vtl.hip).  Scalar loads from the constant cache, issued ahead of the stencil they feed.
    typedef const __attribute__((address_space(4))) char* KArgPtr;
    KArgPtr ka = (KArgPtr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(ka));
    const RArgs& az = MODE == RR_RHS ? *(const RArgs*)(ka + offsetof(RKernArgs, a)) : a;
    // x-neighbours by lane shifts (every lane takes part)
    double L0[NR], R0[NR], Rp[NR], Lm[NR];
#pragma unroll
    for (int r = 0; r < NR - 1; ++r) {
      L0[r] = ALIGNED ? from_left_h(C0[r], H0, r) : from_left(C0[r]);
      Lm[r] = ALIGNED ? from_left_h(Cm[r], Hm, r) : from_left(Cm[r]);
    }
#pragma unroll
    for (int r = 1; r < NR; ++r) {
      R0[r] = ALIGNED ? from_right_h(C0[r], H0, 8 + r) : from_right(C0[r]);
      Rp[r] = ALIGNED ? from_right_h(Cp[r], Hp, 8 + r) : from_right(Cp[r]);
    }
    double eL0[NE], eR0[NE], eRp[NE], eLm[NE];
    if (GUESS) {
#pragma unroll
      for (int r = 0; r < NE - 1; ++r) {
        eL0[r] = ALIGNED ? from_left_h(E0[r], HE0, r) : from_left(E0[r]);
        eLm[r] = ALIGNED ? from_left_h(Em[r], HEm, r) : from_left(Em[r]);
      }
#pragma unroll
      for (int r = 1; r < NE; ++r) {
        eR0[r] = ALIGNED ? from_right_h(E0[r], HE0, 8 + r) : from_right(E0[r]);
        eRp[r] = ALIGNED ? from_right_h(Ep[r], HEp, 8 + r) : from_right(Ep[r]);
      }
    }
    const int tz = axis_type3(z, g.nz, g.z_lo_phys, g.z_hi_phys);
#pragma unroll
    for (int j = 0; j < RY; ++j) {
      const int r = j + 1;
      double v[15];
      v[0] = C0[r];
      v[1] = R0[r];
      v[2] = L0[r];
      v[3] = C0[r + 1];
      v[4] = C0[r - 1];
      v[5] = Cp[r];
      v[6] = Cm[r];
      v[7] = R0[r + 1];
      v[8] = L0[r - 1];
      v[9] = Cp[r + 1];
      v[10] = Cm[r - 1];
      v[11] = Rp[r];
      v[12] = Lm[r];
      v[13] = Rp[r + 1];
      v[14] = Lm[r - 1];
      if (x_out && row_in[r]) {
        const int type = txy[r] + 9 * tz;
        const int64_t gi = (int64_t)z * g.plane + (int64_t)(y0 + r) * g.nx + gx;
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < 15; ++k) s = fma(az.ci.c[k], v[k], s);
        if (MODE == RR_RHS) {
          // without a guess: tab / tab2 = Mass / K, b = C_m Mass v - (1 - theta) dt K v + dt stim, r = dt (stim - K v);
          // with one: tab = B, the second window holds x0 = v + e, b = B v + dt stim, r = b - A x0 (the textbook form:
          // its rounding error is ~1e-16 |b|, far below any stopping threshold rtol |b|) -- two stencils instead of three
          double s2 = 0.0;
          if (!GUESS) {
#pragma unroll
            for (int k = 0; k < 15; ++k) s2 = fma(az.ci2.c[k], v[k], s2);
          }
          double di = s_dinv[13];
          if (type != 13) {
            s = 0.0;
            s2 = 0.0;
#pragma unroll
            for (int k = 0; k < 15; ++k) {
              s = fma(s_tab[type * TABW + k], v[k], s);
              if (!GUESS) s2 = fma(s_tab2[type * TABW + k], v[k], s2);
            }
            di = s_dinv[type];
          }
          double stim = 0.0;
          for (int k = 0; k < az.nstim; ++k) stim = fma(az.amp[k], az.w[k][gi], stim);
          const double b = GUESS ? fma(az.dt, stim, s) : az.cm * s - az.omt_dt * s2 + az.dt * stim;
          double rr = GUESS ? b : az.dt * (stim - s2);
          if (GUESS) {  // - A x0
            constexpr int q = GUESS ? 1 : 0;  // (keeps the indices in range in the instantiations without a window)
            const int re_ = r * q, rp = (r + 1) * q, rm = (r - 1) * q;
            double ev[15];
            ev[0] = E0[re_];
            ev[1] = eR0[re_];
            ev[2] = eL0[re_];
            ev[3] = E0[rp];
            ev[4] = E0[rm];
            ev[5] = Ep[re_];
            ev[6] = Em[re_];
            ev[7] = eR0[rp];
            ev[8] = eL0[rm];
            ev[9] = Ep[rp];
            ev[10] = Em[rm];
            ev[11] = eRp[re_];
            ev[12] = eLm[re_];
            ev[13] = eRp[rp];
            ev[14] = eLm[rm];
            double se = 0.0;
            if (type != 13) {
#pragma unroll
              for (int k = 0; k < 15; ++k) se = fma(s_taba[type * TABW + k], ev[k], se);
            } else {
#pragma unroll
              for (int k = 0; k < 15; ++k) se = fma(az.cia.c[k], ev[k], se);
            }
            rr -= se;
          }
          const double zz = di * rr;
          rr_store(&Y[gi], rr);
          if (Y2 != nullptr) Y2[gi] = v[0];
          acc0 = fma(b, b, acc0);
          acc1 = fma(rr, zz, acc1);
          acc2 = fma(rr, rr, acc2);
        } else {
          double di = s_dinv[13];
          if (type != 13) {
            s = 0.0;
#pragma unroll
            for (int k = 0; k < 15; ++k) s = fma(s_tab[type * TABW + k], v[k], s);
            di = s_dinv[type];
          }
          if (MODE == RR_PDOT) {
            acc0 = fma(v[0], s, acc0);  // p . (A p)
          } else if (MODE == RR_UDOT) {  // u = D^-1 r:  u . (A u),  r . u,  r . r
            const double ri = Rw0[RAW ? j : 0];
            acc0 = fma(v[0], s, acc0);
            acc1 = fma(ri, v[0], acc1);
            acc2 = fma(ri, ri, acc2);
          } else if (MODE == RR_PRUPD) {  // r_new = r - alpha (A p_new)
            rr_store(&Y2[gi], fma(-alpha, s, Rw0[RAW ? j : 0]));
          } else {                      // r -= alpha (A p)
            const double ri = fma(-alpha, s, rv[j]);
            rr_store(&Y[gi], ri);
            acc0 = fma(ri * di, ri, acc0);
            acc1 = fma(ri, ri, acc1);
          }
        }
      }
    }
    }  // z >= zb
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      Cm[r] = C0[r];
      C0[r] = Cp[r];
    }
    if constexpr (ALIGNED) {
      Hm = H0;
      H0 = Hp;
      if (GUESS) {
        HEm = HE0;
        HE0 = HEp;
      }
    }
    if (GUESS) {
#pragma unroll
      for (int r = 0; r < NE; ++r) {
        Em[r] = E0[r];
        E0[r] = Ep[r];
      }
    }
    if (RAW) {
#pragma unroll
      for (int j = 0; j < NW; ++j) Rw0[j] = Rwp[j];
    }
    }  // z < ze
  }
  }

  if (MODE == RR_PDOT) {
    const double s0 = beat_block_sum(acc0, red);
    if (threadIdx.x == 0) a.partials[g.part_off + blockIdx.x] = s0;
  } else if (MODE == RR_RUPD) {
    const double s0 = beat_block_sum(acc0, red);
    const double s1 = beat_block_sum(acc1, red);
    if (threadIdx.x == 0) {
      a.partials[g.part_off + blockIdx.x] = s0;
      a.partials[BEAT_MAX_PARTIALS + g.part_off + blockIdx.x] = s1;
    }
  } else if (MODE == RR_PRUPD) {
  } else {
    const double s0 = beat_block_sum(acc0, red);
    const double s1 = beat_block_sum(acc1, red);
    const double s2 = beat_block_sum(acc2, red);
    if (threadIdx.x == 0) {
      a.partials[g.part_off + blockIdx.x] = s0;
      a.partials[BEAT_MAX_PARTIALS + g.part_off + blockIdx.x] = s1;
      a.partials[2 * BEAT_MAX_PARTIALS + g.part_off + blockIdx.x] = s2;
    }
  }
}

// scalar roll between K_B and the next K_A (beta, iteration count, convergence latch): pcg_next_kernel of beat_pde.hip
__global__ void rr_next_kernel(double* st) {
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

// The scalar step of the single-reduction iteration (Chronopoulos & Gear 1989), after the ONE all-reduce of
//   st[PQ] = u . A u,  st[RZN] = r . u,  st[RRN] = r . r      (u = D^-1 r, r = r_i):
// the stopping test on r_i, then  beta_i = (r_i.u_i) / (r_{i-1}.u_{i-1}),  alpha_i = (r.u) / (u.Au - beta_i (r.u) / alpha_{i-1})
// -- the value p_i . A p_i has in exact arithmetic, without forming p_i first.  Counts the update that follows.
__global__ void rr_merged_next_kernel(double* st, double* alphas, int slot) {
  if (st[STOP] != 0.0) return;
  const double g = st[RZN], d = st[PQ], rr = st[RRN];
  st[RR] = rr;
  const double tr = st[RTOL] * st[RTOL] * st[BB];
  if (rr <= st[TOL2]) {
    st[STOP] = 1.0;
    st[REASON] = rr <= tr ? 2.0 : 3.0;
    return;
  }
  if (st[ITERS] >= st[MAXIT]) {
    st[STOP] = 1.0;
    st[REASON] = -3.0;
    return;
  }
  const bool first = st[ITERS] == 0.0;
  const double beta = first ? 0.0 : g / st[RZ];
  const double alpha = first ? g / d : g / (d - beta * g / st[ALPHA]);
  st[BETA] = beta;
  st[ALPHA] = alpha;
  st[RZ] = g;
  alphas[slot] = alpha;
  st[NUPD] += 1.0;
  st[ITERS] += 1.0;
}

// Rows per wave and blocks per launch: on big slabs (>= 64 M nodes) 4 rows and ~4096 blocks are best (the 6-of-4 row
// halo amortises, chunks stay long); on small ones -- 256^3, or the 512 x 512 x 64 slab one of 8 ranks owns -- the
// launch is over in ~100 us, chunks of a few planes would spend their time in the three-plane prologue, and 2 rows /
// ~2048 blocks (twice the waves per plane, chunks of 16 planes) measured best (tools/rr_small_sweep.sh: 1.69 vs 1.89 ms
// per solve on the slab, 1.61 vs 1.77 at 256^3; the classic kernels: 1.78 / 1.66).  BEAT_RR_RY / BEAT_RR_BLOCKS override.
int rr_rows(int64_t nodes) {
  static const int forced = [] {
    const char* e = std::getenv("BEAT_RR_RY");
    const int v = e ? std::atoi(e) : 0;
    return (v == 2 || v == 4) ? v : 0;
  }();
  if (forced) return forced;
  return nodes >= ((int64_t)64 << 20) ? 4 : 2;
}

int rr_target_blocks(int64_t nodes) {
  if (const char* e = std::getenv("BEAT_RR_BLOCKS")) return std::max(1, std::atoi(e));
  return nodes >= ((int64_t)64 << 20) ? 4096 : 2048;
}

int rr_prefetch() {  // planes fetched ahead of their use (BEAT_RR_PD = 1, 2 or 3)
  static const int pd = [] {
    const char* e = std::getenv("BEAT_RR_PD");
    const int v = e ? std::atoi(e) : 1;
    return (v >= 1 && v <= 3) ? v : 1;
  }();
  return pd;
}

// Decomposition of the planes [z_lo, z_hi) of the slab into waves; block partials are written from slot part_off on.
// BEAT_RR_BY_ROWS: bit m set = launches of MODE m put the four waves of a block on four adjacent row blocks (RGeom::nrg)
int rr_by_rows_mask() {
  static const int mask = [] {
    const char* e = std::getenv("BEAT_RR_BY_ROWS");
    return e ? std::atoi(e) : 0;
  }();
  return mask;
}

RGeom make_geom(const beat_pde* pde, int z_lo, int z_hi, int part_off, int rows = 0, int mode = -1) {
  const Geom& f = pde->g;
  RGeom g{};
  const int64_t nodes = (int64_t)f.nx * f.ny * f.nz;
  const int RY = g.ry = rows ? rows : rr_rows(nodes);
  g.nx = f.nx;
  g.ny = f.ny;
  g.nz = f.nz;
  g.plane = f.plane;
  g.z_lo_phys = f.z_lo_phys;
  g.z_hi_phys = f.z_hi_phys;
  g.z_lo = z_lo;
  g.z_hi = z_hi;
  g.part_off = part_off;
  g.ghost_lo_tz = pde->ghost_lo_tz;
  g.ghost_hi_tz = pde->ghost_hi_tz;
  const int segw = (mode >= 0 && rr_aligned_mode(mode)) ? 64 : SEG;  // (x-nodes per wave: what the kernel instance of this mode assumes)
  g.nsegx = (f.nx + segw - 1) / segw;
  g.nrb = (f.ny + RY - 1) / RY;
  const int nzr = std::max(0, z_hi - z_lo);
  g.nrg = (mode >= 0 && ((rr_by_rows_mask() >> mode) & 1)) ? (g.nrb + 3) / 4 : 0;
  const int64_t per_layer = g.nrg > 0 ? (int64_t)g.nsegx * g.nrg : ((int64_t)g.nsegx * g.nrb + 3) / 4;  // blocks per z-chunk
  const int target = rr_target_blocks(nodes);  // 512^3: 1024 -> 12.3, 2048 -> 11.3, 4096 -> 11.0 ms per solve (early version)
  int nchunks = (int)std::max<int64_t>(1, (target + per_layer - 1) / per_layer);
  nchunks = std::max(1, std::min(nchunks, nzr));
  g.zc = std::max(1, (nzr + nchunks - 1) / nchunks);
  g.nchunks = (nzr + g.zc - 1) / g.zc;
  while ((int64_t)g.nsegx * g.nrb * g.nchunks > (int64_t)2 * BEAT_MAX_PARTIALS && g.nchunks > 1) {  // fewer, longer chunks
    g.zc *= 2;
    g.nchunks = (nzr + g.zc - 1) / g.zc;
  }
  g.total_waves = g.nsegx * g.nrb * g.nchunks;
  g.total_blocks = (g.total_waves + 3) / 4;
  if (g.nrg > 0) {
    g.total_blocks = g.nsegx * g.nrg * g.nchunks;
    g.total_waves = g.total_blocks * 4;
  }
  return g;
}

RGeom make_geom(const beat_pde* pde) { return make_geom(pde, 0, pde->g.nz, 0); }

inline int grid_blocks(const RGeom& g) { return ((g.total_blocks + 7) / 8) * 8; }

Coef interior_row(const double* tab) {
  Coef c;
  for (int k = 0; k < 15; ++k) c.c[k] = tab[13 * 15 + k];
  return c;
}

template <int MODE, bool GUESS = false>
void launch_rr(const beat_pde* pde, const RGeom& g, const RArgs& a) {
  if (g.total_blocks <= 0) return;
  const dim3 grid((unsigned)grid_blocks(g)), block(BEAT_BLOCK);  // xcd_block() deals whole runs to the 8 XCDs
  hipStream_t s = pde->ctx->stream;
#define BEAT_RR_LAUNCH(RYV, PDV) \
  BEAT_KERNEL((rr_kernel<MODE, RYV, PDV, GUESS>), grid, block, 0, s, g, a, a.x, GUESS ? a.e : a.x2, a.y, a.y2)
  const int pd = rr_prefetch();
  if (g.ry == 2) {
    if (pd == 2) BEAT_RR_LAUNCH(2, 2); else if (pd == 3) BEAT_RR_LAUNCH(2, 3); else BEAT_RR_LAUNCH(2, 1);
  } else {
    if (pd == 2) BEAT_RR_LAUNCH(4, 2); else if (pd == 3) BEAT_RR_LAUNCH(4, 3); else BEAT_RR_LAUNCH(4, 1);
  }
#undef BEAT_RR_LAUNCH
}
}  // namespace

bool beat_rr_available(const beat_pde* pde) {
  if (pde->var || pde->pc_ncoef != 1) return false;
  if (const char* e = std::getenv("BEAT_RR")) {
    if (e[0] == '0') return false;
  }
  const RGeom g = make_geom(pde);
  const RGeom gb = make_geom(pde, 0, 1, 0);  // a one-plane boundary launch of the decomposed solve
  return (int64_t)grid_blocks(g) + 2 * grid_blocks(gb) <= BEAT_MAX_PARTIALS;
}

// Right-hand side in residual form (see beat_pde_rhs) without the p output.
int beat_rr_rhs(beat_pde* pde, const double* dev_v_prev, const double* const* host_dev_stim_w, const double* host_stim_amp,
                int n_stim, double* dev_x, double* dev_r, double* dev_st, int part) {
  const GuessTerms& gt = pde->guess;
  const bool guess = gt.d != nullptr && gt.use_e;
  // with a guess the kernel holds two register windows (v_ and e): 2 rows per wave keep it at the other kernels' occupancy
  static const int guess_rows = [] {  // BEAT_RR_GUESS_RY = 2 | 4 (experiments)
    const char* e = std::getenv("BEAT_RR_GUESS_RY");
    const int v = e ? std::atoi(e) : 2;
    return v == 4 ? 4 : 2;
  }();
  const int rows = guess ? guess_rows : 0;
  RArgs a{};
  a.x = dev_v_prev;
  a.y = dev_r;
  a.y2 = (dev_x == dev_v_prev) ? nullptr : dev_x;
  a.tab = pde->d_tab(2);
  a.tab2 = pde->d_tab(3);
  a.dinv = pde->d_dinv();
  a.ci = interior_row(pde->h_mass);
  a.ci2 = interior_row(pde->h_stiff);
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
  a.st = dev_st;
  if (guess) {  // r = b - A (v_ + e): the second register window holds x0; tab = B
    a.tab = pde->d_tab(1);
    a.ci = interior_row(pde->h_B);
    a.e = gt.e;
    a.taba = pde->d_tab(0);
    a.cia = interior_row(pde->h_A);
  }
  auto launch = [&](const RGeom& g) {
    if (guess)
      launch_rr<RR_RHS, true>(pde, g, a);
    else
      launch_rr<RR_RHS>(pde, g, a);
    return g.total_blocks > 0 ? grid_blocks(g) : 0;
  };
  const Geom& f = pde->g;
  if (part < 0) {
    const int nb = launch(make_geom(pde, 0, f.nz, 0, rows, RR_RHS));
    BEAT_LAUNCH_CHECK();
    if (pde->fuse_begin.on) {  // a single-slab solve: its start in the same launch (beat_solve_begin)
      pde->fuse_begin.done = true;
      return beat_pde_launch_reduce(pde, nb, 3, dev_st, nullptr, nullptr, 2, dev_st, pde->fuse_begin.rtol, pde->fuse_begin.atol,
                                    pde->fuse_begin.max_it);
    }
    return beat_pde_launch_reduce(pde, nb, 3, dev_st, nullptr);
  }
  // in two parts on a decomposed grid (as beat_rr_pdot_part): the planes whose stencil needs no ghost plane of v_ / e
  // while those travel, then the one or two slab-boundary planes and the reduction over all block partials
  const int lo = f.z_lo_phys ? 0 : 1, hi = f.nz - (f.z_hi_phys ? 0 : 1);
  const RGeom gi = make_geom(pde, lo, std::max(lo, hi), 0, rows, RR_RHS);
  if (part == 0) {
    launch(gi);
    BEAT_LAUNCH_CHECK();
    return BEAT_OK;
  }
  int off = gi.total_blocks > 0 ? grid_blocks(gi) : 0;
  if (!f.z_lo_phys) off += launch(make_geom(pde, 0, 1, off, rows, RR_RHS));
  if (!f.z_hi_phys && (f.nz > 1 || f.z_lo_phys)) off += launch(make_geom(pde, f.nz - 1, f.nz, off, rows, RR_RHS));
  BEAT_LAUNCH_CHECK();
  BEAT_REQUIRE(off <= BEAT_MAX_PARTIALS, "too many block partials");
  return beat_pde_launch_reduce(pde, off, 3, dev_st, nullptr);
}

// p_new = D^-1 r + st[BETA] p_old (p_old unread while beta = 0), LOCAL p_new . A p_new -> dev_st[PQ].
// In two parts on a decomposed grid (as beat_pde_spmv_dot_part): part 0 = the planes whose stencil needs no ghost
// plane (enqueue it while the ghost planes of r travel), part 1 = the one or two slab-boundary planes -- which also
// keep p_new on the ghost planes next to them -- and the reduction of all block partials.
int beat_rr_pdot_part(beat_pde* pde, double* dev_st, const double* dev_r, const double* dev_p_old, double* dev_p_new,
                      int part) {
  const Geom& f = pde->g;
  RArgs a{};
  a.x = dev_r;
  a.x2 = dev_p_old;
  a.y = dev_p_new;
  a.tab = pde->d_tab(0);
  a.dinv = pde->d_dinv();
  a.ci = interior_row(pde->h_A);
  a.dinv_i = pde->h_dinv[13];
  a.partials = pde->ctx->d_partials;
  a.st = dev_st;
  const int lo = f.z_lo_phys ? 0 : 1, hi = f.nz - (f.z_hi_phys ? 0 : 1);
  const RGeom gi = make_geom(pde, lo, std::max(lo, hi), 0, 0, RR_PDOT);
  if (part == 0) {
    launch_rr<RR_PDOT>(pde, gi, a);
    BEAT_LAUNCH_CHECK();
    return BEAT_OK;
  }
  int off = gi.total_blocks > 0 ? grid_blocks(gi) : 0;
  if (!f.z_lo_phys) {
    const RGeom gb = make_geom(pde, 0, 1, off, 0, RR_PDOT);
    launch_rr<RR_PDOT>(pde, gb, a);
    off += grid_blocks(gb);
  }
  if (!f.z_hi_phys && (f.nz > 1 || f.z_lo_phys)) {
    const RGeom gb = make_geom(pde, f.nz - 1, f.nz, off, 0, RR_PDOT);
    launch_rr<RR_PDOT>(pde, gb, a);
    off += grid_blocks(gb);
  }
  BEAT_LAUNCH_CHECK();
  BEAT_REQUIRE(off <= BEAT_MAX_PARTIALS, "too many block partials");
  return beat_pde_launch_reduce(pde, off, 1, dev_st + PQ, dev_st);
}

int beat_rr_pdot(beat_pde* pde, double* dev_st, const double* dev_r, const double* dev_p_old, double* dev_p_new) {
  if (int rc = beat_rr_pdot_part(pde, dev_st, dev_r, dev_p_old, dev_p_new, 0)) return rc;
  return beat_rr_pdot_part(pde, dev_st, dev_r, dev_p_old, dev_p_new, 1);
}

// alpha = st[RZ]/st[PQ] (kept for `slot`); r_new = r - alpha A p (out of place; ghost planes of p current); LOCAL
// r.D^-1 r, r.r -> dev_st[RZN..RRN]; counts the update; with `roll` the scalar roll (beta, latch) follows -- a
// decomposed solve all-reduces dev_st[RZN..RRN] first and calls beat_rr_next itself.
int beat_rr_rupd(beat_pde* pde, double* dev_st, const double* dev_r, double* dev_r_new, const double* dev_p, int slot,
                 bool roll) {
  const RGeom g = make_geom(pde, 0, pde->g.nz, 0, 0, RR_RUPD);
  RArgs a{};
  a.x = dev_p;
  a.x2 = dev_r;
  a.y = dev_r_new;
  a.tab = pde->d_tab(0);
  a.dinv = pde->d_dinv();
  a.ci = interior_row(pde->h_A);
  a.dinv_i = pde->h_dinv[13];
  a.partials = pde->ctx->d_partials;
  a.st = dev_st;
  a.alphas = pde->d_alphas;
  a.slot = slot;
  launch_rr<RR_RUPD>(pde, g, a);
  BEAT_LAUNCH_CHECK();
  // (with `roll` the scalar step -- beta, iteration count, latch: rr_next_kernel's -- runs in the reduction's launch)
  return beat_pde_launch_reduce(pde, grid_blocks(g), 2, dev_st + RZN, dev_st, dev_st + NUPD, roll ? 1 : 0, dev_st);
}

// ---- single-reduction iteration (decomposed solve, BEAT_DIST_MERGED=1) --------------------------------------------
// Iteration i of the classic loop needs two all-reduces that depend on each other (p.Ap, then r.z / r.r of the updated
// residual).  In Chronopoulos & Gear's form both come from ONE pass over r_i:
//   K_U (udot):   u = D^-1 r_i formed while loading;  partials  u . A u,  r . u,  r . r     -- reads r, writes nothing
//   [one all-reduce of three values; scalar step: stopping test, beta_i, alpha_i]
//   K_P (prupd):  p_i = u + beta_i p_{i-1} formed while loading, stored;  r_{i+1} = r_i - alpha_i A p_i   -- no dot product
// With the operator re-applied from registers neither w = A u nor s = A p is ever stored (the textbook form keeps both and
// updates s by recurrence): 8 + 32 B/node per iteration against 24 + 24, two stencil passes as before, and k + 1 reductions
// per solve of k iterations (the pass that finds r_k converged) against 2 k.
// Part 0 / 1 as beat_rr_pdot_part; part 1 reduces the block partials into dev_st[PQ..RRN] (local sums).
int beat_rr_udot_part(beat_pde* pde, double* dev_st, const double* dev_r, int part) {
  static_assert(RZN == PQ + 1 && RRN == PQ + 2, "the three sums travel as one all-reduce of dev_st[PQ..RRN]");
  const Geom& f = pde->g;
  RArgs a{};
  a.x = dev_r;
  a.x2 = dev_r;
  a.tab = pde->d_tab(0);
  a.dinv = pde->d_dinv();
  a.ci = interior_row(pde->h_A);
  a.dinv_i = pde->h_dinv[13];
  a.partials = pde->ctx->d_partials;
  a.st = dev_st;
  const int lo = f.z_lo_phys ? 0 : 1, hi = f.nz - (f.z_hi_phys ? 0 : 1);
  const RGeom gi = make_geom(pde, lo, std::max(lo, hi), 0, 0, RR_UDOT);
  if (part == 0) {
    launch_rr<RR_UDOT>(pde, gi, a);
    BEAT_LAUNCH_CHECK();
    return BEAT_OK;
  }
  int off = gi.total_blocks > 0 ? grid_blocks(gi) : 0;
  if (!f.z_lo_phys) {
    const RGeom gb = make_geom(pde, 0, 1, off, 0, RR_UDOT);
    launch_rr<RR_UDOT>(pde, gb, a);
    off += grid_blocks(gb);
  }
  if (!f.z_hi_phys && (f.nz > 1 || f.z_lo_phys)) {
    const RGeom gb = make_geom(pde, f.nz - 1, f.nz, off, 0, RR_UDOT);
    launch_rr<RR_UDOT>(pde, gb, a);
    off += grid_blocks(gb);
  }
  BEAT_LAUNCH_CHECK();
  BEAT_REQUIRE(off <= BEAT_MAX_PARTIALS, "too many block partials");
  return beat_pde_launch_reduce(pde, off, 3, dev_st + PQ, dev_st);
}

int beat_rr_merged_next(beat_pde* pde, double* dev_st, int slot) {
  BEAT_KERNEL(rr_merged_next_kernel, dim3(1), dim3(1), 0, pde->ctx->stream, dev_st, pde->d_alphas, slot);
  BEAT_LAUNCH_CHECK();
  return BEAT_OK;
}

// p_new = D^-1 r + st[BETA] p_old on the slab and on the ghost planes next to it (as beat_rr_pdot_part: no exchange of p),
// r_new = r - st[ALPHA] A p_new (out of place).  The ghost planes of r (and of p_old) must be current.
int beat_rr_prupd(beat_pde* pde, double* dev_st, const double* dev_r, const double* dev_p_old, double* dev_p_new,
                  double* dev_r_new) {
  const RGeom g = make_geom(pde, 0, pde->g.nz, 0, 0, RR_PRUPD);
  RArgs a{};
  a.x = dev_r;
  a.x2 = dev_p_old;
  a.y = dev_p_new;
  a.y2 = dev_r_new;
  a.tab = pde->d_tab(0);
  a.dinv = pde->d_dinv();
  a.ci = interior_row(pde->h_A);
  a.dinv_i = pde->h_dinv[13];
  a.partials = pde->ctx->d_partials;
  a.st = dev_st;
  launch_rr<RR_PRUPD>(pde, g, a);
  BEAT_LAUNCH_CHECK();
  return BEAT_OK;
}

int beat_rr_next(beat_pde* pde, double* dev_st) {
  BEAT_KERNEL(rr_next_kernel, dim3(1), dim3(1), 0, pde->ctx->stream, dev_st);
  BEAT_LAUNCH_CHECK();
  return BEAT_OK;
}
