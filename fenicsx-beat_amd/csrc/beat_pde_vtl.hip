// Per-node-coefficient SpMV of the PCG on workgroup tiles: q = A p with every stored forward coefficient and every row
// of p loaded ONCE per tile, the rest shared on chip.
//
// The segment-list kernel (var_spmv_kernel, beat_pde_var.hip) issues 23 wave-level loads per 64 nodes -- 8 forward
// coefficients, 7 backward ones read as the neighbours' forward coefficients a row or a plane away, 7 rows of p, the
// segment's edge values -- and moves 121 B per tissue node from beyond the L2 on the 401^3 shell (PMC, round 3) against
// the 80 B a node owns (8 coefficients, p, q).  The z-marching kernel of round 3 (beat_pde_vrr.hip) loads the forward
// half only but keeps RY rows x 3 planes of p and two sets of coefficients per WAVE (180-240 VGPRs), loads its whole
// footprint whether tissue or not, and reads the tissue bits with a vector load inside the march (a vmcnt(0) per plane
// that drains its prefetch).  Here:
//   * a WORKGROUP owns a tile of 62 x-nodes (lanes 1..62 of a wave; lanes 0 and 63 carry the x-halo) x RY rows, one row
//     per wave, and marches along z over a run of planes that hold tissue;
//   * per plane a wave loads its row's 8 forward coefficients and ONE row of p (the plane after next; both one step ahead
//     of their use); its own row's three planes of p and the four slots that point to the plane above stay in registers;
//   * what a row needs from the rows next to it -- p of the rows above and below, the four forward slots of the row
//     below that point up (+y, +x+y, +y+z, +x+y+z: its backward -y, -x-y, -y-z, -x-y-z) -- goes through LDS: every wave
//     publishes its row once per plane, one barrier per plane, two buffers; the two halo rows of the tile (p above and
//     below, the four slots of the row below) are six more loads per plane, dealt to the waves;
//   * x-neighbours by DPP wave shifts; lanes on nodes outside the tissue issue no loads (per-row 64-bit masks, a table
//     laid out along z and read with scalar loads two steps ahead), so lines without tissue are never fetched.
// Per plane and tile: 9 RY + 6 wave loads for 62 RY nodes (RY = 8: 9.75 per 62 nodes instead of 23 per 64).  The values
// of q are those of var_spmv_kernel bit for bit (same coefficients, the same 15 fused multiply-adds in slot order);
// p.q is summed per TILE and the tiles in list order (the same bits whoever computed what).
// Inside the solver the same pass also FORMS the search direction (PDOT = true, beat_vtl_pdot): it reads r and the previous
// direction instead of p, builds p = D^-1 r + beta p_old wherever the product needs it (own row, rows above and below, halo
// lanes; D^-1 = 1 / the row's own centre coefficient), stores it and computes q = A p -- var_pupdate_oop_kernel's pass over
// r, p_old and 1/diag is gone from the iteration.  Three tile lists per operator: the whole slab, and for decomposed
// grids the planes that need no ghost plane of p and the one or two that do (beat_vtl_spmv_dot_part).
// Measured (profiles/r04_shell400.md, r04_var_resources.md): SpMV 529-543 -> 403-427 us on an 18.6 M-node shell, 121 -> 105
// B/node from beyond the L2; the fused pass 475 us against 375 + 125; 119-125 VGPRs, no scratch, no spilled SGPR.
//
// Replaces, like beat_pde_var.hip, PETSc's MatMult inside KSP.solve (src/beat/base_model.py:236) for operators assembled
// from per-cell conductivity tensors (src/beat/conductivities.py:101-118, demos/biv_endocardial.py:187-282).
#include "beat_pde_internal.h"

#include <algorithm>
#include <cstdlib>
#include <vector>

namespace {
using namespace beat_pde_detail;

constexpr int SEG = 62;  // x-nodes computed per wave (lanes 1..62)
using u64 = unsigned long long;

struct VtlItem {
  int seg, rb, zb, ze;  // tile column (x segment, row block) and the planes [zb, ze) it computes
};

struct VtlArgs {
  int64_t ld;
  int nx, ny, nz;
  int64_t plane;
  int nsegx;  // x segments per row
  int nzp;    // entries per (row, x segment) of the mask table: planes -1 .. nz + padding
  double* partials;
  int part_off;
  const double* st;
  const double* x;         // p
  double* y;               // q
  const double* rows;      // (15, ld) coefficient rows of A
  const unsigned long long* mask;  // lane masks per (row, x segment) along z
  const VtlItem* items;
  const int* xcd_first;    // 9 entries: the part of the tile list each XCD walks
  int* next;               // 8 counters (next tile of each part) + the number of workgroups done
  // PDOT (the solver's iteration): x = the PREVIOUS search direction, the new one is formed while loading --
  // p = D^-1 r + beta p_old, D^-1 = 1 / the row's own centre coefficient -- and stored to pnew
  const double* r;
  double* pnew;
  int first;               // bit 0: iteration 0, p = D^-1 r (beta = 0, p_old not read); bit 1: keep the direction on the lower ghost plane
  // PDOT on a slab with live neighbours (beat_vtl_pdot_part): the centre coefficients of the NEIGHBOURS' boundary planes (one
  // plane each, exchanged once per operator) -- the direction is formed on the ghost planes too, from the exchanged r, the
  // previous direction kept there and these, and stored there, so that p itself is never exchanged.  nullptr: a physical face.
  const double* gc0_lo;
  const double* gc0_hi;
  // The right-hand side of a step in two single-window passes (round 5; the constant-coefficient kernels' formulation,
  // beat_pde_rr.hip: b = B v_ + dt stim, r = b - A x0):
  //   BV  (mode 2): rows = the rows of B, x = v_, y = b (stored), per-tile partials of b.b
  //   RES (mode 4): rows = A, x = the guess increment e (or nullptr), r = v_ -- the window holds x0 = v_ + e, formed while loading
  //                 as PDOT forms the direction --, t = b, y = the residual b - A x0, pnew = D^-1 r (or nullptr: not wanted),
  //                 per-tile partials of r.z and r.r (slots 1 and 2)
  const double* t;
  double dt;
  const double* w[BEAT_MAX_STIM];
  double amp[BEAT_MAX_STIM];
  int nstim;
  int ignore_stop;         // launches outside the iteration (the latch of the previous solve is still set)
};

__device__ __forceinline__ double vtl_from_left(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x138, 0xf, 0xf, true);  // wave_shr:1
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x138, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double vtl_from_right(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x130, 0xf, 0xf, true);  // wave_shl:1
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x130, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}

// One plane of a field (or of a coefficient row) as a raw buffer: a lane that wants nothing passes an offset beyond the
// buffer's end -- its load returns 0 and fetches nothing, its store is dropped -- so neither needs a branch or a change
// of the exec mask, and the compiler can count what is outstanding at every point of the march.
typedef int vtl_v2i __attribute__((ext_vector_type(2)));
constexpr unsigned VTL_OOB = 0x80000000u;
// (-DBEAT_VTL_NT, an experiment of round 5: bit 0 = the coefficient loads non-temporal -- every coefficient is read once per pass
// and should not push the vector rows the neighbouring tiles share out of the L2 --, bit 1 = the stores of p and q; measured in
// profiles/r05_shell400.md; default 0)
#ifndef BEAT_VTL_NT
#define BEAT_VTL_NT 0
#endif
template <int AUX = 0>
__device__ __forceinline__ double vtl_buf_load(const double* base, unsigned bytes, unsigned off) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000);
  const vtl_v2i v = __builtin_amdgcn_raw_buffer_load_b64(r, (int)off, 0, AUX);
  return __hiloint2double(v.y, v.x);
}
__device__ __forceinline__ void vtl_buf_store(double* base, unsigned bytes, unsigned off, double val) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000);
  vtl_v2i v;
  v.x = __double2loint(val);
  v.y = __double2hiint(val);
  __builtin_amdgcn_raw_buffer_store_b64(v, r, (int)off, 0, (BEAT_VTL_NT & 2) ? 2 : 0);
}

// forward slots of the 15-point stencil (beat_stencil_offsets): 0 centre, 1 +x, 3 +y, 5 +z, 7 +x+y, 9 +y+z, 11 +x+z,
// 13 +x+y+z; the backward slot k+1 pairs with the forward slot k.  F[] below holds them in that order.
template <int RY, bool DYN, int MODE>
__global__ __launch_bounds__(RY * 64, 4) void vtl_spmv_kernel(VtlArgs a_) {  // MODE: 0 q = A p | 1 PDOT | 2 BV | 3 PDOT, live neighbours | 4 RES
  constexpr bool PDOT = MODE == 1 || MODE == 3, BV = MODE == 2, RES = MODE == 4;
  constexpr bool FORMED = PDOT || RES;  // the window's values are formed from several loads (PDOT: r, p_old, c0; RES: v_, e)
  constexpr bool GHOST = MODE == 3;  // PDOT on a slab with live neighbours: the direction is formed and kept on the ghost planes too
  constexpr int NPE = 2;  // halo entries that are rows of a vector: p below / above
  // one LDS array: p of plane z+1 for rows y0-1 .. y0+RY of the tile (two buffers), slots 3, 7, 9, 13 of plane z for rows
  // y0-1 .. y0+RY-2 (two buffers), and a row per wave that absorbs the stores of a wave without a (second) halo load
  constexpr int P_PAR = (RY + 2) * 64, C_PAR = RY * 4 * 64;
  constexpr int P_BASE = 0, C_BASE = 2 * P_PAR, DUMMY = C_BASE + 2 * C_PAR;
  constexpr bool TWO = RY < NPE + 4;  // NPE + 4 halo loads per plane: two per wave when a tile has fewer rows than that
  static_assert(!(RES || BV) || !TWO, "the right-hand side passes are built for tiles of 8 rows");
  __shared__ double lds[DUMMY + RY * 64];
  __shared__ double red[3 * RY];
  __shared__ int next_item;
  if (a_.st[STOP] != 0.0 && !a_.ignore_stop) return;
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const u64 lanebit = 1ull << lane;
  // Every XCD (blocks b with b % 8 == x) walks one contiguous, equally heavy part of the tile list, front to back: the
  // tiles its workgroups have in flight are neighbours, the halo rows and lanes they share are lines its L2 holds.  A
  // workgroup that is done takes the part's next tile (one counter per XCD; tiles differ by a factor of ten in tissue,
  // dealt round-robin the average wave was alive 75 % of the launch), and helps the other XCDs out once its own part is
  // exhausted.  p.q is summed per TILE, the reduction adds the tiles in list order: the same bits whoever computed what.
  // DYN = false (the default, see beat_vtl_setup): the tiles of a part dealt round-robin to the XCD's workgroups.
  const int xcd = blockIdx.x & 7;
  int static_it = a_.xcd_first[xcd] + (int)(blockIdx.x >> 3);
  for (;;) {
    // (taken by wave 0 as a whole, uniformly; only the atomic itself is one lane's.  Written as `if (threadIdx.x == 0)` at
    // the head of the loop, the compiler split the loop by lanes and wave 0 executed the barrier below twice per round.)
    if (w == 0) {
      int got = -1;
      if (DYN) {
        for (int k = 0; k < 8 && got < 0; ++k) {
          const int x = (xcd + k) & 7;
          const int first = a_.xcd_first[x], end = a_.xcd_first[x + 1];
          if (first < end) {
            int n = 0;
            if (lane == 0) n = atomicAdd(a_.next + x, 1);
            n = __builtin_amdgcn_readfirstlane(n);
            if (n < end - first) got = first + n;
          }
        }
      } else {
        if (static_it < a_.xcd_first[xcd + 1]) got = static_it;
        static_it += (int)(gridDim.x >> 3);
      }
      if (lane == 0) next_item = got;
    }
    __syncthreads();  // (also: the previous tile's last reads of the LDS buffers are done)
    const int it = __builtin_amdgcn_readfirstlane(next_item);
    if (it < 0) break;
    double acc = 0.0;
    // the kernel arguments through a pointer the optimiser cannot see through, per tile: read where a tile needs them
    // instead of being held in SGPRs from the top of the kernel on (the kernel has none to spare)
    typedef const __attribute__((address_space(4))) char* KArgPtr;
    KArgPtr ka = (KArgPtr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(ka));
    const VtlArgs& a = *(const VtlArgs*)ka;
    const double* __restrict__ X = a.x;
    const double* __restrict__ A = a.rows;
    const u64* __restrict__ MASK = a.mask;
    const VtlItem item = a.items[it];  // wave-uniform
    const int y0 = item.rb * RY, zb = item.zb, ze = item.ze;
    const int gy = y0 + w;
    const int gx = item.seg * SEG - 1 + lane;
    const int cx = min(max(gx, 0), a.nx - 1);
    // rows of the mask table: index y + 1 for y = -1 .. ny (both ends all zero); z + 1 for z = -1 .. nz
    // (32-bit offsets into the table: the set-up refuses grids whose table has 2^31 entries)
    const int ty = min(gy, a.ny), tyu = min(gy + 1, a.ny), tyd = min(gy - 1, a.ny);
    const int mo_own = ((ty + 1) * a.nsegx + item.seg) * a.nzp + 1;
    const unsigned roff = (unsigned)(min(gy, a.ny - 1) * a.nx + cx);
    // Everything inside the march is data, not control flow: a load that is not wanted (a lane outside the tissue, a plane
    // beyond the run, a halo load this wave does not have) gets an empty mask, i.e. out-of-range offsets in every lane.
    // With branches around the loads and the store the compiler cannot count what is outstanding where the paths join
    // and waits for everything (vmcnt(0) right behind the store of q, every plane).
    const unsigned pbytes = (unsigned)a.plane * 8u;
    auto lane_off = [&](u64 mk, unsigned ro) -> unsigned { return (mk & lanebit) ? ro * 8u : VTL_OOB; };
    auto ldp = [&](u64 mk, int z, unsigned ro) -> double { return vtl_buf_load(X + (int64_t)z * a.plane, pbytes, lane_off(mk, ro)); };
    auto ldc = [&](u64 mk, int slot, int z, unsigned ro) -> double {
      return vtl_buf_load(A + (int64_t)slot * a.ld + (int64_t)z * a.plane, pbytes, lane_off(mk, ro));
    };
    // PDOT: a value of p is formed from three loads -- the residual, the previous direction and the row's centre coefficient
    // c0 (D^-1 = 1 / c0, rounded as var_form_A_kernel rounds the stored 1/diag; p = fma(beta, p_old, D^-1 r): the very
    // expression of var_pupdate_oop_kernel, so the same bits) -- wherever the SpMV needs it: the own row, the rows above and
    // below, the halo lanes.  Redundant arithmetic on the halo, no second pass over r, p_old and 1/diag (32 B/node).
    const double beta = PDOT ? a.st[BETA] : 0.0;
    const bool first = PDOT && (a.first & 1) != 0;
    struct Trio {
      double r, q, c0;
    };
    auto form = [&](const Trio& t) -> double {
      if constexpr (RES) {
        return t.r + t.q;  // x0 = v_ + e (a lane that loaded nothing: 0 + 0; without a guess e is not read: + 0)
      } else {
        const double di = 1.0 / t.c0;
        const double zz = di * t.r;
        const double pv = first ? zz : fma(beta, t.q, zz);
        return t.c0 != 0.0 ? pv : 0.0;  // (a lane that loaded nothing: c0 = 0)
      }
    };
    // RES: the second operand of a formed value is the guess increment; nullptr (no guess on record): never read
    const bool no_second = RES ? X == nullptr : first;
    // centre coefficients of plane z: slot 0 of the rows, or the neighbour's plane on a ghost plane
    auto c0_of = [&](int z) -> const double* {
      if constexpr (GHOST) {
        KArgPtr kg = ka;
        asm volatile("" : "+s"(kg));
        const VtlArgs& ag = *(const VtlArgs*)kg;
        return z < 0 ? ag.gc0_lo : (z >= a.nz ? ag.gc0_hi : A + (int64_t)z * a.plane);
      } else {
        return A + (int64_t)z * a.plane;
      }
    };
    auto ld3 = [&](u64 mk, int z, unsigned ro) -> Trio {
      const unsigned off = lane_off(mk, ro);
      KArgPtr kr = ka;
      asm volatile("" : "+s"(kr));
      const double* Rr = ((const VtlArgs*)kr)->r;
      Trio t;
      t.r = vtl_buf_load(Rr + (int64_t)z * a.plane, pbytes, off);
      t.q = vtl_buf_load(X + (int64_t)z * a.plane, pbytes, no_second ? VTL_OOB : off);
      t.c0 = RES ? 0.0 : vtl_buf_load(c0_of(z), pbytes, off);
      return t;
    };
    // (a ghost plane of a physical face: its mask is empty, nothing is loaded, p = 0; the decomposed solve does not come here)
    auto ldpv = [&](u64 mk, int z, unsigned ro) -> double {  // p at (row ro, plane z), straight from memory
      if constexpr (FORMED) {
        return form(ld3(mk, z, ro));
      } else {
        return ldp(mk, z, ro);
      }
    };
    // the halo rows of the tile: p of rows y0-1 and y0+RY (consumed like the waves' own p: loaded for plane z+2 at step
    // z), slots 3, 7, 9, 13 of row y0-1 (like the own coefficients: plane z+1 at step z); wave w takes halo loads w and
    // w + RY of the six (e: 0 = p below, 1 = p above, 2..5 = slot 3 / 7 / 9 / 13 of the row below).  What a halo load
    // needs is re-derived from its number where it is used (a handful of scalar instructions per plane): kept as
    // pointers, the descriptors cost the SGPRs the kernel does not have
    const int ea = w, eb = w + RY;
    auto halo_row = [&](int e) -> int { return (e < NPE && (e & 1)) ? y0 + RY : y0 - 1; };
    const unsigned roff_a = (unsigned)(min(max(halo_row(ea), 0), a.ny - 1) * a.nx + cx);
    const unsigned roff_b = (unsigned)(min(max(halo_row(eb), 0), a.ny - 1) * a.nx + cx);
    const int moff_a = ((min(max(halo_row(ea), -1), a.ny) + 1) * a.nsegx + item.seg) * a.nzp + 1;
    const int moff_b = ((min(max(halo_row(eb), -1), a.ny) + 1) * a.nsegx + item.seg) * a.nzp + 1;
    auto halo_plane = [&](int e, int z) -> int { return z + (e < NPE ? 2 : 1); };  // what step z loads (step z+1 publishes)
    auto halo_mask = [&](int e, int moff, int z) -> u64 { return MASK[moff + halo_plane(e, z)]; };
    auto halo_load = [&](int e, u64 mk, unsigned ro, int z) -> Trio {
      const bool is_p = e < NPE;
      const int zz = halo_plane(e, z);
      const bool want = e < NPE + 4 && zz <= (is_p ? ze : ze - 1);
      const u64 m = want ? mk : 0ull;
      int64_t ld = a.ld;
      asm volatile("" : "+s"(ld));
      const int slot = (0xD973 >> (4 * ((e - NPE) & 3))) & 15;  // 3, 7, 9, 13
      Trio t{0.0, 0.0, 0.0};
      if constexpr (FORMED) {
        // a p entry is three loads (r, p_old, c0; RES: v_, e), a coefficient entry one: the others run with every lane out of range
        KArgPtr kr = ka;
        asm volatile("" : "+s"(kr));
        const double* base = is_p ? ((const VtlArgs*)kr)->r : A + slot * ld;
        const unsigned off = lane_off(m, ro), offp = is_p ? off : VTL_OOB;
        t.r = vtl_buf_load(base + (int64_t)zz * a.plane, pbytes, off);
        t.q = vtl_buf_load(X + (int64_t)zz * a.plane, pbytes, no_second ? VTL_OOB : offp);
        t.c0 = RES ? 0.0 : vtl_buf_load(c0_of(zz), pbytes, offp);
      } else {
        const double* base = is_p ? X : A + slot * ld;
        t.r = vtl_buf_load(base + (int64_t)zz * a.plane, pbytes, lane_off(m, ro));
      }
      return t;
    };
    auto halo_value = [&](int e, const Trio& t) -> double {  // what is published: p of the halo row, or the coefficient
      if constexpr (FORMED) {
        return e < 2 ? form(t) : t.r;
      } else {
        return t.r;
      }
    };
    auto halo_lds = [&](int e, int z) -> int {  // where step z publishes it
      const int par = (e < NPE ? z + 1 : z) & 1;
      const int p_at = P_BASE + par * P_PAR + (e & 1) * (RY + 1) * 64, c_at = C_BASE + par * C_PAR + (e - NPE) * 64;
      const int at = e < NPE ? p_at : c_at;
      return e >= NPE + 4 ? DUMMY + w * 64 : at;
    };

    // ---- prologue: the planes below the run, straight from memory ------------------------------------------------------
    u64 M0 = MASK[mo_own + zb], M1 = MASK[mo_own + zb + 1], M2 = MASK[mo_own + zb + 2];
    double Pm, P0, U0, D0, Dm, K5, K9, K11, K13;
    double C0keep = 0.0;  // PDOT: the centre coefficient of the plane after this one's p was formed from, until it is F[0]
    double acc1 = 0.0, acc2 = 0.0;  // RES: the partials of r.z and r.r
    const u64 out_lanes = ((1ull << SEG) - 1ull) << 1;
    bool direct = zb == 0;
    {
      const int mo_up = ((tyu + 1) * a.nsegx + item.seg) * a.nzp + 1;
      const int mo_dn = ((tyd + 1) * a.nsegx + item.seg) * a.nzp + 1;
      const unsigned roff_up = (unsigned)(min(gy + 1, a.ny - 1) * a.nx + cx);
      const unsigned roff_dn = (unsigned)(max(gy - 1, 0) * a.nx + cx);
      const u64 mo = MASK[mo_own + zb - 1], md = MASK[mo_dn + zb - 1];
      Pm = ldpv(mo, zb - 1, roff);
      if constexpr (RES) {
        P0 = form(ld3(M0, zb, roff));
      } else if constexpr (PDOT) {

        const Trio t0 = ld3(M0, zb, roff);
        P0 = form(t0);
        C0keep = t0.c0;
        // this tile's first plane of the new direction (the later ones are stored as they are formed, one plane ahead) and -- on
        // plane 0 of a slab with a live lower neighbour, bit 1 of `first` -- the direction on the ghost plane below it, kept for
        // the next iteration: the two planes as ONE buffer (a second descriptor costs the SGPRs the kernel does not have)
        if constexpr (GHOST) {
          double* const two = a.pnew + (int64_t)(zb - 1) * a.plane;
          vtl_buf_store(two, 2u * pbytes, (zb == 0 && (a.first & 2) && (mo & lanebit & out_lanes)) ? roff * 8u : VTL_OOB, Pm);
          vtl_buf_store(two, 2u * pbytes, (M0 & lanebit & out_lanes) ? pbytes + roff * 8u : VTL_OOB, P0);
        } else {
          vtl_buf_store(a.pnew + (int64_t)zb * a.plane, pbytes, (M0 & lanebit & out_lanes) ? roff * 8u : VTL_OOB, P0);
        }
      } else {
        P0 = ldp(M0, zb, roff);
      }
      U0 = ldpv(MASK[mo_up + zb], zb, roff_up);
      D0 = ldpv(MASK[mo_dn + zb], zb, roff_dn);
      Dm = ldpv(md, zb - 1, roff_dn);
      // coefficients of the plane below towards this one: slots 5 / 11 of the own row, 9 / 13 of the row below.  Plane 0
      // of a slab has no stored plane below it: its own backward slots 6, 10, 12, 14 are used as they are (what
      // var_spmv_kernel does; zero on a physical face)
      const int zc = direct ? 0 : zb - 1;
      const u64 mk_o = direct ? M0 : mo, mk_d = direct ? M0 : md;
      const unsigned ro_d = direct ? roff : roff_dn;
      K5 = ldc(mk_o, direct ? 6 : 5, zc, roff);
      K11 = ldc(mk_o, direct ? 12 : 11, zc, roff);
      K9 = ldc(mk_d, direct ? 10 : 9, zc, ro_d);
      K13 = ldc(mk_d, direct ? 14 : 13, zc, ro_d);
    }
    // in flight for the first step: this plane's coefficients, p of the next plane, the halo rows' share
    double Fn[8];
    auto load_coefs = [&](u64 mk, int z) {
      const unsigned off = lane_off(mk & ~(1ull << 63), roff);  // lane 63 is the +x halo: nobody needs its coefficients
      // the row stride opaque per plane: otherwise the seven products k ld are loop invariants, hoisted into SGPR pairs
      // the kernel does not have (31 spilled to VGPR lanes)
      int64_t ld = a.ld;
      asm volatile("" : "+s"(ld));
      const double* bc = A + (int64_t)z * a.plane;
      constexpr int CNT = (BEAT_VTL_NT & 1) ? 2 : 0;
      Fn[0] = PDOT ? 0.0 : vtl_buf_load<CNT>(bc, pbytes, off);  // (PDOT: the centre coefficient comes with r and p_old, a plane earlier)
      bc += ld;
      Fn[1] = vtl_buf_load<CNT>(bc, pbytes, off);
      ld += ld;
#pragma unroll
      for (int k = 2; k < 8; ++k) {
        bc += ld;
        Fn[k] = vtl_buf_load<CNT>(bc, pbytes, off);
      }
    };
    load_coefs(M0, zb);
    double Pn = 0.0, Bn = 0.0;
    Trio Tn{0.0, 0.0, 0.0};
    if constexpr (FORMED) {
      Tn = ld3(M1, zb + 1, roff);
    } else {
      Pn = ldp(M1, zb + 1, roff);
    }
    if constexpr (RES) Bn = vtl_buf_load(a.t + (int64_t)zb * a.plane, pbytes, lane_off(M0, roff));  // b of this plane
    Trio Ean = halo_load(ea, halo_mask(ea, moff_a, zb - 1), roff_a, zb - 1);
    Trio Ebn{0.0, 0.0, 0.0};
    if (TWO) Ebn = halo_load(eb, halo_mask(eb, moff_b, zb - 1), roff_b, zb - 1);
    u64 HA = halo_mask(ea, moff_a, zb), HB = TWO ? halo_mask(eb, moff_b, zb) : 0ull;  // masks of the loads step zb issues
    for (int z = zb; z < ze; ++z) {
      // (what a plane uses once -- the output pointers -- is read from the kernel arguments where it is used, every plane: held
      // in SGPRs across the march it was spilled to VGPR lanes)
      KArgPtr kz = ka;
      asm volatile("" : "+s"(kz));
      const VtlArgs& az = *(const VtlArgs*)kz;
      double F[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) F[k] = Fn[k];
      double Pp = Pn;
      if constexpr (RES) Pp = form(Tn);  // x0 of plane z + 1
      if constexpr (PDOT) {
        Pp = form(Tn);  // p of plane z + 1, from what step z - 1 requested
        F[0] = C0keep;
        C0keep = Tn.c0;
        // (the plane above a slab with a live upper neighbour: the direction on the ghost plane, kept for the next iteration)
        const bool keep = z + 1 < ze || (GHOST && z + 1 == a.nz && az.gc0_hi != nullptr);
        vtl_buf_store(az.pnew + (int64_t)(z + 1) * a.plane, pbytes, (keep && (M1 & lanebit & out_lanes)) ? roff * 8u : VTL_OOB, Pp);
      }
      const double Ea = halo_value(ea, Ean), Eb = TWO ? halo_value(eb, Ebn) : 0.0;
      const double Bv = Bn;
      // masks: one step (the halo loads') and two steps (the own row's) ahead of their use
      const u64 M3 = MASK[mo_own + z + 3];
      const u64 HAn = halo_mask(ea, moff_a, z + 1);
      const u64 HBn = TWO ? halo_mask(eb, moff_b, z + 1) : 0ull;
      // next step's operands
      load_coefs(z + 1 < ze ? M1 : 0ull, z + 1);
      if constexpr (FORMED) {
        Tn = ld3(z + 2 <= ze ? M2 : 0ull, z + 2, roff);
      } else {
        Pn = ldp(z + 2 <= ze ? M2 : 0ull, z + 2, roff);
      }
      if constexpr (RES) Bn = vtl_buf_load(az.t + (int64_t)(z + 1) * a.plane, pbytes, lane_off(z + 1 < ze ? M1 : 0ull, roff));
      Ean = halo_load(ea, HA, roff_a, z);
      if (TWO) Ebn = halo_load(eb, HB, roff_b, z);
      // publish this row
      lds[P_BASE + ((z + 1) & 1) * P_PAR + (w + 1) * 64 + lane] = Pp;
      {
        double* __restrict__ cw = lds + (w + 1 < RY ? C_BASE + (z & 1) * C_PAR + (w + 1) * 4 * 64 : DUMMY + w * 64) + lane;
        const int cs = w + 1 < RY ? 64 : 0;
        cw[0] = F[2];
        cw[cs] = F[4];
        cw[2 * cs] = F[5];
        cw[3 * cs] = F[7];
      }
      lds[halo_lds(ea, z) + lane] = Ea;
      if (TWO) lds[halo_lds(eb, z) + lane] = Eb;
      __syncthreads();
      const double* __restrict__ pr = lds + P_BASE + ((z + 1) & 1) * P_PAR + w * 64 + lane;
      const double Dn = pr[0], Un = pr[128];
      const double* __restrict__ cr = lds + C_BASE + (z & 1) * C_PAR + w * 4 * 64 + lane;
      const double H3 = cr[0], H7 = cr[64], H9 = cr[128], H13 = cr[192];
      double c[15], v[15];
      c[0] = F[0];
      c[1] = F[1];
      c[2] = vtl_from_left(F[1]);  // -x: the left neighbour's +x
      c[3] = F[2];
      c[4] = H3;                   // -y: the lower row's +y
      c[5] = F[3];
      c[6] = K5;                   // -z: the lower plane's +z
      c[7] = F[4];
      c[8] = vtl_from_left(H7);    // -x-y
      c[9] = F[5];
      c[10] = K9;                  // -y-z
      c[11] = F[6];
      const double K11l = vtl_from_left(K11), K13l = vtl_from_left(K13);
      c[12] = direct ? K11 : K11l;  // -x-z
      c[13] = F[7];
      c[14] = direct ? K13 : K13l;  // -x-y-z
      v[0] = P0;
      v[1] = vtl_from_right(P0);
      v[2] = vtl_from_left(P0);
      v[3] = U0;
      v[4] = D0;
      v[5] = Pp;
      v[6] = Pm;
      v[7] = vtl_from_right(U0);
      v[8] = vtl_from_left(D0);
      v[9] = Un;
      v[10] = Dm;
      v[11] = vtl_from_right(Pp);
      v[12] = vtl_from_left(Pm);
      v[13] = vtl_from_right(Un);
      v[14] = vtl_from_left(Dm);
      // values are selected, never multiplied by a zero coefficient: a stale ghost plane cannot leak a NaN
      double s = 0.0;
#pragma unroll
      for (int k = 0; k < 15; ++k) s = fma(c[k], (k == 0 || c[k] != 0.0) ? v[k] : 0.0, s);
      const bool out = (M0 & lanebit & out_lanes) != 0ull;
      if constexpr (BV) {
        // b = B v_ + dt stim (rr_kernel's expression for the constant-coefficient grids), stored for the second pass
        double stim = 0.0;
        for (int k = 0; k < az.nstim; ++k)
          stim = fma(az.amp[k], vtl_buf_load(az.w[k] + (int64_t)z * a.plane, pbytes, out ? roff * 8u : VTL_OOB), stim);
        const double b = fma(az.dt, stim, s);
        acc = fma(out ? b : 0.0, b, acc);
        vtl_buf_store(az.y + (int64_t)z * a.plane, pbytes, out ? roff * 8u : VTL_OOB, b);
      } else if constexpr (RES) {
        // s = A x0: r = b - A x0 (the textbook form: its rounding error is ~1e-16 |b|, far below any threshold rtol |b|), z = D^-1 r
        const double rr = Bv - s;
        const double zz = (1.0 / F[0]) * rr;
        if (out) {
          acc1 = fma(rr, zz, acc1);
          acc2 = fma(rr, rr, acc2);
        }
        vtl_buf_store(az.y + (int64_t)z * a.plane, pbytes, out ? roff * 8u : VTL_OOB, rr);
        vtl_buf_store(az.pnew + (int64_t)z * a.plane, pbytes, (out && az.pnew != nullptr) ? roff * 8u : VTL_OOB, zz);
      } else {
        acc = fma(out ? P0 : 0.0, s, acc);  // (an inactive lane's P0 is 0 anyway; s is finite)
        vtl_buf_store(az.y + (int64_t)z * a.plane, pbytes, out ? roff * 8u : VTL_OOB, s);
      }
      // roll: this plane becomes the plane below
      Pm = P0;
      P0 = Pp;
      U0 = Un;
      Dm = D0;
      D0 = Dn;
      K5 = F[3];
      K11 = F[6];
      K9 = H9;
      K13 = H13;
      direct = false;
      M0 = M1;
      M1 = M2;
      M2 = M3;
      HA = HAn;
      HB = HBn;
    }
    // the tile's share of p.q (BV: of b.b in slot 0; RES: of r.z and r.r in slots 1 and 2), summed in wave order
    if constexpr (RES) {
      acc1 = beat_wave_sum(acc1);
      acc2 = beat_wave_sum(acc2);
      if (lane == 0) {
        red[RY + w] = acc1;
        red[2 * RY + w] = acc2;
      }
    } else {
      acc = beat_wave_sum(acc);
      if (lane == 0) red[w] = acc;
    }
    __syncthreads();
    if (w == 0) {
#pragma unroll
      for (int q = (RES ? 1 : 0); q < (RES ? 3 : 1); ++q) {
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < RY; ++k) t += red[q * RY + k];
        if (lane == 0) a.partials[q * BEAT_MAX_PARTIALS + a.part_off + it] = t;
      }
    }
  }
  // the last workgroup to leave rewinds the counters for the next launch
  // (no fence: a __threadfence() here is an L2 write-back and invalidate per workgroup -- a thousand of them while the others
  // still live off the lines they share; the counters are only ever touched by atomics, and a workgroup's last grab, whose
  // result it had to see before it got here, precedes its tick)
  if (DYN && w == 0 && lane == 0) {
    if (atomicAdd(a_.next + 8, 1) == (int)gridDim.x - 1) {
#pragma unroll
      for (int k = 0; k < 9; ++k) a_.next[k] = 0;
    }
  }
}

struct VtlData {
  int ry = 4;
  bool dyn = false;
  bool pdot = true;  // BEAT_VTL_PDOT=0: the three-kernel iteration (SpMV, residual update, direction update)
  bool rhs = true;   // the right-hand side in two tile passes (beat_vtl_rhs); BEAT_VTL_RHS=0: var_rhs_kernel's gathers
  VtlItem* d_items = nullptr;
  int* d_xcd_first = nullptr;  // per list 9 entries: the part of the list each XCD walks; then 9 counters (next tile per XCD, workgroups done)
  u64* d_mask = nullptr;
  // tile lists, back to back in d_items: [0] the whole slab, [1] the planes that need no ghost data, [2] the one or two that do
  // (decomposed grids; empty on a slab with two physical faces)
  int list_base[3] = {0, 0, 0}, list_count[3] = {0, 0, 0};
  int nitems = 0, nsegx = 0, nzp = 0;
  unsigned resident = 0;
};

template <int RY>
unsigned vtl_resident_blocks() {
  auto kernel = vtl_spmv_kernel<RY, false, RY == 8 ? 1 : 0>;
  int dev = 0, cus = 256, per_cu = 1;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, RY * 64, 0) != hipSuccess || per_cu < 1) per_cu = 1;
  return (unsigned)std::min(cus * per_cu, BEAT_MAX_PARTIALS) & ~7u;
}
}  // namespace

void beat_vtl_destroy(beat_pde* pde) {
  VtlData* d = (VtlData*)pde->vtl;
  if (d == nullptr) return;
  if (d->d_items) (void)hipFree(d->d_items);
  if (d->d_xcd_first) (void)hipFree(d->d_xcd_first);
  if (d->d_mask) (void)hipFree(d->d_mask);
  delete d;
  pde->vtl = nullptr;
}

// flags: tissue bits per 64-node segment of the slab (host copy).  Builds the per-row lane masks along z and the list of
// tiles (column x run of planes with tissue), in (z block, row block, x segment) order, cut into eight equally heavy parts.
int beat_vtl_setup(beat_pde* pde, const std::vector<unsigned long long>& flags) {
  // BEAT_VTL=0 keeps the segment-list kernel (var_spmv_kernel), which also serves every launch over part of a slab
  const char* sw = std::getenv("BEAT_VTL");
  if (sw != nullptr && sw[0] == '0') return BEAT_OK;
  const Geom& f = pde->g;
  if (f.nz < 2 || f.nx < 2 || f.ny < 2) return BEAT_OK;  // 1-D / 2-D grids and single planes: the segment list
  if (f.plane >= ((int64_t)1 << 28)) return BEAT_OK;      // 32-bit byte offsets inside a plane
  if ((int64_t)(f.ny + 2) * ((f.nx + SEG - 1) / SEG) * (f.nz + 6) >= ((int64_t)1 << 31)) return BEAT_OK;  // 32-bit mask offsets
  VtlData* d = new VtlData();
  {
    const char* e = std::getenv("BEAT_VTL_RY");
    // eight rows per tile halve the share of the halo rows; four keep thin slabs busy.  Measured on the 401^3 shell, one box,
    // alternating (tools/vtl_ab.sh, rocprofv3 kernel times): 8 rows, runs of 16 planes, dynamic: 403 us; 8 / 32: 414-428;
    // 4 / 32 dealt statically: 423-429, dynamic: 447; 4 / 16: 409-420; 4 / 64: 466 -- the segment-list kernel: 535
    // (16 rows per tile -- 1024-thread workgroups, half the halo share again -- were built and measured too: 11.65 against 11.30 ms
    // per shell step, A B A B on one box: sixteen waves at one barrier per plane; not kept)
    d->ry = e ? (std::atoi(e) == 8 ? 8 : 4) : (f.ny >= 16 ? 8 : 4);
    // tiles dealt round-robin to an XCD's workgroups (default) or taken from a counter per XCD (BEAT_VTL_DYNAMIC=1).  The
    // counter balances better on paper and measured the same on the SpMV alone (tools/bench_voxel.py: 403 against 409 us),
    // but inside the shell's split step (tools/bench_biv.py --n 400, same box, alternating) the round trip of the atomic, with
    // the whole workgroup waiting behind it once per tile, cost 560 against 375 us per launch: 12.85 against 11.24 ms/step
    const char* dy = std::getenv("BEAT_VTL_DYNAMIC");
    d->dyn = dy && dy[0] == '1';
    const char* pd = std::getenv("BEAT_VTL_PDOT");
    d->pdot = !(pd && pd[0] == '0');
    // the right-hand side in two single-window tile passes (beat_vtl_rhs, round 5) is the default on single slabs with tiles
    // of 8 rows; BEAT_VTL_RHS=0 keeps var_rhs_kernel's gathers.  (Round 4's version -- K v_ in a first pass, then ONE pass over A
    // with two vector windows, 166 VGPRs and one workgroup per CU -- measured 1.12 against 1.30 ms and stayed opt-in; it is gone.)
    const char* rh = std::getenv("BEAT_VTL_RHS");
    d->rhs = !(rh && rh[0] == '0');
  }
  int max_run = 16;
  if (const char* e = std::getenv("BEAT_VTL_RUN")) max_run = std::max(1, std::atoi(e));
  int aligned = 0;
  if (const char* e = std::getenv("BEAT_VTL_ALIGN")) aligned = std::atoi(e);
  const int RY = d->ry;
  const int nsegx = (f.nx + SEG - 1) / SEG, nrb = (f.ny + RY - 1) / RY;
  const int nzp = f.nz + 6;  // planes -1 .. nz, and what the march reads ahead of its last plane
  d->nsegx = nsegx;
  d->nzp = nzp;
  auto tissue = [&](int64_t i) -> u64 { return (flags[(size_t)(i >> 6)] >> (i & 63)) & 1ull; };
  // mask[(y + 1) nsegx + seg][z + 1]: bit l = node (62 seg - 1 + l, y, z) is a tissue node of this slab; on a ghost plane
  // that another rank owns: the node is inside the box (its p is live; which of them are tissue is not known here)
  std::vector<u64> mask((size_t)(f.ny + 2) * nsegx * nzp, 0ull);
  for (int y = 0; y < f.ny; ++y)
    for (int s = 0; s < nsegx; ++s) {
      u64* row = mask.data() + ((size_t)(y + 1) * nsegx + s) * nzp + 1;
      u64 box = 0ull;
      for (int l = 0; l < 64; ++l) {
        const int gx = s * SEG - 1 + l;
        if (gx >= 0 && gx < f.nx) box |= 1ull << l;
      }
      if (!f.z_lo_phys) row[-1] = box;
      if (!f.z_hi_phys) row[f.nz] = box;
      for (int z = 0; z < f.nz; ++z) {
        const int64_t base = (int64_t)z * f.plane + (int64_t)y * f.nx + (int64_t)s * SEG - 1;
        u64 m = 0ull;
        for (int l = 0; l < 64; ++l)
          if (((box >> l) & 1ull) && tissue(base + l)) m |= 1ull << l;
        row[z] = m;
      }
    }
  const u64 out_lanes = ((1ull << SEG) - 1ull) << 1;  // lanes 1..62
  std::vector<char> act((size_t)f.nz);
  // the tiles of the planes [z_lo, z_hi), in (z block, row block, x segment) order; false if there are more than partial slots
  auto build = [&](int z_lo, int z_hi, int run, std::vector<VtlItem>& items) -> bool {
    items.clear();
    if (z_hi <= z_lo) return true;
    for (int rb = 0; rb < nrb; ++rb)
      for (int seg = 0; seg < nsegx; ++seg) {
        bool any = false;
        for (int z = z_lo; z < z_hi; ++z) {
          u64 m = 0ull;
          for (int y = rb * RY; y < std::min(f.ny, rb * RY + RY); ++y) m |= mask[((size_t)(y + 1) * nsegx + seg) * nzp + 1 + z];
          act[(size_t)z] = (m & out_lanes) != 0ull;
          any |= act[(size_t)z] != 0;
        }
        if (!any) continue;
        if (aligned) {
          // every tile of a z block covers the block's tissue range as a whole (BEAT_VTL_ALIGN=2: the whole block): measured,
          // within +-2 % of tiles trimmed to their own tissue
          for (int b0 = (z_lo / run) * run; b0 < z_hi; b0 += run) {
            const int c0 = std::max(b0, z_lo), c1 = std::min(z_hi, b0 + run);
            int lo = c1, hi = c0;
            for (int z = c0; z < c1; ++z)
              if (act[(size_t)z]) {
                lo = std::min(lo, z);
                hi = std::max(hi, z + 1);
              }
            if (lo >= hi) continue;
            items.push_back(aligned == 2 ? VtlItem{seg, rb, c0, c1} : VtlItem{seg, rb, lo, hi});
          }
          continue;
        }
        int z = z_lo;
        while (z < z_hi) {
          if (!act[(size_t)z]) {
            ++z;
            continue;
          }
          // a run never crosses a multiple of `run`: tiles of one z block are neighbours in the list
          const int stop = std::min(z_hi, (z / run + 1) * run);
          int e = z + 1, last = z + 1;  // grow the run over gaps of up to two planes
          while (e < stop && (act[(size_t)e] || e - last < 2)) {
            if (act[(size_t)e]) last = e + 1;
            ++e;
          }
          items.push_back(VtlItem{seg, rb, z, last});
          z = last;
        }
      }
    std::stable_sort(items.begin(), items.end(), [&](const VtlItem& p, const VtlItem& q) {
      const int zp = p.zb / run, zq = q.zb / run;
      if (zp != zq) return zp < zq;
      if (p.rb != q.rb) return p.rb < q.rb;
      return p.seg < q.seg;
    });
    return (int64_t)items.size() <= BEAT_MAX_PARTIALS;
  };
  // eight contiguous parts of equal weight (planes + a prologue's worth per tile); indices into the concatenated array
  auto parts = [&](const std::vector<VtlItem>& items, int base, int (&first)[9]) {
    std::vector<int64_t> cum(items.size() + 1, 0);
    for (size_t k = 0; k < items.size(); ++k) cum[k + 1] = cum[k] + (items[k].ze - items[k].zb) + 2;
    first[0] = 0;
    for (int x = 1; x < 8; ++x) {
      const int64_t want = cum.back() * x / 8;
      first[x] = (int)(std::lower_bound(cum.begin(), cum.end(), want) - cum.begin());
      first[x] = std::max(first[x - 1], std::min(first[x], (int)items.size()));
    }
    first[8] = (int)items.size();
    for (int x = 0; x < 9; ++x) first[x] += base;
  };
  std::vector<VtlItem> items, part_items[2];
  // (one partial sum of p.q per tile: longer runs if there would be more tiles than partial slots)
  bool fits = false;
  for (; !(fits = build(0, f.nz, max_run, items)) && max_run < f.nz; max_run *= 2) {
  }
  if (!fits) {  // (a plane of more than 16384 tiles: the segment-list kernel)
    delete d;
    return BEAT_OK;
  }
  if (!(f.z_lo_phys && f.z_hi_phys)) {
    // a decomposed grid: the planes that need no ghost plane of p (their launch overlaps the exchange), then the one or two
    // slab-boundary planes -- the same split as beat_var_spmv_dot_part's
    const int lo = f.z_lo_phys ? 0 : 1, hi = f.nz - (f.z_hi_phys ? 0 : 1);
    std::vector<VtlItem> b_lo, b_hi;
    bool ok = build(lo, std::max(lo, hi), max_run, part_items[0]);
    if (!f.z_lo_phys) ok = build(0, 1, max_run, b_lo) && ok;
    if (!f.z_hi_phys && (f.nz > 1 || f.z_lo_phys)) ok = build(f.nz - 1, f.nz, max_run, b_hi) && ok;
    part_items[1] = b_lo;
    part_items[1].insert(part_items[1].end(), b_hi.begin(), b_hi.end());
    if (!ok || (int64_t)part_items[0].size() + (int64_t)part_items[1].size() > BEAT_MAX_PARTIALS) {
      part_items[0].clear();  // (the split launches keep the segment-list kernel)
      part_items[1].clear();
    }
  }
  int first[3][9];
  d->list_base[0] = 0;
  d->list_count[0] = (int)items.size();
  parts(items, 0, first[0]);
  for (int k = 0; k < 2; ++k) {
    d->list_base[k + 1] = (int)items.size();
    d->list_count[k + 1] = (int)part_items[k].size();
    parts(part_items[k], d->list_base[k + 1], first[k + 1]);
    items.insert(items.end(), part_items[k].begin(), part_items[k].end());
  }
  d->nitems = d->list_count[0];
  pde->vtl = d;
  hipError_t e = hipMalloc(&d->d_items, sizeof(VtlItem) * std::max<size_t>(1, items.size()));
  if (e == hipSuccess && !items.empty()) e = hipMemcpy(d->d_items, items.data(), sizeof(VtlItem) * items.size(), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMalloc(&d->d_xcd_first, sizeof(first) + 9 * sizeof(int));
  if (e == hipSuccess) e = hipMemset(d->d_xcd_first, 0, sizeof(first) + 9 * sizeof(int));
  if (e == hipSuccess) e = hipMemcpy(d->d_xcd_first, first, sizeof(first), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMalloc(&d->d_mask, sizeof(u64) * mask.size());
  if (e == hipSuccess) e = hipMemcpy(d->d_mask, mask.data(), sizeof(u64) * mask.size(), hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    beat_vtl_destroy(pde);
    beat_set_error("beat_vtl_setup: %s", hipGetErrorString(e));
    return BEAT_EHIP;
  }
  d->resident = RY == 8 ? vtl_resident_blocks<8>() : vtl_resident_blocks<4>();
  if (std::getenv("BEAT_VTL_VERBOSE")) {
    long planes = 0;
    for (int k = 0; k < d->nitems; ++k) planes += items[(size_t)k].ze - items[(size_t)k].zb;
    std::fprintf(stderr, "vtl: RY %d, %d x %d columns, %d tiles (+ %d / %d for the split launches), %.1f planes per tile, %u resident blocks\n", RY,
                 nsegx, nrb, d->nitems, d->list_count[1], d->list_count[2], d->nitems ? (double)planes / d->nitems : 0.0, d->resident);
  }
  return BEAT_OK;
}

bool beat_vtl_available(const beat_pde* pde) { return pde->var && pde->vtl != nullptr && ((VtlData*)pde->vtl)->nitems > 0; }

// what the two passes of the right-hand side add to a launch
struct VtlRhs {
  const double* rows;  // coefficient rows of this pass (nullptr: A)
  int mode;            // 2: b = rows v_ + dt stim (rows = B); 4: r = b - A (v_ + e)
  const double* e;     // mode 4: guess increment or nullptr
  const double* t;     // mode 4: b
  double dt;
  const double* w[BEAT_MAX_STIM];
  double amp[BEAT_MAX_STIM];
  int nstim;
  double* red_out;     // b.b, r.z, r.r
};
static int vtl_launch(beat_pde* pde, const double* dev_p, double* dev_q, double* dev_st, const double* dev_r, double* dev_p_new, int first,
                      int list = 0, int part_off = 0, bool reduce = true, int reduce_count = 0, const VtlRhs* rhs = nullptr);

// whole-slab q = A p and the sum p.q
int beat_vtl_spmv_dot(beat_pde* pde, const double* dev_p, double* dev_q, double* dev_st) {
  return vtl_launch(pde, dev_p, dev_q, dev_st, nullptr, nullptr, 0);
}

// the solver's iteration in one pass: p_new = D^-1 r + beta p_old (beta from dev_st; first: p_new = D^-1 r) formed while
// loading and stored, q = A p_new, the sum p_new.q.  Single-slab operators only (both faces physical).
int beat_vtl_pdot(beat_pde* pde, double* dev_st, const double* dev_r, const double* dev_p_old, double* dev_p_new, double* dev_q, int first) {
  BEAT_REQUIRE(pde->g.z_lo_phys && pde->g.z_hi_phys, "beat_vtl_pdot is the single-slab path");
  BEAT_REQUIRE(dev_p_old != dev_p_new, "the direction update is out of place");
  return vtl_launch(pde, dev_p_old, dev_q, dev_st, dev_r, dev_p_new, first);
}

bool beat_vtl_pdot_available(const beat_pde* pde) {
  // (tiles of 8 rows only: with 4 rows a wave carries two halo entries and the fused pass does not fit 128 VGPRs)
  return beat_vtl_available(pde) && ((VtlData*)pde->vtl)->pdot && ((VtlData*)pde->vtl)->ry == 8 && pde->g.z_lo_phys && pde->g.z_hi_phys;
}

static int vtl_launch(beat_pde* pde, const double* dev_p, double* dev_q, double* dev_st, const double* dev_r, double* dev_p_new, int first,
                      int list, int part_off, bool reduce, int reduce_count, const VtlRhs* rhs) {
  const bool bv = rhs != nullptr && rhs->mode == 2, res = rhs != nullptr && rhs->mode == 4;
  const bool pdot = dev_r != nullptr && !res;
  VtlData* d = (VtlData*)pde->vtl;
  const Geom& f = pde->g;
  VtlArgs a{};
  a.ld = pde->v_ld;
  a.nx = f.nx;
  a.ny = f.ny;
  a.nz = f.nz;
  a.plane = f.plane;
  a.nsegx = d->nsegx;
  a.nzp = d->nzp;
  a.partials = pde->ctx->d_partials;
  const int count = d->list_count[list];
  if (count == 0) {  // (a split launch without tiles: only the reduction over what the other part wrote)
    return reduce ? beat_pde_launch_reduce(pde, reduce_count, 1, dev_st + PQ, dev_st) : BEAT_OK;
  }
  a.part_off = part_off - d->list_base[list];  // the kernel indexes its partial by the tile's position in the whole array
  a.st = dev_st;
  // blocks in eights (one per XCD), at most the resident number, at least one tile per block on average
  const unsigned grid = std::max(8u, std::min(d->resident, (unsigned)((count + 7) & ~7)));
  a.x = dev_p;
  a.y = dev_q;
  a.rows = (rhs != nullptr && rhs->rows != nullptr) ? rhs->rows : pde->v_A;
  a.ignore_stop = rhs != nullptr;
  if (bv) {
    a.dt = rhs->dt;
    a.nstim = rhs->nstim;
    for (int k = 0; k < rhs->nstim; ++k) {
      a.w[k] = rhs->w[k];
      a.amp[k] = rhs->amp[k];
    }
  }
  if (res) {
    a.t = rhs->t;
    a.x = rhs->e;  // the second operand of the formed window (nullptr: no guess on record)
  }
  a.mask = d->d_mask;
  a.items = d->d_items;
  a.xcd_first = d->d_xcd_first + 9 * list;
  a.next = d->d_xcd_first + 27;
  a.r = res ? dev_p : dev_r;  // (RES: the first operand of the formed window is v_, passed as dev_p)
  a.pnew = dev_p_new;
  a.first = first ? 1 : 0;
  if (pdot && pde->v_gc0 != nullptr) {  // a slab with live neighbours: the direction is formed and kept on the ghost planes too
    a.gc0_lo = f.z_lo_phys ? nullptr : pde->v_gc0;
    a.gc0_hi = f.z_hi_phys ? nullptr : pde->v_gc0 + f.plane;
    if (a.gc0_lo != nullptr) a.first |= 2;
  }
  auto launch = [&](auto kernel, int ry) { BEAT_KERNEL(kernel, dim3(grid), dim3(ry * 64), 0, pde->ctx->stream, a); };
  if (bv) {
    launch(vtl_spmv_kernel<8, false, 2>, 8);
  } else if (res) {
    launch(vtl_spmv_kernel<8, false, 4>, 8);
  } else if (pdot) {  // (tiles dealt round-robin: the counter of BEAT_VTL_DYNAMIC serves the plain SpMV only)
    (a.gc0_lo != nullptr || a.gc0_hi != nullptr) ? launch(vtl_spmv_kernel<8, false, 3>, 8) : launch(vtl_spmv_kernel<8, false, 1>, 8);
  } else if (d->ry == 8) {
    d->dyn ? launch(vtl_spmv_kernel<8, true, 0>, 8) : launch(vtl_spmv_kernel<8, false, 0>, 8);
  } else {
    d->dyn ? launch(vtl_spmv_kernel<4, true, 0>, 4) : launch(vtl_spmv_kernel<4, false, 0>, 4);
  }
  BEAT_LAUNCH_CHECK();
  if (!reduce) return BEAT_OK;
  if (res) {  // (b.b from the BV pass, r.z and r.r from this one; on a single-slab solve its start in the same launch)
    if (pde->fuse_begin.on) {
      pde->fuse_begin.done = true;
      return beat_pde_launch_reduce(pde, count, 3, rhs->red_out, nullptr, nullptr, 2, rhs->red_out, pde->fuse_begin.rtol, pde->fuse_begin.atol,
                                    pde->fuse_begin.max_it);
    }
    return beat_pde_launch_reduce(pde, count, 3, rhs->red_out, nullptr);
  }
  return beat_pde_launch_reduce(pde, reduce_count ? reduce_count : count, 1, dev_st + PQ, dev_st);  // one partial per tile, in list order
}

// The right-hand side of a step on the tiles (what beat_var_rhs computes with gathers: 352 B/node from beyond the L2 on the 401^3
// shell against 168 algorithmic, 1.30 ms, profiles/r04_shell400.md), in the formulation of the constant-coefficient kernels
// (beat_pde_rr.hip):
//   pass 1 (BV):  b = B v_ + dt sum_k amp_k w_k  -- the plain tile product over the rows of B = C_m Mass - (1 - theta) dt K, which
//                 beat_var_form_A forms beside A (beat_pde::v_B) --, b stored in a work field, per-tile partials of b.b;
//   pass 2 (RES): r = b - A x0 with x0 = v_ + e formed while loading (ONE vector window, as the iteration's fused pass forms its
//                 direction), z = D^-1 r where the three-kernel iteration wants it; per-tile partials of r.z and r.r.
// Two passes at the plain product's cost and register budget (4 waves per SIMD) instead of one pass with two windows at 166 VGPRs.
// r differs from var_rhs_kernel's  dt (stim - K v_) - A e  by rounding (~1e-16 |b|: far below any threshold rtol |b|): the solves
// are the same solves, not the same bits (tests/test_var_gpu.py).  Single slabs, tiles of 8 rows.  dev_t: a work field (the
// solver's q, free until the first iteration).
bool beat_vtl_rhs_available(const beat_pde* pde) {
  if (!(beat_vtl_available(pde) && pde->g.z_lo_phys && pde->g.z_hi_phys)) return false;
  const VtlData* d = (const VtlData*)pde->vtl;
  return d->ry == 8 && d->rhs && pde->v_B != nullptr;
}
bool beat_vtl_rhs_wanted(const beat_pde* pde) {  // (before the rows of B exist: beat_var_form_A allocates them when this says so)
  if (!(beat_vtl_available(pde) && pde->g.z_lo_phys && pde->g.z_hi_phys)) return false;
  const VtlData* d = (const VtlData*)pde->vtl;
  return d->ry == 8 && d->rhs;
}

int beat_vtl_rhs(beat_pde* pde, const double* dev_v_prev, const double* const* host_dev_stim_w, const double* host_stim_amp, int n_stim,
                 double* dev_x, double* dev_r, double* dev_p, double* dev_t, double* dev_red, const double* dev_e) {
  if (dev_x != dev_v_prev)  // nodes outside the tissue keep their value: copy everything first (as beat_var_rhs does)
    BEAT_HIP_CHECK(hipMemcpyAsync(dev_x, dev_v_prev, sizeof(double) * (size_t)pde->n, hipMemcpyDeviceToDevice, pde->ctx->stream));
  VtlRhs one{};
  one.rows = pde->v_B;
  one.mode = 2;
  one.dt = pde->dt;
  for (int k = 0; k < n_stim; ++k) {
    if (host_dev_stim_w[k] == nullptr || host_stim_amp[k] == 0.0) continue;
    one.w[one.nstim] = host_dev_stim_w[k];
    one.amp[one.nstim] = host_stim_amp[k];
    ++one.nstim;
  }
  if (int rc = vtl_launch(pde, dev_v_prev, dev_t, pde->d_st, nullptr, nullptr, 0, 0, 0, false, 0, &one)) return rc;
  VtlRhs two{};
  two.mode = 4;
  two.e = dev_e;
  two.t = dev_t;
  two.red_out = dev_red;
  // (dev_p of vtl_launch carries v_ here; its dev_p_new is where z = D^-1 r goes: only the three-kernel iteration reads it --
  // the fused pass forms its first direction from r itself)
  return vtl_launch(pde, dev_v_prev, dev_r, pde->d_st, nullptr, beat_vtl_pdot_available(pde) ? nullptr : dev_p, 0, 0, 0, true, 0, &two);
}

// The fused pass on a slab with live neighbours, in the two parts of the split launches: part 0 = the planes whose stencil
// needs no ghost plane (enqueue it while the ghost planes of r travel), part 1 = the one or two slab-boundary planes -- which also
// form and keep the direction on the ghost planes next to them, from the exchanged r, the direction kept there by the previous
// iteration and the neighbours' centre coefficients (beat_pde::v_gc0) -- and the sum over both parts' tiles.  p is never exchanged.
bool beat_vtl_pdot_dist_available(const beat_pde* pde) {
  if (!(beat_vtl_available(pde) && beat_vtl_parts_available(pde)) || (pde->g.z_lo_phys && pde->g.z_hi_phys)) return false;
  const VtlData* d = (const VtlData*)pde->vtl;
  static const bool on = [] {  // BEAT_VTL_PDOT_DIST=0: the three-kernel iteration with an exchange of p (A/B runs)
    const char* e = std::getenv("BEAT_VTL_PDOT_DIST");
    return !(e && e[0] == '0');
  }();
  return on && d->pdot && d->ry == 8 && pde->v_gc0 != nullptr;
}

int beat_vtl_pdot_part(beat_pde* pde, double* dev_st, const double* dev_r, const double* dev_p_old, double* dev_p_new, double* dev_q,
                       int first, int part) {
  BEAT_REQUIRE(dev_p_old != dev_p_new, "the direction update is out of place");
  const VtlData* d = (const VtlData*)pde->vtl;
  if (part == 0) return vtl_launch(pde, dev_p_old, dev_q, dev_st, dev_r, dev_p_new, first, 1, 0, false, 0);
  return vtl_launch(pde, dev_p_old, dev_q, dev_st, dev_r, dev_p_new, first, 2, d->list_count[1], true, d->list_count[1] + d->list_count[2]);
}

// the split launches of a decomposed grid: part 0 = the planes that need no ghost plane of p (no reduction), part 1 = the
// slab-boundary planes + the sum over both parts' tiles
bool beat_vtl_parts_available(const beat_pde* pde) {
  if (!(pde->var && pde->vtl != nullptr)) return false;
  const VtlData* d = (const VtlData*)pde->vtl;
  return d->list_count[1] + d->list_count[2] > 0;
}

int beat_vtl_spmv_dot_part(beat_pde* pde, const double* dev_p, double* dev_q, double* dev_st, int part) {
  const VtlData* d = (const VtlData*)pde->vtl;
  if (part == 0) return vtl_launch(pde, dev_p, dev_q, dev_st, nullptr, nullptr, 0, 1, 0, false, 0);
  return vtl_launch(pde, dev_p, dev_q, dev_st, nullptr, nullptr, 0, 2, d->list_count[1], true, d->list_count[1] + d->list_count[2]);
}
