// Per-node-coefficient SpMV of the PCG, marching along z ("register rows" for voxel masks and fibre fields).
//
// q = A p with 15 stored coefficients per node (beat_pde_var.hip) moves, on a voxelised wall, 186 B per node from
// beyond the L2 (PMC, 401^3 box): the 8 "forward" coefficients of the node, the 7 "backward" ones -- read as the
// neighbours' forward coefficients, a plane or a row away, i.e. in another XCD's L2 or in none -- and 7 rows of p, each
// fetched once per row that touches it.  This kernel is the constant-coefficient register-row kernel (beat_pde_rr.hip)
// with the coefficient rows made per-node operands:
//   * a wave owns 62 consecutive x-nodes (lanes 1..62; lanes 0 and 63 carry the x-halo) of RY consecutive rows and
//     marches along z over a run of planes that hold tissue; the three planes of p it needs stay in registers, every
//     row of p is loaded once per wave, x-neighbours come from the adjacent lanes (DPP shifts), the next plane is in
//     flight while the current one is computed;
//   * only the FORWARD coefficients are loaded (8 per node: centre, +x, +y, +z, +x+y, +y+z, +x+z, +x+y+z).  The operator
//     is symmetric, so the backward coefficient towards a neighbour is that neighbour's forward coefficient: towards -x
//     it sits in the adjacent lane, towards -y in the row above in this wave's registers (one extra row of four slots is
//     loaded above the first row), towards the plane below it was loaded one step ago and is kept (4 slots per row).
// Per node and launch: 8 coefficients x (RY+1)/RY-ish + p x (RY+2)/RY + q = ~100 B instead of 186, and half the load
// instructions.  The values of q are those of var_spmv_kernel bit for bit (same coefficients, same order of the 15
// fused multiply-adds); the block partial sums of p.q are added in another order.  It is NOT the default: it moves
// fewer bytes and takes longer (see beat_vrr_setup); BEAT_VRR=1 selects it, tests compare the two kernels bit for bit.
//
// Replaces, like beat_pde_var.hip, PETSc's MatMult inside KSP.solve (src/beat/base_model.py:236) for operators assembled
// from per-cell conductivity tensors (src/beat/conductivities.py:101-118, demos/biv_endocardial.py:187-282).
#include "beat_pde_internal.h"

#include <algorithm>
#include <cstdlib>
#include <vector>

namespace {
using namespace beat_pde_detail;

constexpr int SEG = 62;      // x-nodes computed per wave and row (lanes 1..62)
constexpr int MAX_RUN = 48;  // planes per work item at most (longer runs of tissue are cut)

struct VrrItem {
  int seg, rb, zb, ze;  // column (x segment, row block) and the planes [zb, ze) it computes
};

struct VrrArgs {
  const double* A;  // (15, ld) coefficient rows
  int64_t ld;
  const unsigned long long* flags;  // bit l of flags[s]: node 64 s + l is a tissue node
  const VrrItem* items;
  int nitems;
  int nx, ny, nz;
  int64_t plane;
  int z_lo_phys, z_hi_phys;
  double* partials;
  int part_off;
  const double* st;
};

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

// forward slots of the 15-point stencil (beat_stencil_offsets): 0 centre, 1 +x, 3 +y, 5 +z, 7 +x+y, 9 +y+z, 11 +x+z,
// 13 +x+y+z; the backward slot k+1 pairs with the forward slot k
template <int RY, bool PF>
__global__ __launch_bounds__(BEAT_BLOCK) void vrr_spmv_kernel(VrrArgs a, const double* __restrict__ X, double* __restrict__ Y,
                                                              const double* __restrict__ A,
                                                              const unsigned long long* __restrict__ FLAGS,
                                                              const VrrItem* __restrict__ ITEMS) {
  constexpr int NR = RY + 2;
  __shared__ double red[4];
  if (a.st[STOP] != 0.0) return;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  double acc = 0.0;
  for (int it = blockIdx.x * 4 + wave; it < a.nitems; it += gridDim.x * 4) {
    const VrrItem item = ITEMS[it];  // wave-uniform
    const int gx = item.seg * SEG - 1 + lane;
    const int y0 = item.rb * RY - 1;  // global row of register row 0
    const bool x_in = gx >= 0 && gx < a.nx;
    const bool x_out = x_in && lane >= 1 && lane <= SEG;
    const int cx = min(max(gx, 0), a.nx - 1);
    bool row_in[NR];
    int off[NR];  // in-plane offset of the (clamped, always addressable) element this lane loads of each row
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const int gy = y0 + r;
      row_in[r] = x_in && gy >= 0 && gy < a.ny;
      off[r] = min(max(gy, 0), a.ny - 1) * a.nx + cx;
    }
    // p of one plane, rows 0..NR-1 of this wave's window: 0 outside the box and on the ghost planes of a physical face
    auto load_plane = [&](int k, double (&dst)[NR]) {
      const bool zok = (k >= 0 || !a.z_lo_phys) && (k < a.nz || !a.z_hi_phys);
      const int cz = min(max(k, -1), a.nz);
      const double* __restrict__ bx = X + (int64_t)cz * a.plane;
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        const double v = bx[off[r]];
        dst[r] = (zok && row_in[r]) ? v : 0.0;
      }
    };
    // forward coefficient `slot` of this lane's node in row r of plane k (0 outside the slab's own planes and the box)
    auto coef = [&](int slot, int r, int k) -> double {
      const int ck = min(max(k, 0), a.nz - 1);
      const double v = A[(int64_t)slot * a.ld + (int64_t)ck * a.plane + off[r]];
      return (row_in[r] && k >= 0 && k < a.nz) ? v : 0.0;
    };
    double Xm[NR], X0[NR], Xp[NR];
    load_plane(item.zb - 1, Xm);
    load_plane(item.zb, X0);
    load_plane(item.zb + 1, Xp);
    // coefficients of the plane below the first one, towards this one: K5 / K11 of the own rows (register rows 1..RY),
    // K9 / K13 of the rows above them (register rows 0..RY-1)
    double K5[RY], K11[RY], K9[RY], K13[RY];
#pragma unroll
    for (int j = 0; j < RY; ++j) {
      K5[j] = coef(5, j + 1, item.zb - 1);
      K11[j] = coef(11, j + 1, item.zb - 1);
      K9[j] = coef(9, j, item.zb - 1);
      K13[j] = coef(13, j, item.zb - 1);
    }
    // PF: the coefficients of plane z + 1 are fetched while plane z is computed (one more set of registers)
    double Fn[RY][8], Hn[4];
    auto load_coefs = [&](int z, double (&F_)[RY][8], double (&H_)[4]) {
#pragma unroll
      for (int j = 0; j < RY; ++j) {
        F_[j][0] = coef(0, j + 1, z);
        F_[j][1] = coef(1, j + 1, z);
        F_[j][2] = coef(3, j + 1, z);
        F_[j][3] = coef(5, j + 1, z);
        F_[j][4] = coef(7, j + 1, z);
        F_[j][5] = coef(9, j + 1, z);
        F_[j][6] = coef(11, j + 1, z);
        F_[j][7] = coef(13, j + 1, z);
      }
      H_[0] = coef(3, 0, z);
      H_[1] = coef(7, 0, z);
      H_[2] = coef(9, 0, z);
      H_[3] = coef(13, 0, z);
    };
    if (PF) load_coefs(item.zb, Fn, Hn);
    for (int z = item.zb; z < item.ze; ++z) {
      // this plane's forward coefficients (own rows) and the four slots of the row above that point down-right
      double F[RY][8], H[4];
      if (PF) {
#pragma unroll
        for (int j = 0; j < RY; ++j)
#pragma unroll
          for (int k = 0; k < 8; ++k) F[j][k] = Fn[j][k];
#pragma unroll
        for (int k = 0; k < 4; ++k) H[k] = Hn[k];
        if (z + 1 < item.ze) load_coefs(z + 1, Fn, Hn);
      } else {
        load_coefs(z, F, H);
      }
      const double H3 = H[0], H7 = H[1], H9 = H[2], H13 = H[3];
      // a slab whose lower neighbour is another rank's: the rows of the plane below are not stored here, the first
      // plane reads its own backward slots instead (as var_spmv_kernel does)
      const bool direct_back = z == 0 && !a.z_lo_phys;
      double Xn[NR];
#pragma unroll
      for (int r = 0; r < NR; ++r) Xn[r] = 0.0;
      if (z + 2 <= item.ze) load_plane(z + 2, Xn);  // in flight while this plane is computed
      // tissue bits of the own rows
      bool tissue[RY];
#pragma unroll
      for (int j = 0; j < RY; ++j) {
        const int64_t gi = (int64_t)z * a.plane + off[j + 1];
        tissue[j] = x_out && row_in[j + 1] && ((FLAGS[gi >> 6] >> (gi & 63)) & 1ull);
      }
      double L0[NR], R0[NR], Rp[NR], Lm[NR];
#pragma unroll
      for (int r = 0; r < NR - 1; ++r) {
        L0[r] = from_left(X0[r]);
        Lm[r] = from_left(Xm[r]);
      }
#pragma unroll
      for (int r = 1; r < NR; ++r) {
        R0[r] = from_right(X0[r]);
        Rp[r] = from_right(Xp[r]);
      }
#pragma unroll
      for (int j = 0; j < RY; ++j) {
        const int r = j + 1;
        double c[15], v[15];
        c[0] = F[j][0];
        c[1] = F[j][1];
        c[3] = F[j][2];
        c[5] = F[j][3];
        c[7] = F[j][4];
        c[9] = F[j][5];
        c[11] = F[j][6];
        c[13] = F[j][7];
        c[2] = from_left(F[j][1]);                          // -x: the left neighbour's +x
        c[4] = j == 0 ? H3 : F[j > 0 ? j - 1 : 0][2];       // -y: the upper neighbour's +y
        c[8] = from_left(j == 0 ? H7 : F[j > 0 ? j - 1 : 0][4]);  // -x-y
        if (direct_back) {
          c[6] = coef(6, r, z);
          c[10] = coef(10, r, z);
          c[12] = coef(12, r, z);
          c[14] = coef(14, r, z);
        } else {
          c[6] = K5[j];               // -z: the lower neighbour's +z
          c[10] = K9[j];              // -y-z
          c[12] = from_left(K11[j]);  // -x-z
          c[14] = from_left(K13[j]);  // -x-y-z
        }
        v[0] = X0[r];
        v[1] = R0[r];
        v[2] = L0[r];
        v[3] = X0[r + 1];
        v[4] = X0[r - 1];
        v[5] = Xp[r];
        v[6] = Xm[r];
        v[7] = R0[r + 1];
        v[8] = L0[r - 1];
        v[9] = Xp[r + 1];
        v[10] = Xm[r - 1];
        v[11] = Rp[r];
        v[12] = Lm[r];
        v[13] = Rp[r + 1];
        v[14] = Lm[r - 1];
        // values are selected, never multiplied by a zero coefficient: a stale ghost plane cannot leak a NaN
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < 15; ++k) s = fma(c[k], (k == 0 || c[k] != 0.0) ? v[k] : 0.0, s);
        if (tissue[j]) {
          Y[(int64_t)z * a.plane + (int64_t)(y0 + r) * a.nx + gx] = s;
          acc = fma(v[0], s, acc);
        }
      }
      // roll: this plane becomes the plane below
#pragma unroll
      for (int j = 0; j < RY; ++j) {
        K5[j] = F[j][3];
        K11[j] = F[j][6];
        K9[j] = j == 0 ? H9 : F[j > 0 ? j - 1 : 0][5];
        K13[j] = j == 0 ? H13 : F[j > 0 ? j - 1 : 0][7];
      }
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        Xm[r] = X0[r];
        X0[r] = Xp[r];
        Xp[r] = Xn[r];
      }
    }
  }
  const double s0 = beat_block_sum(acc, red);
  if (threadIdx.x == 0) a.partials[a.part_off + blockIdx.x] = s0;
}

struct VrrData {
  int ry = 2;
  bool pf = true;
  VrrItem* d_items = nullptr;     // [whole | interior | boundary] lists, back to back
  int first[3] = {0, 0, 0}, count[3] = {0, 0, 0};
  unsigned long long* d_flags = nullptr;
  unsigned resident = 0;
};

template <int RY, bool PF>
unsigned resident_blocks_of() {
  int dev = 0, cus = 256, per_cu = 1;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, vrr_spmv_kernel<RY, PF>, BEAT_BLOCK, 0) != hipSuccess || per_cu < 1) per_cu = 1;
  return (unsigned)(cus * per_cu);
}
}  // namespace

void beat_vrr_destroy(beat_pde* pde) {
  VrrData* d = (VrrData*)pde->vrr;
  if (d == nullptr) return;
  if (d->d_items) (void)hipFree(d->d_items);
  if (d->d_flags) (void)hipFree(d->d_flags);
  delete d;
  pde->vrr = nullptr;
}

// flags: tissue bits per 64-node segment of the slab (host copy).  Builds, per column of the wave decomposition, the
// runs of planes that hold tissue -- for the whole slab and, for decomposed grids, for the planes that need no ghost
// data and for the one or two that do.
int beat_vrr_setup(beat_pde* pde, const std::vector<unsigned long long>& flags) {
  // Opt-in (BEAT_VRR=1).  Measured in round 3, same box, against var_spmv_kernel: 401^3 shell (17 M tissue nodes) 0.93
  // against 0.59 ms, dense 257^3 box with a fibre field 0.71 against 0.45 ms -- with 2 or 4 rows per wave, with and
  // without the coefficients of the next plane in flight, runs of 12 to 48 planes.  The bytes are down as designed (24
  // wave loads per 124 nodes instead of 23 per 64), the time is not: the window of p, the forward rows, the kept slots and
  // the prefetched plane are 180-240 VGPRs (two waves per SIMD at two rows per wave, one at four), and on a shell 37 % of
  // the 62 x RY footprint of a wave lies outside the tissue, which the stored-row kernel's per-lane masks never load.
  const char* on = std::getenv("BEAT_VRR");
  if (!(on && on[0] == '1')) return BEAT_OK;
  const Geom& f = pde->g;
  if (f.nz < 1 || f.nx < 2) return BEAT_OK;
  VrrData* d = new VrrData();
  {
    const char* e = std::getenv("BEAT_VRR_RY");
    const int v = e ? std::atoi(e) : 2;
    d->ry = v == 4 ? 4 : 2;
    const char* pfe = std::getenv("BEAT_VRR_PF");
    d->pf = !(pfe && pfe[0] == '0');
  }
  int max_run = MAX_RUN;
  if (const char* e = std::getenv("BEAT_VRR_RUN")) max_run = std::max(1, std::atoi(e));
  const int RY = d->ry;
  const int nsegx = (f.nx + SEG - 1) / SEG, nrb = (f.ny + RY - 1) / RY;
  auto tissue = [&](int64_t i) { return (flags[(size_t)(i >> 6)] >> (i & 63)) & 1ull; };
  std::vector<VrrItem> lists[3];
  std::vector<char> act((size_t)f.nz);
  const int lo = f.z_lo_phys ? 0 : 1, hi = f.nz - (f.z_hi_phys ? 0 : 1);
  for (int rb = 0; rb < nrb; ++rb)
    for (int seg = 0; seg < nsegx; ++seg) {
      const int x0 = seg * SEG, x1 = std::min(f.nx, x0 + SEG), ya = rb * RY, yb = std::min(f.ny, ya + RY);
      bool any = false;
      for (int z = 0; z < f.nz; ++z) {
        char on = 0;
        for (int y = ya; y < yb && !on; ++y) {
          const int64_t base = (int64_t)z * f.plane + (int64_t)y * f.nx;
          for (int x = x0; x < x1; ++x)
            if (tissue(base + x)) {
              on = 1;
              break;
            }
        }
        act[(size_t)z] = on;
        any |= on != 0;
      }
      if (!any) continue;
      auto runs = [&](int z_lo, int z_hi, std::vector<VrrItem>& out) {
        int z = z_lo;
        while (z < z_hi) {
          if (!act[(size_t)z]) {
            ++z;
            continue;
          }
          int e = z + 1, last = z + 1;  // grow the run over gaps of up to two planes
          while (e < z_hi && e - z < max_run && (act[(size_t)e] || e - last < 2)) {
            if (act[(size_t)e]) last = e + 1;
            ++e;
          }
          out.push_back(VrrItem{seg, rb, z, last});
          z = last;
        }
      };
      runs(0, f.nz, lists[0]);
      runs(lo, std::max(lo, hi), lists[1]);
      if (!f.z_lo_phys) runs(0, 1, lists[2]);
      if (!f.z_hi_phys && (f.nz > 1 || f.z_lo_phys)) runs(f.nz - 1, f.nz, lists[2]);
    }
  std::vector<VrrItem> all;
  for (int k = 0; k < 3; ++k) {
    // long runs first: the tail of the launch is made of short ones
    std::stable_sort(lists[k].begin(), lists[k].end(), [](const VrrItem& p, const VrrItem& q) { return p.ze - p.zb > q.ze - q.zb; });
    d->first[k] = (int)all.size();
    d->count[k] = (int)lists[k].size();
    all.insert(all.end(), lists[k].begin(), lists[k].end());
  }
  pde->vrr = d;
  hipError_t e = hipMalloc(&d->d_items, sizeof(VrrItem) * std::max<size_t>(1, all.size()));
  if (e == hipSuccess && !all.empty()) e = hipMemcpy(d->d_items, all.data(), sizeof(VrrItem) * all.size(), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMalloc(&d->d_flags, sizeof(unsigned long long) * std::max<size_t>(1, flags.size()));
  if (e == hipSuccess && !flags.empty())
    e = hipMemcpy(d->d_flags, flags.data(), sizeof(unsigned long long) * flags.size(), hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    beat_vrr_destroy(pde);
    beat_set_error("beat_vrr_setup: %s", hipGetErrorString(e));
    return BEAT_EHIP;
  }
  d->resident = RY == 4 ? (d->pf ? resident_blocks_of<4, true>() : resident_blocks_of<4, false>())
                        : (d->pf ? resident_blocks_of<2, true>() : resident_blocks_of<2, false>());
  if (std::getenv("BEAT_VRR_VERBOSE")) {
    long planes = 0;
    for (int i = 0; i < d->count[0]; ++i) planes += all[(size_t)i].ze - all[(size_t)i].zb;
    std::fprintf(stderr, "vrr: RY %d, prefetch %d, %d columns x rows, %d items (whole slab), %.1f planes per item, %u resident blocks\n", RY,
                 (int)d->pf, nsegx * nrb, d->count[0], d->count[0] ? (double)planes / d->count[0] : 0.0, d->resident);
  }
  return BEAT_OK;
}

bool beat_vrr_available(const beat_pde* pde) { return pde->var && pde->vrr != nullptr; }

// part: -1 = whole slab, 0 = the planes that need no ghost data (no reduction), 1 = the boundary planes + the reduction
int beat_vrr_spmv_dot(beat_pde* pde, const double* dev_p, double* dev_q, double* dev_st, int part) {
  VrrData* d = (VrrData*)pde->vrr;
  const Geom& f = pde->g;
  VrrArgs a{};
  a.A = pde->v_A;
  a.ld = pde->v_ld;
  a.flags = d->d_flags;
  a.nx = f.nx;
  a.ny = f.ny;
  a.nz = f.nz;
  a.plane = f.plane;
  a.z_lo_phys = f.z_lo_phys;
  a.z_hi_phys = f.z_hi_phys;
  a.partials = pde->ctx->d_partials;
  a.st = dev_st;
  auto launch = [&](int list, int part_off) -> int {
    const int n = d->count[list];
    if (n == 0) return 0;
    a.items = d->d_items + d->first[list];
    a.nitems = n;
    a.part_off = part_off;
    const unsigned grid = std::min<unsigned>(d->resident, (unsigned)((n + 3) / 4));
    if (d->ry == 4 && d->pf)
      BEAT_KERNEL((vrr_spmv_kernel<4, true>), dim3(grid), dim3(BEAT_BLOCK), 0, pde->ctx->stream, a, dev_p, dev_q, a.A, a.flags, a.items);
    else if (d->ry == 4)
      BEAT_KERNEL((vrr_spmv_kernel<4, false>), dim3(grid), dim3(BEAT_BLOCK), 0, pde->ctx->stream, a, dev_p, dev_q, a.A, a.flags, a.items);
    else if (d->pf)
      BEAT_KERNEL((vrr_spmv_kernel<2, true>), dim3(grid), dim3(BEAT_BLOCK), 0, pde->ctx->stream, a, dev_p, dev_q, a.A, a.flags, a.items);
    else
      BEAT_KERNEL((vrr_spmv_kernel<2, false>), dim3(grid), dim3(BEAT_BLOCK), 0, pde->ctx->stream, a, dev_p, dev_q, a.A, a.flags, a.items);
    return (int)grid;
  };
  if (part < 0) {
    const int nb = launch(0, 0);
    BEAT_LAUNCH_CHECK();
    return beat_pde_launch_reduce(pde, nb, 1, dev_st + PQ, dev_st);
  }
  if (part == 0) {
    pde->vrr_part_blocks = launch(1, 0);
    BEAT_LAUNCH_CHECK();
    return BEAT_OK;
  }
  int off = pde->vrr_part_blocks;
  off += launch(2, off);
  BEAT_LAUNCH_CHECK();
  BEAT_REQUIRE(off <= BEAT_MAX_PARTIALS, "too many block partials");
  return beat_pde_launch_reduce(pde, off, 1, dev_st + PQ, dev_st);
}
