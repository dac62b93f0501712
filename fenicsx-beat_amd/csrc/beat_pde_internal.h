// Internal declarations shared by the two translation units of the diffusion step:
//   beat_pde.hip      constant-coefficient (27 node types) stencil kernels, PCG kernels, the C ABI
//   beat_pde_var.hip  per-node-coefficient operators, device-side row assembly, Dirichlet elimination
#pragma once
#include "beat_common.h"

#include <algorithm>
#include <cstdlib>
#include <vector>

namespace beat_pde_detail {

// slots of the PCG scalar state `st` (device, caller-owned, >= 16 doubles)
enum St { BB = 0, RZ, RR, PQ, RZN, RRN, TOL2, BETA, STOP, ITERS, REASON, RTOL, ATOL, MAXIT, NUPD, RR0, ALPHA };  // RR0: r.r of the initial guess
// ALPHA: the step length of the single-reduction iteration (beat_rr_merged_next), device side only: the host reads the first
// 16 entries.  The operator's own scalar state (beat_pde::d_st) has BEAT_ST_DOUBLES entries.
constexpr int BEAT_ST_DOUBLES = 32;
constexpr int PRING = 6;  // search directions kept by the deferred-x PCG before x must be brought up to date (default ring)
// Per-node-row operators on a single slab keep PRING_MAX directions (beat_pde::ring): their solves take 9 - 12 iterations at the
// reference's dt (profiles/r05_shell_guess.md), and with a ring of 6 every one of them paid an in-loop flush -- x and the guess's
// fields read and written a second time, 48 B/node per step.  Kernels size their arrays by PRING_MAX and take the ring as an argument.
constexpr int PRING_MAX = 12;
constexpr int TABW = 16;  // padded row width of the device coefficient tables

extern const int kOffsets[45];  // (dx, dy, dz) of the 15 stencil points

struct Geom {
  int nx, ny, nz;
  int64_t plane;
  int tiles_x, tiles_y, nchunks, zc, total;
  int z_lo_phys, z_hi_phys;
  int tile_tx;  // 64, 128 or 256: which Tile<> instantiation the grid was sized for
  int z_lo, z_hi;    // planes [z_lo, z_hi) computed by this launch (whole slab: 0, nz)
  int part_off;      // first block-partial slot this launch writes
};

enum { MODE_APPLY = 0, MODE_SPMV_DOT = 1, MODE_RHS = 2, MODE_PC = 3 };

}  // namespace beat_pde_detail

struct beat_pde {
  beat_ctx* ctx = nullptr;
  beat_pde_detail::Geom g{};
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
  double* d_alphas = nullptr;  // PRING_MAX step lengths of the deferred-x PCG
  // A solve that has been ENQUEUED and not yet looked at by the host (beat_pde_solve_begin / _end, round 5): right-hand side, the
  // iterations the previous solve needed + 1 and the copy of the scalar state are in the stream, `ev_st` marks the copy.  The next
  // ionic launch may be enqueued behind it before the host waits (PendingV::dev_st): the device does not idle while the host wakes up.
  struct OpenSolve {
    bool on = false;
    int kind = 0;  // 0: register-row kernels (constant coefficients), 1: per-node rows
    bool pdot = false;
    const double* v_prev = nullptr;
    double* x = nullptr;
    double* work = nullptr;
    double rtol = 0.0, atol = 0.0;
    int max_it = 0, launched = 0;
    // a decomposed solve (beat_dist.hip: beat_dist_solve_begin / _end): its communicator and which of its loops runs
    void* comm = nullptr;
    bool rr = false, merged = false, vpdot = false;
    int limit = 0;
  } open;
  // set by beat_solve_begin around its right-hand side: the start of the solve (pcg_begin_kernel's step) is to run in the launch that
  // sums the right-hand side's partials; `done` says a right-hand side took it up
  struct FuseBegin {
    bool on = false, done = false;
    double rtol = 0.0, atol = 0.0;
    int max_it = 0;
  } fuse_begin;
  double* h_st = nullptr;       // pinned copy of the scalar state of the open solve (16 doubles)
  hipEvent_t ev_st = nullptr;   // recorded behind that copy
  beat_ksp_info last_info{};    // of the last solve that was finished
  bool applied_behind = false;  // its x update was applied by the launch enqueued behind it, with these terms (beat_pde_guess_traffic)
  beat_pde_detail::GuessTerms applied_terms{};
  int last_rc = 0;
  int ring = beat_pde_detail::PRING;  // search directions kept before x is brought up to date (6; 12: per-node rows on a single slab)
  int last_base = 0;                  // first iteration of the ring cycle the last deferring solve left pending
  double* d_batch_st = nullptr;  // scalar states of the solves of a beat_split_steps batch (BEAT_MAX_BATCH x 16)
  // z node type of the ghost planes (the neighbouring slabs' boundary planes): 1 unless that plane is a face of the
  // whole grid (a neighbour that owns a single plane); set with beat_pde_set_ghost_types
  int ghost_lo_tz = 1, ghost_hi_tz = 1;
  // initial guess from the previous solves' increments (0: x0 = v_; m: + the degree-(m-1) extrapolation of the last m), see GuessTerms
  int single_reduction = -1;               // decomposed solve: 1 one all-reduce per iteration, 0 two, -1 as BEAT_DIST_MERGED says
  int guess_order = 0;                     // as configured: 0..4, or -1 = choose between 3 and 4 per solve (below)
  // adaptive choice (guess_order = -1).  No order is right everywhere: each recorded increment carries an rtol-sized
  // error, which an extrapolation of order m amplifies by the sum of its |coefficients| (1, 3, 7, 15) -- where the
  // increments are smooth in time (plateau, repolarisation, rest) that noise sets the initial residual and the lowest
  // order wins (0.3 iterations per step against 1.6), on a travelling front the truncation error does and the cubic
  // wins (3.6 against 8).  Hill climbing on the order: a running mean of the iteration count per order, the current
  // order used, one of its neighbours tried every 12th solve (up and down in turn), the move made when the neighbour
  // has been costing fewer iterations.  Iteration counts are global: every rank of a decomposed solve decides alike.
  int auto_cur = 3;                        // order the policy currently favours
  int auto_next = 3;                       // order of the guess the NEXT x update prepares (auto_cur or a probe)
  int auto_e_order = 0;                    // order the guess now in e was built with (0: none / not adaptive)
  double auto_score[4] = {0.0, 0.0, 0.0, 0.0};  // running mean of the iterations per solve for orders 1..4
  int auto_seen[4] = {0, 0, 0, 0};
  int auto_since_probe = 0, auto_probe_up = 1;
  double* d_hist[3] = {nullptr, nullptr, nullptr};  // fields with ghost planes: the last increments, newest first
  double* d_guess = nullptr;               // the guess increment e prepared for the next solve
  double* d_hist_alloc = nullptr;
  int hist_fields = 0;                     // fields in d_hist_alloc
  int hist_n = 0;                          // solves recorded since the history was last dropped (capped at the maximal order)
  beat_pde_detail::GuessTerms guess{};     // terms of the solve in progress (out == nullptr: not in use)
  bool guess_pending = false;              // the last solve left x += e + sum alpha_j p_j to its caller ...
  beat_pde_detail::GuessTerms guess_final{};  // ... with these terms
  int rhs_part_blocks = 0;  // block partials written by part 0 of a right-hand side built in two parts
  void* vrr = nullptr;      // work lists of the z-marching per-node SpMV (beat_pde_vrr.hip), or nullptr
  int vrr_part_blocks = 0;
  void* vtl = nullptr;      // tiles and lane masks of the workgroup-tile per-node SpMV (beat_pde_vtl.hip), or nullptr
  bool small_enabled = true;  // grids of a few thousand nodes: whole solve in one launch (beat_pde_small.hip)
  int pc_ncoef = 1;       // 1: Jacobi; m >= 2: Chebyshev polynomial of degree m-1 in D^-1 A (m-1 stencil passes)
  double pc_coef[8] = {1.0};
  // variable-coefficient mode (beat_pde_create_var): caller-owned Mass / K rows, A and 1/diag owned here
  bool var = false;
  const double* v_mass = nullptr;
  const double* v_stiff = nullptr;
  double* v_A = nullptr;
  double* v_dinv = nullptr;
  double* v_B = nullptr;  // rows of B = C_m Mass - (1 - theta) dt K, formed beside A when the right-hand side runs on the tiles (beat_vtl_rhs)
  // a slab with live neighbours: the centre coefficients of the neighbours' boundary planes ([0, plane): below, [plane, 2 plane):
  // above), exchanged by beat_pde_solve_dist when v_gc0_valid is false (the operator changed) -- what the fused tile pass needs to
  // form the search direction on the ghost planes (beat_vtl_pdot_part)
  double* v_gc0 = nullptr;
  bool v_gc0_valid = false;
  bool v_pdot_dist = false;  // all ranks of the decomposition can run the fused tile pass (agreed when v_gc0 is refreshed)
  int64_t v_ld = 0;
  int* v_seg = nullptr;        // device: indices of the 64-node segments that hold tissue nodes (ascending)
  std::vector<int> h_seg;      // host copy (sub-ranges are located by binary search)
  unsigned long long* v_segmask = nullptr;  // device: per list entry, bit l set = node 64 seg + l is a tissue node
  int* v_seg_tiled = nullptr;               // the same list (and masks) ordered by tiles of T rows x T planes (BEAT_VAR_TILE, experiments)
  unsigned long long* v_segmask_tiled = nullptr;
  const double* d_tab(int which) const { return d_tabs + (size_t)which * 27 * beat_pde_detail::TABW; }
  const double* d_dinv() const { return d_tabs + (size_t)4 * 27 * beat_pde_detail::TABW; }
  const double* dinv_arg() const { return var ? v_dinv : d_dinv(); }
};

// Iterations enqueued before the host first looks at the convergence latch: the previous solve's count plus one.  A
// latched iteration costs four empty launches (~20 us); one short costs a host round trip with the GPU idle plus a
// second one after the catch-up iterations, and consecutive time steps differ by one iteration all the time.
inline int beat_pde_first_chunk(const beat_pde* pde) {
  static const int extra = [] {
    const char* e = std::getenv("BEAT_CHUNK_EXTRA");
    return e ? std::max(0, std::atoi(e)) : 1;
  }();
  return pde->last_iters >= 0 ? std::max(1, pde->last_iters + extra) : 8;
}

// fixed-order sum of `count` block partials of `nsum` quantities into out[0..nsum) (beat_pde.hip)
// then / roll_st / rtol / atol / max_it: the scalar step that follows the sums, in the same launch (1: the iteration's roll, 2: the start of a solve)
int beat_pde_launch_reduce(beat_pde* pde, int count, int nsum, double* out, const double* st, double* counter = nullptr, int then = 0,
                           double* roll_st = nullptr, double rtol = 0.0, double atol = 0.0, int max_it = 0);

// per-node-coefficient variants of the stage operations (beat_pde_var.hip); same contracts as the beat_pde_*
// entry points that dispatch to them
int beat_var_form_A(beat_pde* pde);
int beat_var_apply(beat_pde* pde, int which, const double* dev_x, double* dev_y);
int beat_var_rhs(beat_pde* pde, const double* dev_v_prev, const double* const* host_dev_stim_w,
                 const double* host_stim_amp, int n_stim, double* dev_x, double* dev_r, double* dev_p, double* dev_red,
                 const double* dev_e = nullptr,  // dev_e: initial-guess increment (r = b - A (v_ + e)) or nullptr
                 int part = -1);  // decomposed grids: 0 = the planes that need no ghost data, 1 = the boundary planes + the sums; -1: all
int beat_var_spmv_dot(beat_pde* pde, const double* dev_p, double* dev_q, double* dev_st);
int beat_var_spmv_dot_part(beat_pde* pde, const double* dev_p, double* dev_q, double* dev_st, int part);
int beat_var_update_r(beat_pde* pde, double* dev_st, double* dev_r, const double* dev_q, int slot, bool roll = false);
int beat_var_pupdate_oop(beat_pde* pde, double* dev_st, const double* dev_r, const double* dev_p_cur, double* dev_p_next);
int beat_var_flush(beat_pde* pde, const double* dev_st, double* dev_x, const double* dev_ring0, int64_t field_stride,
                   int ring_base, int only_if_full, const beat_pde_detail::GuessTerms& gt);

// initial-guess bookkeeping (beat_pde.hip).  A solve path that supports the guess calls beat_guess_begin before its
// right-hand side, passes beat_guess_terms(pde, ring_base) to every x update and ends with beat_guess_end; every
// other path calls beat_guess_skip (the history does not survive a solve that did not record its increment).
void beat_guess_begin(beat_pde* pde);
void beat_guess_skip(beat_pde* pde);
beat_pde_detail::GuessTerms beat_guess_terms(const beat_pde* pde, int ring_base);
// nupd = executed updates; returns true when an application (e and/or the last partial ring cycle) is still due
bool beat_guess_end(beat_pde* pde, int nupd, bool deferred);
int beat_pde_x_flush_terms(beat_pde* pde, const double* dev_st, double* dev_x, const double* dev_ring0, int64_t field_stride,
                           int ring_base, int only_if_full, const beat_pde_detail::GuessTerms& gt);

void beat_guess_advance(beat_pde* pde);
void beat_guess_observe(beat_pde* pde, int iterations);  // of the solve that just ended (adaptive order)
int beat_guess_policy(beat_pde* pde);                     // hill-climbing move; returns the order to prepare next  // this solve's increment has been recorded: it is the most recent one now

// per-node-coefficient SpMV that marches along z and loads only the forward half of each row (beat_pde_vrr.hip)
int beat_vrr_setup(beat_pde* pde, const std::vector<unsigned long long>& host_tissue_flags);
void beat_vrr_destroy(beat_pde* pde);
bool beat_vrr_available(const beat_pde* pde);
int beat_vrr_spmv_dot(beat_pde* pde, const double* dev_p, double* dev_q, double* dev_st, int part);

// per-node-coefficient SpMV on workgroup tiles: forward coefficients and rows of p loaded once per tile, shared through
// LDS and registers (beat_pde_vtl.hip); whole-slab launches only
int beat_vtl_setup(beat_pde* pde, const std::vector<unsigned long long>& host_tissue_flags);
void beat_vtl_destroy(beat_pde* pde);
bool beat_vtl_available(const beat_pde* pde);
int beat_vtl_spmv_dot(beat_pde* pde, const double* dev_p, double* dev_q, double* dev_st);
bool beat_vtl_parts_available(const beat_pde* pde);
bool beat_vtl_pdot_dist_available(const beat_pde* pde);
int beat_vtl_pdot_part(beat_pde* pde, double* dev_st, const double* dev_r, const double* dev_p_old, double* dev_p_new, double* dev_q,
                       int first, int part);
int beat_vtl_spmv_dot_part(beat_pde* pde, const double* dev_p, double* dev_q, double* dev_st, int part);
bool beat_vtl_pdot_available(const beat_pde* pde);
bool beat_vtl_rhs_available(const beat_pde* pde);
bool beat_vtl_rhs_wanted(const beat_pde* pde);
int beat_vtl_rhs(beat_pde* pde, const double* dev_v_prev, const double* const* host_dev_stim_w, const double* host_stim_amp, int n_stim,
                 double* dev_x, double* dev_r, double* dev_p, double* dev_t, double* dev_red, const double* dev_e);
int beat_vtl_pdot(beat_pde* pde, double* dev_st, const double* dev_r, const double* dev_p_old, double* dev_p_new, double* dev_q, int first);

// the two halves of beat_pde_solve_ex for the Jacobi paths (beat_pde.hip): enqueue without waiting / wait, iterate on if needed, do
// the host's bookkeeping.  beat_solve_end: *needed_more = the first look found the solve unlatched (a launch enqueued behind it
// with PendingV::dev_st has done nothing); a solve that was never opened: the last finished solve's record
bool beat_solve_lazy_available(const beat_pde* pde);
struct beat_comm;
int beat_dist_solve_begin(beat_pde* pde, beat_comm* comm, const double* dev_v_prev, const double* const* host_dev_stim_w,
                          const double* host_stim_amp, int n_stim, double* dev_x, double* dev_work, double rtol, double atol, int max_it);
int beat_dist_solve_end(beat_pde* pde, int defer_flush, beat_ksp_info* info, int* host_pending, bool* needed_more);
int beat_solve_begin(beat_pde* pde, const double* dev_v_prev, const double* const* host_dev_stim_w, const double* host_stim_amp, int n_stim,
                     double* dev_x, double* dev_work, double rtol, double atol, int max_it);
int beat_solve_end(beat_pde* pde, int defer_flush, beat_ksp_info* info, int* host_pending, bool* needed_more);

// one-workgroup solve of small constant-coefficient grids (beat_pde_small.hip)
bool beat_small_available(const beat_pde* pde);
int beat_small_launch(beat_pde* pde, const double* dev_v_prev, const double* const* host_dev_stim_w,
                      const double* host_stim_amp, int n_stim, double* dev_x, double rtol, double atol, int max_it,
                      double* dev_st);
int beat_small_solve(beat_pde* pde, const double* dev_v_prev, const double* const* host_dev_stim_w,
                     const double* host_stim_amp, int n_stim, double* dev_x, double rtol, double atol, int max_it,
                     beat_ksp_info* info);

// register-row kernels of the constant-coefficient Jacobi-PCG that never stores q = A p (beat_pde_rr.hip)
bool beat_rr_available(const beat_pde* pde);
int beat_rr_rhs(beat_pde* pde, const double* dev_v_prev, const double* const* host_dev_stim_w, const double* host_stim_amp,
                int n_stim, double* dev_x, double* dev_r, double* dev_st, int part = -1);  // part: as beat_var_rhs
int beat_rr_pdot(beat_pde* pde, double* dev_st, const double* dev_r, const double* dev_p_old, double* dev_p_new);
int beat_rr_pdot_part(beat_pde* pde, double* dev_st, const double* dev_r, const double* dev_p_old, double* dev_p_new,
                      int part);
int beat_rr_rupd(beat_pde* pde, double* dev_st, const double* dev_r, double* dev_r_new, const double* dev_p, int slot,
                 bool roll = true);
int beat_rr_next(beat_pde* pde, double* dev_st);
// the single-reduction (Chronopoulos-Gear) iteration of a decomposed solve: see beat_pde_rr.hip
int beat_rr_udot_part(beat_pde* pde, double* dev_st, const double* dev_r, int part);
int beat_rr_merged_next(beat_pde* pde, double* dev_st, int slot);
int beat_rr_prupd(beat_pde* pde, double* dev_st, const double* dev_r, const double* dev_p_old, double* dev_p_new,
                  double* dev_r_new);
