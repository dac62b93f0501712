// Ionic (reaction) step kernels: one node per thread, state-major SoA rows, fp64.
// Replaces  states[:] = fun(states=, t=, parameters=, dt=)  (src/beat/odesolver.py:67-79).
//
// HBM traffic per node-update: 16*NS bytes (each state row read once, written once) plus 8 bytes
// when the transmembrane potential is mirrored into the PDE vector (dev_v_copy).
#include "beat_ode_kernel.h"
#include "beat_ode_jit.h"

#include <functional>

// (ode_run_kernel -- many steps inside one launch -- lives in beat_ode_kernel.h: models registered as source instantiate it too)

template <class Model>
static int launch_ode_run(beat_ctx* ctx, double* states, int64_t n, int64_t ld, const double* host_params,
                          int num_params, const double* ppn, int64_t pld, double t0, double dt, int64_t nsteps,
                          int nbeats, int save_freq, const int* track_idx, int ntrack, double* trace) {
  BEAT_REQUIRE(num_params == Model::NP || (host_params == nullptr && ppn == nullptr && Model::NP == 2),
               "model expects %d parameters, got %d", Model::NP, num_params);
  BEAT_REQUIRE(ntrack >= 0 && ntrack <= 8, "at most 8 tracked states");
  BEAT_REQUIRE(ntrack == 0 || (trace != nullptr && save_freq >= 1), "tracking needs a trace buffer and save_freq >= 1");
  ParamPack<Model::NP> prm;
  for (int k = 0; k < Model::NP; ++k) prm.p[k] = host_params ? host_params[k] : 1.0;
  typename Model::Derived drv = Model::derive(prm.p);
  TrackSpec tr{};
  tr.n = ntrack;
  for (int a = 0; a < ntrack; ++a) {
    BEAT_REQUIRE(track_idx[a] >= 0 && track_idx[a] < Model::NS, "tracked state %d out of range", track_idx[a]);
    tr.idx[a] = track_idx[a];
  }
  const unsigned grid = (unsigned)((n + BEAT_BLOCK - 1) / BEAT_BLOCK);
  if (ppn != nullptr) {
    BEAT_REQUIRE(pld >= n, "params_ld %lld < n %lld", (long long)pld, (long long)n);
    BEAT_KERNEL((ode_run_kernel<Model, true>), dim3(grid), dim3(BEAT_BLOCK), 0, ctx->stream, states, n, ld, prm,
                       drv, ppn, pld, t0, dt, nsteps, nbeats, save_freq, tr, trace);
  } else {
    BEAT_KERNEL((ode_run_kernel<Model, false>), dim3(grid), dim3(BEAT_BLOCK), 0, ctx->stream, states, n, ld,
                       prm, drv, ppn, pld, t0, dt, nsteps, nbeats, save_freq, tr, trace);
  }
  BEAT_LAUNCH_CHECK();
  return BEAT_OK;
}

extern "C" int beat_ode_run(beat_ctx* ctx, int model_id, double* dev_states, int64_t n, int64_t ld,
                            const double* host_params, int num_params, const double* dev_params_per_node,
                            int64_t params_ld, double t0, double dt, int64_t nsteps, int nbeats, int save_freq,
                            const int* host_track_idx, int ntrack, double* dev_trace) {
  BEAT_REQUIRE(ctx != nullptr && dev_states != nullptr, "null argument");
  BEAT_REQUIRE(n >= 0 && ld >= n && nsteps >= 0 && nbeats >= 0, "bad shape");
  if (n == 0 || nsteps == 0 || nbeats == 0) return BEAT_OK;
  if (model_id >= BEAT_MODEL_CUSTOM_BASE)  // a model registered as source (beat_ode_model_register)
    return beat_custom_run(ctx, model_id, dev_states, n, ld, host_params, num_params, dev_params_per_node, params_ld, t0, dt, nsteps, nbeats,
                           save_freq, host_track_idx, ntrack, dev_trace);
#define BEAT_RUN(M)                                                                                          \
  return launch_ode_run<M>(ctx, dev_states, n, ld, host_params, num_params, dev_params_per_node, params_ld, \
                           t0, dt, nsteps, nbeats, save_freq, host_track_idx, ntrack, dev_trace)
  switch (model_id) {
    case BEAT_MODEL_SIMPLE_ODE: BEAT_RUN(SimpleOde);
    case BEAT_MODEL_FHN_DEMO: BEAT_RUN(FhnDemo);
    case BEAT_MODEL_FHN_README: BEAT_RUN(FhnReadme);
    case BEAT_MODEL_TP06_GRL1: BEAT_RUN(Tp06Grl1);
    case BEAT_MODEL_TORORD_DYNCL_GRL1: BEAT_RUN(TorordDynClGrl1);
    case BEAT_MODEL_TORORD_LAND_GRL1: BEAT_RUN(TorordLandGrl1);
    default: beat_set_error("unknown model id %d", model_id); return BEAT_EINVAL;
  }
#undef BEAT_RUN
}

// blocks of an ionic launch over n nodes
// (num_states: a block's set-up -- 4 KB of tables into LDS, a barrier -- is 2 % of what a TP06 tile moves and 20 - 50 % of what a tile of
// a 2- or 5-state model moves: models below BEAT_ODE_TILE_MIN_STATES (12) keep the capped, looping launch of rounds 2 - 5; the
// generated 5-state test model 0.455 ms per step at 256^3 with a block per tile, 0.43 with 24 576 looping blocks)
static unsigned ode_grid(int64_t n, int num_states) {
  unsigned grid = (unsigned)((n + BEAT_BLOCK - 1) / BEAT_BLOCK);
  // Round 6: ONE BLOCK PER TILE is the default.  Rounds 2 - 5 capped the launch at 24 576 blocks walking ~21 tiles each: at three
  // waves per SIMD a block's set-up (4 KB of tables into LDS, a barrier) was worth amortising (10.5 - 10.6 against 10.9 - 11.3 ms at
  // 512^3).  At four waves per SIMD (TP06, 118 VGPRs) / three (ToR-ORd) the other blocks of the CU cover it, and blocks handed out
  // in order by the dispatcher keep the launch's accesses in one moving window of the state rows -- what the streaming probe shows
  // for one workgroup per chunk (profiles/r05_streaming.md): TP06 512^3 in one process on the same memory 8.70 -> 8.49 - 8.54 ms
  // (16 384 blocks: 8.75, 32 768: 8.65, 8 192: 8.98), class kernel 8.87 -> 8.57, ToR-ORd 256^3 2.49 -> 2.44; the 512^3 step 13.33 -
  // 13.57 -> 13.04 - 13.27 ms, process by process (profiles/r06_ode_addressing.md).  BEAT_ODE_GRID=<blocks> caps the launch again.
  static const int grid_cap_env = [] {  // blocks per launch (BEAT_ODE_GRID; 0: one block per tile; unset: by the model's size)
    const char* e = std::getenv("BEAT_ODE_GRID");
    return e ? std::atoi(e) : -1;
  }();
  static const int tile_min_states = [] {
    const char* e = std::getenv("BEAT_ODE_TILE_MIN_STATES");
    return e ? std::atoi(e) : 12;
  }();
  const int grid_cap = grid_cap_env >= 0 ? grid_cap_env : (num_states >= tile_min_states ? 0 : 24576);
  static const bool balance = [] {  // BEAT_ODE_BALANCE=0: plain cap (A/B runs)
    const char* e = std::getenv("BEAT_ODE_BALANCE");
    return !(e && e[0] == '0');
  }();
  if (grid_cap > 0 && grid > (unsigned)grid_cap) {
    // a few tiles per block: give every block the same number (the last one aside).  With 2.67 tiles per block -- 256^3, or
    // the slab one of 8 ranks owns at 512^3, on 24 576 blocks -- two thirds of the blocks are on a third tile while the
    // others have finished: 1.36-1.47 ms against 1.25-1.32 with 3 tiles each (A B A B A B on one box, round 3).  With
    // 21.3 tiles per block (512^3) the plain cap measures the same or better (9.80 against 9.83 ms) and stays.
    const unsigned per_block = (grid + (unsigned)grid_cap - 1) / (unsigned)grid_cap;
    grid = (balance && per_block <= 8) ? (grid + per_block - 1) / per_block : (unsigned)grid_cap;
  }
  return grid;
}

template <class Model>
static int launch_ode(beat_ctx* ctx, double* states, int64_t n, int64_t ld, const double* host_params,
                      int num_params, const double* ppn, int64_t pld, double t, double dt,
                      int v_index, double* v_copy, const PendingV& pend, MarkedArgs mk = MarkedArgs{nullptr, nullptr, 0, nullptr, nullptr},
                      SparseRows sp = SparseRows{{0}, 0}) {
  BEAT_REQUIRE(num_params == Model::NP || (host_params == nullptr && ppn == nullptr && Model::NP == 2) || mk.markers != nullptr,
               "model expects %d parameters, got %d", Model::NP, num_params);
  // (with num_params == NP and neither a vector nor rows nor classes every parameter would silently be 1.0 -- what the
  // two-parameter test ODE is called with on purpose and nothing else is)
  BEAT_REQUIRE(host_params != nullptr || ppn != nullptr || mk.markers != nullptr || Model::NP == 2,
               "no parameters given: a host vector, per-node rows or parameter classes");
  BEAT_REQUIRE(v_copy == nullptr || (v_index >= 0 && v_index < Model::NS), "v_index %d out of range", v_index);
  BEAT_REQUIRE(mk.markers == nullptr || v_copy == nullptr || v_index == Model::V_INDEX,
               "the class kernel mirrors the model's potential (row %d), not row %d", Model::V_INDEX, v_index);
  const bool have_pend = pend.count > 0 || pend.gt.d != nullptr || pend.dev_st != nullptr;
  BEAT_REQUIRE(!have_pend || v_index == Model::V_INDEX,
               "a pending update needs v_index = %d (the model's membrane potential), got %d", Model::V_INDEX, v_index);
  ParamPack<Model::NP> prm;
  for (int k = 0; k < Model::NP; ++k) prm.p[k] = host_params ? host_params[k] : 1.0;
  typename Model::Derived drv = Model::derive(prm.p);
  const unsigned grid = ode_grid(n, Model::NS);
  const dim3 g3(grid), b3(BEAT_BLOCK);
#define BEAT_LAUNCH_ODE(PN, PD)                                                                                 \
  BEAT_KERNEL((ode_step_kernel<Model, PN, PD>), g3, b3, 0, ctx->stream, states, n, ld, prm, drv, ppn, pld, t, \
                     dt, v_index, v_copy, pend, mk, sp)
  if (mk.markers != nullptr) {
    BEAT_REQUIRE(ppn == nullptr && mk.table != nullptr, "parameter classes come with a table, not with per-node rows");
    if (have_pend)
      BEAT_KERNEL((ode_step_kernel<Model, false, true, true>), g3, b3, 0, ctx->stream, states, n, ld, prm, drv, ppn, pld, t, dt,
                  v_index, v_copy, pend, mk, sp);
    else
      BEAT_KERNEL((ode_step_kernel<Model, false, false, true>), g3, b3, 0, ctx->stream, states, n, ld, prm, drv, ppn, pld, t, dt,
                  v_index, v_copy, pend, mk, sp);
  } else if (ppn != nullptr && sp.count > 0) {
    BEAT_REQUIRE(pld >= n, "params_ld %lld < n %lld", (long long)pld, (long long)n);
    BEAT_REQUIRE(host_params != nullptr, "sparse rows come with the uniform parameter vector");
    // the instance with these indices as compile-time constants, written and compiled at first use (beat_ode_jit.hip); where
    // that is not to be had (no hipcc, BEAT_JIT=0, a model without accessor-style parameters) the run-time-index kernel below
    {
      const int rc = beat_ode_jit_launch<Model>(ctx, g3, have_pend, states, n, ld, prm, drv, ppn, pld, t, dt, v_index, v_copy, pend, mk, sp);
      if (rc != BEAT_JIT_UNAVAILABLE) return rc;
    }
    // (the shipped kernel finds a row's entry by comparison at run time: four rows; more need the compiled instance)
    BEAT_REQUIRE(sp.count <= BEAT_MAX_SPARSE_ROWS_RT, "%d varying rows need run-time compilation, which is not available here "
                 "(beat_ode_jit_stats; at most %d rows otherwise): pass all rows (beat_ode_step)", sp.count, BEAT_MAX_SPARSE_ROWS_RT);
    if (have_pend)
      BEAT_KERNEL((ode_step_kernel<Model, true, true, false, true>), g3, b3, 0, ctx->stream, states, n, ld, prm, drv, ppn, pld, t, dt,
                  v_index, v_copy, pend, mk, sp);
    else
      BEAT_KERNEL((ode_step_kernel<Model, true, false, false, true>), g3, b3, 0, ctx->stream, states, n, ld, prm, drv, ppn, pld, t, dt,
                  v_index, v_copy, pend, mk, sp);
  } else if (ppn != nullptr) {
    BEAT_REQUIRE(pld >= n, "params_ld %lld < n %lld", (long long)pld, (long long)n);
    if (have_pend)
      BEAT_LAUNCH_ODE(true, true);
    else
      BEAT_LAUNCH_ODE(true, false);
  } else {
    if (have_pend)
      BEAT_LAUNCH_ODE(false, true);
    else
      BEAT_LAUNCH_ODE(false, false);
  }
#undef BEAT_LAUNCH_ODE
  BEAT_LAUNCH_CHECK();
  return BEAT_OK;
}

extern "C" int beat_ode_model_info(int model_id, int* num_states, int* num_params) {
  int ns, np;
  switch (model_id) {
    case BEAT_MODEL_SIMPLE_ODE: ns = SimpleOde::NS; np = SimpleOde::NP; break;
    case BEAT_MODEL_FHN_DEMO: ns = FhnDemo::NS; np = FhnDemo::NP; break;
    case BEAT_MODEL_FHN_README: ns = FhnReadme::NS; np = FhnReadme::NP; break;
    case BEAT_MODEL_TP06_GRL1: ns = Tp06Grl1::NS; np = Tp06Grl1::NP; break;
    case BEAT_MODEL_TORORD_DYNCL_GRL1: ns = TorordDynClGrl1::NS; np = TorordDynClGrl1::NP; break;
    case BEAT_MODEL_TORORD_LAND_GRL1: ns = TorordLandGrl1::NS; np = TorordLandGrl1::NP; break;
    default:
      if (beat_custom_model_info(model_id, &ns, &np, nullptr) == BEAT_OK) break;  // a model registered as source
      beat_set_error("unknown model id %d", model_id);
      return BEAT_EINVAL;
  }
  if (num_states) *num_states = ns;
  if (num_params) *num_params = np;
  return BEAT_OK;
}

static int ode_step_dispatch(beat_ctx* ctx, int model_id, double* dev_states, int64_t n, int64_t ld,
                             const double* host_params, int num_params, const double* dev_params_per_node,
                             int64_t params_ld, double t, double dt, int v_index, double* dev_v_copy,
                             const PendingV& pend, MarkedArgs mk = MarkedArgs{nullptr, nullptr, 0, nullptr, nullptr},
                             SparseRows sp = SparseRows{{0}, 0}) {
  BEAT_REQUIRE(ctx != nullptr, "null context");
  BEAT_REQUIRE(dev_states != nullptr, "null states");
  BEAT_REQUIRE(n >= 0 && ld >= n, "bad shape n=%lld ld=%lld", (long long)n, (long long)ld);
  BEAT_REQUIRE((n + BEAT_BLOCK - 1) / BEAT_BLOCK < (int64_t)0x7fffffff, "n too large");
  if (n == 0) return BEAT_OK;
  if (model_id >= BEAT_MODEL_CUSTOM_BASE) {  // a model registered as source (beat_ode_model_register)
    BEAT_REQUIRE(sp.count == 0, "a model registered as source takes a uniform vector, all per-node rows or classes (no sparse rows)");
    int cns = 0, cnp = 0, cvi = 0;
    if (int rc = beat_custom_model_info(model_id, &cns, &cnp, &cvi)) return rc;
    return beat_custom_step(ctx, model_id, ode_grid(n, cns), dev_states, n, ld, host_params, num_params, dev_params_per_node, params_ld, t, dt,
                            v_index, dev_v_copy, pend, mk);
  }
#define BEAT_STEP(M)                                                                                             \
  return launch_ode<M>(ctx, dev_states, n, ld, host_params, num_params, dev_params_per_node, params_ld, t, dt, \
                       v_index, dev_v_copy, pend, mk, sp)
  switch (model_id) {
    case BEAT_MODEL_SIMPLE_ODE: BEAT_STEP(SimpleOde);
    case BEAT_MODEL_FHN_DEMO: BEAT_STEP(FhnDemo);
    case BEAT_MODEL_FHN_README: BEAT_STEP(FhnReadme);
    case BEAT_MODEL_TP06_GRL1: BEAT_STEP(Tp06Grl1);
    case BEAT_MODEL_TORORD_DYNCL_GRL1: BEAT_STEP(TorordDynClGrl1);
    case BEAT_MODEL_TORORD_LAND_GRL1: BEAT_STEP(TorordLandGrl1);
    default: beat_set_error("unknown model id %d", model_id); return BEAT_EINVAL;
  }
#undef BEAT_STEP
}

extern "C" int beat_ode_step(beat_ctx* ctx, int model_id, double* dev_states, int64_t n, int64_t ld,
                             const double* host_params, int num_params,
                             const double* dev_params_per_node, int64_t params_ld, double t, double dt,
                             int v_index, double* dev_v_copy) {
  return ode_step_dispatch(ctx, model_id, dev_states, n, ld, host_params, num_params, dev_params_per_node, params_ld,
                           t, dt, v_index, dev_v_copy, PendingV{nullptr, 0, nullptr, 0, {}});
}

// More pending directions than the plain kernels' pending path takes (a per-node-row operator on a single slab keeps a ring of
// 12, PRING_MAX; the class kernel takes them all, the uniform / per-node kernels 6 -- their pending values stay live through the
// step): the update is applied by the flush pass instead and the launch proceeds without a pending update.
static int flush_long_ring(beat_pde* pde, double* v_row, const double* dev_ring0, int64_t field_stride, int& pending) {
  if (pending <= BEAT_MAX_PENDING) return BEAT_OK;
  beat_pde_detail::GuessTerms gt{};
  if (pde->guess_pending) {
    gt = pde->guess_final;
    pde->guess_pending = false;
  }
  pending = 0;
  return beat_pde_x_flush_terms(pde, nullptr, v_row, dev_ring0, field_stride, pde->last_base, 0, gt);
}

// pending = -1 (round 5): `pde` has an OPEN solve (beat_pde_solve_begin) whose result the host has not looked at.  The launch is
// enqueued behind it at once, told to read what is pending from the solve's scalar state on the device and to do nothing if the
// solve has not latched (PendingV::dev_st); only then does the host wait for the solve (beat_solve_end: its bookkeeping, more
// iterations if the enqueued ones did not suffice).  The device goes from the solve's last kernel straight into the ionic kernel
// instead of idling while the host wakes up, reads the latch and launches (0.15-0.2 ms of a 13.7 ms step at 512^3: what
// MonodomainSplittingSolver.step cost over .solve).  If the solve did need more iterations the first launch was a no-op and the
// step is launched again, with the host's own count.  Values: those of the two separate calls, bit for bit.
static int step_behind_open_solve(beat_ctx* ctx, beat_pde* pde, const double* dev_ring0, int64_t field_stride, int64_t n, double* v_row,
                                  const std::function<int(const PendingV&)>& launch) {
  BEAT_REQUIRE(pde != nullptr && pde->open.on, "pending = -1 needs an operator with an open solve");
  BEAT_REQUIRE(dev_ring0 != nullptr && field_stride >= n, "bad pending update");
  BEAT_REQUIRE(pde->open.x == v_row, "the open solve does not work on this row");
  PendingV behind{dev_ring0, field_stride, pde->d_alphas, 0, pde->guess, pde->d_st, pde->ring};
  const bool plain_kernel_fits = pde->ring <= BEAT_MAX_PENDING;  // (a longer ring: only the class kernel takes it all)
  (void)plain_kernel_fits;
  if (int rc = launch(behind)) return rc;
  beat_ksp_info info{};
  int pend2[2] = {0, 0};
  bool needed_more = false;
  const int rc_solve = beat_solve_end(pde, 1, &info, pend2, &needed_more);
  if (rc_solve != BEAT_OK && rc_solve != BEAT_ENOTCONV) return rc_solve;
  pde->applied_behind = false;
  if (!needed_more) {  // the launch behind the solve has applied the update: nothing is left to the caller
    if (pde->guess_pending) {
      pde->applied_terms = pde->guess_final;
      pde->applied_behind = true;
    }
    pde->guess_pending = false;
    return BEAT_OK;
  }
  // the first launch saw an unlatched solve and returned at once: again, with what the finished solve left
  PendingV pend{dev_ring0, field_stride, pend2[1] ? pde->d_alphas : nullptr, pend2[1], {}, nullptr, 0};
  if (pde->guess_pending) {
    pend.gt = pde->guess_final;
    pde->guess_pending = false;
  }
  return launch(pend);
}

extern "C" int beat_ode_step_pending(beat_ctx* ctx, int model_id, double* dev_states, int64_t n, int64_t ld,
                                     const double* host_params, int num_params,
                                     const double* dev_params_per_node, int64_t params_ld, double t, double dt,
                                     int v_index, double* dev_v_copy, beat_pde* pde, const double* dev_ring0,
                                     int64_t field_stride, int pending) {
  static_assert(BEAT_MAX_PENDING == beat_pde_detail::PRING && BEAT_MAX_PENDING_CLASS == beat_pde_detail::PRING_MAX, "pending directions = ring size");
  if (pending == -1) {
    BEAT_REQUIRE(pde != nullptr && pde->ring <= BEAT_MAX_PENDING, "this kernel's pending path takes %d directions: finish the solve first", BEAT_MAX_PENDING);
    return step_behind_open_solve(ctx, pde, dev_ring0, field_stride, n, dev_states + (int64_t)v_index * ld, [&](const PendingV& pv) {
      return ode_step_dispatch(ctx, model_id, dev_states, n, ld, host_params, num_params, dev_params_per_node, params_ld, t, dt, v_index,
                               dev_v_copy, pv);
    });
  }
  BEAT_REQUIRE(pending >= 0 && pending <= BEAT_MAX_PENDING_CLASS, "pending count %d out of range", pending);
  BEAT_REQUIRE(pending == 0 || (pde != nullptr && dev_ring0 != nullptr && field_stride >= n), "bad pending update");
  BEAT_REQUIRE(pde == nullptr || !pde->open.on, "the operator has an open solve: finish it (beat_pde_solve_end) or pass pending = -1");
  if (int rc = flush_long_ring(pde, dev_states + (int64_t)v_index * ld, dev_ring0, field_stride, pending)) return rc;
  PendingV pend{dev_ring0, field_stride, pending ? pde->d_alphas : nullptr, pending, {}};
  if (pde != nullptr && pde->guess_pending) {  // this launch is the application the deferring solve left open
    pend.gt = pde->guess_final;
    pde->guess_pending = false;
  }
  return ode_step_dispatch(ctx, model_id, dev_states, n, ld, host_params, num_params, dev_params_per_node, params_ld,
                           t, dt, v_index, dev_v_copy, pend);
}


// Per-node parameters given as the uniform vector plus the rows that vary (see SparseRows): host_row_params[j] is the
// parameter index of row j of dev_rows ((num_rows, rows_ld), num_rows <= 4).  With pde / pending as in
// beat_ode_step_pending (pde == nullptr: a plain step).  The arithmetic is the per-node kernel's -- every parameter per
// lane, Derived per lane -- only the (P - num_rows) rows that do not vary are not read from memory.
extern "C" int beat_ode_step_rows(beat_ctx* ctx, int model_id, double* dev_states, int64_t n, int64_t ld,
                                  const double* host_params, int num_params, const int* host_row_params, int num_rows,
                                  const double* dev_rows, int64_t rows_ld, double t, double dt, int v_index, double* dev_v_copy,
                                  beat_pde* pde, const double* dev_ring0, int64_t field_stride, int pending) {
  BEAT_REQUIRE(host_params != nullptr && host_row_params != nullptr && dev_rows != nullptr, "null argument");
  BEAT_REQUIRE(num_rows >= 1 && num_rows <= BEAT_MAX_SPARSE_ROWS, "1..%d varying rows, got %d", BEAT_MAX_SPARSE_ROWS, num_rows);
  BEAT_REQUIRE(num_rows <= BEAT_MAX_SPARSE_ROWS_RT || beat_jit_enabled(), "%d varying rows need run-time compilation, which is not available "
               "here (beat_ode_jit_stats; at most %d rows otherwise): pass all rows (beat_ode_step)", num_rows, BEAT_MAX_SPARSE_ROWS_RT);
  SparseRows sp{{0}, num_rows};
  for (int j = 0; j < num_rows; ++j) {
    BEAT_REQUIRE(host_row_params[j] >= 0 && host_row_params[j] < num_params, "row %d names parameter %d of %d", j, host_row_params[j], num_params);
    sp.idx[j] = host_row_params[j];
  }
  if (pending == -1) {  // behind an open solve (see beat_ode_step_pending)
    BEAT_REQUIRE(pde != nullptr && pde->ring <= BEAT_MAX_PENDING, "this kernel's pending path takes %d directions: finish the solve first", BEAT_MAX_PENDING);
    return step_behind_open_solve(ctx, pde, dev_ring0, field_stride, n, dev_states + (int64_t)v_index * ld, [&](const PendingV& pv) {
      return ode_step_dispatch(ctx, model_id, dev_states, n, ld, host_params, num_params, dev_rows, rows_ld, t, dt, v_index, dev_v_copy, pv,
                               MarkedArgs{nullptr, nullptr, 0, nullptr, nullptr}, sp);
    });
  }
  BEAT_REQUIRE(pending >= 0 && pending <= BEAT_MAX_PENDING_CLASS, "pending count %d out of range", pending);
  BEAT_REQUIRE(pending == 0 || (pde != nullptr && dev_ring0 != nullptr && field_stride >= n), "bad pending update");
  BEAT_REQUIRE(pde == nullptr || !pde->open.on, "the operator has an open solve: finish it (beat_pde_solve_end) or pass pending = -1");
  if (int rc = flush_long_ring(pde, dev_states + (int64_t)v_index * ld, dev_ring0, field_stride, pending)) return rc;
  PendingV pend{dev_ring0, field_stride, pending ? pde->d_alphas : nullptr, pending, {}};
  if (pde != nullptr && pde->guess_pending) {
    pend.gt = pde->guess_final;
    pde->guess_pending = false;
  }
  return ode_step_dispatch(ctx, model_id, dev_states, n, ld, host_params, num_params, dev_rows, rows_ld, t, dt, v_index, dev_v_copy,
                           pend, MarkedArgs{nullptr, nullptr, 0, nullptr, nullptr}, sp);
}

template <class Model>
static int fill_table(const double* host_params, int num_params, int classes, std::vector<double>& out) {
  BEAT_REQUIRE(num_params == Model::NP, "model expects %d parameters, got %d", Model::NP, num_params);
  static_assert(sizeof(OdeTableEntry<Model>) % sizeof(double) == 0, "table entry is a whole number of doubles");
  const size_t stride = sizeof(OdeTableEntry<Model>) / sizeof(double);
  out.assign(stride * (size_t)classes, 0.0);
  for (int c = 0; c < classes; ++c) {
    OdeTableEntry<Model>* e = (OdeTableEntry<Model>*)(out.data() + stride * c);
    for (int k = 0; k < Model::NP; ++k) e->p[k] = host_params[(size_t)c * Model::NP + k];
    e->d = Model::derive(e->p);
  }
  return BEAT_OK;
}

template <class Model>
static int table_doubles() { return (int)(sizeof(OdeTableEntry<Model>) / sizeof(double)); }

extern "C" int beat_ode_class_table_doubles(int model_id, int* doubles_per_class) {
  BEAT_REQUIRE(doubles_per_class != nullptr, "null argument");
  switch (model_id) {
    case BEAT_MODEL_SIMPLE_ODE: *doubles_per_class = table_doubles<SimpleOde>(); break;
    case BEAT_MODEL_FHN_DEMO: *doubles_per_class = table_doubles<FhnDemo>(); break;
    case BEAT_MODEL_FHN_README: *doubles_per_class = table_doubles<FhnReadme>(); break;
    case BEAT_MODEL_TP06_GRL1: *doubles_per_class = table_doubles<Tp06Grl1>(); break;
    case BEAT_MODEL_TORORD_DYNCL_GRL1: *doubles_per_class = table_doubles<TorordDynClGrl1>(); break;
    case BEAT_MODEL_TORORD_LAND_GRL1: *doubles_per_class = table_doubles<TorordLandGrl1>(); break;
    default: {
      int np = 0;  // a model registered as source: its parameters + the one double of its (empty) Derived
      if (beat_custom_model_info(model_id, nullptr, &np, nullptr) == BEAT_OK) {
        *doubles_per_class = np + 1;
        break;
      }
      beat_set_error("unknown model id %d", model_id);
      return BEAT_EINVAL;
    }
  }
  return BEAT_OK;
}

extern "C" int beat_ode_class_table_fill(beat_ctx* ctx, int model_id, const double* host_params, int num_params, int classes,
                                         double* dev_table) {
  BEAT_REQUIRE(ctx != nullptr && host_params != nullptr && dev_table != nullptr, "null argument");
  BEAT_REQUIRE(classes >= 1 && classes <= BEAT_MAX_CLASSES, "1..%d parameter classes, got %d", BEAT_MAX_CLASSES, classes);
  std::vector<double> tab;
  int rc;
  switch (model_id) {
    case BEAT_MODEL_SIMPLE_ODE: rc = fill_table<SimpleOde>(host_params, num_params, classes, tab); break;
    case BEAT_MODEL_FHN_DEMO: rc = fill_table<FhnDemo>(host_params, num_params, classes, tab); break;
    case BEAT_MODEL_FHN_README: rc = fill_table<FhnReadme>(host_params, num_params, classes, tab); break;
    case BEAT_MODEL_TP06_GRL1: rc = fill_table<Tp06Grl1>(host_params, num_params, classes, tab); break;
    case BEAT_MODEL_TORORD_DYNCL_GRL1: rc = fill_table<TorordDynClGrl1>(host_params, num_params, classes, tab); break;
    case BEAT_MODEL_TORORD_LAND_GRL1: rc = fill_table<TorordLandGrl1>(host_params, num_params, classes, tab); break;
    default: {
      int np = 0;
      if (beat_custom_model_info(model_id, nullptr, &np, nullptr) != BEAT_OK) {
        beat_set_error("unknown model id %d", model_id);
        return BEAT_EINVAL;
      }
      BEAT_REQUIRE(num_params == np, "model expects %d parameters, got %d", np, num_params);
      tab.assign((size_t)(np + 1) * classes, 0.0);
      for (int c = 0; c < classes; ++c)
        for (int k = 0; k < np; ++k) tab[(size_t)c * (np + 1) + k] = host_params[(size_t)c * np + k];
      rc = BEAT_OK;
    }
  }
  if (rc) return rc;
  BEAT_HIP_CHECK(hipMemcpyAsync(dev_table, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  BEAT_HIP_CHECK(hipStreamSynchronize(ctx->stream));  // `tab` goes out of scope
  return BEAT_OK;
}

extern "C" int beat_ode_step_classes(beat_ctx* ctx, int model_id, double* dev_states, int64_t n, int64_t ld,
                                     const double* dev_table, int classes, const unsigned char* dev_markers, double t, double dt,
                                     int v_index, double* dev_v_copy, const int* dev_node_map, double* dev_v_field,
                                     beat_pde* pde, const double* dev_ring0, int64_t field_stride, int pending) {
  BEAT_REQUIRE(dev_table != nullptr && dev_markers != nullptr, "null argument");
  BEAT_REQUIRE((dev_node_map == nullptr) == (dev_v_field == nullptr), "dev_node_map and dev_v_field come together");
  BEAT_REQUIRE(dev_node_map == nullptr || dev_v_copy == nullptr, "with a node map the field IS the mirror of the potential");
  BEAT_REQUIRE(classes >= 1 && classes <= BEAT_MAX_CLASSES, "1..%d parameter classes, got %d", BEAT_MAX_CLASSES, classes);
  int stride = 0;
  if (int rc = beat_ode_class_table_doubles(model_id, &stride)) return rc;
  if (pending == -1) {  // behind an open solve (see beat_ode_step_pending): the class kernel takes the long ring as well
    double* v_row = dev_v_field != nullptr ? dev_v_field : dev_states + (int64_t)v_index * ld;
    return step_behind_open_solve(ctx, pde, dev_ring0, field_stride, dev_node_map != nullptr ? 0 : n, v_row, [&](const PendingV& pv) {
      return ode_step_dispatch(ctx, model_id, dev_states, n, ld, nullptr, 0, nullptr, 0, t, dt, v_index, dev_v_copy, pv,
                               MarkedArgs{dev_markers, dev_table, stride, dev_node_map, dev_v_field});
    });
  }
  BEAT_REQUIRE(pending >= 0 && pending <= BEAT_MAX_PENDING_CLASS, "pending count %d out of range", pending);
  BEAT_REQUIRE(pending == 0 || (pde != nullptr && dev_ring0 != nullptr && (field_stride >= n || dev_node_map != nullptr)),
               "bad pending update");
  BEAT_REQUIRE(pde == nullptr || !pde->open.on, "the operator has an open solve: finish it (beat_pde_solve_end) or pass pending = -1");
  PendingV pend{dev_ring0, field_stride, pending ? pde->d_alphas : nullptr, pending, {}};
  if (pde != nullptr && pde->guess_pending) {
    pend.gt = pde->guess_final;
    pde->guess_pending = false;
  }
  return ode_step_dispatch(ctx, model_id, dev_states, n, ld, nullptr, 0, nullptr, 0, t, dt, v_index, dev_v_copy, pend,
                           MarkedArgs{dev_markers, dev_table, stride, dev_node_map, dev_v_field});
}


// Many split steps of a BIG grid in one call (theta = 1): what a caller's loop of beat_ode_step_pending + beat_pde_solve_ex(defer_flush)
// does, without the caller between the steps.  The host is on the critical path of a step exactly once -- the convergence check of
// the solve, after which the next ionic kernel is launched -- and the device idles for as long as that takes; through the Python
// layers it takes 0.15-0.18 ms per step at 512^3 (1 % of the step) and the ionic kernel that follows an idle gap runs slower on top
// (DESIGN.md 7).  Here it is a wake-up and a launch.  Same kernels, same arguments, same values as the per-step calls.
// pending_in: search directions of an earlier deferred solve still to be applied to the potential row (0: none; the guess
// increment, if one is due, is known to the operator); host_pending[3]: what the LAST solve that ran left pending, as
// beat_pde_solve_ex reports it, and the number of steps done -- a solve that runs out of iterations ends the batch there.
// host_ode_ms (or NULL): per step, the duration of the ionic launch (HIP events).
extern "C" int beat_split_steps_big(beat_ctx* ctx, int model_id, double* dev_states, int64_t n, int64_t ld, const double* host_params,
                                    int num_params, int v_index, beat_pde* pde, double* dev_work, int n_steps, const double* host_t0,
                                    const double* host_dt, const double* const* host_dev_stim_w, const double* host_stim_amp, int n_stim,
                                    double rtol, double atol, int max_it, int pending_in, beat_ksp_info* host_info, int* host_pending,
                                    float* host_ode_ms) {
  BEAT_REQUIRE(ctx != nullptr && pde != nullptr && dev_states != nullptr && dev_work != nullptr && host_pending != nullptr &&
               ((host_t0 != nullptr && host_dt != nullptr) || n_steps == 0), "null argument");
  BEAT_REQUIRE(pde->ctx == ctx, "operator and states belong to different contexts");
  BEAT_REQUIRE(n_steps >= 0 && n_steps <= BEAT_MAX_BATCH, "at most %d steps per call, got %d", BEAT_MAX_BATCH, n_steps);
  BEAT_REQUIRE(pde->n == n, "the operator has %lld nodes, the state array %lld", (long long)pde->n, (long long)n);
  BEAT_REQUIRE(pde->g.z_lo_phys && pde->g.z_hi_phys, "a slab with live neighbours is stepped through beat_pde_solve_dist");
  BEAT_REQUIRE(n_stim == 0 || (host_dev_stim_w != nullptr && host_stim_amp != nullptr), "null stimulus arrays");
  BEAT_REQUIRE(pending_in >= 0 && pending_in <= pde->ring, "pending_in out of range");
  host_pending[0] = 0;
  host_pending[1] = pending_in;
  host_pending[2] = 0;
  if (n_steps == 0) return BEAT_OK;
  const int64_t fld = beat_pde_field_stride(pde);
  double* v_row = dev_states + (int64_t)v_index * ld;
  const double* ring0 = dev_work + pde->g.plane + 3 * fld;  // [r, q, z, ring...], each field behind its lower ghost plane
  // (timing events of the ionic launches: destroyed on every way out of this function)
  struct Events {
    std::vector<hipEvent_t> ev;
    ~Events() {
      for (hipEvent_t e : ev)
        if (e) (void)hipEventDestroy(e);
      (void)hipGetLastError();
    }
  } evs;
  std::vector<hipEvent_t>& ev = evs.ev;
  if (host_ode_ms != nullptr) {
    ev.resize((size_t)2 * n_steps, nullptr);
    for (hipEvent_t& e : ev) BEAT_HIP_CHECK(hipEventCreate(&e));
    for (int s = 0; s < n_steps; ++s) host_ode_ms[s] = -1.f;
  }
  int rc = BEAT_OK;
  int count = pending_in;
  int done = 0;
  for (int s = 0; s < n_steps && rc == BEAT_OK; ++s) {
    if (host_ode_ms) BEAT_HIP_CHECK(hipEventRecord(ev[(size_t)2 * s], ctx->stream));
    rc = beat_ode_step_pending(ctx, model_id, dev_states, n, ld, host_params, num_params, nullptr, 0, host_t0[s], host_dt[s], v_index, nullptr,
                               pde, ring0, fld, count);
    if (host_ode_ms) BEAT_HIP_CHECK(hipEventRecord(ev[(size_t)2 * s + 1], ctx->stream));
    if (rc) break;
    beat_ksp_info info{};
    rc = beat_pde_solve_ex(pde, v_row, host_dev_stim_w, host_stim_amp ? host_stim_amp + (size_t)s * n_stim : nullptr, n_stim, v_row, dev_work,
                           rtol, atol, max_it, 1, &info, host_pending);
    if (host_info) host_info[s] = info;
    count = host_pending[1];
    done = s + 1;
    // a solve that ran out of iterations ENDS the batch (round 5; it used to run on, up to BEAT_MAX_BATCH steps on the bad
    // iterate): the caller sees the state of the failing step, decides (ksp_error_if_not_converged raises there; the
    // reference's loop without it goes on) and calls again for the rest
  }
  host_pending[2] = done;
  if (host_ode_ms != nullptr) {
    (void)hipStreamSynchronize(ctx->stream);
    for (int s = 0; s < done; ++s) {
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, ev[(size_t)2 * s], ev[(size_t)2 * s + 1]) != hipSuccess) ms = -1.f;
      host_ode_ms[s] = ms;
    }
  }
  return rc;
}

// Many split steps of a SMALL grid in one call (theta = 1: ionic step of dt, then the diffusion step of dt in place on
// the potential row; src/beat/monodomain_solver.py:33-79 run n_steps times): per step one ionic launch, the
// one-workgroup solve (beat_pde_small.hip) and, optionally, a probe record -- nothing else, in particular no host
// round trip: the scalar results of every solve are read back once at the end.  A grid of a few thousand nodes is
// otherwise bound by exactly that: the Niederer slab at dx = 0.5 mm spends 0.10 of its 0.15 ms per step on the GPU.
extern "C" int beat_split_steps(beat_ctx* ctx, int model_id, double* dev_states, int64_t n, int64_t ld,
                                const double* host_params, int num_params, int v_index, beat_pde* pde, int n_steps,
                                const double* host_t0, const double* host_dt, const double* const* host_dev_stim_w, const double* host_stim_amp,
                                int n_stim, double rtol, double atol, int max_it, const int64_t* host_probe_idx,
                                const double* host_probe_w, int n_probe, double* dev_probe_out, beat_ksp_info* host_info) {
  BEAT_REQUIRE(ctx != nullptr && pde != nullptr && dev_states != nullptr && ((host_t0 != nullptr && host_dt != nullptr) || n_steps == 0), "null argument");
  BEAT_REQUIRE(pde->ctx == ctx, "operator and states belong to different contexts");
  BEAT_REQUIRE(n_steps >= 0 && n_steps <= BEAT_MAX_BATCH, "at most %d steps per call, got %d", BEAT_MAX_BATCH, n_steps);
  BEAT_REQUIRE(pde->n == n, "the operator has %lld nodes, the state array %lld", (long long)pde->n, (long long)n);
  BEAT_REQUIRE(beat_small_available(pde), "beat_split_steps is for grids the one-launch solve takes "
               "(beat_pde_small_grid_solve_active)");
  BEAT_REQUIRE(!pde->guess_pending, "a deferred update of the potential is pending: apply it first");
  BEAT_REQUIRE(n_stim == 0 || (host_dev_stim_w != nullptr && host_stim_amp != nullptr), "null stimulus arrays");
  BEAT_REQUIRE(n_probe == 0 || (host_probe_idx && host_probe_w && dev_probe_out), "null probe arrays");
  if (n_steps == 0) return BEAT_OK;
  if (pde->d_batch_st == nullptr) BEAT_HIP_CHECK(hipMalloc(&pde->d_batch_st, sizeof(double) * 16 * BEAT_MAX_BATCH));
  double* v_row = dev_states + (int64_t)v_index * ld;
  for (int s = 0; s < n_steps; ++s) {
    if (int rc = ode_step_dispatch(ctx, model_id, dev_states, n, ld, host_params, num_params, nullptr, 0, host_t0[s], host_dt[s],
                                   v_index, nullptr, PendingV{nullptr, 0, nullptr, 0, {}}))
      return rc;
    if (int rc = beat_small_launch(pde, v_row, host_dev_stim_w, host_stim_amp ? host_stim_amp + (size_t)s * n_stim : nullptr,
                                   n_stim, v_row, rtol, atol, max_it, pde->d_batch_st + 16 * s))
      return rc;
    if (pde->guess.d != nullptr) beat_guess_advance(pde);  // (the adaptive order keeps its choice through a batch, see below)
    if (n_probe > 0)
      if (int rc = beat_field_probe_record(ctx, v_row, host_probe_idx, host_probe_w, n_probe, dev_probe_out + (size_t)s * n_probe))
        return rc;
  }
  pde->auto_e_order = 0;
  std::vector<double> h((size_t)16 * n_steps);
  BEAT_HIP_CHECK(hipMemcpyAsync(h.data(), pde->d_batch_st, sizeof(double) * 16 * n_steps, hipMemcpyDeviceToHost, ctx->stream));
  BEAT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  int worst = BEAT_OK;
  for (int s = 0; s < n_steps; ++s) {
    const double* hs = h.data() + 16 * s;
    const int reason = (int)hs[beat_pde_detail::REASON];
    if (host_info) {
      host_info[s].iterations = (int)hs[beat_pde_detail::ITERS];
      host_info[s].converged_reason = reason;
      host_info[s].residual_norm = std::sqrt(hs[beat_pde_detail::RR]);
      host_info[s].rhs_norm = std::sqrt(hs[beat_pde_detail::BB]);
    }
    if (reason < 0 && worst == BEAT_OK) {
      beat_set_error("PCG did not converge in step %d of the batch (%d iterations, ||r|| = %.3e, ||b|| = %.3e)", s,
                     (int)hs[beat_pde_detail::ITERS], std::sqrt(hs[beat_pde_detail::RR]), std::sqrt(hs[beat_pde_detail::BB]));
      worst = BEAT_ENOTCONV;
    }
  }
  pde->last_iters = (int)h[(size_t)16 * (n_steps - 1) + beat_pde_detail::ITERS];
  if (pde->guess_order < 0 && n_steps >= 4) {
    // adaptive order, batch-wise: the whole batch ran with one order (its first solves on the guess the previous
    // batch left); score it by the mean iteration count of the later steps and let the policy move (beat_guess_policy:
    // here one "solve" is one batch)
    double sum = 0.0;
    for (int s = 2; s < n_steps; ++s) sum += h[(size_t)16 * s + beat_pde_detail::ITERS];
    const int k = pde->auto_next - 1;
    const double mean = sum / (n_steps - 2);
    pde->auto_score[k] = pde->auto_seen[k] ? 0.5 * pde->auto_score[k] + 0.5 * mean : mean;
    pde->auto_seen[k] = 1;
    pde->auto_since_probe += 5;  // (a batch stands for many solves: look at a neighbour every second or third batch)
    pde->auto_next = beat_guess_policy(pde);
  }
  return worst;
}
