// Ionic (reaction) step kernels: one node per thread, state-major SoA rows, fp64.
// Replaces  states[:] = fun(states=, t=, parameters=, dt=)  (src/beat/odesolver.py:67-79).
//
// HBM traffic per node-update: 16*NS bytes (each state row read once, written once) plus 8 bytes
// when the transmembrane potential is mirrored into the PDE vector (dev_v_copy).
#include "beat_pde_internal.h"

#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdlib>
#include <vector>
#include "ionic_models.h"

#ifndef BEAT_ODE_PROBE
#define BEAT_ODE_PROBE 0  // 1 / 2: probe builds of the plain step kernel (memory only / arithmetic only), tools/ode_probe.sh
#endif
#include "torord_dyncl.h"

template <int NP>
struct ParamPack {
  double p[NP];
};

// Search directions of the last diffusion solve whose contribution alpha_j p_j has not been added to the
// potential row yet (deferred-x PCG, beat_pde_solve_ex with defer_flush): the ionic kernel reads the row anyway,
// has HBM bandwidth to spare (it is fp64-issue bound) and adds them on the fly, which saves the separate
// x += sum alpha_j p_j pass (8 (k+2) B/node).
struct PendingV {
  const double* ring;    // p_0 (device), p_j = ring + j * fld
  int64_t fld;
  const double* alphas;  // device, step lengths alpha_j
  int count;             // 0: no search direction pending
  // the solve started from an extrapolated guess (beat_pde_set_guess_order): v += inc, inc = e + sum alpha_j p_j,
  // inc is recorded as the step's diffusion increment and the next guess prepared (gt.d == nullptr: no guess)
  beat_pde_detail::GuessTerms gt;
};

// Layout of ode_step_kernel's kernel-argument segment up to the uniform parameters (all members 8-byte aligned): the
// tile loop re-reads them through an opaque copy of the segment pointer (see the kernel).
template <class Model>
struct OdeStepKernArgHead {
  double* states;
  int64_t n, ld;
  ParamPack<Model::NP> prm;
  typename Model::Derived drv;
  const double* ppn;
  int64_t pld;
  double t, dt;
  int v_index;
  double* v_copy;
  PendingV pend;  // (MarkedArgs follows)
};

// Cell types / parameter classes in ONE launch (MARKED): a byte per node selects one of up to BEAT_MAX_CLASSES
// parameter sets (uniform parameters + their Derived constants, a table in device memory laid out as TableEntry);
// 255 = the node belongs to no class and is not advanced.  A wavefront whose nodes all carry the same marker -- the rule
// when the classes are layers or regions -- reads its set with scalar loads exactly as the uniform kernel reads the
// kernel-argument segment; a wavefront that straddles a boundary runs the step once per class present, lanes masked.
// Replaces one launch per marker + scatter / gather of the potential (src/beat/odesolver.py:306-310 loops the markers).
struct MarkedArgs {
  const unsigned char* markers;  // (n) or nullptr
  const double* table;           // classes x (NP + sizeof(Derived) / 8) doubles
  int stride;                    // doubles per table entry
  const int* vmap;               // (n) node of the PDE grid each entry of the state array belongs to, or nullptr (identity)
  double* vfield;                // the PDE's field the potential is read from / mirrored to when vmap is given
};

// Per-node parameters of which only a few ROWS vary (a smooth gradient in one conductance: src/beat/odesolver.py:67-79 hands
// ``fun`` the whole (P, N) array, demos/pace_train.py:133-167 builds such arrays): the varying rows alone live on the
// device, the other parameters come from the uniform vector -- 8 B per varying row and node instead of 8 NP (TP06: 424).
constexpr int BEAT_MAX_SPARSE_ROWS = 4;
struct SparseRows {
  int idx[BEAT_MAX_SPARSE_ROWS];  // parameter index of row j of ppn
  int count;                      // 0: ppn holds all NP rows
};

template <class Model>
struct OdeTableEntry {
  double p[Model::NP];
  typename Model::Derived d;
};

template <class Model, bool PER_NODE, bool PEND, bool MARKED = false, bool SPARSE = false>
__global__ __launch_bounds__(BEAT_BLOCK, PER_NODE ? Model::WAVES_PER_NODE : Model::WAVES) void ode_step_kernel(
    double* __restrict__ states, int64_t n, int64_t ld, ParamPack<Model::NP> prm,
    typename Model::Derived drv, const double* __restrict__ ppn, int64_t pld, double t, double dt,
    int v_index, double* __restrict__ v_copy, PendingV pend, MarkedArgs mk, SparseRows sp) {
  __shared__ double etab[BEAT_EXP_TAB];
  __shared__ LogEntry ltab[128];
  static_assert(BEAT_EXP_TAB == BEAT_BLOCK, "one table entry per thread");
  etab[threadIdx.x] = kExp2Tab[threadIdx.x];
  if (threadIdx.x < 128) ltab[threadIdx.x] = kLogTab[threadIdx.x];
  __syncthreads();
  const FastMath fm{etab, ltab};
  // (Round 3, measured and removed: starting the three blocks that share a CU a third of a tile apart -- s_sleep by
  // (blockIdx.x / 256) % 3 -- to de-phase their load bursts: 9.83 against 9.78 ms at 512^3, A B A B A B on one box.  The
  // 24 576 blocks of a launch replace each other on the CUs 32 times over; whatever phase they start in is gone after
  // the first round.)
  // a block walks over several tiles of 256 nodes (stride gridDim.x) and pays its launch and the table set-up once:
  // at 512^3, 24 576 blocks of ~21 tiles each measured 10.5-10.6 ms against 10.9-11.3 for one block per tile on the
  // same box (768 blocks, i.e. exactly the resident number: 11.5; 3 072: 10.7; 196 608: 10.9)
  for (int64_t tile = blockIdx.x; tile * BEAT_BLOCK < n; tile += gridDim.x) {
  const int64_t i = tile * BEAT_BLOCK + threadIdx.x;
  if (i >= n) break;
  // The row stride, opaque per tile: the base address of each of the NS state rows (states + k ld) is uniform and
  // loop-invariant, so the compiler forms all of them ahead of the tile loop, runs out of SGPRs and parks them in VGPR
  // lanes -- one v_readlane per half address per tile on the VALU this kernel is bound by (144 of 4756 VALU
  // instructions per ToR-ORd node, 58 of 1928 per TP06 node).  Recomputed where used they cost SALU cycles only.
  int64_t ldl = ld;
  asm volatile("" : "+s"(ldl));
  // the kernel-argument segment through a pointer the optimiser cannot see through (see below, at the uniform parameters);
  // the pending-update arguments are read through it too: kept in SGPRs across the tile loop they were spilled as well
  typedef const __attribute__((address_space(4))) char* KArgPtr;
  KArgPtr ka = (KArgPtr)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(ka));
  const PendingV& pendl = *(const PendingV*)(ka + offsetof(OdeStepKernArgHead<Model>, pend));
  if (MARKED) {
    const int m_lane = mk.markers[i];
    // where the node's potential lives: row V_INDEX of the state array, or -- when the array holds only the nodes that
    // carry a cell model (a voxelised wall inside its box) -- the PDE's field at node vmap[i]: the kernel then reads the
    // potential there (pending update included) and writes the new one to both, which is the scatter and the gather of
    // the potential the per-marker route spends two launches per marker on
    const int64_t jn = mk.vmap != nullptr ? (int64_t)mk.vmap[i] : i;
    double* const vptr = mk.vmap != nullptr ? mk.vfield + jn : states + (int64_t)Model::V_INDEX * ld + i;
    struct NodeIOWithV {
      double* __restrict__ base;
      int64_t ld, i;
      double* vout;  // the field entry that mirrors the potential (or nullptr)
      double v;
      __device__ __forceinline__ double load(int k) const { return k == Model::V_INDEX ? v : base[(int64_t)k * ld + i]; }
      __device__ __forceinline__ void store(int k, double x) const {
        base[(int64_t)k * ld + i] = x;
        if (k == Model::V_INDEX && vout != nullptr) *vout = x;
      }
    };
    // 254: a padding entry (the compact layout keeps each class in its own run of whole tiles, so that a wavefront
    // meets one class): nothing is read or written for it
    double v_now = m_lane != 254 ? *vptr : 0.0;
    if (PEND && m_lane != 254) {
      // the potential with the pending update applied (and the guess's bookkeeping done) once, ahead of the passes --
      // same expressions and order as NodeIOPending::load / x_flush_kernel: the pending values die here instead of
      // staying live through every pass (-30 VGPRs, no scratch)
      double pp[BEAT_MAX_PENDING], pa[BEAT_MAX_PENDING];
#pragma unroll
      for (int j = 0; j < BEAT_MAX_PENDING; ++j) {
        pp[j] = j < pendl.count ? __builtin_nontemporal_load(pendl.ring + (int64_t)j * pendl.fld + jn) : 0.0;
        pa[j] = j < pendl.count ? pendl.alphas[j] : 0.0;
      }
      if (pendl.gt.d != nullptr) {
        const beat_pde_detail::GuessTerms& gt = pendl.gt;
        const double ge = beat_pde_detail::beat_guess_needs_e(gt) ? __builtin_nontemporal_load(gt.e + jn) : 0.0;
        const double gd = beat_pde_detail::beat_guess_needs_d(gt) ? __builtin_nontemporal_load(gt.d + jn) : 0.0;
        const double gp0 = beat_pde_detail::beat_guess_needs_dp(gt, 0) ? __builtin_nontemporal_load(gt.dp[0] + jn) : 0.0;
        const double gp1 = beat_pde_detail::beat_guess_needs_dp(gt, 1) ? __builtin_nontemporal_load(gt.dp[1] + jn) : 0.0;
        double inc = gt.accumulate ? 0.0 : ge;
#pragma unroll
        for (int j = 0; j < BEAT_MAX_PENDING; ++j)
          if (j < pendl.count) inc = fma(pa[j], pp[j], inc);
        beat_pde_detail::beat_guess_record(gt, gt.d + jn, gt.e + jn, inc, gd, gp0, gp1, ge);
        v_now += inc;
      } else {
#pragma unroll
        for (int j = 0; j < BEAT_MAX_PENDING; ++j)
          if (j < pendl.count) v_now = fma(pa[j], pp[j], v_now);
      }
    }
    unsigned long long todo = __ballot(m_lane < 254);
    while (todo) {
      const int m = __builtin_amdgcn_readlane(m_lane, __ffsll((long long)todo) - 1);  // wave-uniform
      // the class's entry through a constant-address-space pointer the optimiser cannot see through: scalar loads where
      // the values are used, as for the kernel-argument segment of the uniform kernel (the table is not written here)
      typedef const __attribute__((address_space(4))) char* TabPtr;
      TabPtr tb = (TabPtr)(uintptr_t)(mk.table + (int64_t)m * mk.stride);
      asm volatile("" : "+s"(tb));
      const double* p_c = (const double*)(tb + offsetof(OdeTableEntry<Model>, p));
      const typename Model::Derived& d_c = *(const typename Model::Derived*)(tb + offsetof(OdeTableEntry<Model>, d));
      // (node index and potential are made opaque per pass: otherwise the address of every state row and everything
      // that depends on the potential alone is hoisted out of this loop and kept in registers, +38 VGPRs and scratch)
      asm volatile("" : "+s"(ldl));
      NodeIOWithV iol{states, ldl, i, mk.vmap != nullptr ? vptr : (v_copy != nullptr ? v_copy + i : nullptr), v_now};
      asm volatile("" : "+v"(iol.i), "+v"(iol.v));
      if (m_lane == m) Model::step(iol, p_c, d_c, fm, t, dt);
      todo &= ~__ballot(m_lane == m);
    }
    // a node outside every class still takes part in the diffusion: its potential gets the pending update
    if (PEND && m_lane == 255) {
      *vptr = v_now;
      if (mk.vmap != nullptr) states[(int64_t)Model::V_INDEX * ld + i] = v_now;
    }
    continue;
  }
  // The ~90 uniform doubles (parameters, per-launch derived constants) are scalar loads from the kernel-argument
  // segment.  Left to itself the compiler hoists all of them out of the tile loop, runs out of SGPRs and parks them in
  // VGPR lanes: 640 v_readlane / v_writelane per node on the VALU that is this kernel's bottleneck.  Reading them
  // through a pointer the optimiser cannot see through keeps the loads where they are used (SALU, scalar cache).
  const double* p_uni = (const double*)(ka + offsetof(OdeStepKernArgHead<Model>, prm));
  const typename Model::Derived& d_uni =
      *(const typename Model::Derived*)(ka + offsetof(OdeStepKernArgHead<Model>, drv));
  if (PEND) {
    // all loads issued together (they overlap with the state loads that follow)
    NodeIOPending<Model::V_INDEX> io{states, ldl, i, v_copy, pendl.count, {}, {}, 0.0, 0.0, 0.0, 0.0, {}};
#pragma unroll
    for (int j = 0; j < BEAT_MAX_PENDING; ++j) {
      io.pp[j] = j < pendl.count ? __builtin_nontemporal_load(pendl.ring + (int64_t)j * pendl.fld + i) : 0.0;
      io.pa[j] = j < pendl.count ? pendl.alphas[j] : 0.0;
    }
    if (pendl.gt.d != nullptr) {
      io.gt = pendl.gt;
      io.ge = beat_pde_detail::beat_guess_needs_e(pendl.gt) ? __builtin_nontemporal_load(pendl.gt.e + i) : 0.0;
      io.gd = beat_pde_detail::beat_guess_needs_d(pendl.gt) ? __builtin_nontemporal_load(pendl.gt.d + i) : 0.0;
      io.gp0 = beat_pde_detail::beat_guess_needs_dp(pendl.gt, 0) ? __builtin_nontemporal_load(pendl.gt.dp[0] + i) : 0.0;
      io.gp1 = beat_pde_detail::beat_guess_needs_dp(pendl.gt, 1) ? __builtin_nontemporal_load(pendl.gt.dp[1] + i) : 0.0;
    }
    if (PER_NODE) {
      double pl[Model::NP];
      if (SPARSE) {
        // the uniform vector, then the few rows that vary written over their entries (the index of a row is wave-uniform,
        // the entry it lands in is found by comparison: no dynamic indexing of the register array)
        // (read with VECTOR loads -- one address for the whole wave, a broadcast from one or two cache lines: as scalar
        // loads the NP uniform values were all held in SGPRs until the selects below had consumed them, 500 (TP06) to 1300
        // (Land) SGPRs spilled to VGPR lanes)
        const double* pv = p_uni;
        asm volatile("" : "+v"(pv));
#pragma unroll
        for (int k = 0; k < Model::NP; ++k) pl[k] = pv[k];
#pragma unroll
        for (int j = 0; j < BEAT_MAX_SPARSE_ROWS; ++j) {
          if (j < sp.count) {
            const double vj = ppn[(int64_t)j * pld + i];
            // (the row's index opaque per tile: the NP comparisons with it are loop invariants otherwise -- 4 NP lane masks
            // hoisted out of the tile loop, held in SGPR pairs and spilled)
            int ij = sp.idx[j];
            asm volatile("" : "+s"(ij));
#pragma unroll
            for (int k = 0; k < Model::NP; ++k) pl[k] = k == ij ? vj : pl[k];
          }
        }
      } else {
#pragma unroll
        for (int k = 0; k < Model::NP; ++k) pl[k] = ppn[(int64_t)k * pld + i];
      }
      const typename Model::Derived dl = Model::derive(pl);
      Model::step(io, pl, dl, fm, t, dt);
    } else {
      Model::step(io, p_uni, d_uni, fm, t, dt);
    }
  } else {
    // (the mirror of row v_index -- any row here, unlike in the pending-update form -- is written after the step from
    // the row itself: a store-time test "k == v_index" for each of the NS rows is NS uniform conditions kept, and spilled)
    const NodeIO io{states, ldl, i, nullptr, -1};
    if (PER_NODE) {
      double pl[Model::NP];
      if (SPARSE) {
        // the uniform vector, then the few rows that vary written over their entries (the index of a row is wave-uniform,
        // the entry it lands in is found by comparison: no dynamic indexing of the register array)
        // (read with VECTOR loads -- one address for the whole wave, a broadcast from one or two cache lines: as scalar
        // loads the NP uniform values were all held in SGPRs until the selects below had consumed them, 500 (TP06) to 1300
        // (Land) SGPRs spilled to VGPR lanes)
        const double* pv = p_uni;
        asm volatile("" : "+v"(pv));
#pragma unroll
        for (int k = 0; k < Model::NP; ++k) pl[k] = pv[k];
#pragma unroll
        for (int j = 0; j < BEAT_MAX_SPARSE_ROWS; ++j) {
          if (j < sp.count) {
            const double vj = ppn[(int64_t)j * pld + i];
            // (the row's index opaque per tile: the NP comparisons with it are loop invariants otherwise -- 4 NP lane masks
            // hoisted out of the tile loop, held in SGPR pairs and spilled)
            int ij = sp.idx[j];
            asm volatile("" : "+s"(ij));
#pragma unroll
            for (int k = 0; k < Model::NP; ++k) pl[k] = k == ij ? vj : pl[k];
          }
        }
      } else {
#pragma unroll
        for (int k = 0; k < Model::NP; ++k) pl[k] = ppn[(int64_t)k * pld + i];
      }
      const typename Model::Derived dl = Model::derive(pl);
      Model::step(io, pl, dl, fm, t, dt);
    } else {
#if BEAT_ODE_PROBE == 1
      // probe build (never shipped: -DBEAT_ODE_PROBE=1): the kernel's memory traffic alone -- every state read and
      // written back, same grid and tile loop
      double tmp[Model::NS];
#pragma unroll
      for (int k = 0; k < Model::NS; ++k) tmp[k] = io.load(k);
#pragma unroll
      for (int k = 0; k < Model::NS; ++k) io.store(k, tmp[k] * 1.0000000001);
#elif BEAT_ODE_PROBE == 3
      // probe build (-DBEAT_ODE_PROBE=3): the traffic of probe 1 with the array addressed tile-major -- the NS rows of a
      // tile's 256 nodes next to each other (NS * 2 KB contiguous per tile) instead of NS streams ld apart
      double tmp[Model::NS];
      double* tb = states + tile * (int64_t)(Model::NS * BEAT_BLOCK) + threadIdx.x;
#pragma unroll
      for (int k = 0; k < Model::NS; ++k) tmp[k] = tb[k * BEAT_BLOCK];
#pragma unroll
      for (int k = 0; k < Model::NS; ++k) tb[k * BEAT_BLOCK] = tmp[k] * 1.0000000001;
#elif BEAT_ODE_PROBE == 2
      // probe build (-DBEAT_ODE_PROBE=2): the kernel's arithmetic alone -- states of the block's first tile (cache hits),
      // stores behind a condition that never holds
      struct ProbeIO {
        double* __restrict__ base;
        int64_t ld, i, j;
        __device__ __forceinline__ double load(int k) const { return base[(int64_t)k * ld + j]; }
        __device__ __forceinline__ void store(int k, double v) const {
          if (v == 1.2345e300) base[(int64_t)k * ld + i] = v;
        }
      };
      const ProbeIO pio{states, ldl, i, (int64_t)threadIdx.x};
      Model::step(pio, p_uni, d_uni, fm, t, dt);
#else
      Model::step(io, p_uni, d_uni, fm, t, dt);
#endif
    }
    if (v_copy != nullptr) v_copy[i] = states[(int64_t)v_index * ldl + i];
  }
  }
}

// Many steps inside one launch (single-cell pre-pacing, free-running ODE solves): the node's states stay
// in registers; t restarts at 0 for every beat and advances as j*dt within it (numpy.arange semantics of
// src/beat/single_cell.py:42-65).  Optionally records `ntrack` states every `save_freq` steps.
struct TrackSpec {
  int idx[8];
  int n;
};

// (one wave per SIMD asked of the register allocator: the states of a node stay in registers through the time loop,
// and these launches are a few hundred to a few thousand cells -- occupancy buys nothing, scratch traffic costs)
template <class Model, bool PER_NODE>
__global__ __launch_bounds__(BEAT_BLOCK, 1) void ode_run_kernel(
    double* __restrict__ states, int64_t n, int64_t ld, ParamPack<Model::NP> prm, typename Model::Derived drv,
    const double* __restrict__ ppn, int64_t pld, double t0, double dt, int64_t nsteps, int nbeats, int save_freq,
    TrackSpec track, double* __restrict__ trace) {
  __shared__ double etab[BEAT_EXP_TAB];
  __shared__ LogEntry ltab[128];
  static_assert(BEAT_EXP_TAB == BEAT_BLOCK, "one table entry per thread");
  etab[threadIdx.x] = kExp2Tab[threadIdx.x];
  if (threadIdx.x < 128) ltab[threadIdx.x] = kLogTab[threadIdx.x];
  __syncthreads();
  const FastMath fm{etab, ltab};
  const int64_t i = (int64_t)blockIdx.x * BEAT_BLOCK + threadIdx.x;
  if (i >= n) return;
  double y[Model::NS];
#pragma unroll
  for (int k = 0; k < Model::NS; ++k) y[k] = states[(int64_t)k * ld + i];
  double pl[PER_NODE ? Model::NP : 1];
  typename Model::Derived dl = drv;
  if (PER_NODE) {
#pragma unroll
    for (int k = 0; k < Model::NP; ++k) pl[k] = ppn[(int64_t)k * pld + i];
    dl = Model::derive(pl);
  }
  // Models keep the states in registers across steps (RegIO).  REGISTER_LOOP = false routes a model's step through
  // global memory instead: needed by round 1's generated ToR-ORd step, whose heavy spilling produced wrong values
  // through RegIO with ROCm 7.2; no model in the library uses it any more (the hand-organised ToR-ORd kernel has no
  // spills), tests/test_golden_gpu.py::test_run_kernel_equals_repeated_steps guards every model.
  const RegIO rio{y};
  const NodeIO gio{states, ld, i, nullptr, -1};
  int64_t row = 0;
  for (int beat = 0; beat < nbeats; ++beat) {
    for (int64_t j = 0; j < nsteps; ++j) {
      if (track.n > 0 && j % save_freq == 0) {
        for (int a = 0; a < track.n; ++a) {
          double v = 0.0;
#pragma unroll
          for (int k = 0; k < Model::NS; ++k)
            if (k == track.idx[a]) v = y[k];
          trace[(row * track.n + a) * n + i] = v;
        }
        ++row;
      }
      const double t = t0 + (double)j * dt;
      if (Model::REGISTER_LOOP) {
        if (PER_NODE)
          Model::step(rio, pl, dl, fm, t, dt);
        else
          Model::step(rio, prm.p, dl, fm, t, dt);
      } else {
        if (PER_NODE)
          Model::step(gio, pl, dl, fm, t, dt);
        else
          Model::step(gio, prm.p, dl, fm, t, dt);
#pragma unroll
        for (int k = 0; k < Model::NS; ++k) y[k] = states[(int64_t)k * ld + i];  // for the tracking above
      }
    }
  }
  if (Model::REGISTER_LOOP) {
#pragma unroll
    for (int k = 0; k < Model::NS; ++k) states[(int64_t)k * ld + i] = y[k];
  }
}

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

template <class Model>
static int launch_ode(beat_ctx* ctx, double* states, int64_t n, int64_t ld, const double* host_params,
                      int num_params, const double* ppn, int64_t pld, double t, double dt,
                      int v_index, double* v_copy, const PendingV& pend, MarkedArgs mk = MarkedArgs{nullptr, nullptr, 0, nullptr, nullptr},
                      SparseRows sp = SparseRows{{0, 0, 0, 0}, 0}) {
  BEAT_REQUIRE(num_params == Model::NP || (host_params == nullptr && ppn == nullptr && Model::NP == 2) || mk.markers != nullptr,
               "model expects %d parameters, got %d", Model::NP, num_params);
  // (with num_params == NP and neither a vector nor rows nor classes every parameter would silently be 1.0 -- what the
  // two-parameter test ODE is called with on purpose and nothing else is)
  BEAT_REQUIRE(host_params != nullptr || ppn != nullptr || mk.markers != nullptr || Model::NP == 2,
               "no parameters given: a host vector, per-node rows or parameter classes");
  BEAT_REQUIRE(v_copy == nullptr || (v_index >= 0 && v_index < Model::NS), "v_index %d out of range", v_index);
  BEAT_REQUIRE(mk.markers == nullptr || v_copy == nullptr || v_index == Model::V_INDEX,
               "the class kernel mirrors the model's potential (row %d), not row %d", Model::V_INDEX, v_index);
  const bool have_pend = pend.count > 0 || pend.gt.d != nullptr;
  BEAT_REQUIRE(!have_pend || v_index == Model::V_INDEX,
               "a pending update needs v_index = %d (the model's membrane potential), got %d", Model::V_INDEX, v_index);
  ParamPack<Model::NP> prm;
  for (int k = 0; k < Model::NP; ++k) prm.p[k] = host_params ? host_params[k] : 1.0;
  typename Model::Derived drv = Model::derive(prm.p);
  unsigned grid = (unsigned)((n + BEAT_BLOCK - 1) / BEAT_BLOCK);
  static const int grid_cap = [] {  // blocks per launch (BEAT_ODE_GRID; 0: one block per tile)
    const char* e = std::getenv("BEAT_ODE_GRID");
    return e ? std::atoi(e) : 24576;
  }();
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
    default: beat_set_error("unknown model id %d", model_id); return BEAT_EINVAL;
  }
  if (num_states) *num_states = ns;
  if (num_params) *num_params = np;
  return BEAT_OK;
}

static int ode_step_dispatch(beat_ctx* ctx, int model_id, double* dev_states, int64_t n, int64_t ld,
                             const double* host_params, int num_params, const double* dev_params_per_node,
                             int64_t params_ld, double t, double dt, int v_index, double* dev_v_copy,
                             const PendingV& pend, MarkedArgs mk = MarkedArgs{nullptr, nullptr, 0, nullptr, nullptr},
                             SparseRows sp = SparseRows{{0, 0, 0, 0}, 0}) {
  BEAT_REQUIRE(ctx != nullptr, "null context");
  BEAT_REQUIRE(dev_states != nullptr, "null states");
  BEAT_REQUIRE(n >= 0 && ld >= n, "bad shape n=%lld ld=%lld", (long long)n, (long long)ld);
  BEAT_REQUIRE((n + BEAT_BLOCK - 1) / BEAT_BLOCK < (int64_t)0x7fffffff, "n too large");
  if (n == 0) return BEAT_OK;
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

extern "C" int beat_ode_step_pending(beat_ctx* ctx, int model_id, double* dev_states, int64_t n, int64_t ld,
                                     const double* host_params, int num_params,
                                     const double* dev_params_per_node, int64_t params_ld, double t, double dt,
                                     int v_index, double* dev_v_copy, beat_pde* pde, const double* dev_ring0,
                                     int64_t field_stride, int pending) {
  static_assert(BEAT_MAX_PENDING == beat_pde_detail::PRING, "pending directions = ring size");
  BEAT_REQUIRE(pending >= 0 && pending <= BEAT_MAX_PENDING, "pending count %d out of range", pending);
  BEAT_REQUIRE(pending == 0 || (pde != nullptr && dev_ring0 != nullptr && field_stride >= n), "bad pending update");
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
  BEAT_REQUIRE(pending >= 0 && pending <= BEAT_MAX_PENDING, "pending count %d out of range", pending);
  BEAT_REQUIRE(pending == 0 || (pde != nullptr && dev_ring0 != nullptr && field_stride >= n), "bad pending update");
  SparseRows sp{{0, 0, 0, 0}, num_rows};
  for (int j = 0; j < num_rows; ++j) {
    BEAT_REQUIRE(host_row_params[j] >= 0 && host_row_params[j] < num_params, "row %d names parameter %d of %d", j, host_row_params[j], num_params);
    sp.idx[j] = host_row_params[j];
  }
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
    default: beat_set_error("unknown model id %d", model_id); return BEAT_EINVAL;
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
    default: beat_set_error("unknown model id %d", model_id); return BEAT_EINVAL;
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
  BEAT_REQUIRE(pending >= 0 && pending <= BEAT_MAX_PENDING, "pending count %d out of range", pending);
  BEAT_REQUIRE(pending == 0 || (pde != nullptr && dev_ring0 != nullptr && (field_stride >= n || dev_node_map != nullptr)),
               "bad pending update");
  PendingV pend{dev_ring0, field_stride, pending ? pde->d_alphas : nullptr, pending, {}};
  if (pde != nullptr && pde->guess_pending) {
    pend.gt = pde->guess_final;
    pde->guess_pending = false;
  }
  int stride = 0;
  if (int rc = beat_ode_class_table_doubles(model_id, &stride)) return rc;
  return ode_step_dispatch(ctx, model_id, dev_states, n, ld, nullptr, 0, nullptr, 0, t, dt, v_index, dev_v_copy, pend,
                           MarkedArgs{dev_markers, dev_table, stride, dev_node_map, dev_v_field});
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
