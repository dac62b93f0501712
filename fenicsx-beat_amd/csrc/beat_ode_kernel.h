// The ionic step kernel (template) and its argument structures, in a header so that two translation units can instantiate it:
// beat_ode.hip (every instance the library ships) and the unit beat_ode_jit.hip writes and compiles at run time for ONE instance
// whose varying parameter indices are compile-time constants (sparse per-node parameter rows, see MixedParams below).
// Replaces  states[:] = fun(states=, t=, parameters=, dt=)  (src/beat/odesolver.py:67-79).
#pragma once
#include "beat_pde_internal.h"

#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdlib>
#include <type_traits>
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
  // The launch was enqueued BEHIND a solve the host has not looked at yet (beat_ode_step_pending with pending = -1, round 5): what
  // is pending is read from that solve's scalar state on the device -- dev_st[NUPD] updates executed, `ring_len` directions per ring
  // cycle: count = NUPD % ring, the guess terms accumulate when a full cycle was flushed inside the loop, nothing is due when the
  // guess alone satisfied the stopping test and there is none -- and the kernel does NOTHING if the solve has not latched
  // (dev_st[STOP] == 0: it needs more iterations than were enqueued; the host, which reads the same state right after, enqueues
  // them and launches the step again).  nullptr: count and gt are the host's (every other caller).
  const double* dev_st;
  int ring_len;
};

// what a launch applies: the host's count and guess terms, or (PendingV::dev_st) the ones the open solve's device state says
struct PendingNow {
  int count;
  beat_pde_detail::GuessTerms gt;
};
// How a launch enqueued behind an open solve reads that solve's update count (BEAT_PENDING_READ, measured at 512^3 on one box,
// profiles/r05_step_gap.md): 0 = per tile with a vector load through the generic pointer -- every pending load of the tile waits on
// its round trip, and its s_waitcnt vmcnt(0) on the previous tile's stores as well: +0.10 ms on the ionic kernel of step() against the
// library's loop, which knows the count on the host; 1 = once per launch, kept in an SGPR across the tile loop: the difference is
// gone, but BOTH paths lose 0.1 - 0.4 ms to the changed register allocation; 2 = per tile with a SCALAR load (the state was written
// by earlier kernels and is constant for this one; the constant cache is invalidated at kernel start): no vector-memory wait;
// 3 = thread 0 of a block reads latch and count once, ahead of the barrier that follows the table set-up, into LDS; every tile
// takes the count from there (a volatile LDS read: nothing lives across the tile loop, no global round trip per tile).  3 is the
// build: against 2, alternating on two boxes (13 runs each, profiles/r05_step_gap.md), the ionic kernel of step() AND of the library's
// loop sat at 9.68 - 9.72 ms in every run, where build 2 ranged from 9.63 to 10.03 (the two modes of this kernel that rounds 3 and 4
// chased): mean step 13.55 - 13.57 against 13.62 - 13.76 ms.  (Not immune: 9.65 - 10.07 on a third box.)
#ifndef BEAT_PENDING_READ
#define BEAT_PENDING_READ 3
#endif
__device__ __forceinline__ int beat_pending_read(const PendingV& p) {
#if BEAT_PENDING_READ == 1
  return p.dev_st != nullptr ? __builtin_amdgcn_readfirstlane((int)p.dev_st[beat_pde_detail::NUPD]) : -1;
#else
  return -1;
#endif
}
__device__ __forceinline__ PendingNow beat_pending_now(const PendingV& p, int nupd) {
  PendingNow o{p.count, p.gt};
#if BEAT_PENDING_READ == 0
  if (p.dev_st != nullptr) nupd = (int)p.dev_st[beat_pde_detail::NUPD];
#elif BEAT_PENDING_READ == 2
  if (p.dev_st != nullptr) {
    typedef const __attribute__((address_space(4))) double* ConstD;
    ConstD stc = (ConstD)(uintptr_t)p.dev_st;
    asm volatile("" : "+s"(stc));
    nupd = (int)stc[beat_pde_detail::NUPD];
  }
#endif
  if (nupd >= 0) {
    o.count = nupd % p.ring_len;
    o.gt.accumulate = nupd >= p.ring_len ? 1 : 0;
    const bool e_due = nupd == 0 && p.gt.use_e != 0;
    if (o.count == 0 && !e_due) o.gt.d = nullptr;  // nothing to apply, nothing to record (beat_guess_end)
  }
  return o;
}

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
  PendingV pend;
  MarkedArgs mk;  // (SparseRows follows)
};

// Per-node parameters of which only a few ROWS vary (a smooth gradient in one conductance: src/beat/odesolver.py:67-79 hands
// ``fun`` the whole (P, N) array, demos/pace_train.py:133-167 builds such arrays): the varying rows alone live on the
// device, the other parameters come from the uniform vector -- 8 B per varying row and node instead of 8 NP (TP06: 424).
constexpr int BEAT_MAX_SPARSE_ROWS = 16;     // on an instance compiled for the rows' indices (beat_ode_jit.h)
constexpr int BEAT_MAX_SPARSE_ROWS_RT = 4;   // on the shipped kernel, which finds a row's entry by comparison at run time
struct SparseRows {
  int idx[BEAT_MAX_SPARSE_ROWS];  // parameter index of row j of ppn
  int count;                      // 0: ppn holds all NP rows
};

template <class Model>
struct OdeTableEntry {
  double p[Model::NP];
  typename Model::Derived d;
};

// Varying rows whose parameter indices K... are known at COMPILE time (the instance is written and compiled at first use,
// beat_ode_jit.hip; up to BEAT_MAX_SPARSE_ROWS of them since round 5, four before): every use p[k] of the model's code folds to a
// row's value (k == Kj) or to the uniform vector, read with scalar loads -- everything that does not vary stays on the scalar
// unit, as in the uniform kernel.  IdxPack<>: no compile-time indices (the shipped instances).
template <int... K>
struct IdxPack {
  static constexpr int count = (int)sizeof...(K);
};
template <class CT>
struct MixedParams;
template <int... K>
struct MixedParams<IdxPack<K...>> {
  const double* u;
  double v[sizeof...(K) > 0 ? sizeof...(K) : 1];
  __device__ __forceinline__ double operator[](int k) const {
    double r = u[k];  // (dead when k names a varying row: the load goes with it)
    int j = 0;
    ((r = (k == K ? v[j] : r), ++j), ...);
    return r;
  }
};
// The derived constants (Model::Derived: doubles only): those a varying parameter enters -- bit j of DM0 (entries 0..63) / DM1
// (64..127), found on the host by perturbing the parameter -- from the per-lane evaluation, the others from the uniform set
// the host computed (scalar loads); the per-lane evaluation of an entry that is not taken is dead code and disappears.
template <class D, unsigned long long DM0, unsigned long long DM1>
__device__ __forceinline__ D mix_derived(const D& du, const D& dl) {
  D d;
  constexpr int ND = (int)(sizeof(D) / sizeof(double));
  static_assert(sizeof(D) % sizeof(double) == 0 && ND <= 128, "Derived: up to 128 doubles");
  const double* a = (const double*)&du;
  const double* b = (const double*)&dl;
  double* o = (double*)&d;
#pragma unroll
  for (int j = 0; j < ND; ++j) o[j] = (((j < 64 ? DM0 : DM1) >> (j & 63)) & 1ull) ? b[j] : a[j];
  return d;
}

// An IO type with somewhere to park values (ionic_models.h: beat_stash / beat_unstash): slot j of this lane at lds[j * BEAT_BLOCK]
// (consecutive lanes 8 bytes apart: no bank conflicts), a region no other lane touches -- no barrier
template <class Base>
struct StashIO : Base {
  double* lds;
  __device__ __forceinline__ void stash(int slot, double v) const { lds[slot * BEAT_BLOCK] = v; }
  __device__ __forceinline__ double unstash(int slot) const { return lds[slot * BEAT_BLOCK]; }
};
// the FastMath flavour a model's step takes (Model::FM; FastMath = v_ldexp_f64 scaling unless the model says otherwise)
template <class Model, class = void>
struct beat_fm_type { using type = FastMath; };
template <class Model>
struct beat_fm_type<Model, std::void_t<typename Model::FM>> { using type = typename Model::FM; };
template <class Model, class = void>
struct beat_fm_pin_wanted : std::false_type {};
template <class Model>
struct beat_fm_pin_wanted<Model, std::void_t<decltype(Model::FM_PIN)>> : std::integral_constant<bool, Model::FM_PIN> {};
template <class Model, class = void>
struct beat_stash_slots : std::integral_constant<int, 0> {};
template <class Model>
struct beat_stash_slots<Model, std::void_t<decltype(Model::STASH_SLOTS)>> : std::integral_constant<int, Model::STASH_SLOTS> {};

// 1: the kernel's scalar arguments are re-read through the opaque kernel-argument pointer in every tile (see the tile loop)
#ifndef BEAT_KARGS_PER_TILE
#define BEAT_KARGS_PER_TILE 1
#endif
// 1: the thread's index within the block is recomputed per tile (see the tile loop)
#ifndef BEAT_TID_PER_TILE
#define BEAT_TID_PER_TILE 1
#endif
// cache policy of the class kernel's state rows (BEAT_ODE_CLS_NT: 1 = as the uniform kernels' rows, BEAT_ODE_NT; 0 = plain)
#ifndef BEAT_ODE_CLS_NT
#define BEAT_ODE_CLS_NT 1  // (round 6: ToR-ORd classes 2.547 -> 2.522 ms at 256^3 in one process, shell 401^3 9.87 - 10.05 -> 9.81 - 9.96: profiles/r06_inproc_cls_nt.txt)
#endif
__device__ __forceinline__ double beat_cls_load(const double* p) {
#if BEAT_ODE_CLS_NT
  return beat_row_load(p);
#else
  return *p;
#endif
}
__device__ __forceinline__ void beat_cls_store(double* p, double v) {
#if BEAT_ODE_CLS_NT
  beat_row_store(p, v);
#else
  *p = v;
#endif
}
// a double nobody has computed: the register's content (see the pending values of the tile loop)
__device__ __forceinline__ double beat_any_value() {
  double x;
#ifdef __AMDGCN__
  asm volatile("" : "=v"(x));
#else
  x = 0.0;
#endif
  return x;
}
#if BEAT_PENDING_READ == 3
#define BEAT_PENDING_TILE_COUNT __builtin_amdgcn_readfirstlane(*(volatile int*)&s_pend[1])
#else
#define BEAT_PENDING_TILE_COUNT nupd_dev
#endif
template <class Model, bool PER_NODE, bool PEND, bool MARKED = false, bool SPARSE = false, class CT = IdxPack<>,
          unsigned long long DM0 = 0, unsigned long long DM1 = 0>
__global__ __launch_bounds__(BEAT_BLOCK, (PER_NODE && CT::count == 0) ? Model::WAVES_PER_NODE : Model::WAVES) void ode_step_kernel(
    double* __restrict__ states, int64_t n, int64_t ld, ParamPack<Model::NP> prm,
    typename Model::Derived drv, const double* __restrict__ ppn, int64_t pld, double t, double dt,
    int v_index, double* __restrict__ v_copy, PendingV pend, MarkedArgs mk_arg, SparseRows sp) {
  __shared__ double etab[BEAT_EXP_TAB];
  __shared__ LogEntry ltab[128];
  static_assert(BEAT_EXP_TAB == BEAT_BLOCK, "one table entry per thread");
  etab[threadIdx.x] = beat_exp_tab_entry<beat_fm_type<Model>::type::INT_SCALE>(kExp2Tab[threadIdx.x], (int)threadIdx.x);
  if (threadIdx.x < 128) ltab[threadIdx.x] = kLogTab[threadIdx.x];
  // values a step parks in LDS across the stretch that does not use them (StashIO): the uniform-parameter instances only
  constexpr int NSTASH = (!PER_NODE && !MARKED) ? beat_stash_slots<Model>::value : 0;
  __shared__ double stash_lds[NSTASH > 0 ? NSTASH * BEAT_BLOCK : 1];
  __shared__ int cls_jn[MARKED ? BEAT_BLOCK : 1];  // class kernel: the mapped node of each lane (see NodeIOWithV)
#if BEAT_PENDING_READ == 3
  __shared__ int s_pend[2];  // [0] the solve ahead has latched (or there is none), [1] its update count (-1: the host's count)
  if (PEND && threadIdx.x == 0) {
    s_pend[0] = pend.dev_st != nullptr ? (pend.dev_st[beat_pde_detail::STOP] != 0.0 ? 1 : 0) : 1;
    s_pend[1] = pend.dev_st != nullptr ? (int)pend.dev_st[beat_pde_detail::NUPD] : -1;
  }
  __syncthreads();
  typename beat_fm_type<Model>::type fm{etab, ltab};
  if constexpr (beat_fm_pin_wanted<Model>::value) beat_fm_pin(fm);  // (two VGPRs for the whole step: a model's choice)
  if (PEND && s_pend[0] == 0) return;  // the solve ahead has not latched (see PendingV)
#else
  __syncthreads();
  typename beat_fm_type<Model>::type fm{etab, ltab};
  if constexpr (beat_fm_pin_wanted<Model>::value) beat_fm_pin(fm);  // (two VGPRs for the whole step: a model's choice)
  if (PEND && pend.dev_st != nullptr && pend.dev_st[beat_pde_detail::STOP] == 0.0) return;  // the solve ahead has not latched (see PendingV)
#endif
  const int nupd_dev = PEND ? beat_pending_read(pend) : -1;
  (void)nupd_dev;
  // (Round 3, measured and removed: starting the three blocks that share a CU a third of a tile apart -- s_sleep by
  // (blockIdx.x / 256) % 3 -- to de-phase their load bursts: 9.83 against 9.78 ms at 512^3, A B A B A B on one box.  The
  // 24 576 blocks of a launch replace each other on the CUs 32 times over; whatever phase they start in is gone after
  // the first round.)
  // A block takes the tiles blockIdx.x, blockIdx.x + gridDim.x, ... -- since round 6 the launch has ONE BLOCK PER TILE for models of 12
  // states and more (beat_ode.hip: ode_grid; at four / three waves per SIMD the other blocks of the CU cover a block's table
  // set-up, and blocks dispatched in order keep the launch in one moving window of the rows); rounds 2 - 5 launched 24 576 blocks
  // of ~21 tiles each (at three waves per SIMD: 10.5-10.6 ms against 10.9-11.3 for one block per tile), which small models keep.
  // (the wave's number in the block lives in an SGPR; the lane number is taken from mbcnt per tile, opaque: whatever is derived from
  // threadIdx.x -- its 64-bit extension, its byte offset -- would otherwise be kept in VGPRs across the tile loop, three of the 128
  // that four waves per SIMD have: see BEAT_TID_PER_TILE below)
  const int wave_in_block = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  for (int64_t tile = blockIdx.x;; tile += gridDim.x) {
  // the kernel-argument segment through a pointer the optimiser cannot see through (see below, at the uniform parameters);
  // the pending-update arguments are read through it too: kept in SGPRs across the tile loop they were spilled as well
  typedef const __attribute__((address_space(4))) char* KArgPtr;
  KArgPtr ka = (KArgPtr)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(ka));
  const PendingV& pendl = *(const PendingV*)(ka + offsetof(OdeStepKernArgHead<Model>, pend));
#if BEAT_KARGS_PER_TILE
  // (round 6) the scalar arguments as well, under their own names: as kernel parameters they are loaded once and stay in SGPRs
  // across the tile loop -- with the class arguments the last spilled SGPRs of the class kernels, i.e. a VGPR of lanes to hold them
  const OdeStepKernArgHead<Model>& hd = *(const OdeStepKernArgHead<Model>*)ka;
  double* __restrict__ const states = hd.states;
  const int64_t n = hd.n, ld = hd.ld;
  const double* __restrict__ const ppn = hd.ppn;
  const int64_t pld = hd.pld;
  const double t = hd.t, dt = hd.dt;
  const int v_index = hd.v_index;
  double* __restrict__ const v_copy = hd.v_copy;
  (void)ppn; (void)pld; (void)v_index; (void)v_copy;
#endif
  if (tile * BEAT_BLOCK >= n) break;
#if BEAT_TID_PER_TILE
  int lane_now = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  asm volatile("" : "+v"(lane_now));
  const int tid = (wave_in_block << 6) + lane_now;
#else
  const int tid = (int)threadIdx.x;
#endif
  const int64_t tile0 = tile * BEAT_BLOCK;  // the tile's first node: uniform
  unsigned lane_off = (unsigned)tid * 8u;  // the lane's node within the tile, in BYTES (see NodeIO; not const: beat_at)
  const int64_t i = tile0 + tid;
  if (i >= n) break;
  // The row stride, opaque per tile: the base address of each of the NS state rows (states + k ld) is uniform and
  // loop-invariant, so the compiler forms all of them ahead of the tile loop, runs out of SGPRs and parks them in VGPR
  // lanes -- one v_readlane per half address per tile on the VALU this kernel is bound by (144 of 4756 VALU
  // instructions per ToR-ORd node, 58 of 1928 per TP06 node).  Recomputed where used they cost SALU cycles only.
  int64_t ldl = ld;
  asm volatile("" : "+s"(ldl));
  if (MARKED) {
    // (the class arguments through the opaque kernel-argument pointer too, read where they are used: as kernel parameters they
    // stayed in SGPRs across the tile loop and were spilled to VGPR lanes -- one more VGPR, the one the class kernel was over)
    const MarkedArgs& mk = *(const MarkedArgs*)(ka + offsetof(OdeStepKernArgHead<Model>, mk));
    const int m_lane = mk.markers[i];
    // where the node's potential lives: row V_INDEX of the state array, or -- when the array holds only the nodes that
    // carry a cell model (a voxelised wall inside its box) -- the PDE's field at node vmap[i]: the kernel then reads the
    // potential there (pending update included) and writes the new one to both, which is the scatter and the gather of
    // the potential the per-marker route spends two launches per marker on
    const int64_t jn = mk.vmap != nullptr ? (int64_t)mk.vmap[i] : i;
    double* const vptr = mk.vmap != nullptr ? mk.vfield + jn : states + (int64_t)Model::V_INDEX * ld + i;
    // (round 6) Nothing of this block stays in VGPRs across Model::step but the class byte: the mapped node's index waits in LDS
    // for the one store that needs it (the potential's mirror), the unmapped mirror is addressed like the rows (uniform base +
    // lane offset), and nodes outside every class are dealt with BEFORE the passes -- the pointer and the updated potential used to
    // live through the whole step, 4 of the 7 VGPRs the class kernel lacked for a third wave per SIMD (175 against 168).
    if (mk.vmap != nullptr) cls_jn[tid] = (int)jn;
    struct NodeIOWithV {
      double* __restrict__ base;  // the tile's first node in row 0 (uniform)
      int64_t ld;
      mutable unsigned i;    // the lane's BYTE offset within the tile (see NodeIO)
      double* vbase;  // mirror of the potential: the PDE's field (mapped: entry *jn_slot) or v_copy at the tile (entry = lane), or nullptr
      const int* jn_slot;  // LDS: the block's array (uniform; the lane's entry is found from `i`); nullptr: not mapped
      double v;
      __device__ __forceinline__ double load(int k) const { return k == Model::V_INDEX ? v : beat_cls_load(beat_at(beat_row(base, k, ld), i)); }
      __device__ __forceinline__ void store(int k, double x) const {
        beat_cls_store(beat_at(beat_row(base, k, ld), i), x);
        if (k == Model::V_INDEX && vbase != nullptr) {
          if (jn_slot != nullptr)
            vbase[*(const int*)((const char*)jn_slot + (i >> 1))] = x;  // entry tid = byte offset / 8
          else
            *beat_at(vbase, i) = x;
        }
      }
    };
    // 254: a padding entry (the compact layout keeps each class in its own run of whole tiles, so that a wavefront
    // meets one class): nothing is read or written for it
    double v_now = m_lane != 254 ? *vptr : 0.0;
    if (PEND && m_lane != 254) {
      // the potential with the pending update applied (and the guess's bookkeeping done) once, ahead of the passes --
      // same expressions and order as NodeIOPending::load / x_flush_kernel: the pending values die here instead of
      // staying live through every pass (-30 VGPRs, no scratch)
      const PendingNow now = beat_pending_now(pendl, BEAT_PENDING_TILE_COUNT);
      double pp[BEAT_MAX_PENDING_CLASS], pa[BEAT_MAX_PENDING_CLASS];
#pragma unroll
      for (int j = 0; j < BEAT_MAX_PENDING_CLASS; ++j) {
        pp[j] = j < now.count ? __builtin_nontemporal_load(pendl.ring + (int64_t)j * pendl.fld + jn) : 0.0;
        pa[j] = j < now.count ? pendl.alphas[j] : 0.0;
      }
      if (now.gt.d != nullptr) {
        const beat_pde_detail::GuessTerms& gt = now.gt;
        const double ge = beat_pde_detail::beat_guess_needs_e(gt) ? __builtin_nontemporal_load(gt.e + jn) : 0.0;
        const double gd = beat_pde_detail::beat_guess_needs_d(gt) ? __builtin_nontemporal_load(gt.d + jn) : 0.0;
        const double gp0 = beat_pde_detail::beat_guess_needs_dp(gt, 0) ? __builtin_nontemporal_load(gt.dp[0] + jn) : 0.0;
        const double gp1 = beat_pde_detail::beat_guess_needs_dp(gt, 1) ? __builtin_nontemporal_load(gt.dp[1] + jn) : 0.0;
        double inc = gt.accumulate ? 0.0 : ge;
#pragma unroll
        for (int j = 0; j < BEAT_MAX_PENDING_CLASS; ++j)
          if (j < now.count) inc = fma(pa[j], pp[j], inc);
        beat_pde_detail::beat_guess_record(gt, gt.d + jn, gt.e + jn, inc, gd, gp0, gp1, ge);
        v_now += inc;
      } else {
#pragma unroll
        for (int j = 0; j < BEAT_MAX_PENDING_CLASS; ++j)
          if (j < now.count) v_now = fma(pa[j], pp[j], v_now);
      }
    }
    // a node outside every class still takes part in the diffusion: its potential gets the pending update
    if (PEND && m_lane == 255) {
      *vptr = v_now;
      if (mk.vmap != nullptr) states[(int64_t)Model::V_INDEX * ld + i] = v_now;
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
      NodeIOWithV iol{states + tile0, ldl, lane_off, mk.vmap != nullptr ? mk.vfield : (v_copy != nullptr ? v_copy + tile0 : nullptr),
                      mk.vmap != nullptr ? cls_jn : nullptr, v_now};
      asm volatile("" : "+v"(iol.i), "+v"(iol.v));
      if (m_lane == m) Model::step(iol, p_c, d_c, fm, t, dt);
      todo &= ~__ballot(m_lane == m);
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
    // (every field addressed as its tile's first node -- uniform, SGPRs -- plus the lane's 32-bit offset: see NodeIO)
    PendingNow now = beat_pending_now(pendl, BEAT_PENDING_TILE_COUNT);
    // (round 6: the pending values are NOT zero-filled where nothing is pending -- every use is behind the same uniform condition as the
    // load; `j < count ? load : 0.0` cost 16 register pairs of zeros per tile, 46 of the step's ~1700 VALU instructions.  The
    // alternative value is "whatever the register holds": an asm statement without instructions that DEFINES the value.)
    NodeIOPending<Model::V_INDEX> io{states + tile0, ldl, lane_off, v_copy != nullptr ? v_copy + tile0 : nullptr, now.count, {}, {}, 0.0, 0.0, 0.0, 0.0, {}};
#pragma unroll
    for (int j = 0; j < BEAT_MAX_PENDING; ++j) {
      io.pp[j] = j < now.count ? __builtin_nontemporal_load(beat_at(pendl.ring + ((int64_t)j * pendl.fld + tile0), lane_off)) : beat_any_value();
      io.pa[j] = j < now.count ? pendl.alphas[j] : 0.0;
    }
    io.ge = io.gd = io.gp0 = io.gp1 = 0.0;
    if (now.gt.d != nullptr) {
      now.gt.d += tile0;
      now.gt.e += tile0;
      if (now.gt.dp[0] != nullptr) now.gt.dp[0] += tile0;
      if (now.gt.dp[1] != nullptr) now.gt.dp[1] += tile0;
      io.gt = now.gt;
      io.ge = beat_pde_detail::beat_guess_needs_e(now.gt) ? __builtin_nontemporal_load(beat_at(now.gt.e, lane_off)) : beat_any_value();
      io.gd = beat_pde_detail::beat_guess_needs_d(now.gt) ? __builtin_nontemporal_load(beat_at(now.gt.d, lane_off)) : beat_any_value();
      io.gp0 = beat_pde_detail::beat_guess_needs_dp(now.gt, 0) ? __builtin_nontemporal_load(beat_at(now.gt.dp[0], lane_off)) : beat_any_value();
      io.gp1 = beat_pde_detail::beat_guess_needs_dp(now.gt, 1) ? __builtin_nontemporal_load(beat_at(now.gt.dp[1], lane_off)) : beat_any_value();
    }
    if constexpr (PER_NODE && SPARSE && CT::count > 0) {
      // (row j of ppn belongs to the j-th index of the pack: the launch checks sp.idx against the instance)
      MixedParams<CT> mp;
      mp.u = p_uni;
#pragma unroll
      for (int j = 0; j < CT::count; ++j) mp.v[j] = *beat_at(beat_row(ppn + tile0, j, pld), lane_off);
      if constexpr (DM0 == 0 && DM1 == 0) {  // the varying parameters enter no derived constant: the uniform set where it lies
        Model::step(io, mp, d_uni, fm, t, dt);
      } else {
        const typename Model::Derived dlane = Model::derive(mp);
        const typename Model::Derived dm = mix_derived<typename Model::Derived, DM0, DM1>(d_uni, dlane);
        Model::step(io, mp, dm, fm, t, dt);
      }
    } else if (PER_NODE) {
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
        for (int j = 0; j < BEAT_MAX_SPARSE_ROWS_RT; ++j) {
          if (j < sp.count) {
            const double vj = *beat_at(beat_row(ppn + tile0, j, pld), lane_off);
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
        for (int k = 0; k < Model::NP; ++k) pl[k] = *beat_at(beat_row(ppn + tile0, k, pld), lane_off);
      }
      const typename Model::Derived dl = Model::derive(pl);
      Model::step(io, pl, dl, fm, t, dt);
    } else if constexpr (NSTASH > 0) {
      const StashIO<NodeIOPending<Model::V_INDEX>> sio{io, stash_lds + tid};
      Model::step(sio, p_uni, d_uni, fm, t, dt);
    } else {
      Model::step(io, p_uni, d_uni, fm, t, dt);
    }
  } else {
    // (the mirror of row v_index -- any row here, unlike in the pending-update form -- is written after the step from
    // the row itself: a store-time test "k == v_index" for each of the NS rows is NS uniform conditions kept, and spilled)
    const NodeIO io{states + tile0, ldl, lane_off, nullptr, -1};
    if constexpr (PER_NODE && SPARSE && CT::count > 0) {
      // (row j of ppn belongs to the j-th index of the pack: the launch checks sp.idx against the instance)
      MixedParams<CT> mp;
      mp.u = p_uni;
#pragma unroll
      for (int j = 0; j < CT::count; ++j) mp.v[j] = *beat_at(beat_row(ppn + tile0, j, pld), lane_off);
      if constexpr (DM0 == 0 && DM1 == 0) {  // the varying parameters enter no derived constant: the uniform set where it lies
        Model::step(io, mp, d_uni, fm, t, dt);
      } else {
        const typename Model::Derived dlane = Model::derive(mp);
        const typename Model::Derived dm = mix_derived<typename Model::Derived, DM0, DM1>(d_uni, dlane);
        Model::step(io, mp, dm, fm, t, dt);
      }
    } else if (PER_NODE) {
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
        for (int j = 0; j < BEAT_MAX_SPARSE_ROWS_RT; ++j) {
          if (j < sp.count) {
            const double vj = *beat_at(beat_row(ppn + tile0, j, pld), lane_off);
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
        for (int k = 0; k < Model::NP; ++k) pl[k] = *beat_at(beat_row(ppn + tile0, k, pld), lane_off);
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
      if constexpr (NSTASH > 0) {
        const StashIO<NodeIO> sio{io, stash_lds + tid};
        Model::step(sio, p_uni, d_uni, fm, t, dt);
      } else {
        Model::step(io, p_uni, d_uni, fm, t, dt);
      }
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
  etab[threadIdx.x] = beat_exp_tab_entry<beat_fm_type<Model>::type::INT_SCALE>(kExp2Tab[threadIdx.x], (int)threadIdx.x);
  if (threadIdx.x < 128) ltab[threadIdx.x] = kLogTab[threadIdx.x];
  __syncthreads();
  typename beat_fm_type<Model>::type fm{etab, ltab};
  if constexpr (beat_fm_pin_wanted<Model>::value) beat_fm_pin(fm);  // (two VGPRs for the whole step: a model's choice)
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
    dl = Model::derive((const double*)pl);
  }
  // Models keep the states in registers across steps (RegIO).  REGISTER_LOOP = false routes a model's step through
  // global memory instead: needed by round 1's generated ToR-ORd step, whose heavy spilling produced wrong values
  // through RegIO with ROCm 7.2; no model in the library uses it any more (the hand-organised ToR-ORd kernel has no
  // spills), tests/test_golden_gpu.py::test_run_kernel_equals_repeated_steps guards every model.
  const RegIO rio{y};
  const NodeIO gio{states + (int64_t)blockIdx.x * BEAT_BLOCK, ld, threadIdx.x * 8u, nullptr, -1};
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
          Model::step(rio, (const double*)pl, dl, fm, t, dt);
        else
          Model::step(rio, prm.p, dl, fm, t, dt);
      } else {
        if (PER_NODE)
          Model::step(gio, (const double*)pl, dl, fm, t, dt);
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

