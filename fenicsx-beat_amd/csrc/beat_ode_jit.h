// Sparse per-node parameter rows (up to 16 of them) with the varying indices as COMPILE-TIME constants: the instance of ode_step_kernel for one
// (model, index set) is written as a three-line translation unit, compiled by hipcc for gfx950 at first use (~1.5 s for TP06),
// kept as a code object in a cache directory and loaded with hipModuleLoadData.  Why: with a run-time index every parameter of a
// node has to live in a VGPR (the step's p[k] cannot tell at compile time which k varies): 238 VGPRs, 2 waves, 1.20 x the
// uniform kernel for ONE varying conductance.  With the index known, p[k] folds to "the row's value" for that k and to a scalar
// load for every other k, and the derived constants the parameter does not enter stay the host's: 135 VGPRs, 3 waves, the
// uniform kernel's speed.  Reference: (P, N) parameter arrays handed to ``fun``, src/beat/odesolver.py:67-79,
// demos/pace_train.py:133-167.
//
// Where no instance can be had -- no hipcc on the machine, no kernel sources beside the library, BEAT_JIT=0, a compile that
// fails -- beat_ode_jit_launch returns BEAT_JIT_UNAVAILABLE and the caller launches the run-time-index kernel: a HIP kernel
// either way, never a host path.
#pragma once
#include "beat_ode_kernel.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <string>
#include <vector>

constexpr int BEAT_JIT_UNAVAILABLE = 1000;
constexpr int BEAT_MODEL_CUSTOM_BASE = 100;  // model ids of cell models registered as source (beat_ode_model_register)
int beat_custom_model_info(int model_id, int* ns, int* np, int* v_index);
struct PendingV;
int beat_custom_step(beat_ctx* ctx, int model_id, unsigned grid, double* states, int64_t n, int64_t ld, const double* host_params,
                     int num_params, const double* ppn, int64_t pld, double t, double dt, int v_index, double* v_copy, const PendingV& pend,
                     const MarkedArgs& mk);
int beat_custom_run(beat_ctx* ctx, int model_id, double* states, int64_t n, int64_t ld, const double* host_params, int num_params,
                    const double* ppn, int64_t pld, double t0, double dt, int64_t nsteps, int nbeats, int save_freq, const int* track_idx,
                    int ntrack, double* trace);

// beat_ode_jit.hip
bool beat_jit_enabled();
// the kernel of `key` on ctx's device: from memory, from the cache directory, or compiled from `source` now; nullptr: unavailable
hipFunction_t beat_jit_get(beat_ctx* ctx, const std::string& key, const std::string& source);
// the same from memory only (every step but the first); *known = false when the key has not been asked for yet
hipFunction_t beat_jit_lookup(beat_ctx* ctx, const std::string& key, bool* known);
// an instance that loaded but failed its check against the run-time-index kernel: not used again in this process, reported once
void beat_jit_reject(beat_ctx* ctx, const std::string& key, const std::string& why);

template <class Model, class = void>
struct BeatJitAccessor : std::false_type {};
template <class Model>
struct BeatJitAccessor<Model, std::void_t<decltype(Model::ACCESSOR_PARAMS)>> : std::true_type {};

// the model's type as the generated source spells it
template <class Model>
struct BeatJitName {
  static const char* get() { return nullptr; }
};
template <>
struct BeatJitName<Tp06Grl1> {
  static const char* get() { return "Tp06Grl1"; }
};
template <>
struct BeatJitName<TorordDynClGrl1> {
  static const char* get() { return "TorordDynClGrl1"; }
};
template <>
struct BeatJitName<TorordLandGrl1> {
  static const char* get() { return "TorordLandGrl1"; }
};

// Which derived constants does parameter `idx` enter?  Found numerically: the host evaluates Model::derive with the parameter
// replaced by NaN -- which reaches every constant it enters through arithmetic -- and by a spread of other values -- scaled,
// shifted, negated, and the small integers a cell-type or switch parameter takes -- for the constants it enters through a
// comparison; every entry whose bits change is taken per lane.  (An entry that depends on the parameter ONLY through a
// comparison with a threshold none of these values crosses would be missed; the models here have no such entry -- their
// branches in derive() are on cell type and on flags in {0, 1, 2, 3}.  What would catch such a miss: every instance is checked
// once per process against the run-time-index kernel, beat_jit_self_check below.)
template <class Model>
void beat_jit_derived_mask(const double* p, const SparseRows& sp, unsigned long long dm[2]) {
  using D = typename Model::Derived;
  constexpr int ND = (int)(sizeof(D) / sizeof(double));
  static_assert(sizeof(D) % sizeof(double) == 0 && ND <= 128, "Derived: up to 128 doubles");
  dm[0] = dm[1] = 0;
  double q[Model::NP];
  for (int k = 0; k < Model::NP; ++k) q[k] = p[k];
  const D base = Model::derive((const double*)q);
  for (int j = 0; j < sp.count; ++j) {
    const int k = sp.idx[j];
    const double x = p[k];
    // (NaN reaches every constant the parameter enters ARITHMETICALLY, whatever its value; the finite values are for the
    // constants it enters through a comparison)
    const double variants[] = {std::numeric_limits<double>::quiet_NaN(), std::numeric_limits<double>::infinity(),
                               x * 1.5 + 0.25, x * 0.5 - 0.125, x + 1.0, x + 2.0, x - 1.0, -x - 0.5, 0.0, 1.0, 2.0, 3.0, x * 7.0, x * 0.01};
    for (double v : variants) {
      q[k] = v;
      const D d = Model::derive((const double*)q);
      for (int e = 0; e < ND; ++e)
        if (std::memcmp((const char*)&d + 8 * e, (const char*)&base + 8 * e, 8) != 0) dm[e >> 6] |= 1ull << (e & 63);
    }
    q[k] = x;
  }
}

// One step of the first nodes on two scratch copies: the instance `f` and the run-time-index kernel; BEAT_OK when they agree to
// 1e-9 of each value (+ 1e-12 of its row's scale: the two differ in where the derived constants are rounded, 1e-12 measured),
// BEAT_JIT_UNAVAILABLE (and the instance rejected) when they do not.  Synchronises; runs once per instance and process.
template <class Model>
int beat_jit_self_check(beat_ctx* ctx, hipFunction_t f, const std::string& key, const double* states, int64_t n, int64_t ld,
                        ParamPack<Model::NP> prm, typename Model::Derived drv, const double* ppn, int64_t pld, double t, double dt,
                        int v_index, SparseRows sp) {
  if (const char* e = std::getenv("BEAT_JIT_SELF_CHECK"))
    if (e[0] == '0') return BEAT_OK;
  const bool many = sp.count > BEAT_MAX_SPARSE_ROWS_RT;  // more rows than the run-time-index kernel takes: checked against the all-rows kernel
  int64_t nc = std::min<int64_t>(n, 1024);
  double* scratch = nullptr;
  const size_t bytes = sizeof(double) * 2 * Model::NS * (size_t)nc;
  BEAT_HIP_CHECK(hipMalloc(&scratch, bytes));
  struct Free {
    double* p;
    ~Free() { (void)hipFree(p); }
  } guard{scratch};
  double* sa = scratch;
  double* sb = scratch + (size_t)Model::NS * nc;
  for (int k = 0; k < Model::NS; ++k) {
    BEAT_HIP_CHECK(hipMemcpyAsync(sa + (size_t)k * nc, states + (int64_t)k * ld, sizeof(double) * nc, hipMemcpyDeviceToDevice, ctx->stream));
    BEAT_HIP_CHECK(hipMemcpyAsync(sb + (size_t)k * nc, states + (int64_t)k * ld, sizeof(double) * nc, hipMemcpyDeviceToDevice, ctx->stream));
  }
  PendingV none{nullptr, 0, nullptr, 0, {}};
  MarkedArgs mk{nullptr, nullptr, 0, nullptr, nullptr};
  double* vc = nullptr;
  int64_t ldc = nc;
  const unsigned grid = (unsigned)((nc + BEAT_BLOCK - 1) / BEAT_BLOCK);
  void* args[] = {&sa, &nc, &ldc, &prm, &drv, &ppn, &pld, &t, &dt, &v_index, &vc, &none, &mk, &sp};
  BEAT_HIP_CHECK(hipModuleLaunchKernel(f, grid, 1, 1, BEAT_BLOCK, 1, 1, 0, ctx->stream, args, nullptr));
  if (!many) {
    BEAT_KERNEL((ode_step_kernel<Model, true, false, false, true>), dim3(grid), dim3(BEAT_BLOCK), 0, ctx->stream, sb, nc, ldc, prm, drv, ppn, pld,
                t, dt, v_index, vc, none, mk, sp);
  } else {
    // all NP rows for the first nodes: the uniform vector, the varying rows copied over their entries
    std::vector<double> full((size_t)Model::NP * nc), rows((size_t)sp.count * nc);
    for (int j = 0; j < sp.count; ++j)
      BEAT_HIP_CHECK(hipMemcpyAsync(rows.data() + (size_t)j * nc, ppn + (int64_t)j * pld, sizeof(double) * nc, hipMemcpyDeviceToHost, ctx->stream));
    BEAT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    for (int k = 0; k < Model::NP; ++k)
      for (int64_t i = 0; i < nc; ++i) full[(size_t)k * nc + i] = prm.p[k];
    for (int j = 0; j < sp.count; ++j)
      for (int64_t i = 0; i < nc; ++i) full[(size_t)sp.idx[j] * nc + i] = rows[(size_t)j * nc + i];
    double* dfull = nullptr;
    BEAT_HIP_CHECK(hipMalloc(&dfull, sizeof(double) * full.size()));
    struct Free2 {
      double* p;
      ~Free2() { (void)hipFree(p); }
    } g2{dfull};
    BEAT_HIP_CHECK(hipMemcpyAsync(dfull, full.data(), sizeof(double) * full.size(), hipMemcpyHostToDevice, ctx->stream));
    const double* cfull = dfull;
    SparseRows none_sp{{0}, 0};
    BEAT_KERNEL((ode_step_kernel<Model, true, false, false, false>), dim3(grid), dim3(BEAT_BLOCK), 0, ctx->stream, sb, nc, ldc, prm, drv, cfull,
                ldc, t, dt, v_index, vc, none, mk, none_sp);
    BEAT_LAUNCH_CHECK();
    BEAT_HIP_CHECK(hipStreamSynchronize(ctx->stream));  // (dfull is freed when this block ends)
  }
  BEAT_LAUNCH_CHECK();
  std::vector<double> h((size_t)2 * Model::NS * nc);
  BEAT_HIP_CHECK(hipMemcpyAsync(h.data(), scratch, bytes, hipMemcpyDeviceToHost, ctx->stream));
  BEAT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  const double* a = h.data();
  const double* b = h.data() + (size_t)Model::NS * nc;
  for (int k = 0; k < Model::NS; ++k) {
    double scale = 0.0;
    for (int64_t i = 0; i < nc; ++i) {
      const double v = std::fabs(b[(size_t)k * nc + i]);
      if (v == v && v > scale && v < 1e300) scale = v;
    }
    for (int64_t i = 0; i < nc; ++i) {
      const double x = a[(size_t)k * nc + i], y = b[(size_t)k * nc + i];
      if (x != x && y != y) continue;  // both NaN (a caller's garbage in, the same garbage out)
      if (!(std::fabs(x - y) <= 1e-9 * std::fabs(y) + 1e-12 * scale)) {
        char msg[256];
        std::snprintf(msg, sizeof msg, "self-check against the run-time-index kernel failed (state %d, node %lld: %.17g against %.17g)", k,
                      (long long)i, x, y);
        beat_jit_reject(ctx, key, msg);
        return BEAT_JIT_UNAVAILABLE;
      }
    }
  }
  return BEAT_OK;
}

template <class Model>
int beat_ode_jit_launch(beat_ctx* ctx, dim3 grid, bool have_pend, double* states, int64_t n, int64_t ld, ParamPack<Model::NP> prm,
                        typename Model::Derived drv, const double* ppn, int64_t pld, double t, double dt, int v_index, double* v_copy,
                        PendingV pend, MarkedArgs mk, SparseRows sp) {
  if constexpr (!BeatJitAccessor<Model>::value) {
    return BEAT_JIT_UNAVAILABLE;
  } else {
    if (BeatJitName<Model>::get() == nullptr || sp.count < 1 || sp.count > BEAT_MAX_SPARSE_ROWS || !beat_jit_enabled())
      return BEAT_JIT_UNAVAILABLE;
    // (the mask costs 14 K + 1 evaluations of derive() on the host: kept for as long as the uniform vector and the indices stay
    // what they were -- every step of a run but the first)
    unsigned long long dm[2];
    {
      struct MaskCache {
        bool valid = false;
        double p[Model::NP];
        SparseRows sp;
        unsigned long long dm[2];
      };
      static thread_local MaskCache mc;
      if (mc.valid && std::memcmp(mc.p, prm.p, sizeof mc.p) == 0 && std::memcmp(&mc.sp, &sp, sizeof sp) == 0) {
        dm[0] = mc.dm[0];
        dm[1] = mc.dm[1];
      } else {
        beat_jit_derived_mask<Model>(prm.p, sp, dm);
        std::memcpy(mc.p, prm.p, sizeof mc.p);
        mc.sp = sp;
        mc.dm[0] = dm[0];
        mc.dm[1] = dm[1];
        mc.valid = true;
      }
    }
    std::string pack;  // "12, 27, 3"
    for (int j = 0; j < sp.count; ++j) pack += (j ? ", " : "") + std::to_string(sp.idx[j]);
    char key[512], inst[768];
    std::snprintf(key, sizeof key, "%s_p%d_i%s_m%llx_%llx", BeatJitName<Model>::get(), have_pend ? 1 : 0, pack.c_str(), dm[0], dm[1]);
    for (char* c = key; *c; ++c)
      if (*c == ',' || *c == ' ') *c = '_';
    bool known = false;
    hipFunction_t f = beat_jit_lookup(ctx, key, &known);
    if (known && f == nullptr) return BEAT_JIT_UNAVAILABLE;
    if (!known) {
      std::snprintf(inst, sizeof inst, "ode_step_kernel<%s, true, %s, false, true, IdxPack<%s>, 0x%llxull, 0x%llxull>",
                    BeatJitName<Model>::get(), have_pend ? "true" : "false", pack.c_str(), dm[0], dm[1]);
      std::string src = "// written by libbeat_hip (beat_ode_jit.h): one instance of the ionic step kernel, varying parameter indices compile-time\n"
                        "#include \"beat_ode_kernel.h\"\n"
                        "template __global__ void ";
      src += inst;
      src += "(\n    double*, int64_t, int64_t, ParamPack<";
      src += BeatJitName<Model>::get();
      src += "::NP>, typename ";
      src += BeatJitName<Model>::get();
      src += "::Derived, const double*, int64_t, double, double, int, double*, PendingV, MarkedArgs, SparseRows);\n";
      f = beat_jit_get(ctx, key, src);
      // First use of an instance in this process: it is run beside the run-time-index kernel (every parameter and every derived
      // constant per lane: nothing to get wrong) on a scratch copy of the first nodes, and must agree.  The derived-constant mask
      // is found by perturbation (above); an entry it misses would make the instance use the uniform vector's constant for every
      // node -- silently wrong physics.  A mismatch rejects the instance; the run-time-index kernel runs instead.
      if (f != nullptr) {
        const int rc = beat_jit_self_check<Model>(ctx, f, key, states, n, ld, prm, drv, ppn, pld, t, dt, v_index, sp);
        if (rc == BEAT_JIT_UNAVAILABLE) return BEAT_JIT_UNAVAILABLE;
        if (rc != BEAT_OK) return rc;
      }
    }
    if (f == nullptr) return BEAT_JIT_UNAVAILABLE;
    void* args[] = {&states, &n, &ld, &prm, &drv, &ppn, &pld, &t, &dt, &v_index, &v_copy, &pend, &mk, &sp};
    BEAT_HIP_CHECK(hipModuleLaunchKernel(f, grid.x, 1, 1, BEAT_BLOCK, 1, 1, 0, ctx->stream, args, nullptr));
    return BEAT_OK;
  }
}
