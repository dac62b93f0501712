// Host build of the TP06 generalized-Rush-Larsen step (fenicsx-beat_amd/csrc/ionic_models.h) for the CPU test suite: the
// same source the HIP kernel compiles -- the table-driven exp / log of FastMath included (its 256- and 128-entry tables
// read from host memory instead of LDS); only the hardware reciprocal estimate (v_rcp_f64) is replaced by 1/x, which
// the kernel's Newton step then polishes exactly as on the device.  Twin of tests/torord_host_harness.cpp; built by
// tests/test_tp06_host.py plainly and with -fsanitize=address,undefined.
//   tp06_host <states.bin> <params.bin> <out.bin> n t dt     (states: (19, n) doubles row-major; params: (53,) or (53, n))
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime.h>  // (host side of the HIP headers: g++ sees __device__ / __forceinline__ as plain inline)

// device intrinsics the headers mention
template <class T>
static inline T __shfl_down(T v, int, int) { return v; }
static inline void __syncthreads() {}
struct Dim3Stub { unsigned x = 0, y = 0, z = 0; };
static Dim3Stub threadIdx, blockIdx, blockDim, gridDim;
static inline int __double2hiint(double x) {
  uint64_t b;
  std::memcpy(&b, &x, 8);
  return (int)(uint32_t)(b >> 32);
}
static inline int __double2loint(double x) {
  uint64_t b;
  std::memcpy(&b, &x, 8);
  return (int)(uint32_t)(b & 0xffffffffu);
}
static inline double __hiloint2double(int hi, int lo) {
  const uint64_t b = ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo;
  double x;
  std::memcpy(&x, &b, 8);
  return x;
}
static inline double __builtin_amdgcn_rcp(double x) { return 1.0 / x; }
static inline double __builtin_amdgcn_rsq(double x) { return 1.0 / std::sqrt(x); }
static inline void __builtin_amdgcn_sched_barrier(int) {}
#include "../fenicsx-beat_amd/csrc/ionic_models.h"

using Model = Tp06Grl1;

struct HostIO {
  const double* in;
  double* out;
  long n, i;
  double load(int k) const { return in[(long)k * n + i]; }
  void store(int k, double v) const { out[(long)k * n + i] = v; }
};

int main(int argc, char** argv) {
  if (argc < 7) return 2;
  const long n = std::atol(argv[4]);
  const double t = std::atof(argv[5]), dt = std::atof(argv[6]);
  std::vector<double> S((size_t)Model::NS * n), O((size_t)Model::NS * n, 0.0);
  FILE* f = std::fopen(argv[1], "rb");
  if (!f || std::fread(S.data(), 8, S.size(), f) != S.size()) return 3;
  std::fclose(f);
  f = std::fopen(argv[2], "rb");
  if (!f) return 3;
  std::vector<double> P((size_t)Model::NP * n);
  const size_t got = std::fread(P.data(), 8, P.size(), f);
  std::fclose(f);
  const bool per_node = got == P.size() && n > 1;
  if (!per_node && got < (size_t)Model::NP) return 3;
  const Model::FM fm{kExp2Tab, kLogTab};  // (on the host both flavours scale by ldexp and read the plain table)
  // optional: <nsteps> <steps_per_beat> -- the in-kernel time loop of beat_ode_run on the host: nsteps steps in all, t restarts
  // at the given t every steps_per_beat steps (src/beat/single_cell.py:86-156 paces a cell this way)
  const long nsteps = argc > 7 ? std::atol(argv[7]) : 1;
  const long per_beat = argc > 8 ? std::atol(argv[8]) : nsteps;
  std::vector<typename Model::Derived> Q;
  std::vector<double> PL((size_t)Model::NP * n);
  for (long i = 0; i < n; ++i) {
    for (int k = 0; k < Model::NP; ++k) PL[(size_t)i * Model::NP + k] = per_node ? P[(size_t)k * n + i] : P[k];
    Q.push_back(Model::derive(&PL[(size_t)i * Model::NP]));
  }
  for (long j = 0; j < nsteps; ++j) {
    const double tj = t + (double)(j % per_beat) * dt;
    for (long i = 0; i < n; ++i) {
      const HostIO io{S.data(), O.data(), n, i};
      Model::step(io, &PL[(size_t)i * Model::NP], Q[(size_t)i], fm, tj, dt);
    }
    if (j + 1 < nsteps) S.swap(O);
  }
  f = std::fopen(argv[3], "wb");
  if (!f) return 4;
  std::fwrite(O.data(), 8, O.size(), f);
  std::fclose(f);
  return 0;
}
