// Host build of the hand-organised ToR-ORd-dynCl step (fenicsx-beat_amd/csrc/torord_dyncl.h) for the CPU test suite:
// the same source the HIP kernel compiles, with exp / log / reciprocal from libm.
//   torord_host <states.bin> <params.bin> <out.bin> n t dt     (states: (45, n) doubles row-major; params: (112,) or (112, n))
// Built with -DBEAT_HOST_LAND=1 it runs the Land instance of the same source (52 states, 140 parameters).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define BEAT_TORORD_HOST_TEST 1
#define BEAT_HD
#define BEAT_DV inline
#define BEAT_TFENCE() ((void)0)
#define BEAT_PIN(x) ((void)0)
#define BEAT_SCONST(c) (c)
static inline double beat_rcp(double x) { return 1.0 / x; }
static inline double beat_rsqrt(double x) { return 1.0 / std::sqrt(x); }
static inline double beat_guard(double v) { return std::fabs(v) < 1.0e-4 ? std::copysign(1.0e-4, v) : v; }
struct HostMath {
  double exp(double x) const { return std::exp(x); }
  double log(double x) const { return std::log(x); }
};
using std::exp;
using std::fabs;
using std::floor;
using std::fma;
using std::fmax;
using std::fmin;
using std::pow;
using std::sqrt;
#include "../fenicsx-beat_amd/csrc/torord_dyncl.h"

#ifdef BEAT_HOST_LAND
using Model = TorordLandGrl1;
#else
using Model = TorordDynClGrl1;
#endif

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
  std::vector<double> S(Model::NS * n), O(Model::NS * n, 0.0);
  FILE* f = std::fopen(argv[1], "rb");
  if (!f || std::fread(S.data(), 8, S.size(), f) != S.size()) return 3;
  std::fclose(f);
  f = std::fopen(argv[2], "rb");
  if (!f) return 3;
  std::vector<double> P(Model::NP * n);
  const size_t got = std::fread(P.data(), 8, P.size(), f);
  std::fclose(f);
  const bool per_node = got == P.size() && n > 1;
  if (!per_node && got < (size_t)Model::NP) return 3;
  const HostMath fm;
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
