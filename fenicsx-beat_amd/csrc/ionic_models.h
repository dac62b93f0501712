// Pointwise cell models, one node per thread, states in registers.
//
// Each model is a struct with
//   NS, NP            number of states / parameters
//   struct Derived    parameter-only sub-expressions, computed once per launch on the host
//   derive(p)         host: parameters -> Derived
//   step(io, p, d, fm, t, dt)   device: advances the node's states by one step; states are read
//                         with io.load(k) when first needed and written with io.store(k, value) as
//                         soon as they are final, so that few of them are live at any time
//
// Arithmetic follows the model specifications the reference feeds to its ODE solver
// (src/beat/odesolver.py:67-79): see the citation at each model.
#pragma once

#include <cmath>

#include "beat_common.h"


// ------------------------------------------------------------------------------------------------
// exp() for the cell-model kernels.  x = k ln2/64 + r with |r| <= ln2/128, k = 64 m + j:
//   exp(x) = 2^m * 2^(j/64) * (1 + r + r^2/2 + ... + r^5/120)
// 2^(j/64) comes from a 64-entry table staged in LDS (512 B, lanes hitting different entries are at
// worst a 2-way bank conflict), the truncation error of the degree-5 polynomial is < 4e-17, so the
// result is within ~1 ulp.  No overflow / NaN handling: arguments in these models are bounded
// (|x| < 700); large negative arguments underflow to 0 through v_ldexp_f64.  ~14 instructions
// against ~27 for the library routine -- the ionic kernels are fp64-issue bound.
// ------------------------------------------------------------------------------------------------
__device__ const double kExp2Tab[64] = {
    1.0000000000000000000, 1.0108892860517004600, 1.0218971486541166782, 1.0330248790212284225,
    1.0442737824274138403, 1.0556451783605571588, 1.0671404006768236182, 1.0787607977571197937,
    1.0905077326652576592, 1.1023825833078409436, 1.1143867425958925363, 1.1265216186082418998,
    1.1387886347566916537, 1.1511892299529827058, 1.1637248587775775138, 1.1763969916502812763,
    1.1892071150027210667, 1.2021567314527031421, 1.2152473599804688781, 1.2284805361068700057,
    1.2418578120734840486, 1.2553807570246910896, 1.2690509571917332226, 1.2828700160787782807,
    1.2968395546510096659, 1.3109612115247643419, 1.3252366431597412946, 1.3396675240533030054,
    1.3542555469368927283, 1.3690024229745906119, 1.3839098819638319549, 1.3989796725383111402,
    1.4142135623730950488, 1.4296133383919700112, 1.4451808069770466200, 1.4609177941806469887,
    1.4768261459394993114, 1.4929077282912648492, 1.5091644275934227398, 1.5255981507445383069,
    1.5422108254079408236, 1.5590044002378369670, 1.5759808451078864865, 1.5931421513422668979,
    1.6104903319492543082, 1.6280274218573477668, 1.6457554781539648445, 1.6636765803267364350,
    1.6817928305074290861, 1.7001063537185234695, 1.7186192981224779156, 1.7373338352737062490,
    1.7562521603732994831, 1.7753764925265212526, 1.7947090750031071864, 1.8142521755003987562,
    1.8340080864093424635, 1.8539791250833855684, 1.8741676341102999013, 1.8945759815869656413,
    1.9152065613971472939, 1.9360617934922944506, 1.9571441241754002690, 1.9784560263879509683,
};

struct FastMath {
  const double* __restrict__ tab;  // LDS copy of kExp2Tab
  __device__ __forceinline__ double exp(double x) const {
    const double k = __builtin_rint(x * 92.33248261689366);             // 64 / ln 2
    double r = fma(k, -0.01083042469326756, x);                         // ln2/64, high part (exact product)
    r = fma(k, -2.9815858269852933e-12, r);                             // low part
    const int ki = (int)k;
    const double t = tab[ki & 63];
    double p = fma(r, 1.0 / 120.0, 1.0 / 24.0);
    p = fma(r, p, 1.0 / 6.0);
    p = fma(r, p, 0.5);
    p = fma(r * r, p, r);                                               // exp(r) - 1
    return ldexp(fma(t, p, t), ki >> 6);
  }
};

// Access to the state-major array for one node (row k at base + k*ld).
struct NodeIO {
  double* __restrict__ base;
  int64_t ld, i;
  double* __restrict__ v_copy;  // optional mirror of row v_index (the PDE unknown), may be null
  int v_index;
  __device__ __forceinline__ double load(int k) const { return base[(int64_t)k * ld + i]; }
  __device__ __forceinline__ void store(int k, double v) const {
    base[(int64_t)k * ld + i] = v;
    if (v_copy != nullptr && k == v_index) v_copy[i] = v;
  }
};

// ------------------------------------------------------------------------------------------------
// v' = -a s, s' = b v, forward Euler  (tests/test_odesolver.py:11-17)
// ------------------------------------------------------------------------------------------------
struct SimpleOde {
  static constexpr int NS = 2, NP = 2;
  struct Derived {};
  __host__ __device__ static Derived derive(const double*) { return {}; }
  __device__ static __forceinline__ void step(const NodeIO& io, const double* p, const Derived&, const FastMath&,
                                              double, double dt) {
    const double v = io.load(0), s = io.load(1);
    io.store(0, v - p[0] * s * dt);
    io.store(1, s + p[1] * v * dt);
  }
};

// ------------------------------------------------------------------------------------------------
// FitzHugh-Nagumo, forward Euler, demo variant (demos/fitzhughnagumo.py:45-80, 224-225)
// states [s, V]; parameters [V_peak, V_rest, a, b, c_1, c_2, c_3, stim_amplitude, stim_duration, stim_start]
// ------------------------------------------------------------------------------------------------
struct FhnDemo {
  static constexpr int NS = 2, NP = 10;
  struct Derived {};
  __host__ __device__ static Derived derive(const double*) { return {}; }
  __device__ static __forceinline__ void step(const NodeIO& io, const double* p, const Derived&, const FastMath&,
                                              double t, double dt) {
    const double s = io.load(0), V = io.load(1);
    const double V_peak = p[0], V_rest = p[1], a = p[2], b = p[3], c_1 = p[4], c_2 = p[5],
                 c_3 = p[6], stim_amplitude = p[7], stim_duration = p[8], stim_start = p[9];
    const double V_amp = V_peak - V_rest;
    const double i_Stim = (t >= stim_start && t <= stim_start + stim_duration) ? stim_amplitude : 0.0;
    const double ds_dt = b * (-c_3 * s + (V - V_rest));
    const double V_th = V_amp * a + V_rest;
    const double I = -s * (c_2 / V_amp) * (V - V_rest) +
                     (((c_1 / (V_amp * V_amp)) * (V - V_rest)) * (V - V_th)) * (-V + V_peak);
    const double dV_dt = I + i_Stim;
    io.store(0, s + dt * ds_dt);
    io.store(1, V + dt * dV_dt);
  }
};

// ------------------------------------------------------------------------------------------------
// FitzHugh-Nagumo, forward Euler, README variant (README.md:58-89)
// states [s, v]; parameters [c_1, c_2, c_3, a, b, v_amp, v_rest, v_peak, stim_amplitude, stim_duration, stim_start]
// ------------------------------------------------------------------------------------------------
struct FhnReadme {
  static constexpr int NS = 2, NP = 11;
  struct Derived {};
  __host__ __device__ static Derived derive(const double*) { return {}; }
  __device__ static __forceinline__ void step(const NodeIO& io, const double* p, const Derived&, const FastMath&,
                                              double t, double dt) {
    const double s = io.load(0), v = io.load(1);
    const double c_1 = p[0], c_2 = p[1], c_3 = p[2], a = p[3], b = p[4], v_amp = p[5],
                 v_rest = p[6], v_peak = p[7], stim_amplitude = p[8], stim_duration = p[9],
                 stim_start = p[10];
    const double i_app = (t > stim_start && t < stim_start + stim_duration) ? stim_amplitude : 0.0;
    const double ds_dt = b * (-c_3 * s + (v - v_rest));
    const double v_th = v_amp * a + v_rest;
    const double I = -s * (c_2 / v_amp) * (v - v_rest) +
                     (((c_1 / (v_amp * v_amp)) * (v - v_rest)) * (v - v_th)) * (-v + v_peak);
    const double dV_dt = I + i_app;
    io.store(0, ds_dt * dt + s);
    io.store(1, v + dV_dt * dt);
  }
};

// ------------------------------------------------------------------------------------------------
// ten Tusscher & Panfilov 2006 (epi), first-order generalized Rush-Larsen.
// Specification: odes/tentusscher_panfilov_2006/tentusscher_panfilov_2006_epi_cell.ode:36-322.
// Scheme: gotranx `generalized_rush_larsen` as used by demos/niederer_benchmark.py:82-99:
//   y_i += f_i (exp(J_i dt) - 1) / J_i  if |J_i| > 1e-8 else dt f_i,
//   J_i = d f_i / d y_i with all intermediate expressions resolved (total self-derivative; this
//   variant is the one that reproduces the reference's Niederer table, see oracle/ionic.py).
// State / parameter order = order of appearance in the .ode file.
//
// Arithmetic notes (all fp64, results agree with the literal NumPy restatement to ~1e-13):
//  * gates: f = (inf - y)/tau, J = -1/tau  =>  y += (inf - y) (1 - exp(-dt/tau)); the |J| > 1e-8
//    guard can never trigger there (tau << 1e8 ms).
//  * exponentials whose arguments differ by a constant factor share one exp():
//    exp(+-(V+c)/k) for k in {5, 10, 20} come from E20 = exp(V/20) by squaring, k = 7 and k = 6
//    likewise; exp(-V F/RT) = exp(-0.1 V F/RT)^10, exp((gamma-1) V F/RT) = exp(gamma V F/RT) exp(-V F/RT),
//    the four K1 exponentials come from exp(0.02 u) and exp(0.0002 u), u = V - E_K.
//  * a/b is computed as a * rcp(b) with two Newton steps on v_rcp_f64 (<= 1 ulp for the finite,
//    normal denominators that occur here) instead of the ~12-instruction IEEE division sequence.
//  * parameter-only sub-expressions are evaluated once per launch on the host (Derived).
// ------------------------------------------------------------------------------------------------
struct Tp06Grl1 {
  static constexpr int NS = 19, NP = 53;
  enum S { Xr1, Xr2, Xs, m, h, j, d, f, f2, fCass, s, r, R_prime, Ca_i, Ca_SR, Ca_ss, Na_i, V, K_i };
  enum P {
    P_kna, g_K1, g_Kr, g_Ks, g_Na, g_bna, g_CaL, g_bca, g_to, P_NaK, K_mk, K_mNa, K_NaCa, K_sat,
    alpha, gamma, Km_Ca, Km_Nai, g_pCa, K_pCa, g_pK, Ca_o, k1_prime, k2_prime, k3, k4, EC, max_sr,
    min_sr, V_rel, V_xfer, K_up, V_leak, Vmax_up, Buf_c, K_buf_c, Buf_sr, K_buf_sr, Buf_ss,
    K_buf_ss, V_sr, V_ss, Na_o, R, T, F, Cm, V_c, stim_start, stim_period, stim_duration,
    stim_amplitude, K_o
  };

  struct Derived {
    double RTF, FRT, halfRTF, sqrtKo, gK1s, gKrs, KoPk, cCaL, NaK_B, Nao3, A2c, kNaCaQ, gm1,
        cVF, c1, c2, c3, c4, c5, Kup2, BKc, BKsr, BKss, dsr;
  };
  __host__ __device__ static Derived derive(const double* p) {
    Derived q;
    q.RTF = p[R] * p[T] / p[F];
    q.FRT = p[F] / (p[R] * p[T]);
    q.halfRTF = 0.5 * p[R] * p[T] / p[F];
    q.sqrtKo = sqrt(p[K_o] / 5.4);
    q.gK1s = p[g_K1] * q.sqrtKo;
    q.gKrs = p[g_Kr] * q.sqrtKo;
    q.KoPk = p[K_o] + p[P_kna] * p[Na_o];
    q.cCaL = p[g_CaL] * 4.0 * (p[F] * p[F]) / (p[R] * p[T]);
    q.NaK_B = p[P_NaK] * p[K_o] / (p[K_o] + p[K_mk]);
    q.Nao3 = p[Na_o] * p[Na_o] * p[Na_o];
    q.A2c = q.Nao3 * p[alpha];
    q.kNaCaQ = p[K_NaCa] / ((p[Km_Nai] * p[Km_Nai] * p[Km_Nai] + q.Nao3) * (p[Km_Ca] + p[Ca_o]));
    q.gm1 = p[gamma] - 1.0;
    q.cVF = p[Cm] / (p[V_c] * p[F]);
    q.c1 = p[Cm] / (2.0 * p[V_c] * p[F]);
    q.c2 = p[V_sr] / p[V_c];
    q.c3 = p[Cm] / (2.0 * p[V_ss] * p[F]);
    q.c4 = p[V_sr] / p[V_ss];
    q.c5 = p[V_c] / p[V_ss];
    q.Kup2 = p[K_up] * p[K_up];
    q.BKc = p[Buf_c] * p[K_buf_c];
    q.BKsr = p[Buf_sr] * p[K_buf_sr];
    q.BKss = p[Buf_ss] * p[K_buf_ss];
    q.dsr = p[max_sr] - p[min_sr];
    return q;
  }

  // 1/x: hardware estimate + two Newton steps
  __device__ static __forceinline__ double rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
  }
  __device__ static __forceinline__ double grl1(const FastMath& fm, double y, double fy, double J, double dt) {
    return y + ((fabs(J) > 1e-8) ? fy * (fm.exp(J * dt) - 1.0) * rcp(J) : fy * dt);
  }
  // gate with f = (inf - y)/tau, J = -1/tau
  __device__ static __forceinline__ double gate(const FastMath& fm, double y, double inf, double rtau, double dt) {
    return y + (inf - y) * (1.0 - fm.exp(-dt * rtau));
  }

  // Fence for the instruction scheduler: the step is ~3000 straight-line instructions with ~50
  // independent exp() chains; without fences everything is hoisted and one wave needs the whole
  // register file.  Each fenced block keeps a few chains in flight, which is all the latency hiding
  // 3-4 resident waves per SIMD need.
#define BEAT_FENCE() __builtin_amdgcn_sched_barrier(0)

  __device__ static __forceinline__ void step(const NodeIO& io, const double* p, const Derived& q, const FastMath& fm,
                                              double t, double dt) {
    const double v = io.load(V);
    // exp(c) constants, c written out in the comment
    constexpr double EXP_M3 = 0.049787068367863944;    // exp(-3)
    constexpr double EXP_2P5 = 12.182493960703473;     // exp(2.5)
    constexpr double EXP_M4P5 = 0.011108996538242306;  // exp(-4.5)
    constexpr double EXP_1P3 = 3.6692966676192444;     // exp(1.3)
    constexpr double EXP_3 = 20.085536923187668;       // exp(3)
    constexpr double EXP_M3P2 = 0.04076220397836621;   // exp(-3.2)
    constexpr double EXP_M12 = 6.14421235332821e-06;   // exp(-12)
    constexpr double EXP_7 = 1096.6331584284585;       // exp(7)
    constexpr double EXP_1 = 2.718281828459045;        // exp(1)
    constexpr double EXP_4 = 54.598150033144236;       // exp(4)
    constexpr double EXP_M4 = 0.01831563888873418;     // exp(-4)
    constexpr double EXP_M26_7 = 0.0243728440732796;   // exp(-26/7)
    constexpr double EXP_20_7 = 17.41170806332765;     // exp(20/7)
    constexpr double EXP_5 = 148.4131591025766;        // exp(5)
    constexpr double EXP_5_6 = 2.300975890892825;      // exp(5/6)
    constexpr double EXP_20_6 = 28.03162489452614;     // exp(20/6)
    constexpr double EXP_M1 = 0.36787944117144233;     // exp(-1)
    constexpr double EXP_P02 = 1.0202013400267558;     // exp(0.02)

    // ---- conductances from the OLD gate values (the gates are overwritten below) ----------------------
    const double oXr1 = io.load(Xr1), oXr2 = io.load(Xr2), oXs = io.load(Xs), om = io.load(m), oh = io.load(h),
                 oj = io.load(j), od = io.load(d), of = io.load(f), of2 = io.load(f2), ofCass = io.load(fCass),
                 os = io.load(s), orr = io.load(r);
    const double gNa = p[g_Na] * (om * om * om) * oh * oj;
    const double gKr = q.gKrs * oXr1 * oXr2;
    const double gKs = p[g_Ks] * (oXs * oXs);
    const double gto = p[g_to] * orr * os;
    const double gates_CaL = q.cCaL * od * of * of2 * ofCass;

    // ---- shared exponentials of V ----------------------------------------------------------------------
    const double E20 = fm.exp(0.05 * v), I20 = rcp(E20);
    const double E10 = E20 * E20, I10 = I20 * I20;
    const double E5 = E10 * E10, I5 = I10 * I10;
    const double E7 = fm.exp(v * (1.0 / 7.0)), I7 = rcp(E7);
    const double I6 = fm.exp(v * (-1.0 / 6.0));
    BEAT_FENCE();

    // ---- gates: y += (inf - y)(1 - exp(-dt/tau)); time constants as single quotients ------------------------
    {  // Xr1 (.ode:189-194): tau = 450/(1+ea) * 6/(1+eb)
      const double inf = rcp(1.0 + EXP_M26_7 * I7);                          // exp((-26 - V)/7)
      const double rtau = (1.0 + EXP_M4P5 * I10) * (1.0 + fm.exp((v + 30.0) * (1.0 / 11.5))) * (1.0 / 2700.0);
      io.store(Xr1, gate(fm, oXr1, inf, rtau, dt));
    }
    {  // Xr2 (.ode:196-201): tau = 3/(1+ea) * 1.12/(1+eb)
      const double inf = rcp(1.0 + fm.exp((v + 88.0) * (1.0 / 24.0)));
      const double rtau = (1.0 + EXP_M3 * I20) * (1.0 + EXP_M3 * E20) * (1.0 / 3.36);
      io.store(Xr2, gate(fm, oXr2, inf, rtau, dt));
    }
    BEAT_FENCE();
    {  // Xs (.ode:206-211): tau = 1400/sqrt(1+ea) * 1/(1+eb) + 80 = (1400 + 80 D)/D
      const double inf = rcp(1.0 + fm.exp((-5.0 - v) * (1.0 / 14.0)));
      const double D = sqrt(1.0 + EXP_5_6 * I6) * (1.0 + fm.exp((v - 35.0) * (1.0 / 15.0)));  // exp((5 - V)/6)
      const double rtau = D * rcp(1400.0 + 80.0 * D);
      io.store(Xs, gate(fm, oXs, inf, rtau, dt));
    }
    BEAT_FENCE();
    {  // m (.ode:216-221): tau = 1/(1+ea) * (0.1/(1+eb) + 0.1/(1+ec))
      const double rm = rcp(1.0 + fm.exp((-56.86 - v) * (1.0 / 9.03)));
      const double da = 1.0 + EXP_M12 * I5;                                   // exp((-60 - V)/5)
      const double db = 1.0 + EXP_7 * E5;                                     // exp((V + 35)/5)
      const double dc = 1.0 + fm.exp((v - 50.0) * (1.0 / 200.0));
      const double rtau = da * db * dc * 10.0 * rcp(db + dc);
      io.store(m, gate(fm, om, rm * rm, rtau, dt));
    }
    BEAT_FENCE();
    {  // h, j (.ode:223-235): tau = 1/(alpha + beta), shared steady state
      const double rh = rcp(1.0 + fm.exp((v + 71.55) * (1.0 / 7.43)));
      const double h_inf = rh * rh;
      double ah_bh, aj_bj;
      if (v < -40.0) {
        ah_bh = 0.057 * fm.exp(-(v + 80.0) * (1.0 / 6.8)) + 2.7 * fm.exp(0.079 * v) + 310000.0 * fm.exp(0.3485 * v);
        const double da = 1.0 + fm.exp(0.311 * (v + 79.23)), db = 1.0 + fm.exp(-0.1378 * (v + 40.14));
        const double na = (-25428.0 * fm.exp(0.2444 * v) - 6.948e-6 * fm.exp(-0.04391 * v)) * (v + 37.78);
        const double nb = 0.02424 * fm.exp(-0.01052 * v);
        aj_bj = (na * db + nb * da) * rcp(da * db);
      } else {
        ah_bh = 0.77 * rcp(0.13 * (1.0 + fm.exp((v + 10.66) * (-1.0 / 11.1))));
        aj_bj = 0.6 * fm.exp(0.057 * v) * rcp(1.0 + EXP_M3P2 * I10);         // exp(-0.1 (V + 32))
      }
      io.store(h, gate(fm, oh, h_inf, ah_bh, dt));
      io.store(j, gate(fm, oj, h_inf, aj_bj, dt));
    }
    BEAT_FENCE();
    {  // d (.ode:243-249): tau = (1.4/(1+ea) + 0.25) * 1.4/(1+eb) + 1/(1+ec)
      const double inf = rcp(1.0 + fm.exp((-8.0 - v) * (1.0 / 7.5)));
      const double da = 1.0 + fm.exp((-35.0 - v) * (1.0 / 13.0));
      const double db = 1.0 + EXP_1 * E5;                                     // exp((V + 5)/5)
      const double dc = 1.0 + EXP_2P5 * I20;                                  // exp((50 - V)/20)
      const double num = (1.4 + 0.25 * da) * 1.4 * dc + da * db;
      const double rtau = da * db * dc * rcp(num);
      io.store(d, gate(fm, od, inf, rtau, dt));
    }
    BEAT_FENCE();
    {  // f, f2 (.ode:251-259): tau = c G + A/(1+ea) + B/(1+eb) [+ 20]
      const double v27sq = (v + 27.0) * (v + 27.0);
      const double db = 1.0 + EXP_3 * E10;                                    // exp((V + 30)/10)
      {
        const double inf = rcp(1.0 + EXP_20_7 * E7);                          // exp((V + 20)/7)
        const double da = 1.0 + EXP_1P3 * I10;                                // exp((13 - V)/10)
        const double dab = da * db;
        const double num = (1102.5 * fm.exp(v27sq * (-1.0 / 225.0)) + 20.0) * dab + 200.0 * db + 180.0 * da;
        io.store(f, gate(fm, of, inf, dab * rcp(num), dt));
      }
      {
        const double inf = 0.67 * rcp(1.0 + EXP_5 * E7) + 0.33;               // exp((V + 35)/7)
        const double da = 1.0 + EXP_2P5 * I10;                                // exp((25 - V)/10)
        const double dab = da * db;
        const double num = 562.0 * fm.exp(v27sq * (-1.0 / 240.0)) * dab + 31.0 * db + 80.0 * da;
        io.store(f2, gate(fm, of2, inf, dab * rcp(num), dt));
      }
    }
    BEAT_FENCE();
    {  // s, r (.ode:276-284)
      const double ds_ = 1.0 + EXP_M4 * E5;                                   // exp((V - 20)/5)
      const double s_inf = rcp(1.0 + EXP_4 * E5);                             // exp((V + 20)/5)
      const double num = (85.0 * fm.exp((v + 45.0) * (v + 45.0) * (-1.0 / 320.0)) + 3.0) * ds_ + 5.0;
      io.store(s, gate(fm, os, s_inf, ds_ * rcp(num), dt));
      const double r_inf = rcp(1.0 + EXP_20_6 * I6);                          // exp((20 - V)/6)
      const double rtau_r = rcp(9.5 * fm.exp((v + 40.0) * (v + 40.0) * (-1.0 / 1800.0)) + 0.8);
      io.store(r, gate(fm, orr, r_inf, rtau_r, dt));
    }
    BEAT_FENCE();

    const double vCai = io.load(Ca_i), vCaSR = io.load(Ca_SR), vCass = io.load(Ca_ss), vNai = io.load(Na_i),
                 vKi = io.load(K_i), vR = io.load(R_prime);
    {  // fCass (.ode:261-264): depends on Ca_ss only
      const double c2 = 1.0 + (vCass * 20.0) * (vCass * 20.0);                // 1 + (Ca_ss/0.05)^2
      const double rc2 = rcp(c2);
      io.store(fCass, gate(fm, ofCass, 0.6 * rc2 + 0.4, c2 * rcp(80.0 + 2.0 * c2), dt));
    }

    // ---- reversal potentials ------------------------------------------------------------------------
    const double rNai = rcp(vNai), rKi = rcp(vKi), rCai = rcp(vCai);
    const double rKs = rcp(vKi + p[P_kna] * vNai);
    const double E_Na = q.RTF * log(p[Na_o] * rNai);
    const double E_K = q.RTF * log(p[K_o] * rKi);
    const double E_Ks = q.RTF * log(q.KoPk * rKs);
    const double E_Ca = q.halfRTF * log(p[Ca_o] * rCai);
    const double u = v - E_K;
    BEAT_FENCE();

    // running sums: total membrane current, currents carried by K+, d(sum I)/dV
    double I_tot, I_K, dI_dV, sum_du;
    {  // inward rectifier (.ode:180-184) and its derivative w.r.t. u = V - E_K
      const double G = fm.exp(0.02 * u);
      const double G2 = G * G, G4 = G2 * G2, G5 = G4 * G, G10 = G5 * G5, G25 = G10 * G10 * G5;
      const double e1 = EXP_M12 * (G2 * G);                  // exp(0.06 (u - 200))
      const double e2 = EXP_P02 * fm.exp(0.0002 * u);        // exp(0.0002 (u + 100))
      const double e3 = EXP_M1 * G5;                         // exp(0.1 (u - 10))
      const double e4 = rcp(G25);                            // exp(-0.5 u)
      const double r1 = rcp(1.0 + e1);
      const double aK1 = 0.1 * r1;
      const double daK1 = -0.06 * aK1 * e1 * r1;
      const double rD = rcp(1.0 + e4);
      const double bK1 = (3.0 * e2 + e3) * rD;
      const double dbK1 = (0.0006 * e2 + 0.1 * e3 + 0.5 * e4 * bK1) * rD;
      const double rab = rcp(aK1 + bK1);
      const double xK1 = aK1 * rab;
      const double dxK1 = (daK1 * bK1 - aK1 * dbK1) * rab * rab;
      const double i_K1 = q.gK1s * xK1 * u;
      const double epK = fm.exp((25.0 - v) * (1.0 / 5.98));  // plateau K current (.ode:296)
      const double rpK = rcp(1.0 + epK);
      const double i_p_K = p[g_pK] * u * rpK;
      I_K = i_K1 + gto * u + gKr * u + i_p_K;
      sum_du = q.gK1s * (dxK1 * u + xK1) + gto + gKr + p[g_pK] * rpK;  // d/du of the u-driven currents
      dI_dV = sum_du + p[g_pK] * u * epK * (1.0 / 5.98) * rpK * rpK + gKs;
      I_K += gKs * (v - E_Ks);
      I_tot = I_K;
    }
    BEAT_FENCE();

    // ---- exponentials of V F/(R T); pumps and exchanger (.ode:286-294) --------------------------------------
    const double vF = v * q.FRT;
    const double e5 = fm.exp(-0.1 * vF);
    const double e5_2 = e5 * e5, e5_4 = e5_2 * e5_2, e5_8 = e5_4 * e5_4;
    const double e6 = e5_8 * e5_2;                                // exp(-V F/RT)
    const double eg = fm.exp(p[gamma] * vF);
    const double eg1 = eg * e6;                                   // exp((gamma - 1) V F/RT)
    const double rNaK = rcp(1.0 + 0.1245 * e5 + 0.0353 * e6);
    const double rNaKm = rcp(vNai + p[K_mNa]);
    const double i_NaK = q.NaK_B * vNai * rNaKm * rNaK;
    const double di_NaK_dNai = q.NaK_B * p[K_mNa] * rNaKm * rNaKm * rNaK;
    dI_dV += i_NaK * q.FRT * (0.01245 * e5 + 0.0353 * e6) * rNaK;
    const double A1 = (vNai * vNai * vNai) * p[Ca_o], A2 = q.A2c * vCai;
    const double rS = rcp(1.0 + p[K_sat] * eg1);
    const double NNaCa = eg * A1 - eg1 * A2;
    const double kS = q.kNaCaQ * rS;
    const double i_NaCa = kS * NNaCa;
    dI_dV += kS * q.FRT * ((p[gamma] * eg * A1 - q.gm1 * eg1 * A2) - NNaCa * p[K_sat] * q.gm1 * eg1 * rS);
    const double di_NaCa_dNai = kS * eg * 3.0 * (vNai * vNai) * p[Ca_o];
    const double di_NaCa_dCai = -kS * eg1 * q.A2c;
    const double i_Na_tot = (gNa + p[g_bna]) * (v - E_Na);        // i_Na + i_b_Na
    dI_dV += gNa + p[g_bna] + p[g_bca];
    const double i_b_Ca = p[g_bca] * (v - E_Ca);
    I_tot += i_NaK + i_NaCa + i_Na_tot + i_b_Ca;
    BEAT_FENCE();

    // ---- L-type calcium current (.ode:241) ---------------------------------------------------------------------
    // exp(2 (V - 15) F/RT) is kept as its own exp(): i_CaL divides by (eCaL - 1), which cancels near
    // V = 15 mV and would amplify the few-ulp error of a value derived from e6
    const double eCaL = fm.exp(2.0 * (v - 15.0) * q.FRT);
    const double w15 = v - 15.0;
    const double rDc = rcp(eCaL - 1.0);
    const double NCaL = 0.25 * vCass * eCaL - p[Ca_o];
    const double i_CaL = gates_CaL * w15 * NCaL * rDc;
    dI_dV += gates_CaL * (NCaL * rDc + w15 * (2.0 * q.FRT) * eCaL * (p[Ca_o] - 0.25 * vCass) * rDc * rDc);
    const double di_CaL_dCass = gates_CaL * w15 * 0.25 * eCaL * rDc;
    const double rpCa = rcp(vCai + p[K_pCa]);
    const double i_p_Ca = p[g_pCa] * vCai * rpCa;
    const double di_pCa_dCai = p[g_pCa] * p[K_pCa] * rpCa * rpCa;
    I_tot += i_CaL + i_p_Ca;
    BEAT_FENCE();

    // ---- membrane, sodium, potassium (.ode:318-322) ---------------------------------------------------------------
    const double tmod = t - floor(t / p[stim_period]) * p[stim_period];
    const double i_Stim =
        (tmod >= p[stim_start] && tmod <= p[stim_start] + p[stim_duration]) ? p[stim_amplitude] : 0.0;
    io.store(V, grl1(fm, v, -(I_tot + i_Stim), -dI_dV, dt));
    // dE_K/dK_i = -RTF/K_i, dE_Ks/dK_i = -RTF/(K_i + P_kna Na_i); currents depend on K_i only through them
    io.store(K_i, grl1(fm, vKi, -(I_K + i_Stim - 2.0 * i_NaK) * q.cVF,
                       -(sum_du * q.RTF * rKi + gKs * q.RTF * rKs) * q.cVF, dt));
    io.store(Na_i, grl1(fm, vNai, -(i_Na_tot + 3.0 * i_NaK + 3.0 * i_NaCa) * q.cVF,
                        -((gNa + p[g_bna]) * q.RTF * rNai + 3.0 * di_NaK_dNai + 3.0 * di_NaCa_dNai) * q.cVF, dt));
    BEAT_FENCE();

    // ---- calcium dynamics (.ode:298-316) -----------------------------------------------------------------------------
    const double qup = q.Kup2 * rCai * rCai;
    const double rup = rcp(1.0 + qup);
    const double i_up = p[Vmax_up] * rup;
    const double di_up_dCai = i_up * 2.0 * qup * rCai * rup;
    const double i_leak = p[V_leak] * (vCaSR - vCai);
    const double i_xfer = p[V_xfer] * (vCass - vCai);
    {  // Ca_i
      const double T_i = -(i_b_Ca + i_p_Ca - 2.0 * i_NaCa) * q.c1 + (i_leak - i_up) * q.c2 + i_xfer;
      const double dT_i = -(p[g_bca] * q.halfRTF * rCai + di_pCa_dCai - 2.0 * di_NaCa_dCai) * q.c1 +
                          (-p[V_leak] - di_up_dCai) * q.c2 - p[V_xfer];
      const double rbc = rcp(vCai + p[K_buf_c]);
      const double gci = q.BKc * rbc * rbc;
      const double Fr_i = rcp(1.0 + gci);
      io.store(Ca_i, grl1(fm, vCai, T_i * Fr_i, dT_i * Fr_i + T_i * (Fr_i * Fr_i * 2.0 * gci * rbc), dt));
    }
    BEAT_FENCE();
    const double rCaSR = rcp(vCaSR);
    const double zsr = (p[EC] * rCaSR) * (p[EC] * rCaSR);
    const double rz = rcp(1.0 + zsr);
    const double kcasr = p[max_sr] - q.dsr * rz;
    const double dkcasr = -2.0 * q.dsr * zsr * rCaSR * rz * rz;
    const double rkc = rcp(kcasr);
    const double k1 = p[k1_prime] * rkc;
    const double dk1 = -k1 * dkcasr * rkc;
    const double k2 = p[k2_prime] * kcasr;
    const double css2 = vCass * vCass;
    const double rO = rcp(p[k3] + k1 * css2);
    const double O = k1 * css2 * vR * rO;
    const double dO_dk1 = css2 * vR * p[k3] * rO * rO;
    const double dO_dCass = 2.0 * vCass * k1 * vR * p[k3] * rO * rO;
    const double dsrss = vCaSR - vCass;
    const double i_rel = p[V_rel] * O * dsrss;
    io.store(R_prime, grl1(fm, vR, -k2 * vCass * vR + p[k4] * (1.0 - vR), -vCass * k2 - p[k4], dt));
    {  // Ca_SR
      const double T_sr = i_up - (i_rel + i_leak);
      const double dT_sr = -(p[V_rel] * (dO_dk1 * dk1 * dsrss + O) + p[V_leak]);
      const double rbsr = rcp(vCaSR + p[K_buf_sr]);
      const double gsr = q.BKsr * rbsr * rbsr;
      const double Fr_sr = rcp(1.0 + gsr);
      io.store(Ca_SR, grl1(fm, vCaSR, T_sr * Fr_sr, dT_sr * Fr_sr + T_sr * (Fr_sr * Fr_sr * 2.0 * gsr * rbsr), dt));
    }
    BEAT_FENCE();
    {  // Ca_ss
      const double T_ss = -i_CaL * q.c3 + i_rel * q.c4 - i_xfer * q.c5;
      const double dT_ss = -di_CaL_dCass * q.c3 + p[V_rel] * (dO_dCass * dsrss - O) * q.c4 - p[V_xfer] * q.c5;
      const double rbss = rcp(vCass + p[K_buf_ss]);
      const double gss = q.BKss * rbss * rbss;
      const double Fr_ss = rcp(1.0 + gss);
      io.store(Ca_ss, grl1(fm, vCass, T_ss * Fr_ss, dT_ss * Fr_ss + T_ss * (Fr_ss * Fr_ss * 2.0 * gss * rbss), dt));
    }
  }
#undef BEAT_FENCE
};
