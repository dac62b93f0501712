// Pointwise cell models, one node per thread, states in registers.
//
// Each model is a struct with
//   NS, NP            number of states / parameters
//   struct Derived    parameter-only sub-expressions, computed once per launch on the host
//   derive(p)         host: parameters -> Derived
//   step(y, p, d, t, dt)   device: advances y[NS] in place by one step
//
// Arithmetic follows the model specifications the reference feeds to its ODE solver
// (src/beat/odesolver.py:67-79): see the citation at each model.
#pragma once

#include <cmath>

#include "beat_common.h"

// ------------------------------------------------------------------------------------------------
// v' = -a s, s' = b v, forward Euler  (tests/test_odesolver.py:11-17)
// ------------------------------------------------------------------------------------------------
struct SimpleOde {
  static constexpr int NS = 2, NP = 2;
  struct Derived {};
  __host__ __device__ static Derived derive(const double*) { return {}; }
  __device__ static __forceinline__ void step(double* y, const double* p, const Derived&, double,
                                              double dt) {
    const double v = y[0], s = y[1];
    y[0] = v - p[0] * s * dt;
    y[1] = s + p[1] * v * dt;
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
  __device__ static __forceinline__ void step(double* y, const double* p, const Derived&, double t,
                                              double dt) {
    const double s = y[0], V = y[1];
    const double V_peak = p[0], V_rest = p[1], a = p[2], b = p[3], c_1 = p[4], c_2 = p[5],
                 c_3 = p[6], stim_amplitude = p[7], stim_duration = p[8], stim_start = p[9];
    const double V_amp = V_peak - V_rest;
    const double i_Stim = (t >= stim_start && t <= stim_start + stim_duration) ? stim_amplitude : 0.0;
    const double ds_dt = b * (-c_3 * s + (V - V_rest));
    const double V_th = V_amp * a + V_rest;
    const double I = -s * (c_2 / V_amp) * (V - V_rest) +
                     (((c_1 / (V_amp * V_amp)) * (V - V_rest)) * (V - V_th)) * (-V + V_peak);
    const double dV_dt = I + i_Stim;
    y[0] = s + dt * ds_dt;
    y[1] = V + dt * dV_dt;
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
  __device__ static __forceinline__ void step(double* y, const double* p, const Derived&, double t,
                                              double dt) {
    const double s = y[0], v = y[1];
    const double c_1 = p[0], c_2 = p[1], c_3 = p[2], a = p[3], b = p[4], v_amp = p[5],
                 v_rest = p[6], v_peak = p[7], stim_amplitude = p[8], stim_duration = p[9],
                 stim_start = p[10];
    const double i_app = (t > stim_start && t < stim_start + stim_duration) ? stim_amplitude : 0.0;
    const double ds_dt = b * (-c_3 * s + (v - v_rest));
    const double v_th = v_amp * a + v_rest;
    const double I = -s * (c_2 / v_amp) * (v - v_rest) +
                     (((c_1 / (v_amp * v_amp)) * (v - v_rest)) * (v - v_th)) * (-v + v_peak);
    const double dV_dt = I + i_app;
    y[0] = ds_dt * dt + s;
    y[1] = v + dV_dt * dt;
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
  __device__ static __forceinline__ double grl1(double y, double fy, double J, double dt) {
    return y + ((fabs(J) > 1e-8) ? fy * (exp(J * dt) - 1.0) * rcp(J) : fy * dt);
  }
  // gate with f = (inf - y)/tau, J = -1/tau
  __device__ static __forceinline__ double gate(double y, double inf, double rtau, double dt) {
    return y + (inf - y) * (1.0 - exp(-dt * rtau));
  }

  __device__ static __forceinline__ void step(double* y, const double* p, const Derived& q, double t,
                                              double dt) {
    const double vXr1 = y[Xr1], vXr2 = y[Xr2], vXs = y[Xs], vm = y[m], vh = y[h], vj = y[j],
                 vd = y[d], vf = y[f], vf2 = y[f2], vfCass = y[fCass], vs = y[s], vr = y[r],
                 vR = y[R_prime], vCai = y[Ca_i], vCaSR = y[Ca_SR], vCass = y[Ca_ss],
                 vNai = y[Na_i], v = y[V], vKi = y[K_i];

    // ---- shared exponentials of V ---------------------------------------------------------------
    const double E20 = exp(0.05 * v), I20 = rcp(E20);
    const double E10 = E20 * E20, I10 = I20 * I20;
    const double E5 = E10 * E10, I5 = I10 * I10;
    const double E7 = exp(v * (1.0 / 7.0)), I7 = rcp(E7);
    const double I6 = exp(v * (-1.0 / 6.0));
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
    constexpr double EXP_20_7 = 17.41170806332765;      // exp(20/7)
    constexpr double EXP_5 = 148.4131591025766;          // exp(5)
    constexpr double EXP_5_6 = 2.300975890892825;       // exp(5/6)
    constexpr double EXP_20_6 = 28.03162489452614;      // exp(20/6)
    constexpr double EXP_M1 = 0.36787944117144233;       // exp(-1)
    constexpr double EXP_P02 = 1.0202013400267558;       // exp(0.02)

    // ---- reversal potentials ------------------------------------------------------------------------
    const double rNai = rcp(vNai), rKi = rcp(vKi), rCai = rcp(vCai);
    const double E_Na = q.RTF * log(p[Na_o] * rNai);
    const double E_K = q.RTF * log(p[K_o] * rKi);
    const double rKs = rcp(vKi + p[P_kna] * vNai);
    const double E_Ks = q.RTF * log(q.KoPk * rKs);
    const double E_Ca = q.halfRTF * log(p[Ca_o] * rCai);
    const double u = v - E_K;

    // ---- inward rectifier (.ode:180-184) and its derivative w.r.t. u = V - E_K -------------------------
    const double G = exp(0.02 * u);
    const double G2 = G * G, G4 = G2 * G2, G5 = G4 * G, G10 = G5 * G5, G25 = G10 * G10 * G5;
    const double e1 = EXP_M12 * (G2 * G);                  // exp(0.06 (u - 200))
    const double e2 = EXP_P02 * exp(0.0002 * u);           // exp(0.0002 (u + 100))
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
    const double di_K1_du = q.gK1s * (dxK1 * u + xK1);

    // ---- rapid / slow delayed rectifier (.ode:186-211) ------------------------------------------------
    const double gKr = q.gKrs * vXr1 * vXr2;
    const double i_Kr = gKr * u;
    const double xr1_inf = rcp(1.0 + EXP_M26_7 * I7);             // exp((-26 - V)/7)
    const double a_xr1 = 450.0 * rcp(1.0 + EXP_M4P5 * I10);       // exp((-45 - V)/10)
    const double b_xr1 = 6.0 * rcp(1.0 + exp((v + 30.0) * (1.0 / 11.5)));
    const double rtau_xr1 = rcp(a_xr1 * b_xr1);
    const double xr2_inf = rcp(1.0 + exp((v + 88.0) * (1.0 / 24.0)));
    const double a_xr2 = 3.0 * rcp(1.0 + EXP_M3 * I20);           // exp((-60 - V)/20)
    const double b_xr2 = 1.12 * rcp(1.0 + EXP_M3 * E20);          // exp((V - 60)/20)
    const double rtau_xr2 = rcp(a_xr2 * b_xr2);
    const double gKs = p[g_Ks] * (vXs * vXs);
    const double i_Ks = gKs * (v - E_Ks);
    const double xs_inf = rcp(1.0 + exp((-5.0 - v) * (1.0 / 14.0)));
    const double a_xs = 1400.0 * rcp(sqrt(1.0 + EXP_5_6 * I6));   // exp((5 - V)/6)
    const double b_xs = rcp(1.0 + exp((v - 35.0) * (1.0 / 15.0)));
    const double rtau_xs = rcp(a_xs * b_xs + 80.0);

    // ---- fast sodium (.ode:213-235) -------------------------------------------------------------------
    const double gNa = p[g_Na] * (vm * vm * vm) * vh * vj;
    const double i_Na = gNa * (v - E_Na);
    const double rm = rcp(1.0 + exp((-56.86 - v) * (1.0 / 9.03)));
    const double m_inf = rm * rm;
    const double a_m = rcp(1.0 + EXP_M12 * I5);                   // exp((-60 - V)/5)
    const double b_m = 0.1 * rcp(1.0 + EXP_7 * E5) + 0.1 * rcp(1.0 + exp((v - 50.0) * (1.0 / 200.0)));
    const double rtau_m = rcp(a_m * b_m);
    const double rh = rcp(1.0 + exp((v + 71.55) * (1.0 / 7.43)));
    const double h_inf = rh * rh;
    double ah_bh, aj_bj;  // alpha + beta = 1/tau
    if (v < -40.0) {
      ah_bh = 0.057 * exp(-(v + 80.0) * (1.0 / 6.8)) + 2.7 * exp(0.079 * v) + 310000.0 * exp(0.3485 * v);
      const double aj = (-25428.0 * exp(0.2444 * v) - 6.948e-6 * exp(-0.04391 * v)) * (v + 37.78) *
                        rcp(1.0 + exp(0.311 * (v + 79.23)));
      const double bj = 0.02424 * exp(-0.01052 * v) * rcp(1.0 + exp(-0.1378 * (v + 40.14)));
      aj_bj = aj + bj;
    } else {
      ah_bh = 0.77 * rcp(0.13 * (1.0 + exp((v + 10.66) * (-1.0 / 11.1))));
      aj_bj = 0.6 * exp(0.057 * v) * rcp(1.0 + EXP_M3P2 * I10);   // exp(-0.1 (V + 32))
    }

    // ---- background currents -------------------------------------------------------------------------
    const double i_b_Na = p[g_bna] * (v - E_Na);
    const double i_b_Ca = p[g_bca] * (v - E_Ca);

    // ---- exponentials of V F/(R T) ----------------------------------------------------------------------
    const double vF = v * q.FRT;
    const double e5 = exp(-0.1 * vF);
    const double e5_2 = e5 * e5, e5_4 = e5_2 * e5_2, e5_8 = e5_4 * e5_4;
    const double e6 = e5_8 * e5_2;                                // exp(-V F/RT)
    const double eg = exp(p[gamma] * vF);
    const double eg1 = eg * e6;                                   // exp((gamma - 1) V F/RT)
    // exp(2 (V - 15) F/RT): kept as its own exp() -- i_CaL divides by (eCaL - 1), which cancels
    // near V = 15 mV and would amplify the few-ulp error of a value derived from e6
    const double eCaL = exp(2.0 * (v - 15.0) * q.FRT);

    // ---- L-type calcium (.ode:240-268) ------------------------------------------------------------------
    const double gates_CaL = q.cCaL * vd * vf * vf2 * vfCass;
    const double w15 = v - 15.0;
    const double rDc = rcp(eCaL - 1.0);
    const double NCaL = 0.25 * vCass * eCaL - p[Ca_o];
    const double i_CaL = gates_CaL * w15 * NCaL * rDc;
    const double di_CaL_dV =
        gates_CaL * (NCaL * rDc + w15 * (2.0 * q.FRT) * eCaL * (p[Ca_o] - 0.25 * vCass) * rDc * rDc);
    const double di_CaL_dCass = gates_CaL * w15 * 0.25 * eCaL * rDc;
    const double d_inf = rcp(1.0 + exp((-8.0 - v) * (1.0 / 7.5)));
    const double a_d = 1.4 * rcp(1.0 + exp((-35.0 - v) * (1.0 / 13.0))) + 0.25;
    const double b_d = 1.4 * rcp(1.0 + EXP_1 * E5);               // exp((V + 5)/5)
    const double g_d = rcp(1.0 + EXP_2P5 * I20);                  // exp((50 - V)/20)
    const double rtau_d = rcp(a_d * b_d + g_d);
    const double f_inf = rcp(1.0 + EXP_20_7 * E7);                // exp((V + 20)/7)
    const double v27sq = (v + 27.0) * (v + 27.0);
    const double r30 = rcp(1.0 + EXP_3 * E10);                    // exp((V + 30)/10)
    const double rtau_f = rcp(1102.5 * exp(v27sq * (-1.0 / 225.0)) + 200.0 * rcp(1.0 + EXP_1P3 * I10) +
                              180.0 * r30 + 20.0);                // exp((13 - V)/10)
    const double f2_inf = 0.67 * rcp(1.0 + EXP_5 * E7) + 0.33;    // exp((V + 35)/7)
    const double rtau_f2 = rcp(562.0 * exp(v27sq * (-1.0 / 240.0)) + 31.0 * rcp(1.0 + EXP_2P5 * I10) +
                               80.0 * r30);                       // exp((25 - V)/10)
    const double rc2 = rcp(1.0 + (vCass * 20.0) * (vCass * 20.0));  // 1/(1 + (Ca_ss/0.05)^2)
    const double fCass_inf = 0.6 * rc2 + 0.4;
    const double rtau_fCass = rcp(80.0 * rc2 + 2.0);

    // ---- transient outward (.ode:273-284) ---------------------------------------------------------------
    const double gto = p[g_to] * vr * vs;
    const double i_to = gto * u;
    const double s_inf = rcp(1.0 + EXP_4 * E5);                   // exp((V + 20)/5)
    const double rtau_s = rcp(85.0 * exp((v + 45.0) * (v + 45.0) * (-1.0 / 320.0)) +
                              5.0 * rcp(1.0 + EXP_M4 * E5) + 3.0);  // exp((V - 20)/5)
    const double r_inf = rcp(1.0 + EXP_20_6 * I6);                // exp((20 - V)/6)
    const double rtau_r = rcp(9.5 * exp((v + 40.0) * (v + 40.0) * (-1.0 / 1800.0)) + 0.8);

    // ---- pumps and exchanger (.ode:286-296) ---------------------------------------------------------------
    const double rNaK = rcp(1.0 + 0.1245 * e5 + 0.0353 * e6);
    const double rNaKm = rcp(vNai + p[K_mNa]);
    const double i_NaK = q.NaK_B * vNai * rNaKm * rNaK;
    const double di_NaK_dV = i_NaK * q.FRT * (0.01245 * e5 + 0.0353 * e6) * rNaK;
    const double di_NaK_dNai = q.NaK_B * p[K_mNa] * rNaKm * rNaKm * rNaK;
    const double Nai3 = vNai * vNai * vNai;
    const double A1 = Nai3 * p[Ca_o], A2 = q.A2c * vCai;
    const double rS = rcp(1.0 + p[K_sat] * eg1);
    const double NNaCa = eg * A1 - eg1 * A2;
    const double kS = q.kNaCaQ * rS;
    const double i_NaCa = kS * NNaCa;
    const double di_NaCa_dV =
        kS * q.FRT * ((p[gamma] * eg * A1 - q.gm1 * eg1 * A2) - NNaCa * p[K_sat] * q.gm1 * eg1 * rS);
    const double di_NaCa_dNai = kS * eg * 3.0 * (vNai * vNai) * p[Ca_o];
    const double di_NaCa_dCai = -kS * eg1 * q.A2c;
    const double rpCa = rcp(vCai + p[K_pCa]);
    const double i_p_Ca = p[g_pCa] * vCai * rpCa;
    const double di_pCa_dCai = p[g_pCa] * p[K_pCa] * rpCa * rpCa;
    const double epK = exp((25.0 - v) * (1.0 / 5.98));
    const double rpK = rcp(1.0 + epK);
    const double i_p_K = p[g_pK] * u * rpK;
    const double di_pK_du = p[g_pK] * rpK;
    const double di_pK_dVgate = p[g_pK] * u * epK * (1.0 / 5.98) * rpK * rpK;

    // ---- calcium dynamics (.ode:298-316) ------------------------------------------------------------------
    const double qup = q.Kup2 * rCai * rCai;
    const double rup = rcp(1.0 + qup);
    const double i_up = p[Vmax_up] * rup;
    const double di_up_dCai = i_up * 2.0 * qup * rCai * rup;
    const double i_leak = p[V_leak] * (vCaSR - vCai);
    const double i_xfer = p[V_xfer] * (vCass - vCai);
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
    const double di_rel_dCaSR = p[V_rel] * (dO_dk1 * dk1 * dsrss + O);
    const double di_rel_dCass = p[V_rel] * (dO_dCass * dsrss - O);

    const double T_i = -(i_b_Ca + i_p_Ca - 2.0 * i_NaCa) * q.c1 + (i_leak - i_up) * q.c2 + i_xfer;
    const double dT_i = -(p[g_bca] * q.halfRTF * rCai + di_pCa_dCai - 2.0 * di_NaCa_dCai) * q.c1 +
                        (-p[V_leak] - di_up_dCai) * q.c2 - p[V_xfer];
    const double rbc = rcp(vCai + p[K_buf_c]);
    const double gci = q.BKc * rbc * rbc;
    const double Fr_i = rcp(1.0 + gci);
    const double dCa_i_dt = T_i * Fr_i;
    const double J_Cai = dT_i * Fr_i + T_i * (Fr_i * Fr_i * 2.0 * gci * rbc);

    const double T_sr = i_up - (i_rel + i_leak);
    const double dT_sr = -(di_rel_dCaSR + p[V_leak]);
    const double rbsr = rcp(vCaSR + p[K_buf_sr]);
    const double gsr = q.BKsr * rbsr * rbsr;
    const double Fr_sr = rcp(1.0 + gsr);
    const double dCa_SR_dt = T_sr * Fr_sr;
    const double J_CaSR = dT_sr * Fr_sr + T_sr * (Fr_sr * Fr_sr * 2.0 * gsr * rbsr);

    const double T_ss = -i_CaL * q.c3 + i_rel * q.c4 - i_xfer * q.c5;
    const double dT_ss = -di_CaL_dCass * q.c3 + di_rel_dCass * q.c4 - p[V_xfer] * q.c5;
    const double rbss = rcp(vCass + p[K_buf_ss]);
    const double gss = q.BKss * rbss * rbss;
    const double Fr_ss = rcp(1.0 + gss);
    const double dCa_ss_dt = T_ss * Fr_ss;
    const double J_Cass = dT_ss * Fr_ss + T_ss * (Fr_ss * Fr_ss * 2.0 * gss * rbss);

    const double dR_dt = -k2 * vCass * vR + p[k4] * (1.0 - vR);
    const double J_R = -vCass * k2 - p[k4];

    // ---- sodium, membrane, potassium (.ode:318-322) ----------------------------------------------------------
    const double dNa_i_dt = -(i_Na + i_b_Na + 3.0 * i_NaK + 3.0 * i_NaCa) * q.cVF;
    const double J_Nai = -((gNa + p[g_bna]) * q.RTF * rNai + 3.0 * di_NaK_dNai + 3.0 * di_NaCa_dNai) * q.cVF;
    const double tmod = t - floor(t / p[stim_period]) * p[stim_period];
    const double i_Stim =
        (tmod >= p[stim_start] && tmod <= p[stim_start] + p[stim_duration]) ? p[stim_amplitude] : 0.0;
    const double dV_dt = -(i_K1 + i_to + i_Kr + i_Ks + i_CaL + i_NaK + i_Na + i_b_Na + i_NaCa + i_b_Ca +
                           i_p_K + i_p_Ca + i_Stim);
    const double sum_du = di_K1_du + gto + gKr + di_pK_du;  // d/du of the currents driven by u = V - E_K
    const double J_V = -(sum_du + di_pK_dVgate + gKs + di_CaL_dV + di_NaK_dV + gNa + p[g_bna] +
                         di_NaCa_dV + p[g_bca]);
    const double dK_i_dt = -(i_K1 + i_to + i_Kr + i_Ks + i_p_K + i_Stim - 2.0 * i_NaK) * q.cVF;
    // dE_K/dK_i = -RTF/K_i, dE_Ks/dK_i = -RTF/(K_i + P_kna Na_i); currents depend on K_i only through them
    const double J_Ki = -(sum_du * q.RTF * rKi + gKs * q.RTF * rKs) * q.cVF;

    y[Xr1] = gate(vXr1, xr1_inf, rtau_xr1, dt);
    y[Xr2] = gate(vXr2, xr2_inf, rtau_xr2, dt);
    y[Xs] = gate(vXs, xs_inf, rtau_xs, dt);
    y[m] = gate(vm, m_inf, rtau_m, dt);
    y[h] = gate(vh, h_inf, ah_bh, dt);
    y[j] = gate(vj, h_inf, aj_bj, dt);
    y[d] = gate(vd, d_inf, rtau_d, dt);
    y[f] = gate(vf, f_inf, rtau_f, dt);
    y[f2] = gate(vf2, f2_inf, rtau_f2, dt);
    y[fCass] = gate(vfCass, fCass_inf, rtau_fCass, dt);
    y[s] = gate(vs, s_inf, rtau_s, dt);
    y[r] = gate(vr, r_inf, rtau_r, dt);
    y[R_prime] = grl1(vR, dR_dt, J_R, dt);
    y[Ca_i] = grl1(vCai, dCa_i_dt, J_Cai, dt);
    y[Ca_SR] = grl1(vCaSR, dCa_SR_dt, J_CaSR, dt);
    y[Ca_ss] = grl1(vCass, dCa_ss_dt, J_Cass, dt);
    y[Na_i] = grl1(vNai, dNa_i_dt, J_Nai, dt);
    y[V] = grl1(v, dV_dt, J_V, dt);
    y[K_i] = grl1(vKi, dK_i_dt, J_Ki, dt);
  }
};
