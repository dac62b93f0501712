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
  static Derived derive(const double*) { return {}; }
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
  static Derived derive(const double*) { return {}; }
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
  static Derived derive(const double*) { return {}; }
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
//   y_i += f_i (exp(J_i dt) - 1) / J_i  if |J_i| > 1e-8 else dt f_i,  J_i = d f_i / d y_i of the
//   derivative expression as written (non-zero for the 12 gates and R_prime), forward Euler for
//   Ca_i, Ca_SR, Ca_ss, Na_i, V, K_i.
// State / parameter order = order of appearance in the .ode file.
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
  struct Derived {};
  static Derived derive(const double*) { return {}; }

  __device__ static __forceinline__ double grl1(double y, double fy, double J, double dt) {
    return y + ((fabs(J) > 1e-8) ? fy * (exp(J * dt) - 1.0) / J : fy * dt);
  }

  __device__ static __forceinline__ void step(double* y, const double* p, const Derived&, double t,
                                              double dt) {
    const double vXr1 = y[Xr1], vXr2 = y[Xr2], vXs = y[Xs], vm = y[m], vh = y[h], vj = y[j],
                 vd = y[d], vf = y[f], vf2 = y[f2], vfCass = y[fCass], vs = y[s], vr = y[r],
                 vR_prime = y[R_prime], vCa_i = y[Ca_i], vCa_SR = y[Ca_SR], vCa_ss = y[Ca_ss],
                 vNa_i = y[Na_i], v = y[V], vK_i = y[K_i];

    const double RTF = p[R] * p[T] / p[F];
    const double FRT = p[F] / (p[R] * p[T]);

    // Reversal potentials (.ode:174-178)
    const double E_Na = RTF * log(p[Na_o] / vNa_i);
    const double E_K = RTF * log(p[K_o] / vK_i);
    const double E_Ks = RTF * log((p[K_o] + p[P_kna] * p[Na_o]) / (vK_i + p[P_kna] * vNa_i));
    const double E_Ca = 0.5 * p[R] * p[T] / p[F] * log(p[Ca_o] / vCa_i);

    // Inward rectifier (.ode:180-184)
    const double alpha_K1 = 0.1 / (1.0 + exp(0.06 * (v - E_K - 200.0)));
    const double beta_K1 = (3.0 * exp(0.0002 * (v - E_K + 100.0)) + exp(0.1 * (v - E_K - 10.0))) /
                           (1.0 + exp(-0.5 * (v - E_K)));
    const double xK1_inf = alpha_K1 / (alpha_K1 + beta_K1);
    const double sqrtKo = sqrt(p[K_o] / 5.4);
    const double i_K1 = p[g_K1] * xK1_inf * sqrtKo * (v - E_K);

    // Rapid delayed rectifier (.ode:186-201)
    const double i_Kr = p[g_Kr] * sqrtKo * vXr1 * vXr2 * (v - E_K);
    const double xr1_inf = 1.0 / (1.0 + exp((-26.0 - v) / 7.0));
    const double alpha_xr1 = 450.0 / (1.0 + exp((-45.0 - v) / 10.0));
    const double beta_xr1 = 6.0 / (1.0 + exp((v + 30.0) / 11.5));
    const double tau_xr1 = 1.0 * alpha_xr1 * beta_xr1;
    const double dXr1_dt = (xr1_inf - vXr1) / tau_xr1;
    const double xr2_inf = 1.0 / (1.0 + exp((v + 88.0) / 24.0));
    const double alpha_xr2 = 3.0 / (1.0 + exp((-60.0 - v) / 20.0));
    const double beta_xr2 = 1.12 / (1.0 + exp((v - 60.0) / 20.0));
    const double tau_xr2 = 1.0 * alpha_xr2 * beta_xr2;
    const double dXr2_dt = (xr2_inf - vXr2) / tau_xr2;

    // Slow delayed rectifier (.ode:203-211)
    const double i_Ks = p[g_Ks] * (vXs * vXs) * (v - E_Ks);
    const double xs_inf = 1.0 / (1.0 + exp((-5.0 - v) / 14.0));
    const double alpha_xs = 1400.0 / sqrt(1.0 + exp((5.0 - v) / 6.0));
    const double beta_xs = 1.0 / (1.0 + exp((v - 35.0) / 15.0));
    const double tau_xs = 1.0 * alpha_xs * beta_xs + 80.0;
    const double dXs_dt = (xs_inf - vXs) / tau_xs;

    // Fast sodium (.ode:213-235)
    const double i_Na = p[g_Na] * (vm * vm * vm) * vh * vj * (v - E_Na);
    const double em = 1.0 + exp((-56.86 - v) / 9.03);
    const double m_inf = 1.0 / (em * em);
    const double alpha_m = 1.0 / (1.0 + exp((-60.0 - v) / 5.0));
    const double beta_m = 0.1 / (1.0 + exp((v + 35.0) / 5.0)) + 0.1 / (1.0 + exp((v - 50.0) / 200.0));
    const double tau_m = 1.0 * alpha_m * beta_m;
    const double dm_dt = (m_inf - vm) / tau_m;
    const double eh = 1.0 + exp((v + 71.55) / 7.43);
    const double h_inf = 1.0 / (eh * eh);
    double alpha_h, beta_h, alpha_j, beta_j;
    if (v < -40.0) {
      alpha_h = 0.057 * exp(-(v + 80.0) / 6.8);
      beta_h = 2.7 * exp(0.079 * v) + 310000.0 * exp(0.3485 * v);
      alpha_j = (-25428.0 * exp(0.2444 * v) - 6.948e-6 * exp(-0.04391 * v)) * (v + 37.78) / 1.0 /
                (1.0 + exp(0.311 * (v + 79.23)));
      beta_j = 0.02424 * exp(-0.01052 * v) / (1.0 + exp(-0.1378 * (v + 40.14)));
    } else {
      alpha_h = 0.0;
      beta_h = 0.77 / (0.13 * (1.0 + exp((v + 10.66) / -11.1)));
      alpha_j = 0.0;
      beta_j = 0.6 * exp(0.057 * v) / (1.0 + exp(-0.1 * (v + 32.0)));
    }
    const double tau_h = 1.0 / (alpha_h + beta_h);
    const double dh_dt = (h_inf - vh) / tau_h;
    const double j_inf = h_inf;
    const double tau_j = 1.0 / (alpha_j + beta_j);
    const double dj_dt = (j_inf - vj) / tau_j;

    // Sodium background (.ode:237-238)
    const double i_b_Na = p[g_bna] * (v - E_Na);

    // L-type calcium (.ode:240-268)
    const double eCaL = exp(2.0 * (v - 15.0) * p[F] / (p[R] * p[T]));
    const double i_CaL = p[g_CaL] * vd * vf * vf2 * vfCass * 4.0 * (v - 15.0) * (p[F] * p[F]) /
                         (p[R] * p[T]) * (0.25 * vCa_ss * eCaL - p[Ca_o]) / (eCaL - 1.0);
    const double d_inf = 1.0 / (1.0 + exp((-8.0 - v) / 7.5));
    const double alpha_d = 1.4 / (1.0 + exp((-35.0 - v) / 13.0)) + 0.25;
    const double beta_d = 1.4 / (1.0 + exp((v + 5.0) / 5.0));
    const double gamma_d = 1.0 / (1.0 + exp((50.0 - v) / 20.0));
    const double tau_d = 1.0 * alpha_d * beta_d + gamma_d;
    const double dd_dt = (d_inf - vd) / tau_d;
    const double f_inf = 1.0 / (1.0 + exp((v + 20.0) / 7.0));
    const double v27sq = (v + 27.0) * (v + 27.0);
    const double e30_10 = exp((v + 30.0) / 10.0);
    const double tau_f = 1102.5 * exp(-v27sq / 225.0) + 200.0 / (1.0 + exp((13.0 - v) / 10.0)) +
                         180.0 / (1.0 + e30_10) + 20.0;
    const double df_dt = (f_inf - vf) / tau_f;
    const double f2_inf = 0.67 / (1.0 + exp((v + 35.0) / 7.0)) + 0.33;
    const double tau_f2 = 562.0 * exp(-v27sq / 240.0) + 31.0 / (1.0 + exp((25.0 - v) / 10.0)) +
                          80.0 / (1.0 + e30_10);
    const double df2_dt = (f2_inf - vf2) / tau_f2;
    const double cass2 = (vCa_ss / 0.05) * (vCa_ss / 0.05);
    const double fCass_inf = 0.6 / (1.0 + cass2) + 0.4;
    const double tau_fCass = 80.0 / (1.0 + cass2) + 2.0;
    const double dfCass_dt = (fCass_inf - vfCass) / tau_fCass;

    // Calcium background, transient outward (.ode:270-284)
    const double i_b_Ca = p[g_bca] * (v - E_Ca);
    const double i_to = p[g_to] * vr * vs * (v - E_K);
    const double s_inf = 1.0 / (1.0 + exp((v + 20.0) / 5.0));
    const double tau_s = 85.0 * exp(-((v + 45.0) * (v + 45.0)) / 320.0) +
                         5.0 / (1.0 + exp((v - 20.0) / 5.0)) + 3.0;
    const double ds_dt = (s_inf - vs) / tau_s;
    const double r_inf = 1.0 / (1.0 + exp((20.0 - v) / 6.0));
    const double tau_r = 9.5 * exp(-((v + 40.0) * (v + 40.0)) / 1800.0) + 0.8;
    const double dr_dt = (r_inf - vr) / tau_r;

    // Pumps and exchanger (.ode:286-296)
    const double i_NaK = p[P_NaK] * p[K_o] / (p[K_o] + p[K_mk]) * vNa_i / (vNa_i + p[K_mNa]) /
                         (1.0 + 0.1245 * exp(-0.1 * v * p[F] / (p[R] * p[T])) +
                          0.0353 * exp(-v * p[F] / (p[R] * p[T])));
    const double eg = exp(p[gamma] * v * p[F] / (p[R] * p[T]));
    const double eg1 = exp((p[gamma] - 1.0) * v * p[F] / (p[R] * p[T]));
    const double Nao3 = p[Na_o] * p[Na_o] * p[Na_o];
    const double i_NaCa =
        p[K_NaCa] * (eg * (vNa_i * vNa_i * vNa_i) * p[Ca_o] - eg1 * Nao3 * vCa_i * p[alpha]) /
        ((p[Km_Nai] * p[Km_Nai] * p[Km_Nai] + Nao3) * (p[Km_Ca] + p[Ca_o]) * (1.0 + p[K_sat] * eg1));
    const double i_p_Ca = p[g_pCa] * vCa_i / (vCa_i + p[K_pCa]);
    const double i_p_K = p[g_pK] * (v - E_K) / (1.0 + exp((25.0 - v) / 5.98));

    // Calcium dynamics (.ode:298-316)
    const double i_up = p[Vmax_up] / (1.0 + (p[K_up] * p[K_up]) / (vCa_i * vCa_i));
    const double i_leak = p[V_leak] * (vCa_SR - vCa_i);
    const double i_xfer = p[V_xfer] * (vCa_ss - vCa_i);
    const double ecsr = p[EC] / vCa_SR;
    const double kcasr = p[max_sr] - (p[max_sr] - p[min_sr]) / (1.0 + ecsr * ecsr);
    const double ddt_Ca_i_total = -(i_b_Ca + i_p_Ca - 2.0 * i_NaCa) * p[Cm] / (2.0 * p[V_c] * p[F]) +
                                  (i_leak - i_up) * p[V_sr] / p[V_c] + i_xfer;
    const double bci = vCa_i + p[K_buf_c];
    const double f_JCa_i_free = 1.0 / (1.0 + p[Buf_c] * p[K_buf_c] / (bci * bci));
    const double bsr = vCa_SR + p[K_buf_sr];
    const double f_JCa_sr_free = 1.0 / (1.0 + p[Buf_sr] * p[K_buf_sr] / (bsr * bsr));
    const double bss = vCa_ss + p[K_buf_ss];
    const double f_JCa_ss_free = 1.0 / (1.0 + p[Buf_ss] * p[K_buf_ss] / (bss * bss));
    const double dCa_i_dt = ddt_Ca_i_total * f_JCa_i_free;
    const double k1 = p[k1_prime] / kcasr;
    const double k2 = p[k2_prime] * kcasr;
    const double css2 = vCa_ss * vCa_ss;
    const double O = k1 * css2 * vR_prime / (p[k3] + k1 * css2);
    const double dR_prime_dt = -k2 * vCa_ss * vR_prime + p[k4] * (1.0 - vR_prime);
    const double i_rel = p[V_rel] * O * (vCa_SR - vCa_ss);
    const double ddt_Ca_sr_total = i_up - (i_rel + i_leak);
    const double ddt_Ca_ss_total = -i_CaL * p[Cm] / (2.0 * p[V_ss] * p[F]) +
                                   i_rel * p[V_sr] / p[V_ss] - i_xfer * p[V_c] / p[V_ss];
    const double dCa_SR_dt = ddt_Ca_sr_total * f_JCa_sr_free;
    const double dCa_ss_dt = ddt_Ca_ss_total * f_JCa_ss_free;

    // Sodium, membrane, potassium (.ode:318-322)
    const double dNa_i_dt = -(i_Na + i_b_Na + 3.0 * i_NaK + 3.0 * i_NaCa) / (p[V_c] * p[F]) * p[Cm];
    const double tmod = t - floor(t / p[stim_period]) * p[stim_period];
    const double i_Stim =
        (tmod >= p[stim_start] && tmod <= p[stim_start] + p[stim_duration]) ? p[stim_amplitude] : 0.0;
    const double dV_dt = -(i_K1 + i_to + i_Kr + i_Ks + i_CaL + i_NaK + i_Na + i_b_Na + i_NaCa +
                           i_b_Ca + i_p_K + i_p_Ca + i_Stim);
    const double dK_i_dt =
        -(i_K1 + i_to + i_Kr + i_Ks + i_p_K + i_Stim - 2.0 * i_NaK) / (p[V_c] * p[F]) * p[Cm];
    (void)FRT;

    y[Xr1] = grl1(vXr1, dXr1_dt, -1.0 / tau_xr1, dt);
    y[Xr2] = grl1(vXr2, dXr2_dt, -1.0 / tau_xr2, dt);
    y[Xs] = grl1(vXs, dXs_dt, -1.0 / tau_xs, dt);
    y[m] = grl1(vm, dm_dt, -1.0 / tau_m, dt);
    y[h] = grl1(vh, dh_dt, -1.0 / tau_h, dt);
    y[j] = grl1(vj, dj_dt, -1.0 / tau_j, dt);
    y[d] = grl1(vd, dd_dt, -1.0 / tau_d, dt);
    y[f] = grl1(vf, df_dt, -1.0 / tau_f, dt);
    y[f2] = grl1(vf2, df2_dt, -1.0 / tau_f2, dt);
    y[fCass] = grl1(vfCass, dfCass_dt, -1.0 / tau_fCass, dt);
    y[s] = grl1(vs, ds_dt, -1.0 / tau_s, dt);
    y[r] = grl1(vr, dr_dt, -1.0 / tau_r, dt);
    y[R_prime] = grl1(vR_prime, dR_prime_dt, -vCa_ss * k2 - p[k4], dt);
    y[Ca_i] = vCa_i + dCa_i_dt * dt;
    y[Ca_SR] = vCa_SR + dCa_SR_dt * dt;
    y[Ca_ss] = vCa_ss + dCa_ss_dt * dt;
    y[Na_i] = vNa_i + dNa_i_dt * dt;
    y[V] = v + dV_dt * dt;
    y[K_i] = vK_i + dK_i_dt * dt;
  }
};
