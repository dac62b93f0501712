/* ORACLE (test infrastructure only -- never linked into or called by the product path).
 *
 * Plain C restatement of the hot path for the CPU baseline (bench.py cpu_baseline, kind "port") and as
 * a second checker: TP06 generalized Rush-Larsen step (model spec:
 * odes/tentusscher_panfilov_2006/tentusscher_panfilov_2006_epi_cell.ode:36-322 of the reference; scheme:
 * gotranx GRL1 with total self-derivatives, see oracle/ionic.py) and the theta-rule diffusion step
 * (src/beat/monodomain_model.py:68-98, src/beat/base_model.py:196-236) as a 15-point stencil + Jacobi-PCG.
 * Written literally (libm exp/log, IEEE division, no shared sub-expressions); validated against
 * oracle/ionic.py and oracle/fem.py by tests/test_oracle_c.py.  OpenMP over nodes.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

enum { Xr1, Xr2, Xs, m_, h_, j_, d_, f_, f2_, fCass, s_, r_, R_prime, Ca_i, Ca_SR, Ca_ss, Na_i, V_, K_i, NS };
enum { P_kna, g_K1, g_Kr, g_Ks, g_Na, g_bna, g_CaL, g_bca, g_to, P_NaK, K_mk, K_mNa, K_NaCa, K_sat, alpha_, gamma_,
       Km_Ca, Km_Nai, g_pCa, K_pCa, g_pK, Ca_o, k1_prime, k2_prime, k3, k4, EC, max_sr, min_sr, V_rel, V_xfer, K_up,
       V_leak, Vmax_up, Buf_c, K_buf_c, Buf_sr, K_buf_sr, Buf_ss, K_buf_ss, V_sr, V_ss, Na_o, R_, T_, F_, Cm, V_c,
       stim_start, stim_period, stim_duration, stim_amplitude, K_o, NP };

void oracle_set_num_threads(int n) {
  if (n > 0) omp_set_num_threads(n);
}

int oracle_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

static double grl1(double y, double f, double J, double dt) {
  return y + (fabs(J) > 1e-8 ? f * (exp(J * dt) - 1.0) / J : f * dt);
}

static void tp06_node(double* y, const double* p, double t, double dt) {
  const double vXr1 = y[Xr1], vXr2 = y[Xr2], vXs = y[Xs], m = y[m_], h = y[h_], j = y[j_], d = y[d_], f = y[f_],
               f2 = y[f2_], fC = y[fCass], s = y[s_], r = y[r_], Rp = y[R_prime], Cai = y[Ca_i], CaSR = y[Ca_SR],
               Cass = y[Ca_ss], Nai = y[Na_i], V = y[V_], Ki = y[K_i];
  const double RTF = p[R_] * p[T_] / p[F_], FRT = p[F_] / (p[R_] * p[T_]);
  const double E_Na = RTF * log(p[Na_o] / Nai), E_K = RTF * log(p[K_o] / Ki);
  const double E_Ks = RTF * log((p[K_o] + p[P_kna] * p[Na_o]) / (Ki + p[P_kna] * Nai));
  const double E_Ca = 0.5 * RTF * log(p[Ca_o] / Cai);
  const double u = V - E_K, sq = sqrt(p[K_o] / 5.4);

  /* i_K1 and d/du */
  const double e1 = exp(0.06 * (u - 200)), e2 = exp(0.0002 * (u + 100)), e3 = exp(0.1 * (u - 10)), e4 = exp(-0.5 * u);
  const double aK1 = 0.1 / (1 + e1), daK1 = -0.006 * e1 / ((1 + e1) * (1 + e1));
  const double bK1 = (3 * e2 + e3) / (1 + e4);
  const double dbK1 = ((0.0006 * e2 + 0.1 * e3) * (1 + e4) + (3 * e2 + e3) * 0.5 * e4) / ((1 + e4) * (1 + e4));
  const double xK1 = aK1 / (aK1 + bK1), dxK1 = (daK1 * bK1 - aK1 * dbK1) / ((aK1 + bK1) * (aK1 + bK1));
  const double i_K1 = p[g_K1] * xK1 * sq * u, di_K1 = p[g_K1] * sq * (dxK1 * u + xK1);
  const double gKr = p[g_Kr] * sq * vXr1 * vXr2, i_Kr = gKr * u;
  const double gKs = p[g_Ks] * vXs * vXs, i_Ks = gKs * (V - E_Ks);
  const double gNa = p[g_Na] * m * m * m * h * j, i_Na = gNa * (V - E_Na);
  const double i_b_Na = p[g_bna] * (V - E_Na), i_b_Ca = p[g_bca] * (V - E_Ca);
  const double gto = p[g_to] * r * s, i_to = gto * u;
  const double epK = exp((25 - V) / 5.98), i_p_K = p[g_pK] * u / (1 + epK);
  const double di_pK_du = p[g_pK] / (1 + epK), di_pK_dVg = p[g_pK] * u * epK / (5.98 * (1 + epK) * (1 + epK));

  /* gates: steady states and time constants */
  const double xr1_inf = 1 / (1 + exp((-26 - V) / 7));
  const double tau_xr1 = 450 / (1 + exp((-45 - V) / 10)) * 6 / (1 + exp((V + 30) / 11.5));
  const double xr2_inf = 1 / (1 + exp((V + 88) / 24));
  const double tau_xr2 = 3 / (1 + exp((-60 - V) / 20)) * 1.12 / (1 + exp((V - 60) / 20));
  const double xs_inf = 1 / (1 + exp((-5 - V) / 14));
  const double tau_xs = 1400 / sqrt(1 + exp((5 - V) / 6)) * 1 / (1 + exp((V - 35) / 15)) + 80;
  const double m_inf = 1 / pow(1 + exp((-56.86 - V) / 9.03), 2);
  const double tau_m = 1 / (1 + exp((-60 - V) / 5)) * (0.1 / (1 + exp((V + 35) / 5)) + 0.1 / (1 + exp((V - 50) / 200)));
  const double h_inf = 1 / pow(1 + exp((V + 71.55) / 7.43), 2);
  double ah, bh, aj, bj;
  if (V < -40) {
    ah = 0.057 * exp(-(V + 80) / 6.8);
    bh = 2.7 * exp(0.079 * V) + 310000 * exp(0.3485 * V);
    aj = (-25428 * exp(0.2444 * V) - 6.948e-6 * exp(-0.04391 * V)) * (V + 37.78) / (1 + exp(0.311 * (V + 79.23)));
    bj = 0.02424 * exp(-0.01052 * V) / (1 + exp(-0.1378 * (V + 40.14)));
  } else {
    ah = 0;
    bh = 0.77 / (0.13 * (1 + exp((V + 10.66) / -11.1)));
    aj = 0;
    bj = 0.6 * exp(0.057 * V) / (1 + exp(-0.1 * (V + 32)));
  }
  const double d_inf = 1 / (1 + exp((-8 - V) / 7.5));
  const double tau_d = (1.4 / (1 + exp((-35 - V) / 13)) + 0.25) * (1.4 / (1 + exp((V + 5) / 5))) + 1 / (1 + exp((50 - V) / 20));
  const double f_inf = 1 / (1 + exp((V + 20) / 7));
  const double tau_f = 1102.5 * exp(-pow(V + 27, 2) / 225) + 200 / (1 + exp((13 - V) / 10)) + 180 / (1 + exp((V + 30) / 10)) + 20;
  const double f2_inf = 0.67 / (1 + exp((V + 35) / 7)) + 0.33;
  const double tau_f2 = 562 * exp(-pow(V + 27, 2) / 240) + 31 / (1 + exp((25 - V) / 10)) + 80 / (1 + exp((V + 30) / 10));
  const double c2 = pow(Cass / 0.05, 2);
  const double fC_inf = 0.6 / (1 + c2) + 0.4, tau_fC = 80 / (1 + c2) + 2;
  const double s_inf = 1 / (1 + exp((V + 20) / 5));
  const double tau_s = 85 * exp(-pow(V + 45, 2) / 320) + 5 / (1 + exp((V - 20) / 5)) + 3;
  const double r_inf = 1 / (1 + exp((20 - V) / 6));
  const double tau_r = 9.5 * exp(-pow(V + 40, 2) / 1800) + 0.8;

  /* exchanger, pumps, L-type */
  const double e5 = exp(-0.1 * V * FRT), e6 = exp(-V * FRT);
  const double Dn = 1 + 0.1245 * e5 + 0.0353 * e6;
  const double B = p[P_NaK] * p[K_o] / (p[K_o] + p[K_mk]);
  const double i_NaK = B * Nai / (Nai + p[K_mNa]) / Dn;
  const double di_NaK_dV = i_NaK * FRT * (0.01245 * e5 + 0.0353 * e6) / Dn;
  const double di_NaK_dNai = B * p[K_mNa] / ((Nai + p[K_mNa]) * (Nai + p[K_mNa])) / Dn;
  const double eg = exp(p[gamma_] * V * FRT), eg1 = exp((p[gamma_] - 1) * V * FRT);
  const double Nao3 = pow(p[Na_o], 3), A1 = pow(Nai, 3) * p[Ca_o], A2 = Nao3 * Cai * p[alpha_];
  const double Q = (pow(p[Km_Nai], 3) + Nao3) * (p[Km_Ca] + p[Ca_o]), S = 1 + p[K_sat] * eg1;
  const double Nn = eg * A1 - eg1 * A2;
  const double i_NaCa = p[K_NaCa] * Nn / (Q * S);
  const double dNn = FRT * (p[gamma_] * eg * A1 - (p[gamma_] - 1) * eg1 * A2), dS = p[K_sat] * (p[gamma_] - 1) * FRT * eg1;
  const double di_NaCa_dV = p[K_NaCa] / Q * (dNn * S - Nn * dS) / (S * S);
  const double di_NaCa_dNai = p[K_NaCa] * eg * 3 * Nai * Nai * p[Ca_o] / (Q * S);
  const double di_NaCa_dCai = -p[K_NaCa] * eg1 * Nao3 * p[alpha_] / (Q * S);
  const double i_p_Ca = p[g_pCa] * Cai / (Cai + p[K_pCa]);
  const double di_pCa = p[g_pCa] * p[K_pCa] / ((Cai + p[K_pCa]) * (Cai + p[K_pCa]));
  const double a2 = 2 * FRT, w = V - 15, eL = exp(a2 * w);
  const double CL = p[g_CaL] * d * f * f2 * fC * 4 * p[F_] * p[F_] / (p[R_] * p[T_]);
  const double NL = 0.25 * Cass * eL - p[Ca_o], DL = eL - 1;
  const double i_CaL = CL * w * NL / DL;
  const double di_CaL_dV = CL * (NL / DL + w * a2 * eL * (p[Ca_o] - 0.25 * Cass) / (DL * DL));
  const double di_CaL_dCass = CL * w * 0.25 * eL / DL;

  /* calcium handling */
  const double qup = p[K_up] * p[K_up] / (Cai * Cai), i_up = p[Vmax_up] / (1 + qup);
  const double di_up = i_up * 2 * qup / (Cai * (1 + qup));
  const double i_leak = p[V_leak] * (CaSR - Cai), i_xfer = p[V_xfer] * (Cass - Cai);
  const double z = pow(p[EC] / CaSR, 2), dsr = p[max_sr] - p[min_sr];
  const double kcasr = p[max_sr] - dsr / (1 + z), dkcasr = -2 * dsr * z / (CaSR * (1 + z) * (1 + z));
  const double k1 = p[k1_prime] / kcasr, dk1 = -k1 * dkcasr / kcasr, k2 = p[k2_prime] * kcasr;
  const double cs2 = Cass * Cass, den = p[k3] + k1 * cs2;
  const double O = k1 * cs2 * Rp / den, dO_dk1 = cs2 * Rp * p[k3] / (den * den), dO_dCass = 2 * Cass * k1 * Rp * p[k3] / (den * den);
  const double i_rel = p[V_rel] * O * (CaSR - Cass);
  const double c1 = p[Cm] / (2 * p[V_c] * p[F_]), cc2 = p[V_sr] / p[V_c], c3 = p[Cm] / (2 * p[V_ss] * p[F_]),
               c4 = p[V_sr] / p[V_ss], c5 = p[V_c] / p[V_ss], cVF = p[Cm] / (p[V_c] * p[F_]);
  const double Ti = -(i_b_Ca + i_p_Ca - 2 * i_NaCa) * c1 + (i_leak - i_up) * cc2 + i_xfer;
  const double dTi = -(p[g_bca] * 0.5 * RTF / Cai + di_pCa - 2 * di_NaCa_dCai) * c1 + (-p[V_leak] - di_up) * cc2 - p[V_xfer];
  const double gi = p[Buf_c] * p[K_buf_c] / pow(Cai + p[K_buf_c], 2), Fi = 1 / (1 + gi), dFi = Fi * Fi * 2 * gi / (Cai + p[K_buf_c]);
  const double Tsr = i_up - (i_rel + i_leak), dTsr = -(p[V_rel] * (dO_dk1 * dk1 * (CaSR - Cass) + O) + p[V_leak]);
  const double gs = p[Buf_sr] * p[K_buf_sr] / pow(CaSR + p[K_buf_sr], 2), Fs = 1 / (1 + gs), dFs = Fs * Fs * 2 * gs / (CaSR + p[K_buf_sr]);
  const double Tss = -i_CaL * c3 + i_rel * c4 - i_xfer * c5;
  const double dTss = -di_CaL_dCass * c3 + p[V_rel] * (dO_dCass * (CaSR - Cass) - O) * c4 - p[V_xfer] * c5;
  const double gss = p[Buf_ss] * p[K_buf_ss] / pow(Cass + p[K_buf_ss], 2), Fss = 1 / (1 + gss), dFss = Fss * Fss * 2 * gss / (Cass + p[K_buf_ss]);

  const double tmod = t - floor(t / p[stim_period]) * p[stim_period];
  const double i_Stim = (tmod >= p[stim_start] && tmod <= p[stim_start] + p[stim_duration]) ? p[stim_amplitude] : 0.0;
  const double I_K = i_K1 + i_to + i_Kr + i_Ks + i_p_K;
  const double I_tot = I_K + i_CaL + i_NaK + i_Na + i_b_Na + i_NaCa + i_b_Ca + i_p_Ca + i_Stim;
  const double sum_du = di_K1 + gto + gKr + di_pK_du;
  const double dI_dV = sum_du + di_pK_dVg + gKs + di_CaL_dV + di_NaK_dV + gNa + p[g_bna] + di_NaCa_dV + p[g_bca];

  y[Xr1] = grl1(vXr1, (xr1_inf - vXr1) / tau_xr1, -1 / tau_xr1, dt);
  y[Xr2] = grl1(vXr2, (xr2_inf - vXr2) / tau_xr2, -1 / tau_xr2, dt);
  y[Xs] = grl1(vXs, (xs_inf - vXs) / tau_xs, -1 / tau_xs, dt);
  y[m_] = grl1(m, (m_inf - m) / tau_m, -1 / tau_m, dt);
  y[h_] = grl1(h, (h_inf - h) * (ah + bh), -(ah + bh), dt);
  y[j_] = grl1(j, (h_inf - j) * (aj + bj), -(aj + bj), dt);
  y[d_] = grl1(d, (d_inf - d) / tau_d, -1 / tau_d, dt);
  y[f_] = grl1(f, (f_inf - f) / tau_f, -1 / tau_f, dt);
  y[f2_] = grl1(f2, (f2_inf - f2) / tau_f2, -1 / tau_f2, dt);
  y[fCass] = grl1(fC, (fC_inf - fC) / tau_fC, -1 / tau_fC, dt);
  y[s_] = grl1(s, (s_inf - s) / tau_s, -1 / tau_s, dt);
  y[r_] = grl1(r, (r_inf - r) / tau_r, -1 / tau_r, dt);
  y[R_prime] = grl1(Rp, -k2 * Cass * Rp + p[k4] * (1 - Rp), -k2 * Cass - p[k4], dt);
  y[Ca_i] = grl1(Cai, Ti * Fi, dTi * Fi + Ti * dFi, dt);
  y[Ca_SR] = grl1(CaSR, Tsr * Fs, dTsr * Fs + Tsr * dFs, dt);
  y[Ca_ss] = grl1(Cass, Tss * Fss, dTss * Fss + Tss * dFss, dt);
  y[Na_i] = grl1(Nai, -(i_Na + i_b_Na + 3 * i_NaK + 3 * i_NaCa) * cVF,
                 -((gNa + p[g_bna]) * RTF / Nai + 3 * di_NaK_dNai + 3 * di_NaCa_dNai) * cVF, dt);
  y[V_] = grl1(V, -I_tot, -dI_dV, dt);
  y[K_i] = grl1(Ki, -(I_K + i_Stim - 2 * i_NaK) * cVF,
                -(sum_du * RTF / Ki + gKs * RTF / (Ki + p[P_kna] * Nai)) * cVF, dt);
}

/* states: (19, ld) state-major, n nodes, uniform parameters p[53] */
void oracle_tp06_grl1(double* states, long n, long ld, const double* p, double t, double dt) {
#pragma omp parallel for schedule(static)
  for (long i = 0; i < n; ++i) {
    double y[NS];
    for (int k = 0; k < NS; ++k) y[k] = states[k * ld + i];
    tp06_node(y, p, t, dt);
    for (int k = 0; k < NS; ++k) states[k * ld + i] = y[k];
  }
}

/* ---- 15-point stencil operators on an nx*ny*nz box (physical boundaries on all faces) ---------------- */
static const int OFF[15][3] = {{0, 0, 0}, {1, 0, 0}, {-1, 0, 0}, {0, 1, 0}, {0, -1, 0}, {0, 0, 1}, {0, 0, -1}, {1, 1, 0},
                               {-1, -1, 0}, {0, 1, 1}, {0, -1, -1}, {1, 0, 1}, {-1, 0, -1}, {1, 1, 1}, {-1, -1, -1}};

static int axis_type(long i, long n) { return n == 1 ? 1 : (i == 0 ? 0 : (i == n - 1 ? 2 : 1)); }

/* y = tab x, tab = (27, 15) row-major */
void oracle_stencil_apply(const double* tab, long nx, long ny, long nz, const double* x, double* y) {
#pragma omp parallel for collapse(2) schedule(static)
  for (long iz = 0; iz < nz; ++iz)
    for (long iy = 0; iy < ny; ++iy) {
      const int tyz = 3 * axis_type(iy, ny) + 9 * axis_type(iz, nz);
      for (long ix = 0; ix < nx; ++ix) {
        const double* c = tab + 15 * (axis_type(ix, nx) + tyz);
        double sum = 0.0;
        for (int k = 0; k < 15; ++k) {
          if (c[k] == 0.0) continue;
          sum += c[k] * x[(ix + OFF[k][0]) + nx * ((iy + OFF[k][1]) + ny * (iz + OFF[k][2]))];
        }
        y[ix + nx * (iy + ny * iz)] = sum;
      }
    }
}

static double dot(const double* a, const double* b, long n) {
  double s = 0.0;
#pragma omp parallel for reduction(+ : s) schedule(static)
  for (long i = 0; i < n; ++i) s += a[i] * b[i];
  return s;
}

/* One theta-step, in place on v: A v_new = B v + dt * amp * w (w may be NULL), Jacobi-PCG from x0 = v to
 * ||r|| <= rtol ||b||.  A, B = (27,15) tables.  work: 5*n doubles.  Returns the iteration count. */
int oracle_theta_step(const double* A, const double* B, long nx, long ny, long nz, double* v, const double* w,
                      double amp_dt, double rtol, int max_it, double* work) {
  const long n = nx * ny * nz;
  double *b = work, *r = work + n, *p = work + 2 * n, *q = work + 3 * n, *dinv = work + 4 * n;
  oracle_stencil_apply(B, nx, ny, nz, v, b);
  if (w)
    for (long i = 0; i < n; ++i) b[i] += amp_dt * w[i];
  oracle_stencil_apply(A, nx, ny, nz, v, r);
#pragma omp parallel for collapse(2) schedule(static)
  for (long iz = 0; iz < nz; ++iz)
    for (long iy = 0; iy < ny; ++iy)
      for (long ix = 0; ix < nx; ++ix)
        dinv[ix + nx * (iy + ny * iz)] = 1.0 / A[15 * (axis_type(ix, nx) + 3 * axis_type(iy, ny) + 9 * axis_type(iz, nz))];
#pragma omp parallel for schedule(static)
  for (long i = 0; i < n; ++i) {
    r[i] = b[i] - r[i];
    p[i] = dinv[i] * r[i];
  }
  const double tol2 = rtol * rtol * dot(b, b, n);
  double rz = dot(r, p, n), rr = dot(r, r, n);
  int it = 0;
  while (rr > tol2 && it < max_it) {
    oracle_stencil_apply(A, nx, ny, nz, p, q);
    const double alpha = rz / dot(p, q, n);
    double rzn = 0.0, rrn = 0.0;
#pragma omp parallel for reduction(+ : rzn, rrn) schedule(static)
    for (long i = 0; i < n; ++i) {
      v[i] += alpha * p[i];
      r[i] -= alpha * q[i];
      rzn += r[i] * dinv[i] * r[i];
      rrn += r[i] * r[i];
    }
    const double beta = rzn / rz;
#pragma omp parallel for schedule(static)
    for (long i = 0; i < n; ++i) p[i] = dinv[i] * r[i] + beta * p[i];
    rz = rzn;
    rr = rrn;
    ++it;
  }
  return it;
}
