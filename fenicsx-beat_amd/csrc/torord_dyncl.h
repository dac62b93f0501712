// ToR-ORd-dynCl (Tomek, Rodriguez et al. 2019/2020 with dynamic chloride), 45 states, first-order generalized
// Rush-Larsen -- hand-organised kernel.
// Specification: odes/torord/ToRORd_dynCl_endo.ode (states :1-74, parameters :76-275, expressions :277-633), as
// advanced by the reference's ventricular demos (demos/biv_endocardial.py:124-173, one parameter set per cell type).
// Scheme: gotranx `generalized_rush_larsen` (see Tp06Grl1 in ionic_models.h and oracle/ionic.py):
//   y_i += f_i (exp(J_i dt) - 1) / J_i  if |J_i| > 1e-8 else dt f_i,   J_i = total d f_i / d y_i.
//
// Organisation (what the generated kernel of round 1 could not have: it kept 192 doubles live and ran at one wave
// per SIMD with spills):
//  * the step is a sequence of fenced BLOCKS, one per current / flux family.  A block loads the gate states it owns,
//    computes its current together with ONLY the partial derivatives some state's self-derivative needs, adds both to
//    running sums and stores its gates; its intermediates die with the block.
//  * running sums: for the membrane potential and each of the nine ion-concentration states the total current it
//    sees and that current's derivative w.r.t. the state itself (20 doubles), e.g.
//        d nai / dt = -Acap/(F vmyo) * I_nai + JdiffNa vss/vmyo,   I_nai = INa + INaL + INab + ICaNa_i + 3 INaK + 3 INaCa_i.
//  * derivatives inside a block are taken by forward-mode dual numbers with a COMPILE-TIME sparsity mask (Du<M>): a
//    quantity carries a tangent only for the directions it depends on, so the compiler emits exactly the chain-rule
//    terms that are structurally non-zero and the block reads like the specification.
//  * parameter-only sub-expressions, including every celltype switch, are evaluated once per launch (Derived).
//  * exponentials with a common slope share one exp(); the total derivatives through the Debye-Hueckel activity
//    coefficients and through the CaMK-dependent phosphorylation fraction are kept where a state sees them
//    (cai, nai, ki, cass, nass, kss), as the reference's fully resolved self-derivative has them.
//  * (round 3, profiles/r03_ode_probes.md: 4631 -> 3707 executed VALU instructions per node) there is no scalar fp64 unit,
//    so EVERY product, quotient and reciprocal of parameters alone lives in Derived; reciprocals are batched (one
//    v_rcp_f64 for two to four of them), a rate 1/(c + 1/s) is formed as s/(c s + 1), 1/exp(x) as exp(-x); exponentials
//    of slope 1/20, 1/10, 1/5 mV come from one exp(v/20); the GRL1 increment f (exp(J dt) - 1)/J of a non-gate state is
//    f dt phi(J dt) with phi by its Taylor polynomial when |J dt| <= 1/16 (per lane), and a gate whose time constant has
//    a floor takes the polynomial behind a uniform branch on dt * bound; a * b + c is contracted to fma inside this header
//    (the pragma below; the library as a whole is compiled with contraction off).
// Compiles for the host too (tests/test_torord_host.py builds it with g++ against the golden fixture): everything it
// needs from the device side comes through BEAT_HD / the FM template parameter / beat_rcp / beat_guard / BEAT_TFENCE.
#pragma once

#include <cmath>

#ifndef BEAT_TORORD_HOST_TEST
#include "ionic_models.h"
#define BEAT_HD __host__ __device__
#define BEAT_DV __device__ __forceinline__
#define BEAT_TFENCE() __builtin_amdgcn_sched_barrier(0)
// Pins a running sum at this point of the program: the scheduling fences order only instructions with side effects,
// and a block whose results merely flow into sums that are consumed at the very end (the pumps, the GHK fluxes) would
// otherwise be emitted down there, where everything else is live too.
#define BEAT_PIN(x) asm volatile("" : "+v"(x))
#define BEAT_SCONST(c) beat_sconst(c)
#endif

#if defined(__clang__) && !defined(BEAT_TORORD_NO_CONTRACT)
#pragma clang fp contract(fast)
#endif

namespace torord_detail {

// value + up to three tangents; bit k of M set <=> tangent k is structurally non-zero
template <unsigned M>
struct Du {
  double v;
  double d[3];
};
template <unsigned M>
BEAT_DV Du<M> mk(double v, double d0 = 0.0, double d1 = 0.0, double d2 = 0.0) {
  Du<M> r;
  r.v = v;
  r.d[0] = d0;
  r.d[1] = d1;
  r.d[2] = d2;
  return r;
}
#define BEAT_DU_FOR(k, MASK) \
  _Pragma("unroll") for (int k = 0; k < 3; ++k) if (((MASK) >> k) & 1u)

template <unsigned A, unsigned B>
BEAT_DV Du<A | B> operator+(const Du<A>& a, const Du<B>& b) {
  Du<A | B> r;
  r.v = a.v + b.v;
  BEAT_DU_FOR(k, A | B) r.d[k] = ((A >> k) & 1u) ? (((B >> k) & 1u) ? a.d[k] + b.d[k] : a.d[k]) : b.d[k];
  return r;
}
template <unsigned A, unsigned B>
BEAT_DV Du<A | B> operator-(const Du<A>& a, const Du<B>& b) {
  Du<A | B> r;
  r.v = a.v - b.v;
  BEAT_DU_FOR(k, A | B) r.d[k] = ((A >> k) & 1u) ? (((B >> k) & 1u) ? a.d[k] - b.d[k] : a.d[k]) : -b.d[k];
  return r;
}
template <unsigned A, unsigned B>
BEAT_DV Du<A | B> operator*(const Du<A>& a, const Du<B>& b) {
  Du<A | B> r;
  r.v = a.v * b.v;
  BEAT_DU_FOR(k, A | B) {
    if (((A >> k) & 1u) && ((B >> k) & 1u))
      r.d[k] = fma(a.d[k], b.v, a.v * b.d[k]);
    else if ((A >> k) & 1u)
      r.d[k] = a.d[k] * b.v;
    else
      r.d[k] = a.v * b.d[k];
  }
  return r;
}
template <unsigned A>
BEAT_DV Du<A> operator+(const Du<A>& a, double b) {
  Du<A> r = a;
  r.v = a.v + b;
  return r;
}
template <unsigned A>
BEAT_DV Du<A> operator+(double b, const Du<A>& a) {
  return a + b;
}
template <unsigned A>
BEAT_DV Du<A> operator-(const Du<A>& a, double b) {
  Du<A> r = a;
  r.v = a.v - b;
  return r;
}
template <unsigned A>
BEAT_DV Du<A> operator-(double b, const Du<A>& a) {
  Du<A> r;
  r.v = b - a.v;
  BEAT_DU_FOR(k, A) r.d[k] = -a.d[k];
  return r;
}
template <unsigned A>
BEAT_DV Du<A> operator*(const Du<A>& a, double b) {
  Du<A> r;
  r.v = a.v * b;
  BEAT_DU_FOR(k, A) r.d[k] = a.d[k] * b;
  return r;
}
template <unsigned A>
BEAT_DV Du<A> operator*(double b, const Du<A>& a) {
  return a * b;
}
// 1 / a
template <unsigned A>
BEAT_DV Du<A> inv(const Du<A>& a) {
  Du<A> r;
  r.v = beat_rcp(a.v);
  const double m = -r.v * r.v;
  BEAT_DU_FOR(k, A) r.d[k] = m * a.d[k];
  return r;
}
template <unsigned A, unsigned B>
BEAT_DV Du<A | B> operator/(const Du<A>& a, const Du<B>& b) {
  return a * inv(b);
}
template <unsigned B>
BEAT_DV Du<B> operator/(double a, const Du<B>& b) {
  return inv(b) * a;
}
template <class FM, unsigned A>
BEAT_DV Du<A> dexp(const FM& fm, const Du<A>& a) {
  Du<A> r;
  r.v = fm.exp(a.v);
  BEAT_DU_FOR(k, A) r.d[k] = r.v * a.d[k];
  return r;
}
// Several reciprocals from ONE v_rcp_f64 (quarter rate) + Newton step: 1/a = b c (1/(a b c)), ...  A reciprocal is 7 issue
// slots, a pair 10 instead of 14, a triple 13 instead of 21, a quadruple 16 instead of 28.  The callers' products stay
// far inside the double range (each factor is 1 + exp(.) or a sum of exponentials of the potential: < 1e60 each for
// -135 .. +500 mV) -- where a factor can reach 1e304 (Ito's development term) it keeps its own reciprocal.
BEAT_DV void rcp2(double a, double b, double& ia, double& ib) {
  const double r = beat_rcp(a * b);
  ia = r * b;
  ib = r * a;
}
BEAT_DV void rcp3(double a, double b, double c, double& ia, double& ib, double& ic) {
  const double ab = a * b;
  const double r = beat_rcp(ab * c);
  const double rc = r * c;
  ic = r * ab;
  ia = rc * b;
  ib = rc * a;
}
BEAT_DV void rcp4(double a, double b, double c, double d, double& ia, double& ib, double& ic, double& id) {
  const double ab = a * b, cd = c * d;
  const double r = beat_rcp(ab * cd);
  const double rab = r * cd, rcd = r * ab;
  ia = rab * b;
  ib = rab * a;
  ic = rcd * d;
  id = rcd * c;
}
// 1 / a given r = 1 / a.v
template <unsigned A>
BEAT_DV Du<A> inv_with(const Du<A>& a, double r) {
  Du<A> o;
  o.v = r;
  const double m = -r * r;
  BEAT_DU_FOR(k, A) o.d[k] = m * a.d[k];
  return o;
}
template <unsigned A, unsigned B, unsigned C>
BEAT_DV void inv3(const Du<A>& a, const Du<B>& b, const Du<C>& c, Du<A>& ia, Du<B>& ib, Du<C>& ic) {
  double ra, rb, rc;
  rcp3(a.v, b.v, c.v, ra, rb, rc);
  ia = inv_with(a, ra);
  ib = inv_with(b, rb);
  ic = inv_with(c, rc);
}

template <unsigned A>
BEAT_DV Du<A> sq(const Du<A>& a) {
  return a * a;
}
template <unsigned A>
BEAT_DV Du<A> cube(const Du<A>& a) {
  return a * a * a;
}

}  // namespace torord_detail

// LAND = true: the same cell with the Land contraction model (odes/torord/ToRORd_dynCl_endo_Land.ode): 7 more states
// (cross-bridges XS / XW, troponin-bound calcium CaTrpn, blocked tropomyosin TmB, distortions Zetas / Zetaw, dashpot Cd),
// 28 more parameters, troponin buffering as a flux of the calcium equation instead of a term of Bcai; in that file cai
// is declared with the mechanics states, so its row is 44 and the rows of cajsr .. xs2 are one lower (slot()).
// (round 6: 3 -- with the rows addressed through SGPR bases the uniform and the class kernels need 164 - 168 VGPRs without scratch,
// profiles/r06_ode_resources.md; as a launch bound it keeps a later change from silently falling back to two waves)
#ifndef BEAT_TORORD_WAVES
#define BEAT_TORORD_WAVES 3
#endif
template <bool LAND>
struct TorordGrl1T {
  static constexpr int NS = LAND ? 52 : 45, NP = LAND ? 140 : 112, V_INDEX = LAND ? 41 : 42;  // V_INDEX: membrane potential
  static constexpr bool ACCESSOR_PARAMS = true;  // derive / step take any p indexable by parameter number (beat_ode_jit.hip)
  // ode_run_kernel keeps the states of hand-written models in registers across steps (the generated kernel of round 1
  // spilled so heavily that this miscompiled and had to go through memory)
  static constexpr bool REGISTER_LOOP = true;
  static constexpr int WAVES = BEAT_TORORD_WAVES;  // waves per SIMD the kernels are compiled for
  static constexpr int WAVES_PER_NODE = 1;  // with 112 (140) per-node parameters in registers: one wave per SIMD, no scratch
  enum S {
    S_C1, S_C2, S_C3, S_I_, S_O_, S_CaMKt, S_Jrel_np, S_Jrel_p, S_a, S_ap, S_iF, S_iFp, S_iS, S_iSp, S_cai,
    S_cajsr, S_cansr, S_cass, S_cli, S_clss, S_ki, S_kss, S_nai, S_nass, S_d, S_fcaf, S_fcafp, S_fcas, S_ff_,
    S_ffp, S_fs, S_jca, S_nca_i, S_nca_ss, S_h, S_hp, S_j, S_jp, S_m, S_hL, S_hLp, S_mL, S_v, S_xs1, S_xs2,
    S_XS, S_XW, S_CaTrpn, S_TmB, S_Zetas, S_Zetaw, S_Cd  // LAND only
  };
  // row of a state in the (NS, N) array
  BEAT_HD static constexpr int slot(int s) { return !LAND ? s : (s == S_cai ? 44 : (s > S_cai && s < 45) ? s - 1 : s); }
  template <class IO>
  struct Rows {
    const IO& io;
    BEAT_DV double load(int s) const { return io.load(slot(s)); }
    BEAT_DV void store(int s, double x) const { io.store(slot(s), x); }
  };
  enum P {
    A_atp_, K_atp_, K_o_n_, fkatp_, gkatp_, Aff_, ICaL_fractionSS_, Kmn_, PCa_b_, dielConstant_, k2n_,
    offset_, tjca_, vShift_, BSLmax_, BSRmax_, KmBSL_, KmBSR_, cmdnmax_b_, csqnmax_, kmcmdn_, kmcsqn_,
    kmtrpn_, trpnmax_, CaMKo_, KmCaM_, KmCaMK_, aCaMK_, bCaMK_, EKshift_, Gto_b_, F_, R_, T_, zca_, zcl_,
    zk_, zna_, Fjunc_, GClCa_, GClb_, KdClCa_, GK1_b_, GKb_b_, GKr_b_, alpha_1_, beta_1_, GKs_b_, GNa_,
    GNaL_b_, thL_, Gncx_b_, INaCa_fractionSS_, KmCaAct_, kasymm_, kcaoff_, kcaon_, kna1_, kna2_, kna3_, qca_,
    qna_, wca_, wna_, wnaca_, GpCa_, KmCap_, H_, Khp_, Kki_, Kko_, Kmgatp_, Knai0_, Knao0_, Knap_, Kxkur_,
    MgADP_, MgATP_, Pnak_b_, delta_, eP_, k1m_, k1p_, k2m_, k2p_, k3m_, k3p_, k4m_, k4p_, Jrel_b_, bt_,
    cajsr_half_, Jup_b_, L_, rad__, PCab_, PKNa_, PNab_, cao_, clo_, ko_, nao_, celltype_, i_Stim_Amplitude_,
    i_Stim_End_, i_Stim_Period_, i_Stim_PulseDuration_, i_Stim_Start_, tauCa_, tauCl_, tauK_, tauNa_,
    // LAND only (.ode:648-679)
    emcoupling_, lmbda_, dLambda_, mode_, isacs_, calib_, ktrpn_, ntrpn_, Trpn50_, rw_, rs_, gammas_, gammaw_, phi_,
    Tot_A_, Beta0_, Beta1_, cat50_ref_, Tref_, kuw_, kws_, ku_, ntm_, p_a_, p_b_, p_k_, etal_, etas_
  };

  // parameter-only sub-expressions (celltype switches resolved: .ode PCa, Gto, Pnak, Gncx, GK1, GKb, GKr, GKs, GNaL,
  // upScale, cmdnmax, Jrel_inf)
  struct Derived {
    double FRT, FFRT, RTFna, RTFk, RTFcl;
    double cAF_myo, cAF_ss, cA2F_myo, cA2F_ss, vss_vmyo, vnsr_vmyo, vjsr_vss, vjsr_vnsr;
    double PCa, PCap, Gto, Pnak, Gncx_i, Gncx_ss, GK1s, GKb, GKrs, GKs, GNaL, upScale, cmdnmax, relScale, gKatp;
    double is_epi;
    double cA, gcao_cao, gko_ko, gnao_nao;     // Debye-Hueckel constant; extracellular activities gamma * conc
    double nk_a2, nk_a4, nk_b1, nk_cb3, nk_koK2, nk_1koK2, nk_Pden0;  // INaK
    double nc_k1, nc_h11, nc_cao;              // INaCa
    double a_rel, a_relp, btp;
    double ksu, kwu_kws_kuw, Aw_dL, cw, cs, ca_scale, kb, Cdash;  // LAND
    // Products, quotients and reciprocals of parameters alone that the step used to form per node: there is no scalar
    // fp64 unit, so every one of them was a VALU instruction per wavefront and tile -- an IEEE division (1.0 / p[..],
    // ~13 instructions) nine times over, seven reciprocals, some forty products.
    double nk_sKnai, nk_dKnai, nk_sKnao, nk_dKnao, nk_iKnap, nk_iKxkur, nk_iKki, nk_a3n, nk_cK, nk_cN, nk_cKb1;  // INaK
    double nc_sca, nc_sna, nc_nao_kna3, nc_wna_h11, nc_ikna3, nc_ikna1, nc_ikna2, nc_ikna12, nc_3zna, nc_km2;    // INaCa
    double r_thL, r_3thL, r_tjca, r_tauNa, r_tauK, r_tauCa, Afs, td0;
    double fi, fi_na, fi_k, fs, fs_na, fs_k, m4cA;
    double g_junc0, g_sl0, cu, cup, Jleak_c, J_cansr, GpCa_Km, nao_ko_ks;
    double b_trpn, b_cmdn, b_bsl, b_bsr, b_csqn;
    double nk_iKnai0, nk_nao_iKnao0, nk_cb3eP, half8, r_bt, r_btp;
    double e_er, e_sd;  // exp((EKshift + 70)/20), exp(-(vShift + 6)/20)
    double gate_bound, ito_bound;  // upper bounds of the rates of the gates whose time constant has a floor (see gate_b)
  };
  // P: anything indexable by parameter number (const double*, a per-lane array, MixedParams of beat_ode_kernel.h)
  template <class P>
  BEAT_HD static Derived derive(const P& p) {
    Derived q;
    const double RTF = p[R_] * p[T_] / p[F_];
    q.FRT = p[F_] / (p[R_] * p[T_]);
    q.FFRT = p[F_] * p[F_] / (p[R_] * p[T_]);
    q.RTFna = RTF / p[zna_];
    q.RTFk = RTF / p[zk_];
    q.RTFcl = RTF / p[zcl_];
    const double Ageo = p[L_] * ((2.0 * 3.14) * p[rad__]) + p[rad__] * ((2.0 * 3.14) * p[rad__]);
    const double Acap = 2.0 * Ageo;
    const double vcell = p[L_] * (p[rad__] * ((1000.0 * 3.14) * p[rad__]));
    const double vjsr = 0.0048 * vcell, vmyo = 0.68 * vcell, vnsr = 0.0552 * vcell, vss = 0.02 * vcell;
    q.cAF_myo = Acap / (p[F_] * vmyo);
    q.cAF_ss = Acap / (p[F_] * vss);
    q.cA2F_myo = Acap / ((2.0 * p[F_]) * vmyo);
    q.cA2F_ss = Acap / ((2.0 * p[F_]) * vss);
    q.vss_vmyo = vss / vmyo;
    q.vnsr_vmyo = vnsr / vmyo;
    q.vjsr_vss = vjsr / vss;
    q.vjsr_vnsr = vjsr / vnsr;
    const bool epi = p[celltype_] == 1.0, mid = p[celltype_] == 2.0;
    q.is_epi = epi ? 1.0 : 0.0;
    q.PCa = epi ? 1.2 * p[PCa_b_] : mid ? 2.0 * p[PCa_b_] : p[PCa_b_];
    q.PCap = 1.1 * q.PCa;
    q.Gto = (epi || mid) ? 2.0 * p[Gto_b_] : p[Gto_b_];
    q.Pnak = epi ? 0.9 * p[Pnak_b_] : mid ? 0.7 * p[Pnak_b_] : p[Pnak_b_];
    const double Gncx = epi ? 1.1 * p[Gncx_b_] : mid ? 1.4 * p[Gncx_b_] : p[Gncx_b_];
    q.Gncx_i = Gncx * (1.0 - p[INaCa_fractionSS_]);
    q.Gncx_ss = Gncx * p[INaCa_fractionSS_];
    const double sko = sqrt(p[ko_] / 5.0);
    q.GK1s = (epi ? 1.2 * p[GK1_b_] : mid ? 1.3 * p[GK1_b_] : p[GK1_b_]) * sko;
    q.GKb = epi ? 0.6 * p[GKb_b_] : p[GKb_b_];
    q.GKrs = (epi ? 1.3 * p[GKr_b_] : mid ? 0.8 * p[GKr_b_] : p[GKr_b_]) * sko;
    q.GKs = epi ? 1.4 * p[GKs_b_] : p[GKs_b_];
    q.GNaL = epi ? 0.6 * p[GNaL_b_] : p[GNaL_b_];
    q.upScale = epi ? 1.3 : 1.0;
    q.cmdnmax = epi ? 1.3 * p[cmdnmax_b_] : p[cmdnmax_b_];
    q.relScale = mid ? 1.7 : 1.0;
    q.gKatp = (1.0 / ((p[A_atp_] / p[K_atp_]) * (p[A_atp_] / p[K_atp_]) + 1.0)) * (pow(p[ko_] / p[K_o_n_], 0.24) * (p[fkatp_] * p[gkatp_]));
    const double TD = p[T_] * p[dielConstant_];
    q.cA = 1820000.0 / (TD * sqrt(TD));
    const double Io = (0.5 * (4.0 * p[cao_] + (p[clo_] + (p[ko_] + p[nao_])))) / 1000.0;
    const double gIo = sqrt(Io) / (sqrt(Io) + 1.0) - 0.3 * Io;
    q.gcao_cao = exp((-q.cA * 4.0) * gIo) * p[cao_];
    q.gko_ko = exp((-q.cA * 1.0) * gIo) * p[ko_];
    q.gnao_nao = exp((-q.cA * 1.0) * gIo) * p[nao_];
    q.nk_a2 = p[k2p_];
    q.nk_a4 = ((p[MgATP_] * p[k4p_]) / p[Kmgatp_]) / (1.0 + p[MgATP_] / p[Kmgatp_]);
    q.nk_b1 = p[MgADP_] * p[k1m_];
    q.nk_cb3 = (p[H_] * p[k3m_]) / (1.0 + p[MgATP_] / p[Kmgatp_]);
    q.nk_koK2 = (p[ko_] / p[Kko_]) * (p[ko_] / p[Kko_]);
    q.nk_1koK2 = (1.0 + p[ko_] / p[Kko_]) * (1.0 + p[ko_] / p[Kko_]);
    q.nk_Pden0 = p[H_] / p[Khp_] + 1.0;
    const double h10 = (p[nao_] / p[kna1_]) * (1.0 + p[nao_] / p[kna2_]) + (p[kasymm_] + 1.0);
    q.nc_h11 = (p[nao_] * p[nao_]) / (p[kna2_] * (h10 * p[kna1_]));
    q.nc_k1 = p[kcaon_] * (p[cao_] * (1.0 / h10));
    q.nc_cao = p[cao_];
    q.a_rel = 0.5 * p[bt_];
    q.btp = 1.25 * p[bt_];
    q.a_relp = 0.5 * q.btp;
    q.nk_sKnai = p[delta_] * (1.0 / 3.0);
    q.nk_dKnai = p[delta_] * q.FRT * (1.0 / 3.0);
    q.nk_sKnao = (1.0 - p[delta_]) * (1.0 / 3.0);
    q.nk_dKnao = (1.0 - p[delta_]) * q.FRT * (1.0 / 3.0);
    q.nk_iKnap = 1.0 / p[Knap_];
    q.nk_iKxkur = 1.0 / p[Kxkur_];
    q.nk_iKki = 1.0 / p[Kki_];
    q.nk_a3n = p[k3p_] * q.nk_koK2;
    q.nk_cK = 2.0 * p[zk_];
    q.nk_cN = 3.0 * p[zna_];
    q.nk_cKb1 = q.nk_cK * q.nk_b1;
    q.nc_sca = p[qca_] * q.FRT;
    q.nc_sna = p[qna_] * q.FRT;
    q.nc_nao_kna3 = p[nao_] / p[kna3_];
    q.nc_wna_h11 = p[wna_] * q.nc_h11;
    q.nc_ikna3 = 1.0 / p[kna3_];
    q.nc_ikna1 = 1.0 / p[kna1_];
    q.nc_ikna2 = 1.0 / p[kna2_];
    q.nc_ikna12 = 1.0 / (p[kna2_] * p[kna1_]);
    q.nc_3zna = 3.0 * p[zna_];
    q.nc_km2 = p[KmCaAct_] * p[KmCaAct_];
    q.r_thL = 1.0 / p[thL_];
    q.r_3thL = 1.0 / (3.0 * p[thL_]);
    q.r_tjca = 1.0 / p[tjca_];
    q.r_tauNa = 1.0 / p[tauNa_];
    q.r_tauK = 1.0 / p[tauK_];
    q.r_tauCa = 1.0 / p[tauCa_];
    q.Afs = 1.0 - p[Aff_];
    q.td0 = p[offset_] + 0.6;
    q.fi = 1.0 - p[ICaL_fractionSS_];
    q.fs = p[ICaL_fractionSS_];
    q.fi_na = q.fi * 0.00125;
    q.fi_k = q.fi * 0.0003574;
    q.fs_na = q.fs * 0.00125;
    q.fs_k = q.fs * 0.0003574;
    q.m4cA = -4.0 * q.cA;
    q.g_junc0 = p[Fjunc_] * p[GClCa_];
    q.g_sl0 = p[GClCa_] * (1.0 - p[Fjunc_]);
    q.cu = q.upScale * 0.005425;
    q.cup = (q.upScale * 2.75) * 0.005425;
    q.Jleak_c = 0.0048825 * (1.0 / 15.0);
    q.J_cansr = -p[Jup_b_] * (0.0048825 * (1.0 / 15.0)) - (1.0 / 60.0) * q.vjsr_vnsr;
    q.GpCa_Km = p[GpCa_] * p[KmCap_];
    q.nao_ko_ks = p[PKNa_] * p[nao_] + p[ko_];
    q.b_trpn = p[kmtrpn_] * p[trpnmax_];
    q.b_cmdn = q.cmdnmax * p[kmcmdn_];
    q.b_bsl = p[BSLmax_] * p[KmBSL_];
    q.b_bsr = p[BSRmax_] * p[KmBSR_];
    q.b_csqn = p[csqnmax_] * p[kmcsqn_];
    q.nk_iKnai0 = 1.0 / p[Knai0_];
    q.nk_nao_iKnao0 = p[nao_] / p[Knao0_];
    q.nk_cb3eP = q.nk_cb3 * p[eP_];
    {
      const double h2 = p[cajsr_half_] * p[cajsr_half_], h4 = h2 * h2;
      q.half8 = h4 * h4;
    }
    q.r_bt = 1.0 / p[bt_];
    q.r_btp = 1.0 / q.btp;
    q.e_er = exp((p[EKshift_] + 70.0) * (1.0 / 20.0));
    q.e_sd = exp(-0.05 * (p[vShift_] + 6.0));
    {
      // d: 1/td <= 1/td0; ff, fcaf: 1/7; fs, fcas, xs1: smaller; jca: 1/tjca; hL, hLp: 1/thL; nca: km2n = jca <= 1
      double b = 1.0;
      b = fmax(b, q.td0 > 0.0 ? 1.0 / q.td0 : 1.0e300);
      b = fmax(b, fabs(q.r_tjca));
      b = fmax(b, fabs(q.r_thL));
      q.gate_bound = b;
      // iF, iS: 1/(delta_epi (4.562 + ..)), delta_epi >= 0.05 for epi cells, 1 otherwise; iFp, iSp: times
      // 1/(dti_develop dti_recover) <= 1/(1.354 * 0.5)
      q.ito_bound = (1.0 / (1.354 * 0.5)) / (4.562 * (epi ? 0.05 : 1.0));
    }
    if constexpr (LAND) {  // .ode:682-717, parameter-only
      const double lam12 = p[lmbda_] < 1.2 ? p[lmbda_] : 1.2;
      const double rw = p[rw_], rs = p[rs_];
      q.ksu = p[kws_] * rw * (1.0 / rs - 1.0);
      const double kwu = p[kuw_] * (1.0 / rw - 1.0) - p[kws_];
      q.kwu_kws_kuw = kwu + p[kws_] + p[kuw_];
      q.Aw_dL = (p[Tot_A_] * rs / ((1.0 - rs) * rw + rs)) * p[dLambda_];
      q.cw = p[phi_] * p[kuw_] * ((1.0 - rs) * (1.0 - rw)) / ((1.0 - rs) * rw);
      q.cs = p[phi_] * p[kws_] * ((1.0 - rs) * rw) / rs;
      q.ca_scale = 1000.0 / (p[cat50_ref_] + p[Beta1_] * (lam12 - 1.0));
      q.kb = p[ku_] * pow(p[Trpn50_], p[ntm_]) / (1.0 - rs - (1.0 - rs) * rw);
      q.Cdash = lam12 - 1.0;
    }
    return q;
  }

  BEAT_DV static double grl1(double y, double f, double J, double expm1Jdt, double dt) {
    return y + ((fabs(J) > 1e-8) ? f * beat_rcp(J) * expm1Jdt : f * dt);
  }
  // phi(z) = (exp(z) - 1) / z for |z| <= 1/16 by its Taylor polynomial of degree 8 (first omitted term z^9/10! < 5e-18).
  // The GRL1 increment f (exp(J dt) - 1) / J is f dt phi(J dt); at dt = 0.01 ms nearly every state of nearly every node
  // has |J dt| below 1/16 (time constants above 0.16 ms), and the polynomial is 8 fma against an exp() (13) and, for
  // the non-gate states, a reciprocal (7) and the |J| > 1e-8 selection (phi(0) = 1 is that limit).  It is also the more
  // accurate form: exp(z) - 1 loses log2(1/|z|) bits to cancellation.  The branch is per lane (a node's result does not
  // depend on which nodes share its wavefront); lanes outside the window take the general form below it.
#ifndef BEAT_GRL1_PHI
#define BEAT_GRL1_PHI 1  // 0: the general form for every lane (A/B builds)
#endif
  static constexpr double PHI_WINDOW = 0.0625;
  // (BEAT_SCONST: the coefficient through a scalar register.  A 64-bit literal cannot be an operand; left alone the
  // compiler materialises each one in a VGPR pair -- two v_mov_b32 ahead of every fma of the Horner scheme, three VALU
  // instructions per step instead of one.  Opaque in an SGPR pair it costs two s_mov on the scalar unit.)
  BEAT_DV static double phi_small(double z) {
    double ph = z * BEAT_SCONST(1.0 / 362880.0) + BEAT_SCONST(1.0 / 40320.0);  // (two constants in one fma: one of them in VGPRs)
    ph = fma(z, ph, BEAT_SCONST(1.0 / 5040.0));
    ph = fma(z, ph, BEAT_SCONST(1.0 / 720.0));
    ph = fma(z, ph, BEAT_SCONST(1.0 / 120.0));
    ph = fma(z, ph, BEAT_SCONST(1.0 / 24.0));
    ph = fma(z, ph, BEAT_SCONST(1.0 / 6.0));
    ph = fma(z, ph, 0.5);
    return fma(z, ph, 1.0);
  }
  // gate with f = (inf - y) * rate, J = -rate  =>  y += (inf - y) (1 - exp(-dt rate))
  template <class FM>
  BEAT_DV static double gate(const FM& fm, double y, double inf, double rate, double dt) {
    const double z = -dt * rate;
    if (BEAT_GRL1_PHI > 1 && fabs(z) <= PHI_WINDOW) return fma(inf - y, -z * phi_small(z), y);  // (measured: no gain for gates)
    return y + (inf - y) * (1.0 - fm.exp(fmax(z, -746.0)));  // rates reach 1e21/ms at +350 mV: see FastMath::exp
  }
  // The same update for a gate whose rate has an upper bound B known per parameter set (a time constant c + 1/s has the
  // floor c): when dt B <= 1/32 -- `small`, uniform over the launch, so the branch is a scalar one -- 1 - exp(z), z = -dt rate,
  // is -z phi(z) by the Taylor polynomial of degree 7 (first omitted term z^8/9! < 3e-18): 7 fma in place of the exp().
  static constexpr double GATE_WINDOW = 1.0 / 32.0;
  BEAT_DV static double phi7(double z) {
    double ph = z * BEAT_SCONST(1.0 / 40320.0) + BEAT_SCONST(1.0 / 5040.0);
    ph = fma(z, ph, BEAT_SCONST(1.0 / 720.0));
    ph = fma(z, ph, BEAT_SCONST(1.0 / 120.0));
    ph = fma(z, ph, BEAT_SCONST(1.0 / 24.0));
    ph = fma(z, ph, BEAT_SCONST(1.0 / 6.0));
    ph = fma(z, ph, 0.5);
    return fma(z, ph, 1.0);
  }
  template <class FM>
  BEAT_DV static double gate_b(const FM& fm, double y, double inf, double rate, double dt, bool small) {
    if (small) {
      const double z = -dt * rate;
      return fma(inf - y, -z * phi7(z), y);
    }
    return gate(fm, y, inf, rate, dt);
  }
  template <class FM>
  BEAT_DV static double advance(const FM& fm, double y, double f, double J, double dt) {
    const double z = J * dt;
    if (BEAT_GRL1_PHI && fabs(z) <= PHI_WINDOW) return fma(f * dt, phi_small(z), y);
    return grl1(y, f, J, fm.exp(fmin(fmax(z, -746.0), 710.0)) - 1.0, dt);
  }

  template <class IO, class FM, class P>
  BEAT_DV static void step(const IO& io_, const P& p, const Derived& q, const FM& fm, double t, double dt) {
    using namespace torord_detail;
    const Rows<IO> io{io_};
    // directions of the dual numbers, per block: 0 = v always; 1, 2 = the block's ion concentrations
    constexpr unsigned DV = 1u, D1 = 2u, D2 = 4u;

    const double v = io.load(S_v);
    // Loads and stores retire through one in-order counter on gfx9: a load issued behind a store can only be waited for
    // together with that store (thousands of cycles to reach HBM).  So (1) the eight concentrations several blocks
    // read are loaded once, here, and (2) every block's gates are loaded before the previous block's stores are
    // issued ("pf_" values below) -- no load of this step follows a store of this step.
    const double nai = io.load(S_nai), ki = io.load(S_ki), cai = io.load(S_cai), cass = io.load(S_cass);
    const double nass = io.load(S_nass), kss = io.load(S_kss), cli = io.load(S_cli), clss = io.load(S_clss);
    // The Goldman-Hodgkin-Katz fluxes are 0/0 at v = 0: everything that sees the potential through vF/RT is evaluated
    // at a potential kept 1e-4 mV away from it (beat_guard, derivative 1), as the generated kernel did
    const double vg = beat_guard(v);
    const double vfrt = vg * q.FRT, vffrt = vg * q.FFRT;
    const double e1 = fm.exp(vfrt), e2 = e1 * e1;
    // Exponentials of the potential with slope 1/20, 1/10 and 1/5 mV (INa, Ito, ICaL, IKs: seven of them) are E20 = exp(v/20)
    // times or squared times a constant.
    const double E20 = fm.exp(0.05 * v), E10 = E20 * E20;
    const bool small_g = BEAT_GRL1_PHI && dt * q.gate_bound <= GATE_WINDOW, small_i = BEAT_GRL1_PHI && dt * q.ito_bound <= GATE_WINDOW;

    // running sums: total current seen by the state and its derivative w.r.t. the state
    double Iv = 0.0, dIv = 0.0;
    double Inai = 0.0, dInai = 0.0, Inass = 0.0, dInass = 0.0;
    double Iki = 0.0, dIki = 0.0, Ikss = 0.0, dIkss = 0.0;
    double Icai = 0.0, dIcai = 0.0, Icass = 0.0, dIcass = 0.0;

    // The two pumps come first: their dual-number intermediates are the widest of the step (four x_k with three
    // tangents each), and at this point nothing but the potential and the first running sums is live.
    // ---- INaK (.ode:418-444): directions 0 = v, 1 = nai, 2 = ki --------------------------------------------------------
    {
      // 1 / Knai and nao / Knao as exponentials of the negated argument (the specification divides by exp(.) Knai0)
      const Du<DV> iKnai = dexp(fm, mk<DV>(-(vfrt * q.nk_sKnai), -q.nk_dKnai)) * q.nk_iKnai0;
      const Du<DV> yo = dexp(fm, mk<DV>(-(vfrt * q.nk_sKnao), -q.nk_dKnao)) * q.nk_nao_iKnao0;
      const Du<D1> Nai = mk<D1>(nai, 0.0, 1.0);
      const Du<D2> Ki = mk<D2>(ki, 0.0, 0.0, 1.0);
      const Du<DV | D1> xn = Nai * iKnai;
      const Du<D2> xk = Ki * q.nk_iKki;
      Du<D1 | D2> rP;
      Du<DV | D1 | D2> rD1;
      Du<DV> rD3;
      inv3((Nai * q.nk_iKnap + q.nk_Pden0) + Ki * q.nk_iKxkur, (sq(1.0 + xk) + cube(1.0 + xn)) - 1.0,
           (cube(1.0 + yo) + q.nk_1koK2) - 1.0, rP, rD1, rD3);
      const Du<DV | D1 | D2> a1 = (p[k1p_] * cube(xn)) * rD1;
      const Du<DV | D1 | D2> b4 = (p[k4m_] * sq(xk)) * rD1;
      BEAT_TFENCE();
      const Du<DV> a3 = q.nk_a3n * rD3;
      const Du<DV> b2 = (p[k2m_] * cube(yo)) * rD3;
      const Du<D1 | D2> b3 = q.nk_cb3eP * rP;
      const double a2 = q.nk_a2, a4 = q.nk_a4, b1 = q.nk_b1;
      BEAT_TFENCE();
      // INaK = Pnak (zk JnakK + zna JnakNa), JnakK = 2 (E4 b1 - E3 a1), JnakNa = 3 (E1 a3 - E2 b3), E_k = x_k / sum x:
      // the x_k are formed one after the other and folded into the numerator and the sum (not kept, nor the E_k)
      const double cK = q.nk_cK, cN = q.nk_cN;
      auto x = a2 * (a1 * b3) + (b3 * (a2 * b4) + (a2 * (a1 * a4) + b3 * (b2 * b4)));   // x1
      auto S = x;
      auto N = cN * (x * a3);
      x = b4 * (a2 * a3) + (b4 * (a3 * b1) + (a3 * (a1 * a2) + b4 * (b1 * b2)));        // x2
      S = S + x;
      N = N - cN * (x * b3);
      x = b1 * (a3 * a4) + (a4 * (b1 * b2) + (a4 * (a2 * a3) + b1 * (b2 * b3)));        // x3
      S = S + x;
      N = N - cK * (x * a1);
      x = a1 * (b2 * b3) + (a1 * (a4 * b2) + (a1 * (a3 * a4) + b2 * (b3 * b4)));        // x4
      S = S + x;
      N = N + q.nk_cKb1 * x;
      const auto INaK = q.Pnak * (N * inv(S));
      Iv += INaK.v;
      dIv += INaK.d[0];
      Inai += 3.0 * INaK.v;
      dInai += 3.0 * INaK.d[1];
      Iki += -2.0 * INaK.v;
      dIki += -2.0 * INaK.d[2];
      BEAT_PIN(Iv); BEAT_PIN(dIv); BEAT_PIN(Inai); BEAT_PIN(dInai); BEAT_PIN(Iki); BEAT_PIN(dIki);
    }
    BEAT_TFENCE();

    // ---- INaCa, myoplasm and subspace (.ode:446-525): directions 0 = v, 1 = Na, 2 = Ca --------------------------------
    {
      const Du<DV> rhca = dexp(fm, mk<DV>(-(p[qca_] * vfrt), -q.nc_sca));  // 1 / hca: the only form it is used in
      const Du<DV> hna = dexp(fm, mk<DV>(p[qna_] * vfrt, q.nc_sna));
      // v-only part, shared by both compartments
      const Du<DV> rhna = inv(hna);
      const Du<DV> h7 = q.nc_nao_kna3 * (1.0 + rhna) + 1.0;
      const Du<DV> h9 = inv(h7);
      const Du<DV> h8 = q.nc_nao_kna3 * (rhna * h9);
      const Du<DV> k3pp = h8 * p[wnaca_];
      const Du<DV> k3 = h9 * p[wca_] + k3pp;
      const Du<DV> k8 = q.nc_wna_h11 * h8;
      const double k1 = q.nc_k1, k2 = p[kcaoff_], k5 = p[kcaoff_];
#define BEAT_NCX(NA, CA, GN, IV, DIV, INA, DINA, ICA, DICA, CAF)                                                        \
  {                                                                                                                  \
    const Du<D1> Na = mk<D1>(NA, 0.0, 1.0);                                                                          \
    const Du<D2> Ca = mk<D2>(CA, 0.0, 0.0, 1.0);                                                                     \
    const Du<DV | D1> h1 = (Na * q.nc_ikna3) * (hna + 1.0) + 1.0;                                                    \
    const Du<D1> h4 = (Na * q.nc_ikna1) * (1.0 + Na * q.nc_ikna2) + 1.0;                                             \
    const Du<D2> Ca2 = Ca * Ca;                                                                                      \
    Du<DV | D1> h3;                                                                                                  \
    Du<D1> h6;                                                                                                       \
    Du<D2> rallo;                                                                                                    \
    inv3(h1, h4, Ca2 + q.nc_km2, h3, h6, rallo);                                                                     \
    const Du<DV | D1> h2 = ((hna * Na) * q.nc_ikna3) * h3;                                                           \
    const Du<D1> h5 = ((Na * Na) * q.nc_ikna12) * h6;                                                                \
    const Du<DV | D1> k4pp = h2 * p[wnaca_];                                                                         \
    const Du<DV | D1> k4 = (h3 * p[wca_]) * rhca + k4pp;                                                             \
    const Du<D1 | D2> k6 = p[kcaon_] * (Ca * h6);                                                                    \
    const Du<DV | D1> k7 = p[wna_] * (h2 * h5);                                                                      \
    /* I = allo G (zca JCa + zna JNa), JCa = E2 k2 - E1 k1, JNa = E3 k4pp + 3 (E4 k7 - E1 k8) - E2 k3pp, E_k = x_k / sum x:    */ \
    /* each x_k is folded into the numerator and the sum as soon as it is formed                                          */ \
    Du<DV | D1 | D2> S, N;                                                                                             \
    {                                                                                                                  \
      const auto x1 = (k2 * k4) * (k6 + k7) + (k5 * k7) * (k2 + k3);                                                   \
      S = x1;                                                                                                          \
      N = x1 * (k8 * (-q.nc_3zna) - p[zca_] * k1);                                                                     \
    }                                                                                                                  \
    {                                                                                                                  \
      const auto x2 = (k1 * k7) * (k4 + k5) + (k4 * k6) * (k1 + k8);                                                   \
      S = S + x2;                                                                                                      \
      N = N + x2 * (p[zca_] * k2 - p[zna_] * k3pp);                                                                    \
    }                                                                                                                  \
    {                                                                                                                  \
      const auto x3 = (k1 * k3) * (k6 + k7) + (k6 * k8) * (k2 + k3);                                                   \
      S = S + x3;                                                                                                      \
      N = N + x3 * (p[zna_] * k4pp);                                                                                   \
    }                                                                                                                  \
    {                                                                                                                  \
      const auto x4 = (k2 * k8) * (k4 + k5) + (k3 * k5) * (k1 + k8);                                                   \
      S = S + x4;                                                                                                      \
      N = N + x4 * (q.nc_3zna * k7);                                                                                   \
    }                                                                                                                  \
    const Du<D2> allo = Ca2 * rallo;                                                                                   \
    const auto I = (allo * (GN)) * (N * inv(S));                                                                       \
    IV += I.v;                                                                                                       \
    DIV += I.d[0];                                                                                                   \
    INA += 3.0 * I.v;                                                                                                \
    DINA += 3.0 * I.d[1];                                                                                            \
    ICA += (CAF) * I.v;                                                                                              \
    DICA += (CAF) * I.d[2];                                                                                          \
    BEAT_PIN(IV); BEAT_PIN(DIV); BEAT_PIN(INA); BEAT_PIN(DINA); BEAT_PIN(ICA); BEAT_PIN(DICA);                       \
  }
      // (the Land file's calcium equation has INaCa_i / 3)
      BEAT_NCX(nai, cai, q.Gncx_i, Iv, dIv, Inai, dInai, Icai, dIcai, (LAND ? -2.0 / 3.0 : -2.0))
      BEAT_TFENCE();
      BEAT_NCX(nass, cass, q.Gncx_ss, Iv, dIv, Inass, dInass, Icass, dIcass, -2.0)
#undef BEAT_NCX
    }
    BEAT_TFENCE();

    // ---- CaMK (.ode:413-416): phosphorylated fraction fp = 1/(1 + KmCaMK/CaMKa), shared by INa, INaL, Ito, ICaL, Jup,
    //      Jrel (the specification writes it out six times) ---------------------------------------------------------
    double fp, dfp_dcass;
    double pf_m, pf_h, pf_hp, pf_j, pf_jp, pf_mL, pf_hL, pf_hLp, pf_a, pf_ap, pf_iF, pf_iFp, pf_iS, pf_iSp;
    double pf_d, pf_ff, pf_fs, pf_fcaf, pf_fcas, pf_jca, pf_ffp, pf_fcafp, pf_nca_i, pf_nca_ss;
    double pf_O, pf_C1, pf_C2, pf_C3, pf_I, pf_xs1, pf_xs2, pf_cajsr, pf_cansr, pf_Jrel_np, pf_Jrel_p;
    double pf_XS = 0.0, pf_XW = 0.0, pf_CaTrpn = 0.0, pf_TmB = 0.0, pf_Zetas = 0.0, pf_Zetaw = 0.0, pf_Cd = 0.0;
    {
      const double CaMKt = io.load(S_CaMKt);
      const double rk = beat_rcp(cass + p[KmCaM_]);
      const double rc = cass * rk;                                   // 1/(KmCaM/cass + 1)
      const double CaMKb = p[CaMKo_] * (1.0 - CaMKt) * rc;
      const double dCaMKb_dcass = p[CaMKo_] * (1.0 - CaMKt) * p[KmCaM_] * rk * rk;
      const double dCaMKb_dCaMKt = -p[CaMKo_] * rc;
      const double CaMKa = CaMKb + CaMKt;
      const double ra = beat_rcp(CaMKa + p[KmCaMK_]);
      fp = CaMKa * ra;
      dfp_dcass = p[KmCaMK_] * ra * ra * dCaMKb_dcass;
      const double f = -CaMKt * p[bCaMK_] + (CaMKb * p[aCaMK_]) * (CaMKb + CaMKt);
      const double J = -p[bCaMK_] + p[aCaMK_] * (dCaMKb_dCaMKt * (CaMKb + CaMKt) + CaMKb * (dCaMKb_dCaMKt + 1.0));
      pf_m = io.load(S_m), pf_h = io.load(S_h), pf_hp = io.load(S_hp), pf_j = io.load(S_j), pf_jp = io.load(S_jp);
      io.store(S_CaMKt, advance(fm, CaMKt, f, J, dt));
    }
    BEAT_TFENCE();

    // ---- reversal potentials (.ode:527-532) ---------------------------------------------------------------------------
    double rnai, rki, rcai, rcass;  // (rcai, rcass: the nca states and IKs below)
    rcp4(nai, ki, cai, cass, rnai, rki, rcai, rcass);
    const double ENa = q.RTFna * fm.log(p[nao_] * rnai);
    const double EK = q.RTFk * fm.log(p[ko_] * rki);
    const double uK = v - EK;                                        // driving force of the K currents
    const double dENa = -q.RTFna * rnai, dEK = -q.RTFk * rki;         // d E / d (own concentration)
    BEAT_TFENCE();

    // ---- INa (.ode:577-599) -------------------------------------------------------------------------------------------
    double tm_rate;  // shared with INaL: tm = tmL
    {
      const double m = pf_m, h = pf_h, hp = pf_hp, j = pf_j, jp = pf_jp;
      const double gNa = (m * m * m) * p[GNa_] * (j * (h * (1.0 - fp)) + jp * (fp * hp));
      const double INa = gNa * (v - ENa);
      Iv += INa;
      dIv += gNa;
      Inai += INa;
      dInai += -gNa * dENa;
      BEAT_PIN(Iv); BEAT_PIN(dIv); BEAT_PIN(Inai); BEAT_PIN(dInai);
      BEAT_TFENCE();
      const double tm_den = 0.06487 * fm.exp(-((v - 4.823) * (1.0 / 51.12)) * ((v - 4.823) * (1.0 / 51.12))) +
                            0.1292 * fm.exp(-((v + 45.79) * (1.0 / 15.54)) * ((v + 45.79) * (1.0 / 15.54)));
      BEAT_TFENCE();
      const double em = fm.exp(-(v + 56.86) * (1.0 / 9.03));
      const double eh = fm.exp((v + 71.55) * (1.0 / 7.43));
      double rm, rh, rhp;
      rcp4(tm_den, em + 1.0, eh + 1.0, eh * 2.2423782291926058 + 1.0, tm_rate, rm, rh, rhp);  // exp(6/7.43)
      BEAT_TFENCE();
      double rate_h, rate_j;
      if (v > -40.0) {
        const double ea = fm.exp(0.0900900900900901 * v);
        double r1, r2;
        rcp2(0.13 * ea + 0.0497581410839387, E10 + 0.0407622039783662, r1, r2);
        rate_h = 0.77 * ea * r1;
        rate_j = 0.6 * fm.exp(0.157 * v) * r2;
      } else {
        rate_h = 4.43126792958051e-7 * fm.exp(-0.147058823529412 * v) + (2.7 * fm.exp(0.079 * v) + 310000.0 * fm.exp(0.3485 * v));
        double r1, r2;
        rcp2(50262745825.954 * fm.exp(0.311 * v) + 1.0, 1.0 * fm.exp(0.1378 * v) + 0.00396086833990426, r1, r2);
        const double aj = -(v + 37.78) * (25428.0 * fm.exp(0.28831 * v) + 6.948e-6) * fm.exp(-0.04391 * v) * r1;
        const double bj = 0.02424 * fm.exp(0.12728 * v) * r2;
        rate_j = aj + bj;
      }
      BEAT_TFENCE();
      pf_mL = io.load(S_mL), pf_hL = io.load(S_hL), pf_hLp = io.load(S_hLp);
      io.store(S_m, gate(fm, m, rm * rm, tm_rate, dt));
      io.store(S_h, gate(fm, h, rh * rh, rate_h, dt));
      io.store(S_hp, gate(fm, hp, rhp * rhp, rate_h, dt));
      BEAT_TFENCE();
      io.store(S_j, gate(fm, j, rh * rh, rate_j, dt));
      io.store(S_jp, gate(fm, jp, rh * rh, rate_j * (1.0 / 1.46), dt));
    }
    BEAT_TFENCE();

    // ---- INaL (.ode:561-575) ------------------------------------------------------------------------------------------
    {
      const double mL = pf_mL, hL = pf_hL, hLp = pf_hLp;
      const double gNaL = mL * q.GNaL * (fp * hLp + hL * (1.0 - fp));
      const double INaL = gNaL * (v - ENa);
      Iv += INaL;
      dIv += gNaL;
      Inai += INaL;
      dInai += -gNaL * dENa;
      BEAT_PIN(Iv); BEAT_PIN(dIv); BEAT_PIN(Inai); BEAT_PIN(dInai);
      BEAT_TFENCE();
      const double ehL = fm.exp((v + 87.61) * (1.0 / 7.488));
      pf_a = io.load(S_a), pf_ap = io.load(S_ap), pf_iF = io.load(S_iF), pf_iFp = io.load(S_iFp), pf_iS = io.load(S_iS),
      pf_iSp = io.load(S_iSp);
      double mLss, hLss, hLpss;
      rcp3(fm.exp(-(v + 42.85) * (1.0 / 5.264)) + 1.0, ehL + 1.0, ehL * 2.288717124596482 + 1.0, mLss, hLss, hLpss);  // exp(6.2/7.488)
      io.store(S_mL, gate(fm, mL, mLss, tm_rate, dt));
      io.store(S_hL, gate_b(fm, hL, hLss, q.r_thL, dt, small_g));
      io.store(S_hLp, gate_b(fm, hLp, hLpss, q.r_3thL, dt, small_g));
    }
    BEAT_TFENCE();

    // ---- Ito (.ode:379-403) -------------------------------------------------------------------------------------------
    {
      const double ve = p[EKshift_] + v;
      const double a = pf_a, ap = pf_ap, iF = pf_iF, iFp = pf_iFp, iS = pf_iS, iSp = pf_iSp;
      const double AiF = beat_rcp(fm.exp((ve - 213.6) * (1.0 / 151.2)) + 1.0);
      const double dAiF = -AiF * (1.0 - AiF) * (1.0 / 151.2);
      const double i_ = AiF * iF + (1.0 - AiF) * iS, ip = AiF * iFp + (1.0 - AiF) * iSp;
      const double gto = q.Gto * (i_ * (a * (1.0 - fp)) + ip * (ap * fp));
      const double dgto = q.Gto * (dAiF * (iF - iS) * (a * (1.0 - fp)) + dAiF * (iFp - iSp) * (ap * fp));
      const double Ito = gto * uK;
      Iv += Ito;
      dIv += gto + dgto * uK;
      Iki += Ito;
      dIki += -gto * dEK;
      BEAT_PIN(Iv); BEAT_PIN(dIv); BEAT_PIN(Iki); BEAT_PIN(dIki);
      BEAT_TFENCE();
      const double ea = fm.exp(-(ve - 14.34) * (1.0 / 14.82));
      const double et = fm.exp(-(ve - 18.4099) * (1.0 / 29.3814));
      // ta = 1.0515 / (1/(1.2089 (et + 1)) + 3.5/(exp((ve + 100)/29.3814) + 1)), exp((ve + 100)/29.3814) = exp(118.4099/29.3814) / et:
      // 1/ta over the common denominator -- one reciprocal, shared with the two steady states
      double ass, assp, ta_rate;
      {
        const double c = 56.266384148520984 + et;
        rcp3(ea + 1.0, ea * 1.9635691902911017 + 1.0, (1.2089 * 1.0515) * ((et + 1.0) * c), ass, assp, ta_rate);  // exp(10/14.82)
        ta_rate *= c + (3.5 * 1.2089) * (et * (et + 1.0));
      }
      BEAT_TFENCE();
      const double iss_den = fm.exp((ve + 43.94) * (1.0 / 5.711)) + 1.0;
      const double er = E20 * q.e_er;  // exp((ve + 70)/20), ve = v + EKshift
      double delta_epi = 1.0;
      if (q.is_epi != 0.0) delta_epi = 1.0 - 0.95 * beat_rcp((er * er) * (er * er) + 1.0);  // exp((ve + 70)/5)
      // tiF = delta_epi (4.562 + 1/sF): 1/tiF = sF / (delta_epi (4.562 sF + 1)); tiS likewise
      const double sF = 0.3933 * fm.exp(-(ve + 100.0) * (1.0 / 100.0)) + 0.08004 * fm.exp((ve + 50.0) * (1.0 / 16.59));
      BEAT_TFENCE();
      const double sS = 0.001416 * fm.exp(-(ve + 96.52) * (1.0 / 59.05)) + 1.78e-8 * fm.exp((ve + 114.1) * (1.0 / 8.079));
      double iss, rtiF, rtiS;
      rcp3(iss_den, delta_epi * (4.562 * sF + 1.0), delta_epi * (23.62 * sS + 1.0), iss, rtiF, rtiS);
      rtiF *= sF;
      rtiS *= sS;
      BEAT_TFENCE();
      const double xdev = fmin(-(ve - 12.23) * (1.0 / 0.2154), 700.0);  // exp() of it overflows below -138 mV
      const double dti_develop = 1.354 + 0.0001 * beat_rcp(fm.exp(xdev) + fm.exp((ve - 167.4) * (1.0 / 15.89)));  // (up to 1e304: alone)
      // dti_recover = 1 - 0.5/(er + 1) = (er + 0.5)/(er + 1): 1/(develop recover) with one reciprocal
      BEAT_TFENCE();
      const double rdd = (er + 1.0) * beat_rcp(dti_develop * (er + 0.5));
      pf_d = io.load(S_d), pf_ff = io.load(S_ff_), pf_fs = io.load(S_fs), pf_fcaf = io.load(S_fcaf), pf_fcas = io.load(S_fcas);
      pf_jca = io.load(S_jca), pf_ffp = io.load(S_ffp), pf_fcafp = io.load(S_fcafp), pf_nca_i = io.load(S_nca_i);
      pf_nca_ss = io.load(S_nca_ss);
      io.store(S_a, gate(fm, a, ass, ta_rate, dt));
      io.store(S_ap, gate(fm, ap, assp, ta_rate, dt));
      BEAT_TFENCE();
      io.store(S_iF, gate_b(fm, iF, iss, rtiF, dt, small_i));
      io.store(S_iFp, gate_b(fm, iFp, iss, rtiF * rdd, dt, small_i));
      BEAT_TFENCE();
      io.store(S_iS, gate_b(fm, iS, iss, rtiS, dt, small_i));
      io.store(S_iSp, gate_b(fm, iSp, iss, rtiS * rdd, dt, small_i));
    }
    BEAT_TFENCE();

    // ---- ICaL gates and the common gate factors of ICaL / ICaNa / ICaK (.ode:312-377) ---------------------------------
    //   I_X_c = frac_c * scale_X * Gc * Phi_X_c,  Gc = d [ (1 - fp) (f (1 - nca) + nca fca jca) PCa + fp (fp_ (1 - nca) + nca fcap jca) PCap ]
    double G_i, dG_i_dv, G_ss, dG_ss_dv, dG_ss_dfp;
    {
      const double d = pf_d, ff = pf_ff, fs = pf_fs, fcaf = pf_fcaf, fcas = pf_fcas, jca = pf_jca, ffp = pf_ffp, fcafp = pf_fcafp;
      const double nca_i = pf_nca_i, nca_ss = pf_nca_ss;
      const double sA = beat_rcp(E10 * 0.36787944117144233 + 1.0);  // exp((v - 10)/10) = exp(v/10) exp(-1)
      const double Afcaf = 0.3 + 0.6 * sA, dAfcaf = -0.06 * sA * (1.0 - sA);
      const double Afs = q.Afs;
      const double f = p[Aff_] * ff + Afs * fs, fpx = p[Aff_] * ffp + Afs * fs;
      const double fca = Afcaf * fcaf + (1.0 - Afcaf) * fcas, fcap = Afcaf * fcafp + (1.0 - Afcaf) * fcas;
      const double dfca = dAfcaf * (fcaf - fcas), dfcap = dAfcaf * (fcafp - fcas);
      {
        const double A = f * (1.0 - nca_i) + nca_i * (fca * jca), Ap = fpx * (1.0 - nca_i) + nca_i * (fcap * jca);
        G_i = d * ((1.0 - fp) * A * q.PCa + fp * Ap * q.PCap);
        dG_i_dv = d * ((1.0 - fp) * (nca_i * jca * dfca) * q.PCa + fp * (nca_i * jca * dfcap) * q.PCap);
      }
      {
        const double A = f * (1.0 - nca_ss) + nca_ss * (fca * jca), Ap = fpx * (1.0 - nca_ss) + nca_ss * (fcap * jca);
        G_ss = d * ((1.0 - fp) * A * q.PCa + fp * Ap * q.PCap);
        dG_ss_dv = d * ((1.0 - fp) * (nca_ss * jca * dfca) * q.PCa + fp * (nca_ss * jca * dfcap) * q.PCap);
        dG_ss_dfp = d * (Ap * q.PCap - A * q.PCa);
      }
      BEAT_PIN(G_i); BEAT_PIN(dG_i_dv); BEAT_PIN(G_ss); BEAT_PIN(dG_ss_dv); BEAT_PIN(dG_ss_dfp);
      BEAT_TFENCE();
      // gates
      // Every time constant of the form  tau = c + 1/s  enters as its rate  1/tau = s / (c s + 1): one reciprocal instead
      // of two (three where s itself held a 1/exp: 0.0045/e + 0.0045 e = 0.0045 (1 + e^2)/e), and the seven
      // denominators of the block share two v_rcp_f64.
      const double dss = (v >= 31.4978) ? 1.0 : 1.0763 * fm.exp(-1.007 * fm.exp(-0.0829 * v));
      const double vs = v + p[vShift_];
      // td = td0 + 1/sd, sd = exp(-(vs + 6)/20) + exp(0.09 (vs + 14)) = Wd / E20, Wd = exp(-(vShift + 6)/20) + exp(0.09 (vs + 14)) E20:
      // 1/td = Wd / (td0 Wd + E20)
      const double Wd = q.e_sd + fm.exp(0.09 * (vs + 14.0)) * E20;
      BEAT_TFENCE();
      const double fss_den = fm.exp((v + 19.58) * (1.0 / 3.696)) + 1.0;
      const double e20 = E10 * 7.38905609893065;  // exp((v + 20)/10) = exp(v/10) exp(2)
      const double w20 = 0.0045 * (1.0 + e20 * e20);                                       // tff = 7 + e20 / w20
      BEAT_TFENCE();
      const double sfs = 3.5e-5 * fm.exp(-(v + 5.0) * (1.0 / 4.0)) + 3.5e-5 * fm.exp((v + 5.0) * (1.0 / 6.0));  // tfs = 1000 + 1/sfs
      double rtd, fss, rtff, rtfs;
      rcp4(q.td0 * Wd + E20, fss_den, 7.0 * w20 + e20, 1000.0 * sfs + 1.0, rtd, fss, rtff, rtfs);
      rtd *= Wd;
      rtff *= w20;
      rtfs *= sfs;
      BEAT_TFENCE();
      const double e4 = fm.exp((v - 4.0) * (1.0 / 7.0));
      const double w4 = 0.04 * (1.0 + e4 * e4);                                            // tfcaf = 7 + e4 / w4
      BEAT_TFENCE();
      const double sfcas = 0.00012 * fm.exp(-v * (1.0 / 3.0)) + (0.00012 * 1.770794952435155) * e4;  // tfcas = 100 + 1/sfcas; exp(v/7) = e4 exp(4/7)
      const double jcass_den = fm.exp((v + 18.08) * (1.0 / 2.7916)) + 1.0;
      double rtfcaf, rtfcas, jcass;
      rcp3(7.0 * w4 + e4, 100.0 * sfcas + 1.0, jcass_den, rtfcaf, rtfcas, jcass);
      rtfcaf *= w4;
      rtfcas *= sfcas;
      BEAT_TFENCE();
      pf_O = io.load(S_O_), pf_C1 = io.load(S_C1), pf_C2 = io.load(S_C2), pf_C3 = io.load(S_C3), pf_I = io.load(S_I_);
      io.store(S_d, gate_b(fm, d, dss, rtd, dt, small_g));
      io.store(S_ff_, gate_b(fm, ff, fss, rtff, dt, small_g));
      BEAT_TFENCE();
      io.store(S_ffp, gate_b(fm, ffp, fss, rtff * (1.0 / 2.5), dt, small_g));
      io.store(S_fs, gate_b(fm, fs, fss, rtfs, dt, small_g));
      BEAT_TFENCE();
      io.store(S_fcaf, gate_b(fm, fcaf, fss, rtfcaf, dt, small_g));
      io.store(S_fcafp, gate_b(fm, fcafp, fss, rtfcaf * (1.0 / 2.5), dt, small_g));
      BEAT_TFENCE();
      io.store(S_fcas, gate_b(fm, fcas, fss, rtfcas, dt, small_g));
      io.store(S_jca, gate_b(fm, jca, jcass, q.r_tjca, dt, small_g));
      BEAT_TFENCE();
      // nca: f = anca k2n - km2n nca, km2n = jca, anca = 1/(k2n/km2n + (Kmn/ca + 1)^4) = jca / (k2n + jca (Kmn/ca + 1)^4): J = -jca
      {
        const double xi = p[Kmn_] * rcai + 1.0, xs = p[Kmn_] * rcass + 1.0;
        double anca_i, anca_ss;
        rcp2(p[k2n_] + jca * ((xi * xi) * (xi * xi)), p[k2n_] + jca * ((xs * xs) * (xs * xs)), anca_i, anca_ss);
        anca_i *= jca;
        anca_ss *= jca;
        const double f_i = anca_i * p[k2n_] - jca * nca_i, f_ss = anca_ss * p[k2n_] - jca * nca_ss;
        if (small_g) {  // the increment f (exp(J dt) - 1)/J as f dt phi(J dt), J = -jca (a gate: <= 1)
          const double dph = dt * phi7(-jca * dt);
          io.store(S_nca_i, fma(f_i, dph, nca_i));
          io.store(S_nca_ss, fma(f_ss, dph, nca_ss));
        } else {
          const double em1 = fm.exp(fmax(-jca * dt, -746.0)) - 1.0;
          io.store(S_nca_i, grl1(nca_i, f_i, -jca, em1, dt));
          io.store(S_nca_ss, grl1(nca_ss, f_ss, -jca, em1, dt));
        }
      }
    }
    BEAT_TFENCE();

    // ---- Goldman-Hodgkin-Katz fluxes of the L-type channel, background Ca and Na (.ode:326-347, 600-602, 610) ----------
    double ICaL_ss;  // needed again by the ryanodine receptor
    {
      // dual directions: 0 = v, 1 = the ion the flux carries (cai / nai / ki, cass / nass / kss)
      const Du<DV> Ve1 = mk<DV>(e1, e1 * q.FRT), Ve2 = mk<DV>(e2, 2.0 * e2 * q.FRT);
      const Du<DV> Vff = mk<DV>(vffrt, q.FFRT);
      Du<DV> R1, R2;
      {
        double r1v, r2v;
        rcp2(e1 - 1.0, e2 - 1.0, r1v, r2v);
        R1 = inv_with(Ve1, r1v);
        R2 = inv_with(Ve2, r2v);
      }
      const double fi = q.fi, fs_ = q.fs;
      // myoplasm: activity coefficients gamma = exp(-cA z^2 g(I)), g = sqrt(I)/(1 + sqrt(I)) - 0.3 I, I = ionic strength
      {
        const double Ii = (0.5 * (4.0 * cai + (cli + (ki + nai)))) * (1.0 / 1000.0);
        const double rsI = beat_rsqrt(Ii);                              // 1/sqrt(I) serves sqrt(I) = I/sqrt(I) too
        const double sI = Ii * rsI, r1s = beat_rcp(1.0 + sI);
        const double g = sI * r1s - 0.3 * Ii;
        const double dg = 0.5 * rsI * r1s * r1s - 0.3;                    // dg/dI
        const double g1 = fm.exp(-q.cA * g);                              // z = 1
        const double g2 = (g1 * g1) * (g1 * g1);                          // z = 2: exp(-4 cA g)
        const double dg1_dI = -q.cA * dg * g1, dg2_dI = q.m4cA * dg * g2;
        const Du<DV> Gi = mk<DV>(G_i, dG_i_dv);
        {  // Ca: d I / d cai = 2/1000
          const Du<D1> act = mk<D1>(g2 * cai, 0.0, g2 + cai * dg2_dI * 0.002);
          const Du<DV | D1> Phi = (4.0 * Vff) * (act * Ve2 - q.gcao_cao) * R2;
          const Du<DV | D1> ICaL_i = fi * (Gi * Phi);
          const Du<DV | D1> ICab = p[PCab_] * Phi;
          Iv += ICaL_i.v + ICab.v;
          dIv += ICaL_i.d[0] + ICab.d[0];
          if constexpr (LAND) {  // the Land file's calcium equation has no ICaL_i term
            Icai += ICab.v;
            dIcai += ICab.d[1];
          } else {
            Icai += ICaL_i.v + ICab.v;
            dIcai += ICaL_i.d[1] + ICab.d[1];
          }
          BEAT_PIN(Iv); BEAT_PIN(dIv); BEAT_PIN(Icai); BEAT_PIN(dIcai);
        }
        {  // Na: d I / d nai = 0.5/1000
          const Du<D1> act = mk<D1>(g1 * nai, 0.0, g1 + nai * dg1_dI * 0.0005);
          const Du<DV | D1> ICaNa_i = q.fi_na * (Gi * (Vff * (act * Ve1 - q.gnao_nao) * R1));
          const Du<DV | D1> INab = p[PNab_] * (Vff * (mk<D1>(nai, 0.0, 1.0) * Ve1 - p[nao_]) * R1);
          Iv += ICaNa_i.v + INab.v;
          dIv += ICaNa_i.d[0] + INab.d[0];
          Inai += ICaNa_i.v + INab.v;
          dInai += ICaNa_i.d[1] + INab.d[1];
          BEAT_PIN(Iv); BEAT_PIN(dIv); BEAT_PIN(Inai); BEAT_PIN(dInai);
        }
        {  // K
          const Du<D1> act = mk<D1>(g1 * ki, 0.0, g1 + ki * dg1_dI * 0.0005);
          const Du<DV | D1> ICaK_i = q.fi_k * (Gi * (Vff * (act * Ve1 - q.gko_ko) * R1));
          Iv += ICaK_i.v;
          dIv += ICaK_i.d[0];
          Iki += ICaK_i.v;
          dIki += ICaK_i.d[1];
          BEAT_PIN(Iv); BEAT_PIN(dIv); BEAT_PIN(Iki); BEAT_PIN(dIki);
        }
      }
      BEAT_TFENCE();
      // subspace: the gate factor also depends on cass through fp
      {
        const double Is = (0.5 * (4.0 * cass + (clss + (kss + nass)))) * (1.0 / 1000.0);
        const double rsI = beat_rsqrt(Is);
        const double sI = Is * rsI, r1s = beat_rcp(1.0 + sI);
        const double g = sI * r1s - 0.3 * Is;
        const double dg = 0.5 * rsI * r1s * r1s - 0.3;
        const double g1 = fm.exp(-q.cA * g);
        const double g2 = (g1 * g1) * (g1 * g1);
        const double dg1_dI = -q.cA * dg * g1, dg2_dI = q.m4cA * dg * g2;
        {
          const Du<DV | D1> Gs = mk<DV | D1>(G_ss, dG_ss_dv, dG_ss_dfp * dfp_dcass);
          const Du<D1> act = mk<D1>(g2 * cass, 0.0, g2 + cass * dg2_dI * 0.002);
          const Du<DV | D1> I = fs_ * (Gs * ((4.0 * Vff) * (act * Ve2 - q.gcao_cao) * R2));
          ICaL_ss = I.v;
          Iv += I.v;
          dIv += I.d[0];
          Icass += I.v;
          dIcass += I.d[1];
          BEAT_PIN(ICaL_ss); BEAT_PIN(Iv); BEAT_PIN(dIv); BEAT_PIN(Icass); BEAT_PIN(dIcass);
        }
        const Du<DV> Gs = mk<DV>(G_ss, dG_ss_dv);
        {
          const Du<D1> act = mk<D1>(g1 * nass, 0.0, g1 + nass * dg1_dI * 0.0005);
          const Du<DV | D1> I = q.fs_na * (Gs * (Vff * (act * Ve1 - q.gnao_nao) * R1));
          Iv += I.v;
          dIv += I.d[0];
          Inass += I.v;
          dInass += I.d[1];
          BEAT_PIN(Iv); BEAT_PIN(dIv); BEAT_PIN(Inass); BEAT_PIN(dInass);
        }
        {
          const Du<D1> act = mk<D1>(g1 * kss, 0.0, g1 + kss * dg1_dI * 0.0005);
          const Du<DV | D1> I = q.fs_k * (Gs * (Vff * (act * Ve1 - q.gko_ko) * R1));
          Iv += I.v;
          dIv += I.d[0];
          Ikss += I.v;
          dIkss += I.d[1];
          BEAT_PIN(Iv); BEAT_PIN(dIv); BEAT_PIN(Ikss); BEAT_PIN(dIkss);
        }
      }
    }
    BEAT_TFENCE();

    // ---- K currents: IK1, IKb, IKr (+ Markov states), IKs (+ gates), I_katp (.ode:534-575, 612-615) --------------------
    {
      // IK1: aK1, bK1 functions of u = v - EK
      // K1ss = aK1/(aK1 + bK1) with aK1 = 4.094/A, bK1 = E/B (A = ea + 1, B = eb3 + 1, E = eb1 + eb2) is N/D, N = 4.094 B,
      // D = N + E A, and its derivative (N' - K1ss D')/D: one reciprocal (shared with IKb's) instead of three
      const double ea = fm.exp(0.1217 * (uK - 49.934));
      const double eb1 = 15.72 * fm.exp(0.0674 * (uK - 3.257)), eb2 = fm.exp(0.0618 * (uK - 594.31));
      const double eb3 = fm.exp(-0.1629 * (uK + 14.207));
      BEAT_TFENCE();
      const double ekb = fm.exp(-(v - 10.8968) * (1.0 / 23.9871));
      const double A1 = ea + 1.0, E1 = eb1 + eb2;
      const double N1 = 4.094 * (eb3 + 1.0), D1_ = N1 + E1 * A1;
      const double dN1 = (4.094 * -0.1629) * eb3;
      const double dD1 = dN1 + ((0.0674 * eb1 + 0.0618 * eb2) * A1 + E1 * (0.1217 * ea));
      double rD1_, xkb;
      rcp2(D1_, ekb + 1.0, rD1_, xkb);
      const double K1ss = N1 * rD1_, dK1ss = (dN1 - K1ss * dD1) * rD1_;
      const double gK1 = q.GK1s * K1ss;
      const double dIK1_du = q.GK1s * dK1ss * uK + gK1;
      BEAT_TFENCE();
      // IKb
      const double gKb = q.GKb * xkb;
      const double dIKb_dv = q.GKb * (xkb * (1.0 - xkb) * (1.0 / 23.9871)) * uK + gKb;
      BEAT_TFENCE();
      // IKr
      const double O_ = pf_O;
      const double gKr = O_ * q.GKrs;
      const double Iu = (gK1 + gKb + gKr + q.gKatp) * uK;  // IK1 + IKb + IKr + I_katp
      Iv += Iu;
      dIv += dIK1_du + dIKb_dv + gKr + q.gKatp;
      Iki += Iu;
      dIki += (dIK1_du + gKb + gKr + q.gKatp) * (-dEK);
      BEAT_PIN(Iv); BEAT_PIN(dIv); BEAT_PIN(Iki); BEAT_PIN(dIki);
      {
        const double C1 = pf_C1, C2 = pf_C2, C3 = pf_C3, I_ = pf_I;
        const double alpha = 0.1161 * fm.exp(0.299 * vfrt), alpha_2 = 0.0578 * fm.exp(0.971 * vfrt);
        BEAT_TFENCE();
        const double alpha_C2ToI = 5.2e-5 * fm.exp(1.525 * vfrt), alpha_i = 0.2533 * fm.exp(0.5953 * vfrt);
        BEAT_TFENCE();
        const double beta_ = 0.2442 * fm.exp(-1.604 * vfrt), beta_2 = 0.000349 * fm.exp(-1.062 * vfrt);
        const double beta_i = 0.06525 * fm.exp(-0.8209 * vfrt);
        BEAT_TFENCE();
        const double beta_ItoC2 = (alpha_C2ToI * (beta_2 * beta_i)) * beat_rcp(alpha_2 * alpha_i);
        const double a1_ = p[alpha_1_], b1_ = p[beta_1_];
        pf_xs1 = io.load(S_xs1), pf_xs2 = io.load(S_xs2);
        io.store(S_C1, advance(fm, C1, -C1 * (alpha_C2ToI + (alpha_2 + b1_)) + (I_ * beta_ItoC2 + (C2 * a1_ + O_ * beta_2)),
                               -(alpha_C2ToI + (alpha_2 + b1_)), dt));
        BEAT_TFENCE();
        io.store(S_C2, advance(fm, C2, -C2 * (a1_ + beta_) + (C1 * b1_ + C3 * alpha), -(a1_ + beta_), dt));
        io.store(S_C3, advance(fm, C3, C2 * beta_ - C3 * alpha, -alpha, dt));
        BEAT_TFENCE();
        io.store(S_I_, advance(fm, I_, -I_ * (beta_ItoC2 + beta_i) + (C1 * alpha_C2ToI + O_ * alpha_i), -(beta_ItoC2 + beta_i), dt));
        io.store(S_O_, advance(fm, O_, -O_ * (alpha_i + beta_2) + (C1 * alpha_2 + I_ * beta_i), -(alpha_i + beta_2), dt));
      }
    }
    BEAT_TFENCE();
    {
      // IKs: reversal potential with the Na permeability; KsCa depends on cai (no state's self-derivative sees that)
      const double xs1 = pf_xs1, xs2 = pf_xs2;
      // the four reciprocals of the block from one: 1/(PKNa nai + ki), KsCa's, the steady state's, and 1/txs1 = s/(817.3 s + 1)
      const double sx1 = 0.0002326 * fm.exp((v + 48.28) * (1.0 / 17.8)) + 0.001292 * fm.exp(-(v + 210.0) * (1.0 / 230.0));
      double rks, rKsCa, xsss, rtxs1;
      rcp4(p[PKNa_] * nai + ki, fm.exp(1.4 * fm.log(3.8e-5 * rcai)) + 1.0, fm.exp(-(v + 11.6) * (1.0 / 8.932)) + 1.0,
           817.3 * sx1 + 1.0, rks, rKsCa, xsss, rtxs1);
      rtxs1 *= sx1;
      BEAT_TFENCE();
      const double EKs = q.RTFk * fm.log(q.nao_ko_ks * rks);
      const double KsCa = 1.0 + 0.6 * rKsCa;
      const double gKs = xs2 * (xs1 * (q.GKs * KsCa));
      const double IKs = gKs * (v - EKs);
      Iv += IKs;
      dIv += gKs;
      Iki += IKs;
      dIki += gKs * (q.RTFk * rks);
      BEAT_PIN(Iv); BEAT_PIN(dIv); BEAT_PIN(Iki); BEAT_PIN(dIki);
      BEAT_TFENCE();
      const double rtxs2 = (0.01 * 0.0820849986238988) * E20 + 0.0193 * fm.exp(-(v + 66.54) * (1.0 / 31.0));
      pf_cajsr = io.load(S_cajsr), pf_cansr = io.load(S_cansr), pf_Jrel_np = io.load(S_Jrel_np), pf_Jrel_p = io.load(S_Jrel_p);
      if constexpr (LAND) {
        pf_XS = io.load(S_XS), pf_XW = io.load(S_XW), pf_CaTrpn = io.load(S_CaTrpn), pf_TmB = io.load(S_TmB);
        pf_Zetas = io.load(S_Zetas), pf_Zetaw = io.load(S_Zetaw), pf_Cd = io.load(S_Cd);
      }
      io.store(S_xs1, gate_b(fm, xs1, xsss, rtxs1, dt, small_g));
      io.store(S_xs2, gate(fm, xs2, xsss, rtxs2, dt));
    }
    BEAT_TFENCE();

    // ---- chloride currents and concentrations (.ode:604-608, 403-404) ---------------------------------------------------
    {
      double rcli, rclss, rjn, rsl;
      rcp4(cli, clss, cass + p[KdClCa_], cai + p[KdClCa_], rcli, rclss, rjn, rsl);
      const double ECl = q.RTFcl * fm.log(p[clo_] * rcli), EClss = q.RTFcl * fm.log(p[clo_] * rclss);
      const double g_junc = q.g_junc0 * cass * rjn;
      const double g_sl = q.g_sl0 * cai * rsl;
      const double IClCa_junc = g_junc * (v - EClss), IClCa_sl = g_sl * (v - ECl), IClb = p[GClb_] * (v - ECl);
      Iv += IClCa_junc + IClCa_sl + IClb;
      dIv += g_junc + g_sl + p[GClb_];
      const double rt = q.r_tauNa;
      const double JdiffCl = (clss - cli) * rt;  // the specification divides by tauNa, not tauCl
      // d ECl / d cli = -RTFcl / cli
      const double f_cli = q.cAF_myo * (IClCa_sl + IClb) + JdiffCl * q.vss_vmyo;
      const double J_cli = q.cAF_myo * (g_sl + p[GClb_]) * (q.RTFcl * rcli) - q.vss_vmyo * rt;
      const double f_clss = -JdiffCl + q.cAF_ss * IClCa_junc;
      const double J_clss = -rt + q.cAF_ss * g_junc * (q.RTFcl * rclss);
      io.store(S_cli, advance(fm, cli, f_cli, J_cli, dt));
      io.store(S_clss, advance(fm, clss, f_clss, J_clss, dt));
    }
    BEAT_TFENCE();

    // ---- membrane potential (.ode:600-604) ----------------------------------------------------------------------------
    double Istim = 0.0;
    double ru, rup;  // SERCA's reciprocals, formed with IpCa's (all three are functions of cai)
    {
      const double since = -p[i_Stim_Period_] * floor(-(p[i_Stim_Start_] - t) / p[i_Stim_Period_]) - p[i_Stim_Start_] + t;
      if (p[i_Stim_Start_] <= t && p[i_Stim_PulseDuration_] >= since) Istim = p[i_Stim_Amplitude_];
      // IpCa: sarcolemmal Ca pump
      double rp;
      rcp3(p[KmCap_] + cai, cai + 0.00092, (cai + 0.00092) - 0.00017, rp, ru, rup);
      const double IpCa = p[GpCa_] * cai * rp;
      Iv += IpCa;
      Icai += IpCa;
      dIcai += q.GpCa_Km * rp * rp;
      io.store(S_v, advance(fm, v, -(Istim + Iv), -dIv, dt));
    }
    BEAT_TFENCE();

    // ---- sodium and potassium (.ode:405-411) --------------------------------------------------------------------------
    {
      const double rtNa = q.r_tauNa, rtK = q.r_tauK;
      const double JdiffNa = (nass - nai) * rtNa, JdiffK = (kss - ki) * rtK;
      io.store(S_nai, advance(fm, nai, -q.cAF_myo * Inai + JdiffNa * q.vss_vmyo, -q.cAF_myo * dInai - q.vss_vmyo * rtNa, dt));
      io.store(S_nass, advance(fm, nass, -JdiffNa - q.cAF_ss * Inass, -rtNa - q.cAF_ss * dInass, dt));
      BEAT_TFENCE();
      io.store(S_ki, advance(fm, ki, -q.cAF_myo * (Iki + Istim) + JdiffK * q.vss_vmyo, -q.cAF_myo * dIki - q.vss_vmyo * rtK, dt));
      io.store(S_kss, advance(fm, kss, -JdiffK - q.cAF_ss * Ikss, -rtK - q.cAF_ss * dIkss, dt));
    }
    BEAT_TFENCE();

    // ---- calcium: SERCA, ryanodine receptor, translocation, buffers (.ode:398-402, 617-633) -----------------------------
    {
      const double cajsr = pf_cajsr, cansr = pf_cansr, Jrel_np = pf_Jrel_np, Jrel_p = pf_Jrel_p;
      const double rtCa = q.r_tauCa;
      const double Jdiff = (cass - cai) * rtCa;
      // SERCA
      const double cu = q.cu, cup = q.cup;
      const double Jupnp = cai * cu * ru, Jupp = cai * cup * rup;
      const double Jleak = q.Jleak_c * cansr;
      const double Jup = p[Jup_b_] * (-Jleak + (Jupnp * (1.0 - fp) + Jupp * fp));
      const double dJup_dcai = p[Jup_b_] * ((1.0 - fp) * cu * 0.00092 * ru * ru + fp * cup * (0.00092 - 0.00017) * rup * rup);
      // release
      const double Jrel = p[Jrel_b_] * (Jrel_np * (1.0 - fp) + Jrel_p * fp);
      const double dJrel_dcass = p[Jrel_b_] * (Jrel_p - Jrel_np) * dfp_dcass;
      const double Jtr = (cansr - cajsr) * (1.0 / 60.0);
      double rq;  // 1/(cajsr + kmcsqn), for the cajsr equation below
      {
        // 1/((cajsr_half/cajsr)^8 + 1) = cajsr^8/(cajsr_half^8 + cajsr^8);  tau_rel = max(bt/(1 + 0.0123/cajsr), 0.001),
        // 1/(1 + 0.0123/cajsr) = 1/irc with irc = (cajsr + 0.0123)/cajsr:  1/tau_rel = min(irc/bt, 1000) where irc > 0 and
        // 1000 where it is not (a load that has run NEGATIVE -- perturbed parameter sets do that for a while -- makes the
        // specification's quotient negative, and its max() then returns the floor) -- one reciprocal for the block
        // instead of six.  (A load of exactly zero gives finite values in the specification: kept away from the
        // reciprocal.)
        const double cj = fabs(cajsr) < 1.0e-150 ? 1.0e-150 : cajsr;
        const double c2 = cj * cj, c4 = c2 * c2, c8 = c4 * c4;
        double rcajsr, rh;
        rcp3(cj, q.half8 + c8, cajsr + p[kmcsqn_], rcajsr, rh, rq);
        const double rh8 = c8 * rh;
        const double Jrel_inf = q.relScale * ((ICaL_ss * (-q.a_rel)) * rh8);
        const double Jrel_infp = q.relScale * ((ICaL_ss * (-q.a_relp)) * rh8);
        const double irc = (cj + 0.0123) * rcajsr;
        const double rate_np = irc > 0.0 ? fmin(irc * q.r_bt, 1000.0) : 1000.0;
        const double rate_p = irc > 0.0 ? fmin(irc * q.r_btp, 1000.0) : 1000.0;
        io.store(S_Jrel_np, gate(fm, Jrel_np, Jrel_inf, rate_np, dt));
        io.store(S_Jrel_p, gate(fm, Jrel_p, Jrel_infp, rate_p, dt));
      }
      BEAT_TFENCE();
      if constexpr (LAND) {
        // ---- Land contraction model (ToRORd_dynCl_endo_Land.ode:682-722) ------------------------------------------------
        const double XS = pf_XS, XW = pf_XW, CaTrpn = pf_CaTrpn, TmB = pf_TmB, Zetas = pf_Zetas, Zetaw = pf_Zetaw, Cd = pf_Cd;
        // troponin: d CaTrpn/dt = ktrpn ((1000 cai / cat50)^ntrpn (1 - CaTrpn) - CaTrpn)
        const double pw = fm.exp(p[ntrpn_] * fm.log(cai * q.ca_scale));
        const double fTrpn = p[ktrpn_] * (pw * (1.0 - CaTrpn) - CaTrpn);
        io.store(S_CaTrpn, advance(fm, CaTrpn, fTrpn, -p[ktrpn_] * (pw + 1.0), dt));
        {  // cai: calmodulin buffering in Bcai, troponin as the flux J_TRPN = trpnmax dCaTrpn/dt
          const double rm = beat_rcp(cai + p[kmcmdn_]);
          const double bm = q.b_cmdn * rm * rm;
          const double B = beat_rcp(1.0 + bm);
          const double dB = B * B * (2.0 * bm * rm);
          const double inner = ((-q.cA2F_myo * Icai - Jup * q.vnsr_vmyo) + Jdiff * q.vss_vmyo) - p[trpnmax_] * fTrpn;
          const double dinner = ((-q.cA2F_myo * dIcai - dJup_dcai * q.vnsr_vmyo) - q.vss_vmyo * rtCa) -
                                p[trpnmax_] * p[ktrpn_] * (1.0 - CaTrpn) * p[ntrpn_] * pw * beat_rcp(cai);
          io.store(S_cai, advance(fm, cai, B * inner, dB * inner + B * dinner, dt));
        }
        BEAT_TFENCE();
        const double XU = ((1.0 - TmB) - XS) - XW;
        {  // tropomyosin: kb min(CaTrpn^(-ntm/2), 100) XU - ku CaTrpn^(ntm/2) TmB
          const double chalf = fm.exp((0.5 * p[ntm_]) * fm.log(CaTrpn));
          const double cinv = beat_rcp(chalf);
          const double cm = cinv < 100.0 ? cinv : 100.0;
          io.store(S_TmB, advance(fm, TmB, q.kb * cm * XU - p[ku_] * chalf * TmB, -q.kb * cm - p[ku_] * chalf, dt));
        }
        {  // cross-bridges; the distortion-dependent unbinding rates use relations as numbers (.ode:689-690)
          const double zp = Zetas > 0.0 ? Zetas : 0.0, zn = Zetas < -1.0 ? -Zetas - 1.0 : 0.0;
          const double gsu = p[gammas_] * (zp > zn ? zp : zn), gwu = p[gammaw_] * fabs(Zetaw);
          io.store(S_XS, advance(fm, XS, (p[kws_] * XW - q.ksu * XS) - gsu * XS, -q.ksu - gsu, dt));
          // kuw XU - (kwu + kws + gwu) XW; the self-derivative also sees XU = 1 - TmB - XS - XW
          io.store(S_XW, advance(fm, XW, p[kuw_] * XU - ((q.kwu_kws_kuw - p[kuw_]) + gwu) * XW, -q.kwu_kws_kuw - gwu, dt));
        }
        io.store(S_Zetas, advance(fm, Zetas, q.Aw_dL - q.cs * Zetas, -q.cs, dt));
        io.store(S_Zetaw, advance(fm, Zetaw, q.Aw_dL - q.cw * Zetaw, -q.cw, dt));
        {  // dashpot: p_k (C - Cd) / eta, eta by the sign of C - Cd
          const double dCd = q.Cdash - Cd;
          const double re = p[p_k_] * beat_rcp(dCd < 0.0 ? p[etas_] : p[etal_]);
          io.store(S_Cd, advance(fm, Cd, re * dCd, -re, dt));
        }
      } else {  // cai: d/dt = Bcai * inner
        double rt, rm;
        rcp2(cai + p[kmtrpn_], cai + p[kmcmdn_], rt, rm);
        const double bt_ = q.b_trpn * rt * rt, bm = q.b_cmdn * rm * rm;
        const double B = beat_rcp(bt_ + (bm + 1.0));
        const double dB = B * B * (2.0 * bt_ * rt + 2.0 * bm * rm);
        const double inner = (-q.cA2F_myo * Icai - Jup * q.vnsr_vmyo) + Jdiff * q.vss_vmyo;
        const double dinner = (-q.cA2F_myo * dIcai - dJup_dcai * q.vnsr_vmyo) - q.vss_vmyo * rtCa;
        io.store(S_cai, advance(fm, cai, B * inner, dB * inner + B * dinner, dt));
      }
      BEAT_TFENCE();
      {  // cass
        double rl, rr;
        rcp2(p[KmBSL_] + cass, p[KmBSR_] + cass, rl, rr);
        const double bl = q.b_bsl * rl * rl, br = q.b_bsr * rr * rr;
        const double B = beat_rcp(bl + (br + 1.0));
        const double dB = B * B * (2.0 * bl * rl + 2.0 * br * rr);
        const double inner = -Jdiff + (-q.cA2F_ss * Icass + Jrel * q.vjsr_vss);
        const double dinner = -rtCa + (-q.cA2F_ss * dIcass + dJrel_dcass * q.vjsr_vss);
        io.store(S_cass, advance(fm, cass, B * inner, dB * inner + B * dinner, dt));
      }
      BEAT_TFENCE();
      {  // cajsr
        const double bq = q.b_csqn * rq * rq;
        const double B = beat_rcp(bq + 1.0);
        const double dB = B * B * (2.0 * bq * rq);
        const double inner = -Jrel + Jtr;
        io.store(S_cajsr, advance(fm, cajsr, B * inner, dB * inner - B * (1.0 / 60.0), dt));
      }
      BEAT_TFENCE();
      // cansr
      io.store(S_cansr, advance(fm, cansr, Jup - Jtr * q.vjsr_vnsr, q.J_cansr, dt));
    }
  }
};
using TorordDynClGrl1 = TorordGrl1T<false>;
using TorordLandGrl1 = TorordGrl1T<true>;

#if defined(__clang__) && !defined(BEAT_TORORD_NO_CONTRACT)
#pragma clang fp contract(off)
#endif

