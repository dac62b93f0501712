"""ORACLE (test infrastructure only -- never imported by the product path).

Hand restatement of the ToR-ORd-dynCl ventricular cell model (45 states, 112 parameters) that the
reference's ventricular demos advance with gotranx's first-order generalized Rush-Larsen scheme:

* specification: odes/torord/ToRORd_dynCl_endo.ode (states :1-93, parameters :95-301,
  expressions :303-633), used by demos/biv_endocardial.py:124-173,187-282 (one parameter set per
  cell type: ``celltype`` = 0 endo, 1 epi, 2 mid);
* scheme: ``generalized_rush_larsen`` as restated in oracle/ionic.py (gotranx is an un-vendored,
  un-pinned dependency, pyproject.toml:57-64): y_i += f_i (exp(J_i dt) - 1)/J_i where |J_i| > 1e-8,
  dt f_i elsewhere, J_i = d f_i / d y_i with every intermediate resolved (the variant the
  reference's Niederer table pins for TP06, tests/test_oracle_pins.py).

This file is written by hand from the specification, in dependency order, and is deliberately
independent of tests/golden/ode_spec.py (the ``ast`` walker that produced the committed fixture
tests/golden/torord_spec.npz and that tools/gen_model_data.py shares): it does
not parse the ``.ode`` file, it does not use SymPy, and its self-derivatives come from forward-mode
automatic differentiation with NumPy dual numbers (one pass per state), not from symbolic
differentiation.  tests/test_oracle_pins.py checks it against that fixture (names, defaults, RHS,
J, GRL1 step for the three cell types and along a paced action potential), so a front-end mistake
in the generator chain and a transcription mistake here cannot both go unnoticed.

PARITY UNPINNED for individual gotranx GRL1 step outputs: the reference holds none.
"""

from __future__ import annotations

import numpy as np

# ------------------------------------------------------------------------------------------------
# names and defaults, in order of appearance in the .ode file
# ------------------------------------------------------------------------------------------------
TORORD_STATE_DEFAULTS = dict(
    # IKr (.ode:1-7)
    C1=0.9982511, C2=7.93602000000000023e-4, C3=6.53214300000000045e-4, I_=9.80408300000000003e-6,
    O_=2.92244900000000025e-4,
    # CaMK (.ode:9-11)
    CaMKt=1.09502599999999999e-2,
    # ryr (.ode:13-16)
    Jrel_np=1.80824799999999996e-22, Jrel_p=4.35860800000000030e-21,
    # Ito (.ode:18-25)
    a=8.89925900000000051e-4, ap=4.53416500000000005e-4, iF=0.9996716, iFp=0.9996716, iS=0.5988908,
    iSp=0.6620692,
    # intracellular ions (.ode:27-38)
    cai=7.45348100000000041e-5, cajsr=1.525693, cansr=1.528001, cass=6.49734100000000044e-5,
    cli=29.20698, clss=29.20696, ki=147.7115, kss=147.7114, nai=12.39736, nass=12.3977,
    # ICaL (.ode:40-51)
    d=1.58884100000000000e-31, fcaf=1.0, fcafp=1.0, fcas=0.9999014, ff_=1.0, ffp=1.0, fs=0.9401791,
    jca=0.9999846, nca_i=8.32600900000000053e-4, nca_ss=4.89937800000000024e-4,
    # INa (.ode:53-59)
    h=0.8473267, hp=0.7018454, j=0.8471657, jp=0.8469014, m=6.51715400000000005e-4,
    # INaL (.ode:61-65)
    hL=0.5566017, hLp=0.3115491, mL=1.35120299999999988e-4,
    # membrane (.ode:67-69)
    v=-89.74808,
    # IKs (.ode:71-74)
    xs1=0.243959, xs2=1.58616700000000009e-4,
)
TORORD_STATES = tuple(TORORD_STATE_DEFAULTS)

TORORD_PARAMETER_DEFAULTS = dict(
    # I_katp (.ode:76-82)
    A_atp=2.0, K_atp=0.25, K_o_n=5.0, fkatp=0.0, gkatp=4.3195,
    # ICaL (.ode:84-94)
    Aff=0.6, ICaL_fractionSS=0.8, Kmn=0.002, PCa_b=8.3757e-05, dielConstant=74.0, k2n=500.0, offset=0.0,
    tjca=72.5, vShift=0.0,
    # intracellular ions (.ode:96-107)
    BSLmax=1.124, BSRmax=0.047, KmBSL=0.0087, KmBSR=0.00087, cmdnmax_b=0.05, csqnmax=10.0, kmcmdn=0.00238,
    kmcsqn=0.8, kmtrpn=0.0005, trpnmax=0.07,
    # CaMK (.ode:109-115)
    CaMKo=0.05, KmCaM=0.0015, KmCaMK=0.15, aCaMK=0.05, bCaMK=0.00068,
    # Ito (.ode:117-120)
    EKshift=0.0, Gto_b=0.16,
    # physical constants (.ode:122-130)
    F=96485.0, R=8314.0, T=310.0, zca=2.0, zcl=-1.0, zk=1.0, zna=1.0,
    # ICl (.ode:132-137)
    Fjunc=1.0, GClCa=0.2843, GClb=0.00198, KdClCa=0.1,
    # IK1, IKb, IKr, IKs, INa, INaL (.ode:139-169)
    GK1_b=0.6992, GKb_b=0.0189, GKr_b=0.0321, alpha_1=0.154375, beta_1=0.1911, GKs_b=0.0011, GNa=11.7802,
    GNaL_b=0.0279, thL=200.0,
    # INaCa (.ode:171-186)
    Gncx_b=0.0034, INaCa_fractionSS=0.35, KmCaAct=0.00015, kasymm=12.5, kcaoff=5000.0, kcaon=1500000.0,
    kna1=15.0, kna2=5.0, kna3=88.12, qca=0.167, qna=0.5224, wca=60000.0, wna=60000.0, wnaca=5000.0,
    # IpCa (.ode:188-191)
    GpCa=0.0005, KmCap=0.0005,
    # INaK (.ode:193-216)
    H=1e-07, Khp=1.698e-07, Kki=0.5, Kko=0.3582, Kmgatp=1.698e-07, Knai0=9.073, Knao0=27.78, Knap=224.0,
    Kxkur=292.0, MgADP=0.05, MgATP=9.8, Pnak_b=15.4509, delta=-0.155, eP=4.2, k1m=182.4, k1p=949.5, k2m=39.4,
    k2p=687.2, k3m=79300.0, k3p=1899.0, k4m=40.0, k4p=639.0,
    # ryr, SERCA, cell geometry, ICab, reversal potentials, INab (.ode:218-249)
    Jrel_b=1.5378, bt=4.75, cajsr_half=1.7, Jup_b=1.0, L=0.01, rad_=0.0011, PCab=5.9194e-08, PKNa=0.01833,
    PNab=1.9239e-09,
    # extracellular, environment (.ode:251-260)
    cao=1.8, clo=150.0, ko=5.0, nao=140.0, celltype=0.0,
    # membrane (.ode:262-268)
    i_Stim_Amplitude=-53.0, i_Stim_End=1e17, i_Stim_Period=1000.0, i_Stim_PulseDuration=1.0, i_Stim_Start=0.0,
    # diff (.ode:270-275)
    tauCa=0.2, tauCl=2.0, tauK=2.0, tauNa=2.0,
)
TORORD_PARAMETERS = tuple(TORORD_PARAMETER_DEFAULTS)


def torord_state_index(name: str) -> int:
    return TORORD_STATES.index(name)


def torord_parameter_index(name: str) -> int:
    return TORORD_PARAMETERS.index(name)


def torord_init_state_values(**values) -> np.ndarray:
    d = dict(TORORD_STATE_DEFAULTS)
    for k, v in values.items():
        if k not in d:
            raise KeyError(k)
        d[k] = v
    return np.array([d[k] for k in TORORD_STATES], dtype=np.float64)


def torord_init_parameter_values(**values) -> np.ndarray:
    d = dict(TORORD_PARAMETER_DEFAULTS)
    for k, v in values.items():
        if k not in d:
            raise KeyError(k)
        d[k] = v
    return np.array([d[k] for k in TORORD_PARAMETERS], dtype=np.float64)


# ------------------------------------------------------------------------------------------------
# forward-mode automatic differentiation: value + one directional derivative, NumPy arrays
# ------------------------------------------------------------------------------------------------
class Dual:
    """a + b eps with eps^2 = 0; ``a`` and ``b`` broadcastable NumPy arrays (or floats)."""

    __slots__ = ("a", "b")
    __array_priority__ = 1000.0
    __array_ufunc__ = None  # ndarray / NumPy scalar (op) Dual defers to Dual.__r(op)__

    def __init__(self, a, b=0.0):
        self.a = a
        self.b = b

    @staticmethod
    def _split(x):
        return (x.a, x.b) if isinstance(x, Dual) else (x, 0.0)

    def __neg__(self):
        return Dual(-self.a, -self.b)

    def __pos__(self):
        return self

    def __add__(self, o):
        oa, ob = Dual._split(o)
        return Dual(self.a + oa, self.b + ob)

    __radd__ = __add__

    def __sub__(self, o):
        oa, ob = Dual._split(o)
        return Dual(self.a - oa, self.b - ob)

    def __rsub__(self, o):
        oa, ob = Dual._split(o)
        return Dual(oa - self.a, ob - self.b)

    def __mul__(self, o):
        oa, ob = Dual._split(o)
        return Dual(self.a * oa, self.a * ob + self.b * oa)

    __rmul__ = __mul__

    def __truediv__(self, o):
        oa, ob = Dual._split(o)
        q = self.a / oa
        return Dual(q, (self.b - q * ob) / oa)

    def __rtruediv__(self, o):
        oa, ob = Dual._split(o)
        q = oa / self.a
        return Dual(q, (ob - q * self.b) / self.a)

    def __pow__(self, e):
        if isinstance(e, Dual):
            raise TypeError("only constant exponents occur in the model")
        return Dual(self.a**e, e * self.a ** (e - 1.0) * self.b)

    # comparisons act on the values (the model's conditions select branches, they are not differentiated)
    def __lt__(self, o):
        return self.a < Dual._split(o)[0]

    def __le__(self, o):
        return self.a <= Dual._split(o)[0]

    def __gt__(self, o):
        return self.a > Dual._split(o)[0]

    def __ge__(self, o):
        return self.a >= Dual._split(o)[0]


def _val(x):
    return x.a if isinstance(x, Dual) else x


def _exp(x):
    if isinstance(x, Dual):
        e = np.exp(x.a)
        return Dual(e, e * x.b)
    return np.exp(x)


def _log(x):
    if isinstance(x, Dual):
        return Dual(np.log(x.a), x.b / x.a)
    return np.log(x)


def _sqrt(x):
    if isinstance(x, Dual):
        s = np.sqrt(x.a)
        return Dual(s, 0.5 * x.b / s)
    return np.sqrt(x)


def _abs(x):
    if isinstance(x, Dual):
        return Dual(np.abs(x.a), np.sign(x.a) * x.b)
    return np.abs(x)


def _where(c, a, b):
    if isinstance(a, Dual) or isinstance(b, Dual):
        aa, ab = Dual._split(a)
        ba, bb = Dual._split(b)
        return Dual(np.where(c, aa, ba), np.where(c, ab, bb))
    return np.where(c, a, b)


# ------------------------------------------------------------------------------------------------
# right-hand side (.ode:303-633), written once for floats / arrays / Dual numbers
# ------------------------------------------------------------------------------------------------
def torord_rhs(states, t, parameters, land=None):
    """dy/dt of the 45 states in TORORD_STATES order.  ``states``: sequence of 45 arrays or Dual numbers;
    ``parameters``: sequence of 112 floats or arrays (per-node parameters broadcast against the states).
    ``land``: None, or the 7 extra states followed by the 28 extra parameters of the Land variant (see
    torord_land_rhs): the calcium equation changes and the 7 extra derivatives are appended."""
    exp, log, sqrt, where = _exp, _log, _sqrt, _where
    (C1, C2, C3, I_, O_, CaMKt, Jrel_np, Jrel_p, a, ap, iF, iFp, iS, iSp, cai, cajsr, cansr, cass, cli, clss, ki,
     kss, nai, nass, d, fcaf, fcafp, fcas, ff_, ffp, fs, jca, nca_i, nca_ss, h, hp, j, jp, m, hL, hLp, mL, v, xs1,
     xs2) = states
    (A_atp, K_atp, K_o_n, fkatp, gkatp, Aff, ICaL_fractionSS, Kmn, PCa_b, dielConstant, k2n, offset, tjca, vShift,
     BSLmax, BSRmax, KmBSL, KmBSR, cmdnmax_b, csqnmax, kmcmdn, kmcsqn, kmtrpn, trpnmax, CaMKo, KmCaM, KmCaMK, aCaMK,
     bCaMK, EKshift, Gto_b, F, R, T, zca, zcl, zk, zna, Fjunc, GClCa, GClb, KdClCa, GK1_b, GKb_b, GKr_b, alpha_1,
     beta_1, GKs_b, GNa, GNaL_b, thL, Gncx_b, INaCa_fractionSS, KmCaAct, kasymm, kcaoff, kcaon, kna1, kna2, kna3,
     qca, qna, wca, wna, wnaca, GpCa, KmCap, H, Khp, Kki, Kko, Kmgatp, Knai0, Knao0, Knap, Kxkur, MgADP, MgATP,
     Pnak_b, delta, eP, k1m, k1p, k2m, k2p, k3m, k3p, k4m, k4p, Jrel_b, bt, cajsr_half, Jup_b, L, rad_, PCab, PKNa,
     PNab, cao, clo, ko, nao, celltype, i_Stim_Amplitude, i_Stim_End, i_Stim_Period, i_Stim_PulseDuration,
     i_Stim_Start, tauCa, tauCl, tauK, tauNa) = parameters
    epi = celltype == 1.0
    mid = celltype == 2.0

    # cell geometry (.ode:303-310)
    Ageo = L * ((2.0 * 3.14) * rad_) + rad_ * ((2.0 * 3.14) * rad_)
    Acap = 2.0 * Ageo
    vcell = L * (rad_ * ((1000.0 * 3.14) * rad_))
    vjsr = 0.0048 * vcell
    vmyo = 0.68 * vcell
    vnsr = 0.0552 * vcell
    vss = 0.02 * vcell

    # membrane helpers (.ode:601-603)
    vffrt = (F * (F * v)) / (R * T)
    vfrt = (F * v) / (R * T)

    # CaMK (.ode:413-416)
    CaMKb = (CaMKo * (1.0 - CaMKt)) / (KmCaM / cass + 1.0)
    CaMKa = CaMKb + CaMKt
    dCaMKt_dt = -CaMKt * bCaMK + (CaMKb * aCaMK) * (CaMKb + CaMKt)
    fCaMKp = 1.0 / (1.0 + KmCaMK / CaMKa)  # = fICaLp = fItop = fINaLp = fINap = fJupp = fJrelp (all written alike)

    # reversal potentials (.ode:527-532)
    ECl = ((R * T) / (F * zcl)) * log(clo / cli)
    EClss = ((R * T) / (F * zcl)) * log(clo / clss)
    EK = ((R * T) / (F * zk)) * log(ko / ki)
    EKs = ((R * T) / (F * zk)) * log((PKNa * nao + ko) / (PKNa * nai + ki))
    ENa = ((R * T) / (F * zna)) * log(nao / nai)

    # ---- ICaL (.ode:312-377) ---------------------------------------------------------------------
    Afcaf = 0.3 + 0.6 / (exp((v - 10.0) / 10.0) + 1.0)
    Afcas = 1.0 - Afcaf
    Afs = 1.0 - Aff
    Ii = (0.5 * (4.0 * cai + (cli + (ki + nai)))) / 1000.0
    Io = (0.5 * (4.0 * cao + (clo + (ko + nao)))) / 1000.0
    Iss = (0.5 * (4.0 * cass + (clss + (kss + nass)))) / 1000.0
    PCa = where(epi, 1.2 * PCa_b, where(mid, 2.0 * PCa_b, PCa_b))
    PCap = 1.1 * PCa
    PCaK = 0.0003574 * PCa
    PCaKp = 0.0003574 * PCap
    PCaNa = 0.00125 * PCa
    PCaNap = 0.00125 * PCap
    constA = 1820000.0 / (T * dielConstant) ** 1.5

    def activity(I, z2):
        return exp((-constA * z2) * (sqrt(I) / (sqrt(I) + 1.0) - 0.3 * I))

    gamma_cai, gamma_cao, gamma_cass = activity(Ii, 4.0), activity(Io, 4.0), activity(Iss, 4.0)
    gamma_ki, gamma_ko, gamma_kss = activity(Ii, 1.0), activity(Io, 1.0), activity(Iss, 1.0)
    gamma_nai, gamma_nao, gamma_nass = activity(Ii, 1.0), activity(Io, 1.0), activity(Iss, 1.0)
    e1 = exp(1.0 * vfrt)
    e2 = exp(2.0 * vfrt)
    PhiCaK_i = ((1.0 * vffrt) * (-gamma_ko * ko + (gamma_ki * ki) * e1)) / (e1 - 1.0)
    PhiCaK_ss = ((1.0 * vffrt) * (-gamma_ko * ko + (gamma_kss * kss) * e1)) / (e1 - 1.0)
    PhiCaL_i = ((4.0 * vffrt) * (-cao * gamma_cao + (cai * gamma_cai) * e2)) / (e2 - 1.0)
    PhiCaL_ss = ((4.0 * vffrt) * (-cao * gamma_cao + (cass * gamma_cass) * e2)) / (e2 - 1.0)
    PhiCaNa_i = ((1.0 * vffrt) * (-gamma_nao * nao + (gamma_nai * nai) * e1)) / (e1 - 1.0)
    PhiCaNa_ss = ((1.0 * vffrt) * (-gamma_nao * nao + (gamma_nass * nass) * e1)) / (e1 - 1.0)
    km2n = jca * 1.0
    anca_i = 1.0 / (k2n / km2n + (Kmn / cai + 1.0) ** 4.0)
    anca_ss = 1.0 / (k2n / km2n + (Kmn / cass + 1.0) ** 4.0)
    dss = where(v >= 31.4978, 1.0, 1.0763 * exp(-1.007 * exp(-0.0829 * v)))
    fss = 1.0 / (exp((v + 19.58) / 3.696) + 1.0)
    fcass = fss
    f = Aff * ff_ + Afs * fs
    fp = Aff * ffp + Afs * fs
    fca = Afcaf * fcaf + Afcas * fcas
    fcap = Afcaf * fcafp + Afcas * fcas
    fICaLp = fCaMKp
    jcass = 1.0 / (exp((v + 18.08) / 2.7916) + 1.0)
    td = (offset + 0.6) + 1.0 / (exp(-0.05 * ((v + vShift) + 6.0)) + exp(0.09 * ((v + vShift) + 14.0)))
    tfcaf = 7.0 + 1.0 / (0.04 * exp((-(v - 4.0)) / 7.0) + 0.04 * exp((v - 4.0) / 7.0))
    tfcafp = 2.5 * tfcaf
    tfcas = 100.0 + 1.0 / (0.00012 * exp((-v) / 3.0) + 0.00012 * exp(v / 7.0))
    tff = 7.0 + 1.0 / (0.0045 * exp((-(v + 20.0)) / 10.0) + 0.0045 * exp((v + 20.0) / 10.0))
    tffp = 2.5 * tff
    tfs = 1000.0 + 1.0 / (3.5e-5 * exp((-(v + 5.0)) / 4.0) + 3.5e-5 * exp((v + 5.0) / 6.0))

    def ical(frac, Phi, P, Pp, nca):  # the six channel currents share this form (.ode:316-325)
        return frac * ((d * (Phi * (P * (1.0 - fICaLp)))) * (f * (1.0 - nca) + nca * (fca * jca))
                       + (d * (Phi * (Pp * fICaLp))) * (fp * (1.0 - nca) + nca * (fcap * jca)))

    ICaL_i = ical(1.0 - ICaL_fractionSS, PhiCaL_i, PCa, PCap, nca_i)
    ICaL_ss = ical(ICaL_fractionSS, PhiCaL_ss, PCa, PCap, nca_ss)
    ICaNa_i = ical(1.0 - ICaL_fractionSS, PhiCaNa_i, PCaNa, PCaNap, nca_i)
    ICaNa_ss = ical(ICaL_fractionSS, PhiCaNa_ss, PCaNa, PCaNap, nca_ss)
    ICaK_i = ical(1.0 - ICaL_fractionSS, PhiCaK_i, PCaK, PCaKp, nca_i)
    ICaK_ss = ical(ICaL_fractionSS, PhiCaK_ss, PCaK, PCaKp, nca_ss)
    ICaL_ICaL = ICaL_i + ICaL_ss
    ICaNa = ICaNa_i + ICaNa_ss
    ICaK = ICaK_i + ICaK_ss
    dd_dt = (-d + dss) / td
    dfcaf_dt = (-fcaf + fcass) / tfcaf
    dfcafp_dt = (-fcafp + fcass) / tfcafp
    dfcas_dt = (-fcas + fcass) / tfcas
    dff__dt = (-ff_ + fss) / tff
    dffp_dt = (-ffp + fss) / tffp
    dfs_dt = (-fs + fss) / tfs
    djca_dt = (-jca + jcass) / tjca
    dnca_i_dt = anca_i * k2n - km2n * nca_i
    dnca_ss_dt = anca_ss * k2n - km2n * nca_ss

    # ---- Ito (.ode:379-403) ----------------------------------------------------------------------
    ve = EKshift + v
    AiF = 1.0 / (exp((ve - 213.6) / 151.2) + 1.0)
    AiS = 1.0 - AiF
    Gto = where(np.logical_or(epi, mid), 2.0 * Gto_b, Gto_b)
    ass = 1.0 / (exp((-(ve - 14.34)) / 14.82) + 1.0)
    assp = 1.0 / (exp((-(ve - 24.34)) / 14.82) + 1.0)
    delta_epi = where(epi, 1.0 - 0.95 / (exp((ve + 70.0) / 5.0) + 1.0), 1.0)
    dti_develop = 1.354 + 0.0001 / (exp((-(ve - 12.23)) / 0.2154) + exp((ve - 167.4) / 15.89))
    dti_recover = 1.0 - 0.5 / (exp((ve + 70.0) / 20.0) + 1.0)
    fItop = fCaMKp
    i_ = AiF * iF + AiS * iS
    ip = AiF * iFp + AiS * iSp
    iss = 1.0 / (exp((ve + 43.94) / 5.711) + 1.0)
    ta = 1.0515 / (1.0 / (1.2089 * (exp((-(ve - 18.4099)) / 29.3814) + 1.0)) + 3.5 / (exp((ve + 100.0) / 29.3814) + 1.0))
    tiF_b = 4.562 + 1.0 / (0.3933 * exp((-(ve + 100.0)) / 100.0) + 0.08004 * exp((ve + 50.0) / 16.59))
    tiS_b = 23.62 + 1.0 / (0.001416 * exp((-(ve + 96.52)) / 59.05) + 1.78e-8 * exp((ve + 114.1) / 8.079))
    tiF = delta_epi * tiF_b
    tiS = delta_epi * tiS_b
    tiFp = tiF * (dti_develop * dti_recover)
    tiSp = tiS * (dti_develop * dti_recover)
    Ito_Ito = (Gto * (-EK + v)) * (i_ * (a * (1.0 - fItop)) + ip * (ap * fItop))
    da_dt = (-a + ass) / ta
    dap_dt = (-ap + assp) / ta
    diF_dt = (-iF + iss) / tiF
    diFp_dt = (-iFp + iss) / tiFp
    diS_dt = (-iS + iss) / tiS
    diSp_dt = (-iSp + iss) / tiSp

    # ---- INaK (.ode:418-444) ---------------------------------------------------------------------
    Knai = Knai0 * exp((delta * vfrt) / 3.0)
    Knao = Knao0 * exp((vfrt * (1.0 - delta)) / 3.0)
    P = eP / (((H / Khp + 1.0) + nai / Knap) + ki / Kxkur)
    Pnak = where(epi, 0.9 * Pnak_b, where(mid, 0.7 * Pnak_b, Pnak_b))
    a1 = (k1p * (nai / Knai) ** 3.0) / (((1.0 + ki / Kki) ** 2.0 + (1.0 + nai / Knai) ** 3.0) - 1.0)
    a2 = k2p
    a3 = (k3p * (ko / Kko) ** 2.0) / (((1.0 + ko / Kko) ** 2.0 + (1.0 + nao / Knao) ** 3.0) - 1.0)
    a4 = ((MgATP * k4p) / Kmgatp) / (1.0 + MgATP / Kmgatp)
    b1 = MgADP * k1m
    b2 = (k2m * (nao / Knao) ** 3.0) / (((1.0 + ko / Kko) ** 2.0 + (1.0 + nao / Knao) ** 3.0) - 1.0)
    b3 = (H * (P * k3m)) / (1.0 + MgATP / Kmgatp)
    b4 = (k4m * (ki / Kki) ** 2.0) / (((1.0 + ki / Kki) ** 2.0 + (1.0 + nai / Knai) ** 3.0) - 1.0)
    x1 = a2 * (a1 * b3) + (b3 * (a2 * b4) + (a2 * (a1 * a4) + b3 * (b2 * b4)))
    x2 = b4 * (a2 * a3) + (b4 * (a3 * b1) + (a3 * (a1 * a2) + b4 * (b1 * b2)))
    x3 = b1 * (a3 * a4) + (a4 * (b1 * b2) + (a4 * (a2 * a3) + b1 * (b2 * b3)))
    x4 = a1 * (b2 * b3) + (a1 * (a4 * b2) + (a1 * (a3 * a4) + b2 * (b3 * b4)))
    xsum = x4 + (x3 + (x1 + x2))
    E1_ = x1 / xsum
    E2 = x2 / xsum
    E3 = x3 / xsum
    E4 = x4 / xsum
    JnakK = 2.0 * (-E3 * a1 + E4 * b1)
    JnakNa = 3.0 * (E1_ * a3 - E2 * b3)
    INaK_INaK = Pnak * (JnakK * zk + JnakNa * zna)

    # ---- INaCa (.ode:446-525) --------------------------------------------------------------------
    Gncx = where(epi, 1.1 * Gncx_b, where(mid, 1.4 * Gncx_b, Gncx_b))
    hca = exp(qca * vfrt)
    hna = exp(qna * vfrt)

    def ncx(na, ca, frac):
        allo = 1.0 / ((KmCaAct / ca) ** 2.0 + 1.0)
        h10 = (nao / kna1) * (1.0 + nao / kna2) + (kasymm + 1.0)
        h11 = (nao * nao) / (kna2 * (h10 * kna1))
        h12 = 1.0 / h10
        h1 = (na / kna3) * (hna + 1.0) + 1.0
        h2 = (hna * na) / (h1 * kna3)
        h3 = 1.0 / h1
        h4 = (na / kna1) * (1.0 + na / kna2) + 1.0
        h5 = (na * na) / (kna2 * (h4 * kna1))
        h6 = 1.0 / h4
        h7 = (nao / kna3) * (1.0 + 1.0 / hna) + 1.0
        h8 = nao / (h7 * (hna * kna3))
        h9 = 1.0 / h7
        k1 = kcaon * (cao * h12)
        k2 = kcaoff
        k3p_ = h9 * wca
        k3pp = h8 * wnaca
        k3 = k3p_ + k3pp
        k4p_ = (h3 * wca) / hca
        k4pp = h2 * wnaca
        k4 = k4p_ + k4pp
        k5 = kcaoff
        k6 = kcaon * (ca * h6)
        k7 = wna * (h2 * h5)
        k8 = wna * (h11 * h8)
        y1 = (k2 * k4) * (k6 + k7) + (k5 * k7) * (k2 + k3)
        y2 = (k1 * k7) * (k4 + k5) + (k4 * k6) * (k1 + k8)
        y3 = (k1 * k3) * (k6 + k7) + (k6 * k8) * (k2 + k3)
        y4 = (k2 * k8) * (k4 + k5) + (k3 * k5) * (k1 + k8)
        ysum = y4 + (y3 + (y1 + y2))
        E1n, E2n, E3n, E4n = y1 / ysum, y2 / ysum, y3 / ysum, y4 / ysum
        JncxCa = -E1n * k1 + E2n * k2
        JncxNa = -E2n * k3pp + (E3n * k4pp + 3.0 * (-E1n * k8 + E4n * k7))
        return (allo * (Gncx * frac)) * (JncxCa * zca + JncxNa * zna)

    INaCa_i = ncx(nai, cai, 1.0 - INaCa_fractionSS)
    INaCa_ss = ncx(nass, cass, INaCa_fractionSS)

    # ---- IK1, IKb, IKr, IKs (.ode:534-575) ----------------------------------------------------------
    GK1 = where(epi, 1.2 * GK1_b, where(mid, 1.3 * GK1_b, GK1_b))
    u = -EK + v
    aK1 = 4.094 / (exp(0.1217 * (u - 49.934)) + 1.0)
    bK1 = (15.72 * exp(0.0674 * (u - 3.257)) + exp(0.0618 * (u - 594.31))) / (exp(-0.1629 * (u + 14.207)) + 1.0)
    K1ss = aK1 / (aK1 + bK1)
    IK1_IK1 = (K1ss * (GK1 * sqrt(ko / 5.0))) * u
    GKb = where(epi, 0.6 * GKb_b, GKb_b)
    xkb = 1.0 / (exp((-(v - 10.8968)) / 23.9871) + 1.0)
    IKb_IKb = (GKb * xkb) * u
    GKr = where(epi, 1.3 * GKr_b, where(mid, 0.8 * GKr_b, GKr_b))
    IKr_IKr = (O_ * (GKr * sqrt(ko / 5.0))) * u
    alpha = 0.1161 * exp(0.299 * vfrt)
    alpha_2 = 0.0578 * exp(0.971 * vfrt)
    alpha_C2ToI = 5.2e-5 * exp(1.525 * vfrt)
    alpha_i = 0.2533 * exp(0.5953 * vfrt)
    beta_ = 0.2442 * exp(-1.604 * vfrt)
    beta_2 = 0.000349 * exp(-1.062 * vfrt)
    beta_i = 0.06525 * exp(-0.8209 * vfrt)
    beta_ItoC2 = (alpha_C2ToI * (beta_2 * beta_i)) / (alpha_2 * alpha_i)
    dC1_dt = -C1 * (alpha_C2ToI + (alpha_2 + beta_1)) + (I_ * beta_ItoC2 + (C2 * alpha_1 + O_ * beta_2))
    dC2_dt = -C2 * (alpha_1 + beta_) + (C1 * beta_1 + C3 * alpha)
    dC3_dt = C2 * beta_ - C3 * alpha
    dI__dt = -I_ * (beta_ItoC2 + beta_i) + (C1 * alpha_C2ToI + O_ * alpha_i)
    dO__dt = -O_ * (alpha_i + beta_2) + (C1 * alpha_2 + I_ * beta_i)
    GKs = where(epi, 1.4 * GKs_b, GKs_b)
    KsCa = 1.0 + 0.6 / ((3.8e-5 / cai) ** 1.4 + 1.0)
    IKs_IKs = (xs2 * (xs1 * (GKs * KsCa))) * (-EKs + v)
    txs1 = 817.3 + 1.0 / (0.0002326 * exp((v + 48.28) / 17.8) + 0.001292 * exp((-(v + 210.0)) / 230.0))
    txs2 = 1.0 / (0.01 * exp((v - 50.0) / 20.0) + 0.0193 * exp((-(v + 66.54)) / 31.0))
    xs1ss = 1.0 / (exp((-(v + 11.6)) / 8.932) + 1.0)
    xs2ss = xs1ss
    dxs1_dt = (-xs1 + xs1ss) / txs1
    dxs2_dt = (-xs2 + xs2ss) / txs2

    # ---- INaL, ICab, ICl, INa, INab, I_katp, IpCa (.ode:577-599 and before) -------------------------------
    GNaL = where(epi, 0.6 * GNaL_b, GNaL_b)
    fINaLp = fCaMKp
    INaL_INaL = (mL * (GNaL * (-ENa + v))) * (fINaLp * hLp + hL * (1.0 - fINaLp))
    hLss = 1.0 / (exp((v + 87.61) / 7.488) + 1.0)
    hLssp = 1.0 / (exp((v + 93.81) / 7.488) + 1.0)
    mLss = 1.0 / (exp((-(v + 42.85)) / 5.264) + 1.0)
    thLp = 3.0 * thL
    tmL = 0.06487 * exp(-(((v - 4.823) / 51.12) ** 2.0)) + 0.1292 * exp(-(((v + 45.79) / 15.54) ** 2.0))
    dhL_dt = (-hL + hLss) / thL
    dhLp_dt = (-hLp + hLssp) / thLp
    dmL_dt = (-mL + mLss) / tmL
    ICab_ICab = ((vffrt * (PCab * 4.0)) * (-cao * gamma_cao + (cai * gamma_cai) * e2)) / (e2 - 1.0)
    IClCa_junc = ((Fjunc * GClCa) / (KdClCa / cass + 1.0)) * (-EClss + v)
    IClCa_sl = ((GClCa * (1.0 - Fjunc)) / (KdClCa / cai + 1.0)) * (-ECl + v)
    IClCa = IClCa_junc + IClCa_sl
    IClb = GClb * (-ECl + v)
    fINap = fCaMKp
    INa_INa = (m**3.0 * (GNa * (-ENa + v))) * (j * (h * (1.0 - fINap)) + jp * (fINap * hp))
    gt40 = v > -40.0
    ah = where(gt40, 0.0, 4.43126792958051e-7 * exp(-0.147058823529412 * v))
    aj = where(gt40, 0.0, -(v + 37.78) * (25428.0 * exp(0.28831 * v) + 6.948e-6) * exp(-0.04391 * v)
               / (50262745825.954 * exp(0.311 * v) + 1.0))
    bh = where(gt40, 0.77 * exp(0.0900900900900901 * v) / (0.13 * exp(0.0900900900900901 * v) + 0.0497581410839387),
               2.7 * exp(0.079 * v) + 310000.0 * exp(0.3485 * v))
    bj = where(gt40, 0.6 * exp(0.157 * v) / (1.0 * exp(0.1 * v) + 0.0407622039783662),
               0.02424 * exp(0.12728 * v) / (1.0 * exp(0.1378 * v) + 0.00396086833990426))
    hss = 1.0 / ((exp((v + 71.55) / 7.43) + 1.0) ** 2.0)
    hssp = 1.0 / ((exp((v + 77.55) / 7.43) + 1.0) ** 2.0)
    jss = hss
    mss = 1.0 / ((exp((-(v + 56.86)) / 9.03) + 1.0) ** 2.0)
    th = 1.0 / (ah + bh)
    tj = 1.0 / (aj + bj)
    tjp = 1.46 * tj
    tm = 0.06487 * exp(-(((v - 4.823) / 51.12) ** 2.0)) + 0.1292 * exp(-(((v + 45.79) / 15.54) ** 2.0))
    dh_dt = (-h + hss) / th
    dhp_dt = (-hp + hssp) / th
    dj_dt = (-j + jss) / tj
    djp_dt = (-jp + jss) / tjp
    dm_dt = (-m + mss) / tm
    INab_INab = ((PNab * vffrt) * (nai * e1 - nao)) / (e1 - 1.0)
    akik = (ko / K_o_n) ** 0.24
    bkik = 1.0 / ((A_atp / K_atp) ** 2.0 + 1.0)
    I_katp_I_katp = (bkik * (akik * (fkatp * gkatp))) * u
    IpCa_IpCa = (GpCa * cai) / (KmCap + cai)

    # ---- stimulus and membrane potential (.ode:600-604) ---------------------------------------------
    since = -i_Stim_Period * np.floor(-(i_Stim_Start - t) / i_Stim_Period) - i_Stim_Start + t
    Istim = np.where(np.logical_and(i_Stim_Start <= t, i_Stim_PulseDuration >= since), i_Stim_Amplitude, 0.0)
    dv_dt = -(Istim + (I_katp_I_katp + (IClb + (IClCa + (ICab_ICab + (IpCa_IpCa + (IKb_IKb + (INab_INab + (
        INaK_INaK + (INaCa_ss + (INaCa_i + (IK1_IK1 + (IKs_IKs + (IKr_IKr + (ICaK + (ICaNa + (ICaL_ICaL + (
            Ito_Ito + (INaL_INaL + INa_INa)))))))))))))))))))

    # ---- diffusion fluxes, SERCA, ryanodine receptor, translocation (.ode:606-633) ------------------------
    Jdiff = (-cai + cass) / tauCa
    JdiffCl = (-cli + clss) / tauNa  # the specification divides the chloride flux by tauNa, not tauCl
    JdiffK = (-ki + kss) / tauK
    JdiffNa = (-nai + nass) / tauNa
    upScale = where(epi, 1.3, 1.0)
    Jleak = (0.0048825 * cansr) / 15.0
    Jupnp = (cai * (upScale * 0.005425)) / (cai + 0.00092)
    Jupp = (cai * ((upScale * 2.75) * 0.005425)) / ((cai + 0.00092) - 0.00017)
    fJupp = fCaMKp
    Jup = Jup_b * (-Jleak + (Jupnp * (1.0 - fJupp) + Jupp * fJupp))
    fJrelp = fCaMKp
    Jrel = Jrel_b * (Jrel_np * (1.0 - fJrelp) + Jrel_p * fJrelp)
    a_rel = (0.5 * bt) / 1.0
    btp = 1.25 * bt
    a_relp = (0.5 * btp) / 1.0
    Jrel_inf_b = ((ICaL_ss * (-a_rel)) / 1.0) / ((cajsr_half / cajsr) ** 8.0 + 1.0)
    Jrel_inf = where(mid, 1.7 * Jrel_inf_b, Jrel_inf_b)
    Jrel_infp_b = ((ICaL_ss * (-a_relp)) / 1.0) / ((cajsr_half / cajsr) ** 8.0 + 1.0)
    Jrel_infp = where(mid, 1.7 * Jrel_infp_b, Jrel_infp_b)
    tau_rel_b = bt / (1.0 + 0.0123 / cajsr)
    tau_rel = where(tau_rel_b < 0.001, 0.001, tau_rel_b)
    tau_relp_b = btp / (1.0 + 0.0123 / cajsr)
    tau_relp = where(tau_relp_b < 0.001, 0.001, tau_relp_b)
    dJrel_np_dt = (Jrel_inf - Jrel_np) / tau_rel
    dJrel_p_dt = (Jrel_infp - Jrel_p) / tau_relp
    Jtr = (-cajsr + cansr) / 60.0

    # ---- intracellular ions (.ode:398-411) -------------------------------------------------------------
    cmdnmax = where(epi, 1.3 * cmdnmax_b, cmdnmax_b)
    Bcai = 1.0 / ((kmtrpn * trpnmax) / ((cai + kmtrpn) ** 2.0) + ((cmdnmax * kmcmdn) / ((cai + kmcmdn) ** 2.0) + 1.0))
    Bcajsr = 1.0 / ((csqnmax * kmcsqn) / ((cajsr + kmcsqn) ** 2.0) + 1.0)
    Bcass = 1.0 / ((BSLmax * KmBSL) / ((KmBSL + cass) ** 2.0) + ((BSRmax * KmBSR) / ((KmBSR + cass) ** 2.0) + 1.0))
    dcai_dt = Bcai * (((Acap * (-(-2.0 * INaCa_i + (ICab_ICab + (ICaL_i + IpCa_IpCa))))) / ((2.0 * F) * vmyo)
                       - Jup * vnsr / vmyo) + (Jdiff * vss) / vmyo)
    mechanics = []
    if land is not None:
        # ---- ToRORd_dynCl_endo_Land.ode:632-722: troponin buffering leaves Bcai and becomes the flux J_TRPN; the
        # calcium equation is the file's own (no ICaL_i term, INaCa_i / 3), restated as written ------------------------
        (XS, XW, CaTrpn, TmB, Zetas, Zetaw, Cd, emcoupling, lmbda, dLambda, mode, isacs, calib, ktrpn, ntrpn, Trpn50,
         rw, rs, gammas, gammaw, phi, Tot_A, Beta0, Beta1, cat50_ref, Tref, kuw, kws, ku, ntm, p_a, p_b, p_k, etal,
         etas) = land
        lambda_min12 = where(lmbda < 1.2, lmbda, 1.2)
        kwu = kuw * (1.0 / rw - 1.0) - kws
        ksu = kws * rw * (1.0 / rs - 1.0)
        Aw = Tot_A * rs / ((1.0 - rs) * rw + rs)
        As = Aw
        cw = phi * kuw * ((1.0 - rs) * (1.0 - rw)) / ((1.0 - rs) * rw)
        cs = phi * kws * ((1.0 - rs) * rw) / rs
        XU = (1.0 - TmB) - XS - XW
        gammawu = gammaw * _abs(Zetaw)
        zs_pos = (Zetas > 0) * Zetas                   # relations used as numbers (.ode:690)
        zs_neg = (Zetas < -1) * (-Zetas - 1.0)
        gammasu = gammas * where(zs_pos > zs_neg, zs_pos, zs_neg)
        dXS_dt = kws * XW - ksu * XS - gammasu * XS
        dXW_dt = kuw * XU - kwu * XW - kws * XW - gammawu * XW
        cat50 = cat50_ref + Beta1 * (lambda_min12 - 1.0)
        dCaTrpn_dt = ktrpn * (((cai * 1000.0 / cat50) ** ntrpn) * (1.0 - CaTrpn) - CaTrpn)
        kb = ku * Trpn50**ntm / (1.0 - rs - (1.0 - rs) * rw)
        ctm = CaTrpn ** (-ntm / 2.0)
        dTmB_dt = kb * where(ctm < 100.0, ctm, 100.0) * XU - ku * CaTrpn ** (ntm / 2.0) * TmB
        Bcai = 1.0 / (1.0 + cmdnmax * kmcmdn / (kmcmdn + cai) ** 2.0)
        J_TRPN = dCaTrpn_dt * trpnmax
        dcai_dt = Bcai * (-(IpCa_IpCa + ICab_ICab - 2.0 * INaCa_i / 3.0) * Acap / (2.0 * F * vmyo) - Jup * vnsr / vmyo
                          + Jdiff * vss / vmyo - J_TRPN)
        dZetas_dt = As * dLambda - cs * Zetas
        dZetaw_dt = Aw * dLambda - cw * Zetaw
        C = lambda_min12 - 1.0
        dCd = C - Cd
        eta = where(dCd < 0, etas, etal)
        dCd_dt = p_k * (C - Cd) / eta
        mechanics = [dXS_dt, dXW_dt, dCaTrpn_dt, dTmB_dt, dZetas_dt, dZetaw_dt, dCd_dt]
    dcajsr_dt = Bcajsr * (-Jrel + Jtr)
    dcansr_dt = Jup - Jtr * vjsr / vnsr
    dcass_dt = Bcass * (-Jdiff + ((Acap * (-(ICaL_ss - 2.0 * INaCa_ss))) / ((2.0 * F) * vss) + (Jrel * vjsr) / vss))
    dcli_dt = (Acap * (IClCa_sl + IClb)) / (F * vmyo) + (JdiffCl * vss) / vmyo
    dclss_dt = -JdiffCl + (Acap * IClCa_junc) / (F * vss)
    dki_dt = (Acap * (-(ICaK_i + (-2.0 * INaK_INaK + (Istim + (I_katp_I_katp + (IKb_IKb + (IK1_IK1 + (IKs_IKs + (
        IKr_IKr + Ito_Ito)))))))))) / (F * vmyo) + (JdiffK * vss) / vmyo
    dkss_dt = -JdiffK + (Acap * (-ICaK_ss)) / (F * vss)
    dnai_dt = (Acap * (-(INab_INab + (3.0 * INaK_INaK + (ICaNa_i + (3.0 * INaCa_i + (INaL_INaL + INa_INa))))))) / (
        F * vmyo) + (JdiffNa * vss) / vmyo
    dnass_dt = -JdiffNa + (Acap * (-(ICaNa_ss + 3.0 * INaCa_ss))) / (F * vss)

    return [dC1_dt, dC2_dt, dC3_dt, dI__dt, dO__dt, dCaMKt_dt, dJrel_np_dt, dJrel_p_dt, da_dt, dap_dt, diF_dt, diFp_dt,
            diS_dt, diSp_dt, dcai_dt, dcajsr_dt, dcansr_dt, dcass_dt, dcli_dt, dclss_dt, dki_dt, dkss_dt, dnai_dt,
            dnass_dt, dd_dt, dfcaf_dt, dfcafp_dt, dfcas_dt, dff__dt, dffp_dt, dfs_dt, djca_dt, dnca_i_dt, dnca_ss_dt,
            dh_dt, dhp_dt, dj_dt, djp_dt, dm_dt, dhL_dt, dhLp_dt, dmL_dt, dv_dt, dxs1_dt, dxs2_dt] + mechanics


def _params(parameters, shape, names=None):
    """Parameter entries, each a float (uniform) or an array broadcastable to the nodes (per-node (P, N))."""
    names = TORORD_PARAMETERS if names is None else names
    parameters = np.asarray(parameters, dtype=np.float64)
    if parameters.shape[0] != len(names):
        raise ValueError(f"expected {len(names)} parameters, got {parameters.shape[0]}")
    return [parameters[k] for k in range(parameters.shape[0])]


def _rhs_and_linearized(rhs, names, states, t, parameters):
    states = np.asarray(states, dtype=np.float64)
    ns = states.shape[0]
    ps = _params(parameters, states.shape[1:], names)
    tail = (1,) * (states.ndim - 1)
    eye = np.eye(ns)
    seeded = [Dual(states[k], eye[k].reshape((ns,) + tail)) for k in range(ns)]
    with np.errstate(all="ignore"):
        out = rhs(seeded, t, ps)
    out = [o if isinstance(o, Dual) else Dual(o, 0.0) for o in out]
    f = np.array([np.broadcast_to(np.asarray(o.a, dtype=np.float64), states.shape[1:]) for o in out])
    J = np.array([np.broadcast_to(np.asarray(o.b, dtype=np.float64), (ns,) + states.shape[1:])[i] for i, o in enumerate(out)])
    return f, J


def torord_rhs_and_linearized(states, t, parameters):
    """(f, J): f[i] = dy_i/dt and the total self-derivative J[i] = d f_i / d y_i, both (45, ...) arrays.  One
    forward-mode pass with a 45-component derivative part (state k is seeded with the k-th unit vector, so the pass
    carries every d f_i / d y_k); the diagonal is what generalized Rush-Larsen uses."""
    return _rhs_and_linearized(torord_rhs, TORORD_PARAMETERS, states, t, parameters)


def torord_generalized_rush_larsen(states, t, dt, parameters, delta=1e-8):
    """One GRL1 step of all 45 states (every state has a structurally non-zero self-derivative)."""
    states = np.asarray(states, dtype=np.float64)
    f, J = torord_rhs_and_linearized(states, t, parameters)
    with np.errstate(all="ignore"):
        return states + np.where(np.abs(J) > delta, f * (np.exp(J * dt) - 1) / J, f * dt)


# ------------------------------------------------------------------------------------------------
# ToR-ORd-dynCl + Land contraction model (odes/torord/ToRORd_dynCl_endo_Land.ode; no demo of the reference
# advances it, the file ships next to the one above): 52 states, 140 parameters
# ------------------------------------------------------------------------------------------------
_LAND_EXTRA_STATES = dict(cai=0.0001, XS=0.0, XW=0.0, CaTrpn=1e-8, TmB=1.0, Zetas=0.0, Zetaw=0.0, Cd=0.0)  # .ode:634-646
_LAND_EXTRA_PARAMETERS = dict(  # .ode:648-679
    emcoupling=1.0, lmbda=1.0, dLambda=0.0, mode=1.0, isacs=0.0, calib=1.0, ktrpn=0.1, ntrpn=2.0, Trpn50=0.35, rw=0.5,
    rs=0.25, gammas=0.0085, gammaw=0.615, phi=2.23, Tot_A=25.0, Beta0=2.3, Beta1=-2.4, cat50_ref=0.805, Tref=120.0,
    kuw=0.182, kws=0.012, ku=0.04, ntm=2.4, p_a=2.1, p_b=9.1, p_k=7.0, etal=200.0, etas=20.0,
)
# cai leaves the "intracellular ions" group and is declared with the mechanics states at the end of the file
TORORD_LAND_STATE_DEFAULTS = {k: v for k, v in TORORD_STATE_DEFAULTS.items() if k != "cai"} | _LAND_EXTRA_STATES
TORORD_LAND_STATES = tuple(TORORD_LAND_STATE_DEFAULTS)
TORORD_LAND_PARAMETER_DEFAULTS = TORORD_PARAMETER_DEFAULTS | _LAND_EXTRA_PARAMETERS
TORORD_LAND_PARAMETERS = tuple(TORORD_LAND_PARAMETER_DEFAULTS)
_CAI = TORORD_STATES.index("cai")


def torord_land_init_state_values(**values) -> np.ndarray:
    d = dict(TORORD_LAND_STATE_DEFAULTS)
    for k, v in values.items():
        if k not in d:
            raise KeyError(k)
        d[k] = v
    return np.array([d[k] for k in TORORD_LAND_STATES], dtype=np.float64)


def torord_land_init_parameter_values(**values) -> np.ndarray:
    d = dict(TORORD_LAND_PARAMETER_DEFAULTS)
    for k, v in values.items():
        if k not in d:
            raise KeyError(k)
        d[k] = v
    return np.array([d[k] for k in TORORD_LAND_PARAMETERS], dtype=np.float64)


def torord_land_rhs(states, t, parameters):
    """dy/dt of the 52 states in TORORD_LAND_STATES order (the 44 electrophysiology states without cai, then cai,
    XS, XW, CaTrpn, TmB, Zetas, Zetaw, Cd)."""
    states, parameters = list(states), list(parameters)
    base = states[:_CAI] + [states[44]] + states[_CAI:44]
    out = torord_rhs(base, t, parameters[:112], land=states[45:] + parameters[112:])
    return out[:_CAI] + out[_CAI + 1:45] + [out[_CAI]] + out[45:]


def torord_land_rhs_and_linearized(states, t, parameters):
    return _rhs_and_linearized(torord_land_rhs, TORORD_LAND_PARAMETERS, states, t, parameters)


def torord_land_generalized_rush_larsen(states, t, dt, parameters, delta=1e-8):
    """One GRL1 step of the 52 states."""
    states = np.asarray(states, dtype=np.float64)
    f, J = torord_land_rhs_and_linearized(states, t, parameters)
    with np.errstate(all="ignore"):
        return states + np.where(np.abs(J) > delta, f * (np.exp(J * dt) - 1) / J, f * dt)


def torord_forward_euler(states, t, dt, parameters):
    states = np.asarray(states, dtype=np.float64)
    ps = _params(parameters, states.shape[1:])
    with np.errstate(all="ignore"):
        f = torord_rhs([states[k] for k in range(states.shape[0])], t, ps)
    return states + dt * np.array([np.broadcast_to(np.asarray(_val(v), dtype=np.float64), states.shape[1:]) for v in f])
