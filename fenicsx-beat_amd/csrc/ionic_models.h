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
#include <type_traits>
#include <utility>

#include "beat_common.h"


// ------------------------------------------------------------------------------------------------
// exp() for the cell-model kernels.  x = k ln2/256 + r with |r| <= ln2/512, k = 256 m + j:
//   exp(x) = 2^m * 2^(j/256) * (1 + r + r^2/2 + r^3/6 + r^4/24)
// 2^(j/256) comes from a 256-entry table staged in LDS (2 KB), the truncation error of the degree-4 polynomial
// is < 4e-17, so the result is within ~1 ulp.  k is rounded with the 1.5 * 2^52 trick: one fma gives the rounded
// value in the mantissa and k as a signed integer in the low dword (no v_rndne / v_cvt).  No overflow / underflow / NaN
// handling: arguments in these models are bounded (|x| < 700), and the result must be a normal number (BEAT_EXP_LO / _HI below:
// round 6 scales by an integer add into the exponent field); arguments that are rate * dt products (gate updates, exp(J dt))
// leave that range at unphysiological potentials and are clamped by their callers (beat_clamp_exp_arg).
// 12 VALU instructions (13 with shift + v_ldexp_f64 until round 5, 16 with the 64-entry table and a degree-5 polynomial, ~27 for
// the library routine) -- the ionic kernels are fp64-issue bound and the TP06 step evaluates 51 of them per node.
// ------------------------------------------------------------------------------------------------
constexpr int BEAT_EXP_TAB = 256;
__device__ const double kExp2Tab[BEAT_EXP_TAB] = {
    1.0, 1.0027112750502025, 1.0054299011128027, 1.0081558981184175,
    1.0108892860517005, 1.0136300849514894, 1.016378314910953, 1.019133996077738,
    1.0218971486541166, 1.0246677928971357, 1.0274459491187637, 1.030231637686041,
    1.0330248790212284, 1.0358256936019572, 1.0386341019613787, 1.041450124688316,
    1.0442737824274138, 1.0471050958792898, 1.0499440858006872, 1.0527907730046264,
    1.0556451783605572, 1.0585073227945128, 1.061377227289262, 1.0642549128844645,
    1.0671404006768237, 1.0700337118202419, 1.0729348675259756, 1.075843889062791,
    1.0787607977571199, 1.0816856149932152, 1.0846183622133092, 1.0875590609177697,
    1.0905077326652577, 1.0934643990728858, 1.0964290818163769, 1.099401802630222,
    1.102382583307841, 1.1053714457017412, 1.1083684117236787, 1.1113735033448175,
    1.1143867425958924, 1.1174081515673693, 1.1204377524096067, 1.12347556733302,
    1.1265216186082418, 1.129575928566288, 1.1326385195987192, 1.1357094141578055,
    1.1387886347566916, 1.1418762039695616, 1.1449721444318042, 1.148076478840179,
    1.1511892299529827, 1.154310420590216, 1.1574400736337511, 1.1605782120274988,
    1.1637248587775775, 1.1668800369524817, 1.1700437696832502, 1.1732160801636373,
    1.1763969916502812, 1.1795865274628758, 1.182784710984341, 1.1859915656609938,
    1.189207115002721, 1.1924313825831512, 1.1956643920398273, 1.1989061670743806,
    1.202156731452703, 1.2054161090051239, 1.2086843236265816, 1.2119613992768012,
    1.215247359980469, 1.2185422298274085, 1.2218460329727576, 1.2251587936371455,
    1.22848053610687, 1.2318112847340759, 1.2351510639369334, 1.2384998981998165,
    1.241857812073484, 1.245224830175258, 1.2486009771892048, 1.2519862778663162,
    1.255380757024691, 1.2587844395497165, 1.2621973503942507, 1.2656195145788063,
    1.2690509571917332, 1.2724917033894028, 1.275941778396392, 1.2794012075056693,
    1.2828700160787783, 1.2863482295460256, 1.2898358734066657, 1.2933329732290895,
    1.2968395546510096, 1.3003556433796506, 1.3038812651919358, 1.3074164459346773,
    1.3109612115247644, 1.3145155879493546, 1.318079601266064, 1.3216532776031575,
    1.3252366431597413, 1.3288297242059544, 1.3324325470831615, 1.3360451382041458,
    1.339667524053303, 1.3432997311868353, 1.3469417862329458, 1.3505937158920345,
    1.3542555469368927, 1.3579273062129011, 1.3616090206382248, 1.365300717204012,
    1.3690024229745905, 1.3727141650876684, 1.3764359707545302, 1.380167867260238,
    1.383909881963832, 1.387662042298529, 1.3914243757719262, 1.3951969099662003,
    1.3989796725383112, 1.4027726912202048, 1.4065759938190154, 1.4103896082172707,
    1.4142135623730951, 1.4180478843204152, 1.4218926021691656, 1.4257477441054942,
    1.42961333839197, 1.433489413367789, 1.4373759974489824, 1.4412731191286257,
    1.4451808069770467, 1.449099089642035, 1.4530279958490526, 1.4569675544014438,
    1.460917794180647, 1.4648787441464057, 1.4688504333369818, 1.4728328908693675,
    1.4768261459394993, 1.4808302278224719, 1.4848451658727524, 1.488870989524397,
    1.4929077282912648, 1.4969554117672355, 1.5010140696264256, 1.5050837316234065,
    1.5091644275934228, 1.5132561874526098, 1.5173590411982147, 1.5214730189088146,
    1.5255981507445384, 1.529734466947287, 1.533881997840956, 1.5380407738316568,
    1.5422108254079407, 1.5463921831410214, 1.550584877685, 1.5547889397770887,
    1.559004400237837, 1.5632312899713576, 1.567469639965553, 1.5717194812923414,
    1.5759808451078865, 1.5802537626528246, 1.5845382652524937, 1.588834384317164,
    1.593142151342267, 1.597461597908627, 1.6017927556826934, 1.606135656416771,
    1.6104903319492543, 1.6148568142048607, 1.6192351351948637, 1.6236253270173289,
    1.6280274218573478, 1.632441451987275, 1.6368674497669644, 1.6413054476440063,
    1.645755478153965, 1.6502175739206177, 1.6546917676561943, 1.6591780921616162,
    1.6636765803267364, 1.6681872651305825, 1.6727101796415966, 1.6772453570178785,
    1.681792830507429, 1.6863526334483934, 1.6909247992693053, 1.6955093614893326,
    1.7001063537185235, 1.7047158096580513, 1.709337763100463, 1.713972247929926,
    1.718619298122478, 1.723278947746274, 1.7279512309618377, 1.732636182022311,
    1.7373338352737062, 1.7420442251551564, 1.746767386199169, 1.7515033530318782,
    1.7562521603732995, 1.761013843037584, 1.7657884359332727, 1.7705759740635547,
    1.7753764925265212, 1.7801900265154245, 1.785016611318935, 1.789856282321401,
    1.7947090750031072, 1.7995750249405351, 1.804454167806624, 1.809346539371032,
    1.8142521755003989, 1.8191711121586085, 1.8241033854070534, 1.8290490314048973,
    1.8340080864093424, 1.8389805867758937, 1.843966568958626, 1.8489660695104508,
    1.8539791250833855, 1.8590057724288205, 1.864046048397789, 1.8690999899412386,
    1.8741676341103, 1.8792490180565602, 1.8843441790323345, 1.8894531543909392,
    1.8945759815869656, 1.8997126981765553, 1.9048633418176741, 1.9100279502703899,
    1.9152065613971474, 1.9203992131630474, 1.925605943636125, 1.930826790987627,
    1.9360617934922943, 1.9413109895286405, 1.9465744175792332, 1.9518521162309783,
    1.9571441241754002, 1.9624504802089273, 1.9677712232331759, 1.9731063922552343,
    1.978456026387951, 1.9838201648502194, 1.9891988469672663, 1.9945921121709402,
};

// log(): x = 2^e * m, m in [1, 2); the top 7 mantissa bits pick c_j = 1 + (j + 0.5)/128 from a table of
// (1/c_j, log c_j) pairs (2 KB in LDS, one ds_read_b128); r = m/c_j - 1, |r| <= 2^-8, and
// log x = e ln2 + log c_j + (r - r^2/2 + ... - r^6/6), truncation < 2e-18.  Arguments here are
// concentration ratios (positive, normal, far from 1), so no special cases.
struct LogEntry {
  double inv, logc;
};
__device__ const LogEntry kLogTab[128] = {
    {0.9961089494163424, 0.0038986404156573091361},
    {0.9884169884169884, 0.011650617219975250717},
    {0.9808429118773946, 0.019342962843130986244},
    {0.973384030418251, 0.026976587698202081386},
    {0.9660377358490566, 0.034552381506659725601},
    {0.9588014981273408, 0.042071213920687043533},
    {0.9516728624535316, 0.04953393512227667772},
    {0.9446494464944649, 0.056941376400138453816},
    {0.9377289377289377, 0.064294350705397258084},
    {0.9309090909090909, 0.071593653187008818793},
    {0.924187725631769, 0.07884006170777598897},
    {0.9175627240143369, 0.086034337341803154249},
    {0.9110320284697508, 0.093177224854183340943},
    {0.9045936395759717, 0.10026945316367516232},
    {0.8982456140350877, 0.10731173578908803202},
    {0.89198606271777, 0.1143047712800586345},
    {0.8858131487889274, 0.1212492436328696548},
    {0.8797250859106529, 0.12814582269193005726},
    {0.8737201365187713, 0.13499516453750482063},
    {0.8677966101694915, 0.14179791186025737894},
    {0.8619528619528619, 0.14855469432313718671},
    {0.8561872909698997, 0.1552661289111239693},
    {0.8504983388704319, 0.16193282026931322982},
    {0.8448844884488449, 0.1685553610298066548},
    {0.839344262295082, 0.17513433212784914888},
    {0.8338762214983714, 0.18167030310763463359},
    {0.8284789644012945, 0.18816383241818293955},
    {0.8231511254019293, 0.19461546769967167013},
    {0.8178913738019169, 0.20102574606059078297},
    {0.8126984126984127, 0.20739519434607058803},
    {0.807570977917981, 0.21372432939771816965},
    {0.8025078369905956, 0.22001365830528213494},
    {0.7975077881619937, 0.22626367865045341755},
    {0.7925696594427245, 0.23247487874309400507},
    {0.7876923076923077, 0.23864773785017501078},
    {0.7828746177370031, 0.24478272641769090819},
    {0.7781155015197568, 0.25088030628580942049},
    {0.7734138972809668, 0.25694093089750042631},
    {0.7687687687687688, 0.26296504550088136049},
    {0.764179104477612, 0.26895308734550394896},
    {0.7596439169139466, 0.27490548587279921274},
    {0.7551622418879056, 0.28082266290088779852},
    {0.750733137829912, 0.28670503280395431552},
    {0.7463556851311953, 0.29255300268637745602},
    {0.7420289855072464, 0.29836697255179726932},
    {0.7377521613832853, 0.3041473354672967825},
    {0.7335243553008596, 0.30989447772286471865},
    {0.7293447293447294, 0.31560877898630328503},
    {0.7252124645892352, 0.32129061245373424479},
    {0.7211267605633803, 0.32694034499585326725},
    {0.7170868347338936, 0.33255833730007659317},
    {0.713091922005571, 0.33814494400871639467},
    {0.7091412742382271, 0.34370051385331844251},
    {0.7052341597796143, 0.34922538978528827643},
    {0.7013698630136986, 0.35471990910292902055},
    {0.6975476839237057, 0.36018440357500783228},
    {0.6937669376693767, 0.36561919956096471219},
    {0.6900269541778976, 0.37102461812787261014},
    {0.6863270777479893, 0.37640097516425304692},
    {0.6826666666666666, 0.38174858149084837325},
    {0.6790450928381963, 0.38706774296844833951},
    {0.6754617414248021, 0.39235876060286390804},
    {0.6719160104986877, 0.39762193064713850298},
    {0.6684073107049608, 0.40285754470108350164},
    {0.6649350649350649, 0.4080658898082217493},
    {0.661498708010336, 0.41324724855021928886},
    {0.6580976863753213, 0.41840189913888388994},
    {0.6547314578005116, 0.42353011550580321375},
    {0.6513994910941476, 0.42863216738969868561},
    {0.6481012658227848, 0.43370832042155937479},
    {0.6448362720403022, 0.43875883620762796463},
    {0.6416040100250626, 0.44378397241030103668},
    {0.6384039900249376, 0.44878398282700673567},
    {0.6352357320099256, 0.45375911746712050751},
    {0.6320987654320988, 0.45870962262697668523},
    {0.628992628992629, 0.46363574096303253781},
    {0.6259168704156479, 0.46853771156323927471},
    {0.6228710462287105, 0.47341577001667210492},
    {0.6198547215496368, 0.47827014848147023268},
    {0.6168674698795181, 0.4831010757511357347},
    {0.6139088729016786, 0.48790877731923904177},
    {0.6109785202863962, 0.49269347544257520972},
    {0.6080760095011877, 0.49745538920281889802},
    {0.6052009456264775, 0.50219473456671551856},
    {0.6023529411764705, 0.50691172444485444172},
    {0.5995316159250585, 0.51160656874906207939},
    {0.5967365967365967, 0.51627947444845449704},
    {0.5939675174013921, 0.52093064562418535404},
    {0.5912240184757506, 0.52556028352292738743},
    {0.5885057471264368, 0.53016858660912163172},
    {0.585812356979405, 0.53475575061602767331},
    {0.5831435079726651, 0.53932196859560892193},
    {0.5804988662131519, 0.54386743096728349121},
    {0.5778781038374717, 0.54839232556557324645},
    {0.5752808988764045, 0.55289683768667764954},
    {0.5727069351230425, 0.55738115013400637873},
    {0.5701559020044543, 0.56184544326269186172},
    {0.5676274944567627, 0.56628989502311587346},
    {0.565121412803532, 0.57071468100347154572},
    {0.5626373626373626, 0.57511997447138794129},
    {0.5601750547045952, 0.57950594641464226028},
    {0.5577342047930284, 0.58387276558098258239},
    {0.5553145336225597, 0.58822059851708601311},
    {0.5529157667386609, 0.59254960960667153759},
    {0.5505376344086022, 0.59685996110779383745},
    {0.5481798715203426, 0.60115181318933478025},
    {0.5458422174840085, 0.60542532396671690852},
    {0.5435244161358811, 0.60968064953685529126},
    {0.5412262156448203, 0.61391794401237045013},
    {0.5389473684210526, 0.61813735955507875642},
    {0.5366876310272537, 0.62233904640877868782},
    {0.534446764091858, 0.62652315293135285953},
    {0.5322245322245323, 0.63068982562619864957},
    {0.5300207039337475, 0.63483920917301013998},
    {0.5278350515463918, 0.63897144645792069839},
    {0.5256673511293635, 0.6430866786030272781},
    {0.523517382413088, 0.64718504499530945601},
    {0.5213849287169042, 0.65126668331495819751},
    {0.5192697768762677, 0.65533172956312764597},
    {0.5171717171717172, 0.65938031808912782698},
    {0.5150905432595574, 0.66341258161706616696},
    {0.5130260521042084, 0.66742865127195627278},
    {0.5109780439121756, 0.67142865660530240745},
    {0.5089463220675944, 0.67541272562017683386},
    {0.5069306930693069, 0.679380984795797338},
    {0.504930966469428, 0.68333355911162063829},
    {0.5029469548133595, 0.68727057207096033905},
    {0.5009784735812133, 0.69119214572414201437},
};

// Minimum resident waves per SIMD the register allocator must leave room for (2nd argument of
// __launch_bounds__, Model::WAVES): the TP06 step has ~50 independent exp() chains that the scheduler would
// otherwise hoist until one wave owns the whole register file.
#ifndef BEAT_ODE_WAVES
#define BEAT_ODE_WAVES 2
#endif
#ifndef BEAT_ODE_WAVES_PER_NODE
#define BEAT_ODE_WAVES_PER_NODE 2  // kernels whose parameters are per-node rows (TP06: 53 more doubles per lane)
#endif

// The range FastMath::exp takes (round 6): on the device the factor 2^m goes into the table value's exponent field by an INTEGER add
// (one instruction in place of the shift + v_ldexp_f64 pair), which neither underflows to 0 nor overflows to inf -- the result must be a
// normal number: exp(-708) = 3.3e-308 (m = -1022) ... exp(709) = 8.2e307.  Callers whose argument is a rate * dt product clamp it into
// that range: below -708 the clamp changes no RESULT (1 - 3e-308 = 1, 3e-308 - 1 = -1 in fp64: what the gate updates and the GRL1
// increments do with it), above 709.78 the callers put the overflow back (beat_exp_overflow: the scheme's literal expression gives
// inf there, and so does the oracle).
constexpr double BEAT_EXP_LO = -708.0, BEAT_EXP_HI = 709.0;
__device__ __forceinline__ double beat_clamp_exp_arg(double x) { return fmin(fmax(x, BEAT_EXP_LO), BEAT_EXP_HI); }
// exp(x) for any x >= BEAT_EXP_LO given e = FastMath::exp(min(x, BEAT_EXP_HI)): inf where exp overflows (x > 709.78...)
__device__ __forceinline__ double beat_exp_overflow(double x, double e) { return x > 709.782712893384 ? HUGE_VAL : e; }
// The table entry as the device's exp() wants it: 2^(j/256) with j << 12 taken off its high word, so that adding the whole
// k = 256 m + j, shifted by 12, leaves m in the exponent field (k << 12 = (m << 20) + (j << 12))
template <bool INT_SCALE>
__device__ __forceinline__ double beat_exp_tab_entry(double t, int j) {
#ifdef __AMDGCN__
  if constexpr (INT_SCALE) return __hiloint2double(__double2hiint(t) - (j << 12), __double2loint(t));
#endif
  (void)j;
  return t;
}

// 1/x: hardware estimate (~26 bits) + one third-order step r (1 + e + e^2), e = 1 - x r
__device__ __forceinline__ double beat_rcp(double x) {
  const double r = __builtin_amdgcn_rcp(x);
  const double e = fma(-x, r, 1.0);
  return fma(r, fma(e, e, e), r);
}

// 1/sqrt(x): hardware estimate + one third-order step r (1 + e/2 + 3 e^2/8), e = 1 - x r^2 (9 issue slots; sqrt(x) = x r)
__device__ __forceinline__ double beat_rsqrt(double x) {
  const double r = __builtin_amdgcn_rsq(x);
  const double e = fma(-(x * r), r, 1.0);
  return fma(r * fma(e, 0.375, 0.5), e, r);
}

// a double constant through a scalar register pair (see torord_dyncl.h: phi_small)
__device__ __forceinline__ double beat_sconst(double c) {
#ifdef __AMDGCN__  // (tests/tp06_host_harness.cpp builds this header with g++)
  asm volatile("" : "+s"(c));
#endif
  return c;
}

#ifndef BEAT_FM_PIN
#define BEAT_FM_PIN 1
#endif
// INT_SCALE (round 6, the TP06 step): exp() puts 2^m into the table value's exponent field by an integer add -- see exp() and
// BEAT_EXP_LO / _HI; the table then holds adjusted entries (beat_exp_tab_entry).  false: v_ldexp_f64, any argument down to -5.8e6
// underflows to 0 (ToR-ORd, the generated models, the forward-Euler models).
template <bool INT_SCALE_>
struct FastMathT {
  static constexpr bool INT_SCALE = INT_SCALE_;
  const double* __restrict__ tab;  // LDS copy of kExp2Tab
  const LogEntry* __restrict__ ltab;  // LDS copy of kLogTab
  // 1.5 * 2^52, the rounding constant of exp(), in a VGPR pair for the whole step (the kernels make it opaque once, beat_fm_pin): an
  // fma takes ONE scalar constant on gfx9, so `fma(x, 256/ln2, 1.5 * 2^52)` needs the second in VGPRs, and as a plain literal the
  // register allocator re-materialised it (two v_mov_b32) at ten places of the TP06 step rather than keep two registers live
  double magic = 6755399441055744.0;
  __device__ __forceinline__ double log(double x) const {
    const int hi = __double2hiint(x);
    const int e = ((hi >> 20) & 0x7ff) - 1023;
    const LogEntry t = ltab[(hi >> 13) & 127];
    const double m = __hiloint2double((hi & 0x000fffff) | 0x3ff00000, __double2loint(x));  // mantissa in [1, 2)
    const double r = fma(m, t.inv, -1.0);
    double p = fma(r, -1.0 / 6.0, 1.0 / 5.0);
    p = fma(r, p, -0.25);
    p = fma(r, p, 1.0 / 3.0);
    p = fma(r, p, -0.5);
    p = fma(r * r, p, r);
    const double ed = (double)e;
    return fma(ed, 0.693147180559663, t.logc + fma(ed, 2.8235290563031577e-13, p));
  }
  __device__ __forceinline__ double exp(double x) const {
    const double kb = fma(x, 369.3299304675746, magic);    // 256 / ln 2; 1.5 * 2^52: k in the low dword
    const double k = kb - magic;
    double r = fma(k, -0.002707606173999011, x);                        // ln2/256, high part (34 bits: exact product)
    r = fma(k, -6.327543041662719e-14, r);                              // low part
    const int ki = __double2loint(kb);
    const double t = tab[ki & (BEAT_EXP_TAB - 1)];
    // exp(r) - 1 = r + r^2 (1/2 + r/6 + r^2/24), every fma with at most ONE constant that is not an inline one (an fma
    // with two takes the second from VGPRs: two v_mov_b32 per use)
    const double r2 = r * r;
    double p = fma(r, beat_sconst(1.0 / 6.0), 0.5);
    p = fma(r2, beat_sconst(1.0 / 24.0), p);
    p = fma(r2, p, r);
#ifdef __AMDGCN__
    if constexpr (INT_SCALE) {
      // 2^m 2^(j/256): k << 12 added to the (adjusted, beat_exp_tab_entry) table value's high word -- one v_lshl_add_u32; x in
      // [BEAT_EXP_LO, BEAT_EXP_HI] (NaN stays NaN: r is NaN).  Bit for bit what ldexp gives there: scaling by 2^m commutes with the fma.
      const double ts = __hiloint2double(__double2hiint(t) + (int)((unsigned)ki << 12), __double2loint(t));
      return fma(ts, p, ts);
    }
#endif
    return ldexp(fma(t, p, t), ki >> 8);  // (also the host harnesses' form: the plain table)
  }
};

using FastMath = FastMathT<false>;
template <bool I>
__device__ __forceinline__ void beat_fm_pin(FastMathT<I>& fm) {
#if defined(__AMDGCN__) && BEAT_FM_PIN
  asm volatile("" : "+v"(fm.magic));
#endif
}

// The generated models' Goldman-Hodgkin-Katz fluxes  v g / (exp(c v) - 1)  are 0/0 at v = 0 and their v-derivative
// loses all accuracy next to it (relative error ~ ulp / (c v F/RT)^2; round-1 notes in DESIGN.md): the intermediates
// they go through are evaluated at a potential kept at least 1e-4 mV away from the singular value.  That leaves the
// derivative five correct digits; on the rare step a node spends inside the window its rates are evaluated up to
// 1e-4 mV off, which moves an increment by |d increment / dv| * 1e-4 mV (~2e-5 of the increment).
__device__ __forceinline__ double beat_guard(double v) { return fabs(v) < 1.0e-4 ? copysign(1.0e-4, v) : v; }

// Access to the state-major array for one node (row k at base + k*ld).
// Cache policy of the state rows' loads and stores (-DBEAT_ODE_NT: bit 0 non-temporal loads, bit 1 non-temporal stores; default
// 0 = plain).  Round 5 measured it because the library's own streaming probe reaches its best in-place rate with both
// (csrc/beat_probe.hip, profiles/r05_streaming.md: 6.5 against 6.0 TB/s for ONE stream per wave; 5.85 against 5.73 for 19 row
// streams, the pattern of this kernel); the result for the kernels themselves is in profiles/r05_streaming.md.
// Round 6: 3 is the default.  With one block per tile and four waves per SIMD (beat_ode.hip: ode_grid) the kernel behaves as the
// probe does: TP06 512^3 in one process on the same memory 8.60 - 8.64 -> 8.41 ms (loads only: 8.55 - 8.57, stores only: no change),
// the 512^3 step 12.68 / 13.12 -> 12.41 / 12.80 ms on the two levels consecutive processes alternate between
// (profiles/r06_ode_addressing.md); ToR-ORd: 2.465 -> 2.44 ms at 256^3, its class kernel (plain accesses) unchanged.
#ifndef BEAT_ODE_NT
#define BEAT_ODE_NT 3
#endif
__device__ __forceinline__ double beat_row_load(const double* p) {
#if (BEAT_ODE_NT & 1) && defined(__clang__)  // (the host harnesses build this header with g++: plain accesses there)
  return __builtin_nontemporal_load(p);
#else
  return *p;
#endif
}
__device__ __forceinline__ void beat_row_store(double* p, double v) {
#if (BEAT_ODE_NT & 2) && defined(__clang__)
  __builtin_nontemporal_store(v, p);
#else
  *p = v;
#endif
}

// Addressing (round 6): `base` points at the first node of the wave's TILE in row 0 (wave-uniform: it lives in SGPRs, and so
// does base + k ld for every row), `i` is the lane's node within the tile as an UNSIGNED 32-bit number: address = uniform 64-bit
// base + zero-extended 32-bit lane offset, the form global_load / global_store take with the base in an SGPR pair and ONE offset
// VGPR shared by every row.  With the node's 64-bit index in `i` (rounds 1 - 5) each row's address was a VGPR pair of its own
// (v_lshl_add_u64 per row), kept from the row's load to its store: 38 of the TP06 step's 134 VGPRs, 90 of ToR-ORd's 227.
// (the offset in BYTES, formed in 32-bit arithmetic: `base + zext(lane)` scales the 64-bit extension by 8, which the instruction
// selector cannot prove to fit the 32-bit offset register)
#ifndef BEAT_AT_OPAQUE
#define BEAT_AT_OPAQUE 1
#endif
// base + k ld with the stride opaque per access (BEAT_ROW_OPAQUE): the row's base is then formed on the scalar unit where the access
// is (s_mul + s_add, a handful of otherwise idle SALU cycles) instead of being kept in an SGPR pair from the row's load to its
// store -- 19 (TP06) to 52 (ToR-ORd + Land) pairs of the ~100 SGPRs a wave has, spilled to VGPR lanes beyond that
#ifndef BEAT_ROW_OPAQUE
#define BEAT_ROW_OPAQUE 1
#endif
template <class T>
__device__ __forceinline__ T* beat_row(T* base, int k, int64_t ld) {
#if defined(__AMDGCN__) && BEAT_ROW_OPAQUE
  asm volatile("" : "+s"(ld));
#endif
  return base + (int64_t)k * ld;
}
template <class T>
__device__ __forceinline__ T* beat_at(T* uniform_base, unsigned& byte_off) {
  typedef typename std::conditional<std::is_const<T>::value, const char, char>::type Byte;
#if defined(__AMDGCN__) && BEAT_AT_OPAQUE
  // (the offset made opaque at every access, IN PLACE -- the caller's own variable, so that there is one value chain and no copy:
  // otherwise ONE 64-bit extension of the offset is formed at the top of the tile, outside the blocks of the accesses, and
  // instruction selection -- block by block -- no longer sees "uniform base + zext(32-bit offset)", the form that goes into the
  // instruction's own address operands.  On a by-value copy the same statement cost a v_mov_b32 per access.)
  asm volatile("" : "+v"(byte_off));
#endif
  return (T*)((Byte*)uniform_base + byte_off);
}
struct NodeIO {
  double* __restrict__ base;
  int64_t ld;
  mutable unsigned i;  // BYTE offset of the lane's node within the tile (mutable: beat_at makes it opaque in place)
  double* __restrict__ v_copy;  // optional mirror of row v_index (the PDE unknown; offset like `base`), may be null
  int v_index;
  __device__ __forceinline__ double load(int k) const { return beat_row_load(beat_at(beat_row(base, k, ld), i)); }
  __device__ __forceinline__ void store(int k, double v) const {
    beat_row_store(beat_at(beat_row(base, k, ld), i), v);
    if (v_copy != nullptr && k == v_index) *beat_at(v_copy, i) = v;
  }
};

// The same with a pending update of the membrane potential (beat_ode_step_pending): row VIDX -- a compile-time
// constant, Model::V_INDEX, so that every other load stays a plain load -- is read as
// V + sum_j pa[j] pp[j], accumulated in the order x_flush_kernel uses (bit-identical to a flushed row); the store
// writes the complete new value.
constexpr int BEAT_MAX_PENDING = 6;  // = default ring size of the deferred-x PCG: what the plain kernels' pending path takes
constexpr int BEAT_MAX_PENDING_CLASS = 12;  // the class kernel consumes the pending directions ahead of its passes: the long ring (PRING_MAX)
template <int VIDX>
struct NodeIOPending {
  double* __restrict__ base;  // (tile base and 32-bit lane offset: see NodeIO)
  int64_t ld;
  mutable unsigned i;  // BYTE offset (see NodeIO)
  double* __restrict__ v_copy;
  int npend;
  double pa[BEAT_MAX_PENDING], pp[BEAT_MAX_PENDING];
  double ge, gd, gp0, gp1;          // what the fields e, d, dp[0], dp[1] held at this node (when gt.d != nullptr; read up front)
  beat_pde_detail::GuessTerms gt;   // where the step's diffusion increment is recorded and the next guess prepared
  __device__ __forceinline__ double load(int k) const {
    double x = beat_row_load(beat_at(beat_row(base, k, ld), i));
    if (k == VIDX) {
      if (gt.d != nullptr) {  // same expressions and order as x_flush_kernel's guess branch
        double inc = (!gt.accumulate && gt.use_e) ? ge : 0.0;  // (ge is only loaded when it is due: beat_guess_needs_e)
#pragma unroll
        for (int j = 0; j < BEAT_MAX_PENDING; ++j)
          if (j < npend) inc = fma(pa[j], pp[j], inc);
        // (recorded here, where the increment is formed: doing it with the potential's own store near the end of the
        // step keeps five more values alive through the step and measured no faster)
        beat_pde_detail::beat_guess_record(gt, beat_at(gt.d, i), beat_at(gt.e, i), inc, gd, gp0, gp1, ge);
        return x + inc;
      }
#pragma unroll
      for (int j = 0; j < BEAT_MAX_PENDING; ++j)
        if (j < npend) x = fma(pa[j], pp[j], x);
    }
    return x;
  }
  __device__ __forceinline__ void store(int k, double v) const {
    beat_row_store(beat_at(beat_row(base, k, ld), i), v);
    if (k == VIDX && v_copy != nullptr) *beat_at(v_copy, i) = v;
  }
};

// The same interface on a register-resident copy of the node's states (in-kernel time loops).  Every
// model loads a state before it stores it and never reloads it afterwards, so one array suffices.
struct RegIO {
  double* y;
  __device__ __forceinline__ double load(int k) const { return y[k]; }
  __device__ __forceinline__ void store(int k, double v) const { y[k] = v; }
};

// Parking a value in LDS across the part of a step that does not use it (an IO type that has somewhere to park it offers
// stash / unstash: StashIO in beat_ode_kernel.h; every other IO type keeps the value where it is).  The TP06 step holds 19
// loaded states, nine shared exponentials and five conductances while its twelve gate blocks run: at four waves per SIMD (128
// VGPRs) the register allocator spilled 28 B per lane to scratch -- memory operations that also queue on the in-order vector-memory
// counter behind the gate stores.  LDS reads and writes count on lgkmcnt and cost 64 B per node and value of LDS traffic.
template <class IO, class = void>
struct beat_has_stash : std::false_type {};
template <class IO>
struct beat_has_stash<IO, std::void_t<decltype(std::declval<const IO&>().stash(0, 0.0))>> : std::true_type {};
template <class IO>
__device__ __forceinline__ void beat_stash(const IO& io, int slot, double v) {
  if constexpr (beat_has_stash<IO>::value) io.stash(slot, v);
}
template <class IO>
__device__ __forceinline__ double beat_unstash(const IO& io, int slot, double kept) {
  if constexpr (beat_has_stash<IO>::value)
    return io.unstash(slot);
  else
    return kept;
}

// ------------------------------------------------------------------------------------------------
// v' = -a s, s' = b v, forward Euler  (tests/test_odesolver.py:11-17)
// ------------------------------------------------------------------------------------------------
struct SimpleOde {
  static constexpr int NS = 2, NP = 2, V_INDEX = 0;
  static constexpr bool REGISTER_LOOP = true;
  static constexpr int WAVES = BEAT_ODE_WAVES;
  static constexpr int WAVES_PER_NODE = BEAT_ODE_WAVES_PER_NODE;  // per-node parameter rows: NP more values per lane
  struct Derived {};
  __host__ __device__ static Derived derive(const double*) { return {}; }
  template <class IO>
  __device__ static __forceinline__ void step(const IO& io, const double* p, const Derived&, const FastMath&,
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
  static constexpr int NS = 2, NP = 10, V_INDEX = 1;
  static constexpr bool REGISTER_LOOP = true;
  static constexpr int WAVES = BEAT_ODE_WAVES;
  static constexpr int WAVES_PER_NODE = BEAT_ODE_WAVES_PER_NODE;  // per-node parameter rows: NP more values per lane
  struct Derived {};
  __host__ __device__ static Derived derive(const double*) { return {}; }
  template <class IO>
  __device__ static __forceinline__ void step(const IO& io, const double* p, const Derived&, const FastMath&,
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
  static constexpr int NS = 2, NP = 11, V_INDEX = 1;
  static constexpr bool REGISTER_LOOP = true;
  static constexpr int WAVES = BEAT_ODE_WAVES;
  static constexpr int WAVES_PER_NODE = BEAT_ODE_WAVES_PER_NODE;  // per-node parameter rows: NP more values per lane
  struct Derived {};
  __host__ __device__ static Derived derive(const double*) { return {}; }
  template <class IO>
  __device__ static __forceinline__ void step(const IO& io, const double* p, const Derived&, const FastMath&,
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
#if defined(__clang__) && !defined(BEAT_TP06_NO_CONTRACT)
#pragma clang fp contract(fast)  // a * b + c as one fma inside the step (the library as a whole is built with contraction off)
#endif
struct Tp06Grl1 {
  static constexpr int NS = 19, NP = 53, V_INDEX = 17;
  static constexpr bool ACCESSOR_PARAMS = true;  // derive / step take any p indexable by parameter number (beat_ode_jit.hip)
  static constexpr bool REGISTER_LOOP = true;
#ifndef BEAT_TP06_WAVES
#define BEAT_TP06_WAVES 4  // (round 6: 118 - 124 VGPRs in the uniform and the class kernels; the bound keeps it that way)
#endif
#ifndef BEAT_TP06_STASH
#define BEAT_TP06_STASH 0
#endif
  static constexpr int WAVES = BEAT_TP06_WAVES;
  static constexpr int WAVES_PER_NODE = BEAT_ODE_WAVES_PER_NODE;  // per-node parameter rows: NP more values per lane
  static constexpr int STASH_SLOTS = BEAT_TP06_STASH;  // values parked in LDS while the gate blocks run (beat_stash)
  // exp() with the factor 2^m added into the exponent field (FastMathT<true>: one instruction less per exp(), 51 per node; results in
  // the normal range only, i.e. |V| < ~370 mV -- beyond it the Gaussian time constants' arguments leave [BEAT_EXP_LO, BEAT_EXP_HI])
  // Measured (profiles/r06_ode_addressing.md): 1778 -> 1729 static VALU instructions, 8.10 -> 8.00 ms in one process, 12.20 -> 12.165 ms per
  // 512^3 step -- 0.3 %, for a step that would return garbage instead of NaN beyond its range: OFF.
#ifndef BEAT_TP06_EXP_INT
#define BEAT_TP06_EXP_INT 0
#endif
  using FM = FastMathT<BEAT_TP06_EXP_INT != 0>;
  static constexpr bool FM_PIN = true;  // exp()'s rounding constant in a VGPR pair for the whole step (FastMath::magic): 118 -> 120 VGPRs, -18 VALU instructions
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
  // P: anything indexable by parameter number (const double*, a per-lane array, MixedParams of beat_ode.hip)
  template <class P>
  __host__ __device__ static Derived derive(const P& p) {
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

  __device__ static __forceinline__ double rcp(double x) { return beat_rcp(x); }
  // Reciprocals of 2-4 independent values from ONE v_rcp_f64 (Montgomery's trick): 1/(ab) b = 1/a ...
  // Every group here is a set of finite, normal, same-sign-insensitive denominators whose product
  // stays far inside the double range; each result carries ~2 extra roundings.
  __device__ static __forceinline__ void rcp2(double a, double b, double& ia, double& ib) {
    const double r = rcp(a * b);
    ia = r * b;
    ib = r * a;
  }
  __device__ static __forceinline__ void rcp3(double a, double b, double c, double& ia, double& ib, double& ic) {
    const double ab = a * b;
    const double r = rcp(ab * c);
    ic = r * ab;
    const double t = r * c;
    ia = t * b;
    ib = t * a;
  }
  __device__ static __forceinline__ void rcp4(double a, double b, double c, double d, double& ia, double& ib,
                                              double& ic, double& id) {
    const double ab = a * b, cd = c * d;
    const double r = rcp(ab * cd);
    const double rab = r * cd, rcd = r * ab;
    ia = rab * b;
    ib = rab * a;
    ic = rcd * d;
    id = rcd * c;
  }
  __device__ static __forceinline__ double grl1(const FM& fm, double y, double fy, double J, double dt) {
    return y + ((fabs(J) > 1e-8) ? fy * (beat_exp_overflow(J * dt, fm.exp(beat_clamp_exp_arg(J * dt))) - 1.0) * rcp(J) : fy * dt);
  }
  // same with 1/J supplied by the caller (batched); rJ is only used where |J| > 1e-8
  __device__ static __forceinline__ double grl1r(const FM& fm, double y, double fy, double J, double rJ,
                                                 double dt) {
    return y + ((fabs(J) > 1e-8) ? fy * (beat_exp_overflow(J * dt, fm.exp(beat_clamp_exp_arg(J * dt))) - 1.0) * rJ : fy * dt);
  }
  __device__ static __forceinline__ double guard(double J) { return (fabs(J) > 1e-8) ? J : 1.0; }
  // The GRL1 increment f (exp(J dt) - 1) / J of a non-gate state as f dt phi(J dt), phi(z) = (exp(z) - 1) / z by its Taylor polynomial
  // of degree 8 when |J dt| <= 1/16 (first omitted term z^9 / 10! < 5e-18; phi(0) = 1 is the |J| <= 1e-8 limit of the scheme), the
  // scheme's literal expression otherwise -- the form the ToR-ORd kernel has had since round 3 (torord_dyncl.h: advance).  At dt = 0.01 -
  // 0.05 ms nearly every node is inside the window for its seven non-gate states (the potential during an upstroke is not): 13
  // instructions in place of exp() + reciprocal + selection (~26), and no cancellation in exp(z) - 1.  Round 6: the kernel is bound by
  // fp64 issue once more (four waves per SIMD, 86 % VALU busy: profiles/r06_512.md).  BEAT_TP06_PHI=0: the literal expression always.
#ifndef BEAT_TP06_PHI
#define BEAT_TP06_PHI 1
#endif
  __device__ static __forceinline__ double phi_small(double z) {
    double ph = z * beat_sconst(1.0 / 362880.0) + beat_sconst(1.0 / 40320.0);
    ph = fma(z, ph, beat_sconst(1.0 / 5040.0));
    ph = fma(z, ph, beat_sconst(1.0 / 720.0));
    ph = fma(z, ph, beat_sconst(1.0 / 120.0));
    ph = fma(z, ph, beat_sconst(1.0 / 24.0));
    ph = fma(z, ph, beat_sconst(1.0 / 6.0));
    ph = fma(z, ph, 0.5);
    return fma(z, ph, 1.0);
  }
  __device__ static __forceinline__ double advance(const FM& fm, double y, double fy, double J, double dt) {
    const double z = J * dt;
    if (BEAT_TP06_PHI && fabs(z) <= 0.0625) return fma(fy * dt, phi_small(z), y);
    return grl1(fm, y, fy, J, dt);
  }
  // gate with f = (inf - y)/tau, J = -1/tau
#ifndef BEAT_TP06_GATE_PHI
#define BEAT_TP06_GATE_PHI 0  // 1: 1 - exp(z), z = -dt/tau, as -z phi7(z) when |z| <= 1/32 (per lane); measured, see profiles/r06_ode_addressing.md
#endif
  __device__ static __forceinline__ double phi7(double z) {  // (exp(z) - 1) / z, |z| <= 1/32: first omitted term z^8 / 9! < 3e-18
    double ph = z * beat_sconst(1.0 / 40320.0) + beat_sconst(1.0 / 5040.0);
    ph = fma(z, ph, beat_sconst(1.0 / 720.0));
    ph = fma(z, ph, beat_sconst(1.0 / 120.0));
    ph = fma(z, ph, beat_sconst(1.0 / 24.0));
    ph = fma(z, ph, beat_sconst(1.0 / 6.0));
    ph = fma(z, ph, 0.5);
    return fma(z, ph, 1.0);
  }
  __device__ static __forceinline__ double gate(const FM& fm, double y, double inf, double rtau, double dt) {
#if BEAT_TP06_GATE_PHI
    const double z = -dt * rtau;
    if (z >= -0.03125) return fma(inf - y, -z * phi7(z), y);
#endif
    return y + (inf - y) * (1.0 - fm.exp(fmax(-dt * rtau, BEAT_EXP_LO)));  // (-dt / tau <= 0)
  }

  // Fence for the instruction scheduler: the step is ~3000 straight-line instructions with ~50
  // independent exp() chains; without fences everything is hoisted and one wave needs the whole
  // register file.  Each fenced block keeps a few chains in flight, which is all the latency hiding
  // 3-4 resident waves per SIMD need.
#ifndef BEAT_TP06_FENCES
#define BEAT_TP06_FENCES 1  // 0: none (experiments)
#endif
#if BEAT_TP06_FENCES
#define BEAT_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define BEAT_FENCE()
#endif

  template <class IO, class P>
  __device__ static __forceinline__ void step(const IO& io, const P& p, const Derived& q, const FM& fm,
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
    // The concentrations are not needed before the gates are done, but their loads are issued HERE, ahead of the
    // gates' stores: on gfx9 loads and stores retire through one in-order counter, so a load issued behind the twelve
    // gate stores could only be waited for together with them (s_waitcnt vmcnt: ~2-4 k cycles for stores to reach HBM,
    // once per tile and wave) -- 12 more VGPRs through the gate section, which the three-wave budget has.  512^3, same
    // box, runs in pairs: 10.34-10.43 against 10.50-10.55 ms.  (Loading V after the gates, so that the two stores a
    // pending update makes at V's load precede no other load, added nothing to that: 10.34-10.39.)
    const double vCai0 = io.load(Ca_i), vCaSR0 = io.load(Ca_SR), vCass0 = io.load(Ca_ss), vNai0 = io.load(Na_i),
                 vKi0 = io.load(K_i), vR0 = io.load(R_prime);
    // parked in LDS while the gate blocks run (see beat_stash): the ones needed last first
    if (STASH_SLOTS >= 1) beat_stash(io, 0, vR0);
    if (STASH_SLOTS >= 2) beat_stash(io, 1, vCaSR0);
    if (STASH_SLOTS >= 3) beat_stash(io, 2, vKi0);
    if (STASH_SLOTS >= 4) beat_stash(io, 3, vNai0);
    if (STASH_SLOTS >= 5) beat_stash(io, 4, vCai0);
    if (STASH_SLOTS >= 6) beat_stash(io, 5, vCass0);

    // ---- shared exponentials of V ----------------------------------------------------------------------
    const double E20 = fm.exp(0.05 * v), E7 = fm.exp(v * (1.0 / 7.0));
    double I20, I7;
    rcp2(E20, E7, I20, I7);
    const double E10 = E20 * E20, I10 = I20 * I20;
    const double E5 = E10 * E10, I5 = I10 * I10;
    const double I6 = fm.exp(v * (-1.0 / 6.0));
    BEAT_FENCE();

    // ---- gates: y += (inf - y)(1 - exp(-dt/tau)); time constants as single quotients ------------------------
    {  // Xr1 (.ode:189-194): tau = 450/(1+ea) * 6/(1+eb);  Xr2 (.ode:196-201): tau = 3/(1+ea) * 1.12/(1+eb)
      double inf1, inf2;
      rcp2(1.0 + EXP_M26_7 * I7, 1.0 + fm.exp((v + 88.0) * (1.0 / 24.0)), inf1, inf2);  // exp((-26 - V)/7)
      const double rtau1 = (1.0 + EXP_M4P5 * I10) * (1.0 + fm.exp((v + 30.0) * (1.0 / 11.5))) * (1.0 / 2700.0);
      io.store(Xr1, gate(fm, oXr1, inf1, rtau1, dt));
      const double rtau2 = (1.0 + EXP_M3 * I20) * (1.0 + EXP_M3 * E20) * (1.0 / 3.36);
      io.store(Xr2, gate(fm, oXr2, inf2, rtau2, dt));
    }
    BEAT_FENCE();
    {  // Xs (.ode:206-211): tau = 1400/sqrt(1+ea) * 1/(1+eb) + 80 = (1400 + 80 D)/D
      const double D = sqrt(1.0 + EXP_5_6 * I6) * (1.0 + fm.exp((v - 35.0) * (1.0 / 15.0)));  // exp((5 - V)/6)
      double inf, rnum;
      rcp2(1.0 + fm.exp((-5.0 - v) * (1.0 / 14.0)), 1400.0 + 80.0 * D, inf, rnum);
      io.store(Xs, gate(fm, oXs, inf, D * rnum, dt));
    }
    BEAT_FENCE();
    {  // m (.ode:216-221): tau = 1/(1+ea) * (0.1/(1+eb) + 0.1/(1+ec))
      const double da = 1.0 + EXP_M12 * I5;                                   // exp((-60 - V)/5)
      const double db = 1.0 + EXP_7 * E5;                                     // exp((V + 35)/5)
      const double dc = 1.0 + fm.exp((v - 50.0) * (1.0 / 200.0));
      double rm, rsum;
      rcp2(1.0 + fm.exp((-56.86 - v) * (1.0 / 9.03)), db + dc, rm, rsum);
      io.store(m, gate(fm, om, rm * rm, da * db * dc * 10.0 * rsum, dt));
    }
    BEAT_FENCE();
    {  // h, j (.ode:223-235): tau = 1/(alpha + beta), shared steady state
      const double dh = 1.0 + fm.exp((v + 71.55) * (1.0 / 7.43));
      double rh, ah_bh, aj_bj;
      if (v < -40.0) {
        ah_bh = 0.057 * fm.exp(-(v + 80.0) * (1.0 / 6.8)) + 2.7 * fm.exp(0.079 * v) + 310000.0 * fm.exp(0.3485 * v);
        const double da = 1.0 + fm.exp(0.311 * (v + 79.23)), db = 1.0 + fm.exp(-0.1378 * (v + 40.14));
        const double na = (-25428.0 * fm.exp(0.2444 * v) - 6.948e-6 * fm.exp(-0.04391 * v)) * (v + 37.78);
        const double nb = 0.02424 * fm.exp(-0.01052 * v);
        double rab;
        rcp2(dh, da * db, rh, rab);
        aj_bj = (na * db + nb * da) * rab;
      } else {
        double rbh, rbj;
        rcp3(dh, 0.13 * (1.0 + fm.exp((v + 10.66) * (-1.0 / 11.1))), 1.0 + EXP_M3P2 * I10, rh, rbh, rbj);
        ah_bh = 0.77 * rbh;
        aj_bj = 0.6 * fm.exp(0.057 * v) * rbj;                               // exp(-0.1 (V + 32))
      }
      const double h_inf = rh * rh;
      io.store(h, gate(fm, oh, h_inf, ah_bh, dt));
      io.store(j, gate(fm, oj, h_inf, aj_bj, dt));
    }
    BEAT_FENCE();
    {  // d (.ode:243-249): tau = (1.4/(1+ea) + 0.25) * 1.4/(1+eb) + 1/(1+ec)
      const double da = 1.0 + fm.exp((-35.0 - v) * (1.0 / 13.0));
      const double db = 1.0 + EXP_1 * E5;                                     // exp((V + 5)/5)
      const double dc = 1.0 + EXP_2P5 * I20;                                  // exp((50 - V)/20)
      const double num = (1.4 + 0.25 * da) * 1.4 * dc + da * db;
      double inf, rnum;
      rcp2(1.0 + fm.exp((-8.0 - v) * (1.0 / 7.5)), num, inf, rnum);
      io.store(d, gate(fm, od, inf, da * db * dc * rnum, dt));
    }
    BEAT_FENCE();
    {  // f, f2 (.ode:251-259): tau = c G + A/(1+ea) + B/(1+eb) [+ 20]
      const double v27sq = (v + 27.0) * (v + 27.0);
      const double db = 1.0 + EXP_3 * E10;                                    // exp((V + 30)/10)
      const double da1 = 1.0 + EXP_1P3 * I10;                                 // exp((13 - V)/10)
      const double dab1 = da1 * db;
      const double num1 = (1102.5 * fm.exp(v27sq * (-1.0 / 225.0)) + 20.0) * dab1 + 200.0 * db + 180.0 * da1;
      const double da2 = 1.0 + EXP_2P5 * I10;                                 // exp((25 - V)/10)
      const double dab2 = da2 * db;
      const double num2 = 562.0 * fm.exp(v27sq * (-1.0 / 240.0)) * dab2 + 31.0 * db + 80.0 * da2;
      double inf1, rnum1, rinf2, rnum2;
      rcp4(1.0 + EXP_20_7 * E7, num1, 1.0 + EXP_5 * E7, num2, inf1, rnum1, rinf2, rnum2);  // exp((V+20)/7), exp((V+35)/7)
      io.store(f, gate(fm, of, inf1, dab1 * rnum1, dt));
      io.store(f2, gate(fm, of2, 0.67 * rinf2 + 0.33, dab2 * rnum2, dt));
    }
    BEAT_FENCE();
    {  // s, r (.ode:276-284)
      const double ds_ = 1.0 + EXP_M4 * E5;                                   // exp((V - 20)/5)
      const double num = (85.0 * fm.exp((v + 45.0) * (v + 45.0) * (-1.0 / 320.0)) + 3.0) * ds_ + 5.0;
      double s_inf, rnum, r_inf, rtau_r;
      rcp4(1.0 + EXP_4 * E5, num, 1.0 + EXP_20_6 * I6,                        // exp((V + 20)/5), exp((20 - V)/6)
           9.5 * fm.exp((v + 40.0) * (v + 40.0) * (-1.0 / 1800.0)) + 0.8, s_inf, rnum, r_inf, rtau_r);
      io.store(s, gate(fm, os, s_inf, ds_ * rnum, dt));
      io.store(r, gate(fm, orr, r_inf, rtau_r, dt));
    }
    BEAT_FENCE();

    const double vCass = STASH_SLOTS >= 6 ? beat_unstash(io, 5, vCass0) : vCass0;
    {  // fCass (.ode:261-264): depends on Ca_ss only
      const double c2 = 1.0 + (vCass * 20.0) * (vCass * 20.0);                // 1 + (Ca_ss/0.05)^2
      double rc2, rt;
      rcp2(c2, 80.0 + 2.0 * c2, rc2, rt);
      io.store(fCass, gate(fm, ofCass, 0.6 * rc2 + 0.4, c2 * rt, dt));
    }

    // ---- reversal potentials ------------------------------------------------------------------------
    const double vKi = STASH_SLOTS >= 3 ? beat_unstash(io, 2, vKi0) : vKi0;
    const double vNai = STASH_SLOTS >= 4 ? beat_unstash(io, 3, vNai0) : vNai0;
    const double vCai = STASH_SLOTS >= 5 ? beat_unstash(io, 4, vCai0) : vCai0;
    double rNai, rKi, rCai, rKs;
    rcp4(vNai, vKi, vCai, vKi + p[P_kna] * vNai, rNai, rKi, rCai, rKs);
    const double E_Na = q.RTF * fm.log(p[Na_o] * rNai);
    const double E_K = q.RTF * fm.log(p[K_o] * rKi);
    const double E_Ks = q.RTF * fm.log(q.KoPk * rKs);
    const double E_Ca = q.halfRTF * fm.log(p[Ca_o] * rCai);
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
      double e4, r1, rG1;                                    // exp(-0.5 u) = 1/G25, 1/(1 + e1), 1/(G25 + 1)
      rcp3(G25, 1.0 + e1, G25 + 1.0, e4, r1, rG1);
      const double aK1 = 0.1 * r1;
      const double daK1 = -0.06 * aK1 * e1 * r1;
      const double rD = G25 * rG1;                           // 1/(1 + e4)
      const double bK1 = (3.0 * e2 + e3) * rD;
      const double dbK1 = (0.0006 * e2 + 0.1 * e3 + 0.5 * e4 * bK1) * rD;
      const double epK = fm.exp((25.0 - v) * (1.0 / 5.98));  // plateau K current (.ode:296)
      double rab, rpK;
      rcp2(aK1 + bK1, 1.0 + epK, rab, rpK);
      const double xK1 = aK1 * rab;
      const double dxK1 = (daK1 * bK1 - aK1 * dbK1) * rab * rab;
      const double i_K1 = q.gK1s * xK1 * u;
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
    double rNaK, rNaKm, rS;
    rcp3(1.0 + 0.1245 * e5 + 0.0353 * e6, vNai + p[K_mNa], 1.0 + p[K_sat] * eg1, rNaK, rNaKm, rS);
    const double i_NaK = q.NaK_B * vNai * rNaKm * rNaK;
    const double di_NaK_dNai = q.NaK_B * p[K_mNa] * rNaKm * rNaKm * rNaK;
    dI_dV += i_NaK * q.FRT * (0.01245 * e5 + 0.0353 * e6) * rNaK;
    const double A1 = (vNai * vNai * vNai) * p[Ca_o], A2 = q.A2c * vCai;
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
    const double w15 = v - 15.0;
    const double xCaL = 2.0 * w15 * q.FRT;
    const double eCaL = fm.exp(xCaL);
    // The specification's w15 (..)/(eCaL - 1) is 0/0 at V = 15 mV.  Its value is fine next to the singularity, but
    // its V-derivative is a difference of two O(1/x) terms: the relative error of J_V grows like ulp/x^2, and a
    // node that passes within ~1e-10 mV of 15 mV (it happens: ~1e8 node-crossings per simulated beat at 512^3) gets
    // |J_V| ~ 1e8 of either sign, and exp(J_V dt) overflows.  Inside |x| < 1e-2 (|V - 15| < 0.13 mV) the factor
    // x/(e^x - 1) and its derivative are therefore taken from their series; outside nothing changes, bit for bit.
    const bool nearCaL = fabs(xCaL) < 1.0e-2;
    double rDc, rpCa;
    rcp2(nearCaL ? 1.0 : eCaL - 1.0, vCai + p[K_pCa], rDc, rpCa);
    const double NCaL = 0.25 * vCass * eCaL - p[Ca_o];
    double i_CaL = gates_CaL * w15 * NCaL * rDc;
    double dCaL_dV = gates_CaL * (NCaL * rDc + w15 * (2.0 * q.FRT) * eCaL * (p[Ca_o] - 0.25 * vCass) * rDc * rDc);
    double di_CaL_dCass = gates_CaL * w15 * 0.25 * eCaL * rDc;
    if (nearCaL) {
      const double x2 = xCaL * xCaL;
      const double phi = 1.0 + xCaL * (-0.5 + xCaL * (1.0 / 12.0 + x2 * (-1.0 / 720.0 + x2 * (1.0 / 30240.0))));
      const double dphi = -0.5 + xCaL * (1.0 / 6.0 + x2 * (-1.0 / 180.0 + x2 * (1.0 / 5040.0)));
      const double B = phi * q.halfRTF;  // (V - 15)/(e^x - 1)
      i_CaL = gates_CaL * NCaL * B;
      dCaL_dV = gates_CaL * (0.25 * vCass * eCaL * phi + NCaL * dphi);
      di_CaL_dCass = gates_CaL * 0.25 * eCaL * B;
    }
    dI_dV += dCaL_dV;
    const double i_p_Ca = p[g_pCa] * vCai * rpCa;
    const double di_pCa_dCai = p[g_pCa] * p[K_pCa] * rpCa * rpCa;
    I_tot += i_CaL + i_p_Ca;
    BEAT_FENCE();

    // ---- membrane, sodium, potassium (.ode:318-322) ---------------------------------------------------------------
    const double tmod = t - floor(t / p[stim_period]) * p[stim_period];
    const double i_Stim =
        (tmod >= p[stim_start] && tmod <= p[stim_start] + p[stim_duration]) ? p[stim_amplitude] : 0.0;
    {
      const double J_V = -dI_dV;
      // dE_K/dK_i = -RTF/K_i, dE_Ks/dK_i = -RTF/(K_i + P_kna Na_i); currents depend on K_i only through them
      const double J_Ki = -(sum_du * q.RTF * rKi + gKs * q.RTF * rKs) * q.cVF;
      const double J_Nai = -((gNa + p[g_bna]) * q.RTF * rNai + 3.0 * di_NaK_dNai + 3.0 * di_NaCa_dNai) * q.cVF;
#if BEAT_TP06_PHI
      io.store(V, advance(fm, v, -(I_tot + i_Stim), J_V, dt));
      io.store(K_i, advance(fm, vKi, -(I_K + i_Stim - 2.0 * i_NaK) * q.cVF, J_Ki, dt));
      io.store(Na_i, advance(fm, vNai, -(i_Na_tot + 3.0 * i_NaK + 3.0 * i_NaCa) * q.cVF, J_Nai, dt));
#else
      double rJV, rJK, rJN;
      rcp3(guard(J_V), guard(J_Ki), guard(J_Nai), rJV, rJK, rJN);
      io.store(V, grl1r(fm, v, -(I_tot + i_Stim), J_V, rJV, dt));
      io.store(K_i, grl1r(fm, vKi, -(I_K + i_Stim - 2.0 * i_NaK) * q.cVF, J_Ki, rJK, dt));
      io.store(Na_i, grl1r(fm, vNai, -(i_Na_tot + 3.0 * i_NaK + 3.0 * i_NaCa) * q.cVF, J_Nai, rJN, dt));
#endif
    }
    BEAT_FENCE();

    // ---- calcium dynamics (.ode:298-316) -----------------------------------------------------------------------------
    const double vR = STASH_SLOTS >= 1 ? beat_unstash(io, 0, vR0) : vR0;
    const double vCaSR = STASH_SLOTS >= 2 ? beat_unstash(io, 1, vCaSR0) : vCaSR0;
    const double qup = q.Kup2 * rCai * rCai;
    double rup, rbc, rCaSR, rbsr, rbss;
    rcp3(1.0 + qup, vCai + p[K_buf_c], vCaSR, rup, rbc, rCaSR);
    rcp2(vCaSR + p[K_buf_sr], vCass + p[K_buf_ss], rbsr, rbss);
    const double zsr = (p[EC] * rCaSR) * (p[EC] * rCaSR);
    const double gci = q.BKc * rbc * rbc, gsr = q.BKsr * rbsr * rbsr, gss = q.BKss * rbss * rbss;
    double rz, Fr_i, Fr_sr, Fr_ss;
    rcp4(1.0 + zsr, 1.0 + gci, 1.0 + gsr, 1.0 + gss, rz, Fr_i, Fr_sr, Fr_ss);
    const double i_up = p[Vmax_up] * rup;
    const double di_up_dCai = i_up * 2.0 * qup * rCai * rup;
    const double i_leak = p[V_leak] * (vCaSR - vCai);
    const double i_xfer = p[V_xfer] * (vCass - vCai);
    {  // Ca_i
      const double T_i = -(i_b_Ca + i_p_Ca - 2.0 * i_NaCa) * q.c1 + (i_leak - i_up) * q.c2 + i_xfer;
      const double dT_i = -(p[g_bca] * q.halfRTF * rCai + di_pCa_dCai - 2.0 * di_NaCa_dCai) * q.c1 +
                          (-p[V_leak] - di_up_dCai) * q.c2 - p[V_xfer];
      io.store(Ca_i, advance(fm, vCai, T_i * Fr_i, dT_i * Fr_i + T_i * (Fr_i * Fr_i * 2.0 * gci * rbc), dt));
    }
    BEAT_FENCE();
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
    io.store(R_prime, advance(fm, vR, -k2 * vCass * vR + p[k4] * (1.0 - vR), -vCass * k2 - p[k4], dt));
    {  // Ca_SR
      const double T_sr = i_up - (i_rel + i_leak);
      const double dT_sr = -(p[V_rel] * (dO_dk1 * dk1 * dsrss + O) + p[V_leak]);
      io.store(Ca_SR, advance(fm, vCaSR, T_sr * Fr_sr, dT_sr * Fr_sr + T_sr * (Fr_sr * Fr_sr * 2.0 * gsr * rbsr), dt));
    }
    BEAT_FENCE();
    {  // Ca_ss
      const double T_ss = -i_CaL * q.c3 + i_rel * q.c4 - i_xfer * q.c5;
      const double dT_ss = -di_CaL_dCass * q.c3 + p[V_rel] * (dO_dCass * dsrss - O) * q.c4 - p[V_xfer] * q.c5;
      io.store(Ca_ss, advance(fm, vCass, T_ss * Fr_ss, dT_ss * Fr_ss + T_ss * (Fr_ss * Fr_ss * 2.0 * gss * rbss), dt));
    }
  }
#undef BEAT_FENCE
};
#if defined(__clang__) && !defined(BEAT_TP06_NO_CONTRACT)
#pragma clang fp contract(off)
#endif
