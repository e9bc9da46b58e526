// gel_physics.h -- fp64 device functions of the 3-DoF rocket RHS for gfx950.
//
// What is computed follows the reference's production (C++) path; each function
// cites the reference lines whose arithmetic it reproduces.  How it is computed
// is device-first: values that several finite-difference sweeps share (the
// atmosphere at a position, sin/cos of the Earth angle, the wind vector in ECI,
// thrust direction, gravity) are produced once per node and kept in registers,
// instead of re-running the whole chain for every perturbed variable as the
// reference does (lib/con_dynamics.py:353-474 -> src/pybind_dynamics.cpp:42-68).
// Re-use is exact: a shared value is bit-identical to what a full re-evaluation
// would produce, because the perturbed variable does not enter it.
#pragma once
#include <hip/hip_runtime.h>

namespace gel {

#define GEL_DEV __device__ __forceinline__

// src/Earth.cpp:41-47
constexpr double kMu = 3.986004418e14;
constexpr double kOmega = 7.2921151467e-5;
constexpr double kRa = 6378137.0;
constexpr double kF = 1.0 / 298.257223563;
constexpr double kRb = kRa * (1.0 - kF);
constexpr double kE2 = (kRa * kRa - kRb * kRb) / kRa / kRa;
constexpr double kEp2 = (kRa * kRa - kRb * kRb) / kRb / kRb;
constexpr double kPi = 3.14159265358979323846;

// LDS-resident tables: US-1976 layers (src/Air.cpp:31-45) + wind + CA.
// atm is [7][11]: Lmb, Tmb, Pb, R (= Rstar / mb, src/Air.cpp:67), and the two per-layer constants of the
// pressure formula (src/Air.cpp:93-97) precomputed on the host with the reference's own operation
// order: pexp = -g0 / Lmb / R (lapse layers), gR = g0 / R (isothermal layers).
struct Tables {
  const double* atm;
  const double* wind;   // [Kw][3]
  const double* ca;     // [Kc][2]
  const double* winds;  // [Kw-1][2] slopes of the two wind components per table interval (host: (yu - yl) / (xu - xl))
  const double* cas;    // [Kc-1]    slopes of CA per Mach interval
  int Kw, Kc;
};
// Lmb, Tmb, Pb, R, pressure exponent, g0/R, layer base altitude, 1/Tmb: 11 layers each
constexpr int kAtmDoubles = 88;
// doubles of the staged tables: atmosphere | wind rows | CA rows | wind slopes | CA slopes
GEL_DEV constexpr int table_doubles(int Kw, int Kc) { return kAtmDoubles + 3 * Kw + 2 * Kc + 2 * (Kw - 1) + (Kc - 1); }

// ---------------------------------------------------------------------------
// fp64 square root and division without the range guards.
//
// hipcc expands sqrt(double) into v_rsq_f64 + a 9-operation Goldschmidt refinement wrapped in a
// 2^+-256 rescale (inputs below 2^-767) and a class test (0, inf, nan): 8 of its 19 instructions; and
// a/b into v_rcp_f64 + 7 fused operations wrapped in v_div_scale x2 / v_div_fmas / v_div_fixup, which
// only act when an exponent is within ~2^100 of the format's ends.  Everything on this path is a
// physical quantity in SI units or a scaled O(1) variable -- nowhere near either end -- so the
// refinement alone is used: the SAME operations in the SAME order, hence the same bits as the
// compiler's sequence for every finite positive normal operand (checked in tests/test_gpu_parity.py).
// fsqrt(0) is NaN; the one call site that can legitimately see 0 (air-relative speed) clamps first.
// -DGEL_STD_MATH restores the compiler's sequences (A/B and a fallback for doubts).
// ---------------------------------------------------------------------------
#ifndef GEL_STD_MATH
GEL_DEV double fsqrt(double x) {
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y, h = 0.5 * y;
  const double r = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, r, g);
  h = __builtin_fma(h, r, h);
  double d = __builtin_fma(-g, g, x);
  g = __builtin_fma(d, h, g);
  d = __builtin_fma(-g, g, x);
  return __builtin_fma(d, h, g);
}
GEL_DEV double fdiv(double a, double b) {
  double r = __builtin_amdgcn_rcp(b);
  double e = __builtin_fma(-b, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-b, r, 1.0);
  r = __builtin_fma(r, e, r);
  const double q = a * r;
  const double t = __builtin_fma(-b, q, a);
  return __builtin_fma(t, r, q);
}
#else
GEL_DEV double fsqrt(double x) { return sqrt(x); }
GEL_DEV double fdiv(double a, double b) { return a / b; }
#endif
GEL_DEV double frcp(double b) { return fdiv(1.0, b); }

// A wave-uniform double moved into an SGPR pair (two v_readfirstlane_b32): frees the VGPR pair it would occupy for the rest
// of the kernel; vector instructions read it as their scalar operand.
GEL_DEV double wave_uniform(double v) {
  const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v)), hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
  return __hiloint2double(hi, lo);
}
// x * s (s wave-uniform) as a product the compiler can neither merge with an earlier identical one nor hoist: the cheap way
// to NOT carry a scaled copy of a value in registers across the kernel -- recompute it where it is used.  Same bits as x * s.
GEL_DEV double fresh_product(double x, double s) {
  double r;
  asm volatile("v_mul_f64 %0, %1, %2" : "=v"(r) : "v"(x), "s"(s));
  return r;
}
GEL_DEV double fresh_mul(double a, double b) {   // likewise, both factors per-lane
  double r;
  asm volatile("v_mul_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// sqrt and 1/sqrt of the same argument from ONE Goldschmidt iteration: s is bit-identical to fsqrt(x); r comes
// from one more step on the companion value (<= 1 ulp from 1/sqrt(x)).  13 operations instead of the 18 of
// fsqrt + frcp, and one quarter-rate instruction (v_rsq_f64) instead of two (v_rsq_f64, v_rcp_f64).  Used where
// the reciprocal feeds nothing that is later differenced against a 6.4e6 m cancellation (see pos_part()).
GEL_DEV void fsqrt_rsqrt(double x, double& s, double& r) {
#ifndef GEL_STD_MATH
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y, h = 0.5 * y;
  const double r0 = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, r0, g);
  h = __builtin_fma(h, r0, h);
  double d = __builtin_fma(-g, g, x);
  g = __builtin_fma(d, h, g);
  d = __builtin_fma(-g, g, x);
  s = __builtin_fma(d, h, g);
  const double r1 = __builtin_fma(-h, s, 0.5);
  h = __builtin_fma(h, r1, h);
  r = h + h;
#else
  s = sqrt(x);
  r = 1.0 / s;
#endif
}
// 1/sqrt(x) alone (9 operations)
GEL_DEV double frsqrt(double x) {
#ifndef GEL_STD_MATH
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y, h = 0.5 * y;
  const double r0 = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, r0, g);
  h = __builtin_fma(h, r0, h);
  const double r1 = __builtin_fma(-h, g, 0.5);
  h = __builtin_fma(h, r1, h);
  return h + h;
#else
  return 1.0 / sqrt(x);
#endif
}

// ---------------------------------------------------------------------------
// log of the temperature ratio of a lapse layer, and sin/cos of a latitude or of half the Earth angle.
// tools/microbench/piece_time.hip times every piece of the chain at the kernel's occupancy: ocml's fp64 log
// costs 164 ns per call and SIMD (4x its exp), sincos 132 ns -- together a quarter of pos_part().
//
// flog_ratio: inside a layer's own altitude range the ratio lies in (0.65, 1.3); there
//   log x = 2 atanh(z), z = (x - 1)/(x + 1)   (x - 1 exact by Sterbenz, |z| <= 1/3 for x in [0.5, 2])
// as the odd series to z^33 (truncation < 3e-18 relative), ~1.5 ulp, 59 ns.  Anything else (the top layer is
// extrapolated upwards without bound, src/Air.cpp:56-61) takes the library's log.
// fsincos: |x| <= 3 pi/4 needs at most one subtraction of pi/2, exact against a two-part pi/2 with the rounding
// tail carried into the classic k_sin / k_cos minimax kernels on [-pi/4, pi/4]; 1 ulp like the library's
// (3.0 % vs 3.1 % of results differ from glibc's on 2.6 M arguments), 84 ns.  Larger arguments: the library's.
// -DGEL_STD_MATH restores log() / sincos() everywhere.
// ---------------------------------------------------------------------------
// One Horner step p * w + c with the coefficient as a SCALAR operand.  hipcc compiles __builtin_fma(p, w, literal) to
// v_fmac_f64 with the literal first moved into the destination VGPR pair -- two v_mov_b32 per term, which on a
// kernel bound by vector issue doubles the cost of every polynomial.  With the coefficient in an SGPR pair (two s_mov_b32
// on the scalar unit) the step is the one v_fma_f64.  Same operation, same bits.  -DGEL_HORNER_VGPR restores the plain form.
#ifndef GEL_HORNER_VGPR
GEL_DEV double horner(double p, double w, double c) {
  double r;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(p), "v"(w), "s"(c));
  return r;
}
#else
GEL_DEV double horner(double p, double w, double c) { return __builtin_fma(p, w, c); }
#endif

// exp(x) for the arguments of this path (pressure ratios: |x| <= 40; the thermosphere's temperature law: -3 < x <= 0).  The library's own
// algorithm (ocml expD: n = rint(x log2 e), t = x - n ln 2 in two parts, degree-11 polynomial, ldexp) with its coefficients as scalar
// operands (horner()) and without the overflow / underflow selects that cannot act for |x| < 700: the same operations in the same
// order -- the library's bits (tests/test_gpu_parity.py) -- in 24 vector instructions instead of 45.  A wavefront with a lane
// outside (-700, 700) takes the library's.
GEL_DEV double fexp(double x) {
#ifndef GEL_STD_MATH
#define GEL_F64(bits) __builtin_bit_cast(double, (unsigned long long)(bits))   // the library's constants, bit for bit
  if (__builtin_amdgcn_ballot_w64(!(fabs(x) < 700.0)) == 0) {
    const double n = __builtin_rint(x * GEL_F64(0x3ff71547652b82feULL));     // log2 e
    double t = __builtin_fma(n, GEL_F64(0xbfe62e42fefa39efULL), x);          // - ln 2, high part
    t = __builtin_fma(n, GEL_F64(0xbc7abc9e3b39803fULL), t);                 // - ln 2, low part
    double p = __builtin_fma(t, GEL_F64(0x3e5ade156a5dcb37ULL), GEL_F64(0x3e928af3fca7ab0cULL));
    p = horner(p, t, GEL_F64(0x3ec71dee623fde64ULL));
    p = horner(p, t, GEL_F64(0x3efa01997c89e6b0ULL));
    p = horner(p, t, GEL_F64(0x3f2a01a014761f6eULL));
    p = horner(p, t, GEL_F64(0x3f56c16c1852b7b0ULL));
    p = horner(p, t, GEL_F64(0x3f81111111122322ULL));
    p = horner(p, t, GEL_F64(0x3fa55555555502a1ULL));
    p = horner(p, t, GEL_F64(0x3fc5555555555511ULL));
    p = horner(p, t, GEL_F64(0x3fe000000000000bULL));
    p = __builtin_fma(t, p, 1.0);
    p = __builtin_fma(t, p, 1.0);
    return __builtin_ldexp(p, (int)n);
  }
#undef GEL_F64
#endif
  return exp(x);
}

GEL_DEV double flog_ratio(double x) {
#ifndef GEL_STD_MATH
  if (x > 0.5 && x < 2.0) {
    const double z = fdiv(x - 1.0, x + 1.0), w = z * z;
    double p = 1.0 / 33.0;
    p = horner(p, w, 1.0 / 31.0); p = horner(p, w, 1.0 / 29.0); p = horner(p, w, 1.0 / 27.0);
    p = horner(p, w, 1.0 / 25.0); p = horner(p, w, 1.0 / 23.0); p = horner(p, w, 1.0 / 21.0);
    p = horner(p, w, 1.0 / 19.0); p = horner(p, w, 1.0 / 17.0); p = horner(p, w, 1.0 / 15.0);
    p = horner(p, w, 1.0 / 13.0); p = horner(p, w, 1.0 / 11.0); p = horner(p, w, 1.0 / 9.0);
    p = horner(p, w, 1.0 / 7.0); p = horner(p, w, 1.0 / 5.0); p = horner(p, w, 1.0 / 3.0);
    return __builtin_fma(2.0 * z * w, p, 2.0 * z);
  }
#endif
  return log(x);
}

GEL_DEV void fsincos(double x, double* sn, double* cs) {
#ifdef GEL_STD_MATH
  sincos(x, sn, cs);
#else
  const double ax = fabs(x);
  if (ax > 2.35619449019234492885) { sincos(x, sn, cs); return; }  // 3 pi / 4
  const double n = (ax > 0.78539816339744830962) ? 1.0 : 0.0;       // pi / 4
  const double kPio2Hi = 1.57079632679489655800e+00, kPio2Lo = 6.12323399573676603587e-17;
  const double hi = __builtin_fma(-n, kPio2Hi, ax);                  // exact
  const double r = __builtin_fma(-n, kPio2Lo, hi);
  const double y = __builtin_fma(-n, kPio2Lo, hi - r);               // r + y = |x| - n pi/2 to ~1e-33
  const double z = r * r;
  const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
               S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  const double v = z * r;
  const double ps = horner(horner(horner(horner(S6, z, S5), z, S4), z, S3), z, S2);
  const double ks = r - ((z * (0.5 * y - v * ps) - y) - v * S1);
  const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
               C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  const double pc = z * horner(horner(horner(horner(horner(C6, z, C5), z, C4), z, C3), z, C2), z, C1);
  const double hz = 0.5 * z, w = 1.0 - hz;
  const double kc = w + (((1.0 - w) - hz) + (z * pc - r * y));
  const bool q = n != 0.0;        // |x| = n pi/2 + (r + y):  n = 0 -> (ks, kc);  n = 1 -> (kc, -ks); sin is odd in x
  *sn = copysign(q ? kc : ks, x);
  *cs = q ? -ks : kc;
#endif
}

// ---------------------------------------------------------------------------
// US Standard Atmosphere 1976.  One layer search serves T, P, rho and a; the
// reference repeats it five times per node (src/Air.cpp:100-111).
// ---------------------------------------------------------------------------
GEL_DEV double geopotential_altitude(double z) {  // src/Air.cpp:47-54
  return (z < 86000.0) ? fdiv(1.0 * (6356766.0 * z), 6356766.0 + z) : z;
}

GEL_DEV int us76_layer(double h) {  // src/Air.cpp:56-61
  int k = 0;
  k = (h >= 11000.0) ? 1 : k;
  k = (h >= 20000.0) ? 2 : k;
  k = (h >= 32000.0) ? 3 : k;
  k = (h >= 47000.0) ? 4 : k;
  k = (h >= 51000.0) ? 5 : k;
  k = (h >= 71000.0) ? 6 : k;
  k = (h >= 86000.0) ? 7 : k;
  k = (h >= 91000.0) ? 8 : k;
  k = (h >= 110000.0) ? 9 : k;
  k = (h >= 120000.0) ? 10 : k;
  return k;
}

struct Air { double T, P, rho, a, inv_a; };

GEL_DEV Air atmosphere(double h, const double* atm) {
  const int k = us76_layer(h);
  const double Hb = atm[66 + k];  // layer base altitude from the table (a 30-instruction select chain otherwise)
  const double Lmb = atm[k], Tmb = atm[11 + k], Pb = atm[22 + k], R = atm[33 + k];
  const double r0 = 6356766.0;
  Air o;
  // temperature: src/Air.cpp:71-88
  if (h <= 91000.0) {
    o.T = Tmb + Lmb * (h - Hb);
  } else if (h <= 110000.0) {
    const double Tc = 263.1905, A = -76.3232, a = -19942.9;
    o.T = Tc + A * fsqrt(1.0 - fdiv(fdiv((h - 91000.0) * (h - 91000.0), a), a));   // fdiv: the compiler's division without its range guards, same bits
  } else if (h <= 120000.0) {
    o.T = Tmb + Lmb * (h - Hb);
  } else {
    const double Tinf = 1000.0;
    const double xi = fdiv((h - Hb) * (r0 + Hb), r0 + h);
    o.T = Tinf - (Tinf - Tmb) * fexp(-0.01875e-3 * xi);
  }
  // pressure: src/Air.cpp:90-98
  if (fabs(Lmb) > 1.0e-6) {
    // (T/Tmb)^(-g0/Lmb/R), exponent from the table.  Inside a layer the base lies in (0.65, 1.3) and the exponent
    // is a per-layer constant of magnitude <= 35, so exp(y*log(x)) is within ~|y log x| ulp (<= 10) of pow(x, y)
    // at a quarter of its cost (piece_time.hip: pow 336 ns, exp(y*log x) 201 ns, with flog_ratio 95 ns).
    // the temperature ratio by the tabulated 1/Tmb (<= 1 ulp from the division; amplified by the exponent, |y| <= 35,
    // that stays inside the exp(y log x) budget above)
    o.P = Pb * fexp(atm[44 + k] * flog_ratio((Tmb + Lmb * (h - Hb)) * atm[77 + k]));
  } else {
    o.P = Pb * fexp((atm[55 + k] * (Hb - h)) * atm[77 + k]);           // g0/R and 1/Tmb from the table
  }
  o.rho = fdiv(o.P, R * o.T);     // src/Air.cpp:100-105 (P/R/T)
  fsqrt_rsqrt(1.4 * R * o.T, o.a, o.inv_a);   // src/Air.cpp:107-111; 1/a for the Mach number
  return o;
}

// ---------------------------------------------------------------------------
// geodesy: src/Earth.cpp:49-61 (Bowring one step)
// ---------------------------------------------------------------------------
// theta = atan2(z Ra, p Rb) is only used through sin(theta), cos(theta): they are formed algebraically
// (a/h, b/h with h = hypot(a, b)), identical up to rounding to sincos(atan2(a, b)).
// inv_p = 1/p (<= 1 ulp; it only scales the longitude pair of wind_eci()); exactly on the polar axis p = 0 and
// inv_p = 0, which wind_eci() reads as longitude 0 = the reference's atan2(0, 0)
// Optional exports for the exact-difference position sweeps (pos_delta()): 1/hypot(z Ra, p Rb) and 1/hypot(zz, pp) of the
// latitude's atan2 arguments.
GEL_DEV void geodetic_lat_p(double x, double y, double z, double& lat, double& p, double& inv_p, double* ih_out = nullptr,
                            double* ihy_out = nullptr) {
  const double p2 = x * x + y * y;
  fsqrt_rsqrt(fmax(p2, 1.0e-300), p, inv_p);
  if (!(p2 > 0.0)) { p = 0.0; inv_p = 0.0; }
  const double a = z * kRa, b = p * kRb;
  // only sin/cos(theta) = a/h, b/h are used, and they enter the latitude through the 0.7 % Bowring correction
  // terms: the reciprocal root directly (<= 1 ulp on st, ct = <= 0.01 ulp on the arguments of atan2)
  const double h2 = a * a + b * b;
  const double ih = frsqrt(fmax(h2, 1.0e-300));
  const double st = (h2 > 0.0) ? a * ih : 0.0;
  const double ct = (h2 > 0.0) ? b * ih : 1.0;
  const double zz = z + kEp2 * kRb * (st * st * st), pp = p - kE2 * kRa * (ct * ct * ct);
  lat = atan2(zz, pp);
  if (ih_out) *ih_out = ih;
  if (ihy_out) *ihy_out = frsqrt(fmax(zz * zz + pp * pp, 1.0e-300));
}

// Bowring's one-step latitude (src/Earth.cpp:49-57) delivered as (sin lat, cos lat): the reference forms lat = atan2(zz, pp)
// and then only ever uses sin / cos of it (and of lat / 2), so the pair is taken directly as (zz, pp) / hypot(zz, pp) -- the
// same two numbers up to 2 ulp, without the atan2 -> sincos round trip (166 fp64 operations of the 507 of the position part).
// Rounds 1-2 kept the round trip because the position SWEEPS differenced the altitude p / cos(lat) - N computed from the
// pair, so its rounding had to correlate with the reference's; the sweeps now take the exact-difference form (pos_delta) and
// no longer see how the centre pair was rounded.  pp == 0 (the polar axis, where the reference gets cos(pi/2) = 6.1e-17 and
// an altitude of -N) returns the reference's pair.  ih_out / ihy_out: 1/hypot(z Ra, p Rb) and 1/hypot(zz, pp) for pos_delta().
// the head of geodetic_sincos_p() on its own: p, 1/p and 1/hypot(z Ra, p Rb) -- the same statements, hence the same bits.  The
// AERO instantiation of the fused kernel forms them again in every position sweep instead of parking them (their three park slots
// hold the aero rows' centre values there).
GEL_DEV void geodetic_p_ih(double x, double y, double z, double& p, double& inv_p, double& ih) {
  const double p2 = x * x + y * y;
  fsqrt_rsqrt(fmax(p2, 1.0e-300), p, inv_p);
  if (!(p2 > 0.0)) { p = 0.0; inv_p = 0.0; }
  const double a = z * kRa, b = p * kRb;
  const double h2 = a * a + b * b;
  ih = frsqrt(fmax(h2, 1.0e-300));
}
GEL_DEV void geodetic_sincos_p(double x, double y, double z, double& sl, double& cl, double& p, double& inv_p,
                               double* ih_out = nullptr, double* ihy_out = nullptr) {
  const double p2 = x * x + y * y;
  fsqrt_rsqrt(fmax(p2, 1.0e-300), p, inv_p);
  if (!(p2 > 0.0)) { p = 0.0; inv_p = 0.0; }
  const double a = z * kRa, b = p * kRb;
  const double h2 = a * a + b * b;
  const double ih = frsqrt(fmax(h2, 1.0e-300));
  const double st = (h2 > 0.0) ? a * ih : 0.0;
  const double ct = (h2 > 0.0) ? b * ih : 1.0;
  const double zz = z + kEp2 * kRb * (st * st * st), pp = p - kE2 * kRa * (ct * ct * ct);
  const double ihy = frsqrt(fmax(zz * zz + pp * pp, 1.0e-300));
  sl = zz * ihy;
  cl = pp * ihy;
  if (pp == 0.0) {   // sincos(atan2(zz, 0)): (+-1, cos(pi/2) as fp64 has it); atan2(0, 0) = 0
    sl = (zz == 0.0) ? 0.0 : copysign(1.0, zz);
    cl = (zz == 0.0) ? 1.0 : 6.123233995736766e-17;
  }
  if (ih_out) *ih_out = ih;
  if (ihy_out) *ihy_out = ihy;
}

// altitude from (p, sin lat, cos lat): src/Earth.cpp:58-59
GEL_DEV double geodetic_alt_from(double p, double sl, double cl) {
  const double N = fdiv(kRa, fsqrt(1.0 - kE2 * sl * sl));
  return fdiv(p, cl) - N;
}

GEL_DEV double geodetic_altitude(double x, double y, double z) {
  double lat, p, ip;
  geodetic_lat_p(x, y, z, lat, p, ip);
  double sl, cl;
  fsincos(lat, &sl, &cl);
  return geodetic_alt_from(p, sl, cl);
}

GEL_DEV void geodetic_full(double x, double y, double z, double& lat, double& lon, double& alt) {
  double p, ip;
  geodetic_lat_p(x, y, z, lat, p, ip);
  lon = atan2(y, x);
  double sl, cl;
  fsincos(lat, &sl, &cl);
  const double N = fdiv(kRa, fsqrt(1.0 - kE2 * sl * sl));
  alt = fdiv(p, cl) - N;
}

// ---------------------------------------------------------------------------
// J2 gravity: src/gravity.cpp:11-57
// ---------------------------------------------------------------------------
GEL_DEV void gravity_eci(const double r3[3], double barC20, double g[3]) {
  const double a = 6378137.0, mu = kMu;
  const double b = a * (1.0 - 1.0 / 298.257223563);
  const double x = r3[0], y = r3[1], z = r3[2];
  double r, inv_r;  // one reciprocal root serves x/r, y/r, z/r, a/r, mu/r^2 (each <= 1 ulp from the division)
  fsqrt_rsqrt(fmax(x * x + y * y + z * z, 1.0e-300), r, inv_r);
  double irx = 0.0, iry = 0.0, irz = 0.0;
  if (r > 1.0e-150) { irx = x * inv_r; iry = y * inv_r; irz = z * inv_r; }
  const double s5 = 2.23606797749978969641;  // sqrt(5.0)
  const double barP20 = s5 * (3.0 * irz * irz - 1.0) * 0.5;
  const double barP20d = s5 * 3.0 * irz;
  if (r < b) { r = b; inv_r = 1.0 / b; }
  const double mur2 = mu * (inv_r * inv_r);
  const double ar = a * inv_r;
  const double g_ir = -mur2 * (1.0 + barC20 * ar * ar * (3.0 * barP20 + irz * barP20d));
  const double g_iz = mur2 * ar * ar * barC20 * barP20d;
  g[0] = g_ir * irx;
  g[1] = g_ir * iry;
  g[2] = g_ir * irz + g_iz;
}

// ---------------------------------------------------------------------------
// quaternions: src/wrapper_coordinate.hpp:50-78
// ---------------------------------------------------------------------------
GEL_DEV void quatmult(const double q[4], const double p[4], double o[4]) {
  o[0] = q[0] * p[0] - q[1] * p[1] - q[2] * p[2] - q[3] * p[3];
  o[1] = q[0] * p[1] + q[1] * p[0] + q[2] * p[3] - q[3] * p[2];
  o[2] = q[0] * p[2] - q[1] * p[3] + q[2] * p[0] + q[3] * p[1];
  o[3] = q[0] * p[3] + q[1] * p[2] - q[2] * p[1] + q[3] * p[0];
}

// quatrot(q, v) = vec( conj(q) * (0,v) * q )
GEL_DEV void quatrot(const double q[4], const double v[3], double out[3]) {
  const double vq[4] = {0.0, v[0], v[1], v[2]};
  const double qc[4] = {q[0], -q[1], -q[2], -q[3]};
  double t1[4], r[4];
  quatmult(vq, q, t1);
  quatmult(qc, t1, r);
  out[0] = r[1]; out[1] = r[2]; out[2] = r[3];
}

// ---------------------------------------------------------------------------
// clamped linear interpolation: src/wrapper_utils.hpp:51-80, with np.interp's
// value at x == xp[0].  Tables live in LDS.
// ---------------------------------------------------------------------------
// std::lower_bound on a sorted LDS column = number of entries < x.  Small tables (the usual case: 7-9
// rows) are counted with independent broadcast reads -- no dependent LDS round trips, no divergent
// loop; long tables fall back to the binary search.
GEL_DEV int lower_count(double x, const double* tab, int n, int stride) {
  int lo = 0;
  if (n <= 32) {
    for (int i = 0; i < n; i++) lo += (tab[i * stride] < x) ? 1 : 0;
  } else {
    int hi = n;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (tab[mid * stride] < x) lo = mid + 1; else hi = mid;
    }
  }
  return lo;
}

// yl + alpha (yu - yl), alpha = (x - xl)/(xu - xl), as yl + (x - xl) * slope with the slope (yu - yl)/(xu - xl) of the
// interval tabulated on the host: one division and one LDS read less per lookup, <= 2 ulp of the increment away.
GEL_DEV double interp_tab(double x, const double* tab, const double* slope, int n, int stride, int ycol) {
  const int idx = min(max(lower_count(x, tab, n, stride) - 1, 0), n - 2);
  const double v = tab[idx * stride + ycol] + (x - tab[idx * stride]) * slope[idx];
  // clamps of the reference (and yp[0] at x == xp[0]) as selects instead of early returns
  return (x <= tab[0]) ? tab[ycol] : ((x > tab[(n - 1) * stride]) ? tab[(n - 1) * stride + ycol] : v);
}

// The interval of the last lookup, kept by the caller across the sweeps of one node: a perturbed evaluation moves the
// abscissa by 1e-8 of itself, so it falls into the same interval (xl < x <= xu) almost surely; then the count over the
// table and the clamps are skipped.  The same interval means the same operands in the same expression: bit-identical to the
// full lookup.  A wavefront takes the short way only if all of its lanes can.  Only the interval's INDEX is kept (one
// register): its four numbers are read from the LDS table again -- they used to occupy eight registers for the whole kernel.
struct Bracket { int idx; };
GEL_DEV Bracket no_bracket() { return Bracket{-1}; }
GEL_DEV double interp_tab_cached(double x, const double* tab, const double* slope, int n, int stride, int ycol, Bracket& br) {
  {
    const int i = max(br.idx, 0);
    const double xl = tab[i * stride], xu = tab[(i + 1) * stride];
    const bool hit = (br.idx >= 0) && (xl < x) && (x <= xu);
    if (__builtin_amdgcn_ballot_w64(!hit) == 0) return tab[i * stride + ycol] + (x - xl) * slope[i];   // wave-uniform branch
  }
  const int idx = min(max(lower_count(x, tab, n, stride) - 1, 0), n - 2);
  br.idx = idx;
  const double v = tab[idx * stride + ycol] + (x - tab[idx * stride]) * slope[idx];
  return (x <= tab[0]) ? tab[ycol] : ((x > tab[(n - 1) * stride]) ? tab[(n - 1) * stride + ycol] : v);
}

// both wind components share one bracket search (src/wrapper_utils.hpp:82-87 runs it twice)
GEL_DEV void wind_ned2(double h, const double* tab, const double* slope, int n, double& wn, double& we) {
  const int idx = min(max(lower_count(h, tab, n, 3) - 1, 0), n - 2);
  const double dxl = h - tab[idx * 3];
  const double vn = tab[idx * 3 + 1] + dxl * slope[2 * idx];
  const double ve = tab[idx * 3 + 2] + dxl * slope[2 * idx + 1];
  const bool below = h <= tab[0], above = h > tab[(n - 1) * 3];
  wn = below ? tab[1] : (above ? tab[(n - 1) * 3 + 1] : vn);
  we = below ? tab[2] : (above ? tab[(n - 1) * 3 + 2] : ve);
}

// wind_ned2 with the altitude interval of the node's previous lookup (see Bracket): the three position sweeps and the
// centre evaluation of a node sit within 1e-8 of each other
struct Bracket2 { double xl, xu, y0, y1, s0, s1; };
GEL_DEV Bracket2 no_bracket2() { return Bracket2{1.79769313486231570815e308, -1.79769313486231570815e308, 0.0, 0.0, 0.0, 0.0}; }
GEL_DEV void wind_ned2_cached(double h, const double* tab, const double* slope, int n, double& wn, double& we, Bracket2& br) {
  const bool hit = (br.xl < h) && (h <= br.xu);
  if (__builtin_amdgcn_ballot_w64(!hit) == 0) {   // wave-uniform branch
    const double dxl = h - br.xl;
    wn = br.y0 + dxl * br.s0;
    we = br.y1 + dxl * br.s1;
    return;
  }
  const int idx = min(max(lower_count(h, tab, n, 3) - 1, 0), n - 2);
  br.xl = tab[idx * 3]; br.xu = tab[idx * 3 + 3]; br.y0 = tab[idx * 3 + 1]; br.y1 = tab[idx * 3 + 2];
  br.s0 = slope[2 * idx]; br.s1 = slope[2 * idx + 1];
  const double dxl = h - br.xl;
  const double vn = br.y0 + dxl * br.s0;
  const double ve = br.y1 + dxl * br.s1;
  const bool below = h <= tab[0], above = h > tab[(n - 1) * 3];
  wn = below ? tab[1] : (above ? tab[(n - 1) * 3 + 1] : vn);
  we = below ? tab[2] : (above ? tab[(n - 1) * 3 + 2] : ve);
}

// wind_ned2 for the centre evaluation of a node whose position sweeps take the exact-difference form: the same value
// expressions (bit-identical to wind_ned2), plus the index of the node's piece of the table: >= 0 inside the table,
// -1 below its first row, -2 above its last (the two clamped ends, where the wind does not change with altitude).
GEL_DEV void wind_ned2_centre(double h, const double* tab, const double* slope, int n, double& wn, double& we, int& piece) {
  const int idx = min(max(lower_count(h, tab, n, 3) - 1, 0), n - 2);
  const double dxl = h - tab[idx * 3];
  const double vn = tab[idx * 3 + 1] + dxl * slope[2 * idx];
  const double ve = tab[idx * 3 + 2] + dxl * slope[2 * idx + 1];
  const bool below = h <= tab[0], above = h > tab[(n - 1) * 3];
  wn = below ? tab[1] : (above ? tab[(n - 1) * 3 + 1] : vn);
  we = below ? tab[2] : (above ? tab[(n - 1) * 3 + 2] : ve);
  piece = below ? -1 : (above ? -2 : idx);
}
// What a position sweep in exact-difference form reads back from (h, piece): the wind at h (the same expressions: bit-identical
// to the centre's), the slopes of its piece (zero in the clamped ends) and how far h may move without leaving the piece.
// Branch-free: every operand is read, then selected.
GEL_DEV void wind_piece(double h, int piece, const double* tab, const double* slope, int n, double& wn, double& we, double& s0,
                        double& s1, double& margin) {
  const int idx = max(piece, 0);
  const double xl = tab[idx * 3], xu = tab[idx * 3 + 3];
  const double sl0 = slope[2 * idx], sl1 = slope[2 * idx + 1];
  const double dxl = h - xl;
  const double vn = tab[idx * 3 + 1] + dxl * sl0;
  const double ve = tab[idx * 3 + 2] + dxl * sl1;
  const double x0 = tab[0], xe = tab[(n - 1) * 3];
  const bool below = piece == -1, above = piece == -2, in = piece >= 0;
  wn = below ? tab[1] : (above ? tab[(n - 1) * 3 + 1] : vn);
  we = below ? tab[2] : (above ? tab[(n - 1) * 3 + 2] : ve);
  s0 = in ? sl0 : 0.0;
  s1 = in ? sl1 : 0.0;
  margin = below ? x0 - h : (above ? h - xe : fmin(dxl, xu - h));
}

// ---------------------------------------------------------------------------
// Exact-difference atmosphere: the state at h + dh from the state at h, for a dh that stays inside the layer (the caller
// checks that).  Every change is an algebraic identity of the reference's formula (src/Air.cpp:71-111), not a recomputation:
//   T linear in h                         dT = Lmb dh
//   T = Tc + A sqrt(1 - y^2) (91..110 km) ds = s (sqrt(1 + us) - 1), us = -d(y^2)/s^2
//   T = Tinf - (Tinf - Tmb) exp(-l xi)    dT = (T - Tinf) expm1(-l dxi)
//   P = Pb (Tl/Tmb)^y                     P'/P = exp(y log1p(dTl/Tl));   P = Pb exp(gR (Hb - h)/Tmb): P'/P = exp(-gR dh/Tmb)
//   rho = P/(R T)                         rho'/rho = (1 + uP)/(1 + uT);   1/a = (1.4 R T)^-1/2: (1 + uT)^-1/2
// with the series of log1p / expm1 / (1 + u)^-1/2 cut where the next term is below 1e-13 of the change for |dh| <= 1 m
// (gel_host.hip selects the recomputing form for larger steps).  The result differs from the reference's recomputation
// by the rounding noise of THAT (1e-16 of the value = 1e-8 .. 1e-4 of the change); tests/test_delta_model.py has the model.
// ---------------------------------------------------------------------------
struct AirCentre { double iT, h; int k; };   // what the difference form needs besides Air: 1/T, geopotential altitude, layer
// geopotential altitude of 86 km geometric, the end of the geopotential branch (src/Air.cpp:47-54), rounded DOWN: a node of layer 6
// (71 .. 86 km geopotential) below it has altitude < 86 km
constexpr double kGeopot86 = 84852.0;
GEL_DEV void atmosphere_delta(const Air& c, const AirCentre& ac, double dh, const double* atm, double& P1, double& rho1, double& inv_a1) {
  const int k = ac.k;
  const double r0 = 6356766.0;
  const double Lmb = atm[k];
  const double dTl = Lmb * dh;
  double dT = dTl, iTl = ac.iT;   // h <= 91 km and 110 .. 120 km: T is the layer's linear profile
  if (k == 8) {                   // 91 .. 110 km: elliptic profile (src/Air.cpp:75-78); the pressure keeps the linear one
    const double ia = 1.0 / -19942.9;
    const double yv = (ac.h - 91000.0) * ia, dyv = dh * ia;
    const double dy2 = dyv * (2.0 * yv + dyv);
    const double s = (frcp(ac.iT) - 263.1905) * (1.0 / -76.3232);
    const double us = -dy2 * frcp(s * s);
    dT = (-76.3232 * s) * (us * (0.5 + us * (-0.125 + 0.0625 * us)));
    iTl = frcp(atm[11 + k] + Lmb * (ac.h - atm[66 + k]));
  } else if (k == 10) {           // above 120 km: exponential profile (:83-87)
    const double Hb = atm[66 + k];
    const double irh = frcp(r0 + ac.h);
    const double dxi = ((r0 + Hb) * (r0 + Hb)) * dh * irh * irh * (1.0 - dh * irh);
    const double zc = -0.01875e-3 * dxi;
    dT = (frcp(ac.iT) - 1000.0) * (zc * (1.0 + zc * (0.5 + zc * (1.0 / 6.0))));
    iTl = frcp(atm[11 + k] + Lmb * (ac.h - Hb));
  }
  const double uT = dT * ac.iT;
  const double ul = dTl * iTl;
  const double e_lapse = atm[44 + k] * (ul * (1.0 + ul * (-0.5 + ul * (1.0 / 3.0))));
  const double e_iso = -(atm[55 + k] * atm[77 + k]) * dh;
  const double ee = (fabs(Lmb) > 1.0e-6) ? e_lapse : e_iso;
  const double uP = ee * (1.0 + ee * (0.5 + ee * (1.0 / 6.0)));
  P1 = __builtin_fma(c.P, uP, c.P);
  rho1 = __builtin_fma(c.rho, (uP - uT) * (1.0 + uT * (uT - 1.0)), c.rho);
  inv_a1 = __builtin_fma(c.inv_a, uT * (-0.5 + uT * (0.375 - 0.3125 * uT)), c.inv_a);
}

}  // namespace gel
