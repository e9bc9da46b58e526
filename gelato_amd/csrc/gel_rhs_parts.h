// gel_rhs_parts.h -- the air RHS split by what each piece depends on, so that every
// finite-difference sweep recomputes only what its perturbed variable enters.
//
//   pos_part(r)            atmosphere, wind (NED), gravity, geodetic latitude   <- position only
//   earth_angle(t)         cos/sin(omega t), cos/sin(omega t / 2)               <- time only
//   wind_eci(r, ea, ...)   wind rotated NED -> ECI                              <- position, time
//   aero_force(...)        axial aerodynamic force                               <- velocity + the above
//   accel(...)             (thrust + aero)/m + g, normalised                     <- mass, thrust direction
//
// Arithmetic policy (fp64, no fast-math): each piece follows the reference's formulas; where
// the reference's C++ divides by a loop-invariant (mass, unit_vel, sqrt(2), dx) the device
// multiplies by its once-computed reciprocal.  That changes a result by <= 1 ulp -- the same
// class of difference as ocml-vs-glibc libm -- and is applied identically to the centre and the
// perturbed evaluations, so finite-difference quotients see no extra noise.
#pragma once
#include "gel_physics.h"

#ifndef GEL_CALM_SHORTCUT
#define GEL_CALM_SHORTCUT 1
#endif

namespace gel {

// depends on position only
struct PosPart {
  double rho, P, inv_a;  // atmosphere at the node: density, pressure, 1 / speed of sound
  double wn, we;     // wind, NED
  double g[3];       // gravity, ECI
  double shp, chp;   // sin, cos of half the geodetic latitude (for the NED quaternion)
  double inv_p;      // 1 / sqrt(x^2 + y^2)
};

// Bowring's one-step geodetic latitude (src/Earth.cpp:49-57) delivered as (sin lat, cos lat): the
// reference forms lat = atan2(zz, pp) and then only ever uses sin/cos of it (and of lat/2), so the pair is
// taken directly as zz/hypot, pp/hypot -- the same two numbers up to rounding, without the atan2 -> sincos
// round trip.  Likewise sin/cos(theta) of theta = atan2(z Ra, p Rb).
GEL_DEV void geodetic_sincos(double x, double y, double z, double& sl, double& cl, double& p) {
  p = sqrt(x * x + y * y);
  const double a = z * kRa, b = p * kRb;
  const double ih = 1.0 / sqrt(a * a + b * b);
  const double st = a * ih, ct = b * ih;
  const double zz = z + kEp2 * kRb * (st * st * st);
  const double pp = p - kE2 * kRa * (ct * ct * ct);
  const double ihy = 1.0 / sqrt(zz * zz + pp * pp);
  sl = zz * ihy;
  cl = pp * ihy;
}

// What the exact-difference position sweeps (pos_delta) need of the centre evaluation besides PosPart.
struct PosCentre {
  double p, ih, ihy;     // sqrt(x^2 + y^2); 1/hypot(z Ra, p Rb) (Bowring's auxiliary angle); 1/hypot of the latitude's atan2 arguments
  double sl, cl, icl;    // sin, cos of the geodetic latitude; 1/cos
  double N;              // prime-vertical radius Ra / sqrt(1 - e^2 sin^2 lat)
  double G;              // d(geopotential altitude)/d(altitude) = (r0/(r0 + alt))^2 is G^2; 1 above 86 km
  double s0, s1;         // slopes of the wind components on the node's piece of the table (0 in the clamped ends)
  double margin;         // how far altitude / geopotential altitude may move without leaving the atmosphere layer, the wind
                         // table's piece or the geopotential branch
  AirCentre air;         // 1/temperature, geopotential altitude, layer
};
// Where pos_part() leaves the PosCentre members: the first eight go to the sink AS SOON AS they exist (the fused kernel parks
// them in LDS, so that they do not occupy registers across the atmosphere chain), the rest come back in the struct.
enum PosCentreSlot { PCS_P = 0, PCS_IH, PCS_IHY, PCS_SL, PCS_CL, PCS_ICL, PCS_N, PCS_G, PCS_COUNT };
// ... the rest is re-derived by pos_centre_tail() from three small items, where a sweep needs it
struct PosCentreTail { double h; int k, piece; };   // geopotential altitude, atmosphere layer, piece of the wind table
struct NoSink { GEL_DEV void put(int, double) const {} };
struct PosCentreSink {   // collects into a PosCentre (hooks, aero kernel)
  PosCentre* pc;
  GEL_DEV void put(int i, double v) const {
    if (i == PCS_P) pc->p = v; else if (i == PCS_IH) pc->ih = v; else if (i == PCS_IHY) pc->ihy = v; else if (i == PCS_SL) pc->sl = v;
    else if (i == PCS_CL) pc->cl = v; else if (i == PCS_ICL) pc->icl = v; else if (i == PCS_N) pc->N = v; else pc->G = v;
  }
};

// GRAV = false (aero path constraints): no gravity.
template <bool CENTRE = false, class Sink = NoSink, bool GRAV = true>
GEL_DEV PosPart pos_part(const double r[3], const Tables& tb, double barC20, Bracket2* wbr = nullptr, Sink sink = Sink(),
                         PosCentreTail* tail = nullptr) {
  PosPart o;
  // the reference feeds the ECI position to ecef2geodetic for altitude (src/pybind_dynamics.cpp:43)
  double p, sl, cl;
  if (CENTRE) {
    double ih, ihy;
    geodetic_sincos_p(r[0], r[1], r[2], sl, cl, p, o.inv_p, &ih, &ihy);
    sink.put(PCS_IH, ih); sink.put(PCS_IHY, ihy); sink.put(PCS_P, p);
  } else geodetic_sincos_p(r[0], r[1], r[2], sl, cl, p, o.inv_p);
  // half-angle pair of the NED quaternion (src/Coordinate.cpp:89-90): cos(lat/2) = sqrt((1+cos lat)/2)
  // (cos lat >= 0), sin(lat/2) = sin lat / (2 cos(lat/2)); root and reciprocal root from one iteration
  {
    double irt;
    fsqrt_rsqrt(0.5 * (1.0 + cl), o.chp, irt);
    o.shp = (0.5 * sl) * irt;
  }
  // src/Earth.cpp:58-59 (geodetic_alt_from(), spelled out for the exports)
  const double N = fdiv(kRa, fsqrt(1.0 - kE2 * sl * sl));
  const double q = fdiv(p, cl);
  const double alt = q - N;
  if (CENTRE) {
    sink.put(PCS_SL, sl); sink.put(PCS_CL, cl); sink.put(PCS_N, N);
    sink.put(PCS_ICL, q * o.inv_p);                                          // 1/cos(lat) = (p/cos lat)/p
    sink.put(PCS_G, (alt < 86000.0) ? fdiv(6356766.0, 6356766.0 + alt) : 1.0);
  }
  const double h = geopotential_altitude(alt);
  const Air air = atmosphere(h, tb.atm);
  o.rho = air.rho; o.P = air.P; o.inv_a = air.inv_a;
  // wind looked up at geopotential altitude (:44,49); wbr: the caller's altitude interval of this node's previous evaluations
  if (CENTRE) {
    wind_ned2_centre(h, tb.wind, tb.winds, tb.Kw, o.wn, o.we, tail->piece);
    tail->h = h; tail->k = us76_layer(h);
  } else if (wbr) wind_ned2_cached(h, tb.wind, tb.winds, tb.Kw, o.wn, o.we, *wbr);
  else wind_ned2(h, tb.wind, tb.winds, tb.Kw, o.wn, o.we);
  if (GRAV) gravity_eci(r, barC20, o.g);
  else o.g[0] = o.g[1] = o.g[2] = 0.0;
  return o;
}

// The members of PosCentre that are not parked, re-derived for a sweep from (h, layer, wind piece) and the centre's
// density and pressure: the wind slopes and the margin (how far the altitude may move inside the atmosphere layer -- for
// layer 6 also inside the geopotential branch, which ends at 84852 m -- and inside the piece of the wind table), and 1/T = rho R / P.
// Also returns the centre's wind (bit-identical to pos_part's), so that it need not live in registers either.
GEL_DEV void pos_centre_tail(const PosCentreTail& t, double rho, double P, const Tables& tb, PosCentre& pc, double& wn, double& we) {
  double wm;
  wind_piece(t.h, t.piece, tb.wind, tb.winds, tb.Kw, wn, we, pc.s0, pc.s1, wm);
  const double lo = (t.k == 0) ? -1.0e300 : tb.atm[66 + t.k];   // the bottom layer extends below sea level (src/Air.cpp:56-61)
  const double hi = (t.k == 6) ? kGeopot86 : ((t.k < 10) ? tb.atm[67 + min(t.k, 9)] : 1.0e300);
  pc.margin = fmin(wm, fmin(t.h - lo, hi - t.h));
  pc.air.iT = rho * tb.atm[33 + t.k] * frcp(P);
  pc.air.h = t.h; pc.air.k = t.k;
}

// ---------------------------------------------------------------------------
// Exact-difference form of pos_part() for a position sweep: component kk of r moved by dlt = r'_kk - r_kk (an exact
// floating-point difference).  The reference re-runs the whole chain on r' (lib/con_dynamics.py:381-400) and differences
// two values of 6.4e6 m magnitude; here the CHANGE of every intermediate is formed directly from an algebraic identity
// of the reference's own formula (src/Earth.cpp:49-61, src/Air.cpp:47-111, src/wrapper_utils.hpp:82-87):
//   p = sqrt(x^2 + y^2)          d(p^2) = dlt (2 x_k + dlt),  dp = p (sqrt(1 + u) - 1),  u = d(p^2)/p^2
//   sin/cos(theta) = a/h, b/h    dh2 = da (2a + da) + db (2b + db),  d(1/h) = (1/h)((1 + uh)^-1/2 - 1)
//   lat = atan2(zz, pp)          dlat = atan((dzz pp - zz dpp)/(zz zz' + pp pp'))   (tan of a difference)
//   sin, cos, half-angle pair    angle-addition with sin(dlat) = dlat, 1 - cos(dlat) = dlat^2/2
//   alt = p/cos(lat) - N         d(p/cl) = (dp cl - p dcl)/(cl cl'),  dN = N ((1 + uw)^-1/2 - 1)
//   h = r0 alt/(r0 + alt)        dh = r0^2 dalt/((r0 + alt)(r0 + alt'))
//   atmosphere, wind             atmosphere_delta(); linear piece of the table
// with every series cut where the next term is below 1e-13 of the change for |dlt| <= 1 m.  ~130 fp64 operations instead
// of ~450, and the change is accurate to ~1e-12 of itself, where the recomputation's is accurate to 1e-16 of the VALUE
// (1e-8 .. 1e-4 of the change: the reference's own finite-difference noise).  Gravity is recomputed by the caller.
// Returns false for a lane whose perturbed point leaves the centre's atmosphere layer / table piece / geopotential branch
// or sits too close to the polar axis for the series: the caller then recomputes the wavefront's sweep in full.
// ---------------------------------------------------------------------------
GEL_DEV bool pos_delta(const double r[3], int kk, double dlt, const PosPart& c, const PosCentre& pc, const Tables& tb, PosPart& o) {
  // kk is a constant where the sweeps are unrolled (wave-uniform otherwise): a step along z leaves p alone, a step along x or y
  // leaves z alone, and the terms that the other direction would contribute -- products with an exact zero, sums with it -- are
  // not formed (same values; only the sign of an exact zero can differ)
  const bool along_z = kk == 2;
  const double a = r[2] * kRa, b = pc.p * kRb;
  double u = 0.0, dp = 0.0, dh2, dst, dct, dih, ih1;
  o.inv_p = c.inv_p;
  if (!along_z) {
    const double xs = (kk == 0) ? r[0] : r[1];
    const double dp2 = dlt * __builtin_fma(2.0, xs, dlt);
    u = dp2 * c.inv_p * c.inv_p;
    dp = (0.5 * dp2 * c.inv_p) * (1.0 + u * (-0.25 + 0.125 * u));
    o.inv_p = c.inv_p * (1.0 + u * (-0.5 + 0.375 * u));
    // Bowring's auxiliary angle (src/Earth.cpp:53-55)
    const double db = dp * kRb;
    dh2 = db * __builtin_fma(2.0, b, db);
    const double uh = dh2 * pc.ih * pc.ih;
    dih = pc.ih * (uh * (-0.5 + 0.375 * uh));
    ih1 = pc.ih + dih;
    dst = a * dih;
    dct = db * ih1 + b * dih;
  } else {
    const double da = dlt * kRa;
    dh2 = da * __builtin_fma(2.0, a, da);
    const double uh = dh2 * pc.ih * pc.ih;
    dih = pc.ih * (uh * (-0.5 + 0.375 * uh));
    ih1 = pc.ih + dih;
    dst = da * ih1 + a * dih;
    dct = b * dih;
  }
  const double st = a * pc.ih, ct = b * pc.ih;
  const double dst3 = dst * (3.0 * st * (st + dst) + dst * dst);
  const double dct3 = dct * (3.0 * ct * (ct + dct) + dct * dct);
  const double dzz = along_z ? dlt + (kEp2 * kRb) * dst3 : (kEp2 * kRb) * dst3;
  const double dpp = along_z ? -((kE2 * kRa) * dct3) : dp - (kE2 * kRa) * dct3;
  // latitude: tan(lat' - lat) = (dzz pp - zz dpp)/(zz zz' + pp pp'), with (zz, pp)/hypot = (sin, cos) lat
  const double e = (pc.cl * dpp + pc.sl * dzz) * pc.ihy;
  const double dlat = ((dzz * pc.cl - dpp * pc.sl) * pc.ihy) * (1.0 + e * (e - 1.0));
  const double hl2 = 0.5 * dlat * dlat;
  const double dsl = pc.cl * dlat - pc.sl * hl2, dcl = -(pc.sl * dlat + pc.cl * hl2);
  // half-angle pair of the NED quaternion
  const double hd = 0.5 * dlat, hd2 = 0.5 * hd * hd;
  o.shp = c.shp + (c.chp * hd - c.shp * hd2);
  o.chp = c.chp - (c.shp * hd + c.chp * hd2);
  // altitude (src/Earth.cpp:58-59)
  const double v = dcl * pc.icl;
  const double dq = ((dp - (pc.p * pc.icl) * dcl) * pc.icl) * (1.0 + v * (v - 1.0));
  const double nr = pc.N * (1.0 / kRa);
  const double uw = (-kE2 * dsl * __builtin_fma(2.0, pc.sl, dsl)) * (nr * nr);   // 1/(1 - e^2 sin^2 lat) = (N/Ra)^2
  const double dN = pc.N * (uw * (-0.5 + 0.375 * uw));
  const double dalt = dq - dN;
  // geopotential altitude (src/Air.cpp:47-54)
  const double g1 = (pc.air.k <= 6) ? pc.G * (1.0 / 6356766.0) : 0.0;           // 1/(r0 + alt) below 86 km (h < 84852 m: layers 0..6)
  const double dh = (pc.G * pc.G) * dalt * (1.0 - dalt * g1);
  Air ca_;
  ca_.P = c.P; ca_.rho = c.rho; ca_.inv_a = c.inv_a; ca_.T = 0.0; ca_.a = 0.0;
  atmosphere_delta(ca_, pc.air, dh, tb.atm, o.P, o.rho, o.inv_a);
  o.wn = __builtin_fma(pc.s0, dh, c.wn);
  o.we = __builtin_fma(pc.s1, dh, c.we);
  return (c.inv_p > 0.0) && (fabs(u) < 1.0e-4) && (fabs(v) < 1.0e-4) && (fmax(fabs(dalt), fabs(dh)) < pc.margin);
}

// depends on time only: src/Coordinate.cpp:41-59 (cos/sin(omega t)), :75-79 (half angle)
struct EarthAngle { double c, s, ch, sh; };

// One sincos of the half angle; the full angle by the double-angle identities (<= 2 ulp from a second
// sincos, and the same pair serves the centre and every sweep that does not move t).
// The half-angle pair alone (what the fused kernel keeps across its sweeps) and the full set formed from it where it is
// used: the same two products as in earth_angle(), as products the compiler cannot merge and carry (fresh_mul).
struct EarthHalf { double ch, sh; };
GEL_DEV EarthAngle full_angle(const EarthHalf& h) {
  EarthAngle e;
  e.ch = h.ch; e.sh = h.sh;
  e.s = fresh_mul(2.0 * h.sh, h.ch);
  e.c = fresh_mul(h.ch - h.sh, h.ch + h.sh);
  return e;
}
GEL_DEV EarthAngle earth_angle(double t) {
  EarthAngle e;
  fsincos(kOmega * t / 2.0, &e.sh, &e.ch);
  e.s = 2.0 * e.sh * e.ch;
  e.c = (e.ch - e.sh) * (e.ch + e.sh);
  return e;
}

// Wind vector in ECI = quatrot(quat_nedg2eci(pos, t), wind_ned)  (src/pybind_dynamics.cpp:51-52).
// quat_nedg2eci = conj( q_eci2ecef(t) * q_ecef2ned( Rz(-omega t) pos ) )  (src/Coordinate.cpp:75-110).
// The reference re-runs the Bowring latitude on the rotated position (Coordinate.cpp:86); a rotation
// about z leaves (sqrt(x^2+y^2), z) and therefore the latitude unchanged, so pos_part's latitude is
// reused (equal to a recomputation up to rounding).  The longitude IS taken from the rotated position;
// the reference forms lon = atan2(py, px) and uses only cos/sin(lon/2) (Coordinate.cpp:87-88): they are
// obtained from (cos lon, sin lon) = (px, py)/p by the half-angle identities, on the branch that has no
// cancellation (lon/2 in (-pi/2, pi/2], so cos(lon/2) >= 0).
GEL_DEV void wind_eci(const double r[3], const EarthAngle& e, double s_hp, double c_hp, double inv_p, double wn,
                      double we, double w[3]) {
  // eci2ecef(pos, t): src/Coordinate.cpp:51-59
  const double px = r[0] * e.c + r[1] * e.s;
  const double py = -r[0] * e.s + r[1] * e.c;
  const double clon = (inv_p > 0.0) ? px * inv_p : 1.0, slon = py * inv_p;  // on the polar axis: longitude 0
  double th, ith;  // |cos| or |sin| of lon/2, whichever is >= 0.707, and its reciprocal from the same iteration
  fsqrt_rsqrt(0.5 * (1.0 + fabs(clon)), th, ith);
  const double uh = (0.5 * slon) * ith;
  const double c_hl = (clon >= 0.0) ? th : fabs(uh);
  const double s_hl = (clon >= 0.0) ? uh : copysign(th, slon);
  // quat_ecef2ned: src/Coordinate.cpp:85-98
  const double irt2 = 0.70710678118654752440;  // 1/sqrt(2)
  const double b0 = c_hl * (c_hp - s_hp) * irt2, b1 = s_hl * (c_hp + s_hp) * irt2;
  const double b2 = -c_hl * (c_hp + s_hp) * irt2, b3 = s_hl * (c_hp - s_hp) * irt2;
  // (ch,0,0,sh) * b, then conjugate  -> q = quat_ned2eci
  const double q0 = e.ch * b0 - e.sh * b3;
  const double q1 = -(e.ch * b1 - e.sh * b2);
  const double q2 = -(e.ch * b2 + e.sh * b1);
  const double q3 = -(e.ch * b3 + e.sh * b0);
  // quatrot(q, (wn, we, 0)) = vec( conj(q) * ((0,wn,we,0) * q) )   (src/wrapper_coordinate.hpp:70-78)
  const double t0 = -wn * q1 - we * q2;
  const double t1 = wn * q0 + we * q3;
  const double t2 = -wn * q3 + we * q0;
  const double t3 = wn * q2 - we * q1;
  w[0] = q0 * t1 - q1 * t0 - q2 * t3 + q3 * t2;
  w[1] = q0 * t2 + q1 * t3 - q2 * t0 - q3 * t1;
  w[2] = q0 * t3 - q1 * t2 + q2 * t1 - q3 * t0;
}

// wind_eci unless the whole wavefront is in calm air: where both wind components are exactly zero (below / above the
// measured part of a wind table, as in the shipped example from 23 km up) the rotation of the zero vector is the zero
// vector, so its ~110 instructions are skipped -- six times per node in the fused kernel.  Non-finite components are not
// zero and take the full path.
GEL_DEV void wind_eci_or_calm(const double r[3], const EarthAngle& e, double s_hp, double c_hp, double inv_p, double wn,
                              double we, double w[3]) {
#if GEL_CALM_SHORTCUT
  if (__builtin_amdgcn_ballot_w64(!(wn == 0.0 && we == 0.0)) == 0) {   // wave-uniform branch
    w[0] = 0.0; w[1] = 0.0; w[2] = 0.0;
    return;
  }
#endif
  wind_eci(r, e, s_hp, c_hp, inv_p, wn, we, w);
}

// aerodynamic force (ECI): src/pybind_dynamics.cpp:48-59 given the shared parts.
// Air-relative velocity: the reference forms vel_eci2ecef(v, r, t) = Rz(-omega t) (v - omega x r) and then rotates it straight
// back, ecef2eci(., t) = Rz(+omega t) (src/Coordinate.cpp:69-73, 41-49, src/pybind_dynamics.cpp:48,53).  The two rotations
// cancel; they are not performed here (GEL_AERO_ROTATE=1 restores them): v - omega x r - w_eci differs from the round trip by
// its rounding (<= 4 ulp of |v|, i.e. 1e-15 of the force), identically in the centre and in every perturbed evaluation.
#ifndef GEL_AERO_ROTATE
#define GEL_AERO_ROTATE 0
#endif
GEL_DEV void aero_force(const double r[3], const double v[3], double rho, double inv_a_sound, const EarthAngle& e,
                        const double w[3], double area, const Tables& tb, double F[3], Bracket* br = nullptr) {
  // omega x r = (-w y, w x, 0)
  const double d0 = v[0] + kOmega * r[1];
  const double d1 = v[1] - kOmega * r[0];
#if GEL_AERO_ROTATE
  const double e0 = d0 * e.c + d1 * e.s;
  const double e1 = -d0 * e.s + d1 * e.c;
  const double a0 = (e0 * e.c - e1 * e.s) - w[0];
  const double a1 = (e0 * e.s + e1 * e.c) - w[1];
#else
  const double a0 = d0 - w[0];
  const double a1 = d1 - w[1];
#endif
  const double a2 = v[2] - w[2];
  // a vehicle at rest in the air (vn = 0) is a legitimate input: clamp below anything physical so that
  // fsqrt stays defined; the force is k * (-a) = 0 either way
  const double vn = fsqrt(fmax(a0 * a0 + a1 * a1 + a2 * a2, 1.0e-200));
  const double mach = vn * inv_a_sound;
  // br: the caller's Mach interval of this node's previous evaluations (fused kernel), or none
  const double ca = br ? interp_tab_cached(mach, tb.ca, tb.cas, tb.Kc, 2, 1, *br) : interp_tab(mach, tb.ca, tb.cas, tb.Kc, 2, 1);
  const double k = 0.5 * rho * area * ca * vn;
  F[0] = k * -a0; F[1] = k * -a1; F[2] = k * -a2;
}

// acc/unit_vel = ((thrust_eci + aero)/m + g)/uv        src/pybind_dynamics.cpp:66-70
GEL_DEV void accel(const double Td[3], const double F[3], double inv_m, const double g[3], double inv_uv,
                   double out[3]) {
#pragma unroll
  for (int c = 0; c < 3; c++) out[c] = ((Td[c] + F[c]) * inv_m + g[c]) * inv_uv;
}
// the same, also returning (thrust_eci + aero)/m (what the mass sweep's closed form scales); same operations, same bits
GEL_DEV void accel_parts(const double Td[3], const double F[3], double inv_m, const double g[3], double inv_uv, double tm[3],
                         double out[3]) {
#pragma unroll
  for (int c = 0; c < 3; c++) { tm[c] = (Td[c] + F[c]) * inv_m; out[c] = (tm[c] + g[c]) * inv_uv; }
}
// NoAir: (thrust_eci/m + g)/uv                          src/pybind_dynamics.cpp:85-91
GEL_DEV void accel_noair(const double Td[3], double inv_m, const double g[3], double inv_uv, double out[3]) {
#pragma unroll
  for (int c = 0; c < 3; c++) out[c] = (Td[c] * inv_m + g[c]) * inv_uv;
}

// thrust direction = quatrot(conj(q), (1,0,0))   (src/pybind_dynamics.cpp:62-63), zero terms dropped
GEL_DEV void thrust_dir(const double q[4], double dir[3]) {
  dir[0] = q[0] * q[0] + q[1] * q[1] - q[2] * q[2] - q[3] * q[3];
  dir[1] = q[0] * q[3] + q[1] * q[2] + q[2] * q[1] + q[3] * q[0];
  dir[2] = -q[0] * q[2] + q[1] * q[3] - q[2] * q[0] + q[3] * q[1];
}

// quaternion kinematics: src/pybind_dynamics.cpp:94-106; om = u*unit_u*pi/180, zero terms dropped:
// dq = 0.5 * q * (0, 0, oy, oz)
GEL_DEV void quat_rate(const double q[4], double u0, double u1, double unit_u, double dq[4]) {
  const double d2r = 0.017453292519943295769;  // pi/180
  const double oy = (u0 * unit_u) * d2r, oz = (u1 * unit_u) * d2r;
  dq[0] = 0.5 * (-q[2] * oy - q[3] * oz);
  dq[1] = 0.5 * (q[2] * oz - q[3] * oy);
  dq[2] = 0.5 * (q[0] * oy - q[1] * oz);
  dq[3] = 0.5 * (q[0] * oz + q[1] * oy);
}

// ---------------------------------------------------------------------------
// Aero path constraints (lib/con_aero.py, src/wrapper_utils.hpp:89-206): the pieces aero_kernel (gel_kernels.hip) and the
// aero rows of the fused kernel (gel_eval_kernel.h, AERO instantiation) share -- the same expressions, hence the same bits.
// ---------------------------------------------------------------------------
// air-relative velocity in ECI (wrapper_utils.hpp:93-100) and its SQUARED norm (q needs no root; alpha takes the reciprocal root)
GEL_DEV double aero_vair2(const double r[3], const double v[3], const double w[3], double a[3]) {
  a[0] = (v[0] + kOmega * r[1]) - w[0]; a[1] = (v[1] - kOmega * r[0]) - w[1]; a[2] = v[2] - w[2];
  return a[0] * a[0] + a[1] * a[1] + a[2] * a[2];
}
// cosine of the angle of attack (wrapper_utils.hpp:101-106): one dot product times the two reciprocal norms (the reference divides
// component by component: <= 3 ulp of the cosine apart); d = thrust_dir(q), ind = 1/|d|
GEL_DEV double aero_cos(const double a[3], double nv2, const double d[3], double ind) {
  return ((a[0] * d[0] + a[1] * d[1] + a[2] * d[2]) * frsqrt(fmax(nv2, 1.0e-300))) * ind;
}
// the reference's clamps (wrapper_utils.hpp:107-111): cos > 1 -> 0, |v_air| < 1e-6 -> 0
GEL_DEV double aero_acos(double c, double nv2) { return (c > 1.0) ? 0.0 : ((nv2 < 1.0e-12) ? 0.0 : acos(c)); }
// Exact-difference form of alpha: with c = cos(alpha_c), s = sin(alpha_c) and the perturbed cosine c_p,
//   c_p - c = c (cos t - 1) - s sin t   =>   t = -(c_p - c)/s - (c / 2s) t^2 + t^3/6 - ...        (t = alpha_p - alpha_c)
// solved by two substitutions.  With x = (c / 2s) t0 the fixed point is t0 (1 - x + 2x^2 - 5x^3 + ...) and two substitutions
// give t0 (1 - x + 2x^2 - x^3): 4 |x|^3 of t short, i.e. <= 4e-12 of t under the guard |x| < 1e-4 below (t ~ 1e-8 .. 1e-5, so
// the guard refuses only angles of attack below a few degrees at the largest steps) -- four orders below what a second acos
// would carry.  One reciprocal root and a dozen operations instead of that acos.  false: a lane where the form does not
// apply (a clamp of the reference is active at either point, sin(alpha) < 1e-6, or the step is not small against sin(alpha)).
GEL_DEV bool aero_dalpha(double c_p, double nv2_p, double c_c, double inv_s, bool centre_ok, double& t) {
  const double t0 = (c_c - c_p) * inv_s, k = (0.5 * c_c) * inv_s;
  const double t1 = t0 - (k * t0) * t0;
  t = (t0 - (k * t1) * t1) + (t1 * t1) * (t1 * (1.0 / 6.0));
  return centre_ok && (c_p <= 1.0) && (nv2_p >= 1.0e-12) && (fabs(k * t0) < 1.0e-4);
}


}  // namespace gel
