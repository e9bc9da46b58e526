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

GEL_DEV PosPart pos_part(const double r[3], const Tables& tb, double barC20, Bracket2* wbr = nullptr) {
  PosPart o;
  // the reference feeds the ECI position to ecef2geodetic for altitude (src/pybind_dynamics.cpp:43)
  // The altitude p/cos(lat) - N cancels 6.4e6 m down to the altitude and the position sweeps difference
  // exactly that round-off, so sin/cos(lat) are taken the reference's way (atan2, then sincos): measured,
  // the algebraic pair of geodetic_sincos() is equally accurate but decorrelates the FD noise from the
  // reference's (3e-4 on vel/position entries of a polar, high-dynamic-pressure test state).
  double p, sl, cl;
#ifdef GEL_AB_LATALG  // A/B switch for tools/variant.sh only: the algebraic pair (see the comment above)
  geodetic_sincos(r[0], r[1], r[2], sl, cl, p);
  o.inv_p = frcp(p);
#else
  double lat;
  geodetic_lat_p(r[0], r[1], r[2], lat, p, o.inv_p);
  fsincos(lat, &sl, &cl);
#endif
  // half-angle pair of the NED quaternion (src/Coordinate.cpp:89-90): cos(lat/2) = sqrt((1+cos lat)/2)
  // (cos lat >= 0), sin(lat/2) = sin lat / (2 cos(lat/2)); root and reciprocal root from one iteration
  {
    double irt;
    fsqrt_rsqrt(0.5 * (1.0 + cl), o.chp, irt);
    o.shp = (0.5 * sl) * irt;
  }
  const double alt = geodetic_alt_from(p, sl, cl);
  const double h = geopotential_altitude(alt);
  const Air air = atmosphere(h, tb.atm);
  o.rho = air.rho; o.P = air.P; o.inv_a = air.inv_a;
  // wind looked up at geopotential altitude (:44,49); wbr: the caller's altitude interval of this node's previous evaluations
  if (wbr) wind_ned2_cached(h, tb.wind, tb.winds, tb.Kw, o.wn, o.we, *wbr);
  else wind_ned2(h, tb.wind, tb.winds, tb.Kw, o.wn, o.we);
  gravity_eci(r, barC20, o.g);
  return o;
}

// depends on time only: src/Coordinate.cpp:41-59 (cos/sin(omega t)), :75-79 (half angle)
struct EarthAngle { double c, s, ch, sh; };

// One sincos of the half angle; the full angle by the double-angle identities (<= 2 ulp from a second
// sincos, and the same pair serves the centre and every sweep that does not move t).
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

// aerodynamic force (ECI): src/pybind_dynamics.cpp:48-59 given the shared parts
GEL_DEV void aero_force(const double r[3], const double v[3], double rho, double inv_a_sound, const EarthAngle& e,
                        const double w[3], double area, const Tables& tb, double F[3], Bracket* br = nullptr) {
  // vel_eci2ecef: src/Coordinate.cpp:69-73 (omega x r = (-w y, w x, 0)), then ecef2eci (:41-49), minus wind
  const double d0 = v[0] + kOmega * r[1];
  const double d1 = v[1] - kOmega * r[0];
  const double e0 = d0 * e.c + d1 * e.s;
  const double e1 = -d0 * e.s + d1 * e.c;
  const double a0 = (e0 * e.c - e1 * e.s) - w[0];
  const double a1 = (e0 * e.s + e1 * e.c) - w[1];
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

}  // namespace gel
