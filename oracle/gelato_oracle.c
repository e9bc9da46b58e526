/*
 * gelato_oracle.c -- CPU ORACLE (test infrastructure, NOT the product).
 * See gelato_oracle.h for scope, provenance and how parity is pinned.
 *
 * Every function cites the reference file:line it restates.  The structure
 * deliberately mirrors the reference (one full RHS sweep per perturbed
 * variable, per phase; in-place "+= dx ... -= dx" on a private copy of x;
 * COO emission in the reference's order) so that it is both the checker and a
 * like-for-like single-thread CPU baseline.
 */
#include "gelato_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* Per-thread scratch for the dense sub-matrices of the velocity / quaternion Jacobians (300-530 KB at n = 64):
 * allocated once per thread and kept.  Fresh malloc()s of that size are mmap()ed and page-faulted on every
 * call, which serialises the OpenMP threads of orc_eval_batch on the kernel's mm lock. */
static double* tls_buf(int slot, size_t n) {
  static __thread double* buf[2];
  static __thread size_t cap[2];
  if (cap[slot] < n) {
    free(buf[slot]);
    buf[slot] = (double*)malloc(n * sizeof(double));
    cap[slot] = buf[slot] ? n : 0;
  }
  return buf[slot];
}

/* ------------------------------------------------------------------ */
/* constants: src/Earth.cpp:41-47                                      */
/* ------------------------------------------------------------------ */
static const double MU = 3.986004418e14;
static const double OMEGA = 7.2921151467e-5;
static const double RA = 6378137.0;
#define ONE_F 298.257223563

static double earth_Rb(void) { return RA * (1.0 - 1.0 / ONE_F); }

/* ------------------------------------------------------------------ */
/* US Standard Atmosphere 1976: src/Air.cpp:28-111                      */
/* ------------------------------------------------------------------ */
static const double AIR_RSTAR = 8314.32, AIR_G0 = 9.80665, AIR_R0 = 6356766.0;
static const double AIR_HB[11] = {0.0, 11000.0, 20000.0, 32000.0, 47000.0, 51000.0,
                                  71000.0, 86000.0, 91000.0, 110000.0, 120000.0};
static const double AIR_LMB[11] = {-0.0065, 0.0, 0.001, 0.0028, 0.0, -0.0028, -0.002, 0.0, 0.0025, 0.012, 0.012};
static const double AIR_TMB[11] = {288.15, 216.65, 216.65, 228.65, 270.65, 270.65,
                                   214.65, 186.8673, 186.8673, 240.0, 360.0};
static const double AIR_PB[11] = {101325.0, 22632.0, 5474.9, 868.02, 110.91, 66.939,
                                  3.9564, 0.37338, 0.15381, 7.1042e-3, 2.5382e-3};
static const double AIR_MB[11] = {28.9644, 28.9644, 28.9644, 28.9644, 28.9644, 28.9644,
                                  28.9644, 28.9522, 28.89, 27.27, 26.20};

typedef struct { double Hb, Lmb, Tmb, Pb, R; } air_params;

double orc_geopotential_altitude(double z) { /* Air.cpp:47-54 */
  if (z < 86000.0) return 1.0 * (AIR_R0 * z) / (AIR_R0 + z);
  return z;
}

static air_params us76_params(double altitude) { /* Air.cpp:56-69 */
  int k = 0;
  for (int i = 0; i < 11; i++)
    if (altitude >= AIR_HB[i]) k = i;
  air_params p;
  p.Hb = AIR_HB[k]; p.Lmb = AIR_LMB[k]; p.Tmb = AIR_TMB[k]; p.Pb = AIR_PB[k];
  p.R = AIR_RSTAR / AIR_MB[k];
  return p;
}

double orc_air_temperature(double h) { /* Air.cpp:71-88 */
  air_params p = us76_params(h);
  if (h <= 91000.0) {
    return p.Tmb + p.Lmb * (h - p.Hb);
  } else if (h <= 110000.0) {
    double Tc = 263.1905, A = -76.3232, a = -19942.9;
    return Tc + A * sqrt(1.0 - (h - 91000.0) * (h - 91000.0) / a / a);
  } else if (h <= 120000.0) {
    return p.Tmb + p.Lmb * (h - p.Hb);
  } else {
    double Tinf = 1000.0;
    double xi = (h - p.Hb) * (AIR_R0 + p.Hb) / (AIR_R0 + h);
    return Tinf - (Tinf - p.Tmb) * exp(-0.01875e-3 * xi);
  }
}

double orc_air_pressure(double h) { /* Air.cpp:90-98 */
  air_params p = us76_params(h);
  if (fabs(p.Lmb) > 1.0e-6)
    return p.Pb * pow((p.Tmb + p.Lmb * (h - p.Hb)) / p.Tmb, -AIR_G0 / p.Lmb / p.R);
  return p.Pb * exp(AIR_G0 / p.R * (p.Hb - h) / p.Tmb);
}

double orc_air_density(double h) { /* Air.cpp:100-105 */
  air_params p = us76_params(h);
  double T = orc_air_temperature(h);
  double P = orc_air_pressure(h);
  return P / p.R / T;
}

double orc_speed_of_sound(double h) { /* Air.cpp:107-111 */
  air_params p = us76_params(h);
  double T = orc_air_temperature(h);
  return sqrt(1.4 * p.R * T);
}

/* ------------------------------------------------------------------ */
/* geodesy / frames                                                    */
/* ------------------------------------------------------------------ */
static void ecef2geodetic_rad(const double pos[3], double out[3]) { /* Earth.cpp:49-61 */
  const double Rb = earth_Rb();
  const double e2 = (RA * RA - Rb * Rb) / RA / RA;
  const double ep2 = (RA * RA - Rb * Rb) / Rb / Rb;
  double p = sqrt(pos[0] * pos[0] + pos[1] * pos[1]);
  double theta = atan2(pos[2] * RA, p * Rb);
  double lat = atan2(pos[2] + ep2 * Rb * (sin(theta) * sin(theta) * sin(theta)),
                     p - e2 * RA * (cos(theta) * cos(theta) * cos(theta)));
  double lon = atan2(pos[1], pos[0]);
  double N = RA / sqrt(1.0 - e2 * sin(lat) * sin(lat));
  double alt = p / cos(lat) - N;
  out[0] = lat; out[1] = lon; out[2] = alt;
}

void orc_ecef2geodetic(double x, double y, double z, double out[3]) { /* wrapper_coordinate.hpp:105-111 */
  double pos[3] = {x, y, z};
  ecef2geodetic_rad(pos, out);
  out[0] = out[0] * 180.0 / M_PI;
  out[1] = out[1] * 180.0 / M_PI;
}

void orc_gravity(const double pos[3], double barC20, double g[3]) { /* gravity.cpp:11-57 */
  double a = 6378137.0, one_f = ONE_F, mu = MU;
  double f = 1.0 / one_f;
  double b = a * (1.0 - f);
  double x = pos[0], y = pos[1], z = pos[2];
  double r = sqrt(x * x + y * y + z * z);
  double irx, iry, irz;
  if (r == 0.0) { irx = iry = irz = 0; }
  else { irx = x / r; iry = y / r; irz = z / r; }
  double barP20 = sqrt(5.0) * (3.0 * irz * irz - 1.0) * 0.5;
  double barP20d = sqrt(5.0) * 3.0 * irz;
  if (r < b) r = b;
  double g_ir = -mu / (r * r) * (1.0 + barC20 * (a / r) * (a / r) * (3.0 * barP20 + irz * barP20d));
  double g_iz = mu / (r * r) * (a / r) * (a / r) * barC20 * barP20d;
  g[0] = g_ir * irx;
  g[1] = g_ir * iry;
  g[2] = g_ir * irz + g_iz;
}

void orc_ecef2eci(const double v[3], double t, double out[3]) { /* Coordinate.cpp:41-49 */
  double o0 = v[0] * cos(OMEGA * t) - v[1] * sin(OMEGA * t);
  double o1 = v[0] * sin(OMEGA * t) + v[1] * cos(OMEGA * t);
  out[2] = v[2]; out[0] = o0; out[1] = o1;
}

void orc_eci2ecef(const double v[3], double t, double out[3]) { /* Coordinate.cpp:51-59 */
  double o0 = v[0] * cos(OMEGA * t) + v[1] * sin(OMEGA * t);
  double o1 = -v[0] * sin(OMEGA * t) + v[1] * cos(OMEGA * t);
  out[2] = v[2]; out[0] = o0; out[1] = o1;
}

void orc_vel_eci2ecef(const double vel[3], const double pos[3], double t, double out[3]) { /* Coordinate.cpp:69-73 */
  /* omega_vec.cross(pos) with omega_vec = (0,0,w): (-w*y, w*x, 0) */
  double w[3] = {0.0, 0.0, OMEGA};
  double cr[3] = {w[1] * pos[2] - w[2] * pos[1], w[2] * pos[0] - w[0] * pos[2], w[0] * pos[1] - w[1] * pos[0]};
  double d[3] = {vel[0] - cr[0], vel[1] - cr[1], vel[2] - cr[2]};
  orc_eci2ecef(d, t, out);
}

void orc_quatmult(const double q[4], const double p[4], double o[4]) { /* wrapper_coordinate.hpp:50-57 */
  double o0 = q[0] * p[0] - q[1] * p[1] - q[2] * p[2] - q[3] * p[3];
  double o1 = q[0] * p[1] + q[1] * p[0] + q[2] * p[3] - q[3] * p[2];
  double o2 = q[0] * p[2] - q[1] * p[3] + q[2] * p[0] + q[3] * p[1];
  double o3 = q[0] * p[3] + q[1] * p[2] - q[2] * p[1] + q[3] * p[0];
  o[0] = o0; o[1] = o1; o[2] = o2; o[3] = o3;
}

static void quat_conj(const double q[4], double o[4]) { /* wrapper_coordinate.hpp:59-66 */
  o[0] = q[0]; o[1] = -q[1]; o[2] = -q[2]; o[3] = -q[3];
}

void orc_quatrot(const double q[4], const double v[3], double out[3]) { /* wrapper_coordinate.hpp:70-78 */
  double vq[4] = {0, v[0], v[1], v[2]};
  double qc[4], t1[4], r[4];
  quat_conj(q, qc);
  orc_quatmult(vq, q, t1);
  orc_quatmult(qc, t1, r);
  out[0] = r[1]; out[1] = r[2]; out[2] = r[3];
}

static void quat_eci2ecef(double t, double q[4]) { /* Coordinate.cpp:75-79 */
  q[0] = cos(OMEGA * t / 2.0); q[1] = 0.0; q[2] = 0.0; q[3] = sin(OMEGA * t / 2.0);
}

static void quat_ecef2ned(const double pos_ecef[3], double q[4]) { /* Coordinate.cpp:85-98 */
  double g[3];
  ecef2geodetic_rad(pos_ecef, g);
  double c_hl = cos(g[1] / 2.0), s_hl = sin(g[1] / 2.0);
  double c_hp = cos(g[0] / 2.0), s_hp = sin(g[0] / 2.0);
  q[0] = c_hl * (c_hp - s_hp) / sqrt(2.0);
  q[1] = s_hl * (c_hp + s_hp) / sqrt(2.0);
  q[2] = -c_hl * (c_hp + s_hp) / sqrt(2.0);
  q[3] = s_hl * (c_hp - s_hp) / sqrt(2.0);
}

void orc_quat_nedg2eci(const double pos[3], double t, double out[4]) { /* Coordinate.cpp:104-110 */
  double a[4], b[4], pe[3], ab[4];
  quat_eci2ecef(t, a);
  orc_eci2ecef(pos, t, pe);
  quat_ecef2ned(pe, b);
  orc_quatmult(a, b, ab); /* Eigen Quaterniond product == Hamilton product */
  quat_conj(ab, out);
}

/* wrapper_utils.hpp:51-80.  At x == xp[0] the reference's lower_bound arithmetic
 * indexes xp[-1]; the well-defined np.interp value yp[0]
 * (the Python twin's behaviour, lib/utils.py:83-89) is returned instead. */
double orc_interp(double x, const double* xp, const double* yp, int n, int stride) {
  if (x <= xp[0]) return yp[0];
  if (x > xp[(n - 1) * stride]) return yp[(n - 1) * stride];
  int lo = 0, hi = n; /* std::lower_bound: first index with xp[idx] >= x */
  while (lo < hi) {
    int mid = (lo + hi) / 2;
    if (xp[mid * stride] < x) lo = mid + 1; else hi = mid;
  }
  int idx = lo - 1;
  double x_lower = xp[idx * stride], x_upper = xp[(idx + 1) * stride];
  double y_lower = yp[idx * stride], y_upper = yp[(idx + 1) * stride];
  double alpha = (x - x_lower) / (x_upper - x_lower);
  return y_lower + alpha * (y_upper - y_lower);
}

void orc_wind_ned(double alt, const double* wind, int K, double out[3]) { /* wrapper_utils.hpp:82-87 */
  out[0] = orc_interp(alt, wind, wind + 1, K, 3);
  out[1] = orc_interp(alt, wind, wind + 2, K, 3);
  out[2] = 0.0;
}

/* ------------------------------------------------------------------ */
/* RHS: src/pybind_dynamics.cpp:30-106                                 */
/* ------------------------------------------------------------------ */
void orc_dynamics_velocity(int n, const double* mass_e, const double* pos_e, const double* vel_e,
                           const double* quat, const double* t, const double param[5],
                           const double* wind, int Kw, const double* ca_tab, int Kc,
                           const double units[3], double barC20, double* out) {
  double thrust_vac = param[0], air_area = param[2], nozzle_area = param[4];
  for (int i = 0; i < n; i++) {
    double mass = mass_e[i] * units[0];
    double pos[3] = {pos_e[3 * i] * units[1], pos_e[3 * i + 1] * units[1], pos_e[3 * i + 2] * units[1]};
    double vel[3] = {vel_e[3 * i] * units[2], vel_e[3 * i + 1] * units[2], vel_e[3 * i + 2] * units[2]};
    const double* q = quat + 4 * i;
    double llh[3];
    orc_ecef2geodetic(pos[0], pos[1], pos[2], llh);
    double altitude = orc_geopotential_altitude(llh[2]);
    double rho = orc_air_density(altitude);
    double p = orc_air_pressure(altitude);

    double vel_ecef[3], wned[3], qn2i[4], wind_eci[3], vair[3];
    orc_vel_eci2ecef(vel, pos, t[i], vel_ecef);
    orc_wind_ned(altitude, wind, Kw, wned);
    orc_quat_nedg2eci(pos, t[i], qn2i);
    orc_quatrot(qn2i, wned, wind_eci);
    orc_ecef2eci(vel_ecef, t[i], vair);
    vair[0] -= wind_eci[0]; vair[1] -= wind_eci[1]; vair[2] -= wind_eci[2];
    double vnorm = sqrt(vair[0] * vair[0] + vair[1] * vair[1] + vair[2] * vair[2]);
    double mach = vnorm / orc_speed_of_sound(altitude);
    double ca = orc_interp(mach, ca_tab, ca_tab + 1, Kc, 2);
    double k = 0.5 * rho * air_area * ca * vnorm;
    double aero[3] = {k * -vair[0], k * -vair[1], k * -vair[2]};

    double thrust = thrust_vac - nozzle_area * p;
    double qc[4], ex[3] = {1.0, 0.0, 0.0}, dir[3], g[3];
    quat_conj(q, qc);
    orc_quatrot(qc, ex, dir);
    orc_gravity(pos, barC20, g);
    for (int c = 0; c < 3; c++) {
      double acc = (thrust * dir[c] + aero[c]) / mass + g[c];
      out[3 * i + c] = acc / units[2];
    }
  }
}

void orc_dynamics_velocity_NoAir(int n, const double* mass_e, const double* pos_e, const double* quat,
                                 const double param[5], const double units[3], double barC20, double* out) {
  double thrust = param[0];
  for (int i = 0; i < n; i++) {
    double mass = mass_e[i] * units[0];
    double pos[3] = {pos_e[3 * i] * units[1], pos_e[3 * i + 1] * units[1], pos_e[3 * i + 2] * units[1]};
    double qc[4], ex[3] = {1.0, 0.0, 0.0}, dir[3], g[3];
    quat_conj(quat + 4 * i, qc);
    orc_quatrot(qc, ex, dir);
    orc_gravity(pos, barC20, g);
    for (int c = 0; c < 3; c++) {
      double acc = (thrust * dir[c]) / mass + g[c];
      out[3 * i + c] = acc / units[2];
    }
  }
}

void orc_dynamics_quaternion(int n, const double* quat, const double* u_e, double unit_u, double* out) {
  for (int i = 0; i < n; i++) {
    double om[4] = {0.0, 0.0, u_e[2 * i] * unit_u, u_e[2 * i + 1] * unit_u};
    for (int c = 0; c < 4; c++) om[c] = om[c] * M_PI / 180.0;
    double qp[4];
    orc_quatmult(quat + 4 * i, om, qp);
    for (int c = 0; c < 4; c++) out[4 * i + c] = 0.5 * qp[c];
  }
}

/* ------------------------------------------------------------------ */
/* LGR nodes and differentiation matrix: lib/PSfunctions.py:64-88,149-208 */
/* ------------------------------------------------------------------ */
/* Jacobi polynomial P_m^{(0,1)} and derivative at x (three-term recurrence). */
static void jacobi01(int m, double x, double* P, double* dP) {
  const double a = 0.0, b = 1.0;
  double p0 = 1.0, p1 = 0.5 * ((a + b + 2.0) * x + a - b);
  if (m == 0) { *P = 1.0; *dP = 0.0; return; }
  for (int k = 2; k <= m; k++) {
    double c = 2.0 * k + a + b;
    double a1 = 2.0 * k * (k + a + b) * (c - 2.0);
    double a2 = (c - 1.0) * (a * a - b * b);
    double a3 = (c - 2.0) * (c - 1.0) * c;
    double a4 = 2.0 * (k + a - 1.0) * (k + b - 1.0) * c;
    double p2 = ((a2 + a3 * x) * p1 - a4 * p0) / a1;
    p0 = p1; p1 = p2;
  }
  /* (2m+a+b)(1-x^2) P' = m[a-b-(2m+a+b)x] P_m + 2(m+a)(m+b) P_{m-1} */
  double c = 2.0 * m + a + b;
  *P = p1;
  *dP = (m * (a - b - c * x) * p1 + 2.0 * (m + a) * (m + b) * p0) / (c * (1.0 - x * x));
}

/* symmetric tridiagonal eigenvalues by implicit QL (Golub-Welsch, as SciPy's
 * special.j_roots used by lib/PSfunctions.py:163). d[m] diag, e[m] offdiag (e[m-1] unused). */
static int tqli_eigs(int m, double* d, double* e) {
  for (int l = 0; l < m; l++) {
    int iter = 0, mm;
    do {
      for (mm = l; mm < m - 1; mm++) {
        double dd = fabs(d[mm]) + fabs(d[mm + 1]);
        if (fabs(e[mm]) <= 2.3e-16 * dd) break;
      }
      if (mm != l) {
        if (iter++ == 200) return -1;
        double g = (d[l + 1] - d[l]) / (2.0 * e[l]);
        double r = hypot(g, 1.0);
        g = d[mm] - d[l] + e[l] / (g + (g >= 0 ? fabs(r) : -fabs(r)));
        double s = 1.0, c = 1.0, p = 0.0;
        int i;
        for (i = mm - 1; i >= l; i--) {
          double f = s * e[i], b = c * e[i];
          e[i + 1] = (r = hypot(f, g));
          if (r == 0.0) { d[i + 1] -= p; e[mm] = 0.0; break; }
          s = f / r; c = g / r;
          g = d[i + 1] - p;
          r = (d[i] - g) * s + 2.0 * c * b;
          d[i + 1] = g + (p = s * r);
          g = c * r - b;
        }
        if (r == 0.0 && i >= l) continue;
        d[l] -= p; e[l] = g; e[mm] = 0.0;
      }
    } while (mm != l);
  }
  return 0;
}

static int cmp_double(const void* a, const void* b) {
  double x = *(const double*)a, y = *(const double*)b;
  return (x > y) - (x < y);
}

int orc_lgr_nodes(int n, double* tau) { /* PSfunctions.py:149-168 (reverse=True) */
  if (n < 2) return -1;
  int m = n - 1; /* roots of P_{n-1}^{(0,1)} */
  double* d = (double*)calloc(m + 1, sizeof(double));
  double* e = (double*)calloc(m + 1, sizeof(double));
  const double a = 0.0, b = 1.0;
  for (int k = 0; k < m; k++) {
    double c = 2.0 * k + a + b;
    d[k] = (k == 0) ? (b - a) / (a + b + 2.0) : (b * b - a * a) / (c * (c + 2.0));
    if (k >= 1) {
      double num = 4.0 * k * (k + a) * (k + b) * (k + a + b);
      double den = c * c * (c + 1.0) * (c - 1.0);
      e[k - 1] = sqrt(num / den);
    }
  }
  int rc = tqli_eigs(m, d, e);
  for (int k = 0; k < m; k++) { /* Newton polish on the polynomial itself */
    double x = d[k];
    for (int it = 0; it < 3; it++) {
      double P, dP;
      jacobi01(m, x, &P, &dP);
      x -= P / dP;
    }
    d[k] = x;
  }
  /* nodes = sort(-hstack((-1, roots))) */
  tau[0] = 1.0;
  for (int k = 0; k < m; k++) tau[k + 1] = -d[k];
  qsort(tau, n, sizeof(double), cmp_double);
  free(d); free(e);
  return rc;
}

static double lagrangeD(const double* tn, int N, int k, double t) { /* PSfunctions.py:64-88 */
  double den = 1.0;
  for (int i = 0; i < N; i++)
    if (i != k) den = den * (tn[k] - tn[i]);
  double num = 0.0;
  for (int j = 0; j < N; j++) {
    double num_j = 1.0;
    if (j != k) {
      for (int i = 0; i < N; i++)
        if (i != k && i != j) num_j = num_j * (t - tn[i]);
      num = num + num_j;
    }
  }
  return num / den;
}

int orc_lgr_diffmat(int n, double* D) { /* PSfunctions.py:182-208 (reverse=True) */
  double* tk = (double*)malloc((n + 1) * sizeof(double));
  tk[0] = -1.0;
  int rc = orc_lgr_nodes(n, tk + 1);
  for (int k = 0; k < n; k++)
    for (int i = 0; i < n + 1; i++) D[k * (n + 1) + i] = lagrangeD(tk, n + 1, i, tk[k + 1]);
  free(tk);
  return rc;
}

/* ------------------------------------------------------------------ */
/* problem                                                             */
/* ------------------------------------------------------------------ */
struct orc_problem {
  int S, N, M;
  int* n;
  int* ua; /* index_start_u, SectionParameters.py:36-38 */
  double *thrust, *massflow, *ref_area, *nozzle_area;
  int *engine_on, *att_hold;
  double um, up, uv, uu, ut, dx, barC20;
  int Kw, Kc;
  double *wind, *ca;
  double **tau, **D;
};

static double* dupd(const double* s, size_t n) {
  double* d = (double*)malloc(n * sizeof(double));
  memcpy(d, s, n * sizeof(double));
  return d;
}

orc_problem* orc_problem_create(int S, const int32_t* num_nodes, const double* thrust, const double* massflow,
                                const double* reference_area, const double* nozzle_area,
                                const int32_t* engine_on, const int32_t* attitude_hold,
                                const double units[5], double dx, double barC20,
                                const double* wind, int Kw, const double* ca, int Kc,
                                const double* D_all, const double* tau_all) {
  orc_problem* p = (orc_problem*)calloc(1, sizeof(orc_problem));
  p->S = S;
  p->n = (int*)malloc(S * sizeof(int));
  p->ua = (int*)malloc(S * sizeof(int));
  p->engine_on = (int*)malloc(S * sizeof(int));
  p->att_hold = (int*)malloc(S * sizeof(int));
  int N = 0;
  for (int i = 0; i < S; i++) {
    p->n[i] = num_nodes[i]; p->ua[i] = N; N += num_nodes[i];
    p->engine_on[i] = engine_on[i]; p->att_hold[i] = attitude_hold[i];
  }
  p->N = N; p->M = N + S;
  p->thrust = dupd(thrust, S); p->massflow = dupd(massflow, S);
  p->ref_area = dupd(reference_area, S); p->nozzle_area = dupd(nozzle_area, S);
  p->um = units[0]; p->up = units[1]; p->uv = units[2]; p->uu = units[3]; p->ut = units[4];
  p->dx = dx; p->barC20 = barC20;
  p->Kw = Kw; p->Kc = Kc;
  p->wind = dupd(wind, (size_t)Kw * 3); p->ca = dupd(ca, (size_t)Kc * 2);
  p->tau = (double**)calloc(S, sizeof(double*));
  p->D = (double**)calloc(S, sizeof(double*));
  size_t offD = 0, offT = 0;
  for (int i = 0; i < S; i++) {
    int n = p->n[i];
    int shared = -1;
    for (int j = 0; j < i; j++) if (p->n[j] == n) { shared = j; break; }
    if (D_all && tau_all) {
      p->D[i] = dupd(D_all + offD, (size_t)n * (n + 1));
      p->tau[i] = dupd(tau_all + offT, n);
    } else if (shared >= 0) {
      p->D[i] = dupd(p->D[shared], (size_t)n * (n + 1));
      p->tau[i] = dupd(p->tau[shared], n);
    } else {
      p->D[i] = (double*)malloc((size_t)n * (n + 1) * sizeof(double));
      p->tau[i] = (double*)malloc(n * sizeof(double));
      orc_lgr_nodes(n, p->tau[i]);
      orc_lgr_diffmat(n, p->D[i]);
    }
    offD += (size_t)n * (n + 1); offT += n;
  }
  return p;
}

void orc_problem_destroy(orc_problem* p) {
  if (!p) return;
  for (int i = 0; i < p->S; i++) { free(p->tau[i]); free(p->D[i]); }
  free(p->tau); free(p->D); free(p->n); free(p->ua); free(p->engine_on); free(p->att_hold);
  free(p->thrust); free(p->massflow); free(p->ref_area); free(p->nozzle_area); free(p->wind); free(p->ca);
  free(p);
}

const double* orc_problem_D(const orc_problem* p, int i) { return p->D[i]; }
const double* orc_problem_tau(const orc_problem* p, int i) { return p->tau[i]; }
int orc_num_vars(const orc_problem* p) { return 11 * p->M + 2 * p->N + p->S + 1; }
int orc_num_rows(const orc_problem* p, int g) { static const int k[4] = {1, 3, 3, 4}; return k[g] * p->N; }
int orc_num_blocks(int g) { static const int k[4] = {2, 3, 5, 3}; return k[g]; }

/* per-phase nnz of (group, block): SURVEY appendix B */
static int64_t phase_nnz(const orc_problem* p, int g, int blk, int i) {
  int64_t n = p->n[i];
  switch (g) {
    case 0: if (blk == 0) return p->engine_on[i] ? n * (n + 1) : 2 * n;
            return p->engine_on[i] ? 2 * n : 0;
    case 1: if (blk == 0) return 3 * n * (n + 1);
            if (blk == 1) return 3 * n;
            return 6 * n;
    case 2: if (blk == 0) return 3 * n;
            if (blk == 1) return 9 * n;
            if (blk == 2) return 9 * n * (n + 1);
            if (blk == 3) return 12 * n;
            return 6 * n;
    default: if (blk == 0) return p->att_hold[i] ? 8 * n : 16 * n * (n + 1);
             return p->att_hold[i] ? 0 : 8 * n;
  }
}

int64_t orc_block_nnz(const orc_problem* p, int g, int blk) {
  int64_t s = 0;
  for (int i = 0; i < p->S; i++) s += phase_nnz(p, g, blk, i);
  return s;
}

int64_t orc_total_nnz(const orc_problem* p) {
  int64_t s = 0;
  for (int g = 0; g < 4; g++)
    for (int b = 0; b < orc_num_blocks(g); b++) s += orc_block_nnz(p, g, b);
  return s;
}

void orc_block_shape(const orc_problem* p, int g, int blk, int64_t sh[2]) {
  /* con_dynamics.py:75-76,168-170,314-318,550-552 */
  int64_t N = p->N, M = p->M, T = p->S + 1;
  static const int rk[4] = {1, 3, 3, 4};
  sh[0] = rk[g] * N;
  switch (g) {
    case 0: sh[1] = blk == 0 ? M : T; break;
    case 1: sh[1] = blk == 2 ? T : 3 * M; break;
    case 2: sh[1] = blk == 0 ? M : blk == 1 ? 3 * M : blk == 2 ? 3 * M : blk == 3 ? 4 * M : T; break;
    default: sh[1] = blk == 0 ? 4 * M : blk == 1 ? 2 * N : T; break;
  }
}

/* views into the packed decision vector */
typedef struct { const double *mass, *pos, *vel, *quat, *u, *t; } xview;
typedef struct { double *mass, *pos, *vel, *quat, *u, *t; } xmut;
static xview view(const orc_problem* p, const double* x) {
  xview v; int M = p->M, N = p->N;
  v.mass = x; v.pos = x + M; v.vel = x + 4 * M; v.quat = x + 7 * M; v.u = x + 11 * M; v.t = x + 11 * M + 2 * N;
  return v;
}
static xmut mview(const orc_problem* p, double* x) {
  xmut v; int M = p->M, N = p->N;
  v.mass = x; v.pos = x + M; v.vel = x + 4 * M; v.quat = x + 7 * M; v.u = x + 11 * M; v.t = x + 11 * M + 2 * N;
  return v;
}

/* SectionParameters.py:77-81 (entries 1..n only; entry 0 = to is never used by the RHS) */
static void time_nodes(const orc_problem* p, int i, double to, double tf, double* t1n) {
  for (int j = 0; j < p->n[i]; j++) t1n[j] = p->tau[i][j] * (tf - to) / 2 + (tf + to) / 2;
}

/* D(i).dot(X) for k interleaved columns: lh[j*k+c] = sum_i D[j][i] * X[i*k+c] */
static void D_dot(const double* D, int n, const double* X, int k, double* lh) {
  for (int j = 0; j < n; j++)
    for (int c = 0; c < k; c++) {
      double s = 0.0;
      for (int i = 0; i < n + 1; i++) s += D[j * (n + 1) + i] * X[i * k + c];
      lh[j * k + c] = s;
    }
}

static void fill_param(const orc_problem* p, int i, double param[5]) { /* con_dynamics.py:249-252 */
  param[0] = p->thrust[i]; param[1] = p->massflow[i]; param[2] = p->ref_area[i];
  param[3] = 0.0; param[4] = p->nozzle_area[i];
}

/* the closure `dynamics` of con_dynamics.py:345-351 */
static void dyn(const orc_problem* p, const double param[5], int n, const double* m, const double* r,
                const double* v, const double* q, const double* t, double* out) {
  double units[3] = {p->um, p->up, p->uv};
  if (param[2] == 0.0) orc_dynamics_velocity_NoAir(n, m, r, q, param, units, p->barC20, out);
  else orc_dynamics_velocity(n, m, r, v, q, t, param, p->wind, p->Kw, p->ca, p->Kc, units, p->barC20, out);
}

/* ------------------------------------------------------------------ */
/* residuals                                                           */
/* ------------------------------------------------------------------ */
static void res_mass(const orc_problem* p, const double* x, double* out) { /* con_dynamics.py:34-63 */
  xview X = view(p, x);
  for (int i = 0; i < p->S; i++) {
    int n = p->n[i], ua = p->ua[i], xa = ua + i;
    const double* m = X.mass + xa;
    double to = X.t[i], tf = X.t[i + 1];
    if (p->engine_on[i]) {
      D_dot(p->D[i], n, m, 1, out + ua);
      double rh = -p->massflow[i] / p->um * (tf - to) * p->ut / 2.0;
      for (int j = 0; j < n; j++) out[ua + j] = out[ua + j] - rh;
    } else {
      for (int j = 0; j < n; j++) out[ua + j] = m[j + 1] - m[0];
    }
  }
}

static void res_pos(const orc_problem* p, const double* x, double* out) { /* con_dynamics.py:116-152 */
  xview X = view(p, x);
  for (int i = 0; i < p->S; i++) {
    int n = p->n[i], ua = p->ua[i], xa = ua + i;
    double to = X.t[i], tf = X.t[i + 1];
    D_dot(p->D[i], n, X.pos + 3 * xa, 3, out + 3 * ua);
    const double* v1 = X.vel + 3 * (xa + 1);
    for (int j = 0; j < 3 * n; j++) {
      double rh = v1[j] * p->uv * (tf - to) * p->ut / 2.0 / p->up;
      out[3 * ua + j] = out[3 * ua + j] - rh;
    }
  }
}

static void res_vel(const orc_problem* p, const double* x, double* out) { /* con_dynamics.py:216-289 */
  xview X = view(p, x);
  int nmax = 0;
  for (int i = 0; i < p->S; i++) if (p->n[i] > nmax) nmax = p->n[i];
  double* tn = (double*)malloc(nmax * sizeof(double));
  double* f = (double*)malloc(3 * nmax * sizeof(double));
  for (int i = 0; i < p->S; i++) {
    int n = p->n[i], ua = p->ua[i], xa = ua + i;
    double to = X.t[i], tf = X.t[i + 1], param[5];
    time_nodes(p, i, to, tf, tn);
    fill_param(p, i, param);
    D_dot(p->D[i], n, X.vel + 3 * xa, 3, out + 3 * ua);
    dyn(p, param, n, X.mass + xa + 1, X.pos + 3 * (xa + 1), X.vel + 3 * (xa + 1), X.quat + 4 * (xa + 1), tn, f);
    for (int j = 0; j < 3 * n; j++) {
      double rh = f[j] * (tf - to) * p->ut / 2.0;
      out[3 * ua + j] = out[3 * ua + j] - rh;
    }
  }
  free(tn); free(f);
}

static void res_quat(const orc_problem* p, const double* x, double* out) { /* con_dynamics.py:499-533 */
  xview X = view(p, x);
  int nmax = 0;
  for (int i = 0; i < p->S; i++) if (p->n[i] > nmax) nmax = p->n[i];
  double* f = (double*)malloc(4 * nmax * sizeof(double));
  for (int i = 0; i < p->S; i++) {
    int n = p->n[i], ua = p->ua[i], xa = ua + i;
    const double* q = X.quat + 4 * xa;
    double to = X.t[i], tf = X.t[i + 1];
    if (p->att_hold[i]) {
      for (int j = 0; j < n; j++)
        for (int c = 0; c < 4; c++) out[4 * (ua + j) + c] = q[4 * (j + 1) + c] - q[c];
    } else {
      D_dot(p->D[i], n, q, 4, out + 4 * ua);
      orc_dynamics_quaternion(n, q + 4, X.u + 2 * ua, p->uu, f);
      for (int j = 0; j < 4 * n; j++) {
        double rh = f[j] * (tf - to) * p->ut / 2.0;
        out[4 * ua + j] = out[4 * ua + j] - rh;
      }
    }
  }
  free(f);
}

void orc_residual(const orc_problem* p, int g, const double* x, double* out) {
  switch (g) {
    case 0: res_mass(p, x, out); break;
    case 1: res_pos(p, x, out); break;
    case 2: res_vel(p, x, out); break;
    default: res_quat(p, x, out); break;
  }
}

/* ------------------------------------------------------------------ */
/* COO Jacobians                                                       */
/* ------------------------------------------------------------------ */
typedef struct { int32_t *r, *c; double* v; int64_t k; } coo;
static inline void put(coo* o, int32_t r, int32_t c, double v) {
  if (o->r) { o->r[o->k] = r; o->c[o->k] = c; }
  o->v[o->k] = v; o->k++;
}
static void coo_init(const orc_problem* p, int g, int32_t* rows, int32_t* cols, double* vals, coo* o) {
  int64_t off = 0;
  for (int b = 0; b < orc_num_blocks(g); b++) {
    o[b].r = rows ? rows + off : NULL; o[b].c = cols ? cols + off : NULL; o[b].v = vals + off; o[b].k = 0;
    off += orc_block_nnz(p, g, b);
  }
}

static void jac_mass(const orc_problem* p, const double* x, int32_t* rows, int32_t* cols, double* vals) {
  /* con_dynamics.py:66-113 */
  (void)x;
  coo o[2]; coo_init(p, 0, rows, cols, vals, o);
  coo *Jm = &o[0], *Jt = &o[1];
  for (int i = 0; i < p->S; i++) {
    int n = p->n[i], ua = p->ua[i], ub = ua + n, xa = ua + i, xb = xa + n + 1;
    if (p->engine_on[i]) {
      for (int j = ua; j < ub; j++)
        for (int c = xa; c < xb; c++) put(Jm, j, c, p->D[i][(j - ua) * (n + 1) + (c - xa)]);
      for (int j = ua; j < ub; j++) put(Jt, j, i, -p->massflow[i] / p->um * p->ut / 2.0);
      for (int j = ua; j < ub; j++) put(Jt, j, i + 1, p->massflow[i] / p->um * p->ut / 2.0);
    } else {
      for (int j = ua; j < ub; j++) put(Jm, j, xa, -1.0);
      for (int j = ua; j < ub; j++) put(Jm, j, xa + 1 + (j - ua), 1.0);
    }
  }
}

static void jac_pos(const orc_problem* p, const double* x, int32_t* rows, int32_t* cols, double* vals) {
  /* con_dynamics.py:155-213; key order position, velocity, t */
  xview X = view(p, x);
  coo o[3]; coo_init(p, 1, rows, cols, vals, o);
  coo *Jp = &o[0], *Jv = &o[1], *Jt = &o[2];
  for (int i = 0; i < p->S; i++) {
    int n = p->n[i], ua = p->ua[i], ub = ua + n, xa = ua + i, xb = xa + n + 1;
    double to = X.t[i], tf = X.t[i + 1];
    const double* v1 = X.vel + 3 * (xa + 1);
    double rh_vel = -p->uv * (tf - to) * p->ut / 2.0 / p->up;
    for (int j = 0; j < 3 * n; j++) put(Jv, 3 * ua + j, 3 * (xa + 1) + j, rh_vel);
    for (int j = 0; j < 3 * n; j++) put(Jt, 3 * ua + j, i, v1[j] * p->uv * p->ut / 2.0 / p->up);
    for (int j = 0; j < 3 * n; j++) put(Jt, 3 * ua + j, i + 1, -(v1[j] * p->uv * p->ut / 2.0 / p->up));
    for (int ki = 0; ki < 3; ki++)
      for (int j = ua * 3 + ki, jj = 0; j < ub * 3 + ki; j += 3, jj++)
        for (int c = xa * 3 + ki, cc = 0; c < xb * 3 + ki; c += 3, cc++) put(Jp, j, c, p->D[i][jj * (n + 1) + cc]);
  }
}

static void jac_vel(const orc_problem* p, const double* x_in, int32_t* rows, int32_t* cols, double* vals) {
  /* con_dynamics.py:292-496; key order mass, position, velocity, quaternion, t.
   * The reference perturbs views of xdict in place (+= dx ... -= dx); the same
   * operations are applied here to a private copy so later sweeps see the same
   * (possibly 1-ulp drifted) base values the reference's do. */
  int nv = orc_num_vars(p);
  double* x = (double*)malloc(nv * sizeof(double));
  memcpy(x, x_in, nv * sizeof(double));
  xmut X = mview(p, x);
  coo o[5]; coo_init(p, 2, rows, cols, vals, o);
  coo *Jm = &o[0], *Jp = &o[1], *Jv = &o[2], *Jq = &o[3], *Jt = &o[4];
  int nmax = 0;
  for (int i = 0; i < p->S; i++) if (p->n[i] > nmax) nmax = p->n[i];
  double* tn = (double*)malloc(nmax * sizeof(double));
  double* tn2 = (double*)malloc(nmax * sizeof(double));
  double* fc = (double*)malloc(3 * nmax * sizeof(double));
  double* fp = (double*)malloc(3 * nmax * sizeof(double));
  double* rh = (double*)malloc(3 * nmax * sizeof(double));
  double* sub = tls_buf(0, (size_t)9 * nmax * (nmax + 1));
  const double dx = p->dx;
  for (int i = 0; i < p->S; i++) {
    int n = p->n[i], ua = p->ua[i], ub = ua + n, xa = ua + i, xb = xa + n + 1;
    double* m1 = X.mass + xa + 1;
    double* r1 = X.pos + 3 * (xa + 1);
    double* v1 = X.vel + 3 * (xa + 1);
    double* q1 = X.quat + 4 * (xa + 1);
    double to = X.t[i], tf = X.t[i + 1], param[5];
    time_nodes(p, i, to, tf, tn);
    fill_param(p, i, param);
    int W = 3 * (n + 1);
    memset(sub, 0, (size_t)3 * n * W * sizeof(double));
    for (int k = 0; k < 3; k++)
      for (int j = 0; j < n; j++)
        for (int c = 0; c < n + 1; c++) sub[(3 * j + k) * W + 3 * c + k] = p->D[i][j * (n + 1) + c];

    dyn(p, param, n, m1, r1, v1, q1, tn, fc);

    /* mass */
    for (int j = 0; j < n; j++) m1[j] += dx;
    dyn(p, param, n, m1, r1, v1, q1, tn, fp);
    for (int j = 0; j < n; j++) m1[j] -= dx;
    for (int j = 0; j < 3 * n; j++) rh[j] = -(fp[j] - fc[j]) / dx * (tf - to) * p->ut / 2.0;
    for (int j = ua; j < ub; j++)
      for (int c = 0; c < 3; c++) put(Jm, 3 * j + c, xa + 1 + (j - ua), rh[3 * (j - ua) + c]);

    /* position */
    for (int k = 0; k < 3; k++) {
      for (int j = 0; j < n; j++) r1[3 * j + k] += dx;
      dyn(p, param, n, m1, r1, v1, q1, tn, fp);
      for (int j = 0; j < n; j++) r1[3 * j + k] -= dx;
      for (int j = 0; j < 3 * n; j++) rh[j] = -(fp[j] - fc[j]) / dx * (tf - to) * p->ut / 2.0;
      for (int j = ua; j < ub; j++)
        for (int c = 0; c < 3; c++) put(Jp, 3 * j + c, 3 * (xa + 1 + (j - ua)) + k, rh[3 * (j - ua) + c]);
    }

    /* velocity */
    if (param[2] > 0.0) {
      for (int k = 0; k < 3; k++) {
        for (int j = 0; j < n; j++) v1[3 * j + k] += dx;
        dyn(p, param, n, m1, r1, v1, q1, tn, fp);
        for (int j = 0; j < n; j++) v1[3 * j + k] -= dx;
        for (int j = 0; j < 3 * n; j++) rh[j] = -(fp[j] - fc[j]) / dx * (tf - to) * p->ut / 2.0;
        for (int j = 0; j < n; j++)
          for (int c = 0; c < 3; c++) sub[(3 * j + c) * W + 3 * (j + 1) + k] += rh[3 * j + c];
      }
    }
    for (int ki = 0; ki < 3; ki++)
      for (int kj = 0; kj < 3; kj++)
        for (int j = ua * 3 + ki, jj = 0; j < ub * 3 + ki; j += 3, jj++)
          for (int c = xa * 3 + kj, cc = 0; c < xb * 3 + kj; c += 3, cc++)
            put(Jv, j, c, sub[(3 * jj + ki) * W + 3 * cc + kj]);

    /* quaternion */
    for (int k = 0; k < 4; k++) {
      for (int j = 0; j < n; j++) q1[4 * j + k] += dx;
      dyn(p, param, n, m1, r1, v1, q1, tn, fp);
      for (int j = 0; j < n; j++) q1[4 * j + k] -= dx;
      for (int j = 0; j < 3 * n; j++) rh[j] = -(fp[j] - fc[j]) / dx * (tf - to) * p->ut / 2.0;
      for (int j = ua; j < ub; j++)
        for (int c = 0; c < 3; c++) put(Jq, 3 * j + c, 4 * (xa + 1 + (j - ua)) + k, rh[3 * (j - ua) + c]);
    }

    /* t_o, t_f */
    double to_p = to + dx;
    if (param[2] > 0.0) {
      time_nodes(p, i, to_p, tf, tn2);
      dyn(p, param, n, m1, r1, v1, q1, tn2, fp);
      for (int j = 0; j < 3 * n; j++) rh[j] = -(fp[j] * (tf - to_p) - fc[j] * (tf - to)) / dx * p->ut / 2.0;
      for (int j = 0; j < 3 * n; j++) put(Jt, 3 * ua + j, i, rh[j]);
      double tf_p = tf + dx;
      time_nodes(p, i, to, tf_p, tn2);
      dyn(p, param, n, m1, r1, v1, q1, tn2, fp);
      for (int j = 0; j < 3 * n; j++) rh[j] = -(fp[j] * (tf_p - to) - fc[j] * (tf - to)) / dx * p->ut / 2.0;
      for (int j = 0; j < 3 * n; j++) put(Jt, 3 * ua + j, i + 1, rh[j]);
    } else {
      for (int j = 0; j < 3 * n; j++) put(Jt, 3 * ua + j, i, fc[j] * p->ut / 2.0);
      for (int j = 0; j < 3 * n; j++) put(Jt, 3 * ua + j, i + 1, -(fc[j] * p->ut / 2.0));
    }
    (void)xb;
  }
  free(x); free(tn); free(tn2); free(fc); free(fp); free(rh);
}

static void jac_quat(const orc_problem* p, const double* x_in, int32_t* rows, int32_t* cols, double* vals) {
  /* con_dynamics.py:536-632; key order quaternion, u, t */
  int nv = orc_num_vars(p);
  double* x = (double*)malloc(nv * sizeof(double));
  memcpy(x, x_in, nv * sizeof(double));
  xmut X = mview(p, x);
  coo o[3]; coo_init(p, 3, rows, cols, vals, o);
  coo *Jq = &o[0], *Ju = &o[1], *Jt = &o[2];
  int nmax = 0;
  for (int i = 0; i < p->S; i++) if (p->n[i] > nmax) nmax = p->n[i];
  double* fc = (double*)malloc(4 * nmax * sizeof(double));
  double* fp = (double*)malloc(4 * nmax * sizeof(double));
  double* rh = (double*)malloc(4 * nmax * sizeof(double));
  double* sub = tls_buf(1, (size_t)16 * nmax * (nmax + 1));
  const double dx = p->dx;
  for (int i = 0; i < p->S; i++) {
    int n = p->n[i], ua = p->ua[i], ub = ua + n, xa = ua + i, xb = xa + n + 1;
    double* q1 = X.quat + 4 * (xa + 1);
    double* u = X.u + 2 * ua;
    double to = X.t[i], tf = X.t[i + 1];
    if (p->att_hold[i]) {
      for (int j = 4 * ua; j < 4 * ub; j++) put(Jq, j, 4 * xa + (j - 4 * ua) % 4, -1.0);
      for (int j = 4 * ua; j < 4 * ub; j++) put(Jq, j, 4 * (xa + 1) + (j - 4 * ua), 1.0);
      continue;
    }
    int W = 4 * (n + 1);
    memset(sub, 0, (size_t)4 * n * W * sizeof(double));
    for (int k = 0; k < 4; k++)
      for (int j = 0; j < n; j++)
        for (int c = 0; c < n + 1; c++) sub[(4 * j + k) * W + 4 * c + k] = p->D[i][j * (n + 1) + c];
    orc_dynamics_quaternion(n, q1, u, p->uu, fc);
    for (int k = 0; k < 4; k++) {
      for (int j = 0; j < n; j++) q1[4 * j + k] += dx;
      orc_dynamics_quaternion(n, q1, u, p->uu, fp);
      for (int j = 0; j < n; j++) q1[4 * j + k] -= dx;
      for (int j = 0; j < 4 * n; j++) rh[j] = -(fp[j] - fc[j]) / dx * (tf - to) * p->ut / 2.0;
      for (int j = 0; j < n; j++)
        for (int c = 0; c < 4; c++) sub[(4 * j + c) * W + 4 * (j + 1) + k] += rh[4 * j + c];
    }
    for (int j = 4 * ua; j < 4 * ub; j++)
      for (int c = 4 * xa; c < 4 * xb; c++) put(Jq, j, c, sub[(size_t)(j - 4 * ua) * W + (c - 4 * xa)]);
    for (int k = 0; k < 2; k++) {
      for (int j = 0; j < n; j++) u[2 * j + k] += dx;
      orc_dynamics_quaternion(n, q1, u, p->uu, fp);
      for (int j = 0; j < n; j++) u[2 * j + k] -= dx;
      for (int j = 0; j < 4 * n; j++) rh[j] = -(fp[j] - fc[j]) / dx * (tf - to) * p->ut / 2.0;
      for (int j = ua; j < ub; j++)
        for (int c = 0; c < 4; c++) put(Ju, 4 * j + c, 2 * j + k, rh[4 * (j - ua) + c]);
    }
    for (int j = 0; j < 4 * n; j++) put(Jt, 4 * ua + j, i, fc[j] * p->ut / 2.0);
    for (int j = 0; j < 4 * n; j++) put(Jt, 4 * ua + j, i + 1, -(fc[j] * p->ut / 2.0));
  }
  free(x); free(fc); free(fp); free(rh);
}

void orc_jacobian(const orc_problem* p, int g, const double* x, int32_t* rows, int32_t* cols, double* vals) {
  switch (g) {
    case 0: jac_mass(p, x, rows, cols, vals); break;
    case 1: jac_pos(p, x, rows, cols, vals); break;
    case 2: jac_vel(p, x, rows, cols, vals); break;
    default: jac_quat(p, x, rows, cols, vals); break;
  }
}

/* ------------------------------------------------------------------ */
/* generic dense forward difference: lib/jac_fd.py:29-62               */
/* ------------------------------------------------------------------ */
void orc_jac_fd(const orc_problem* p, int g, const double* x_in, double* J) {
  int nv = orc_num_vars(p), nr = orc_num_rows(p, g);
  double* x = (double*)malloc(nv * sizeof(double));
  double* g0 = (double*)malloc(nr * sizeof(double));
  double* g1 = (double*)malloc(nr * sizeof(double));
  memcpy(x, x_in, nv * sizeof(double));
  orc_residual(p, g, x, g0);
  for (int i = 0; i < nv; i++) {
    x[i] += p->dx;
    orc_residual(p, g, x, g1);
    for (int r = 0; r < nr; r++) J[(size_t)r * nv + i] = (g1[r] - g0[r]) / p->dx;
    x[i] -= p->dx;
  }
  free(x); free(g0); free(g1);
}

/* lib/cost_gradient.py:29-47 */
double orc_cost(const orc_problem* p, const double* x, int payload_mode) {
  xview X = view(p, x);
  return payload_mode ? -X.mass[0] : X.t[p->S];
}
void orc_cost_jac(const orc_problem* p, const double* x, int payload_mode, double* grad) {
  (void)x;
  if (payload_mode) { memset(grad, 0, p->M * sizeof(double)); grad[0] = -1.0; }
  else { memset(grad, 0, (p->S + 1) * sizeof(double)); grad[p->S] = 1.0; }
}

void orc_eval_batch(const orc_problem* p, int B, const double* x, double* res, double* vals, int nthreads) {
  int nv = orc_num_vars(p);
  int64_t nr = 11 * (int64_t)p->N, nnz = orc_total_nnz(p);
  int64_t goff[4], roff[4] = {0, p->N, 4 * (int64_t)p->N, 7 * (int64_t)p->N};
  int64_t s = 0;
  for (int g = 0; g < 4; g++) {
    goff[g] = s;
    for (int b = 0; b < orc_num_blocks(g); b++) s += orc_block_nnz(p, g, b);
  }
  (void)nthreads;
  /* vals == NULL: timing mode -- every thread writes its Jacobian values into one scratch array it
   * allocates once, instead of B fresh 5-MB rows (whose page faults serialise the threads) */
#ifdef _OPENMP
#pragma omp parallel num_threads(nthreads > 0 ? nthreads : 1)
#endif
  {
    double* scratch = vals ? NULL : (double*)malloc(sizeof(double) * (size_t)(nnz > 0 ? nnz : 1));
#ifdef _OPENMP
#pragma omp for schedule(dynamic)
#endif
    for (int b = 0; b < B; b++) {
      const double* xb = x + (size_t)b * nv;
      double* vb = vals ? vals + (size_t)b * nnz : scratch;
      for (int g = 0; g < 4; g++) {
        orc_residual(p, g, xb, res + (size_t)b * nr + roff[g]);
        orc_jacobian(p, g, xb, NULL, NULL, vb + goff[g]);
      }
    }
    free(scratch);
  }
}

/* ================================================================== */
/* aerodynamic path constraints                   */
/* (lib/con_aero.py; src/wrapper_utils.hpp:89-206)                     */
/* ================================================================== */
static void air_velocity_eci(const double pos[3], const double vel[3], double t, const double* wind, int Kw,
                             double vair[3], double* altitude_out) {
  /* the chain shared by the three functions, wrapper_utils.hpp:93-100,165-172 */
  double llh[3], vel_ecef[3], wned[3], qn2i[4], wind_eci[3];
  orc_ecef2geodetic(pos[0], pos[1], pos[2], llh);
  double altitude = orc_geopotential_altitude(llh[2]);
  orc_vel_eci2ecef(vel, pos, t, vel_ecef);
  orc_wind_ned(altitude, wind, Kw, wned);
  orc_quat_nedg2eci(pos, t, qn2i);
  orc_quatrot(qn2i, wned, wind_eci);
  orc_ecef2eci(vel_ecef, t, vair);
  vair[0] -= wind_eci[0]; vair[1] -= wind_eci[1]; vair[2] -= wind_eci[2];
  *altitude_out = altitude;
}

double orc_angle_of_attack_all_rad(const double pos[3], const double vel[3], const double quat[4], double t,
                                   const double* wind, int Kw) { /* wrapper_utils.hpp:89-111 */
  double qc[4], ex[3] = {1.0, 0.0, 0.0}, dir[3], vair[3], alt;
  quat_conj(quat, qc);
  orc_quatrot(qc, ex, dir);
  air_velocity_eci(pos, vel, t, wind, Kw, vair, &alt);
  double nv = sqrt(vair[0] * vair[0] + vair[1] * vair[1] + vair[2] * vair[2]);
  double nd = sqrt(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]);
  double c_alpha = (vair[0] / nv) * (dir[0] / nd) + (vair[1] / nv) * (dir[1] / nd) + (vair[2] / nv) * (dir[2] / nd);
  if (c_alpha > 1.0) return 0.0;
  else if (nv < 1e-6) return 0.0;
  else return acos(c_alpha);
}

double orc_dynamic_pressure_pa(const double pos[3], const double vel[3], double t, const double* wind, int Kw) {
  /* wrapper_utils.hpp:163-175 */
  double vair[3], alt;
  air_velocity_eci(pos, vel, t, wind, Kw, vair, &alt);
  double rho = orc_air_density(alt);
  double nv = sqrt(vair[0] * vair[0] + vair[1] * vair[1] + vair[2] * vair[2]);
  return 0.5 * rho * nv * nv;
}

double orc_q_alpha_pa_rad(const double pos[3], const double vel[3], const double quat[4], double t,
                          const double* wind, int Kw) { /* wrapper_utils.hpp:190-195 */
  double alpha = orc_angle_of_attack_all_rad(pos, vel, quat, t, wind, Kw);
  double q = orc_dynamic_pressure_pa(pos, vel, t, wind, Kw);
  return q * alpha;
}

/* constraint specs: kind 0 = AOA_max, 1 = dynamic_pressure_max, 2 = Q_alpha_max (condition[...] of con_aero.py) */
typedef struct { int phase, all; double limit; } aero_spec;
#define ORC_MAX_AERO 64
static aero_spec g_unused_spec; /* silence unused warnings on some compilers */

typedef struct { int n; aero_spec s[ORC_MAX_AERO]; } aero_list;
/* stored beside the problem (kept out of struct orc_problem's original layout) */
typedef struct aero_store { const orc_problem* p; aero_list k[3]; struct aero_store* next; } aero_store;
static aero_store* g_aero = NULL;

static aero_store* aero_of(const orc_problem* p, int create) {
  for (aero_store* a = g_aero; a; a = a->next) if (a->p == p) return a;
  if (!create) return NULL;
  aero_store* a = (aero_store*)calloc(1, sizeof(aero_store));
  a->p = p; a->next = g_aero; g_aero = a;
  (void)g_unused_spec;
  return a;
}

int orc_aero_configure(const orc_problem* p, int kind, int nspec, const int32_t* phase, const int32_t* range_all,
                       const double* limit) {
  if (kind < 0 || kind > 2 || nspec > ORC_MAX_AERO) return -1;
  aero_store* a = aero_of(p, 1);
  a->k[kind].n = nspec;
  for (int i = 0; i < nspec; i++) { a->k[kind].s[i].phase = phase[i]; a->k[kind].s[i].all = range_all[i]; a->k[kind].s[i].limit = limit[i]; }
  return 0;
}

static int aero_nk(const orc_problem* p, const aero_spec* s) { return s->all ? p->n[s->phase] + 1 : 1; }

int orc_aero_rows(const orc_problem* p, int kind) { /* inequality_length_max_*: con_aero.py:254-309 */
  aero_store* a = aero_of(p, 0);
  int r = 0;
  if (a) for (int i = 0; i < a->k[kind].n; i++) r += aero_nk(p, &a->k[kind].s[i]);
  return r;
}

/* the *_array_dimless helpers, con_aero.py:39-87: scale, evaluate, divide by units[3] */
static double aero_value(const orc_problem* p, int kind, const double* pos_e, const double* vel_e, const double* q,
                         double t_e, double limit) {
  double pos[3] = {pos_e[0] * p->up, pos_e[1] * p->up, pos_e[2] * p->up};
  double vel[3] = {vel_e[0] * p->uv, vel_e[1] * p->uv, vel_e[2] * p->uv};
  double t = t_e * p->ut;
  double f = kind == 0 ? orc_angle_of_attack_all_rad(pos, vel, q, t, p->wind, p->Kw)
           : kind == 1 ? orc_dynamic_pressure_pa(pos, vel, t, p->wind, p->Kw)
                       : orc_q_alpha_pa_rad(pos, vel, q, t, p->wind, p->Kw);
  return f / limit;
}

static double time_node(const orc_problem* p, int i, int k, double to, double tf) { /* SectionParameters.py:77-81 */
  return k == 0 ? to : p->tau[i][k - 1] * (tf - to) / 2 + (tf + to) / 2;
}

void orc_aero_residual(const orc_problem* p, int kind, const double* x, double* out) {
  /* inequality_max_alpha / _q / _qalpha: con_aero.py:90-252 */
  aero_store* a = aero_of(p, 0);
  if (!a) return;
  xview X = view(p, x);
  int row = 0;
  for (int s = 0; s < a->k[kind].n; s++) {
    const aero_spec* sp = &a->k[kind].s[s];
    int i = sp->phase, xa = p->ua[i] + i, nk = aero_nk(p, sp);
    double to = X.t[i], tf = X.t[i + 1];
    for (int k = 0; k < nk; k++)
      out[row++] = 1.0 - aero_value(p, kind, X.pos + 3 * (xa + k), X.vel + 3 * (xa + k), X.quat + 4 * (xa + k),
                                    time_node(p, i, k, to, tf), sp->limit);
  }
}

int64_t orc_aero_nnz(const orc_problem* p, int kind, int var) { /* var: 0 position, 1 velocity, 2 quaternion, 3 t */
  int64_t rows = orc_aero_rows(p, kind);
  static const int w[4] = {3, 3, 4, 2};
  if (var == 2 && kind == 1) return 0;
  return rows * w[var];
}

void orc_aero_jacobian(const orc_problem* p, int kind, const double* x_in, int32_t* rows, int32_t* cols, double* vals) {
  /* inequality_jac_max_*: con_aero.py:311-471 (+ the q and q-alpha twins); blocks concatenated in key order
   * position, velocity, quaternion, t.  Gradients by forward differences with the reference's in-place
   * "+= dx ... -= dx" on copies (con_aero.py:324-371 fancy-indexes, i.e. copies, the node rows). */
  aero_store* a = aero_of(p, 0);
  if (!a) return;
  xview X = view(p, x_in);
  const double dx = p->dx;
  coo o[4];
  int64_t off = 0;
  for (int v = 0; v < 4; v++) {
    o[v].r = rows ? rows + off : NULL; o[v].c = cols ? cols + off : NULL; o[v].v = vals + off; o[v].k = 0;
    off += orc_aero_nnz(p, kind, v);
  }
  int iRow = 0;
  for (int s = 0; s < a->k[kind].n; s++) {
    const aero_spec* sp = &a->k[kind].s[s];
    int i = sp->phase, xa = p->ua[i] + i, nk = aero_nk(p, sp);
    double to = X.t[i], tf = X.t[i + 1];
    double* gp = (double*)malloc(sizeof(double) * nk * 12);
    for (int k = 0; k < nk; k++) {
      double r[3], v[3], q[4];
      memcpy(r, X.pos + 3 * (xa + k), 24); memcpy(v, X.vel + 3 * (xa + k), 24); memcpy(q, X.quat + 4 * (xa + k), 32);
      double tk = time_node(p, i, k, to, tf);
      double fc = aero_value(p, kind, r, v, q, tk, sp->limit);
      for (int j = 0; j < 3; j++) { r[j] += dx; gp[k * 12 + j] = (aero_value(p, kind, r, v, q, tk, sp->limit) - fc) / dx; r[j] -= dx; }
      for (int j = 0; j < 3; j++) { v[j] += dx; gp[k * 12 + 3 + j] = (aero_value(p, kind, r, v, q, tk, sp->limit) - fc) / dx; v[j] -= dx; }
      for (int j = 0; j < 4; j++) { q[j] += dx; gp[k * 12 + 6 + j] = (aero_value(p, kind, r, v, q, tk, sp->limit) - fc) / dx; q[j] -= dx; }
      gp[k * 12 + 10] = (aero_value(p, kind, r, v, q, time_node(p, i, k, to + dx, tf), sp->limit) - fc) / dx;
      gp[k * 12 + 11] = (aero_value(p, kind, r, v, q, time_node(p, i, k, to, tf + dx), sp->limit) - fc) / dx;
    }
    for (int j = 0; j < 3; j++) for (int k = 0; k < nk; k++) put(&o[0], iRow + k, (xa + k) * 3 + j, -gp[k * 12 + j]);
    for (int j = 0; j < 3; j++) for (int k = 0; k < nk; k++) put(&o[1], iRow + k, (xa + k) * 3 + j, -gp[k * 12 + 3 + j]);
    if (kind != 1)
      for (int j = 0; j < 4; j++) for (int k = 0; k < nk; k++) put(&o[2], iRow + k, (xa + k) * 4 + j, -gp[k * 12 + 6 + j]);
    for (int k = 0; k < nk; k++) put(&o[3], iRow + k, i, -gp[k * 12 + 10]);
    for (int k = 0; k < nk; k++) put(&o[3], iRow + k, i + 1, -gp[k * 12 + 11]);
    iRow += nk;
    free(gp);
  }
}
