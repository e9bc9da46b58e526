// gel_kernels.hip -- HIP kernels (gfx950 / CDNA4, fp64) of the LGR defect-residual
// and forward-difference-Jacobian hot path, plus their launchers.
//
// Kernel inventory
//   eval_kernel<JAC,MFMA>  one wavefront per (decision vector, phase, 64-node chunk), one lane per node:
//                      D.X rows, centre RHS, every FD sweep, residual rows and the
//                      x-dependent Jacobian entries of that node, fused.
//   expand_kernel      compact Jacobian entries + constant template -> full COO values
//   perturb_kernel / quotient_kernel   column-batched generic forward difference
//   rhs_*_kernel, point_kernel         node-batched RHS / point-function hooks
//
// Data layout (HBM):
//   x     [B][nvars]  packed xdict (AoS per node, as the boundary hands it over)
//   res   [B][11N]    mass N | pos 3N | vel 3N | quat 4N
//   jvar  [B][V]      per phase [slot][node]: consecutive lanes (nodes) write
//                     consecutive doubles -> every store instruction is one
//                     contiguous 512-byte segment per wavefront.
//   Dt    per phase, transposed (Dt[i*n + j] = D[j][i]) so lane j reads
//         consecutive addresses while all lanes share X[i][c] (broadcast).
#include <hip/hip_runtime.h>

#include "gel_tables.h"

#include "gel_eval_kernel.h"
namespace gel {

// ---------------------------------------------------------------------------
// compact -> full COO values.  src[i] = -1: constant cval[i]; s >= 0: jvar[b][s]; s <= -2: -jvar[b][-2 - s].
// Two doubles (16 B) per lane: wide coalesced stores.  A thread keeps its two template entries
// (cval, src: 12 B per entry, 7.3 MB at 6x64 -- larger than one XCD's L2) in registers and re-uses them
// for kExpandGroup decision vectors, so the template is read once per group instead of once per vector
// and the kernel is bounded by its 8 B/entry of HBM writes (5.8 TB/s with non-temporal stores).
// ---------------------------------------------------------------------------
constexpr int kExpandGroup = 8;
__global__ __launch_bounds__(kBlock) void expand_kernel(long long nnz, long long V, int B, const double* __restrict__ cval,
                                                        const int32_t* __restrict__ src,
                                                        const double* __restrict__ jvar, double* __restrict__ full) {
  const int b0 = blockIdx.y * kExpandGroup;
  const int nb = min(kExpandGroup, B - b0);
  const bool even = (nnz & 1) == 0;  // then every vector's slice starts 16-byte aligned
  for (long long i = 2 * ((long long)blockIdx.x * kBlock + threadIdx.x); i < nnz; i += 2LL * gridDim.x * kBlock) {
    const bool two = i + 1 < nnz;
    const int s0 = src[i], s1 = two ? src[i + 1] : -1;
    const double c0 = (s0 == -1) ? cval[i] : 0.0, c1 = (two && s1 == -1) ? cval[i + 1] : 0.0;
    const int g0 = (s0 >= 0) ? s0 : -2 - s0, g1 = (s1 >= 0) ? s1 : -2 - s1;  // compact index (unused for constants)
    for (int g = 0; g < nb; g++) {
      const double* jv = jvar + (size_t)(b0 + g) * V;
      double* out = full + (size_t)(b0 + g) * nnz + i;
      const double v0 = (s0 == -1) ? c0 : ((s0 >= 0) ? jv[g0] : -jv[g0]);
      const double v1 = (s1 == -1) ? c1 : ((s1 >= 0) ? jv[g1] : -jv[g1]);
      if (two && (even || ((b0 + g) & 1) == 0)) {
        // written once, read by someone else later: non-temporal (4.2 -> 5.8 TB/s measured)
        typedef double gel_d2 __attribute__((ext_vector_type(2)));
        __builtin_nontemporal_store(gel_d2{v0, v1}, reinterpret_cast<gel_d2*>(out));
      } else {
        __builtin_nontemporal_store(v0, out);
        if (two) __builtin_nontemporal_store(v1, out + 1);
      }
    }
  }
}

// ---------------------------------------------------------------------------
// Update of a full COO value buffer that already holds the constants: only the x-dependent
// entries are written -- full[b][vdst[i]] = +-jvar[b][|vsrc[i]|] for the nvar entries the gather map takes from the compact
// vector (26 k of the 607 k at mixed-6x64), vdst ascending, so the runs that are contiguous in the reference's emission order
// (the [node][xyz] blocks: four fifths of the entries) are coalesced stores and the rest (the diagonal of the dense velocity
// blocks, the pairs of the quaternion blocks) are single 8 / 16-byte writes.  4 % of the bytes expand_kernel writes.
// fill_full_kernel lays the constant template down once (or again whenever the buffer is re-used for other data).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void update_full_kernel(long long nnz, long long V, int nvar, int B, const int32_t* __restrict__ vdst,
                                                             const int32_t* __restrict__ vsrc, const double* __restrict__ jvar,
                                                             double* __restrict__ full) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= nvar) return;
  const int d = vdst[i], s = vsrc[i];
  const int g = (s >= 0) ? s : -2 - s;
  for (int b = blockIdx.y; b < B; b += gridDim.y) {
    const double v = jvar[(size_t)b * V + g];
    full[(size_t)b * nnz + d] = (s >= 0) ? v : -v;
  }
}
// The same update by whole 64-byte lines: every line of a vector's value array that holds at least one x-dependent entry is written
// in full, its constants from the template -- eight lanes per line.  An entry that sits alone between constants (the diagonal of the
// dense velocity blocks: 2,880 per vector at mixed-6x64) is then a full-line write instead of an 8-byte one, which HBM (ECC words of
// 32 / 64 bytes) turns into a read-modify-write: measured 0.26 -> see DESIGN.md ms at B = 1024 for 8 x fewer partial writes.  Needs
// nnz % 8 == 0 (every vector's array starts on a line); launch_update_full() falls back to the entry-wise kernel otherwise.
__global__ __launch_bounds__(kBlock) void update_lines_kernel(long long nnz, long long V, int nlines, int B, const int32_t* __restrict__ vline,
                                                              const int32_t* __restrict__ src, const double* __restrict__ cval,
                                                              const double* __restrict__ jvar, double* __restrict__ full) {
  const int t = blockIdx.x * kBlock + threadIdx.x;
  if (t >= 8 * nlines) return;
  const long long i = 8LL * vline[t >> 3] + (t & 7);
  const int s = src[i];
  const double c = (s == -1) ? cval[i] : 0.0;
  const int g = (s >= 0) ? s : -2 - s;
  for (int b = blockIdx.y; b < B; b += gridDim.y) {
    double v = c;
    if (s != -1) { const double w = jvar[(size_t)b * V + g]; v = (s >= 0) ? w : -w; }
    __builtin_nontemporal_store(v, full + (size_t)b * nnz + i);
  }
}
__global__ __launch_bounds__(kBlock) void fill_full_kernel(long long nnz, int B, const double* __restrict__ cval, double* __restrict__ full) {
  for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < nnz; i += (long long)gridDim.x * kBlock) {
    const double c = cval[i];
    for (int b = blockIdx.y; b < B; b += gridDim.y) __builtin_nontemporal_store(c, full + (size_t)b * nnz + i);
  }
}

// ---------------------------------------------------------------------------
// generic column-batched forward difference (lib/jac_fd.py:29-62) of the four defect residuals, PHASE BY PHASE.
// The residual rows of a phase depend only on the 13 n + 13 columns of that phase (its state and control nodes and its
// two knot times; lib/con_dynamics.py:46,132,237,512 are loops over phases with no cross-phase data): every other
// column gives bit-identical residual rows, i.e. an exact zero in the reference's dense Jacobian.  So each phase is
// differenced as its own one-phase problem: its local decision vector [mass n+1 | pos 3(n+1) | vel 3(n+1) | quat 4(n+1)
// | u 2n | t0 tf] is gathered from x (colmap: local column -> global column), the 13 n + 14 perturbed copies are
// formed, one residual-only launch evaluates them, and the quotients go straight to their rows and columns of the
// dense Jacobian.  No (num_vars + 1) x num_vars copy of x is ever built.
// ---------------------------------------------------------------------------
// Xp[0] = gathered local x; Xp[c + 1] = the same with local column c += dx
__global__ void perturb_local_kernel(int nloc, double dx, const double* __restrict__ x, const int32_t* __restrict__ colmap,
                                     double* __restrict__ Xp) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)(nloc + 1) * nloc) return;
  const int row = (int)(t / nloc), col = (int)(t - (long long)row * nloc);
  double v = x[colmap[col]];
  if (row == col + 1) v += dx;
  Xp[t] = v;
}

// J[(row0 + r) * ldJ + colmap[c]] = (res[c + 1][roff + r] - res[0][roff + r]) / dx     (tiled transpose through LDS; no colmap: c itself)
__global__ void quotient_local_kernel(int nloc, int nres, int roff, int nrows, double dx, const double* __restrict__ res,
                                      double* __restrict__ J, long long ldJ, int row0, const int32_t* __restrict__ colmap) {
  __shared__ double tile[32][33];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  for (int k = threadIdx.y; k < 32; k += blockDim.y) {  // read: r fast
    const int c = c0 + k, r = r0 + threadIdx.x;
    if (c < nloc && r < nrows) tile[k][threadIdx.x] = res[(size_t)(c + 1) * nres + roff + r];
  }
  __syncthreads();
  for (int k = threadIdx.y; k < 32; k += blockDim.y) {  // write: column fast (local columns of one variable are contiguous globally)
    const int r = r0 + k, c = c0 + threadIdx.x;
    if (c < nloc && r < nrows) J[(size_t)(row0 + r) * ldJ + (colmap ? colmap[c] : c)] = (tile[threadIdx.x][k] - res[roff + r]) / dx;
  }
}

// ---------------------------------------------------------------------------
// node-batched RHS hooks (dynamics_c replacements, src/pybind_dynamics.cpp:30-106)
// ---------------------------------------------------------------------------
struct RhsArgs {
  int n, Kw, Kc;
  const double *mass_e, *pos_e, *vel_e, *quat, *t, *tables;
  double thrust, area, nozzle, um, up, uv, barC20;
  double* out;
};

__global__ void rhs_vel_air_kernel(RhsArgs A) {
  extern __shared__ double lds[];
  ProblemDev P{};
  P.Kw = A.Kw; P.Kc = A.Kc; P.tables = A.tables;
  const Tables tb = stage_tables(P, lds);
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= A.n) return;
  const double m = A.mass_e[i] * A.um;
  const double r[3] = {A.pos_e[3 * i] * A.up, A.pos_e[3 * i + 1] * A.up, A.pos_e[3 * i + 2] * A.up};
  const double v[3] = {A.vel_e[3 * i] * A.uv, A.vel_e[3 * i + 1] * A.uv, A.vel_e[3 * i + 2] * A.uv};
  const double q[4] = {A.quat[4 * i], A.quat[4 * i + 1], A.quat[4 * i + 2], A.quat[4 * i + 3]};
  const PosPart pp = pos_part(r, tb, A.barC20);
  const EarthAngle ea = earth_angle(A.t[i]);
  double w[3], F[3], dir[3], f[3];
  wind_eci(r, ea, pp.shp, pp.chp, pp.inv_p, pp.wn, pp.we, w);
  aero_force(r, v, pp.rho, pp.inv_a, ea, w, A.area, tb, F);
  thrust_dir(q, dir);
  const double T = A.thrust - A.nozzle * pp.P;
  const double Td[3] = {T * dir[0], T * dir[1], T * dir[2]};
  accel(Td, F, 1.0 / m, pp.g, 1.0 / A.uv, f);
  for (int c = 0; c < 3; c++) A.out[3 * i + c] = f[c];
}

__global__ void rhs_vel_noair_kernel(RhsArgs A) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= A.n) return;
  const double m = A.mass_e[i] * A.um;
  const double r[3] = {A.pos_e[3 * i] * A.up, A.pos_e[3 * i + 1] * A.up, A.pos_e[3 * i + 2] * A.up};
  const double q[4] = {A.quat[4 * i], A.quat[4 * i + 1], A.quat[4 * i + 2], A.quat[4 * i + 3]};
  double dir[3], g[3], f[3];
  thrust_dir(q, dir);
  gravity_eci(r, A.barC20, g);
  const double Td[3] = {A.thrust * dir[0], A.thrust * dir[1], A.thrust * dir[2]};
  accel_noair(Td, 1.0 / m, g, 1.0 / A.uv, f);
  for (int c = 0; c < 3; c++) A.out[3 * i + c] = f[c];
}

__global__ void rhs_quat_kernel(int n, const double* quat, const double* u_e, double unit_u, double* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double q[4] = {quat[4 * i], quat[4 * i + 1], quat[4 * i + 2], quat[4 * i + 3]};
  double dq[4];
  quat_rate(q, u_e[2 * i], u_e[2 * i + 1], unit_u, dq);
  for (int c = 0; c < 4; c++) out[4 * i + c] = dq[c];
}

__global__ void point_kernel(int kind, int n, const double* in, const double* aux, int aux_rows, double* out) {
  extern __shared__ double lds[];
  // aux table (wind [K][3] or generic [K][2]) and the atmosphere table are staged in LDS
  // kinds 5 / 6: rows followed by the per-interval slopes (built by the host entry point)
  const int naux = (kind == 5 || kind == 10) ? 3 * aux_rows + 2 * (aux_rows - 1)
                   : (kind == 6 || kind == 9) ? 2 * aux_rows + (aux_rows - 1) : (kind == 0 ? kAtmDoubles : 0);
  for (int i = threadIdx.x; i < naux; i += blockDim.x) lds[i] = aux[i];
  __syncthreads();
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  switch (kind) {
    case 0: {
      const double h = geopotential_altitude(in[i]);
      const Air a = atmosphere(h, lds);
      out[5 * i] = h; out[5 * i + 1] = a.T; out[5 * i + 2] = a.P; out[5 * i + 3] = a.rho; out[5 * i + 4] = a.a;
    } break;
    case 1: {
      // latitude / longitude in degrees (atan2, as the reference reports them); the altitude exactly as pos_part forms it
      // (from the algebraic sine / cosine pair of the latitude)
      double lat, sl, cl, p, ip;
      geodetic_lat_p(in[3 * i], in[3 * i + 1], in[3 * i + 2], lat, p, ip);
      geodetic_sincos_p(in[3 * i], in[3 * i + 1], in[3 * i + 2], sl, cl, p, ip);
      const double lon = atan2(in[3 * i + 1], in[3 * i]);
      out[3 * i] = lat * 180.0 / kPi; out[3 * i + 1] = lon * 180.0 / kPi;
      out[3 * i + 2] = geodetic_alt_from(p, sl, cl);
    } break;
    case 2: {
      const double r[3] = {in[3 * i], in[3 * i + 1], in[3 * i + 2]};
      double g[3];
      gravity_eci(r, aux[0], g);
      out[3 * i] = g[0]; out[3 * i + 1] = g[1]; out[3 * i + 2] = g[2];
    } break;
    case 3: {
      // wind vector NED -> ECI exactly as the hot path does it: in = pos[3], t, wn, we
      const double* a = in + 6 * i;
      const double r[3] = {a[0], a[1], a[2]};
      double sl, cl, p, ip, w[3];
      geodetic_sincos_p(r[0], r[1], r[2], sl, cl, p, ip);
      const EarthAngle ea = earth_angle(a[3]);
      double chp, irt;
      fsqrt_rsqrt(0.5 * (1.0 + cl), chp, irt);
      wind_eci(r, ea, (0.5 * sl) * irt, chp, ip, a[4], a[5], w);
      out[3 * i] = w[0]; out[3 * i + 1] = w[1]; out[3 * i + 2] = w[2];
    } break;
    case 4: {
      const double* a = in + 7 * i;  // vel[3], pos[3], t
      double s, c;
      sincos(kOmega * a[6], &s, &c);
      const double d0 = a[0] - (0.0 * a[5] - kOmega * a[4]);
      const double d1 = a[1] - (kOmega * a[3] - 0.0 * a[5]);
      const double d2 = a[2] - (0.0 * a[4] - 0.0 * a[3]);
      out[3 * i] = d0 * c + d1 * s; out[3 * i + 1] = -d0 * s + d1 * c; out[3 * i + 2] = d2;
    } break;
    case 5: {
      double wn, we;
      wind_ned2(in[i], lds, lds + 3 * aux_rows, aux_rows, wn, we);
      out[3 * i] = wn; out[3 * i + 1] = we; out[3 * i + 2] = 0.0;
    } break;
    case 6: out[i] = interp_tab(in[i], lds, lds + 2 * aux_rows, aux_rows, 2, 1); break;
    case 7: {  // the path's guard-free sqrt / division beside the compiler's, for the bit-identity test
      const double a = in[2 * i], b = in[2 * i + 1];
      out[4 * i] = fsqrt(a); out[4 * i + 1] = sqrt(a); out[4 * i + 2] = fdiv(a, b); out[4 * i + 3] = a / b;
    } break;
    case 8: {  // the path's sincos / log beside the library's, for the accuracy test
      double s1, c1, s2, c2;
      fsincos(in[2 * i], &s1, &c1);
      sincos(in[2 * i], &s2, &c2);
      out[6 * i] = s1; out[6 * i + 1] = c1; out[6 * i + 2] = s2; out[6 * i + 3] = c2;
      out[6 * i + 4] = flog_ratio(in[2 * i + 1]); out[6 * i + 5] = log(in[2 * i + 1]);
    } break;
    case 9: {  // eight lookups in a row through the interval kept from the previous one (the fused kernel's CA lookups)
      Bracket br = no_bracket();
      for (int k = 0; k < 8; k++) out[8 * i + k] = interp_tab_cached(in[8 * i + k], lds, lds + 2 * aux_rows, aux_rows, 2, 1, br);
    } break;
    case 10: {  // likewise the wind lookups of a node's position evaluations
      Bracket2 br = no_bracket2();
      for (int k = 0; k < 8; k++) {
        double wn, we;
        wind_ned2_cached(in[8 * i + k], lds, lds + 3 * aux_rows, aux_rows, wn, we, br);
        out[16 * i + 2 * k] = wn; out[16 * i + 2 * k + 1] = we;
      }
    } break;
    case 11: {  // quatmult (src/wrapper_coordinate.hpp:50-57): in = q[4], p[4]
      double o[4];
      quatmult(in + 8 * i, in + 8 * i + 4, o);
      for (int c = 0; c < 4; c++) out[4 * i + c] = o[c];
    } break;
    case 12: {  // quatrot (:70-78) = vec(conj(q) (0, v) q): in = q[4], v[3]
      double o[3];
      quatrot(in + 7 * i, in + 7 * i + 4, o);
      for (int c = 0; c < 3; c++) out[3 * i + c] = o[c];
    } break;
    case 13: {  // conj (:59-61) and the thrust direction quatrot(conj(q), (1, 0, 0)) the hot path forms without the zero terms: in = q[4]
      const double* q = in + 4 * i;
      out[7 * i] = q[0]; out[7 * i + 1] = -q[1]; out[7 * i + 2] = -q[2]; out[7 * i + 3] = -q[3];
      double d[3];
      thrust_dir(q, d);
      for (int c = 0; c < 3; c++) out[7 * i + 4 + c] = d[c];
    } break;
    case 14: out[2 * i] = fexp(in[i]); out[2 * i + 1] = exp(in[i]); break;   // the path's exp beside the library's, for the bit-identity test
    default: break;
  }
}

// ---------------------------------------------------------------------------
// Aero path constraints: lib/con_aero.py:90-252 (values) and :311-756 (forward-
// difference gradients), device functions of src/wrapper_utils.hpp:89-206.  kind 0 = angle of attack,
// 1 = dynamic pressure, 2 = q*alpha.
//
// ONE launch serves all three kinds.  A constrained state node (phase, k) appears once, whatever kinds constrain it,
// and the chain (geodetic -> wind -> Earth angle -> wind in ECI -> air-relative velocity; thrust direction / atmosphere)
// is split by what each piece depends on, as in gel_rhs_parts.h: a sweep recomputes only what its perturbed variable
// enters (a reused piece is bit-identical to what the reference's full re-evaluation produces, because the perturbed
// variable does not enter it).  Eight lanes per node: lane 0 the centre value plus the light sweeps (velocity 3:
// only the air-relative velocity changes; quaternion 4: only the thrust direction), lanes 1..3 one position sweep
// each (the whole chain), lanes 4..5 the t0 / tf sweeps (Earth angle -> wind rotation -> air-relative velocity).  The
// centre values reach the other lanes by a wavefront shuffle.  4 + 2 heavy evaluations per node instead of the
// 3 kinds x 13 of one-lane-per-row.
// ---------------------------------------------------------------------------
struct AeroPos {            // depends on position only
  double rho;               // density at the node
  double wn, we;            // wind, NED
  double shp, chp, inv_p;   // NED half-latitude pair, 1/p
};

GEL_DEV AeroPos aero_pos_part(const double r[3], const Tables& tb) {
  AeroPos o;
  double lat, p, sl, cl;
  geodetic_lat_p(r[0], r[1], r[2], lat, p, o.inv_p);
  fsincos(lat, &sl, &cl);
  const double h = geopotential_altitude(geodetic_alt_from(p, sl, cl));
  wind_ned2(h, tb.wind, tb.winds, tb.Kw, o.wn, o.we);
  double irt;
  fsqrt_rsqrt(0.5 * (1.0 + cl), o.chp, irt);
  o.shp = (0.5 * sl) * irt;
  o.rho = atmosphere(h, tb.atm).rho;      // wrapper_utils.hpp:163-175
  return o;
}

// air-relative velocity in ECI (wrapper_utils.hpp:93-100) and its norm
GEL_DEV double aero_vair(const double r[3], const double v[3], const EarthAngle& ea, const double w[3], double a[3]) {
  const double d0 = v[0] + kOmega * r[1], d1 = v[1] - kOmega * r[0];
  const double e0 = d0 * ea.c + d1 * ea.s, e1 = -d0 * ea.s + d1 * ea.c;
  a[0] = (e0 * ea.c - e1 * ea.s) - w[0]; a[1] = (e0 * ea.s + e1 * ea.c) - w[1]; a[2] = v[2] - w[2];
  return sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
}

// angle of attack (wrapper_utils.hpp:89-111)
GEL_DEV double aero_alpha(const double a[3], double nv, const double q[4]) {
  double dir[3];
  thrust_dir(q, dir);
  const double nd = sqrt(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]);
  const double c_alpha = (a[0] / nv) * (dir[0] / nd) + (a[1] / nv) * (dir[1] / nd) + (a[2] / nv) * (dir[2] / nd);
  return (c_alpha > 1.0) ? 0.0 : ((nv < 1e-6) ? 0.0 : acos(c_alpha));
}

struct AeroOut {
  double* con[3];   // [B][nrows[kind]]
  double* jac[3];   // [B][nrows[kind] * (8 + 4 (kind != 1))]: position | velocity | quaternion | t blocks
  int32_t nrows[3];
  int64_t ld;       // != 0: the outputs are per-vector RECORDS of ld doubles (gel_eval_batch_aero_device): con / jac point at the
                    // kind's part of record 0, a vector's part lies b * ld doubles on; 0: dense [B][...] arrays per kind
  int32_t sm;       // != 0 (records only): SPEC-MAJOR part A (gel_device.h AeroPhaseDev) -- a node's row0[kind] is the first double of
                    // its spec's block in the record, row[kind] its constraint value's place; con / jac all point at the record
};

// One WAVEFRONT = 64 consecutive constrained nodes of one decision vector, every sweep of the node in the same lane (round 2 ran
// one wavefront per sweep role and the whole chain in each: six, then four runs of the atmosphere chain per node):
//   centre      pos_part<CENTRE> (the fused kernel's position part without gravity), wind into ECI, air-relative velocity
//               (the rotation by omega t and back is not performed: gel_rhs_parts.h aero_force), alpha, q;
//   position    EXACT-DIFFERENCE form (pos_delta(), gel_rhs_parts.h): the change of latitude pair, altitude, density and wind
//               from algebraic identities of the reference's formulas, ~130 operations instead of a second run of the chain,
//               then wind -> air velocity -> alpha, q at the perturbed point; a wavefront with a lane the form does not
//               cover (another atmosphere layer / table piece, next to the polar axis) recomputes that sweep in full;
//   velocity    only the air-relative velocity changes;   quaternion: only the body axis;
//   t0 / tf     zeros: their sweeps (con_aero.py:452-463 and twins) move only the Earth angle, which the air-relative velocity
//               does not depend on -- the rotation is applied and undone, and the NED axes at an inertial position do not move
//               with t -- so the reference's quotients are rounding noise around zero (<= 9e-5 beside position entries of 3e3).
// GEL_FLAG_FD_RECOMPUTE: every position sweep and the two t sweeps re-run the chain like the reference.  alpha is only
// evaluated by wavefronts that have an alpha or q-alpha row.  Consecutive lanes are consecutive nodes of a spec, so every
// store of a gradient block is one contiguous segment (which is what lets the B = 1 callback write straight into pinned
// host memory).  Four wavefronts (four tiles) per workgroup.
#ifndef GEL_AERO_UNROLL
#define GEL_AERO_UNROLL 1   // the position / velocity / quaternion sweeps of the aero rows as copies of their loop bodies (constant columns and
                            // directions): 164 VGPRs, -1.5 % launch time at B = 16384 (three alternating runs each)
#endif
constexpr int kAeroWaves = 4;
// aero_vair2(), aero_cos(), aero_acos(), aero_dalpha(): gel_rhs_parts.h (shared with the fused kernel's aero rows)
// park slots of one wavefront (doubles; the last six hold up to 12 ints per lane)
enum { AP_ILIM = 0, AP_AC = 3, AP_QC, AP_CC, AP_IS, AP_IND, AP_W = 8, AP_A0 = 11, AP_DIR = 14, AP_RHO = 17, AP_NV2 = 18, AP_INTS = 19,
       kAeroParkSlots = 25 };
typedef __attribute__((address_space(3))) int lds_int;
typedef __attribute__((address_space(3))) double lds_f64;
// ROLES (the B = 1 callback launch, where the length of one wavefront's chain is what counts): the four wavefronts of a
// workgroup share ONE tile -- wavefront 0 the centre values and the light sweeps, wavefronts 1..3 the centre and one position
// sweep each.  Same operations on the same operands per entry: bit-identical to the one-wavefront form.
// WIDE (the rows the fused kernel's lanes do not reach -- state node 0 of every phase, phases without aerodynamics -- of a launch
// whose outputs are the per-vector records of gel_eval_batch_aero_device): any number of vectors per wavefront (entry f of the
// (vector, node) sequence -> vector f / nnodes), outputs at 64-bit addresses b * ld + ... by plain global stores.  Same
// expressions per entry, same bits.
// SM (records only): SPEC-MAJOR part A (AeroOut::sm) -- an instantiation of its own, so that the ordinary launches carry neither
// its selects nor a third form of every store (as a run-time switch it cost aero_kernel 8 %)
template <bool ROLES, bool WIDE = false, bool SM = false>
__device__ __forceinline__ void aero_body(const ProblemDev P, int nnodes, const AeroNodeDev* __restrict__ nodes, int tiles, int B,
                                          const double* __restrict__ x, const AeroOut O, const unsigned vblk) {
  extern __shared__ double lds[];
  const Tables tb = stage_tables(P, lds);
  const int lane = (int)(threadIdx.x & 63);
  const int sw = ROLES ? (int)(threadIdx.x >> 6) : -1;   // ROLES: 0 centre + light sweeps, 1..3 position sweep sw - 1
  const long long wid = ROLES ? (long long)vblk : (long long)vblk * kAeroWaves + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  // FLAT (batch launches, tiles == 0) [r5]: a wavefront takes 64 consecutive entries of the (vector, node) sequence of the whole batch,
  // whichever vectors they belong to -- 325 constrained nodes per vector are 5.08 wavefronts' worth, not six tiles with the last one
  // 59 / 64 empty.  A wavefront then straddles at most two vectors (nnodes >= 64), so the vector is a per-lane value.
  const bool flat = !ROLES && !WIDE && tiles == 0;
  int b, ni_raw;
  bool live;
  if (WIDE) {
    const long long total = (long long)B * nnodes, f = wid * 64 + lane;
    if (wid * 64 >= total) return;                      // after the tables' barrier
    live = f < total;
    const long long fc = live ? f : total - 1;
    b = (int)(fc / nnodes);
    ni_raw = live ? (int)(fc - (long long)b * nnodes) : nnodes;
  } else if (flat) {
    const long long total = (long long)B * nnodes, f0 = wid * 64;
    if (f0 >= total) return;                           // after the tables' barrier
    const int b0 = (int)(f0 / nnodes), r0 = (int)(f0 - (long long)b0 * nnodes);
    const bool wrap = r0 + lane >= nnodes;
    live = f0 + lane < total;
    b = live ? (wrap ? b0 + 1 : b0) : b0;
    ni_raw = live ? (wrap ? r0 + lane - nnodes : r0 + lane) : nnodes;
  } else {
    if (wid >= (long long)B * tiles) return;           // after the tables' barrier
    b = (int)(wid / tiles);
    ni_raw = (int)(wid - (long long)b * tiles) * 64 + lane;
    live = ni_raw < nnodes;
  }
  const int ni = live ? ni_raw : nnodes - 1;
  const AeroNodeDev Nd = nodes[ni];
  const PhaseDev& ph = P.phases[Nd.phase];
  const double* xb = x + (size_t)b * P.nvars;
  const int M = P.M, N = P.N, xi = ph.xa + Nd.k;
  const double dx = P.dx;
  double re[3];
#pragma unroll
  for (int c = 0; c < 3; c++) re[c] = xb[M + 3 * xi + c];
  const bool want_jac = O.jac[0] || O.jac[1] || O.jac[2];
  // does any lane of this wavefront have an alpha or q-alpha row?  (wave-uniform)
  const bool need_alpha = __builtin_amdgcn_ballot_w64(live && ((O.con[0] && Nd.row[0] >= 0) || (O.con[2] && Nd.row[2] >= 0))) != 0;
  // The park: this wavefront's [slot][lane] region of LDS behind the tables.  What every store needs (1/limit, the centre's alpha,
  // q, cos, 1/sin, the node's rows) and what only the light sweeps need (centre wind, air velocity, body axis) wait there
  // instead of in registers across the position sweeps (234 -> 164 VGPRs).
  lds_f64* park = (lds_f64*)lds + ((table_doubles(P.Kw, P.Kc) + 1) & ~1) + (size_t)(threadIdx.x >> 6) * (kAeroParkSlots * 64) + lane;
  // ints (12 per lane): [kind] 8 nk; then per kind and block width w the BYTE offset 8 (w row0 + k) of this node's entries inside a
  // block of that width (or -1: no row of this kind / no gradient asked) -- alpha: w = 3, 4, 2 at 3, 4, 5; q: w = 3, 2 at 6, 7;
  // q-alpha: w = 3, 4, 2 at 8, 9, 10; [11] centre_ok.  An entry of block (bo, w), column col then sits at byte
  // 8 bo R + col (8 nk) + that offset of the kind's gradient vector: one 32-bit multiply-add per store, the rest on the scalar unit
  // (round 4 formed bo R + w row0 + col nk + k in 64-bit vector arithmetic per entry and kind: 369 of the 2,349 vector instructions
  // of a tile were integer address work)
  lds_int* ipark = (lds_int*)(park - lane + AP_INTS * 64) + lane;
#define AP_AIDX(kind, w) ((kind) == 0 ? ((w) == 3 ? 3 : ((w) == 4 ? 4 : 5)) : ((kind) == 1 ? ((w) == 3 ? 6 : 7) : ((w) == 3 ? 8 : ((w) == 4 ? 9 : 10))))
#define AP_GET(i) (park[(i) * 64])
#define AP_SET(i, v) (park[(i) * 64] = (v))
  // ---- centre (con_aero.py:39-87: scale, evaluate)
  PosCentre pc;
  PosPart pp;
  EarthAngle ea{1.0, 0.0, 1.0, 0.0};
  // Calm air [r5]: where both wind components of every lane are exactly zero (below and above the measured part of the wind table: in
  // the shipped example below 1 km and from 23 km up) the rotation of the zero vector into ECI is the zero vector -- wind_eci_or_calm()
  // skips its ~70 instructions, four times per tile -- and the Earth angle, which enters the air-relative velocity only through that
  // rotation (aero_vair2), is not formed at all (as in the fused kernel).  A sweep whose perturbed point leaves the calm (a step across
  // the table's end: the recomputing fallback) forms it on first need, from the node's time read again.
  bool have_ea = false;
#define GEL_AERO_NEED_EA(wn_, we_)                                                                                     \
  do {                                                                                                                 \
    if (!have_ea && __builtin_amdgcn_ballot_w64(!((wn_) == 0.0 && (we_) == 0.0)) != 0) {                                \
      const int kn_ = nodes[ni].k, phn_ = nodes[ni].phase;                                                             \
      const double to_ = xb[11 * M + 2 * N + phn_], tf_ = xb[11 * M + 2 * N + phn_ + 1];                               \
      const double tau_ = (kn_ == 0) ? 0.0 : P.tau[P.phases[phn_].toff + kn_ - 1];                                     \
      /* PSparams.time_nodes (SectionParameters.py:77-81): node 0 is t0 itself; t in seconds here (con_aero.py:45) */   \
      ea = earth_angle(((kn_ == 0) ? to_ : (tau_ * (tf_ - to_) / 2 + (tf_ + to_) / 2)) * P.ut);                         \
      have_ea = true;                                                                                                  \
    }                                                                                                                  \
  } while (0)
  double chk = 0.0;
  {
    double r[3], v[3], q[4], w[3], a0[3], dir[3];
#pragma unroll
    for (int c = 0; c < 3; c++) { r[c] = re[c] * P.up; v[c] = xb[4 * M + 3 * xi + c] * P.uv; }
#pragma unroll
    for (int c = 0; c < 4; c++) q[c] = xb[7 * M + 4 * xi + c];
    PosCentreTail pt;
    pp = pos_part<true, PosCentreSink, false>(r, tb, 0.0, nullptr, PosCentreSink{&pc}, &pt);
    pos_centre_tail(pt, pp.rho, pp.P, tb, pc, pp.wn, pp.we);
    GEL_AERO_NEED_EA(pp.wn, pp.we);
    wind_eci_or_calm(r, ea, pp.shp, pp.chp, pp.inv_p, pp.wn, pp.we, w);
    const double nv2 = aero_vair2(r, v, w, a0);
    thrust_dir(q, dir);
    const double ind = frsqrt(fmax(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2], 1.0e-300));
    const double cc = need_alpha ? aero_cos(a0, nv2, dir, ind) : 0.0;
    const double alpha_c = need_alpha ? aero_acos(cc, nv2) : 0.0;
    const double qdyn_c = 0.5 * pp.rho * nv2;                       // 0.5 rho |v_air|^2 (wrapper_utils.hpp:163-175)
    double sc, isc;
    fsqrt_rsqrt(fmax((1.0 - cc) * (1.0 + cc), 1.0e-300), sc, isc);  // sin(alpha_c) and its reciprocal
    const bool centre_ok = (cc <= 1.0) && (nv2 >= 1.0e-12) && (sc > 1.0e-6);
    // f / units[3] (con_aero.py:85-87) as f * (1 / units[3]) (one rounding apart from the division); con = 1 - f (:127-139)
#pragma unroll
    for (int kind = 0; kind < 3; kind++) {
      const double il = frcp(Nd.limit[kind]);
      AP_SET(AP_ILIM + kind, il * P.inv_dx);   // 1 / (limit dx): what every gradient entry of the kind is scaled by
      const int row = Nd.row[kind];
      {
        const bool has = live && O.jac[kind] && row >= 0;
        // FLAT: the vector's gradient values start b R (8 + nq) doubles into the kind's buffer (unsigned 32-bit byte offsets: the
        // launcher keeps to the per-vector mapping where a batch's gradients exceed 4 GB); -1 = no entry
        const unsigned vb = flat ? (unsigned)b * (unsigned)(O.ld ? (int)O.ld * 8 : O.nrows[kind] * (8 + ((kind == 1) ? 0 : 4)) * 8) : 0u;
        ipark[kind * 64] = 8 * Nd.nk[kind];
        ipark[AP_AIDX(kind, 3) * 64] = has ? (int)(vb + 8u * (unsigned)((SM ? 1 : 3) * Nd.row0[kind] + Nd.ko)) : -1;
        ipark[AP_AIDX(kind, 2) * 64] = has ? (int)(vb + 8u * (unsigned)((SM ? 1 : 2) * Nd.row0[kind] + Nd.ko)) : -1;
        if (kind != 1) ipark[AP_AIDX(kind, 4) * 64] = has ? (int)(vb + 8u * (unsigned)((SM ? 1 : 4) * Nd.row0[kind] + Nd.ko)) : -1;
      }
      if (!live || row < 0 || !O.con[kind] || (ROLES && sw != 0)) continue;
      const double cv = 1.0 - ((kind == 0) ? alpha_c : (kind == 1) ? qdyn_c : qdyn_c * alpha_c) * il;
      if (ROLES) __builtin_nontemporal_store(cv, O.con[kind] + (size_t)b * O.nrows[kind] + row);
      else if (WIDE) O.con[kind][(size_t)b * O.ld + row] = cv;
      else O.con[kind][(size_t)b * (O.ld ? (size_t)O.ld : (size_t)O.nrows[kind]) + row] = cv;
      chk += cv;
    }
    ipark[11 * 64] = centre_ok ? 1 : 0;
    AP_SET(AP_AC, alpha_c); AP_SET(AP_QC, qdyn_c); AP_SET(AP_CC, cc); AP_SET(AP_IS, isc); AP_SET(AP_IND, ind);
#pragma unroll
    for (int c = 0; c < 3; c++) { AP_SET(AP_W + c, w[c]); AP_SET(AP_A0 + c, a0[c]); AP_SET(AP_DIR + c, dir[c]); }
    AP_SET(AP_RHO, pp.rho); AP_SET(AP_NV2, nv2);
  }
  // One gradient entry of every kind that has this node, jac = -(f_p - f_c)/dx (con_aero.py:437-463), from the perturbed point's
  // air velocity a (squared norm nv2), body axis d (1/|d| = ind) and density: alpha_p - alpha_c in exact-difference form
  // (aero_dalpha; a lane it does not cover takes two acos like the reference -- per lane, so that a node's values do not depend on
  // which nodes share its wavefront: the flat mapping puts two vectors' nodes into one), q_p - q_c as it is, and
  //   alpha:   d f = t / limit        q:   d f = (q_p - q_c) / limit        q alpha:   d f = (q_p t + (q_p - q_c) alpha_c) / limit
  // (q_p alpha_p - q_c alpha_c, regrouped).  Block offset `boff` in units of R rows (0 position, 3 velocity, 6 quaternion, -1 = t:
  // 6 + nq), `width` columns per row, column `col`; zero: the entry is an exact zero.
#define GEL_AERO_EMIT(boff, width, col, a, nv2, d, ind, rho, skip_q, zero, skip_lane)                                      \
  do {                                                                                                            \
    double t_ = 0.0, dq_ = 0.0, qp_ = 0.0;                                                                        \
    const double ac_ = AP_GET(AP_AC);                                                                             \
    if (!(zero)) {                                                                                                \
      const double qc_ = AP_GET(AP_QC);                                                                           \
      qp_ = 0.5 * (rho) * (nv2);                                                                                  \
      dq_ = qp_ - qc_;                                                                                            \
      if (need_alpha) {                                                                                           \
        const double cp_ = aero_cos(a, nv2, d, ind);                                                              \
        const bool okd_ = aero_dalpha(cp_, nv2, AP_GET(AP_CC), AP_GET(AP_IS), ipark[11 * 64] != 0, t_);           \
        if (__builtin_amdgcn_ballot_w64(!okd_) != 0) {                                                            \
          const double t2_ = aero_acos(cp_, nv2) - ac_;                                                           \
          t_ = okd_ ? t_ : t2_;                                                                                   \
        }                                                                                                         \
      }                                                                                                           \
    }                                                                                                             \
    _Pragma("unroll") for (int kind = 0; kind < 3; kind++) {                                                      \
      if ((skip_q) && kind == 1) continue;                 /* dynamic pressure has no quaternion block */        \
      const int a8_ = ipark[AP_AIDX(kind, width) * 64];                                                           \
      if (a8_ == -1 || (skip_lane)) continue;                                                                     \
      if (SM && (zero)) continue;                          /* spec-major records do not hold the exact zeros of the t columns */ \
      const int nq = (kind == 1) ? 0 : 4;                                                                         \
      const int bo = ((boff) < 0) ? (6 + nq) : (boff);                                                            \
      const double df_ = (kind == 0) ? t_ : ((kind == 1) ? dq_ : qp_ * t_ + dq_ * ac_);                           \
      const double gv = (zero) ? 0.0 : -(df_ * AP_GET(AP_ILIM + kind));                                           \
      gel_au2 gd_;                                                                                                \
      __builtin_memcpy(&gd_, &gv, 8);                                                                             \
      if (WIDE) O.jac[kind][(size_t)b * O.ld + (size_t)bo * O.nrows[kind] + ((a8_ + (col) * ipark[kind * 64]) >> 3)] = gv;                     \
      else if (SM) __builtin_amdgcn_raw_buffer_store_b64(gd_, jrs[kind], a8_ + ((((boff) < 0) ? 11 : ((boff) == 0 ? 1 : ((boff) == 3 ? 4 : 7))) + (col)) * ipark[kind * 64], 0, 0); \
      else __builtin_amdgcn_raw_buffer_store_b64(gd_, jrs[kind], ((col) == 0) ? a8_ : a8_ + (col) * ipark[kind * 64], 8 * bo * O.nrows[kind], ROLES ? 2 : 0); /* ROLES = the one-vector callback: streamed to pinned host memory */ \
      chk += gv;                                                                                                  \
    }                                                                                                             \
  } while (0)
  // one buffer resource per kind: base = this vector's gradient values, [R][8 + nq] doubles (wave-uniform: scalar registers)
  typedef unsigned gel_au2 __attribute__((ext_vector_type(2)));
  __amdgpu_buffer_rsrc_t jrs[3];
#pragma unroll
  for (int kind = 0; kind < 3; kind++)
    jrs[kind] = __builtin_amdgcn_make_buffer_rsrc((O.jac[kind] && !WIDE) ? (void*)(O.jac[kind] + (flat ? (size_t)0 : (size_t)__builtin_amdgcn_readfirstlane(b) * (O.ld ? (size_t)O.ld : (size_t)O.nrows[kind] * (8 + ((kind == 1) ? 0 : 4))))) : (void*)nullptr,
                                                  0, -1, 0x00020000);
  if (want_jac) {
    // ---- t0 / tf columns
#pragma unroll 1
    for (int c = (ROLES && sw != 0) ? 2 : 0; c < 2; c++) {
      if (P.fd_recompute) {   // audit form: the knot times and the node's abscissa are read again (not carried in registers)
        const int kn = nodes[ni].k, phn = nodes[ni].phase;
        const double to2 = xb[11 * M + 2 * N + phn], tf2 = xb[11 * M + 2 * N + phn + 1];
        const double tau2 = (kn == 0) ? 0.0 : P.tau[P.phases[phn].toff + kn - 1];
        const double to_p = (c == 0) ? to2 + dx : to2, tf_p = (c == 1) ? tf2 + dx : tf2;
        const double tp = ((kn == 0) ? to_p : (tau2 * (tf_p - to_p) / 2 + (tf_p + to_p) / 2)) * P.ut;
        const EarthAngle eq = earth_angle(tp);
        const double r[3] = {fresh_product(re[0], P.up), fresh_product(re[1], P.up), fresh_product(re[2], P.up)};
        const double vq[3] = {xb[4 * M + 3 * xi] * P.uv, xb[4 * M + 3 * xi + 1] * P.uv, xb[4 * M + 3 * xi + 2] * P.uv};
        double wq[3], aq[3];
        wind_eci(r, eq, pp.shp, pp.chp, pp.inv_p, pp.wn, pp.we, wq);
        const double nvq = aero_vair2(r, vq, wq, aq);
        const double dq[3] = {AP_GET(AP_DIR), AP_GET(AP_DIR + 1), AP_GET(AP_DIR + 2)};
        GEL_AERO_EMIT(-1, 2, c, aq, nvq, dq, AP_GET(AP_IND), pp.rho, false, false, false);
      } else {
        const double none[3] = {0.0, 0.0, 0.0};
        GEL_AERO_EMIT(-1, 2, c, none, 0.0, none, 0.0, 0.0, false, true, false);
      }
    }
    // ---- position sweeps
#define GEL_AERO_POS_TAIL(c, rp, pq, skip_)                                                          \
  do {                                                                                               \
    double wq_[3], a_[3];                                                                            \
    GEL_AERO_NEED_EA((pq).wn, (pq).we);                                                              \
    wind_eci_or_calm(rp, ea, (pq).shp, (pq).chp, (pq).inv_p, (pq).wn, (pq).we, wq_);                 \
    const double vq_[3] = {xb[4 * M + 3 * xi] * P.uv, xb[4 * M + 3 * xi + 1] * P.uv, xb[4 * M + 3 * xi + 2] * P.uv}; \
    const double nv2_ = aero_vair2(rp, vq_, wq_, a_);                                                \
    const double dd_[3] = {AP_GET(AP_DIR), AP_GET(AP_DIR + 1), AP_GET(AP_DIR + 2)};                  \
    GEL_AERO_EMIT(0, 3, c, a_, nv2_, dd_, AP_GET(AP_IND), (pq).rho, false, false, skip_);                  \
  } while (0)
    const unsigned mine = ROLES ? ((sw == 0) ? 0u : (1u << (sw - 1))) : 7u;   // this wavefront's position sweeps
    unsigned todo = P.fd_recompute ? mine : 0u;   // sweeps with a lane the difference form does not cover (wave-uniform)
    unsigned bad = P.fd_recompute ? 7u : 0u;      // this lane's sweeps among them: only these take the recomputed values
    if (!P.fd_recompute) {
#if GEL_AERO_UNROLL
#pragma unroll
#else
#pragma unroll 1
#endif
      for (int c = 0; c < 3; c++) {
        if (!((mine >> c) & 1u)) continue;
        asm volatile("" ::: "memory");
        // the scaled position is formed where it is used (fresh_product: not carried across the trips), the velocity read again
        const double r[3] = {fresh_product(re[0], P.up), fresh_product(re[1], P.up), fresh_product(re[2], P.up)};
        double rp[3];
#pragma unroll
        for (int d = 0; d < 3; d++) rp[d] = (d == c) ? (re[d] + dx) * P.up : r[d];
        const double dlt = (c == 0) ? rp[0] - r[0] : ((c == 1) ? rp[1] - r[1] : rp[2] - r[2]);   // exact
        PosPart pq;
        const bool okp = pos_delta(r, c, dlt, pp, pc, tb, pq);
        if (__builtin_amdgcn_ballot_w64(!okp) != 0) { todo |= 1u << c; if (!okp) bad |= 1u << c; }
        GEL_AERO_POS_TAIL(c, rp, pq, !okp);
      }
    }
    if (todo) {   // the chain once more on the perturbed position (like the reference), kept by the lanes that need it
      asm volatile("" ::: "memory");
#pragma unroll 1
      for (int c = 0; c < 3; c++) {
        if (!((todo >> c) & 1u)) continue;
        double rp[3];
#pragma unroll
        for (int d = 0; d < 3; d++) rp[d] = ((d == c) ? xb[M + 3 * xi + d] + dx : xb[M + 3 * xi + d]) * P.up;
        const PosPart pf = pos_part<false, NoSink, false>(rp, tb, 0.0);
        GEL_AERO_POS_TAIL(c, rp, pf, !((bad >> c) & 1u));
      }
    }
#undef GEL_AERO_POS_TAIL
    asm volatile("" ::: "memory");   // the light sweeps read their inputs again
    // ---- velocity sweeps: only the air-relative velocity changes
#if GEL_AERO_UNROLL
#pragma unroll
#else
#pragma unroll 1
#endif
    for (int c = (ROLES && sw != 0) ? 3 : 0; c < 3; c++) {
      double vp[3], a[3];
      const double r[3] = {xb[M + 3 * xi] * P.up, xb[M + 3 * xi + 1] * P.up, xb[M + 3 * xi + 2] * P.up};
#pragma unroll
      for (int d = 0; d < 3; d++) vp[d] = ((d == c) ? xb[4 * M + 3 * xi + d] + dx : xb[4 * M + 3 * xi + d]) * P.uv;
      const double wc[3] = {AP_GET(AP_W), AP_GET(AP_W + 1), AP_GET(AP_W + 2)};
      const double dc[3] = {AP_GET(AP_DIR), AP_GET(AP_DIR + 1), AP_GET(AP_DIR + 2)};
      const double nv2 = aero_vair2(r, vp, wc, a);
      GEL_AERO_EMIT(3, 3, c, a, nv2, dc, AP_GET(AP_IND), AP_GET(AP_RHO), false, false, false);
    }
    // ---- quaternion sweeps: only the body axis changes
    if (need_alpha && !(ROLES && sw != 0)) {
#if GEL_AERO_UNROLL
#pragma unroll
#else
#pragma unroll 1
#endif
      for (int c = 0; c < 4; c++) {
        double qp[4], dp[3];
#pragma unroll
        for (int d = 0; d < 4; d++) qp[d] = (d == c) ? xb[7 * M + 4 * xi + d] + dx : xb[7 * M + 4 * xi + d];
        thrust_dir(qp, dp);
        const double indp = frsqrt(fmax(dp[0] * dp[0] + dp[1] * dp[1] + dp[2] * dp[2], 1.0e-300));
        const double ac[3] = {AP_GET(AP_A0), AP_GET(AP_A0 + 1), AP_GET(AP_A0 + 2)};
        GEL_AERO_EMIT(6, 4, c, ac, AP_GET(AP_NV2), dp, indp, AP_GET(AP_RHO), true, false, false);
      }
    }
  }
#undef GEL_AERO_EMIT
#undef GEL_AERO_NEED_EA
#undef AP_AIDX
#undef AP_GET
#undef AP_SET
  if (!(fabs(chk) <= 1.79769313486231570815e308)) *(volatile int32_t*)P.flag = 1;  // every writer stores the same 1
}

#ifndef GEL_AERO_MIN_WAVES
#define GEL_AERO_MIN_WAVES 3
#endif
// Workgroup p runs on XCD p % 8: in dispatch order every XCD works on every eighth run of entries, i.e. on every page of x and of the
// outputs that is in flight.  The workgroup index is permuted so that XCD j takes a contiguous eighth of the launch (as the fused
// kernel's vector groups, gel_eval_kernel.h GEL_XCD_RANGES); the workgroups behind the last multiple of eight keep their place.
#ifndef GEL_AERO_XCD_RANGES
#define GEL_AERO_XCD_RANGES 1
#endif
__device__ __forceinline__ unsigned aero_xcd_block(unsigned p, unsigned n) {
  const unsigned n8 = n >> 3;
  return (GEL_AERO_XCD_RANGES && p < 8u * n8) ? (p & 7u) * n8 + (p >> 3) : p;
}
__global__ __launch_bounds__(64 * kAeroWaves, GEL_AERO_MIN_WAVES) void aero_kernel(ProblemDev P, int nnodes, const AeroNodeDev* __restrict__ nodes,
                                                                int tiles, int B, const double* __restrict__ x, AeroOut O) {
  aero_body<false>(P, nnodes, nodes, tiles, B, x, O, aero_xcd_block(blockIdx.x, gridDim.x));
}
__global__ __launch_bounds__(64 * kAeroWaves, GEL_AERO_MIN_WAVES) void aero_sm_kernel(ProblemDev P, int nnodes, const AeroNodeDev* __restrict__ nodes,
                                                                   int tiles, int B, const double* __restrict__ x, AeroOut O) {
  aero_body<false, false, true>(P, nnodes, nodes, tiles, B, x, O, aero_xcd_block(blockIdx.x, gridDim.x));
}
__global__ __launch_bounds__(64 * kAeroWaves, 2) void aero_wide_kernel(ProblemDev P, int nnodes, const AeroNodeDev* __restrict__ nodes,
                                                                int B, const double* __restrict__ x, AeroOut O) {
  aero_body<false, true>(P, nnodes, nodes, 0, B, x, O, blockIdx.x);
}

// the listed nodes' rows into per-vector records: out.con / out.jac point at each kind's part of record 0, ld doubles per record
hipError_t launch_aero_wide(const ProblemDev& P, int nnodes, const AeroNodeDev* nodes, int B, const double* d_x,
                            const AeroLaunchOut& out, long long ld, hipStream_t s) {
  if (B <= 0 || nnodes <= 0) return hipSuccess;
  AeroOut O;
  for (int k = 0; k < 3; k++) { O.con[k] = out.con[k]; O.jac[k] = out.jac[k]; O.nrows[k] = out.nrows[k]; }
  O.ld = ld; O.sm = 0;
  const size_t lds = sizeof(double) * (((staged_table_doubles(P.Kw, P.Kc) + 1) & ~(size_t)1) + (size_t)kAeroWaves * kAeroParkSlots * 64);
  const long long waves = ((long long)B * nnodes + 63) / 64;
  hipLaunchKernelGGL(aero_wide_kernel, dim3((unsigned)((waves + kAeroWaves - 1) / kAeroWaves)), dim3(64 * kAeroWaves), lds, s, P, nnodes, nodes, B, d_x, O);
  return hipGetLastError();
}

// ld != 0: per-vector records (see AeroOut).  The flat mapping addresses a batch's gradient values with 32-bit byte offsets: a
// batch whose records span more is launched in runs of vectors that do not.
hipError_t launch_aero(const ProblemDev& P, int nnodes, const AeroNodeDev* nodes, int B, const double* d_x,
                       const AeroLaunchOut& out, hipStream_t s, long long ld, bool spec_major) {
  if (B <= 0 || nnodes <= 0) return hipSuccess;
  if (ld > 0 && B > 1 && nnodes >= 64 && (nnodes & 63) != 0) {
    const long long run = ((1LL << 32) - (1LL << 25)) / (ld * 8);
    if (run >= 1 && run < B) {
      for (long long b0 = 0; b0 < B; b0 += run) {
        AeroLaunchOut o = out;
        for (int k = 0; k < 3; k++) { if (o.con[k]) o.con[k] += b0 * ld; if (o.jac[k]) o.jac[k] += b0 * ld; }
        const hipError_t e = launch_aero(P, nnodes, nodes, (int)std::min<long long>(run, B - b0), d_x + b0 * P.nvars, o, s, ld, spec_major);
        if (e != hipSuccess) return e;
      }
      return hipSuccess;
    }
  }
  AeroOut O;
  for (int k = 0; k < 3; k++) { O.con[k] = out.con[k]; O.jac[k] = out.jac[k]; O.nrows[k] = out.nrows[k]; }
  O.ld = ld; O.sm = (ld > 0 && spec_major) ? 1 : 0;
  int tiles = (nnodes + 63) / 64;
  const size_t lds = sizeof(double) * (((staged_table_doubles(P.Kw, P.Kc) + 1) & ~(size_t)1) + (size_t)kAeroWaves * kAeroParkSlots * 64);
  long long waves = (long long)B * tiles;
  // flat (vector, node) mapping (tiles = 0 tells the kernel): no mostly-empty last tile per vector; needs every wavefront inside two
  // vectors and the batch's gradient values of a kind inside 32-bit byte offsets
  long long maxbytes = 0;
  for (int k = 0; k < 3; k++) maxbytes = std::max(maxbytes, (long long)B * (ld ? ld : (long long)O.nrows[k] * (8 + ((k == 1) ? 0 : 4))) * 8);
  if (B > 1 && nnodes >= 64 && (nnodes & 63) != 0 && maxbytes < (1LL << 32) - (1LL << 24)) {
    waves = ((long long)B * nnodes + 63) / 64;
    tiles = 0;
  }
  const unsigned grid = (unsigned)((waves + kAeroWaves - 1) / kAeroWaves);
  if (O.sm) hipLaunchKernelGGL(aero_sm_kernel, dim3(grid), dim3(64 * kAeroWaves), lds, s, P, nnodes, nodes, tiles, B, d_x, O);
  else hipLaunchKernelGGL(aero_kernel, dim3(grid), dim3(64 * kAeroWaves), lds, s, P, nnodes, nodes, tiles, B, d_x, O);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Knot / terminal / user rows.
//
// Linear rows: (coef0 x[idx0] + coef1 x[idx1]) + c0 -- equality_init, equality_time, inequality_time and
// equality_knot_LGR (lib/con_init_terminal_knot.py:39-52,118-141,174-252,422-436) are differences of single decision
// variables plus a constant; with coefficients +-1 the expression rounds exactly like the reference's
// (a - b) - c.  Their Jacobians are the coefficients (constant COO, laid out by the caller).
//
// Node-function rows: g = f(r, v) / p0 - p1 at ONE state node, r = position * unit, v = velocity * unit, and the forward
// difference (g(x + dx e_c) - g(x)) / dx over exactly the six columns the row can see (lib/con_init_terminal_knot.py:
// 378-405 perturbs the last node's six columns; lib/jac_fd.py:29-62 perturbs every column of x, and every column but
// these six returns an exact zero).  Eight lanes per row: lane 0 the centre, lanes 1..6 one perturbed state column each,
// lane 7 the knot time of rows that read it (the waypoint rows of lib/con_waypoint.py), formed in the kernel -- no
// perturbed copies of x exist anywhere.  Functions: see node_fn().
// ---------------------------------------------------------------------------
// Instantaneous impact point, FAA algorithm (lib/IIP.py:30-135 posLLH_IIP_FAA, fill_na = True, n_iter = 5): geodetic
// latitude and East longitude [deg] of the vacuum impact of (posECEF, velECEF); (0, 0) where the algorithm has no solution.
GEL_DEV void iip_faa(const double pe[3], const double ve[3], double& lat_deg, double& lon_deg) {
  const double a = 6378137.0, f = 1.0 / 298.257223563, b = a * (1.0 - f), e2 = 2.0 * f - f * f;
  lat_deg = 0.0; lon_deg = 0.0;
  double r_k1 = b;
  const double r0 = sqrt(pe[0] * pe[0] + pe[1] * pe[1] + pe[2] * pe[2]);
  if (r0 < r_k1) return;                                          // below the surface
  const double vi[3] = {ve[0] - kOmega * pe[1], ve[1] + kOmega * pe[0], ve[2]};   // + omega x r
  const double v0 = sqrt(vi[0] * vi[0] + vi[1] * vi[1] + vi[2] * vi[2]);
  const double eps_cos = (r0 * (v0 * v0) / kMu) - 1.0;
  if (eps_cos >= 1.0) return;                                     // not elliptical
  const double a_t = r0 / (1.0 - eps_cos);
  const double eps_sin = (pe[0] * vi[0] + pe[1] * vi[1] + pe[2] * vi[2]) / sqrt(kMu * a_t);
  const double eps2 = eps_cos * eps_cos + eps_sin * eps_sin;
  if (sqrt(eps2) <= 1.0 && a_t * (1.0 - sqrt(eps2)) - a >= 0.0) return;   // positive perigee height
  double Ek = 0.0, Fk = 0.0, Gk = 0.0, r_k2 = 0.0, r_prev = 0.0, eps_k_sin = 0.0, dcos = 0.0, dsin = 0.0;
  const double root = sqrt((a_t * a_t * a_t) / kMu);
  for (int it = 0; it < 5; it++) {
    const double eps_k_cos = (a_t - r_k1) / a_t;
    if (eps2 - eps_k_cos * eps_k_cos < 0.0) return;              // no intersection with the surface
    eps_k_sin = -sqrt(eps2 - eps_k_cos * eps_k_cos);
    dcos = (eps_k_cos * eps_cos + eps_k_sin * eps_sin) / eps2;
    dsin = (eps_k_sin * eps_cos - eps_k_cos * eps_sin) / eps2;
    const double fs = (dcos - eps_cos) / (1.0 - eps_cos);
    const double gs = (dsin + eps_sin - eps_k_sin) * root;
    Ek = fs * pe[0] + gs * vi[0]; Fk = fs * pe[1] + gs * vi[1]; Gk = fs * pe[2] + gs * vi[2];
    const double q = Gk / r_k1;
    r_k2 = a / sqrt((e2 / (1.0 - e2)) * (q * q) + 1.0);
    r_prev = r_k1;
    r_k1 = r_k2;
  }
  if (fabs(r_prev - r_k2) > 1.0) return;                          // not converged
  const double delta = atan2(dsin, dcos);
  const double time_sec = (delta + eps_sin - eps_k_sin) * root;
  const double phi = atan2(tan(asin(Gk / r_k2)), 1.0 - e2);
  const double lam = atan2(Fk, Ek) - kOmega * time_sec;
  lat_deg = phi * 180.0 / kPi;
  lon_deg = lam * 180.0 / kPi;
}

// Vincenty's inverse formula, lib/downrange.py:32-111 (iteration limit 5000, |d lambda| < 1e-12)
GEL_DEV double distance_vincenty(double lat_o, double lon_o, double lat_t, double lon_t) {
  const double Ra = 6378137.0, f = 1.0 / 298.257223563, Rb = Ra * (1.0 - f);
  const double lat1 = lat_o * kPi / 180.0, lon1 = lon_o * kPi / 180.0, lat2 = lat_t * kPi / 180.0, lon2 = lon_t * kPi / 180.0;
  if (lon2 - lon1 == 0.0) return 0.0;
  const double U1 = atan((1.0 - f) * tan(lat1)), U2 = atan((1.0 - f) * tan(lat2)), dl = lon2 - lon1;
  const double sU1 = sin(U1), cU1 = cos(U1), sU2 = sin(U2), cU2 = cos(U2);
  double lam = dl, sin_sigma = 0.0, cos_sigma = 0.0, sigma = 0.0, cos_alpha = 0.0, cos_2sm = 0.0;
  for (int it = 0; it < 5000; it++) {
    const double sl = sin(lam), cl = cos(lam);
    const double t1 = cU2 * sl, t2 = cU1 * sU2 - sU1 * cU2 * cl;
    sin_sigma = sqrt(t1 * t1 + t2 * t2);
    cos_sigma = sU1 * sU2 + cU1 * cU2 * cl;
    sigma = atan2(sin_sigma, cos_sigma);
    const double sin_alpha = cU1 * cU2 * sl / sin_sigma;
    cos_alpha = sqrt(1.0 - sin_alpha * sin_alpha);
    cos_2sm = cos_sigma - 2.0 * sU1 * sU2 / (cos_alpha * cos_alpha);
    const double coeff = f / 16.0 * (cos_alpha * cos_alpha) * (4.0 + f * (4.0 - 3.0 * (cos_alpha * cos_alpha)));
    const double prev = lam;
    lam = dl + (1.0 - coeff) * f * sin_alpha * (sigma + coeff * sin_sigma * (cos_2sm + coeff * cos_sigma * (-1.0 + 2.0 * cos_2sm)));
    if (fabs(lam - prev) < 1e-12) break;
  }
  const double u2 = (cos_alpha * cos_alpha) * (Ra * Ra - Rb * Rb) / (Rb * Rb);
  const double A = 1.0 + u2 / 16384.0 * (4096.0 + u2 * (-768.0 + u2 * (320.0 - 175.0 * u2)));
  const double Bc = u2 / 1024.0 * (256.0 + u2 * (-128.0 + u2 * (74.0 - 47.0 * u2)));
  const double ds = Bc * sin_sigma * (cos_2sm + 0.25 * Bc * (cos_sigma * (-1.0 + 2.0 * (cos_2sm * cos_2sm)) -
                    (1.0 / 6.0) * Bc * cos_2sm * (-3.0 + 4.0 * (sin_sigma * sin_sigma)) * (-3.0 + 4.0 * (cos_2sm * cos_2sm))));
  return Rb * A * (sigma - ds);
}

// functions of ONE knot state (r, v in SI units, t in seconds, row parameters p):
//   0 orbit energy (src/wrapper_coordinate.hpp:246-250)   1 |angular momentum| (:222-228)   2 inclination [rad] (:229-236)
//   3 a   4 e   5 a (1 - e)   6 a (1 + e)   (src/Coordinate.cpp:197-245)   7 |r|   8 |v|
//   9 / 10 / 11 geodetic latitude [deg] / longitude [deg] / altitude [m] of the ECEF position at time t
//       (eci2geodetic, lib/coordinate.py; lib/con_waypoint.py:507-560)
//   12 / 13 latitude / longitude [deg] of the instantaneous impact point (lib/con_waypoint.py:164-207, lib/IIP.py)
//   14 sine of the elevation above an antenna's horizon, p[2..4] = antenna ECEF, p[5..7] = its local vertical
//       (lib/con_waypoint.py:45-51)
//   15 downrange [m]: Vincenty distance from the launch point p[2] = latitude, p[3] = longitude [deg] to the geodetic
//       latitude / longitude of the position (lib/con_waypoint.py:531-534,590-598; lib/downrange.py:32-111)
GEL_DEV double node_fn(int fn, const double r[3], const double v[3], double t, const double* p) {
  if (fn >= 9) {
    double sn, cs;
    sincos(kOmega * t, &sn, &cs);
    const double pe[3] = {r[0] * cs + r[1] * sn, -r[0] * sn + r[1] * cs, r[2]};      // eci2ecef (src/Coordinate.cpp:51-59)
    if (fn <= 11) {
      double lat, lon, alt;
      geodetic_full(pe[0], pe[1], pe[2], lat, lon, alt);
      return (fn == 9) ? lat * (180.0 / kPi) : (fn == 10) ? lon * (180.0 / kPi) : alt;   // math.degrees
    }
    if (fn == 15) {
      double lat, lon, alt;
      geodetic_full(pe[0], pe[1], pe[2], lat, lon, alt);
      return distance_vincenty(p[2], p[3], lat * (180.0 / kPi), lon * (180.0 / kPi));
    }
    if (fn <= 13) {
      const double d0 = v[0] + kOmega * r[1], d1 = v[1] - kOmega * r[0];             // vel_eci2ecef (:69-73)
      const double ve[3] = {d0 * cs + d1 * sn, -d0 * sn + d1 * cs, v[2]};
      double la, lo;
      iip_faa(pe, ve, la, lo);
      return (fn == 12) ? la : lo;
    }
    const double d[3] = {pe[0] - p[2], pe[1] - p[3], pe[2] - p[4]};
    const double dn = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    return (d[0] / dn) * p[5] + (d[1] / dn) * p[6] + (d[2] / dn) * p[7];
  }
  const double rn = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
  const double vn = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
  if (fn == 0) return 0.5 * vn * vn - kMu / rn;
  if (fn == 7) return rn;
  if (fn == 8) return vn;
  const double c[3] = {r[1] * v[2] - r[2] * v[1], r[2] * v[0] - r[0] * v[2], r[0] * v[1] - r[1] * v[0]};  // r x v
  const double c2 = c[0] * c[0] + c[1] * c[1] + c[2] * c[2];
  if (fn == 1) return sqrt(c2);
  if (fn == 2) return acos(c[2] / sqrt(c2));
  // Laplace vector f = v x c - mu r/|r|
  const double f[3] = {v[1] * c[2] - v[2] * c[1] - kMu * (r[0] / rn), v[2] * c[0] - v[0] * c[2] - kMu * (r[1] / rn),
                       v[0] * c[1] - v[1] * c[0] - kMu * (r[2] / rn)};
  const double e = sqrt(f[0] * f[0] + f[1] * f[1] + f[2] * f[2]) / kMu;
  if (fn == 4) return e;
  const double a = (c2 / kMu) / (1.0 - e * e);
  if (fn == 3) return a;
  return (fn == 5) ? a * (1.0 - e) : a * (1.0 + e);
}

__device__ __forceinline__ void rows_body(const ProblemDev P, int nlin, const LinRowDev* __restrict__ lin, int nfn,
                                          const FnRowDev* __restrict__ fr, int B, int lin_blocks, const double* __restrict__ x,
                                          double* __restrict__ con, double* __restrict__ jfn, const unsigned vblk) {
  const int R = nlin + nfn;
  double chk = 0.0;
  if ((int)vblk < lin_blocks) {
    const long long t = (long long)vblk * blockDim.x + threadIdx.x;
    if (t >= (long long)B * nlin) return;
    const int b = (int)(t / nlin), r = (int)(t - (long long)b * nlin);
    const LinRowDev L = lin[r];
    const double* xb = x + (size_t)b * P.nvars;
    double s = L.coef0 * xb[L.idx0];
    if (L.idx1 >= 0) s += L.coef1 * xb[L.idx1];
    chk = con[(size_t)b * R + r] = s + L.c0;
  } else {
    const long long t = (long long)(vblk - lin_blocks) * blockDim.x + threadIdx.x;
    const long long grp = t >> 3;
    const int sw = (int)(t & 7);                       // 0 centre, 1..3 position xyz + dx, 4..6 velocity xyz + dx, 7 knot time + dx
    const bool live = grp < (long long)B * nfn;
    const long long g2 = live ? grp : 0;
    const int b = (int)(g2 / nfn), row = (int)(g2 - (long long)b * nfn);
    const FnRowDev F = fr[row];
    const double* xb = x + (size_t)b * P.nvars;
    double r[3], v[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
      // xdict[key][j] += dx, then * unit (con_init_terminal_knot.py:340-341,392-394; con_waypoint.py:219-232,565-577)
      const double pe = xb[P.M + 3 * F.node + c], ve = xb[4 * P.M + 3 * F.node + c];
      r[c] = ((sw == 1 + c) ? pe + P.dx : pe) * P.up;
      v[c] = ((sw == 4 + c) ? ve + P.dx : ve) * P.uv;
    }
    double tk = 0.0;
    if (F.tcol >= 0) {
      const double te = xb[11 * P.M + 2 * P.N + F.tcol];
      tk = ((sw == 7) ? te + P.dx : te) * P.ut;
    }
    const double f = node_fn(F.fn, r, v, tk, F.p);
    const int vm = F.mode & 3;
    double g = (vm == 0) ? f / F.p[0] - F.p[1] : (f - F.p[1]) / F.p[0];
    if (F.mode & 8) g = -g;
    const int lane0 = (int)(threadIdx.x & 63 & ~7);
    const double gc = __shfl(g, lane0, 64), fc = __shfl(f, lane0, 64);   // the centre values of this row's lane group
    if (!live) return;
    if (sw == 0) chk = con[(size_t)b * R + nlin + row] = g;
    else if (jfn) {
      double d;
      if (F.mode & 4) { d = ((f - fc) / P.dx) / F.p[0]; if (F.mode & 8) d = -d; }   // the reference scales the raw difference
      else d = (g - gc) / P.dx;                                                       // difference of the row's own value
      chk = jfn[((size_t)b * nfn + row) * 7 + (sw - 1)] = d;
    }
  }
  if (!(fabs(chk) <= 1.79769313486231570815e308)) *(volatile int32_t*)P.flag = 1;  // every writer stores the same 1
}

__global__ void rows_kernel(ProblemDev P, int nlin, const LinRowDev* __restrict__ lin, int nfn,
                            const FnRowDev* __restrict__ fr, int B, int lin_blocks, const double* __restrict__ x,
                            double* __restrict__ con, double* __restrict__ jfn) {
  rows_body(P, nlin, lin, nfn, fr, B, lin_blocks, x, con, jfn, blockIdx.x);
}

// ---------------------------------------------------------------------------
// ONE launch for one callback of the optimiser (Trajectory_Optimization.py:194-312 at B = 1): the four defect groups in the
// split latency form (its own wavefronts, one unit each), the aero path constraints and the row table, as workgroup
// ranges of the same grid -- nothing waits for anything, so the three run side by side instead of one launch after the
// other (three launches on one stream: 30 us of kernels back to back + two more launch latencies; three streams cost
// more than they saved: 94 us against 65).  Same device functions, same bits as the separate launches.
// 256 threads: the aero workgroup (one wavefront per sweep role) and four units of the split form.
// ---------------------------------------------------------------------------
struct CallbackArgs {
  int32_t nb_eval, nb_aero;
  int32_t nnodes, tiles;
  const AeroNodeDev* nodes;
  AeroOut O;
  int32_t nlin, nfn, lin_blocks;
  const LinRowDev* lin;
  const FnRowDev* fr;
  double* con;
  double* jfn;
};
static_assert(64 * kAeroWaves == 256, "callback_kernel's workgroup is the aero workgroup (four roles of one tile)");
template <bool JAC, bool MFMA>
__global__ __launch_bounds__(256) void callback_kernel(ProblemDev P, const double* __restrict__ x, double* __restrict__ res,
                                                       double* __restrict__ jvar, CallbackArgs A) {
  const unsigned b = blockIdx.x;
  if (b < (unsigned)A.nb_eval) eval_body<JAC, MFMA, true, false, 17>(P, 1, x, res, jvar, b);
  else if (b < (unsigned)(A.nb_eval + A.nb_aero)) aero_body<true>(P, A.nnodes, A.nodes, A.tiles, 1, x, A.O, b - (unsigned)A.nb_eval);
  else rows_body(P, A.nlin, A.lin, A.nfn, A.fr, 1, A.lin_blocks, x, A.con, A.jfn, b - (unsigned)(A.nb_eval + A.nb_aero));
  signal_done(P);
}

hipError_t launch_callback(const ProblemDev& P0, bool want_jac, const double* d_x, double* d_res, double* d_jvar,
                           int nnodes, const AeroNodeDev* nodes, const AeroLaunchOut* aero,
                           int nlin, const LinRowDev* lin, int nfn, const FnRowDev* fr, double* d_con, double* d_jfn, hipStream_t s) {
  ProblemDev P = P0;
  P.unit0 = 4 * P0.chunk0;
  P.nunits = 4 * P0.nchunks;                       // every unit of the problem, split form
  CallbackArgs A{};
  A.nb_eval = (P.nunits + 3) / 4;                  // four wavefronts = four units per workgroup
  if (aero && nnodes > 0) {
    for (int k = 0; k < 3; k++) { A.O.con[k] = aero->con[k]; A.O.jac[k] = aero->jac[k]; A.O.nrows[k] = aero->nrows[k]; }
    A.O.ld = 0; A.O.sm = 0;
    A.nnodes = nnodes; A.nodes = nodes; A.tiles = (nnodes + 63) / 64; A.nb_aero = A.tiles;   // ROLES form: one workgroup per tile
  }
  int rows_blocks = 0;
  if (d_con && nlin + nfn > 0) {
    A.nlin = nlin; A.nfn = nfn; A.lin = lin; A.fr = fr; A.con = d_con; A.jfn = d_jfn;
    A.lin_blocks = (nlin + 255) / 256;
    rows_blocks = A.lin_blocks + (nfn * 8 + 255) / 256;
  }
  const size_t lds_eval = sizeof(double) * ((size_t)P.park_off + (size_t)wave_lds_doubles(want_jac, P.use_mfma != 0, false, true) * 4);
  const size_t lds_aero = sizeof(double) * (((staged_table_doubles(P.Kw, P.Kc) + 1) & ~(size_t)1) + (size_t)kAeroWaves * kAeroParkSlots * 64);
  const size_t lds = lds_eval > lds_aero ? lds_eval : lds_aero;
  const dim3 grid((unsigned)(A.nb_eval + A.nb_aero + rows_blocks));
  if (P.done_flag) P.done_total = (int32_t)grid.x;
  if (want_jac) {
    if (P.use_mfma) hipLaunchKernelGGL((callback_kernel<true, true>), grid, dim3(256), lds, s, P, d_x, d_res, d_jvar, A);
    else hipLaunchKernelGGL((callback_kernel<true, false>), grid, dim3(256), lds, s, P, d_x, d_res, d_jvar, A);
  } else {
    if (P.use_mfma) hipLaunchKernelGGL((callback_kernel<false, true>), grid, dim3(256), lds, s, P, d_x, d_res, d_jvar, A);
    else hipLaunchKernelGGL((callback_kernel<false, false>), grid, dim3(256), lds, s, P, d_x, d_res, d_jvar, A);
  }
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Post-processing table (output_result.py:37-263): one thread per state node, kOutputCols values
// per node in the order of include/gelato_amd.h GEL_OUTPUT_*.  Runs once after the optimiser has finished: written for
// agreement with the reference's formulas (same operation order, libm calls where the reference has them), not speed.
// ---------------------------------------------------------------------------
struct Quat { double w, x, y, z; };
GEL_DEV Quat qmul(const Quat& q, const Quat& p) {   // lib/coordinate.py:31-37
  return Quat{q.w * p.w - q.x * p.x - q.y * p.y - q.z * p.z, q.x * p.w + q.w * p.x - q.z * p.y + q.y * p.z,
              q.y * p.w + q.z * p.x + q.w * p.y - q.x * p.z, q.z * p.w - q.y * p.x + q.x * p.y + q.w * p.z};
}
GEL_DEV Quat qconj(const Quat& q) { return Quat{q.w, -q.x, -q.y, -q.z}; }
GEL_DEV void qrot(const Quat& q, const double v[3], double o[3]) {   // conj(q) * v * q (:55-68)
  const Quat r = qmul(qconj(q), qmul(Quat{0.0, v[0], v[1], v[2]}, q));
  o[0] = r.x; o[1] = r.y; o[2] = r.z;
}
GEL_DEV double norm3(const double v[3]) { return sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }

// quat_ecef2nedg (lib/coordinate.py:335-359) from the geodetic latitude / longitude [rad]
GEL_DEV Quat quat_ecef2nedg(double lat, double lon) {
  double s_hl, c_hl, s_hp, c_hp;
  sincos(lon / 2.0, &s_hl, &c_hl);
  sincos(lat / 2.0, &s_hp, &c_hp);
  const double r2 = sqrt(2.0);
  return Quat{c_hl * (c_hp - s_hp) / r2, s_hl * (c_hp + s_hp) / r2, -c_hl * (c_hp + s_hp) / r2, s_hl * (c_hp - s_hp) / r2};
}

constexpr int kOutputCols = kOutputColumns;
__global__ void output_kernel(ProblemDev P, int M, const double* __restrict__ x, const double* __restrict__ tx,
                              const int32_t* __restrict__ node_sec, double lat0, double lon0, double* __restrict__ out) {
  extern __shared__ double lds[];
  const Tables tb = stage_tables(P, lds);
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M) return;
  const PhaseDev ph = P.phases[node_sec[i]];
  const double mass = x[i] * P.um;
  const double pos[3] = {x[M + 3 * i] * P.up, x[M + 3 * i + 1] * P.up, x[M + 3 * i + 2] * P.up};
  const double vel[3] = {x[4 * M + 3 * i] * P.uv, x[4 * M + 3 * i + 1] * P.uv, x[4 * M + 3 * i + 2] * P.uv};
  Quat q{x[7 * M + 4 * i], x[7 * M + 4 * i + 1], x[7 * M + 4 * i + 2], x[7 * M + 4 * i + 3]};
  {
    const double qn = sqrt(q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z);   // normalize(quat_[i]) (:129)
    q = Quat{q.w / qn, q.x / qn, q.y / qn, q.z / qn};
  }
  const double t = tx[i];
  double* o = out + (size_t)i * kOutputCols;
  double sn, cs;
  sincos(kOmega * t, &sn, &cs);
  // eci2ecef, vel_eci2ecef (src/Coordinate.cpp:51-73)
  const double pe[3] = {pos[0] * cs + pos[1] * sn, -pos[0] * sn + pos[1] * cs, pos[2]};
  const double g0 = vel[0] + kOmega * pos[1], g1 = vel[1] - kOmega * pos[0];
  const double ve[3] = {g0 * cs + g1 * sn, -g0 * sn + g1 * cs, vel[2]};
  double lat, lon, alt;
  geodetic_full(pe[0], pe[1], pe[2], lat, lon, alt);
  const double lat_d = lat * (180.0 / kPi), lon_d = lon * (180.0 / kPi);
  const double h = geopotential_altitude(alt);
  o[1] = lat_d; o[2] = lon_d; o[6] = alt;
  o[5] = distance_vincenty(lat0, lon0, lat_d, lon_d);
  // orbital elements (lib/coordinate.py:591-649)
  {
    const double rn = norm3(pos);
    const double nr[3] = {pos[0] / rn, pos[1] / rn, pos[2] / rn};
    const double c[3] = {pos[1] * vel[2] - pos[2] * vel[1], pos[2] * vel[0] - pos[0] * vel[2], pos[0] * vel[1] - pos[1] * vel[0]};
    const double f[3] = {vel[1] * c[2] - vel[2] * c[1] - kMu * nr[0], vel[2] * c[0] - vel[0] * c[2] - kMu * nr[1],
                         vel[0] * c[1] - vel[1] * c[0] - kMu * nr[2]};
    const double cn = norm3(c), fn = norm3(f);
    const double c1[3] = {c[0] / cn, c[1] / cn, c[2] / cn}, f1[3] = {f[0] / fn, f[1] / fn, f[2] / fn};
    const double inc = acos(c1[2]);
    double asc, argp;
    if (inc > 1e-10) {
      asc = atan2(c1[0], -c1[1]);
      argp = acos(cos(asc) * f1[0] + sin(asc) * f1[1]);
      if (f[2] < 0) argp *= -1.0;
    } else {
      asc = 0.0;
      argp = atan2(f[1], f[0]);
    }
    const double pp = cn * cn / kMu, e = fn / kMu, a = pp / (1.0 - e * e);
    double ta = acos(f1[0] * nr[0] + f1[1] * nr[1] + f1[2] * nr[2]);
    if (vel[0] * pos[0] + vel[1] * pos[1] + vel[2] * pos[2] < 0.0) ta = 2.0 * kPi - ta;
    if (asc < 0.0) asc += 2.0 * kPi;
    if (argp < 0.0) argp += 2.0 * kPi;
    if (ta < 0.0) ta += 2.0 * kPi;
    o[7] = a * (1.0 + e) - 6378137;
    o[8] = a * (1.0 - e) - 6378137;
    o[9] = inc * (180.0 / kPi); o[11] = asc * (180.0 / kPi); o[10] = argp * (180.0 / kPi); o[12] = ta * (180.0 / kPi);
  }
  // ground velocity in NED, inertial velocity in NED (output_result.py:166-186)
  const Quat q_e2n = quat_ecef2nedg(lat, lon);
  double sh, ch;
  sincos(kOmega * t / 2.0, &sh, &ch);
  const Quat q_i2n = qmul(Quat{ch, 0.0, 0.0, sh}, q_e2n);    // quat_eci2nedg (:386-397)
  double vg_ned[3], v_ned[3];
  qrot(q_e2n, ve, vg_ned);
  qrot(q_i2n, vel, v_ned);
  o[13] = vg_ned[0]; o[14] = vg_ned[1]; o[15] = vg_ned[2];
  o[26] = norm3(ve);
  o[22] = atan2(v_ned[1], v_ned[0]) * (180.0 / kPi);
  o[21] = asin(-v_ned[2] / norm3(v_ned)) * (180.0 / kPi);
  const Air air = atmosphere(h, tb.atm);
  double wn, we;
  wind_ned2(h, tb.wind, tb.winds, tb.Kw, wn, we);
  const double va_ned[3] = {vg_ned[0] - wn, vg_ned[1] - we, vg_ned[2] - 0.0};
  const double van = norm3(va_ned);
  const double qdyn = 0.5 * (van * van) * air.rho;
  o[31] = qdyn;
  // air velocity in ECI (:225-231) = ecef2eci(vel_ecef) - quatrot(quat_nedg2eci, wind)
  const double w_ned[3] = {wn, we, 0.0};
  double w_eci[3];
  qrot(qconj(q_i2n), w_ned, w_eci);
  const double va[3] = {(ve[0] * cs - ve[1] * sn) - w_eci[0], (ve[0] * sn + ve[1] * cs) - w_eci[1], ve[2] - w_eci[2]};
  const double vn = norm3(va);
  const double ux[3] = {1.0, 0.0, 0.0};
  double tdir[3];
  qrot(qconj(q), ux, tdir);
  o[23] = tdir[0]; o[24] = tdir[1]; o[25] = tdir[2];
  // angles of attack (lib/utils.py:92-161).  They take the altitude for the wind from the ECI position fed to
  // ecef2geodetic; a rotation about z leaves that altitude unchanged up to rounding, so the wind above is reused.
  {
    const double tn = norm3(tdir);
    const double ca_ = (va[0] / vn) * (tdir[0] / tn) + (va[1] / vn) * (tdir[1] / tn) + (va[2] / vn) * (tdir[2] / tn);
    const double a_all = (ca_ >= 1.0 || vn < 0.001) ? 0.0 : acos(ca_);
    const double a_deg = a_all * 180.0 / kPi;
    o[28] = a_deg;
    o[32] = a_deg * qdyn;
    double vb[3];
    qrot(q, va, vb);
    const bool none = vb[0] < 0.001;
    o[29] = none ? 0.0 : atan2(vb[2], vb[0]) * 180.0 / kPi;
    o[30] = none ? 0.0 : atan2(vb[1], vb[0]) * 180.0 / kPi;
  }
  // euler_from_quat(quat_nedg2body) (lib/coordinate.py:488-528)
  {
    const Quat qb = qmul(qconj(q_i2n), q);
    double az, el, ro;
    if (2.0 * (qb.w * qb.y - qb.z * qb.x) >= 1.0) { el = kPi / 2; az = 0.0; ro = 0.0; }
    else {
      az = atan2(2.0 * (qb.w * qb.z + qb.x * qb.y), 1.0 - 2.0 * (qb.y * qb.y + qb.z * qb.z));
      el = asin(2.0 * (qb.w * qb.y - qb.z * qb.x));
      ro = atan2(2.0 * (qb.w * qb.x + qb.y * qb.z), 1.0 - 2.0 * (qb.x * qb.x + qb.y * qb.y));
    }
    if (az < 0.0) az += 2.0 * kPi;
    o[18] = az * (180.0 / kPi); o[19] = el * (180.0 / kPi); o[20] = ro * (180.0 / kPi);
  }
  // Mach number, axial force and acceleration, thrust (:233-256)
  const double mach = vn / air.a;
  o[33] = mach;
  o[27] = vn;
  const double ca = interp_tab(mach, tb.ca, tb.cas, tb.Kc, 2, 1);
  const double kf = 0.5 * air.rho * vn;
  const double aero[3] = {kf * -va[0] * ph.area * ca, kf * -va[1] * ph.area * ca, kf * -va[2] * ph.area * ca};
  double aero_b[3];
  qrot(q, aero, aero_b);
  const double thrust = ph.thrust - ph.nozzle * air.P;
  o[0] = thrust;
  o[17] = aero_b[0];
  o[16] = (thrust + aero_b[0]) / mass;
  // impact point, NaN where there is none (posLLH_IIP_FAA(.., fill_na = False), :258-260)
  double la, lo;
  iip_faa(pe, ve, la, lo);
  const bool no_iip = (la == 0.0 && lo == 0.0);
  o[3] = no_iip ? __builtin_nan("") : la;
  o[4] = no_iip ? __builtin_nan("") : lo;
}

hipError_t launch_output(const ProblemDev& P, int M, const double* d_x, const double* d_tx, const int32_t* d_node_sec,
                         double lat0, double lon0, double* d_out, hipStream_t s) {
  hipLaunchKernelGGL(output_kernel, dim3((M + 63) / 64), dim3(64), sizeof(double) * staged_table_doubles(P.Kw, P.Kc), s, P, M, d_x, d_tx, d_node_sec,
                     lat0, lon0, d_out);
  return hipGetLastError();
}

hipError_t launch_rows(const ProblemDev& P, int nlin, const LinRowDev* lin, int nfn, const FnRowDev* fr, int B,
                       const double* d_x, double* d_con, double* d_jfn, hipStream_t s) {
  if (B <= 0 || nlin + nfn <= 0) return hipSuccess;
  const int lb = (int)(((long long)B * nlin + 255) / 256), fb = (int)(((long long)B * nfn * 8 + 255) / 256);
  hipLaunchKernelGGL(rows_kernel, dim3((unsigned)(lb + fb)), dim3(256), 0, s, P, nlin, lin, nfn, fr, B, lb, d_x, d_con, d_jfn);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------
// launches with at most this many wavefronts (after the x4) use the split latency form: one per SIMD
constexpr long long kSplitMaxWaves = 1024;

static size_t table_lds_bytes(int Kw, int Kc) { return sizeof(double) * staged_table_doubles(Kw, Kc); }

// which instantiation a launch takes (also reported through gel_launch_info)
EvalForm eval_form(const ProblemDev& P, int B, bool want_res, bool want_jac) {
  EvalForm f;
  const long long waves = (long long)B * P.nchunks;
  f.jac = want_jac;
  // D.X on the matrix pipe only when residual rows are requested at all and the problem asks for it
  f.mfma = P.use_mfma && want_res;
  // a handful of vectors cannot fill 1024 SIMDs: trade recomputation of the centre for a shorter serial chain;
  // P.nunits > 0: the caller asked for a range of units (unit-sharded launch)
  f.split = want_jac && (P.nunits > 0 || waves * 4 <= kSplitMaxWaves);
  // two decision vectors per wavefront: cooperative form (matrix pipe, not split) of a problem whose phases all fit 32 lanes
  f.pack = f.mfma && !f.split && P.pack;
  const int vw = f.pack ? 8 : 4;   // decision vectors per workgroup of the cooperative form
  f.waves = f.split ? (long long)B * (P.nunits > 0 ? P.nunits : 4 * P.nchunks)
                    : (f.mfma ? 4LL * P.nchunks * ((B + vw - 1) / vw) : waves);
  return f;
}

template <bool JAC, bool PACK, bool LONGP, bool NTS = true>
static void launch_coop(const ProblemDev& P, int B, const double* d_x, double* d_res, double* d_jvar, hipStream_t s) {
  const unsigned nb = (unsigned)((B + (PACK ? 7 : 3)) / (PACK ? 8 : 4));
  // vector-group major order deals the groups to the eight XCDs in blocks of eight (short last block: idle workgroups leave at once)
  const unsigned grid = P.vmajor ? (unsigned)P.nchunks * 8u * ((nb + 7u) / 8u) : (unsigned)P.nchunks * nb;
  const size_t lds = sizeof(double) * ((size_t)P.park_off + (size_t)wave_lds_doubles(JAC, true, PACK, false, LONGP) * (kBlock / 64));
  hipLaunchKernelGGL((eval_kernel<JAC, true, false, PACK, LONGP, NTS>), dim3(grid), dim3(kBlock), lds, s, P, B, d_x, d_res, d_jvar);
}

hipError_t launch_eval(const ProblemDev& P, int B, const double* d_x, double* d_res, double* d_jvar, hipStream_t s) {
  if (B <= 0) return hipSuccess;
  const EvalForm f = eval_form(P, B, d_res != nullptr, d_jvar != nullptr);
  if (f.mfma && !f.split) {
    // cooperative D.X form (matrix pipe, not split): one workgroup = one work item x four (PACK: eight) decision vectors
    // Jacobian values of a launch that fits the Infinity Cache (256 MB) and is read again at once (gel_eval_full_device: the update of
    // a full COO buffer) by ordinary stores: the reader finds them there (fused + update at B = 1024: 0.26 -> 0.165 ms).  A launch on
    // its own is better off streaming them past the caches even when it is small (B = 256 on repeat: ordinary stores +9 %)
    const bool small = P.cached_out && (double)B * 8.0 * ((double)P.V + 11.0 * P.N) <= 192.0e6;
    if (f.jac && f.pack) { if (small) launch_coop<true, true, false, false>(P, B, d_x, d_res, d_jvar, s); else launch_coop<true, true, false, true>(P, B, d_x, d_res, d_jvar, s); }
    // one Jacobian instantiation for long and short phases (without the slab loop it measured 0.4 % SLOWER at 6 x 64, with 52
    // fewer instructions per wavefront); the residual-only form of a problem without a long phase drops to 72 VGPRs without it
    // (7 waves/SIMD: -10 % launch time at 6 x 64)
    else if (f.jac) { if (small) launch_coop<true, false, true, false>(P, B, d_x, d_res, d_jvar, s); else launch_coop<true, false, true, true>(P, B, d_x, d_res, d_jvar, s); }
    else if (f.pack) launch_coop<false, true, false>(P, B, d_x, d_res, d_jvar, s);
    else if (P.longp) launch_coop<false, false, true>(P, B, d_x, d_res, d_jvar, s);
    else launch_coop<false, false, false>(P, B, d_x, d_res, d_jvar, s);
    return hipGetLastError();
  }
  const unsigned grid = (unsigned)((f.waves * 64 + kBlock - 1) / kBlock);
  // tables | one region per wavefront (the park, or only what a residual-only launch parks)
  const size_t lds = sizeof(double) * ((size_t)P.park_off + (size_t)wave_lds_doubles(f.jac, f.mfma, false) * (kBlock / 64));
  if (f.split) {
    ProblemDev Q = P;
    if (Q.nunits <= 0) { Q.unit0 = 4 * P.chunk0; Q.nunits = 4 * P.nchunks; }  // the whole list, split
    if (Q.done_flag) Q.done_total = (int32_t)grid;
    if (f.mfma)
      hipLaunchKernelGGL((eval_kernel<true, true, true>), dim3(grid), dim3(kBlock), lds, s, Q, B, d_x, d_res, d_jvar);
    else
      hipLaunchKernelGGL((eval_kernel<true, false, true>), dim3(grid), dim3(kBlock), lds, s, Q, B, d_x, d_res, d_jvar);
    return hipGetLastError();
  }
  if (f.jac)
    hipLaunchKernelGGL((eval_kernel<true, false>), dim3(grid), dim3(kBlock), lds, s, P, B, d_x, d_res, d_jvar);
  else
    hipLaunchKernelGGL((eval_kernel<false, false>), dim3(grid), dim3(kBlock), lds, s, P, B, d_x, d_res, d_jvar);
  return hipGetLastError();
}

hipError_t launch_expand(long long nnz, long long V, int B, const double* cval, const int32_t* src,
                         const double* d_jvar, double* d_full, hipStream_t s) {
  if (B <= 0) return hipSuccess;
  long long pairs = (nnz + 1) / 2;
  unsigned gx = (unsigned)((pairs + kBlock - 1) / kBlock);
  if (gx > 4096) gx = 4096;
  const int per_launch = 65535 * kExpandGroup;  // grid.y limit
  for (int b0 = 0; b0 < B; b0 += per_launch) {
    const int nb = (B - b0 < per_launch) ? (B - b0) : per_launch;
    hipLaunchKernelGGL(expand_kernel, dim3(gx, (nb + kExpandGroup - 1) / kExpandGroup), dim3(kBlock), 0, s, nnz, V, nb,
                       cval, src, d_jvar + (size_t)b0 * V, d_full + (size_t)b0 * nnz);
  }
  return hipGetLastError();
}

hipError_t launch_update_full(long long nnz, long long V, int nvar, int B, const int32_t* vdst, const int32_t* vsrc,
                              int nlines, const int32_t* vline, const int32_t* src, const double* cval,
                              const double* d_jvar, double* d_full, hipStream_t s) {
  if (B <= 0 || nvar <= 0) return hipSuccess;
  if (nlines > 0 && (nnz & 7) == 0 && ((uintptr_t)d_full & 63) == 0) {
    const unsigned gx = (unsigned)((8LL * nlines + kBlock - 1) / kBlock);
    const unsigned gy = (unsigned)std::min<long long>(B, std::max<long long>(1, 16384 / gx));
    hipLaunchKernelGGL(update_lines_kernel, dim3(gx, gy), dim3(kBlock), 0, s, nnz, V, nlines, B, vline, src, cval, d_jvar, d_full);
    return hipGetLastError();
  }
  const unsigned gx = (unsigned)((nvar + kBlock - 1) / kBlock);
  // enough workgroups to fill the chip whatever the batch; every workgroup strides over the vectors
  const unsigned gy = (unsigned)std::min<long long>(B, std::max<long long>(1, 8192 / gx));
  hipLaunchKernelGGL(update_full_kernel, dim3(gx, gy), dim3(kBlock), 0, s, nnz, V, nvar, B, vdst, vsrc, d_jvar, d_full);
  return hipGetLastError();
}
hipError_t launch_fill_full(long long nnz, int B, const double* cval, double* d_full, hipStream_t s) {
  if (B <= 0) return hipSuccess;
  unsigned gx = (unsigned)((nnz + kBlock - 1) / kBlock);
  if (gx > 2048) gx = 2048;
  hipLaunchKernelGGL(fill_full_kernel, dim3(gx, (unsigned)std::min(B, 16)), dim3(kBlock), 0, s, nnz, B, cval, d_full);
  return hipGetLastError();
}

// pos[i] = rank * width + offset of output entry i (res entries first, then compact Jacobian values): entry i of vector b
// sits at out[(rank * B + b) * width + offset]
__global__ __launch_bounds__(kBlock) void shard_unpack_kernel(long long nres, long long V, long long width, int B,
                                                              const int64_t* __restrict__ pos, const double* __restrict__ out,
                                                              double* __restrict__ res, double* __restrict__ jvar) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nres + V) return;
  if ((i < nres) ? (res == nullptr) : (jvar == nullptr)) return;
  const long long pz = pos[i], r = pz / width, off = pz - r * width;
  for (int b = blockIdx.y; b < B; b += gridDim.y) {
    const double v = out[((size_t)r * B + b) * width + off];
    if (i < nres) res[(size_t)b * nres + i] = v;
    else jvar[(size_t)b * V + (i - nres)] = v;
  }
}

hipError_t launch_shard_unpack(long long nres, long long V, long long width, int B, const int64_t* pos, const double* out,
                               double* d_res, double* d_jvar, hipStream_t s) {
  const unsigned gx = (unsigned)((nres + V + kBlock - 1) / kBlock);
  const unsigned gy = (unsigned)std::min(B, 4096);
  hipLaunchKernelGGL(shard_unpack_kernel, dim3(gx, gy), dim3(kBlock), 0, s, nres, V, width, B, pos, out, d_res, d_jvar);
  return hipGetLastError();
}

hipError_t launch_perturb_local(int nloc, double dx, const double* d_x, const int32_t* d_colmap, double* d_Xp, hipStream_t s) {
  const long long tot = (long long)(nloc + 1) * nloc;
  hipLaunchKernelGGL(perturb_local_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, nloc, dx, d_x, d_colmap, d_Xp);
  return hipGetLastError();
}

hipError_t launch_quotient_local(int nloc, int nres, int roff, int nrows, double dx, const double* d_res, double* d_J,
                                 long long ldJ, int row0, const int32_t* d_colmap, hipStream_t s) {
  dim3 grid((nloc + 31) / 32, (nrows + 31) / 32);
  hipLaunchKernelGGL(quotient_local_kernel, grid, dim3(32, 8), 0, s, nloc, nres, roff, nrows, dx, d_res, d_J, ldJ, row0, d_colmap);
  return hipGetLastError();
}

hipError_t launch_rhs_vel(bool air, int n, const double* mass_e, const double* pos_e, const double* vel_e,
                          const double* quat, const double* t, const double* tables, int Kw, int Kc, double thrust,
                          double area, double nozzle, double um, double up, double uv, double barC20, double* out,
                          hipStream_t s) {
  RhsArgs A{n, Kw, Kc, mass_e, pos_e, vel_e, quat, t, tables, thrust, area, nozzle, um, up, uv, barC20, out};
  const unsigned grid = (n + 63) / 64;
  if (air)
    hipLaunchKernelGGL(rhs_vel_air_kernel, dim3(grid), dim3(64), table_lds_bytes(Kw, Kc), s, A);
  else
    hipLaunchKernelGGL(rhs_vel_noair_kernel, dim3(grid), dim3(64), 0, s, A);
  return hipGetLastError();
}

hipError_t launch_rhs_quat(int n, const double* quat, const double* u_e, double unit_u, double* out, hipStream_t s) {
  hipLaunchKernelGGL(rhs_quat_kernel, dim3((n + 63) / 64), dim3(64), 0, s, n, quat, u_e, unit_u, out);
  return hipGetLastError();
}

hipError_t launch_point(int kind, int n, const double* in, const double* aux, int aux_rows, double* out,
                        hipStream_t s) {
  const size_t lds = sizeof(double) * (size_t)(kAtmDoubles + 5 * (aux_rows > 0 ? aux_rows : 0));
  hipLaunchKernelGGL(point_kernel, dim3((n + 63) / 64), dim3(64), lds, s, kind, n, in, aux, aux_rows, out);
  return hipGetLastError();
}

}  // namespace gel

