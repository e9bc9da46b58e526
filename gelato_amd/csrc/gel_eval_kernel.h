// gel_eval_kernel.h -- the fused evaluation kernel (included by gel_kernels.hip).
//
// One wavefront = one (decision vector b, phase, 64-node chunk) work item; one
// lane = one collocation node.  The phase is wave-uniform: its parameters sit in
// SGPRs, every phase-type branch (air / NoAir, hold, engine off) is a scalar
// branch, and the X rows of the D.X product are scalar (broadcast) loads.
//
// Register discipline (fp64 = 2 VGPRs per value, 64-wide): the heavy chain
// (geodetic -> atmosphere -> wind -> Earth-angle quaternion -> aero) has ONE
// instance, in a loop whose LAST trip is the centre point and leaves through
// `break`; nothing the later sweeps need is carried across a heavy trip.  Inputs
// that are only needed again late (quaternion, velocity, mass) are re-read from
// L1/L2 through a laundered pointer instead of being kept live.
#pragma once

namespace gel {

// make the compiler forget what it knows about a (wave-uniform) pointer, so a
// later load through it is a real (cached) load, not a value kept in registers
template <class T>
GEL_DEV const T* relaunder(const T* p) {
  asm volatile("" : "+s"(p));
  return p;
}

template <bool JAC>
__global__ __launch_bounds__(kBlock) void eval_kernel(ProblemDev P, int B, const double* __restrict__ x,
                                                      double* __restrict__ res, double* __restrict__ jvar) {
  extern __shared__ double lds[];
  const Tables tb = stage_tables(P, lds);

  const int lane = threadIdx.x & 63;
  const long long item = __builtin_amdgcn_readfirstlane((int)(((long long)blockIdx.x * kBlock + threadIdx.x) >> 6));
  if (item >= (long long)B * P.nchunks) return;
  const int b = (int)(item / P.nchunks);
  const int2 ck = P.chunks[(int)(item - (long long)b * P.nchunks)];
  const int sec = __builtin_amdgcn_readfirstlane(ck.x);
  const int j = __builtin_amdgcn_readfirstlane(ck.y) + lane;  // node inside the phase
  const PhaseDev& ph = P.phases[sec];
  const int n = ph.n;
  if (j >= n) return;
  const int g = ph.ua + j;       // global collocation node
  const int xj = ph.xa + 1 + j;  // its state row (x-node j+1)
  const int M = P.M, N = P.N;

  const double* xb = x + (size_t)b * P.nvars;
  const double* xt = xb + 11 * M + 2 * N;
  const double to = xt[sec], tf = xt[sec + 1];
  const double dx = P.dx, ut = P.ut;
  double chk = 0.0;  // running sum of everything written: NaN/Inf detector

  double* rb = res ? res + (size_t)b * 11 * N : nullptr;
  double* jb = JAC ? jvar + (size_t)b * P.V + ph.voff + j : nullptr;
#define EMIT(slot, val)                  \
  do {                                   \
    const double _v = (val);             \
    jb[(size_t)(slot) * n] = _v;         \
    chk += _v;                           \
  } while (0)
  // Jacobian entry from a perturbed/centre pair: -(f_p - f_c)/dx*(tf-to)*unit_t/2  (con_dynamics.py:372)
#define FDQ(fp, fc) (-((fp) - (fc)) / dx * (tf - to) * ut / 2.0)

  // ---------------- velocity RHS + FD Jacobian (lib/con_dynamics.py:216-496) ----------------
  double fc[3];
  {
    const double* xm = xb;
    const double* xr = xb + M;
    const double* xv = xb + 4 * M;
    const double* xq = xb + 7 * M;
    const double me = xm[xj];
    const double re[3] = {xr[3 * xj], xr[3 * xj + 1], xr[3 * xj + 2]};
    const double tau = P.tau[ph.toff + j];
    const double tn = tau * (tf - to) / 2 + (tf + to) / 2;  // PSparams.time_nodes, SectionParameters.py:77-81
    const double m = me * P.um;
    double dir[3];
    {
      const double q[4] = {xq[4 * xj], xq[4 * xj + 1], xq[4 * xj + 2], xq[4 * xj + 3]};
      thrust_dir(q, dir);
    }

    if (ph.air) {
      const double v[3] = {xv[3 * xj] * P.uv, xv[3 * xj + 1] * P.uv, xv[3 * xj + 2] * P.uv};
      // Trips k = 0,1,2: position sweeps (pos_k + dx); trip k = 3: centre, leaves by break.
      double fp[3][3];
      PosPart pp;
      TimePart tp;
      double F[3], T;
#pragma unroll 1
      for (int k = JAC ? 0 : 3;; k++) {
        double r[3];
#pragma unroll
        for (int c = 0; c < 3; c++) r[c] = ((k == c) ? (re[c] + dx) : re[c]) * P.up;
        pp = pos_part(r, tb, P.barC20);
        tp = time_part(r, tn, pp.wn, pp.we);
        aero_force(r, v, pp, tp, ph.area, tb, F);
        T = ph.thrust - ph.nozzle * pp.P;
        const double Td[3] = {T * dir[0], T * dir[1], T * dir[2]};
        accel(Td, F, m, pp.g, P.uv, fc);
        if (k == 3) break;
#pragma unroll
        for (int kk = 0; kk < 3; kk++)
          if (k == kk) {
#pragma unroll
            for (int c = 0; c < 3; c++) fp[kk][c] = fc[c];
          }
      }
      // here pp, tp, F, T, fc are the centre values
      if (JAC) {
#pragma unroll
        for (int k = 0; k < 3; k++)
#pragma unroll
          for (int c = 0; c < 3; c++) EMIT(12 + 3 * k + c, FDQ(fp[k][c], fc[c]));

        const double r[3] = {re[0] * P.up, re[1] * P.up, re[2] * P.up};
        double f[3];
        // velocity sweeps: only the aerodynamic force changes
        if (ph.air_fd) {
          const double djj = P.Dt[ph.doff + (size_t)(j + 1) * n + j];  // D[j][j+1]
          const double Tdc[3] = {T * dir[0], T * dir[1], T * dir[2]};
#pragma unroll 1
          for (int k = 0; k < 3; k++) {
            double vp[3], Fp[3];
#pragma unroll
            for (int c = 0; c < 3; c++) vp[c] = ((k == c) ? (xv[3 * xj + c] + dx) : xv[3 * xj + c]) * P.uv;
            aero_force(r, vp, pp, tp, ph.area, tb, Fp);
            accel(Tdc, Fp, m, pp.g, P.uv, f);
            // submat_vel[3j+c, 3(j+1)+k] = D[j][j+1]*(c==k) + rh_vel   (con_dynamics.py:341-343,415-416)
#pragma unroll
            for (int c = 0; c < 3; c++) EMIT(ph.s_vv + 3 * k + c, ((c == k) ? djj : 0.0) + FDQ(f[c], fc[c]));
          }
        }
        // quaternion sweeps: only the thrust direction changes
        {
          const double* xq2 = relaunder(xq);
          const double q[4] = {xq2[4 * xj], xq2[4 * xj + 1], xq2[4 * xj + 2], xq2[4 * xj + 3]};
#pragma unroll 1
          for (int k = 0; k < 4; k++) {
            double qp[4];
#pragma unroll
            for (int c = 0; c < 4; c++) qp[c] = (k == c) ? (q[c] + dx) : q[c];
            double dp[3];
            thrust_dir(qp, dp);
            const double Td[3] = {T * dp[0], T * dp[1], T * dp[2]};
            accel(Td, F, m, pp.g, P.uv, f);
#pragma unroll
            for (int c = 0; c < 3; c++) EMIT(ph.s_vq + 3 * k + c, FDQ(f[c], fc[c]));
          }
        }
        const double Tdc[3] = {T * dir[0], T * dir[1], T * dir[2]};
        // mass sweep: only the division by mass changes
        accel(Tdc, F, (me + dx) * P.um, pp.g, P.uv, f);
#pragma unroll
        for (int c = 0; c < 3; c++) EMIT(9 + c, FDQ(f[c], fc[c]));
        // t0 / tf sweeps (con_dynamics.py:452-480): only the Earth angle changes
        if (ph.air_fd) {
#pragma unroll 1
          for (int k = 0; k < 2; k++) {
            const double to_p = (k == 0) ? to + dx : to;
            const double tf_p = (k == 1) ? tf + dx : tf;
            const double tnp = tau * (tf_p - to_p) / 2 + (tf_p + to_p) / 2;
            const TimePart tq = time_part(r, tnp, pp.wn, pp.we);
            double Fp[3];
            aero_force(r, v, pp, tq, ph.area, tb, Fp);
            accel(Tdc, Fp, m, pp.g, P.uv, f);
#pragma unroll
            for (int c = 0; c < 3; c++)
              EMIT(ph.s_vt + 3 * k + c, -(f[c] * (tf_p - to_p) - fc[c] * (tf - to)) / dx * ut / 2.0);
          }
        } else {
#pragma unroll
          for (int c = 0; c < 3; c++) {
            const double rh_to = fc[c] * ut / 2.0;
            EMIT(ph.s_vt + c, rh_to);
            EMIT(ph.s_vt + 3 + c, -rh_to);
          }
        }
      }
    } else {
      // NoAir (reference_area == 0): thrust + gravity only (src/pybind_dynamics.cpp:73-92)
      const double T = ph.thrust;
      const double Td[3] = {T * dir[0], T * dir[1], T * dir[2]};
      double gc[3];
      {
        const double r[3] = {re[0] * P.up, re[1] * P.up, re[2] * P.up};
        gravity_eci(r, P.barC20, gc);
      }
      accel_noair(Td, m, gc, P.uv, fc);
      if (JAC) {
        double f[3];
        accel_noair(Td, (me + dx) * P.um, gc, P.uv, f);
#pragma unroll
        for (int c = 0; c < 3; c++) EMIT(9 + c, FDQ(f[c], fc[c]));
#pragma unroll 1
        for (int k = 0; k < 3; k++) {
          double r[3], gp[3];
#pragma unroll
          for (int c = 0; c < 3; c++) r[c] = ((k == c) ? (re[c] + dx) : re[c]) * P.up;
          gravity_eci(r, P.barC20, gp);
          accel_noair(Td, m, gp, P.uv, f);
#pragma unroll
          for (int c = 0; c < 3; c++) EMIT(12 + 3 * k + c, FDQ(f[c], fc[c]));
        }
        const double q[4] = {xq[4 * xj], xq[4 * xj + 1], xq[4 * xj + 2], xq[4 * xj + 3]};
#pragma unroll 1
        for (int k = 0; k < 4; k++) {
          double qp[4];
#pragma unroll
          for (int c = 0; c < 4; c++) qp[c] = (k == c) ? (q[c] + dx) : q[c];
          double dp[3];
          thrust_dir(qp, dp);
          const double Tp[3] = {T * dp[0], T * dp[1], T * dp[2]};
          accel_noair(Tp, m, gc, P.uv, f);
#pragma unroll
          for (int c = 0; c < 3; c++) EMIT(ph.s_vq + 3 * k + c, FDQ(f[c], fc[c]));
        }
#pragma unroll
        for (int c = 0; c < 3; c++) {
          const double rh_to = fc[c] * ut / 2.0;
          EMIT(ph.s_vt + c, rh_to);
          EMIT(ph.s_vt + 3 + c, -rh_to);
        }
      }
    }
  }

  // from here on the node state is re-read (L1/L2 hits) rather than kept live above
  const double* xb2 = relaunder(xb);
  const double* xm = xb2;
  const double* xr = xb2 + M;
  const double* xv = xb2 + 4 * M;
  const double* xq = xb2 + 7 * M;
  const double* xu = xb2 + 11 * M;
  const double ve[3] = {xv[3 * xj], xv[3 * xj + 1], xv[3 * xj + 2]};
  const double q[4] = {xq[4 * xj], xq[4 * xj + 1], xq[4 * xj + 2], xq[4 * xj + 3]};

  // ---------------- position / quaternion Jacobian entries (:155-213, :536-632) ----------------
  if (JAC) {
    const double rh_vel = -P.uv * (tf - to) * ut / 2.0 / P.up;
#pragma unroll
    for (int c = 0; c < 3; c++) {
      EMIT(0 + c, rh_vel);
      const double rh_to = ve[c] * P.uv * ut / 2.0 / P.up;
      EMIT(3 + c, rh_to);
      EMIT(6 + c, -rh_to);
    }
    if (!ph.hold) {
      const double u0 = xu[2 * g], u1 = xu[2 * g + 1];
      double fq[4], f[4];
      quat_rate(q, u0, u1, P.uu, fq);
      const double djj = P.Dt[ph.doff + (size_t)(j + 1) * n + j];
#pragma unroll 1
      for (int k = 0; k < 4; k++) {
        double qp[4];
#pragma unroll
        for (int c = 0; c < 4; c++) qp[c] = (k == c) ? (q[c] + dx) : q[c];
        quat_rate(qp, u0, u1, P.uu, f);
        // submat_quat[4j+c, 4(j+1)+k] = D[j][j+1]*(c==k) + rh_quat   (con_dynamics.py:575-589)
#pragma unroll
        for (int c = 0; c < 4; c++) EMIT(ph.s_qq + 4 * k + c, ((c == k) ? djj : 0.0) + FDQ(f[c], fq[c]));
      }
#pragma unroll 1
      for (int k = 0; k < 2; k++) {
        quat_rate(q, (k == 0) ? u0 + dx : u0, (k == 1) ? u1 + dx : u1, P.uu, f);
#pragma unroll
        for (int c = 0; c < 4; c++) EMIT(ph.s_qq + 16 + 4 * k + c, FDQ(f[c], fq[c]));
      }
#pragma unroll
      for (int c = 0; c < 4; c++) {
        const double rh_to = fq[c] * ut / 2.0;
        EMIT(ph.s_qq + 24 + c, rh_to);
        EMIT(ph.s_qq + 28 + c, -rh_to);
      }
    }
  }

  // ---------------- D.X rows and the four defect residuals (:34-63,116-152,216-289,499-533) ----------------
  if (rb) {
    double lm = 0.0, lr[3] = {0, 0, 0}, lv[3] = {0, 0, 0}, lq[4] = {0, 0, 0, 0};
    {
      const double* Dt = P.Dt + ph.doff + j;  // lane j: consecutive addresses
      const double* pm = xm + ph.xa;           // wave-uniform rows: scalar loads
      const double* pr = xr + 3 * ph.xa;
      const double* pv = xv + 3 * ph.xa;
      const double* pq = xq + 4 * ph.xa;
      for (int i = 0; i <= n; i++) {
        const double d = Dt[(size_t)i * n];
        lm += d * pm[i];
#pragma unroll
        for (int c = 0; c < 3; c++) lr[c] += d * pr[3 * i + c];
#pragma unroll
        for (int c = 0; c < 3; c++) lv[c] += d * pv[3 * i + c];
#pragma unroll
        for (int c = 0; c < 4; c++) lq[c] += d * pq[4 * i + c];
      }
    }
    double cm;
    if (ph.engine_on) {
      const double rh = -ph.massflow / P.um * (tf - to) * ut / 2.0;
      cm = lm - rh;
    } else {
      cm = xm[xj] - xm[ph.xa];
    }
    rb[g] = cm;
    chk += cm;
#pragma unroll
    for (int c = 0; c < 3; c++) {
      const double rh = ve[c] * P.uv * (tf - to) * ut / 2.0 / P.up;
      const double cp = lr[c] - rh;
      rb[N + 3 * g + c] = cp;
      chk += cp;
    }
#pragma unroll
    for (int c = 0; c < 3; c++) {
      const double rh = fc[c] * (tf - to) * ut / 2.0;
      const double cv = lv[c] - rh;
      rb[4 * N + 3 * g + c] = cv;
      chk += cv;
    }
    if (ph.hold) {
#pragma unroll
      for (int c = 0; c < 4; c++) {
        const double cq = q[c] - xq[4 * ph.xa + c];
        rb[7 * N + 4 * g + c] = cq;
        chk += cq;
      }
    } else {
      double fq[4];
      quat_rate(q, xu[2 * g], xu[2 * g + 1], P.uu, fq);
#pragma unroll
      for (int c = 0; c < 4; c++) {
        const double rh = fq[c] * (tf - to) * ut / 2.0;
        const double cq = lq[c] - rh;
        rb[7 * N + 4 * g + c] = cq;
        chk += cq;
      }
    }
  }
#undef EMIT
#undef FDQ
  if (!(fabs(chk) <= 1.79769313486231570815e308)) atomicOr(P.flag, 1);
}

}  // namespace gel
