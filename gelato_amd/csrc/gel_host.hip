// gel_host.hip -- host side of the engine: LGR generator, fixed sparsity pattern,
// device-resident problem, and the C-ABI of include/gelato_amd.h.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <chrono>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <thread>
#include <vector>

#include "../../include/gelato_amd.h"
#include "gel_device.h"
#include "gel_launch.h"

namespace {

thread_local std::string g_err;
int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}
#define HIPCHK(expr)                                                                         \
  do {                                                                                       \
    hipError_t _e = (expr);                                                                  \
    if (_e != hipSuccess)                                                                    \
      return fail(GEL_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));           \
  } while (0)

// ---------------------------------------------------------------------------
// LGR nodes / differentiation matrix.
// What: lib/PSfunctions.py:149-168 (flipped LGR nodes, ending at +1) and :182-208
// (D[k][i] = l_i'(tau_x[k+1]) on tau_x = [-1, tau]).  How: the flipped LGR points
// are the negated roots of P_{n-1}(x) + P_n(x) (Legendre), found by Newton in
// extended precision from the classic -cos(2 pi k/(2n-1)) guesses; D comes from
// barycentric weights in O(n^2) instead of the reference's O(n^4) products.
// ---------------------------------------------------------------------------
typedef long double ld;

void legendre_pair(int n, ld x, ld& Pn, ld& Pn1) {  // P_n, P_{n-1}
  ld p0 = 1.0L, p1 = x;
  if (n == 0) { Pn = 1.0L; Pn1 = 0.0L; return; }
  for (int k = 2; k <= n; k++) {
    const ld p2 = ((2 * k - 1) * x * p1 - (k - 1) * p0) / k;
    p0 = p1; p1 = p2;
  }
  Pn = p1; Pn1 = p0;
}

int lgr_nodes_ld(int n, std::vector<ld>& tau) {
  if (n < 2) return -1;
  std::vector<ld> xs(n);
  xs[0] = -1.0L;  // the Radau end point of the un-flipped set
  const ld pi = 3.141592653589793238462643383279502884L;
  for (int k = 1; k < n; k++) {
    ld x = -cosl(2.0L * pi * k / (2 * n - 1));
    for (int it = 0; it < 100; it++) {
      ld Pn, Pn1;
      legendre_pair(n, x, Pn, Pn1);  // P_n, P_{n-1}
      ld Pm, Pm1;
      legendre_pair(n - 1, x, Pm, Pm1);  // P_{n-1}, P_{n-2}
      const ld f = Pn1 + Pn;
      // P_k'(x) = k (x P_k - P_{k-1}) / (x^2 - 1)
      const ld dPn = n * (x * Pn - Pn1) / (x * x - 1.0L);
      const ld dPn1 = (n - 1) * (x * Pm - Pm1) / (x * x - 1.0L);
      // deflate the known root at -1: g = f/(1+x)
      const ld gval = f / (1.0L + x);
      const ld dg = (dPn + dPn1) / (1.0L + x) - f / ((1.0L + x) * (1.0L + x));
      const ld step = gval / dg;
      x -= step;
      if (fabsl(step) < 1e-19L) break;
    }
    xs[k] = x;
  }
  tau.resize(n);
  for (int k = 0; k < n; k++) tau[k] = -xs[k];
  std::sort(tau.begin(), tau.end());
  tau[n - 1] = 1.0L;
  return 0;
}

int lgr_diffmat_ld(int n, std::vector<double>& D, std::vector<double>& tau_out) {
  std::vector<ld> tau;
  if (lgr_nodes_ld(n, tau)) return -1;
  std::vector<ld> t(n + 1), w(n + 1);
  t[0] = -1.0L;
  for (int k = 0; k < n; k++) t[k + 1] = tau[k];
  for (int i = 0; i <= n; i++) {
    ld p = 1.0L;
    for (int m = 0; m <= n; m++)
      if (m != i) p *= (t[i] - t[m]);
    w[i] = 1.0L / p;
  }
  D.assign((size_t)n * (n + 1), 0.0);
  for (int k = 0; k < n; k++) {
    ld diag = 0.0L;
    for (int i = 0; i <= n; i++) {
      if (i == k + 1) continue;
      const ld v = (w[i] / w[k + 1]) / (t[k + 1] - t[i]);
      D[(size_t)k * (n + 1) + i] = (double)v;
      diag -= v;
    }
    D[(size_t)k * (n + 1) + k + 1] = (double)diag;
  }
  tau_out.resize(n);
  for (int k = 0; k < n; k++) tau_out[k] = (double)tau[k];
  return 0;
}

// ---------------------------------------------------------------------------
struct HostPhase {
  int n, ua, xa;
  int air, air_fd, t_fd, q_fd, engine_on, hold;
  int K, s_vv, s_vq, s_vt, s_qq;
  int64_t voff;
  double thrust, massflow, area, nozzle;
  std::vector<double> D, tau;
};

}  // namespace

struct gel_problem {
  int device = 0;
  bool fd_recompute = false;   // GEL_FLAG_FD_RECOMPUTE (or a step too long for the difference form): the reference's recomputing sweeps
  hipStream_t stream = nullptr;
  gel::ProblemDev dev{};
  std::vector<HostPhase> ph;
  gel_dims dims{};
  int64_t block_off[GEL_NUM_BLOCKS + 1]{};
  double um, up, uv, uu, ut, dx, barC20;
  // host copies of the pattern-derived data
  std::vector<double> cval;      // [total_nnz] constants (x-dependent entries 0)
  std::vector<int32_t> src;      // [total_nnz] gather map: -1 constant; s >= 0: compact[s]; s <= -2: -compact[-2 - s]
  std::vector<int64_t> var_idx;  // [V] compact slot -> the first full index that takes it with a plus sign
  struct Run { int64_t dst0, dstride, src0, sstride, len; double sign; };
  std::vector<Run> var_runs;     // the gather map as constant-stride runs (about one per (phase, slot, use): the n nodes)
  std::vector<int32_t> chunk_phase;  // [nchunks] phase of every 64-node work item
  // packed unit-shard exchange (gel_shard_plan): contiguous unit ranges per rank; every unit's entries of a vector form one
  // block at unit_base[unit] inside its rank's slice; shard_pos = the map back to the ordinary layouts
  std::vector<int32_t> shard_begin;  // [nranks + 1]
  std::vector<int64_t> unit_base;    // [4 * nchunks]
  int64_t shard_width = 0;
  int64_t* d_unit_base = nullptr;
  int64_t* d_shard_pos = nullptr;    // [11N + V]: rank * width + offset of every res entry, then of every compact value
  // aero path constraints: kind 0 = AOA_max, 1 = dynamic_pressure_max, 2 = Q_alpha_max
  std::vector<gel::AeroRowDev> aero_rows[3];
  std::vector<gel::AeroNodeDev> aero_nodes;                   // the constrained state nodes, shared by the kinds
  gel::AeroNodeDev* d_aero_nodes = nullptr;
  // gel_eval_batch_aero_device (defect groups + aero rows, one output record per vector): the record's layout, the per-phase
  // records of the rows the fused kernel's lanes write, and the constrained nodes they do not reach (state node 0 of a phase,
  // phases without aerodynamics) -- left to aero_wide_kernel
  int64_t aero_ld = 0, aero_off_con[2][3] = {{0, 0, 0}, {0, 0, 0}}, aero_off_jac[2][3] = {{0, 0, 0}, {0, 0, 0}};
  std::vector<gel::AeroRowDev> aero_part_rows[2][3];          // the rows of part A (the lanes') and part B (the rest), per kind
  std::vector<int32_t> aero_part_of[3];                       // per row of a kind: part | (row inside the part) << 1
  std::vector<int64_t> aero_partA_base[3];                    // per row of part A: first double of its spec's block in the record
  int64_t aero_partA_len = 0;                                 // doubles of part A (part B's sections follow)
  int64_t aero_dump = 0;                                      // first double of the dump area at the end of part A
  std::vector<gel::AeroNodeDev> aero_part_nodes[2];
  std::vector<gel::AeroPhaseDev> aero_ph;
  gel::AeroPhaseDev* d_aero_ph = nullptr;
  gel::AeroNodeDev* d_aero_part_nodes[2] = {nullptr, nullptr};
  double *d_aero_x = nullptr, *d_aero_out = nullptr;          // working set of large host-buffer calls
  size_t d_aero_x_cap = 0, d_aero_out_cap = 0;                // doubles
  // device buffers (static)
  gel::PhaseDev* d_phases = nullptr;
  int32_t* d_node_phase = nullptr;
  int4* d_chunks = nullptr;         // work items in phase order (what gel_chunk_phase / shard ranges index)
  int4* d_chunks_sorted = nullptr;  // the same items, dearest phase type first (whole launches)
  double* d_Dsw = nullptr;          // D per work item in the feed order of v_mfma_f64_16x16x4_f64
  double* d_Dst = nullptr;          // the same, row-tile major (one wavefront per row tile)
  double* d_Dt = nullptr;
  double* d_tau = nullptr;
  double* d_tables = nullptr;
  double* d_cval = nullptr;
  int32_t* d_src = nullptr;
  int32_t *d_vdst = nullptr, *d_vsrc = nullptr;   // the x-dependent entries of the gather map: destination (ascending), signed source
  int32_t nvar_entries = 0;
  int32_t* d_vline = nullptr;     // the 64-byte lines (index / 8) of a value vector that hold an x-dependent entry
  int32_t nvar_lines = 0;
  // COO-direct output of the one-vector latency path (gel_eval_kernel.h "COO-DIRECT"): first entry of the eight groups of runs per
  // phase; the runs of the gather map that are left to the host (dense velocity / quaternion blocks); the engine's own full value
  // array in pinned host memory (constants laid down once; the kernel and the host scatter rewrite the x-dependent entries)
  std::vector<int32_t> coo_tab;   // [8 * S]; empty: the mode is not available for this problem
  std::vector<Run> rest_runs;
  int32_t* d_coo = nullptr;
  double* h_full = nullptr;
  double* cb_res = nullptr;       // pinned residual vector a caller may name as its own output (gel_pinned_buffers): no copy then
  double* cb_x[2] = {nullptr, nullptr};   // two pinned decision-vector buffers a caller may fill in turn and pass as x: read in place
  int32_t* d_flag = nullptr;
  // B = 1 / small-batch working set
  int capB = 0;
  double *d_x = nullptr, *d_res = nullptr, *d_jv = nullptr;
  double *h_x = nullptr, *h_res = nullptr, *h_jv = nullptr;  // pinned
  int32_t* h_flag = nullptr;                                  // pinned
  // gel_jac_fd working set: kept between calls; the residuals of all num_vars + 1 perturbed vectors are
  // re-used while x stays the same (the four groups are asked for one after the other)
  double *jfd_x = nullptr, *jfd_Xp = nullptr, *jfd_res = nullptr, *jfd_J = nullptr;
  size_t jfd_J_cap = 0;                                       // doubles
  // every phase as its own one-phase problem (gel_jac_fd differences phase by phase)
  gel::PhaseDev* d_subphases = nullptr;                       // [S] the phase records with ua = xa = 0
  int4* d_subchunks = nullptr;                                // the phase-ordered work items with phase index 0
  int32_t* d_colmap = nullptr;                                // per phase: local column -> global column
  std::vector<int32_t> sub_chunk0, sub_nchunks, sub_col0;     // [S] first work item, work items, first colmap entry
  std::vector<size_t> sub_res0;                               // [S] first double of the phase's residuals in jfd_res
  std::vector<double> jfd_last_x;                             // empty = nothing cached
  int32_t* d_done = nullptr;                                  // self-signalling one-vector launches (ProblemDev::done_flag): device counter,
  volatile int32_t* h_done = nullptr;                         // pinned host word, sequence number of the last armed launch
  int32_t done_seq = 0;
  long long done_ema_us = 100;                                // running estimate of a self-signalling launch's wait (sets the spin budget)
  int jfd_status = GEL_OK;
  // knot / terminal / user rows (gel_rows_configure)
  std::vector<gel::LinRowDev> lin_rows;
  std::vector<gel::FnRowDev> fn_rows;
  gel::LinRowDev* d_lin_rows = nullptr;
  gel::FnRowDev* d_fn_rows = nullptr;
  double* h_rows = nullptr;                                   // pinned outputs of small gel_rows_eval calls: con | jfn
  size_t h_rows_cap = 0;                                      // doubles
  double *d_rows_x = nullptr, *d_rows_out = nullptr;          // working set of large host-buffer calls
  size_t d_rows_x_cap = 0, d_rows_out_cap = 0;                // doubles
  double* h_aero = nullptr;                                   // pinned outputs of small gel_eval_aero calls
  size_t h_aero_cap = 0;                                      // doubles
  // large host batches (gel_eval_batch): two staging slots of kPipeEvals decision vectors each, every
  // slot with its own stream, so that PCIe in, kernel, PCIe out and the host copies of neighbouring
  // sub-batches overlap
  struct Slot {
    double *d_x = nullptr, *d_res = nullptr, *d_jv = nullptr;
    double *h_x = nullptr, *h_res = nullptr, *h_jv = nullptr;  // pinned
    int32_t *d_flag = nullptr, *h_flag = nullptr;
    hipStream_t stream = nullptr;
    int64_t first = 0;  // sub-batch in flight: [first, first + count)
    int count = 0;
    bool res = false, jac = false;
  } slot[2];
  int pipe_evals = 0;  // capacity of a slot (0 = not allocated)
};

namespace {

// kind: 0 = constant value; 1 = x-dependent, value = compact[slot]; 2 = x-dependent, value = -compact[slot]
// (slot = compact index within the eval; several entries may name the same slot)
using Visitor = std::function<void(int block, int64_t k, int32_t row, int32_t col, int kind, double cval, int64_t slot)>;

// Walks all 13 blocks in the reference's emission order (lib/con_dynamics.py:66-113,
// 155-213,292-496,536-632).  k is the index inside the block.
// Compact slots: see gel_eval_kernel.h (kSlotPT ...).  Layout of a phase's K * n node values: 64-node chunk major,
// [chunk][slot][node of the chunk] -- everything one wavefront writes is one contiguous block (a [slot][n] layout makes a
// wavefront of a phase above 64 nodes write 512 bytes out of every n * 8: measured 8 % slower on 12 x 128).  The per-phase
// scalar sits behind them at voff + K * n.
static inline int64_t compact_index(const HostPhase& h, int slot, int j) {
  const int j0 = j & ~63, w = std::min(64, h.n - j0);
  return h.voff + (int64_t)j0 * h.K + (int64_t)slot * w + (j - j0);
}
void walk_pattern(const gel_problem& P, const Visitor& vis) {
  const int S = (int)P.ph.size();
  int64_t k[GEL_NUM_BLOCKS] = {0};
  auto cs = [&](const HostPhase& h, int slot, int j) { return compact_index(h, slot, j); };
  for (int i = 0; i < S; i++) {
    const HostPhase& h = P.ph[i];
    const int n = h.n, ua = h.ua, xa = h.xa;
    const int64_t scalar = h.voff + (int64_t)h.K * n;  // pos/velocity diagonal value of this phase
    auto Dji = [&](int j, int c) { return h.D[(size_t)j * (n + 1) + c]; };
    // ---- group 0: mass ----
    if (h.engine_on) {
      for (int j = 0; j < n; j++)
        for (int c = 0; c <= n; c++) vis(0, k[0]++, ua + j, xa + c, 0, Dji(j, c), -1);
      const double a = -h.massflow / P.um * P.ut / 2.0, b = h.massflow / P.um * P.ut / 2.0;
      for (int j = 0; j < n; j++) vis(1, k[1]++, ua + j, i, 0, a, -1);
      for (int j = 0; j < n; j++) vis(1, k[1]++, ua + j, i + 1, 0, b, -1);
    } else {
      for (int j = 0; j < n; j++) vis(0, k[0]++, ua + j, xa, 0, -1.0, -1);
      for (int j = 0; j < n; j++) vis(0, k[0]++, ua + j, xa + 1 + j, 0, 1.0, -1);
    }
    // ---- group 1: position ----
    for (int ki = 0; ki < 3; ki++)
      for (int j = 0; j < n; j++)
        for (int c = 0; c <= n; c++) vis(2, k[2]++, 3 * (ua + j) + ki, 3 * (xa + c) + ki, 0, Dji(j, c), -1);
    for (int jj = 0; jj < 3 * n; jj++) vis(3, k[3]++, 3 * ua + jj, 3 * (xa + 1) + jj, 1, 0, scalar);
    for (int jj = 0; jj < 3 * n; jj++) vis(4, k[4]++, 3 * ua + jj, i, 1, 0, cs(h, 0 + jj % 3, jj / 3));
    for (int jj = 0; jj < 3 * n; jj++) vis(4, k[4]++, 3 * ua + jj, i + 1, 2, 0, cs(h, 0 + jj % 3, jj / 3));
    // ---- group 2: velocity ----
    for (int j = 0; j < n; j++)
      for (int c = 0; c < 3; c++) vis(5, k[5]++, 3 * (ua + j) + c, xa + 1 + j, 1, 0, cs(h, 3 + c, j));
    for (int kk = 0; kk < 3; kk++)
      for (int j = 0; j < n; j++)
        for (int c = 0; c < 3; c++) vis(6, k[6]++, 3 * (ua + j) + c, 3 * (xa + 1 + j) + kk, 1, 0, cs(h, 6 + 3 * kk + c, j));
    for (int ki = 0; ki < 3; ki++)
      for (int kj = 0; kj < 3; kj++)
        for (int j = 0; j < n; j++)
          for (int cc = 0; cc <= n; cc++) {
            const double dv = (ki == kj) ? Dji(j, cc) : 0.0;
            if (cc == j + 1 && h.air_fd)
              vis(7, k[7]++, 3 * (ua + j) + ki, 3 * (xa + cc) + kj, 1, 0, cs(h, h.s_vv + 3 * kj + ki, j));
            else
              vis(7, k[7]++, 3 * (ua + j) + ki, 3 * (xa + cc) + kj, 0, dv, -1);
          }
    for (int kk = 0; kk < 4; kk++)
      for (int j = 0; j < n; j++)
        for (int c = 0; c < 3; c++) vis(8, k[8]++, 3 * (ua + j) + c, 4 * (xa + 1 + j) + kk, 1, 0, cs(h, h.s_vq + 3 * kk + c, j));
    // t0 column, then tf column: two sweeps when the velocity / time derivatives are finite differences
    // (reference_area > 0), else the tf column is the exact negative of the t0 column (:478-480)
    for (int jj = 0; jj < 3 * n; jj++) vis(9, k[9]++, 3 * ua + jj, i, 1, 0, cs(h, h.s_vt + jj % 3, jj / 3));
    for (int jj = 0; jj < 3 * n; jj++)
      vis(9, k[9]++, 3 * ua + jj, i + 1, h.t_fd ? 1 : 2, 0, cs(h, h.s_vt + (h.t_fd ? 3 : 0) + jj % 3, jj / 3));
    // ---- group 3: quaternion ----
    if (h.hold) {
      for (int jj = 0; jj < 4 * n; jj++) vis(10, k[10]++, 4 * ua + jj, 4 * xa + jj % 4, 0, -1.0, -1);
      for (int jj = 0; jj < 4 * n; jj++) vis(10, k[10]++, 4 * ua + jj, 4 * (xa + 1) + jj, 0, 1.0, -1);
    } else {
      for (int j = 0; j < n; j++)
        for (int c = 0; c < 4; c++)
          for (int cc = 0; cc <= n; cc++)
            for (int kk = 0; kk < 4; kk++) {
              // on the block diagonal dq_c/dq_kk is a finite difference only where dq_c contains q_kk
              // (src/pybind_dynamics.cpp:94-106: rows {0,1} x columns {2,3} and rows {2,3} x columns {0,1});
              // elsewhere the reference's difference is exactly zero and the entry is D[j][j+1] or 0
              if (cc == j + 1 && ((c < 2) != (kk < 2))) {
                if (h.q_fd)
                  vis(10, k[10]++, 4 * (ua + j) + c, 4 * (xa + cc) + kk, 1, 0, cs(h, h.s_qq + 2 * kk + (c & 1), j));
                else {
                  // closed form (dq is linear in q): two values per node, A = omega_y S / 2 (slot s_qq) and B = omega_z S / 2
                  // (s_qq + 1), S = (tf - to) unit_t / 2; entry (c, kk) = -d(dq_c)/d(q_kk) S  (src/pybind_dynamics.cpp:94-106)
                  static const int which[4][4] = {{-1, -1, 0, 1}, {-1, -1, 1, 0}, {0, 1, -1, -1}, {1, 0, -1, -1}};   // [c][kk]
                  static const int sign[4][4] = {{0, 0, +1, +1}, {0, 0, -1, +1}, {-1, +1, 0, 0}, {-1, -1, 0, 0}};
                  vis(10, k[10]++, 4 * (ua + j) + c, 4 * (xa + cc) + kk, sign[c][kk] > 0 ? 1 : 2, 0, cs(h, h.s_qq + which[c][kk], j));
                }
              } else
                vis(10, k[10]++, 4 * (ua + j) + c, 4 * (xa + cc) + kk, 0, (c == kk) ? Dji(j, cc) : 0.0, -1);
            }
      for (int kk = 0; kk < 2; kk++)
        for (int j = 0; j < n; j++)
          for (int c = 0; c < 4; c++) {
            if (h.q_fd)
              vis(11, k[11]++, 4 * (ua + j) + c, 2 * (ua + j) + kk, 1, 0, cs(h, h.s_qq + 8 + 4 * kk + c, j));
            else {
              // closed form (dq is linear in u): slots s_qq + 2 .. 5 hold C_0' = -C_0, C_1, C_2, C_3 with C_i = unit_u (pi/180) q_i S / 2
              // (the first negated so that every slot has a positive use); entry (c, kk) = -d(dq_c)/d(u_kk) S
              static const int which[2][4] = {{2, 3, 0, 1}, {3, 2, 1, 0}};          // [kk][c]: which C
              static const int sign[2][4] = {{+1, +1, -1, -1}, {+1, -1, +1, -1}};
              const int w = which[kk][c];
              const int sg = (w == 0) ? -sign[kk][c] : sign[kk][c];                  // slot of C_0 stores -C_0
              vis(11, k[11]++, 4 * (ua + j) + c, 2 * (ua + j) + kk, sg > 0 ? 1 : 2, 0, cs(h, h.s_qq + 2 + w, j));
            }
          }
      const int s_qt = h.s_qq + (h.q_fd ? 16 : 6);
      for (int jj = 0; jj < 4 * n; jj++) vis(12, k[12]++, 4 * ua + jj, i, 1, 0, cs(h, s_qt + jj % 4, jj / 4));
      for (int jj = 0; jj < 4 * n; jj++) vis(12, k[12]++, 4 * ua + jj, i + 1, 2, 0, cs(h, s_qt + jj % 4, jj / 4));
    }
  }
}

int64_t phase_block_nnz(const HostPhase& h, int blk) {
  const int64_t n = h.n;
  switch (blk) {
    case 0: return h.engine_on ? n * (n + 1) : 2 * n;
    case 1: return h.engine_on ? 2 * n : 0;
    case 2: return 3 * n * (n + 1);
    case 3: return 3 * n;
    case 4: return 6 * n;
    case 5: return 3 * n;
    case 6: return 9 * n;
    case 7: return 9 * n * (n + 1);
    case 8: return 12 * n;
    case 9: return 6 * n;
    case 10: return h.hold ? 8 * n : 16 * n * (n + 1);
    case 11: return h.hold ? 0 : 8 * n;
    default: return h.hold ? 0 : 8 * n;
  }
}

template <class T>
int upload(T** d, const std::vector<T>& h) {
  HIPCHK(hipMalloc((void**)d, std::max<size_t>(1, h.size()) * sizeof(T)));
  if (!h.empty()) HIPCHK(hipMemcpy(*d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
  return GEL_OK;
}

// global column of every local column of phase i as a one-phase problem: [mass n+1 | pos 3(n+1) | vel 3(n+1) | quat 4(n+1) | u 2n | t0 tf]
static void phase_columns(const gel_problem* p, int i, std::vector<int32_t>& colmap) {
  const HostPhase& h = p->ph[i];
  const int n = h.n, M = (int)p->dims.M, N = (int)p->dims.N;
  for (int k = 0; k <= n; k++) colmap.push_back(h.xa + k);                                  // mass
  for (int k = 0; k < 3 * (n + 1); k++) colmap.push_back(M + 3 * h.xa + k);                  // position
  for (int k = 0; k < 3 * (n + 1); k++) colmap.push_back(4 * M + 3 * h.xa + k);              // velocity
  for (int k = 0; k < 4 * (n + 1); k++) colmap.push_back(7 * M + 4 * h.xa + k);              // quaternion
  for (int k = 0; k < 2 * n; k++) colmap.push_back(11 * M + 2 * h.ua + k);                   // u
  colmap.push_back(11 * M + 2 * N + i); colmap.push_back(11 * M + 2 * N + i + 1);            // t0, tf
}

#define NEED_DEVICE(p)                                                                              \
  do {                                                                                              \
    if ((p)->device == GEL_DEVICE_NONE)                                                             \
      return fail(GEL_ERR_HIP, "host-only handle: nothing can be evaluated without a GPU (no CPU fallback)"); \
  } while (0)

// calls moving less than this are served zero-copy out of the pinned staging buffers (run_host)
constexpr size_t kZeroCopyBytes = (size_t)1 << 20;

int ensure_capacity(gel_problem* p, int B) {
  NEED_DEVICE(p);
  if (B <= p->capB) return GEL_OK;
  HIPCHK(hipSetDevice(p->device));
  // release first and forget the old capacity: a failed allocation below must not leave freed pointers behind
  hipFree(p->d_x); hipFree(p->d_res); hipFree(p->d_jv);
  if (p->h_x) hipHostFree(p->h_x);
  if (p->h_res) hipHostFree(p->h_res);
  if (p->h_jv) hipHostFree(p->h_jv);
  p->d_x = p->d_res = p->d_jv = p->h_x = p->h_res = p->h_jv = nullptr;
  p->capB = 0;
  const size_t nx = (size_t)B * p->dims.num_vars, nr = (size_t)B * 11 * p->dims.N, nj = (size_t)B * std::max<int64_t>(1, p->dims.num_var_entries);
  HIPCHK(hipMalloc((void**)&p->d_x, nx * 8));
  HIPCHK(hipMalloc((void**)&p->d_res, nr * 8));
  HIPCHK(hipMalloc((void**)&p->d_jv, nj * 8));
  HIPCHK(hipHostMalloc((void**)&p->h_x, nx * 8));
  HIPCHK(hipHostMalloc((void**)&p->h_res, nr * 8));
  HIPCHK(hipHostMalloc((void**)&p->h_jv, nj * 8));
  p->capB = B;
  return GEL_OK;
}

// Wait for a short launch.  Round 5, measured on the one-vector path (three resident processes taking turns, same box): polling
// hipStreamQuery 39.1 us per Engine.eval, hipStreamSynchronize 37.3 us, the same after hipSetDeviceFlags(hipDeviceScheduleSpin) 37.0 us
// -- this runtime's synchronise spins on the completion signal itself before it sleeps, and a query is a full API call per poll
// (rounds 1-4 measured the opposite on an earlier runtime and polled).  GEL_WAIT_MODE=0 restores the polling loop (measurement switch).
// Self-signalling one-vector launches: the kernel's last workgroup stores the launch's sequence number to a pinned host word once
// every workgroup's results are visible system-wide (gel_eval_kernel.h signal_done); the host spins on that word instead of waiting
// for the end-of-kernel signal, which the runtime hands over 4 us later at 6 x 64 (29 -> 25 us launch-to-results).  The stream still
// holds the kernel's tail; whatever is launched next on it is ordered behind.  GEL_DONE_FLAG=0: off (measurement switch).
static bool done_flag_on() {
  static const bool on = [] { const char* e = getenv("GEL_DONE_FLAG"); return !e || atoi(e) != 0; }();
  return on;
}
static void arm_done(gel_problem* p, gel::ProblemDev& dv) {
  if (!p->d_done || !p->h_done || !done_flag_on()) return;
  dv.done_ctr = p->d_done;
  dv.done_flag = const_cast<int32_t*>(p->h_done);
  dv.done_seq = ++p->done_seq;
}
static hipError_t spin_wait(hipStream_t s);
static inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
  __builtin_ia32_pause();
#elif defined(__aarch64__)
  asm volatile("yield" ::: "memory");
#else
  std::this_thread::yield();
#endif
}
static hipError_t wait_done(gel_problem* p, const gel::ProblemDev& dv) {
  if (dv.done_flag) {
    // Spin budget: a few times what this handle's launches have taken so far (floor 200 us, ceiling 5 ms) -- with several resident
    // processes taking turns on the GPU, or a preempted queue, a fixed 5 ms burnt a core per call before the runtime's wait was
    // taken anyway.  A launch that never signals (it should not happen) falls back to that wait as well.
    const auto t0 = std::chrono::steady_clock::now();
    const auto budget = std::chrono::microseconds(std::min<long long>(5000, std::max<long long>(200, 8 * p->done_ema_us)));
    for (unsigned it = 1;; it++) {
      if (*p->h_done == dv.done_seq) {
        std::atomic_thread_fence(std::memory_order_acquire);
        const long long us = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
        p->done_ema_us = (3 * p->done_ema_us + us + 3) / 4;
        return hipSuccess;
      }
      cpu_relax();
      if ((it & 1023u) == 0 && std::chrono::steady_clock::now() - t0 > budget) break;
    }
    p->done_ema_us = std::min<long long>(5000, 2 * p->done_ema_us + 50);   // the wait outlasted the budget: a longer one next time
  }
  return spin_wait(p->stream);
}

static hipError_t spin_wait(hipStream_t s) {
  static const int mode = [] { const char* e = getenv("GEL_WAIT_MODE"); return e ? atoi(e) : 1; }();
  if (mode) return hipStreamSynchronize(s);
  hipError_t q;
  while ((q = hipStreamQuery(s)) == hipErrorNotReady) {}
  return q;
}

// The engine's own full COO value array and residual vector in pinned host memory (one-vector latency path): the constants are
// laid down here, once; the COO-direct kernel and the host's scatter of the few entries left to it rewrite the x-dependent ones.
int ensure_full(gel_problem* p) {
  if (p->h_full && p->cb_res && p->cb_x[0] && p->cb_x[1]) return GEL_OK;
  HIPCHK(hipSetDevice(p->device));
  // each buffer on its own: a failed allocation leaves the others as they are and the next call tries again (no half-made set is
  // ever handed out: gel_pinned_buffers and the zero-copy checks only run after this function has returned GEL_OK)
  if (!p->h_full) {
    HIPCHK(hipHostMalloc((void**)&p->h_full, std::max<size_t>(1, p->cval.size()) * 8));
    std::memcpy(p->h_full, p->cval.data(), p->cval.size() * 8);
  }
  if (!p->cb_res) HIPCHK(hipHostMalloc((void**)&p->cb_res, (size_t)11 * p->dims.N * 8));
  for (int i = 0; i < 2; i++)
    if (!p->cb_x[i]) HIPCHK(hipHostMalloc((void**)&p->cb_x[i], (size_t)p->dims.num_vars * 8));
  return GEL_OK;
}
// GEL_NO_COO_DIRECT=1 (measurement switch): the compact layout + the host scatter of every x-dependent entry, as before round 5
bool coo_direct(const gel_problem* p) {
  static const bool off = [] { const char* e = getenv("GEL_NO_COO_DIRECT"); return e && atoi(e) != 0; }();
  return !off && !p->coo_tab.empty() && p->d_coo != nullptr;
}
// After a COO-direct evaluation: the entries the kernel left to the host (the diagonal of the dense velocity block, the pairs of the
// dense quaternion block: 11 of a node's 49 slots) go from the compact vector into the engine's array; a caller that brought its
// own array gets the all-x-dependent blocks as whole copies and the same few scattered entries.
void finish_full(gel_problem* p, double* vals_full, int fill) {
  auto rest = [&](double* out) {
    for (const gel_problem::Run& r : p->rest_runs) {
      double* d = out + r.dst0;
      const double* v = p->h_jv + r.src0;
      if (r.sign > 0) for (int64_t k = 0; k < r.len; k++) d[k * r.dstride] = v[k * r.sstride];
      else for (int64_t k = 0; k < r.len; k++) d[k * r.dstride] = -v[k * r.sstride];
    }
  };
  rest(p->h_full);
  if (vals_full == p->h_full) return;
  if (fill) std::memcpy(vals_full, p->cval.data(), p->cval.size() * 8);
  for (int b : {3, 4, 5, 6, 8, 9, 11, 12})
    std::memcpy(vals_full + p->block_off[b], p->h_full + p->block_off[b], (size_t)(p->block_off[b + 1] - p->block_off[b]) * 8);
  rest(vals_full);
}

// run B evals from host x; leaves results in the pinned staging buffers (res_to: where the kernel writes the residual vector
// of a one-vector call -- the handle's staging buffer or the pinned vector the caller named; coo: COO-direct Jacobian output)
int run_host(gel_problem* p, int B, const double* x, bool want_res, bool want_jac, double* res_to = nullptr, bool* coo_io = nullptr) {
  const bool coo = coo_io && *coo_io;
  if (coo_io) *coo_io = false;   // set again where the mode is really used (the zero-copy branch)
  int rc = ensure_capacity(p, B);
  if (rc) return rc;
  HIPCHK(hipSetDevice(p->device));
  const size_t nx = (size_t)B * p->dims.num_vars, nr = (size_t)B * 11 * p->dims.N, nj = (size_t)B * p->dims.num_var_entries;
  // a one-vector call whose x IS one of the handle's pinned decision-vector buffers (gel_pinned_buffers) is read in place
  const bool x_pinned = B == 1 && p->cb_x[0] && (x == p->cb_x[0] || x == p->cb_x[1]);
  if (!x_pinned) std::memcpy(p->h_x, x, nx * 8);
  const double* const xin = x_pinned ? x : p->h_x;
  if ((nx + (want_res ? nr : 0) + (want_jac ? nj : 0)) * 8 <= kZeroCopyBytes) {
    // Small calls (the optimiser's one-vector callbacks): the kernel reads x from and writes its results
    // to the pinned staging buffers itself -- one launch and one synchronise instead of launch + four
    // copy commands (measured 91 -> see DESIGN.md 5, B = 1).  The non-finite flag is a plain store of 1,
    // so it may live in host memory too.
    gel::ProblemDev dv = p->dev;
    dv.flag = p->h_flag;
    // COO-direct output exists in the latency form only: a problem of more than 256 work items takes a cooperative form even for one
    // vector (gel::eval_form) and keeps the compact path
    if (coo && B == 1 && want_jac && gel::eval_form(dv, 1, want_res, true).split) { dv.coo_full = p->h_full; dv.coo = p->d_coo; *coo_io = true; }
    dv.split_vel = 1;   // a whole evaluation: every part of every work item is in this launch
    if (gel::eval_form(dv, B, want_res, want_jac).split) arm_done(p, dv);   // the latency form tells the host itself when its results are there
    HIPCHK(gel::launch_eval(dv, B, xin, want_res ? (res_to ? res_to : p->h_res) : nullptr, want_jac ? p->h_jv : nullptr, p->stream));
    HIPCHK(wait_done(p, dv));
    if (*p->h_flag) { *p->h_flag = 0; return GEL_NONFINITE; }
    return GEL_OK;
  }
  HIPCHK(hipMemcpyAsync(p->d_x, xin, nx * 8, hipMemcpyHostToDevice, p->stream));
  HIPCHK(gel::launch_eval(p->dev, B, p->d_x, want_res ? p->d_res : nullptr, want_jac ? p->d_jv : nullptr, p->stream));
  if (want_res) HIPCHK(hipMemcpyAsync(p->h_res, p->d_res, nr * 8, hipMemcpyDeviceToHost, p->stream));
  if (want_jac && nj) HIPCHK(hipMemcpyAsync(p->h_jv, p->d_jv, nj * 8, hipMemcpyDeviceToHost, p->stream));
  HIPCHK(hipMemcpyAsync(p->h_flag, p->d_flag, 4, hipMemcpyDeviceToHost, p->stream));
  HIPCHK(hipStreamSynchronize(p->stream));
  if (want_res && res_to) std::memcpy(res_to, p->h_res, nr * 8);
  if (*p->h_flag) {
    HIPCHK(hipMemsetAsync(p->d_flag, 0, 4, p->stream));
    return GEL_NONFINITE;
  }
  return GEL_OK;
}

// ---- pipelined host batch ----
constexpr size_t kPipeBytes = (size_t)16 << 20;  // staging per slot and direction

// memcpy on a few threads: one core moves ~10 GB/s, PCIe delivers 2-5x that
void par_copy(void* dst, const void* src, size_t bytes) {
  const size_t kMin = (size_t)2 << 20;
  const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
  const size_t nt = std::min<size_t>(std::min<size_t>(4, hw), bytes / kMin);
  if (nt <= 1) { std::memcpy(dst, src, bytes); return; }
  std::vector<std::thread> th;
  const size_t per = ((bytes / nt) + 63) & ~(size_t)63;
  for (size_t t = 1; t < nt; t++) {
    const size_t off = t * per, len = (t + 1 == nt) ? bytes - off : per;
    th.emplace_back([=] { std::memcpy((char*)dst + off, (const char*)src + off, len); });
  }
  std::memcpy(dst, src, per);
  for (auto& t : th) t.join();
}

void free_slots(gel_problem* p);

int ensure_slots(gel_problem* p) {
  NEED_DEVICE(p);
  if (p->pipe_evals) return GEL_OK;
  HIPCHK(hipSetDevice(p->device));
  const size_t per_eval = 8 * (size_t)std::max<int64_t>(p->dims.num_var_entries, 1);
  const int cap = (int)std::max<size_t>(1, kPipeBytes / per_eval);
  const size_t nx = (size_t)cap * p->dims.num_vars, nr = (size_t)cap * 11 * p->dims.N, nj = (size_t)cap * std::max<int64_t>(1, p->dims.num_var_entries);
  for (auto& sl : p->slot) {
    const bool ok = hipMalloc((void**)&sl.d_x, nx * 8) == hipSuccess && hipMalloc((void**)&sl.d_res, nr * 8) == hipSuccess &&
                    hipMalloc((void**)&sl.d_jv, nj * 8) == hipSuccess && hipMalloc((void**)&sl.d_flag, 4) == hipSuccess &&
                    hipMemset(sl.d_flag, 0, 4) == hipSuccess && hipHostMalloc((void**)&sl.h_x, nx * 8) == hipSuccess &&
                    hipHostMalloc((void**)&sl.h_res, nr * 8) == hipSuccess && hipHostMalloc((void**)&sl.h_jv, nj * 8) == hipSuccess &&
                    hipHostMalloc((void**)&sl.h_flag, 4) == hipSuccess && hipStreamCreate(&sl.stream) == hipSuccess;
    if (!ok) {
      free_slots(p);   // nothing half-allocated survives: the next call starts from scratch instead of leaking
      return fail(GEL_ERR_ALLOC, "staging slots: allocation failed");
    }
    *sl.h_flag = 0;
    sl.count = 0;
  }
  p->pipe_evals = cap;
  return GEL_OK;
}

void free_slots(gel_problem* p) {
  for (auto& sl : p->slot) {
    if (sl.stream) { hipStreamSynchronize(sl.stream); hipStreamDestroy(sl.stream); }
    hipFree(sl.d_x); hipFree(sl.d_res); hipFree(sl.d_jv); hipFree(sl.d_flag);
    if (sl.h_x) hipHostFree(sl.h_x);
    if (sl.h_res) hipHostFree(sl.h_res);
    if (sl.h_jv) hipHostFree(sl.h_jv);
    if (sl.h_flag) hipHostFree(sl.h_flag);
    sl = gel_problem::Slot{};
  }
  p->pipe_evals = 0;
}

// B evals from / to pageable host arrays in sub-batches: while sub-batch i runs (H2D, kernel, D2H on its
// slot's stream), the host retires sub-batch i-1 of the other slot (pinned -> caller) and stages i+1.
// Is this host pointer page-locked memory the runtime knows (hipHostMalloc / hipHostRegister, e.g. a torch tensor made with
// pin_memory = True)?  Then the copy engines read and write it directly and the staging copy through the handle's slots is skipped.
static bool is_pinned(const void* ptr) {
  if (!ptr) return false;
  hipPointerAttribute_t at;
  if (hipPointerGetAttributes(&at, ptr) != hipSuccess) { (void)hipGetLastError(); return false; }
  return at.type == hipMemoryTypeHost;
}

int run_host_pipelined(gel_problem* p, int B, const double* x, double* res, double* jvar) {
  int rc = ensure_slots(p);
  if (rc) return rc;
  HIPCHK(hipSetDevice(p->device));
  // page-locked caller buffers (all that are given): no staging copies on the host at all -- H2D straight from x, D2H straight into
  // res / jvar, sub-batch by sub-batch on the two slots' streams; the host only waits
  const bool direct = is_pinned(x) && (!res || is_pinned(res)) && (!jvar || is_pinned(jvar));
  const int cap = p->pipe_evals;
  const size_t nv = (size_t)p->dims.num_vars, nr = (size_t)11 * p->dims.N, nj = (size_t)p->dims.num_var_entries;
  const int nsub = (B + cap - 1) / cap;
  int status = GEL_OK;
  auto retire = [&](gel_problem::Slot& sl) -> int {
    if (!sl.count) return GEL_OK;
    HIPCHK(hipStreamSynchronize(sl.stream));
    if (*sl.h_flag) {
      *sl.h_flag = 0;
      HIPCHK(hipMemsetAsync(sl.d_flag, 0, 4, sl.stream));
      status = GEL_NONFINITE;
    }
    if (!direct) {
      if (sl.res) par_copy(res + (size_t)sl.first * nr, sl.h_res, (size_t)sl.count * nr * 8);
      if (sl.jac && nj) par_copy(jvar + (size_t)sl.first * nj, sl.h_jv, (size_t)sl.count * nj * 8);
    }
    sl.count = 0;
    return GEL_OK;
  };
  for (int i = 0; i < nsub; i++) {
    gel_problem::Slot& sl = p->slot[i & 1];
    if ((rc = retire(sl))) break;
    const int64_t first = (int64_t)i * cap;
    const int count = (int)std::min<int64_t>(cap, B - first);
    if (!direct) par_copy(sl.h_x, x + (size_t)first * nv, (size_t)count * nv * 8);
    sl.first = first; sl.count = count; sl.res = res != nullptr; sl.jac = jvar != nullptr;
    gel::ProblemDev dv = p->dev;
    dv.flag = sl.d_flag;
    const double* const hx = direct ? x + (size_t)first * nv : sl.h_x;
    double* const hres = direct ? res + (size_t)first * nr : sl.h_res;
    double* const hjv = direct ? jvar + (size_t)first * nj : sl.h_jv;
    if (hipMemcpyAsync(sl.d_x, hx, (size_t)count * nv * 8, hipMemcpyHostToDevice, sl.stream) != hipSuccess ||
        gel::launch_eval(dv, count, sl.d_x, res ? sl.d_res : nullptr, jvar ? sl.d_jv : nullptr, sl.stream) != hipSuccess ||
        (res && hipMemcpyAsync(hres, sl.d_res, (size_t)count * nr * 8, hipMemcpyDeviceToHost, sl.stream) != hipSuccess) ||
        (jvar && nj && hipMemcpyAsync(hjv, sl.d_jv, (size_t)count * nj * 8, hipMemcpyDeviceToHost, sl.stream) != hipSuccess) ||
        hipMemcpyAsync(sl.h_flag, sl.d_flag, 4, hipMemcpyDeviceToHost, sl.stream) != hipSuccess) {
      rc = fail(GEL_ERR_HIP, "pipelined batch: enqueue failed");
      break;
    }
  }
  // drain both slots (also after an error, so that nothing is left in flight)
  for (int k = 0; k < 2; k++) {
    const int r2 = retire(p->slot[(nsub + k) & 1]);
    if (!rc) rc = r2;
  }
  return rc ? rc : status;
}

void scatter_full(const gel_problem* p, const double* jv, double* vals_full, int fill) {
  if (fill) std::memcpy(vals_full, p->cval.data(), p->cval.size() * 8);
  // the index map is a few hundred constant-stride runs: no index loads, predictable strides
  for (const gel_problem::Run& r : p->var_runs) {
    double* d = vals_full + r.dst0;
    const double* v = jv + r.src0;
    if (r.sign > 0) for (int64_t k = 0; k < r.len; k++) d[k * r.dstride] = v[k * r.sstride];
    else for (int64_t k = 0; k < r.len; k++) d[k * r.dstride] = -v[k * r.sstride];
  }
}

}  // namespace

namespace gel {
void fill_atmosphere_table(double* atm) {
  // src/Air.cpp:31-45
  const double lmb[11] = {-0.0065, 0.0, 0.001, 0.0028, 0.0, -0.0028, -0.002, 0.0, 0.0025, 0.012, 0.012};
  const double tmb[11] = {288.15, 216.65, 216.65, 228.65, 270.65, 270.65, 214.65, 186.8673, 186.8673, 240.0, 360.0};
  const double pb[11] = {101325.0, 22632.0, 5474.9, 868.02, 110.91, 66.939, 3.9564, 0.37338, 0.15381, 7.1042e-3, 2.5382e-3};
  const double mb[11] = {28.9644, 28.9644, 28.9644, 28.9644, 28.9644, 28.9644, 28.9644, 28.9522, 28.89, 27.27, 26.20};
  for (int k = 0; k < 11; k++) {
    atm[k] = lmb[k];
    atm[11 + k] = tmb[k];
    atm[22 + k] = pb[k];
    atm[33 + k] = 8314.32 / mb[k];  // Rstar / mb[k], src/Air.cpp:67
    // the two per-layer constants of Air::pressure (src/Air.cpp:93-97), reference operation order
    const double R = atm[33 + k], g0 = 9.80665;
    atm[44 + k] = (std::fabs(lmb[k]) > 1.0e-6) ? (-g0 / lmb[k] / R) : 0.0;
    atm[55 + k] = g0 / R;
    static const double hb[11] = {0.0, 11000.0, 20000.0, 32000.0, 47000.0, 51000.0, 71000.0, 86000.0, 91000.0, 110000.0, 120000.0};
    atm[66 + k] = hb[k];  // src/Air.cpp:29-30
    atm[77 + k] = 1.0 / tmb[k];
  }
}

// validates the interpolation tables (np.interp's precondition; the bracket search and the tabulated slopes of
// the device lookup rely on >= 2 rows and strictly increasing abscissae) -- nullptr if fine, else the complaint
const char* check_tables(const double* wind, int Kw, const double* ca, int Kc) {
  if (!wind || !ca || Kw < 2 || Kc < 2) return "wind and CA tables need at least two rows";
  for (int k = 1; k < Kw; k++)
    if (!(wind[3 * k] > wind[3 * (k - 1)])) return "wind table altitudes must increase strictly";
  for (int k = 1; k < Kc; k++)
    if (!(ca[2 * k] > ca[2 * (k - 1)])) return "CA table Mach numbers must increase strictly";
  return nullptr;
}

// rows [K][w] followed by the slopes (y[k+1][c] - y[k][c]) / (x[k+1] - x[k]) of every interval, c = 1..w-1
void append_rows_and_slopes(std::vector<double>& rows_out, std::vector<double>& slopes_out, const double* tab, int K, int w) {
  rows_out.insert(rows_out.end(), tab, tab + (size_t)K * w);
  for (int k = 0; k + 1 < K; k++)
    for (int c = 1; c < w; c++)
      slopes_out.push_back((tab[(size_t)(k + 1) * w + c] - tab[(size_t)k * w + c]) / (tab[(size_t)(k + 1) * w] - tab[(size_t)k * w]));
}

// atmosphere | wind rows | CA rows | wind slopes | CA slopes  (gel_physics.h table_doubles())
std::vector<double> build_tables(const double* wind, int Kw, const double* ca, int Kc) {
  std::vector<double> t(kAtmTableDoubles), slopes;
  fill_atmosphere_table(t.data());
  append_rows_and_slopes(t, slopes, wind, Kw, 3);
  append_rows_and_slopes(t, slopes, ca, Kc, 2);
  t.insert(t.end(), slopes.begin(), slopes.end());
  return t;
}
}  // namespace gel

extern "C" {

const char* gel_last_error(void) { return g_err.c_str(); }
const char* gel_version(void) { return "gelato_amd 0.1 (gfx950, fp64)"; }

int gel_lgr_nodes(int32_t n, double* tau) {
  std::vector<ld> t;
  if (!tau || lgr_nodes_ld(n, t)) return fail(GEL_ERR_ARG, "gel_lgr_nodes: n >= 2 required");
  for (int k = 0; k < n; k++) tau[k] = (double)t[k];
  return GEL_OK;
}

int gel_lgr_diffmat(int32_t n, double* D) {
  std::vector<double> d, t;
  if (!D || lgr_diffmat_ld(n, d, t)) return fail(GEL_ERR_ARG, "gel_lgr_diffmat: n >= 2 required");
  std::memcpy(D, d.data(), d.size() * 8);
  return GEL_OK;
}

int gel_problem_create(const gel_problem_desc* d, gel_problem** out) {
  if (!d || !out) return fail(GEL_ERR_ARG, "null argument");
  if (d->num_sections < 1 || !d->num_nodes || !d->thrust || !d->massflow || !d->reference_area || !d->nozzle_area ||
      !d->engine_on || !d->attitude_hold || !d->wind_table || !d->ca_table || d->wind_rows < 2 || d->ca_rows < 2)
    return fail(GEL_ERR_ARG, "incomplete problem description");
  if (!(d->dx > 0.0)) return fail(GEL_ERR_ARG, "dx must be positive");
  if (const char* why = gel::check_tables(d->wind_table, d->wind_rows, d->ca_table, d->ca_rows)) return fail(GEL_ERR_ARG, why);
  for (int i = 0; i < d->num_sections; i++)
    if (d->num_nodes[i] < 2) return fail(GEL_ERR_ARG, "every phase needs >= 2 LGR nodes (nodes_LGR requires n >= 2)");
  // device == GEL_DEVICE_NONE: a host-only handle (dims, LGR data, sparsity pattern, constant values,
  // work partition) that can describe the problem but never evaluate it -- there is no CPU fallback.
  if (d->device != GEL_DEVICE_NONE) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
      return fail(GEL_ERR_HIP, "no HIP device: the engine has no CPU fallback");
    if (d->device < 0 || d->device >= ndev) return fail(GEL_ERR_ARG, "device ordinal out of range");
  }

  gel_problem* p = new gel_problem();
  p->device = d->device;
  p->um = d->unit_mass; p->up = d->unit_position; p->uv = d->unit_velocity; p->uu = d->unit_u; p->ut = d->unit_t;
  p->dx = d->dx;
  // Position sweeps in exact-difference form and the t0 / tf columns in closed form (gel_rhs_parts.h pos_delta, gel_eval_kernel.h),
  // unless the caller asks for the reference's recomputing sweeps or the step dx * unit_position is beyond the 1 m the
  // truncated series of the difference form are sized for (the reference's own dx = 1e-8 gives 0.064 m).
  p->fd_recompute = ((d->flags & GEL_FLAG_FD_RECOMPUTE) != 0) || !(std::fabs(d->dx * d->unit_position) <= 1.0);
  p->barC20 = (d->barC20 == 0.0) ? -0.484165371736e-3 : d->barC20;
  const int S = d->num_sections;
  int N = 0;
  size_t offD = 0, offT = 0;
  int64_t V = 0;
  p->ph.resize(S);
  for (int i = 0; i < S; i++) {
    HostPhase& h = p->ph[i];
    h.n = d->num_nodes[i]; h.ua = N; h.xa = N + i; N += h.n;
    h.thrust = d->thrust[i]; h.massflow = d->massflow[i]; h.area = d->reference_area[i]; h.nozzle = d->nozzle_area[i];
    h.air = (h.area != 0.0); h.air_fd = (h.area > 0.0);
    h.t_fd = h.air_fd && p->fd_recompute;
    h.engine_on = d->engine_on[i] != 0; h.hold = d->attitude_hold[i] != 0;
    h.s_vv = 15; h.s_vq = h.air_fd ? 24 : 15; h.s_vt = h.s_vq + 12; h.s_qq = h.s_vt + (h.t_fd ? 6 : 3);
    h.q_fd = !h.hold && p->fd_recompute;   // quaternion kinematics: finite differences (20 slots) or closed form (10)
    h.K = h.s_qq + (h.hold ? 0 : (h.q_fd ? 20 : 10));
    h.voff = V; V += (int64_t)h.K * h.n + 1;  // node slots + the phase's pos/velocity diagonal scalar
    const size_t nd = (size_t)h.n * (h.n + 1);
    if (d->D && d->tau) {
      h.D.assign(d->D + offD, d->D + offD + nd);
      h.tau.assign(d->tau + offT, d->tau + offT + h.n);
    } else {
      int same = -1;
      for (int j = 0; j < i; j++) if (p->ph[j].n == h.n) { same = j; break; }
      if (same >= 0) { h.D = p->ph[same].D; h.tau = p->ph[same].tau; }
      else lgr_diffmat_ld(h.n, h.D, h.tau);
    }
    offD += nd; offT += h.n;
  }
  const int M = N + S;
  gel_dims& dm = p->dims;
  dm.S = S; dm.N = N; dm.M = M; dm.num_vars = 11 * M + 2 * N + S + 1;
  dm.num_rows[0] = N; dm.num_rows[1] = 3 * N; dm.num_rows[2] = 3 * N; dm.num_rows[3] = 4 * N;
  const int64_t T = S + 1;
  const int64_t rk[13] = {1, 1, 3, 3, 3, 3, 3, 3, 3, 3, 4, 4, 4};
  const int64_t ck[13] = {M, T, 3 * M, 3 * M, T, M, 3 * M, 3 * M, 4 * M, T, 4 * M, 2 * N, T};
  int64_t tot = 0;
  for (int b = 0; b < GEL_NUM_BLOCKS; b++) {
    int64_t s = 0;
    for (int i = 0; i < S; i++) s += phase_block_nnz(p->ph[i], b);
    dm.block_nnz[b] = s;
    dm.block_shape[b][0] = rk[b] * N;
    dm.block_shape[b][1] = ck[b];
    p->block_off[b] = tot;
    tot += s;
  }
  p->block_off[GEL_NUM_BLOCKS] = tot;
  dm.total_nnz = tot;
  dm.num_var_entries = V;
  {
    // A_min = read x once + write the residual once + write every x-dependent value the reference
    // computes once: V_ref = sum over phases of n * (9 [pos] + 39 | 30 [vel with | without velocity / time FD] + 32 [free
    // attitude]).  The engine's compact vector holds the DISTINCT values among them (stored_bytes).
    int64_t Vref = 0;
    for (int i = 0; i < S; i++) Vref += (int64_t)p->ph[i].n * (9 + (p->ph[i].air_fd ? 39 : 30) + (p->ph[i].hold ? 0 : 32));
    dm.algorithmic_bytes = 8 * ((int64_t)dm.num_vars + 11 * (int64_t)N + Vref);
    dm.stored_bytes = 8 * (11 * (int64_t)N + V);
  }

  // pattern-derived host arrays
  p->cval.assign((size_t)tot, 0.0);
  p->src.assign((size_t)tot, -1);
  p->var_idx.assign((size_t)V, -1);
  struct Ref { int64_t slot, f; int kind; int64_t occ; };
  std::vector<Ref> refs;
  walk_pattern(*p, [&](int blk, int64_t k, int32_t, int32_t, int kind, double cv, int64_t slot) {
    const int64_t f = p->block_off[blk] + k;
    if (kind == 0) { p->cval[(size_t)f] = cv; return; }
    p->src[(size_t)f] = (kind == 1) ? (int32_t)slot : (int32_t)(-2 - slot);
    if (kind == 1 && p->var_idx[(size_t)slot] < 0) p->var_idx[(size_t)slot] = f;
    refs.push_back({slot, f, kind, 0});
  });
  for (int64_t s = 0; s < V; s++)
    if (p->var_idx[(size_t)s] < 0) { delete p; return fail(GEL_ERR_ARG, "internal: compact slot without a COO entry"); }
  {
    // The gather map as constant-stride runs for the host scatter.  A per-node slot is used once or twice (a t0
    // column and its negated tf column): its k-th use over consecutive nodes is one run.  A phase scalar is used
    // 3n times: its uses in full order are one run with source stride 0.
    std::vector<char> is_scalar((size_t)V, 0);
    for (int i = 0; i < S; i++) is_scalar[(size_t)(p->ph[i].voff + (int64_t)p->ph[i].K * p->ph[i].n)] = 1;
    std::stable_sort(refs.begin(), refs.end(), [](const Ref& a, const Ref& b) { return a.slot < b.slot; });
    for (size_t i = 0; i < refs.size(); i++)
      refs[i].occ = (i > 0 && refs[i - 1].slot == refs[i].slot && !is_scalar[(size_t)refs[i].slot]) ? refs[i - 1].occ + 1 : 0;
    std::stable_sort(refs.begin(), refs.end(), [](const Ref& a, const Ref& b) { return a.occ < b.occ; });  // (occ, slot, f)
    for (size_t i = 0; i < refs.size();) {
      size_t j = i + 1;
      const bool sc = is_scalar[(size_t)refs[i].slot] != 0;
      const int64_t ds = (j < refs.size()) ? refs[j].f - refs[i].f : 1;
      while (j < refs.size() && refs[j].kind == refs[i].kind && refs[j].occ == refs[i].occ &&
             refs[j].slot == refs[j - 1].slot + (sc ? 0 : 1) && refs[j].f - refs[j - 1].f == ds &&
             (is_scalar[(size_t)refs[j].slot] != 0) == sc)
        j++;
      p->var_runs.push_back({refs[i].f, ds, refs[i].slot, sc ? 0 : 1, (int64_t)(j - i), refs[i].kind == 1 ? 1.0 : -1.0});
      i = j;
    }
  }

  if (!p->fd_recompute) {
    // COO-direct groups: where the runs of the all-x-dependent blocks start in the full value vector (the emission order of
    // walk_pattern() = the reference's), checked against the gather map itself -- any surprise switches the mode off
    int64_t kb[GEL_NUM_BLOCKS] = {0};
    std::vector<int32_t> tab((size_t)8 * S, -1);
    bool ok = tot < (int64_t)1 << 31;
    for (int i = 0; i < S && ok; i++) {
      const HostPhase& h = p->ph[i];
      const int n = h.n;
      auto at = [&](int b) { return p->block_off[b] + kb[b]; };
      int32_t* t = &tab[(size_t)8 * i];
      t[0] = (int32_t)at(3); t[1] = (int32_t)at(4); t[2] = (int32_t)at(5); t[3] = (int32_t)at(6); t[4] = (int32_t)at(8); t[5] = (int32_t)at(9);
      t[6] = h.hold ? -1 : (int32_t)at(11); t[7] = h.hold ? -1 : (int32_t)at(12);
      // spot checks against the map: node j, component c of group g sits at t[g] + w (n k + j) + c and takes slot(g, k, c) of node j
      auto expect = [&](int64_t f, int slot, int j, int sign) {
        const int64_t s = compact_index(h, slot, j);
        ok = ok && f >= 0 && f < tot && p->src[(size_t)f] == (sign > 0 ? (int32_t)s : (int32_t)(-2 - s));
      };
      for (int j : {0, n - 1})
        for (int c = 0; c < 3; c++) {
          ok = ok && p->src[(size_t)(t[0] + 3 * j + c)] == (int32_t)(h.voff + (int64_t)h.K * n);
          expect(t[1] + 3 * j + c, 0 + c, j, +1); expect(t[1] + 3 * (n + j) + c, 0 + c, j, -1);
          expect(t[2] + 3 * j + c, 3 + c, j, +1);
          for (int k = 0; k < 3; k++) expect(t[3] + 3 * (n * k + j) + c, 6 + 3 * k + c, j, +1);
          for (int k = 0; k < 4; k++) expect(t[4] + 3 * (n * k + j) + c, h.s_vq + 3 * k + c, j, +1);
          expect(t[5] + 3 * j + c, h.s_vt + c, j, +1); expect(t[5] + 3 * (n + j) + c, h.s_vt + c, j, -1);
        }
      if (!h.hold)
        for (int j : {0, n - 1})
          for (int c = 0; c < 4; c++) {
            // quat/u: the kernel's COO form writes {C2, C3, -C0, -C1 | C3, -C2, C1, -C0} with slots s_qq + 2 .. 5 = -C0, C1, C2, C3
            static const int which[2][4] = {{2, 3, 0, 1}, {3, 2, 1, 0}}, sg[2][4] = {{+1, +1, -1, -1}, {+1, -1, +1, -1}};
            for (int k = 0; k < 2; k++) expect(t[6] + 4 * (n * k + j) + c, h.s_qq + 2 + which[k][c], j, which[k][c] == 0 ? -sg[k][c] : sg[k][c]);
            expect(t[7] + 4 * j + c, h.s_qq + 6 + c, j, +1); expect(t[7] + 4 * (n + j) + c, h.s_qq + 6 + c, j, -1);
          }
      for (int b = 0; b < GEL_NUM_BLOCKS; b++) kb[b] += phase_block_nnz(h, b);
    }
    if (ok) {
      p->coo_tab = tab;
      for (const gel_problem::Run& r : p->var_runs) {
        const int b = (int)(std::upper_bound(p->block_off, p->block_off + GEL_NUM_BLOCKS + 1, r.dst0) - p->block_off) - 1;
        if (b == 7 || b == 10) p->rest_runs.push_back(r);
      }
    }
  }

  for (int i = 0; i < S; i++)
    for (int j0 = 0; j0 < p->ph[i].n; j0 += 64) p->chunk_phase.push_back(i);
  if (p->device == GEL_DEVICE_NONE) {
    p->dev.nchunks = (int32_t)p->chunk_phase.size();
    *out = p;
    return GEL_OK;
  }

  // device side
  if (hipSetDevice(p->device) != hipSuccess) { delete p; return fail(GEL_ERR_HIP, "hipSetDevice failed"); }
  std::vector<gel::PhaseDev> dph(S);
  std::vector<int32_t> node_phase(N);
  std::vector<double> Dt, tau;
  for (int i = 0; i < S; i++) {
    const HostPhase& h = p->ph[i];
    gel::PhaseDev& q = dph[i];
    q.n = h.n; q.ua = h.ua; q.xa = h.xa; q.air = h.air; q.air_fd = h.air_fd; q.t_fd = h.t_fd; q.q_fd = h.q_fd; q.engine_on = h.engine_on; q.hold = h.hold;
    q.K = h.K; q.s_vv = h.s_vv; q.s_vq = h.s_vq; q.s_vt = h.s_vt; q.s_qq = h.s_qq;
    q.doff = (int32_t)Dt.size(); q.toff = (int32_t)tau.size(); q.voff = h.voff;
    q.thrust = h.thrust; q.massflow = h.massflow; q.area = h.area; q.nozzle = h.nozzle;
    q.mf_um = -h.massflow / p->um;
    for (int c = 0; c <= h.n; c++)
      for (int j = 0; j < h.n; j++) Dt.push_back(h.D[(size_t)j * (h.n + 1) + c]);
    tau.insert(tau.end(), h.tau.begin(), h.tau.end());
    for (int j = 0; j < h.n; j++) node_phase[h.ua + j] = i;
  }
  const std::vector<double> tables = gel::build_tables(d->wind_table, d->wind_rows, d->ca_table, d->ca_rows);
  // Work items, and D laid out the way the matrix pipe consumes it.  A operand of v_mfma_f64_16x16x4_f64:
  // lane l holds A[row l & 15][k l >> 4]; a wavefront covers 64 nodes = 4 row tiles, and k-step ks covers
  // columns 4 ks .. 4 ks + 3 of D.  Dsw[((item_off + ks) * 64 + l) * 4 + t] = D[j0 + 16 t + (l & 15)][4 ks + (l >> 4)]
  // (0 beyond column n; rows beyond the phase repeat its last row: computed, never read back), so that one lane
  // fetches its four tile operands of a k-step with a single 32-byte access at a fixed stride.
  std::vector<int4> chunks;
  std::vector<double> Dsw, Dst;
  for (int i = 0; i < S; i++) {
    const HostPhase& h = p->ph[i];
    const int n = h.n, ksteps = (n + 4) >> 2;
    for (int j0 = 0; j0 < n; j0 += 64) {
      // .w: where the chunk stands in its phase's run and how long the run is (the cooperative launch order keeps the chunks
      // of a phase and a group of vectors together, gel_eval_kernel.h)
      chunks.push_back(make_int4(i, j0, (int)(Dsw.size() / 4), ((j0 / 64) << 16) | ((h.n + 63) / 64)));
      for (int ks = 0; ks < ksteps; ks++)
        for (int l = 0; l < 64; l++)
          for (int t = 0; t < 4; t++) {
            const int k = 4 * ks + (l >> 4), row = std::min(j0 + 16 * t + (l & 15), n - 1);
            Dsw.push_back(k <= n ? h.D[(size_t)row * (n + 1) + k] : 0.0);
          }
      // Dst[((item_off + ks) * 4 + t) * 64 + l]: the same element, row tile t contiguous over the lanes
      for (int ks = 0; ks < ksteps; ks++)
        for (int t = 0; t < 4; t++)
          for (int l = 0; l < 64; l++) {
            const int k = 4 * ks + (l >> 4), row = std::min(j0 + 16 * t + (l & 15), n - 1);
            Dst.push_back(k <= n ? h.D[(size_t)row * (n + 1) + k] : 0.0);
          }
    }
  }

  // launch order of a whole evaluation: aerodynamic phases (the long chain, 4x + sweeps) before NoAir
  // ones, free attitude before held -- the same weights gelato_amd/parallel.py balances shards with
  std::vector<int4> sorted_chunks = chunks;
  auto weight = [&](const int4& c) { return (p->ph[c.x].air ? 10.0 : 1.5) + (p->ph[c.x].hold ? 0.0 : 0.5); };
  std::stable_sort(sorted_chunks.begin(), sorted_chunks.end(), [&](const int4& a, const int4& b) { return weight(a) > weight(b); });

  // the phase-ordered list serves shard launches over ARBITRARY work-item ranges: no runs there
  std::vector<int4> shard_chunks = chunks;
  for (int4& c : shard_chunks) c.w = 1;
  int rc = GEL_OK;
  if ((rc = upload(&p->d_chunks, shard_chunks)) || (rc = upload(&p->d_chunks_sorted, sorted_chunks)) || (rc = upload(&p->d_Dsw, Dsw)) || (rc = upload(&p->d_Dst, Dst)) || (rc = upload(&p->d_phases, dph)) || (rc = upload(&p->d_node_phase, node_phase)) || (rc = upload(&p->d_Dt, Dt)) ||
      (rc = upload(&p->d_tau, tau)) || (rc = upload(&p->d_tables, tables)) || (rc = upload(&p->d_cval, p->cval)) ||
      (rc = upload(&p->d_src, p->src))) {
    gel_problem_destroy(p);
    return rc;
  }
  {
    std::vector<int32_t> vdst, vsrc;
    for (size_t i = 0; i < p->src.size(); i++)
      if (p->src[i] != -1) { vdst.push_back((int32_t)i); vsrc.push_back(p->src[i]); }
    p->nvar_entries = (int32_t)vdst.size();
    if (!vdst.empty() && ((rc = upload(&p->d_vdst, vdst)) || (rc = upload(&p->d_vsrc, vsrc)))) {
      gel_problem_destroy(p);
      return rc;
    }
    std::vector<int32_t> vline;
    for (int32_t d : vdst)
      if (vline.empty() || vline.back() != d / 8) vline.push_back(d / 8);
    p->nvar_lines = (int32_t)vline.size();
    if (!vline.empty() && (rc = upload(&p->d_vline, vline))) {
      gel_problem_destroy(p);
      return rc;
    }
    if (!p->coo_tab.empty() && (rc = upload(&p->d_coo, p->coo_tab))) {
      gel_problem_destroy(p);
      return rc;
    }
  }
  {
    // one-phase sub-problems for the phase-by-phase forward difference (gel_jac_fd)
    std::vector<gel::PhaseDev> sub = dph;
    std::vector<int4> subchunks = chunks;
    std::vector<int32_t> colmap;
    int c0 = 0;
    for (int i = 0; i < S; i++) {
      sub[i].ua = 0; sub[i].xa = 0; sub[i].voff = 0;
      p->sub_chunk0.push_back(c0);
      int cnt = 0;
      while (c0 + cnt < (int)chunks.size() && chunks[c0 + cnt].x == i) { subchunks[c0 + cnt].x = 0; cnt++; }
      p->sub_nchunks.push_back(cnt);
      c0 += cnt;
      p->sub_col0.push_back((int32_t)colmap.size());
      phase_columns(p, i, colmap);
    }
    if ((rc = upload(&p->d_subphases, sub)) || (rc = upload(&p->d_subchunks, subchunks)) || (rc = upload(&p->d_colmap, colmap))) {
      gel_problem_destroy(p);
      return rc;
    }
  }
  if (hipMalloc((void**)&p->d_flag, 4) != hipSuccess || hipMemset(p->d_flag, 0, 4) != hipSuccess ||
      hipMalloc((void**)&p->d_done, 4) != hipSuccess || hipMemset(p->d_done, 0, 4) != hipSuccess ||
      hipHostMalloc((void**)&p->h_done, 4) != hipSuccess ||
      hipHostMalloc((void**)&p->h_flag, 4) != hipSuccess || hipStreamCreate(&p->stream) != hipSuccess) {
    gel_problem_destroy(p);
    return fail(GEL_ERR_HIP, "device allocation failed");
  }
  *p->h_flag = 0;
  gel::ProblemDev& dv = p->dev;
  dv.S = S; dv.N = N; dv.M = M; dv.nvars = dm.num_vars; dv.Kw = d->wind_rows; dv.Kc = d->ca_rows; dv.V = V;
  dv.phases = p->d_phases; dv.node_phase = p->d_node_phase; dv.Dt = p->d_Dt; dv.tau = p->d_tau; dv.tables = p->d_tables;
  dv.flag = p->d_flag;
  dv.nchunks = (int32_t)chunks.size(); dv.chunks = p->d_chunks_sorted; dv.Dsw = p->d_Dsw; dv.Dst = p->d_Dst;
  dv.park_off = (int32_t)((tables.size() + 1) / 2 * 2);
  {
    // D.X path.  fp64 MFMA and fp64 VALU instructions share the SIMD's fp64 datapath on this part (measured: their busy
    // times add up), so the matrix pipe is not free throughput; what it buys is fewer issue slots and the
    // cooperative form: four decision vectors side by side (44 of 48 tile columns used, a quarter of the D traffic), and --
    // when every phase fits 32 lanes -- two decision vectors per wavefront, which no wavefront dot-product form offers.
    // Same-box A/B, evals/s at B = 32768, matrix pipe vs VALU (tools/dx_paths.sh): 3x8 97.2 M vs 53.3 M, 3x16 77.7 M vs
    // 49.1 M, 3x32 55.2 M vs 40.1 M, 6x64 13.9 M vs 12.7 M; residual-only 3x8 265 M vs 163 M, 3x16 217 M vs 129 M, 3x32
    // 161 M vs 91 M, 6x64 31.3 M vs 21.4 M.  Between 33 and 63 nodes per phase the matrix pipe wins as well (one vector per
    // wavefront).  Only a mesh that mixes phases above 32 nodes with many tiny ones would prefer the dot-products; none
    // of the reference's configurations does.  GEL_FLAG_DX_VALU / _MFMA force a path.
    int nmax = 0;
    for (int i = 0; i < S; i++) nmax = std::max(nmax, p->ph[i].n);
    dv.use_mfma = (d->flags & GEL_FLAG_DX_VALU) ? 0 : 1;
    dv.pack = (nmax <= 32) && !(d->flags & GEL_FLAG_NO_PACK);
    dv.longp = (nmax >= gel::kLongPhaseFrom) ? 1 : 0;
  }
  dv.fd_recompute = p->fd_recompute ? 1 : 0;
  {
    int nmax = 0;
    for (int i = 0; i < S; i++) nmax = std::max(nmax, p->ph[i].n);
    // measured (same box): 3 x 32 +4 % (fused and residual-only), 6 x 64 -1..2 %: only the small-phase meshes, whose rows share most lines
    dv.vmajor = (nmax <= 32 && !(d->flags & GEL_FLAG_ITEM_MAJOR)) ? 1 : 0;
  }
  dv.um = p->um; dv.up = p->up; dv.uv = p->uv; dv.uu = p->uu; dv.ut = p->ut; dv.dx = p->dx; dv.barC20 = p->barC20;
  dv.inv_uv = 1.0 / p->uv; dv.inv_dx = 1.0 / p->dx; dv.kpt = p->uv * p->ut / 2.0 / p->up; dv.hT = p->ut * 0.5;
  *out = p;
  return GEL_OK;
}

int gel_problem_destroy(gel_problem* p) {
  if (!p) return GEL_OK;
  if (p->device == GEL_DEVICE_NONE) { delete p; return GEL_OK; }
  hipSetDevice(p->device);
  if (p->stream) { hipStreamSynchronize(p->stream); hipStreamDestroy(p->stream); }
  hipFree(p->d_phases); hipFree(p->d_node_phase); hipFree(p->d_chunks); hipFree(p->d_chunks_sorted); hipFree(p->d_Dsw); hipFree(p->d_Dst); hipFree(p->d_Dt); hipFree(p->d_tau); hipFree(p->d_tables);
  hipFree(p->d_vdst); hipFree(p->d_vsrc); hipFree(p->d_vline); hipFree(p->d_coo);
  if (p->h_full) hipHostFree(p->h_full);
  if (p->cb_res) hipHostFree(p->cb_res);
  for (int i = 0; i < 2; i++) if (p->cb_x[i]) hipHostFree(p->cb_x[i]);
  hipFree(p->d_cval); hipFree(p->d_src); hipFree(p->d_flag); hipFree(p->d_unit_base); hipFree(p->d_shard_pos);
  hipFree(p->d_aero_nodes); hipFree(p->d_aero_x); hipFree(p->d_aero_out); hipFree(p->d_aero_ph); hipFree(p->d_aero_part_nodes[0]); hipFree(p->d_aero_part_nodes[1]);
  free_slots(p);
  hipFree(p->d_x); hipFree(p->d_res); hipFree(p->d_jv);
  if (p->h_x) hipHostFree(p->h_x);
  if (p->h_res) hipHostFree(p->h_res);
  if (p->h_jv) hipHostFree(p->h_jv);
  if (p->h_flag) hipHostFree(p->h_flag);
  if (p->h_aero) hipHostFree(p->h_aero);
  if (p->h_rows) hipHostFree(p->h_rows);
  hipFree(p->d_lin_rows); hipFree(p->d_fn_rows); hipFree(p->d_rows_x); hipFree(p->d_rows_out);
  hipFree(p->jfd_x); hipFree(p->jfd_Xp); hipFree(p->jfd_res); hipFree(p->jfd_J);
  hipFree(p->d_done); if (p->h_done) hipHostFree(const_cast<int32_t*>(p->h_done));
  hipFree(p->d_subphases); hipFree(p->d_subchunks); hipFree(p->d_colmap);
  delete p;
  return GEL_OK;
}

int gel_problem_dims(const gel_problem* p, gel_dims* out) {
  if (!p || !out) return fail(GEL_ERR_ARG, "null argument");
  *out = p->dims;
  return GEL_OK;
}

int gel_problem_D(const gel_problem* p, int32_t i, double* D) {
  if (!p || !D || i < 0 || i >= (int)p->ph.size()) return fail(GEL_ERR_ARG, "Index out of range");
  std::memcpy(D, p->ph[i].D.data(), p->ph[i].D.size() * 8);
  return GEL_OK;
}

int gel_problem_tau(const gel_problem* p, int32_t i, double* tau) {
  if (!p || !tau || i < 0 || i >= (int)p->ph.size()) return fail(GEL_ERR_ARG, "Index out of range");
  std::memcpy(tau, p->ph[i].tau.data(), p->ph[i].tau.size() * 8);
  return GEL_OK;
}

int gel_pattern(const gel_problem* p, int32_t block, int32_t* rows, int32_t* cols) {
  if (!p || !rows || !cols || block < 0 || block >= GEL_NUM_BLOCKS) return fail(GEL_ERR_ARG, "bad block");
  walk_pattern(*p, [&](int blk, int64_t k, int32_t r, int32_t c, int, double, int64_t) {
    if (blk == block) { rows[k] = r; cols[k] = c; }
  });
  return GEL_OK;
}

int gel_pattern_all(const gel_problem* p, int32_t* rows_full, int32_t* cols_full) {
  if (!p || !rows_full || !cols_full) return fail(GEL_ERR_ARG, "null argument");
  walk_pattern(*p, [&](int blk, int64_t k, int32_t r, int32_t c, int, double, int64_t) {
    rows_full[p->block_off[blk] + k] = r;
    cols_full[p->block_off[blk] + k] = c;
  });
  return GEL_OK;
}

int gel_const_values(const gel_problem* p, double* vals_full) {
  if (!p || !vals_full) return fail(GEL_ERR_ARG, "null argument");
  std::memcpy(vals_full, p->cval.data(), p->cval.size() * 8);
  return GEL_OK;
}

int gel_var_index(const gel_problem* p, int64_t* idx) {
  if (!p || !idx) return fail(GEL_ERR_ARG, "null argument");
  std::memcpy(idx, p->var_idx.data(), p->var_idx.size() * 8);
  return GEL_OK;
}

int gel_full_source(const gel_problem* p, int32_t* src) {
  if (!p || !src) return fail(GEL_ERR_ARG, "null argument");
  std::memcpy(src, p->src.data(), p->src.size() * sizeof(int32_t));
  return GEL_OK;
}

int gel_pinned_buffers(gel_problem* p, double** res, double** vals_full, double** x0, double** x1) {
  if (!p) return fail(GEL_ERR_ARG, "null argument");
  NEED_DEVICE(p);
  if (int rc = ensure_full(p)) return rc;
  if (res) *res = p->cb_res;
  if (vals_full) *vals_full = p->h_full;
  if (x0) *x0 = p->cb_x[0];
  if (x1) *x1 = p->cb_x[1];
  return GEL_OK;
}

int gel_eval_residual(gel_problem* p, const double* x, double* res) {
  if (!p || !x || !res) return fail(GEL_ERR_ARG, "null argument");
  const bool own = p->cb_res && res == p->cb_res;   // the caller named the handle's pinned vector: the kernel writes it, no copy
  const int rc = run_host(p, 1, x, true, false, own ? p->cb_res : nullptr);
  if (rc < 0) return rc;
  if (!own) std::memcpy(res, p->h_res, (size_t)11 * p->dims.N * 8);
  return rc;
}

int gel_eval_jacobian(gel_problem* p, const double* x, double* vals_full, int32_t fill_constants) {
  return gel_eval(p, x, nullptr, vals_full, fill_constants);
}

int gel_eval(gel_problem* p, const double* x, double* res, double* vals_full, int32_t fill_constants) {
  if (!p || !x || !vals_full) return fail(GEL_ERR_ARG, "null argument");
  NEED_DEVICE(p);
  bool coo = coo_direct(p);
  if (coo) { if (int rc0 = ensure_full(p)) return rc0; }
  const bool own = res && p->cb_res && res == p->cb_res;
  const int rc = run_host(p, 1, x, res != nullptr, true, own ? p->cb_res : nullptr, &coo);
  if (rc < 0) return rc;
  if (res && !own) std::memcpy(res, p->h_res, (size_t)11 * p->dims.N * 8);
  if (coo) finish_full(p, vals_full, fill_constants);
  else scatter_full(p, p->h_jv, vals_full, fill_constants);
  return rc;
}

int gel_eval_batch(gel_problem* p, int32_t B, const double* x, double* res, double* jvar) {
  if (!p || !x || B < 1 || (!res && !jvar)) return fail(GEL_ERR_ARG, "bad argument");
  NEED_DEVICE(p);
  // more than one staging slot's worth: sub-batches through the two-slot pipeline
  if ((size_t)B * 8 * (size_t)std::max<int64_t>(p->dims.num_var_entries, 1) > kPipeBytes) return run_host_pipelined(p, B, x, res, jvar);
  const int rc = run_host(p, B, x, res != nullptr, jvar != nullptr);
  if (rc < 0) return rc;
  if (res) std::memcpy(res, p->h_res, (size_t)B * 11 * p->dims.N * 8);
  if (jvar) std::memcpy(jvar, p->h_jv, (size_t)B * p->dims.num_var_entries * 8);
  return rc;
}

int gel_eval_batch_device(gel_problem* p, int32_t B, const double* d_x, double* d_res, double* d_jvar, void* stream) {
  if (!p || !d_x || B < 1 || (!d_res && !d_jvar)) return fail(GEL_ERR_ARG, "bad argument");
  NEED_DEVICE(p);
  HIPCHK(gel::launch_eval(p->dev, B, d_x, d_res, d_jvar, stream ? (hipStream_t)stream : p->stream));
  return GEL_OK;
}

int gel_eval_shard_units_device(gel_problem* p, int32_t B, const double* d_x, double* d_res, double* d_jvar,
                                int32_t unit_begin, int32_t unit_count, void* stream) {
  if (!p || !d_x || B < 1 || !d_jvar) return fail(GEL_ERR_ARG, "bad argument (the unit form always writes Jacobian values)");
  NEED_DEVICE(p);
  const int32_t total = 4 * (int32_t)p->chunk_phase.size();
  if (unit_begin < 0 || unit_count < 0 || unit_begin + unit_count > total)
    return fail(GEL_ERR_ARG, "unit range out of bounds");
  if (unit_count == 0) return GEL_OK;
  gel::ProblemDev dv = p->dev;
  dv.chunks = p->d_chunks;  // unit ids refer to the phase-ordered list
  dv.chunk0 = 0;
  dv.unit0 = unit_begin;
  dv.nunits = unit_count;
  HIPCHK(gel::launch_eval(dv, B, d_x, d_res, d_jvar, stream ? (hipStream_t)stream : p->stream));
  return GEL_OK;
}

int gel_num_chunks(const gel_problem* p, int32_t* nchunks) {
  if (!p || !nchunks) return fail(GEL_ERR_ARG, "null argument");
  *nchunks = (int32_t)p->chunk_phase.size();
  return GEL_OK;
}

int gel_chunk_phase(const gel_problem* p, int32_t* phase) {
  if (!p || !phase) return fail(GEL_ERR_ARG, "null argument");
  std::memcpy(phase, p->chunk_phase.data(), p->chunk_phase.size() * sizeof(int32_t));
  return GEL_OK;
}

int gel_launch_info(const gel_problem* p, int32_t B, int32_t want_res, int32_t want_jac, int32_t* info) {
  if (!p || !info || B < 1) return fail(GEL_ERR_ARG, "bad argument");
  NEED_DEVICE(p);
  const gel::EvalForm f = gel::eval_form(p->dev, B, want_res != 0, want_jac != 0);
  info[0] = f.jac; info[1] = f.mfma; info[2] = f.split; info[3] = (int32_t)std::min<long long>(f.waves, INT32_MAX);
  info[4] = f.pack;
  return GEL_OK;
}

int gel_unit_owner(const gel_problem* p, int32_t* res_owner, int32_t* jvar_owner) {
  if (!p || !res_owner || !jvar_owner) return fail(GEL_ERR_ARG, "null argument");
  // unit = 4 * work item + part (phase-ordered work items): part 0 writes the residual rows of the item's nodes and
  // every compact value of those nodes except the position-sweep columns of an aerodynamic phase, which parts 1..3
  // write (one column = three slots each); the phase scalar belongs to part 0 of the phase's first work item
  const int N = p->dims.N;
  int item = 0;
  for (size_t i = 0; i < p->ph.size(); i++) {
    const HostPhase& h = p->ph[i];
    for (int j0 = 0; j0 < h.n; j0 += 64, item++) {
      const int j1 = std::min(j0 + 64, h.n);
      for (int j = j0; j < j1; j++) {
        const int g = h.ua + j;
        res_owner[g] = 4 * item;
        for (int c = 0; c < 3; c++) { res_owner[N + 3 * g + c] = 4 * item; res_owner[4 * N + 3 * g + c] = 4 * item; }
        for (int c = 0; c < 4; c++) res_owner[7 * N + 4 * g + c] = 4 * item;
        for (int s = 0; s < h.K; s++) {
          const int part = (h.air && s >= 6 && s < 15) ? 1 + (s - 6) / 3 : 0;
          jvar_owner[compact_index(h, s, j)] = 4 * item + part;
        }
      }
      if (j0 == 0) jvar_owner[h.voff + (int64_t)h.K * h.n] = 4 * item;
    }
  }
  return GEL_OK;
}

// Packed unit-shard exchange: the layout every rank agrees on.  Rank r holds units [unit_begin[r], unit_begin[r + 1]); the block
// of a unit inside its rank's per-vector block: the compact slots it owns as [slot][node of the chunk] (part 0 of an aerodynamic
// phase: all but slots 6 .. 14; part k > 0: slots 6 + 3 (k - 1) .. + 2), the phase scalar behind them (part 0 of a phase's first
// chunk), then (part 0) the residual rows mass | position 3 | velocity 3 | quaternion 4 of the chunk's nodes -- what
// eval_body<.., SPLIT> writes when ProblemDev::shard_width != 0.
int gel_shard_plan(gel_problem* p, int32_t nranks, const int32_t* unit_begin, int64_t* width, int64_t* res_pos, int64_t* jvar_pos) {
  if (!p || nranks < 1 || !unit_begin || !width) return fail(GEL_ERR_ARG, "bad argument");
  const int32_t nunits = 4 * (int32_t)p->chunk_phase.size();
  if (unit_begin[0] != 0 || unit_begin[nranks] != nunits) return fail(GEL_ERR_ARG, "the unit ranges must cover every unit");
  for (int r = 0; r < nranks; r++)
    if (unit_begin[r + 1] < unit_begin[r]) return fail(GEL_ERR_ARG, "the unit ranges must be ordered");
  const int N = p->dims.N;
  const int64_t V = p->dims.num_var_entries;
  std::vector<int64_t> usize(nunits, 0), pos((size_t)11 * N + V, -1);
  std::vector<int32_t> urank(nunits, 0);
  for (int r = 0; r < nranks; r++)
    for (int u = unit_begin[r]; u < unit_begin[r + 1]; u++) urank[u] = r;
  // sizes first (bases need them), then positions
  {
    int item = 0;
    for (const HostPhase& h : p->ph)
      for (int j0 = 0; j0 < h.n; j0 += 64, item++) {
        const int64_t nn = std::min(64, h.n - j0);
        const int skip = h.air ? 9 : 0;
        usize[4 * item] = (int64_t)(h.K - skip) * nn + (j0 == 0 ? 1 : 0) + 11 * nn;
        for (int k = 1; k < 4; k++) usize[4 * item + k] = h.air ? 3 * nn : 0;
      }
  }
  std::vector<int64_t> base(nunits, 0);
  int64_t w = 0;
  for (int r = 0; r < nranks; r++) {
    int64_t off = 0;
    for (int u = unit_begin[r]; u < unit_begin[r + 1]; u++) { base[u] = off; off += usize[u]; }
    w = std::max(w, off);
  }
  w = (w + 1) & ~(int64_t)1;   // 16-byte multiples: every rank's slice and every vector's block start aligned
  {
    int item = 0;
    for (const HostPhase& h : p->ph)
      for (int j0 = 0; j0 < h.n; j0 += 64, item++) {
        const int nn = std::min(64, h.n - j0);
        const int skip = h.air ? 9 : 0;
        const int64_t nj = (int64_t)(h.K - skip) * nn + (j0 == 0 ? 1 : 0);
        const int u0 = 4 * item;
        const int64_t b0 = (int64_t)urank[u0] * w + base[u0];
        for (int jl = 0; jl < nn; jl++) {
          const int j = j0 + jl, g = h.ua + j;
          pos[g] = b0 + nj + jl;
          for (int c = 0; c < 3; c++) { pos[N + 3 * g + c] = b0 + nj + nn + 3 * jl + c; pos[4 * N + 3 * g + c] = b0 + nj + 4 * nn + 3 * jl + c; }
          for (int c = 0; c < 4; c++) pos[7 * N + 4 * g + c] = b0 + nj + 7 * nn + 4 * jl + c;
          for (int sl = 0; sl < h.K; sl++) {
            const int part = (h.air && sl >= 6 && sl < 15) ? 1 + (sl - 6) / 3 : 0;
            const int u = u0 + part;
            const int local = part ? sl - (6 + 3 * (part - 1)) : (sl >= 6 ? sl - skip : sl);
            pos[(size_t)11 * N + compact_index(h, sl, j)] = (int64_t)urank[u] * w + base[u] + (int64_t)local * nn + jl;
          }
        }
        if (j0 == 0) pos[(size_t)11 * N + h.voff + (int64_t)h.K * h.n] = b0 + (int64_t)(h.K - skip) * nn;
      }
  }
  for (int64_t v : pos)
    if (v < 0) return fail(GEL_ERR_ARG, "shard plan: an output entry without an owner (internal)");
  p->shard_begin.assign(unit_begin, unit_begin + nranks + 1);
  p->unit_base = base;
  p->shard_width = w;
  *width = w;
  if (res_pos) std::memcpy(res_pos, pos.data(), sizeof(int64_t) * 11 * (size_t)N);
  if (jvar_pos) std::memcpy(jvar_pos, pos.data() + (size_t)11 * N, sizeof(int64_t) * (size_t)V);
  if (p->device >= 0) {
    HIPCHK(hipSetDevice(p->device));
    hipFree(p->d_unit_base); hipFree(p->d_shard_pos);
    p->d_unit_base = nullptr; p->d_shard_pos = nullptr;
    HIPCHK(hipMalloc(&p->d_unit_base, sizeof(int64_t) * (size_t)nunits));
    HIPCHK(hipMemcpy(p->d_unit_base, base.data(), sizeof(int64_t) * (size_t)nunits, hipMemcpyHostToDevice));
    HIPCHK(hipMalloc(&p->d_shard_pos, sizeof(int64_t) * pos.size()));
    HIPCHK(hipMemcpy(p->d_shard_pos, pos.data(), sizeof(int64_t) * pos.size(), hipMemcpyHostToDevice));
  }
  return GEL_OK;
}

// The plan is state of the handle and the next gel_shard_plan call replaces it, so the caller states the plan its buffer was sized
// for (nranks, width): a buffer allocated as [old nranks][B][old width] must never be written with the new plan's offsets.
static int check_plan(const gel_problem* p, int32_t nranks, int64_t width) {
  if (p->shard_width <= 0) return fail(GEL_ERR_ARG, "gel_shard_plan has not been called on this handle");
  if (nranks != (int32_t)p->shard_begin.size() - 1 || width != p->shard_width)
    return fail(GEL_ERR_ARG, "the handle's shard plan is not the one this buffer was sized for (gel_shard_plan was called again: plan once more, or use one handle per plan)");
  return GEL_OK;
}

int gel_eval_shard_packed_device(gel_problem* p, int32_t B, const double* d_x, double* d_out, int32_t rank, int32_t nranks_expected,
                                 int64_t width_expected, void* stream) {
  if (!p || !d_x || !d_out || B < 1) return fail(GEL_ERR_ARG, "bad argument");
  NEED_DEVICE(p);
  if (int rc = check_plan(p, nranks_expected, width_expected)) return rc;
  if (!p->d_unit_base) return fail(GEL_ERR_ARG, "gel_shard_plan has not been called on this handle");
  const int nranks = (int)p->shard_begin.size() - 1;
  if (rank < 0 || rank >= nranks) return fail(GEL_ERR_ARG, "rank outside the plan");
  const int32_t u0 = p->shard_begin[rank], cnt = p->shard_begin[rank + 1] - u0;
  if (cnt == 0) return GEL_OK;
  gel::ProblemDev dv = p->dev;
  dv.chunks = p->d_chunks;  // unit ids refer to the phase-ordered list
  dv.chunk0 = 0;
  dv.unit0 = u0;
  dv.nunits = cnt;
  dv.shard_width = p->shard_width;
  dv.unit_base = p->d_unit_base;
  double* slice = d_out + (size_t)rank * (size_t)B * (size_t)p->shard_width;
  HIPCHK(gel::launch_eval(dv, B, d_x, slice, slice, stream ? (hipStream_t)stream : p->stream));
  return GEL_OK;
}

int gel_shard_unpack_device(gel_problem* p, int32_t B, const double* d_out, double* d_res, double* d_jvar, int32_t nranks_expected,
                            int64_t width_expected, void* stream) {
  if (!p || !d_out || B < 1 || (!d_res && !d_jvar)) return fail(GEL_ERR_ARG, "bad argument");
  NEED_DEVICE(p);
  if (int rc = check_plan(p, nranks_expected, width_expected)) return rc;
  if (!p->d_shard_pos) return fail(GEL_ERR_ARG, "gel_shard_plan has not been called on this handle");
  HIPCHK(gel::launch_shard_unpack(11 * p->dims.N, p->dims.num_var_entries, p->shard_width, B, p->d_shard_pos, d_out, d_res, d_jvar,
                                  stream ? (hipStream_t)stream : p->stream));
  return GEL_OK;
}

int gel_fill_full_device(gel_problem* p, int32_t B, double* d_jfull, void* stream) {
  if (!p || !d_jfull || B < 1) return fail(GEL_ERR_ARG, "bad argument");
  NEED_DEVICE(p);
  HIPCHK(gel::launch_fill_full(p->dims.total_nnz, B, p->d_cval, d_jfull, stream ? (hipStream_t)stream : p->stream));
  return GEL_OK;
}

int gel_update_full_device(gel_problem* p, int32_t B, const double* d_jvar, double* d_jfull, void* stream) {
  if (!p || !d_jvar || !d_jfull || B < 1) return fail(GEL_ERR_ARG, "bad argument");
  NEED_DEVICE(p);
  HIPCHK(gel::launch_update_full(p->dims.total_nnz, p->dims.num_var_entries, p->nvar_entries, B, p->d_vdst, p->d_vsrc,
                                 p->nvar_lines, p->d_vline, p->d_src, p->d_cval, d_jvar, d_jfull,
                                 stream ? (hipStream_t)stream : p->stream));
  return GEL_OK;
}

int gel_eval_full_device(gel_problem* p, int32_t B, const double* d_x, double* d_res, double* d_jvar, double* d_jfull, void* stream) {
  if (!p || !d_x || !d_jvar || !d_jfull || B < 1) return fail(GEL_ERR_ARG, "bad argument");
  NEED_DEVICE(p);
  hipStream_t s = stream ? (hipStream_t)stream : p->stream;
  gel::ProblemDev dv = p->dev;
  dv.cached_out = 1;   // the compact values are read again by the update below: kept in the caches when the launch fits them
  HIPCHK(gel::launch_eval(dv, B, d_x, d_res, d_jvar, s));
  HIPCHK(gel::launch_update_full(p->dims.total_nnz, p->dims.num_var_entries, p->nvar_entries, B, p->d_vdst, p->d_vsrc,
                                 p->nvar_lines, p->d_vline, p->d_src, p->d_cval, d_jvar, d_jfull, s));
  return GEL_OK;
}

int gel_expand_full_device(gel_problem* p, int32_t B, const double* d_jvar, double* d_jfull, void* stream) {
  if (!p || !d_jvar || !d_jfull || B < 1) return fail(GEL_ERR_ARG, "bad argument");
  NEED_DEVICE(p);
  HIPCHK(gel::launch_expand(p->dims.total_nnz, p->dims.num_var_entries, B, p->d_cval, p->d_src, d_jvar, d_jfull,
                            stream ? (hipStream_t)stream : p->stream));
  return GEL_OK;
}

int gel_sync(gel_problem* p, void* stream) {
  if (!p) return fail(GEL_ERR_ARG, "null argument");
  NEED_DEVICE(p);
  hipStream_t s = stream ? (hipStream_t)stream : p->stream;
  HIPCHK(hipMemcpyAsync(p->h_flag, p->d_flag, 4, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  if (*p->h_flag) {
    HIPCHK(hipMemsetAsync(p->d_flag, 0, 4, s));
    HIPCHK(hipStreamSynchronize(s));
    *p->h_flag = 0;
    return GEL_NONFINITE;
  }
  return GEL_OK;
}

// ---- generic forward difference, phase by phase (lib/jac_fd.py:29-62 on the four defect residuals) ----
// working set: x, the perturbed local vectors of the LARGEST phase (reused phase after phase), the residuals of every phase's
// perturbed vectors (kept while x stays the same: the four groups are asked for one after the other)
static int jfd_allocate(gel_problem* p) {
  if (p->jfd_x) return GEL_OK;
  const size_t nv = (size_t)p->dims.num_vars;
  size_t xp_max = 0, res_tot = 0;
  p->sub_res0.clear();
  for (int i = 0; i < p->dims.S; i++) {
    const size_t n = p->ph[i].n, nloc = 13 * n + 13;
    xp_max = std::max(xp_max, (nloc + 1) * nloc);
    p->sub_res0.push_back(res_tot);
    res_tot += (nloc + 1) * 11 * n;
  }
  if (hipMalloc((void**)&p->jfd_x, nv * 8) != hipSuccess || hipMalloc((void**)&p->jfd_Xp, xp_max * 8) != hipSuccess ||
      hipMalloc((void**)&p->jfd_res, res_tot * 8) != hipSuccess) {
    hipFree(p->jfd_x); hipFree(p->jfd_Xp); hipFree(p->jfd_res);
    p->jfd_x = p->jfd_Xp = p->jfd_res = nullptr;   // all or nothing: a later call must not find a partial set
    return fail(GEL_ERR_ALLOC, "gel_jac_fd: device allocation failed");
  }
  return GEL_OK;
}

// one residual evaluation per (phase, local column) of the vector at d_x: launches only
static int jfd_evaluate(gel_problem* p, const double* d_x, hipStream_t s) {
  for (int i = 0; i < p->dims.S; i++) {
    const int n = p->ph[i].n, nloc = 13 * n + 13;
    gel::ProblemDev dv = p->dev;   // the phase as a one-phase problem: same tables, D, tau; local index space
    dv.S = 1; dv.N = n; dv.M = n + 1; dv.nvars = nloc; dv.V = 0;
    dv.phases = p->d_subphases + i;
    dv.chunks = p->d_subchunks + p->sub_chunk0[i]; dv.nchunks = p->sub_nchunks[i]; dv.chunk0 = 0;
    HIPCHK(gel::launch_perturb_local(nloc, p->dx, d_x, p->d_colmap + p->sub_col0[i], p->jfd_Xp, s));
    HIPCHK(gel::launch_eval(dv, nloc + 1, p->jfd_Xp, p->jfd_res + p->sub_res0[i], nullptr, s));
  }
  return GEL_OK;
}

static inline int jfd_w(int group) { return (group == 0) ? 1 : (group == 3) ? 4 : 3; }   // rows per node of the group

// the group's quotients from the residuals of jfd_evaluate: dense [num_rows[group]][num_vars] (zeros, then every phase's block at its
// rows and mapped columns) or the blocks alone, one after the other, [w n_i][13 n_i + 13] each; -> doubles written
static int jfd_quotients(gel_problem* p, int group, int blocks, double* d_J, hipStream_t s, size_t* count) {
  const size_t nv = (size_t)p->dims.num_vars, nrows = (size_t)p->dims.num_rows[group];
  const int w = jfd_w(group);
  if (!blocks) HIPCHK(hipMemsetAsync(d_J, 0, nrows * nv * 8, s));
  size_t off = 0;
  for (int i = 0; i < p->dims.S; i++) {
    const int n = p->ph[i].n, nloc = 13 * n + 13;
    const int roff_loc = (group == 0) ? 0 : (group == 1) ? n : (group == 2) ? 4 * n : 7 * n;
    if (blocks)
      HIPCHK(gel::launch_quotient_local(nloc, 11 * n, roff_loc, w * n, p->dx, p->jfd_res + p->sub_res0[i], d_J + off, (long long)nloc, 0,
                                        nullptr, s));
    else
      HIPCHK(gel::launch_quotient_local(nloc, 11 * n, roff_loc, w * n, p->dx, p->jfd_res + p->sub_res0[i], d_J, (long long)nv,
                                        w * p->ph[i].ua, p->d_colmap + p->sub_col0[i], s));
    off += (size_t)w * n * nloc;
  }
  *count = blocks ? off : nrows * nv;
  return GEL_OK;
}

static size_t jfd_block_doubles(const gel_problem* p, int group) {
  size_t tot = 0;
  for (int i = 0; i < p->dims.S; i++) tot += (size_t)jfd_w(group) * p->ph[i].n * (13 * (size_t)p->ph[i].n + 13);
  return tot;
}

static int jfd_host(gel_problem* p, int32_t group, const double* x, double* J, int blocks) {
  if (!p || !x || !J || group < 0 || group >= GEL_NUM_GROUPS) return fail(GEL_ERR_ARG, "bad argument");
  NEED_DEVICE(p);
  HIPCHK(hipSetDevice(p->device));
  int rc = ensure_slots(p);  // the two pinned staging slots also carry J back to the caller
  if (rc) return rc;
  if ((rc = jfd_allocate(p))) return rc;
  const size_t nv = (size_t)p->dims.num_vars, nrows = (size_t)p->dims.num_rows[group];
  const size_t need = blocks ? jfd_block_doubles(p, group) : nrows * nv;
  if (p->jfd_J_cap < need) {
    hipFree(p->jfd_J);
    p->jfd_J = nullptr; p->jfd_J_cap = 0;
    HIPCHK(hipMalloc((void**)&p->jfd_J, need * 8));
    p->jfd_J_cap = need;
  }
  // the perturbed evaluations -- unless the same x was just differenced
  if (p->jfd_last_x.size() != nv || std::memcmp(p->jfd_last_x.data(), x, nv * 8) != 0) {
    p->jfd_last_x.clear();
    HIPCHK(hipMemcpyAsync(p->jfd_x, x, nv * 8, hipMemcpyHostToDevice, p->stream));
    if ((rc = jfd_evaluate(p, p->jfd_x, p->stream))) return rc;
    HIPCHK(hipMemcpyAsync(p->h_flag, p->d_flag, 4, hipMemcpyDeviceToHost, p->stream));
    HIPCHK(hipStreamSynchronize(p->stream));
    p->jfd_status = GEL_OK;
    if (*p->h_flag) { *p->h_flag = 0; HIPCHK(hipMemsetAsync(p->d_flag, 0, 4, p->stream)); p->jfd_status = GEL_NONFINITE; }
    p->jfd_last_x.assign(x, x + nv);
  }
  size_t total = 0;
  if ((rc = jfd_quotients(p, group, blocks, p->jfd_J, p->stream, &total))) return rc;
  HIPCHK(hipStreamSynchronize(p->stream));
  // J -> caller through the two pinned slots: D2H of piece i+1 overlaps the host copy of piece i
  const size_t piece = (size_t)p->pipe_evals * (size_t)std::max<int64_t>(p->dims.num_var_entries, 1);
  size_t pend_off[2] = {0, 0}, pend_n[2] = {0, 0};
  for (size_t off = 0, i = 0; off < total || pend_n[0] || pend_n[1]; i++) {
    gel_problem::Slot& sl = p->slot[i & 1];
    if (pend_n[i & 1]) {
      HIPCHK(hipStreamSynchronize(sl.stream));
      par_copy(J + pend_off[i & 1], sl.h_jv, pend_n[i & 1] * 8);
      pend_n[i & 1] = 0;
    }
    if (off < total) {
      const size_t n = std::min(piece, total - off);
      HIPCHK(hipMemcpyAsync(sl.h_jv, p->jfd_J + off, n * 8, hipMemcpyDeviceToHost, sl.stream));
      pend_off[i & 1] = off; pend_n[i & 1] = n;
      off += n;
    }
  }
  return p->jfd_status;
}

int gel_jac_fd(gel_problem* p, int32_t group, const double* x, double* J) { return jfd_host(p, group, x, J, 0); }

int gel_jac_fd_blocks(gel_problem* p, int32_t group, const double* x, double* blocks) { return jfd_host(p, group, x, blocks, 1); }

int gel_jac_fd_block_dims(const gel_problem* p, int32_t group, int64_t* rows, int64_t* cols, int64_t* row0, int64_t* offset) {
  if (!p || group < 0 || group >= GEL_NUM_GROUPS) return fail(GEL_ERR_ARG, "bad argument");
  int64_t off = 0;
  for (int i = 0; i < p->dims.S; i++) {
    const int64_t r = (int64_t)jfd_w(group) * p->ph[i].n, c = 13 * (int64_t)p->ph[i].n + 13;
    if (rows) rows[i] = r;
    if (cols) cols[i] = c;
    if (row0) row0[i] = (int64_t)jfd_w(group) * p->ph[i].ua;
    if (offset) offset[i] = off;
    off += r * c;
  }
  if (offset) offset[p->dims.S] = off;
  return GEL_OK;
}

int gel_jac_fd_block_cols(const gel_problem* p, int32_t phase, int32_t* cols) {
  if (!p || !cols || phase < 0 || phase >= p->dims.S) return fail(GEL_ERR_ARG, "bad argument");
  std::vector<int32_t> c;
  phase_columns(p, phase, c);
  std::memcpy(cols, c.data(), sizeof(int32_t) * c.size());
  return GEL_OK;
}

int gel_jac_fd_device(gel_problem* p, int32_t group, const double* d_x, double* d_J, int32_t blocks, void* stream) {
  if (!p || !d_x || !d_J || group < 0 || group >= GEL_NUM_GROUPS) return fail(GEL_ERR_ARG, "bad argument");
  NEED_DEVICE(p);
  HIPCHK(hipSetDevice(p->device));
  int rc = jfd_allocate(p);
  if (rc) return rc;
  hipStream_t s = stream ? (hipStream_t)stream : p->stream;
  p->jfd_last_x.clear();   // the residuals kept for gel_jac_fd's host callers are overwritten
  if ((rc = jfd_evaluate(p, d_x, s))) return rc;
  size_t total = 0;
  return jfd_quotients(p, group, blocks ? 1 : 0, d_J, s, &total);
}

// --------------------------- RHS / point hooks ---------------------------
namespace {
struct DevBuf {
  double* p = nullptr;
  ~DevBuf() { if (p) hipFree(p); }
  int put(const double* h, size_t n) {
    HIPCHK(hipMalloc((void**)&p, std::max<size_t>(1, n) * 8));
    if (h && n) HIPCHK(hipMemcpy(p, h, n * 8, hipMemcpyHostToDevice));
    return GEL_OK;
  }
};
int need_device() {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(GEL_ERR_HIP, "no HIP device: the engine has no CPU fallback");
  return GEL_OK;
}
}  // namespace

int gel_dynamics_velocity(int32_t n, const double* mass_e, const double* pos_e, const double* vel_e, const double* quat,
                          const double* t, const double* param, const double* wind, int32_t Kw, const double* ca,
                          int32_t Kc, const double* units, double barC20, double* out) {
  if (n < 0 || !mass_e || !pos_e || !vel_e || !quat || !t || !param || !wind || !ca || !units || !out)
    return fail(GEL_ERR_ARG, "null argument");
  if (n == 0) return GEL_OK;
  int rc = need_device();
  if (rc) return rc;
  if (const char* why = gel::check_tables(wind, Kw, ca, Kc)) return fail(GEL_ERR_ARG, why);
  const std::vector<double> tables = gel::build_tables(wind, Kw, ca, Kc);
  DevBuf m, r, v, q, tt, tb, o;
  if ((rc = m.put(mass_e, n)) || (rc = r.put(pos_e, 3 * (size_t)n)) || (rc = v.put(vel_e, 3 * (size_t)n)) ||
      (rc = q.put(quat, 4 * (size_t)n)) || (rc = tt.put(t, n)) || (rc = tb.put(tables.data(), tables.size())) ||
      (rc = o.put(nullptr, 3 * (size_t)n)))
    return rc;
  if (barC20 == 0.0) barC20 = -0.484165371736e-3;
  HIPCHK(gel::launch_rhs_vel(true, n, m.p, r.p, v.p, q.p, tt.p, tb.p, Kw, Kc, param[0], param[2], param[4], units[0],
                             units[1], units[2], barC20, o.p, nullptr));
  HIPCHK(hipMemcpy(out, o.p, 3 * (size_t)n * 8, hipMemcpyDeviceToHost));
  return GEL_OK;
}

int gel_dynamics_velocity_NoAir(int32_t n, const double* mass_e, const double* pos_e, const double* quat,
                                const double* param, const double* units, double barC20, double* out) {
  if (n < 0 || !mass_e || !pos_e || !quat || !param || !units || !out) return fail(GEL_ERR_ARG, "null argument");
  if (n == 0) return GEL_OK;
  int rc = need_device();
  if (rc) return rc;
  DevBuf m, r, q, o;
  if ((rc = m.put(mass_e, n)) || (rc = r.put(pos_e, 3 * (size_t)n)) || (rc = q.put(quat, 4 * (size_t)n)) ||
      (rc = o.put(nullptr, 3 * (size_t)n)))
    return rc;
  if (barC20 == 0.0) barC20 = -0.484165371736e-3;
  HIPCHK(gel::launch_rhs_vel(false, n, m.p, r.p, nullptr, q.p, nullptr, nullptr, 0, 0, param[0], 0.0, 0.0, units[0],
                             units[1], units[2], barC20, o.p, nullptr));
  HIPCHK(hipMemcpy(out, o.p, 3 * (size_t)n * 8, hipMemcpyDeviceToHost));
  return GEL_OK;
}

int gel_dynamics_quaternion(int32_t n, const double* quat, const double* u_e, double unit_u, double* out) {
  if (n < 0 || !quat || !u_e || !out) return fail(GEL_ERR_ARG, "null argument");
  if (n == 0) return GEL_OK;
  int rc = need_device();
  if (rc) return rc;
  DevBuf q, u, o;
  if ((rc = q.put(quat, 4 * (size_t)n)) || (rc = u.put(u_e, 2 * (size_t)n)) || (rc = o.put(nullptr, 4 * (size_t)n))) return rc;
  HIPCHK(gel::launch_rhs_quat(n, q.p, u.p, unit_u, o.p, nullptr));
  HIPCHK(hipMemcpy(out, o.p, 4 * (size_t)n * 8, hipMemcpyDeviceToHost));
  return GEL_OK;
}

namespace {
// grows a device or pinned buffer; on failure the buffer is gone and its capacity is 0 (never a stale pointer)
int grow(double** buf, size_t* cap, size_t need, bool pinned) {
  if (*cap >= need) return GEL_OK;
  if (*buf) { if (pinned) hipHostFree(*buf); else hipFree(*buf); }
  *buf = nullptr; *cap = 0;
  HIPCHK(pinned ? hipHostMalloc((void**)buf, need * 8) : hipMalloc((void**)buf, need * 8));
  *cap = need;
  return GEL_OK;
}
}  // namespace

// ------------------ aero path constraints (lib/con_aero.py) ------------------
int gel_aero_configure(gel_problem* p, int32_t kind, int32_t nspec, const int32_t* phase, const int32_t* range_all,
                       const double* limit) {
  if (!p || kind < 0 || kind > 2 || nspec < 0 || (nspec && (!phase || !range_all || !limit)))
    return fail(GEL_ERR_ARG, "bad argument");
  std::vector<gel::AeroRowDev> rows;
  int prev = -1;
  for (int s = 0; s < nspec; s++) {
    // the reference walks range(num_sections - 1) in order (con_aero.py:108): the last phase is never constrained
    if (phase[s] < 0 || phase[s] >= (int)p->ph.size() - 1 || phase[s] <= prev)
      return fail(GEL_ERR_ARG, "aero specs must name increasing phases in [0, num_sections - 1)");
    if (!(limit[s] != 0.0)) return fail(GEL_ERR_ARG, "aero limit must be non-zero");
    prev = phase[s];
    const int nk = range_all[s] ? p->ph[phase[s]].n + 1 : 1;
    const int row0 = (int)rows.size();
    for (int k = 0; k < nk; k++) rows.push_back(gel::AeroRowDev{phase[s], k, nk, row0, limit[s]});
  }
  p->aero_rows[kind] = rows;
  // the union of the constrained state nodes over the three kinds, in (phase, node) order
  std::vector<gel::AeroNodeDev> nodes;
  for (int i = 0; i + 1 < (int)p->ph.size(); i++)
    for (int k = 0; k <= p->ph[i].n; k++) {
      gel::AeroNodeDev nd{i, k, {-1, -1, -1}, {0, 0, 0}, {0, 0, 0}, k, {1.0, 1.0, 1.0}};
      bool any = false;
      for (int kd = 0; kd < 3; kd++) {
        const auto& A = p->aero_rows[kd];
        for (size_t r = 0; r < A.size(); r++)
          if (A[r].phase == i && A[r].k == k) {
            nd.row[kd] = (int32_t)r; nd.nk[kd] = A[r].nk; nd.row0[kd] = A[r].row0; nd.limit[kd] = A[r].limit;
            any = true;
          }
      }
      if (any) nodes.push_back(nd);
    }
  p->aero_nodes = nodes;
  // ---- the per-vector record of gel_eval_batch_aero_device: two parts, each in gel_eval_aero_all's layout for ITS rows.
  //      Part A: nodes 1 .. n of the aerodynamic phases' "all nodes" specs -- what the fused kernel's lanes can write, a spec's row of
  //      a column n doubles long.  Part B: every other row.  Every section starts on a multiple of eight doubles.
  auto fusable = [&](const gel::AeroRowDev& r) { return p->ph[r.phase].air && r.nk == p->ph[r.phase].n + 1 && r.k >= 1; };
  for (int part = 0; part < 2; part++)
    for (int kd = 0; kd < 3; kd++) {
      p->aero_part_rows[part][kd].clear();
      p->aero_part_of[kd].assign(p->aero_rows[kd].size(), 0);
    }
  for (int kd = 0; kd < 3; kd++) {
    const auto& A = p->aero_rows[kd];
    for (size_t r0 = 0; r0 < A.size(); r0 += A[r0].nk) {
      // the spec's rows of part A and of part B, each a spec of its own there
      int rowA0 = (int)p->aero_part_rows[0][kd].size(), rowB0 = (int)p->aero_part_rows[1][kd].size(), nA = 0, nB = 0;
      for (int k = 0; k < A[r0].nk; k++) (fusable(A[r0 + k]) ? nA : nB)++;
      int iA = 0, iB = 0;
      for (int k = 0; k < A[r0].nk; k++) {
        const bool fa = fusable(A[r0 + k]);
        gel::AeroRowDev q = A[r0 + k];
        q.nk = fa ? nA : nB; q.row0 = fa ? rowA0 : rowB0;
        p->aero_part_of[kd][r0 + k] = (fa ? 0 : 1) | ((fa ? rowA0 + iA : rowB0 + iB) << 1);   // part, row inside the part
        p->aero_part_rows[fa ? 0 : 1][kd].push_back(q);
        (fa ? iA : iB)++;
      }
    }
  }
  int64_t off = 0;
  auto pad8 = [](int64_t v) { return (v + 7) / 8 * 8; };
  // part A, spec-major (gel_device.h AeroPhaseDev): one block of 13 n doubles per (kind, phase) spec
  for (int kd = 0; kd < 3; kd++) {
    const auto& A = p->aero_part_rows[0][kd];
    p->aero_partA_base[kd].assign(A.size(), 0);
    for (size_t r0 = 0; r0 < A.size(); r0 += A[r0].nk) {
      for (int k = 0; k < A[r0].nk; k++) p->aero_partA_base[kd][r0 + k] = off;
      off = pad8(off + (int64_t)(p->fd_recompute ? gel::kAeroSpecColsWithT : gel::kAeroSpecCols) * A[r0].nk);
    }
  }
  {   // the dump area: where the lanes' stores of a kind their phase does not have go (AeroPhaseDev::base)
    int nmax = 0;
    for (const auto& h : p->ph) nmax = std::max(nmax, h.n);
    p->aero_dump = off;
    off = pad8(off + (int64_t)gel::kAeroSpecColsWithT * nmax);
  }
  p->aero_partA_len = off;
  // part B: gel_eval_aero_all's layout for its rows
  for (int kd = 0; kd < 3; kd++) { p->aero_off_con[1][kd] = off; off = pad8(off + (int64_t)p->aero_part_rows[1][kd].size()); }
  for (int kd = 0; kd < 3; kd++) { p->aero_off_jac[1][kd] = off; off = pad8(off + (int64_t)p->aero_part_rows[1][kd].size() * ((kd == 1) ? 8 : 12)); }
  for (int kd = 0; kd < 3; kd++) { p->aero_off_con[0][kd] = 0; p->aero_off_jac[0][kd] = 0; }
  p->aero_ld = off;
  // node tables of the two parts.  Part A: row0 = first double of the spec's block in the record, row = the constraint value's place
  // (block + node), ko = the node's place in a column's row; part B: the ordinary tables of its own rows
  for (int part = 0; part < 2; part++) {
    auto& nodes_p = p->aero_part_nodes[part];
    nodes_p.clear();
    for (const auto& nd0 : p->aero_nodes) {
      gel::AeroNodeDev nd = nd0;
      bool any = false;
      for (int kd = 0; kd < 3; kd++) {
        nd.row[kd] = -1;
        if (nd0.row[kd] < 0) continue;
        const int32_t po = p->aero_part_of[kd][nd0.row[kd]];
        if ((po & 1) != part) continue;
        const auto& q = p->aero_part_rows[part][kd][po >> 1];
        nd.ko = (po >> 1) - q.row0;
        nd.nk[kd] = q.nk;
        if (part == 0) {
          nd.row0[kd] = (int32_t)p->aero_partA_base[kd][po >> 1];
          nd.row[kd] = nd.row0[kd] + nd.ko;
        } else {
          nd.row[kd] = po >> 1; nd.row0[kd] = q.row0;
        }
        any = true;
      }
      if (any) nodes_p.push_back(nd);
    }
  }
  // per-phase records of part A for the fused kernel's lanes
  p->aero_ph.assign(p->ph.size(), gel::AeroPhaseDev{});
  for (size_t i = 0; i < p->ph.size(); i++) {
    gel::AeroPhaseDev& a = p->aero_ph[i];
    for (int kd = 0; kd < 3; kd++) { a.il[kd] = 0.0; a.ilx[kd] = 0.0; a.base[kd] = (int32_t)(8 * p->aero_dump); }
    for (int kd = 0; kd < 3; kd++) {
      const auto& A = p->aero_part_rows[0][kd];
      for (size_t r0 = 0; r0 < A.size(); r0 += A[r0].nk) {
        if (A[r0].phase != (int)i) continue;
        a.kinds |= 1 << kd;
        a.base[kd] = (int32_t)(8 * p->aero_partA_base[kd][r0]);
        a.il[kd] = 1.0 / A[r0].limit;            // the kernels' frcp(limit): the correctly rounded quotient
        a.ilx[kd] = a.il[kd] * p->dev.inv_dx;
      }
    }
  }
  if (8 * p->aero_ld >= (int64_t)1 << 31) return fail(GEL_ERR_ARG, "aero record too long for 32-bit byte offsets");
  if (p->device != GEL_DEVICE_NONE) {
    HIPCHK(hipSetDevice(p->device));
    HIPCHK(hipStreamSynchronize(p->stream));
    hipFree(p->d_aero_nodes); hipFree(p->d_aero_ph);
    p->d_aero_nodes = nullptr; p->d_aero_ph = nullptr;
    hipFree(p->d_aero_part_nodes[0]); hipFree(p->d_aero_part_nodes[1]);
    p->d_aero_part_nodes[0] = p->d_aero_part_nodes[1] = nullptr;
    int rc = upload(&p->d_aero_nodes, p->aero_nodes);
    if (rc) return rc;
    if ((rc = upload(&p->d_aero_ph, p->aero_ph))) return rc;
    for (int part = 0; part < 2; part++)
      if (!p->aero_part_nodes[part].empty() && (rc = upload(&p->d_aero_part_nodes[part], p->aero_part_nodes[part]))) return rc;
  }
  return GEL_OK;
}

int gel_aero_record_layout(const gel_problem* p, int64_t* width, int64_t* off_con, int64_t* off_jac) {
  if (!p || !width || !off_con || !off_jac) return fail(GEL_ERR_ARG, "bad argument");
  *width = p->aero_ld;
  for (int part = 0; part < 2; part++)
    for (int k = 0; k < 3; k++) { off_con[3 * part + k] = p->aero_off_con[part][k]; off_jac[3 * part + k] = p->aero_off_jac[part][k]; }
  return GEL_OK;
}

// record index of every entry of gel_eval_aero_all's arrays: var = -1 the constraint vector [nrows], var 0..3 the position /
// velocity / quaternion / t block of the gradient values (in the order of gel_aero_pattern)
int gel_aero_record_map(const gel_problem* p, int32_t kind, int32_t var, int64_t* idx) {
  if (!p || kind < 0 || kind > 2 || var < -1 || var > 3 || !idx) return fail(GEL_ERR_ARG, "bad argument");
  const auto& A = p->aero_rows[kind];
  const int nq = (kind == 1) ? 0 : 4;
  int64_t o = 0;
  if (var == -1) {
    for (size_t r = 0; r < A.size(); r++) {
      const int32_t po = p->aero_part_of[kind][r];
      if (po & 1) idx[o++] = p->aero_off_con[1][kind] + (po >> 1);
      else idx[o++] = p->aero_partA_base[kind][po >> 1] + ((po >> 1) - p->aero_part_rows[0][kind][po >> 1].row0);
    }
    return GEL_OK;
  }
  if (var == 2 && kind == 1) return GEL_OK;   // dynamic pressure has no quaternion block
  const int64_t w = (var == 2) ? 4 : ((var == 3) ? 2 : 3);
  const int64_t bo = (var == 0) ? 0 : ((var == 1) ? 3 : ((var == 2) ? 6 : 6 + nq));
  for (size_t r0 = 0; r0 < A.size(); r0 += A[r0].nk)
    for (int64_t j = 0; j < w; j++)
      for (int k = 0; k < A[r0].nk; k++) {     // the reference's emission order: spec, column, node (gel_aero_pattern)
        const int32_t po = p->aero_part_of[kind][r0 + k];
        const int part = po & 1;
        const auto& q = p->aero_part_rows[part][kind][po >> 1];
        const int64_t R = (int64_t)p->aero_part_rows[part][kind].size();
        const int64_t ko = (po >> 1) - q.row0;
        if (part == 1) idx[o++] = p->aero_off_jac[1][kind] + bo * R + w * q.row0 + j * q.nk + ko;
        else if (var == 3 && !p->fd_recompute) idx[o++] = -1;      // an exact zero, not stored (AeroPhaseDev)
        else idx[o++] = p->aero_partA_base[kind][po >> 1] + (((var == 0) ? 1 : ((var == 1) ? 4 : ((var == 2) ? 7 : 11))) + j) * q.nk + ko;
      }
  return GEL_OK;
}

// Defect groups AND aero path constraints of a resident batch: d_res [B][11N], d_jvar [B][V] as gel_eval_batch_device writes them,
// d_aero [B][width] one record per vector (gel_aero_record_layout / gel_aero_record_map).
int gel_eval_batch_aero_device(gel_problem* p, int32_t B, const double* d_x, double* d_res, double* d_jvar, double* d_aero,
                               void* stream) {
  if (!p || !d_x || B < 1 || !d_res || !d_jvar || !d_aero) return fail(GEL_ERR_ARG, "bad argument");
  NEED_DEVICE(p);
  if (p->aero_nodes.empty()) return fail(GEL_ERR_ARG, "no aero path constraints configured (gel_aero_configure)");
  hipStream_t s = stream ? (hipStream_t)stream : p->stream;
  gel::AeroLaunchOut out[2];
  for (int part = 0; part < 2; part++)
    for (int k = 0; k < 3; k++) {   // part A (spec-major): every pointer is the record itself
      out[part].nrows[k] = (int32_t)p->aero_part_rows[part][k].size();
      out[part].con[k] = out[part].nrows[k] ? d_aero + p->aero_off_con[part][k] : nullptr;
      out[part].jac[k] = out[part].nrows[k] ? d_aero + p->aero_off_jac[part][k] : nullptr;
    }
  // GEL_AERO_FUSED=0: the two kernels one after the other (same record, same bits)
  const char* e = std::getenv("GEL_AERO_FUSED");
  const bool fused = gel::eval_aero_fusable(p->dev, B) && !p->aero_part_nodes[0].empty() && !(e && e[0] == '0');
  if (fused) {
    gel::ProblemDev dv = p->dev;
    dv.aero_ph = p->d_aero_ph; dv.aero_out = d_aero; dv.aero_ld = p->aero_ld;
    HIPCHK(gel::launch_eval_aero(dv, B, d_x, d_res, d_jvar, s));
  } else {
    HIPCHK(gel::launch_eval(p->dev, B, d_x, d_res, d_jvar, s));
    HIPCHK(gel::launch_aero(p->dev, (int)p->aero_part_nodes[0].size(), p->d_aero_part_nodes[0], B, d_x, out[0], s, p->aero_ld, true));
  }
  HIPCHK(gel::launch_aero_wide(p->dev, (int)p->aero_part_nodes[1].size(), p->d_aero_part_nodes[1], B, d_x, out[1], p->aero_ld, s));
  return GEL_OK;
}

int gel_aero_dims(const gel_problem* p, int32_t kind, int32_t* nrows, int64_t* nnz4) {
  if (!p || kind < 0 || kind > 2 || !nrows || !nnz4) return fail(GEL_ERR_ARG, "bad argument");
  const int64_t R = (int64_t)p->aero_rows[kind].size();
  *nrows = (int32_t)R;
  nnz4[0] = 3 * R; nnz4[1] = 3 * R; nnz4[2] = (kind == 1) ? 0 : 4 * R; nnz4[3] = 2 * R;
  return GEL_OK;
}

int gel_aero_pattern(const gel_problem* p, int32_t kind, int32_t var, int32_t* rows, int32_t* cols) {
  // emission order of inequality_jac_max_*: con_aero.py:437-463
  if (!p || kind < 0 || kind > 2 || var < 0 || var > 3 || !rows || !cols) return fail(GEL_ERR_ARG, "bad argument");
  const auto& A = p->aero_rows[kind];
  int64_t o = 0;
  for (size_t r0 = 0; r0 < A.size(); r0 += A[r0].nk) {
    const int nk = A[r0].nk, i = A[r0].phase, xa = p->ph[i].xa, iRow = (int)r0;
    if (var == 3) {
      for (int k = 0; k < nk; k++) { rows[o] = iRow + k; cols[o++] = i; }
      for (int k = 0; k < nk; k++) { rows[o] = iRow + k; cols[o++] = i + 1; }
    } else if (!(var == 2 && kind == 1)) {
      const int w = (var == 2) ? 4 : 3;
      for (int j = 0; j < w; j++)
        for (int k = 0; k < nk; k++) { rows[o] = iRow + k; cols[o++] = (xa + k) * w + j; }
    }
  }
  return GEL_OK;
}

namespace {
size_t aero_jac_len(const gel_problem* p, int kind) { return p->aero_rows[kind].size() * ((kind == 1) ? 8 : 12); }
}  // namespace

int gel_eval_aero_all_device(gel_problem* p, int32_t B, const double* d_x, double* const* d_con, double* const* d_jac,
                             void* stream) {
  if (!p || B < 1 || !d_x || !d_con) return fail(GEL_ERR_ARG, "bad argument");
  NEED_DEVICE(p);
  gel::AeroLaunchOut out;
  for (int k = 0; k < 3; k++) {
    out.nrows[k] = (int32_t)p->aero_rows[k].size();
    out.con[k] = out.nrows[k] ? d_con[k] : nullptr;
    out.jac[k] = (out.con[k] && d_jac) ? d_jac[k] : nullptr;
  }
  HIPCHK(gel::launch_aero(p->dev, (int)p->aero_nodes.size(), p->d_aero_nodes, B, d_x, out, stream ? (hipStream_t)stream : p->stream));
  return GEL_OK;
}

int gel_eval_aero_all(gel_problem* p, int32_t B, const double* x, double* const* con, double* const* jac) {
  if (!p || B < 1 || !x || !con) return fail(GEL_ERR_ARG, "bad argument");
  NEED_DEVICE(p);
  if (p->aero_nodes.empty()) return GEL_OK;
  HIPCHK(hipSetDevice(p->device));
  // one contiguous output area: con[0] | jac[0] | con[1] | jac[1] | con[2] | jac[2] (only what was asked for)
  size_t off_c[3], off_j[3], total = 0;
  for (int k = 0; k < 3; k++) {
    const size_t R = p->aero_rows[k].size();
    off_c[k] = total; total += (con[k] && R) ? (size_t)B * R : 0;
    off_j[k] = total; total += (con[k] && R && jac && jac[k]) ? (size_t)B * aero_jac_len(p, k) : 0;
  }
  if (total == 0) return GEL_OK;
  const size_t nx = (size_t)B * p->dims.num_vars;
  const bool zero_copy = (nx + total) * 8 <= kZeroCopyBytes;
  int rc;
  double* base;
  if (zero_copy) {
    // the optimiser's callback: x and every output in pinned host memory, one launch + one synchronise
    if ((rc = ensure_capacity(p, B)) || (rc = grow(&p->h_aero, &p->h_aero_cap, total, true))) return rc;
    std::memcpy(p->h_x, x, nx * 8);
    base = p->h_aero;
  } else {
    if ((rc = grow(&p->d_aero_x, &p->d_aero_x_cap, nx, false)) || (rc = grow(&p->d_aero_out, &p->d_aero_out_cap, total, false))) return rc;
    HIPCHK(hipMemcpyAsync(p->d_aero_x, x, nx * 8, hipMemcpyHostToDevice, p->stream));
    base = p->d_aero_out;
  }
  gel::AeroLaunchOut out;
  for (int k = 0; k < 3; k++) {
    const size_t R = p->aero_rows[k].size();
    out.nrows[k] = (int32_t)R;
    out.con[k] = (con[k] && R) ? base + off_c[k] : nullptr;
    out.jac[k] = (out.con[k] && jac && jac[k]) ? base + off_j[k] : nullptr;
  }
  gel::ProblemDev dv = p->dev;
  if (zero_copy) dv.flag = p->h_flag;
  HIPCHK(gel::launch_aero(dv, (int)p->aero_nodes.size(), p->d_aero_nodes, B, zero_copy ? p->h_x : p->d_aero_x, out, p->stream));
  if (!zero_copy) {
    for (int k = 0; k < 3; k++) {
      if (out.con[k]) HIPCHK(hipMemcpyAsync(con[k], out.con[k], (size_t)B * p->aero_rows[k].size() * 8, hipMemcpyDeviceToHost, p->stream));
      if (out.jac[k]) HIPCHK(hipMemcpyAsync(jac[k], out.jac[k], (size_t)B * aero_jac_len(p, k) * 8, hipMemcpyDeviceToHost, p->stream));
    }
    HIPCHK(hipMemcpyAsync(p->h_flag, p->d_flag, 4, hipMemcpyDeviceToHost, p->stream));
  }
  HIPCHK(hipStreamSynchronize(p->stream));
  if (zero_copy)
    for (int k = 0; k < 3; k++) {
      if (out.con[k]) std::memcpy(con[k], out.con[k], (size_t)B * p->aero_rows[k].size() * 8);
      if (out.jac[k]) std::memcpy(jac[k], out.jac[k], (size_t)B * aero_jac_len(p, k) * 8);
    }
  if (*p->h_flag) {
    *p->h_flag = 0;
    if (!zero_copy) HIPCHK(hipMemsetAsync(p->d_flag, 0, 4, p->stream));
    return GEL_NONFINITE;
  }
  return GEL_OK;
}

int gel_eval_aero(gel_problem* p, int32_t kind, int32_t B, const double* x, double* con, double* jac_vals) {
  if (!p || kind < 0 || kind > 2 || B < 1 || !x || !con) return fail(GEL_ERR_ARG, "bad argument");
  double* c[3] = {nullptr, nullptr, nullptr};
  double* j[3] = {nullptr, nullptr, nullptr};
  c[kind] = con; j[kind] = jac_vals;
  return gel_eval_aero_all(p, B, x, c, j);
}

// ------------- knot / terminal / user rows (lib/con_init_terminal_knot.py, example/user_constraints.py) -------------
int gel_rows_configure(gel_problem* p, int32_t nlin, const gel_linear_row* lin, int32_t nfn, const gel_nodefn_row* fn) {
  if (!p || nlin < 0 || nfn < 0 || (nlin && !lin) || (nfn && !fn)) return fail(GEL_ERR_ARG, "bad argument");
  for (int i = 0; i < nlin; i++)
    if (lin[i].idx0 < 0 || lin[i].idx0 >= p->dims.num_vars || lin[i].idx1 >= p->dims.num_vars)
      return fail(GEL_ERR_ARG, "linear row: variable index out of range");
  for (int i = 0; i < nfn; i++)
    if (fn[i].fn < 0 || fn[i].fn > 15 || fn[i].node < 0 || fn[i].node >= p->dims.M || !(fn[i].p[0] != 0.0) ||
        fn[i].tcol < -1 || fn[i].tcol > p->dims.S || (fn[i].fn >= 9 && fn[i].tcol < 0) || (fn[i].mode & ~15) || (fn[i].mode & 3) > 1)
      return fail(GEL_ERR_ARG, "node-function row: unknown function or mode, node / time column out of range, or zero scale");
  p->lin_rows.resize(nlin);
  p->fn_rows.resize(nfn);
  for (int i = 0; i < nlin; i++) p->lin_rows[i] = gel::LinRowDev{lin[i].idx0, lin[i].idx1 < 0 ? -1 : lin[i].idx1, lin[i].coef0, lin[i].coef1, lin[i].c0};
  for (int i = 0; i < nfn; i++) {
    gel::FnRowDev& d = p->fn_rows[i];
    d.fn = fn[i].fn; d.node = fn[i].node; d.tcol = fn[i].tcol; d.mode = fn[i].mode;
    for (int k = 0; k < 8; k++) d.p[k] = fn[i].p[k];
  }
  if (p->device == GEL_DEVICE_NONE) return GEL_OK;
  HIPCHK(hipSetDevice(p->device));
  HIPCHK(hipStreamSynchronize(p->stream));
  hipFree(p->d_lin_rows); hipFree(p->d_fn_rows);
  p->d_lin_rows = nullptr; p->d_fn_rows = nullptr;
  int rc = GEL_OK;
  if ((rc = upload(&p->d_lin_rows, p->lin_rows)) || (rc = upload(&p->d_fn_rows, p->fn_rows))) return rc;
  return GEL_OK;
}

int gel_rows_dims(const gel_problem* p, int32_t* nlin, int32_t* nfn) {
  if (!p || !nlin || !nfn) return fail(GEL_ERR_ARG, "null argument");
  *nlin = (int32_t)p->lin_rows.size();
  *nfn = (int32_t)p->fn_rows.size();
  return GEL_OK;
}

int gel_rows_eval_device(gel_problem* p, int32_t B, const double* d_x, double* d_con, double* d_jfn, void* stream) {
  if (!p || B < 1 || !d_x || !d_con) return fail(GEL_ERR_ARG, "bad argument");
  NEED_DEVICE(p);
  HIPCHK(gel::launch_rows(p->dev, (int)p->lin_rows.size(), p->d_lin_rows, (int)p->fn_rows.size(), p->d_fn_rows, B, d_x,
                          d_con, d_jfn, stream ? (hipStream_t)stream : p->stream));
  return GEL_OK;
}


int gel_rows_eval(gel_problem* p, int32_t B, const double* x, double* con, double* jfn) {
  if (!p || B < 1 || !x || !con) return fail(GEL_ERR_ARG, "bad argument");
  NEED_DEVICE(p);
  const size_t R = p->lin_rows.size() + p->fn_rows.size(), nf = p->fn_rows.size();
  if (R == 0) return GEL_OK;
  HIPCHK(hipSetDevice(p->device));
  const size_t nx = (size_t)B * p->dims.num_vars, nc = (size_t)B * R, nj = jfn ? (size_t)B * nf * 7 : 0;
  int rc;
  if ((nx + nc + nj) * 8 <= kZeroCopyBytes) {
    // the optimiser's callback: the kernel reads x from and writes to pinned host memory, one launch + one synchronise
    if ((rc = ensure_capacity(p, B)) || (rc = grow(&p->h_rows, &p->h_rows_cap, nc + nj + 1, true))) return rc;
    std::memcpy(p->h_x, x, nx * 8);
    gel::ProblemDev dv = p->dev;
    dv.flag = p->h_flag;
    HIPCHK(gel::launch_rows(dv, (int)p->lin_rows.size(), p->d_lin_rows, (int)nf, p->d_fn_rows, B, p->h_x, p->h_rows,
                            jfn ? p->h_rows + nc : nullptr, p->stream));
    HIPCHK(hipStreamSynchronize(p->stream));
    std::memcpy(con, p->h_rows, nc * 8);
    if (jfn) std::memcpy(jfn, p->h_rows + nc, nj * 8);
    if (*p->h_flag) { *p->h_flag = 0; return GEL_NONFINITE; }
    return GEL_OK;
  }
  if ((rc = grow(&p->d_rows_x, &p->d_rows_x_cap, nx, false)) || (rc = grow(&p->d_rows_out, &p->d_rows_out_cap, nc + nj + 1, false))) return rc;
  HIPCHK(hipMemcpyAsync(p->d_rows_x, x, nx * 8, hipMemcpyHostToDevice, p->stream));
  HIPCHK(gel::launch_rows(p->dev, (int)p->lin_rows.size(), p->d_lin_rows, (int)nf, p->d_fn_rows, B, p->d_rows_x,
                          p->d_rows_out, jfn ? p->d_rows_out + nc : nullptr, p->stream));
  HIPCHK(hipMemcpyAsync(con, p->d_rows_out, nc * 8, hipMemcpyDeviceToHost, p->stream));
  if (jfn) HIPCHK(hipMemcpyAsync(jfn, p->d_rows_out + nc, nj * 8, hipMemcpyDeviceToHost, p->stream));
  HIPCHK(hipMemcpyAsync(p->h_flag, p->d_flag, 4, hipMemcpyDeviceToHost, p->stream));
  HIPCHK(hipStreamSynchronize(p->stream));
  if (*p->h_flag) { *p->h_flag = 0; HIPCHK(hipMemsetAsync(p->d_flag, 0, 4, p->stream)); return GEL_NONFINITE; }
  return GEL_OK;
}

// ------------- post-processing table (output_result.py:37-263) -------------
int gel_output_table(gel_problem* p, const double* x, const double* tx_res, double launch_lat_deg, double launch_lon_deg,
                     double* out) {
  if (!p || !x || !tx_res || !out) return fail(GEL_ERR_ARG, "null argument");
  NEED_DEVICE(p);
  HIPCHK(hipSetDevice(p->device));
  const int M = p->dims.M;
  // the section of every state node (output_result.py:121-143: section s owns its n + 1 state nodes)
  std::vector<int32_t> sec((size_t)M);
  for (size_t i = 0; i < p->ph.size(); i++)
    for (int k = 0; k <= p->ph[i].n; k++) sec[(size_t)p->ph[i].xa + k] = (int32_t)i;
  const size_t nx = (size_t)p->dims.num_vars, no = (size_t)M * gel::kOutputColumns;
  int rc;
  // one scratch buffer: x | tx | out | node sections (as doubles' worth of bytes)
  const size_t words = nx + (size_t)M + no + ((size_t)M + 1) / 2;
  if ((rc = grow(&p->d_rows_x, &p->d_rows_x_cap, words, false))) return rc;
  double* d_x = p->d_rows_x;
  double* d_tx = d_x + nx;
  double* d_out = d_tx + M;
  int32_t* d_sec = reinterpret_cast<int32_t*>(d_out + no);
  HIPCHK(hipMemcpyAsync(d_x, x, nx * 8, hipMemcpyHostToDevice, p->stream));
  HIPCHK(hipMemcpyAsync(d_tx, tx_res, (size_t)M * 8, hipMemcpyHostToDevice, p->stream));
  HIPCHK(hipMemcpyAsync(d_sec, sec.data(), (size_t)M * 4, hipMemcpyHostToDevice, p->stream));
  HIPCHK(gel::launch_output(p->dev, M, d_x, d_tx, d_sec, launch_lat_deg, launch_lon_deg, d_out, p->stream));
  HIPCHK(hipMemcpyAsync(out, d_out, no * 8, hipMemcpyDeviceToHost, p->stream));
  HIPCHK(hipStreamSynchronize(p->stream));
  return GEL_OK;
}

// ------------- from-file initial guess on the host (initialize.py:322-409) -------------
int gel_initial_guess(const gel_problem* p, int32_t nref, const double* t_ref, const double* table,
                      const double* knot_times, double* x) {
  if (!p || !t_ref || !table || !knot_times || !x || nref < 2) return fail(GEL_ERR_ARG, "bad argument (>= 2 reference rows)");
  // repeated times are legal (the example's table repeats every knot: end of one section = start of the next); a
  // bracket [lo, hi] found by lower_bound never has zero width unless the table STARTS with a repeat
  for (int k = 1; k < nref; k++)
    if (!(t_ref[k] >= t_ref[k - 1])) return fail(GEL_ERR_ARG, "reference times must not decrease");
  const int S = p->dims.S, M = p->dims.M, N = p->dims.N;
  // scipy interp1d(kind="linear", fill_value="extrapolate"): hi = first knot >= t clipped to [1, nref - 1], lo = hi - 1,
  // y = slope * (t - t_lo) + y_lo with slope = (y_hi - y_lo) / (t_hi - t_lo): the end intervals extend beyond the table
  auto interp = [&](double t, int col0, int ncol, double unit, double* out) {
    int hi = (int)(std::lower_bound(t_ref, t_ref + nref, t) - t_ref);
    hi = std::min(std::max(hi, 1), nref - 1);
    const int lo = hi - 1;
    for (int c = 0; c < ncol; c++) {
      const double ylo = table[(size_t)lo * 13 + col0 + c], yhi = table[(size_t)hi * 13 + col0 + c];
      const double slope = (yhi - ylo) / (t_ref[hi] - t_ref[lo]);
      const double rise = slope * (t - t_ref[lo]);  // its own statement: no fused multiply-add, the bits scipy produces
      out[c] = (rise + ylo) / unit;
    }
  };
  double* xm = x; double* xr = x + M; double* xv = x + 4 * (size_t)M; double* xq = x + 7 * (size_t)M;
  double* xu = x + 11 * (size_t)M; double* xt = x + 11 * (size_t)M + 2 * (size_t)N;
  for (int i = 0; i < S; i++) {
    const HostPhase& h = p->ph[i];
    const double to = knot_times[i], tf = knot_times[i + 1];
    for (int k = 0; k <= h.n; k++) {  // state nodes: tau_x = [-1, tau]
      const double tau = (k == 0) ? -1.0 : h.tau[k - 1];
      const double t = tau * (tf - to) / 2.0 + (tf + to) / 2.0;
      const int node = h.xa + k;
      interp(t, 0, 1, p->um, xm + node);
      interp(t, 1, 3, p->up, xr + 3 * (size_t)node);
      interp(t, 4, 3, p->uv, xv + 3 * (size_t)node);
      interp(t, 7, 4, 1.0, xq + 4 * (size_t)node);
    }
    for (int k = 0; k < h.n; k++) {   // control nodes: tau
      const double t = h.tau[k] * (tf - to) / 2.0 + (tf + to) / 2.0;
      interp(t, 11, 2, p->uu, xu + 2 * (size_t)(h.ua + k));
    }
  }
  for (int i = 0; i <= S; i++) xt[i] = knot_times[i] / p->ut;
  // a node time beyond a reference table that ends (or starts) with a repeated time extrapolates over a zero-width interval
  // (scipy does the same and returns NaN / Inf silently): report it
  for (int64_t i = 0; i < p->dims.num_vars; i++)
    if (!std::isfinite(x[i])) return GEL_NONFINITE;
  return GEL_OK;
}

// ------------- one callback = one device round trip -------------
int gel_eval_callback(gel_problem* p, const double* x, const gel_callback_io* io) {
  if (!p || !x || !io) return fail(GEL_ERR_ARG, "null argument");
  NEED_DEVICE(p);
  HIPCHK(hipSetDevice(p->device));
  int rc = ensure_capacity(p, 1);
  if (rc) return rc;
  const size_t nlin = p->lin_rows.size(), nfn = p->fn_rows.size(), R = nlin + nfn;
  const bool rows = io->rows_con && R;
  size_t off_c[3], off_j[3], atotal = 0;
  bool aero = false;
  for (int k = 0; k < 3; k++) {
    const size_t n = p->aero_rows[k].size();
    off_c[k] = atotal; atotal += (io->aero_con[k] && n) ? n : 0;
    off_j[k] = atotal; atotal += (io->aero_con[k] && n && io->aero_jac[k]) ? aero_jac_len(p, k) : 0;
    aero = aero || (io->aero_con[k] && n);
  }
  if (rows && (rc = grow(&p->h_rows, &p->h_rows_cap, R + 7 * nfn + 1, true))) return rc;
  if (aero && (rc = grow(&p->h_aero, &p->h_aero_cap, atotal, true))) return rc;
  const bool x_pinned = p->cb_x[0] && (x == p->cb_x[0] || x == p->cb_x[1]);
  if (!x_pinned) std::memcpy(p->h_x, x, (size_t)p->dims.num_vars * 8);
  const double* const xin = x_pinned ? x : p->h_x;
  // everything reads x from and writes to pinned host memory: no copy commands
  gel::ProblemDev dv = p->dev;
  dv.flag = p->h_flag;
  const bool want_jac = io->vals_full != nullptr;
  const bool fused = io->res || want_jac;
  // GEL_CB_MODE=3 (measurement switch): the three launches of rounds 1-2, back to back on the handle's stream (compact path)
  static const int cb_mode = [] { const char* e = getenv("GEL_CB_MODE"); return e ? atoi(e) : 0; }();
  const bool coo = want_jac && cb_mode != 3 && coo_direct(p);
  if (coo) {
    if ((rc = ensure_full(p))) return rc;
    dv.coo_full = p->h_full; dv.coo = p->d_coo;
  }
  dv.split_vel = 1;   // a whole evaluation: every part of every work item is in this launch
  const bool own_res = io->res && p->cb_res && io->res == p->cb_res;
  double* const res_to = own_res ? p->cb_res : p->h_res;
  gel::AeroLaunchOut out;
  if (aero)
    for (int k = 0; k < 3; k++) {
      const size_t n = p->aero_rows[k].size();
      out.nrows[k] = (int32_t)n;
      out.con[k] = (io->aero_con[k] && n) ? p->h_aero + off_c[k] : nullptr;
      out.jac[k] = (out.con[k] && io->aero_jac[k]) ? p->h_aero + off_j[k] : nullptr;
    }
  if (cb_mode == 3 || !fused) {
    if (fused) HIPCHK(gel::launch_eval(dv, 1, xin, res_to, want_jac ? p->h_jv : nullptr, p->stream));
    if (rows) HIPCHK(gel::launch_rows(dv, (int)nlin, p->d_lin_rows, (int)nfn, p->d_fn_rows, 1, xin, p->h_rows,
                                      io->rows_jfn ? p->h_rows + R : nullptr, p->stream));
    if (aero) HIPCHK(gel::launch_aero(dv, (int)p->aero_nodes.size(), p->d_aero_nodes, 1, xin, out, p->stream));
  } else {
    // ONE launch: defect groups, aero kinds and row table as workgroup ranges of one grid (gel_kernels.hip callback_kernel)
    arm_done(p, dv);
    HIPCHK(gel::launch_callback(dv, want_jac, xin, res_to, want_jac ? p->h_jv : nullptr,
                                aero ? (int)p->aero_nodes.size() : 0, p->d_aero_nodes, aero ? &out : nullptr,
                                (int)nlin, p->d_lin_rows, (int)nfn, p->d_fn_rows, rows ? p->h_rows : nullptr,
                                (rows && io->rows_jfn) ? p->h_rows + R : nullptr, p->stream));
  }
  HIPCHK(wait_done(p, dv));   // the ONE wait of the callback: the kernel's own word where it signals, else the runtime's synchronise
  if (io->res && !own_res) std::memcpy(io->res, p->h_res, (size_t)11 * p->dims.N * 8);
  if (want_jac) {
    if (coo) finish_full(p, io->vals_full, io->fill_constants);
    else scatter_full(p, p->h_jv, io->vals_full, io->fill_constants);
  }
  if (rows) {
    std::memcpy(io->rows_con, p->h_rows, R * 8);
    if (io->rows_jfn) std::memcpy(io->rows_jfn, p->h_rows + R, 7 * nfn * 8);
  }
  if (aero)
    for (int k = 0; k < 3; k++) {
      if (out.con[k]) std::memcpy(io->aero_con[k], out.con[k], p->aero_rows[k].size() * 8);
      if (out.jac[k]) std::memcpy(io->aero_jac[k], out.jac[k], aero_jac_len(p, k) * 8);
    }
  if (*p->h_flag) { *p->h_flag = 0; return GEL_NONFINITE; }
  return GEL_OK;
}

int gel_point_eval(int32_t kind, int32_t n, const double* in, const double* aux, int32_t aux_rows, double* out) {
  static const int nin[15] = {1, 3, 3, 6, 7, 1, 1, 2, 2, 8, 8, 8, 7, 4, 1}, nout[15] = {5, 3, 3, 3, 3, 3, 1, 4, 6, 8, 16, 4, 3, 7, 2};
  if (kind < 0 || kind > 14 || n < 0 || !in || !out) return fail(GEL_ERR_ARG, "bad argument");
  if (n == 0) return GEL_OK;
  int rc = need_device();
  if (rc) return rc;
  double atm[gel::kAtmTableDoubles];
  size_t naux = 0;
  const double* hax = aux;
  std::vector<double> ext;  // kinds 5 / 6: the table rows followed by their per-interval slopes
  if (kind == 0) { gel::fill_atmosphere_table(atm); hax = atm; naux = gel::kAtmTableDoubles; aux_rows = 0; }
  else if (kind == 2) { if (!aux) return fail(GEL_ERR_ARG, "kind 2 needs aux[0] = barC20"); naux = 1; aux_rows = 0; }
  else if (kind == 5 || kind == 6 || kind == 9 || kind == 10) {
    const int w = (kind == 5 || kind == 10) ? 3 : 2;
    if (!aux || aux_rows < 2) return fail(GEL_ERR_ARG, "kinds 5 and 6 need a table of at least two rows");
    for (int k = 1; k < aux_rows; k++)
      if (!(aux[(size_t)w * k] > aux[(size_t)w * (k - 1)])) return fail(GEL_ERR_ARG, "table abscissae must increase strictly");
    std::vector<double> slopes;
    gel::append_rows_and_slopes(ext, slopes, aux, aux_rows, w);
    ext.insert(ext.end(), slopes.begin(), slopes.end());
    hax = ext.data(); naux = ext.size();
  }
  else { aux_rows = 0; }
  DevBuf i, a, o;
  if ((rc = i.put(in, (size_t)n * nin[kind])) || (rc = a.put(hax, naux)) || (rc = o.put(nullptr, (size_t)n * nout[kind]))) return rc;
  HIPCHK(gel::launch_point(kind, n, i.p, a.p, aux_rows, o.p, nullptr));
  HIPCHK(hipMemcpy(out, o.p, (size_t)n * nout[kind] * 8, hipMemcpyDeviceToHost));
  return GEL_OK;
}

}  // extern "C"
