// gel_device.h -- device-resident problem description shared by host and kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gel {

// One LGR phase (section).  Index arithmetic follows PSparams.get_index
// (lib/SectionParameters.py:83-103): ua = sum n_{<i}, xa = ua + i.
struct PhaseDev {
  int32_t n, ua, xa;
  int32_t air;        // reference_area != 0  -> dynamics_velocity, else _NoAir (lib/con_dynamics.py:257,346)
  int32_t air_fd;     // reference_area  > 0  -> velocity / t0 / tf sweeps are finite differences (:403,454)
  int32_t t_fd;       // air_fd and GEL_FLAG_FD_RECOMPUTE: the t0 / tf columns by two more sweeps (else in closed form, tf = -t0)
  int32_t q_fd;       // free attitude and GEL_FLAG_FD_RECOMPUTE: quaternion-kinematics entries by finite differences (else closed form)
  int32_t engine_on;  // lib/con_dynamics.py:53,80
  int32_t hold;       // attitude in ("hold","vertical")  (lib/con_dynamics.py:521,559)
  int32_t K;          // compact Jacobian slots per node of this phase
  int32_t s_vv, s_vq, s_vt, s_qq;  // slot bases (s_pos = 0, s_vm = 9, s_vp = 12)
  int32_t doff;       // offset of this phase's transposed D in Dt (doubles)
  int32_t toff;       // offset of tau
  int64_t voff;       // offset of this phase in the compact Jacobian vector
  double thrust, massflow, area, nozzle;
  double mf_um;       // -massflow / unit_mass, divided on the host (the same IEEE division the kernel used to repeat in every lane)
};

#ifdef __HIPCC__
// A phase record fetched through the constant address space: the loads are scalar (s_load) and -- unlike
// loads through a plain global pointer -- known not to alias the kernel's own stores.  Without this the
// compiler re-loaded single fields with VECTOR loads in the middle of the sweeps, and the s_waitcnt
// vmcnt(0) behind each of them also waited for every Jacobian store in flight (vmcnt counts in order).
template <class T>
__device__ __forceinline__ T load_const(const T* p) {
  return *(const T __attribute__((address_space(4)))*)p;
}
__device__ __forceinline__ int4 load_const_int4(const int4* p) {   // one s_load_dwordx4
  const int32_t* ip = (const int32_t*)p;
  int4 v;
  v.x = load_const(ip); v.y = load_const(ip + 1); v.z = load_const(ip + 2); v.w = load_const(ip + 3);
  return v;
}
__device__ __forceinline__ PhaseDev load_phase(const PhaseDev* g) {
  PhaseDev q;
  q.n = load_const(&g->n); q.ua = load_const(&g->ua); q.xa = load_const(&g->xa);
  q.air = load_const(&g->air); q.air_fd = load_const(&g->air_fd); q.t_fd = load_const(&g->t_fd); q.q_fd = load_const(&g->q_fd); q.engine_on = load_const(&g->engine_on);
  q.hold = load_const(&g->hold); q.K = load_const(&g->K);
  q.s_vv = load_const(&g->s_vv); q.s_vq = load_const(&g->s_vq); q.s_vt = load_const(&g->s_vt); q.s_qq = load_const(&g->s_qq);
  q.doff = load_const(&g->doff); q.toff = load_const(&g->toff); q.voff = load_const(&g->voff);
  q.thrust = load_const(&g->thrust); q.massflow = load_const(&g->massflow); q.area = load_const(&g->area);
  q.nozzle = load_const(&g->nozzle); q.mf_um = load_const(&g->mf_um);
  return q;
}
#endif

// one aero path-constraint row = one state node of a constrained phase (lib/con_aero.py)
struct AeroRowDev {
  int32_t phase, k, nk, row0;  // phase, node 0..nk-1 inside it, rows of its spec, first row of its spec
  double limit;                // units[3] of con_aero.py
};
// one constrained state node, shared by the kinds that constrain it (kind 0 alpha, 1 q, 2 q-alpha): row = its row in
// that kind's constraint vector (-1: not constrained by that kind), nk / row0 / limit as in AeroRowDev
struct AeroNodeDev {
  int32_t phase, k;
  int32_t row[3], nk[3], row0[3];
  int32_t ko;        // the node's place inside its spec's rows in the OUTPUT (k, or k - 1 in the lanes' part of a record: AeroPhaseDev)
  double limit[3];
};

// Aero rows written by the FUSED kernel (gel_eval_kernel.h, AERO instantiation; gel_eval_batch_aero_device): one record per phase.
// The lanes of an aerodynamic phase's wavefront are its state nodes 1 .. n; where the phase has an "all nodes" spec of a kind they
// also write that kind's constraint value and gradient entries of their node, from the centre evaluation and the position /
// velocity / quaternion sweeps they run anyway.  Outputs: ONE record of aero_ld doubles per decision vector in two parts.
// Part A = the rows the lanes write, SPEC-MAJOR: every (kind, phase) spec is one block of 11 n doubles,
//   [con n | position 3 n | velocity 3 n | quaternion 4 n]      (columns [column][node]; the dynamic pressure leaves its
// quaternion columns unwritten; the t columns are exact zeros -- the air-relative velocity does not depend on the Earth angle --
// and are not stored at all: gel_aero_record_map names them -1, like the constants of gel_full_source; a problem created with
// GEL_FLAG_FD_RECOMPUTE runs the t sweeps and has them as columns 11, 12 of blocks of 13 n),
// so that a lane's store address is base[kind] + ((first column of the block + column) n + node) 8
// -- ONE scalar per kind and phase instead of a table of block offsets -- and a column's row is the phase's n nodes: whole 64-byte
// lines at n = 64.  Part B = every other row (state node 0 of a phase, phases without aerodynamics, "initial" specs) in
// gel_eval_aero_all's per-kind block layout, written by aero_wide_kernel.  gel_aero_record_map gives the record index of every entry
// of the reference's layout.
struct AeroPhaseDev {
  int32_t kinds;       // bit k: nodes 1 .. n of this phase have a row of kind k AND the phase runs the aerodynamic chain
  int32_t base[3];     // kind k: BYTE offset of the spec's block inside the vector's record; a kind the phase does not have: the record's
                       // dump area (13 n doubles nobody reads), with il = ilx = 0 -- the lanes store unconditionally
  double il[3];        // 1 / limit (limit = units[3] of con_aero.py), divided on the host (the same IEEE quotient the kernels' frcp forms)
  double ilx[3];       // (1 / limit) * (1 / dx): what every gradient entry of the kind is scaled by
};
// first column (in units of n doubles) of a block inside a spec-major block: con 0, position 1, velocity 4, quaternion 7, (t 11)
constexpr int kAeroSpecCols = 11, kAeroSpecColsWithT = 13;

// knot / terminal / user rows (lib/con_init_terminal_knot.py, example/user_constraints.py): see gel_kernels.hip rows_kernel
struct LinRowDev { int32_t idx0, idx1; double coef0, coef1, c0; };  // (coef0 x[idx0] + coef1 x[idx1]) + c0; idx1 < 0: one term
// f(position, velocity of state node `node`[, knot time x_t[tcol]]) mapped by `mode` (gel_kernels.hip rows_kernel):
// value: mode & 3 == 0: f / p[0] - p[1];  == 1: (f - p[1]) / p[0];  bit 3: negated
// difference: bit 2 clear: (value(x + dx e_c) - value(x)) / dx;  set: ((f(x + dx e_c) - f(x)) / dx) / p[0], negated with bit 3
struct FnRowDev { int32_t fn, node, tcol, mode; double p[8]; };

#ifndef GEL_LONG_PHASE_FROM
#define GEL_LONG_PHASE_FROM 68
#endif
constexpr int kLongPhaseFrom = GEL_LONG_PHASE_FROM;   // phases of this many nodes and more take the slab loop of the cooperative form (gel_eval_kernel.h)

struct ProblemDev {
  int32_t S, N, M, nvars;
  int32_t Kw, Kc;
  int32_t nchunks;           // sum over phases of ceil(n/64): wavefront work items per eval
  int32_t use_mfma;          // D.X on v_mfma_f64_16x16x4_f64 instead of VALU FMAs
  int32_t pack;              // every phase has at most 32 nodes: the cooperative form carries two decision vectors per wavefront
  int32_t longp;             // some phase has kXldsPipeFrom (68) nodes or more: the cooperative form with the slab loop
  int32_t cached_out;        // this launch's Jacobian values are read again at once (gel_eval_full_device): ordinary stores if they fit the Infinity Cache
  // one-vector launches that tell the host themselves when their results have arrived (pinned host memory): every workgroup counts
  // itself in done_ctr (device memory) after a system-scope release of its stores, the last one stores done_seq to done_flag
  // (pinned host word the host spins on) -- the host does not wait for the end-of-kernel signal (4 us later at 6 x 64).  null: off
  int32_t* done_ctr;
  int32_t* done_flag;
  int32_t done_total, done_seq;
  int32_t chunk0;            // first work item of this launch (phase-sharded launches), else 0
  int32_t unit0, nunits;     // split form only: first unit and number of units (unit = 4 * work item + part)
  int32_t park_off;          // first double of the per-lane LDS park (after the staged tables)
  int32_t vmajor;            // cooperative launches in vector-group major, XCD-aware order (every phase one chunk)
  int32_t fd_recompute;      // GEL_FLAG_FD_RECOMPUTE: every finite-difference sweep re-runs the reference's chain
  int64_t shard_width;       // split form, packed unit-shard output (gel_eval_shard_packed_device): doubles per vector in a rank's
                             // slice of the exchange buffer; 0: the ordinary res / jvar layouts
  const int64_t* unit_base;  // packed output: first double of every unit's block inside its rank's per-vector block [4 * work items]
  const int4* chunks;        // [nchunks] {phase, first node of the chunk, offset of its MFMA-ordered D in Dsw / 4,
                             //  (position in the run of this phase's chunks) << 16 | chunks of the phase in the list}:
                             // dearest phase type first for a whole launch, the natural (phase) order for a
                             // phase-sharded one
  const double* Dsw;         // D in matrix-pipe feed order: [chunk][k-step][lane][row tile 0..3] (see gel_host.hip):
                             // one wavefront feeds all four row tiles (split latency form)
  const double* Dst;         // the same values as [chunk][k-step][row tile 0..3][lane]: wavefront w of a workgroup
                             // feeds row tile w only (cooperative throughput form), 512 contiguous bytes per k-step
  int64_t V;                 // compact entries per eval
  const PhaseDev* phases;    // [S]
  const int32_t* node_phase; // [N]
  const double* Dt;          // per phase, transposed: Dt[doff + i*n + j] = D[j][i]
  const double* tau;         // per phase
  const double* tables;      // atm[77] (gel_physics.h kAtmDoubles) | wind[Kw*3] | ca[Kc*2]
  int32_t* flag;             // non-finite flag
  double um, up, uv, uu, ut, dx, barC20;
  // wave-uniform quotients formed once on the host instead of by a division sequence (12 instructions) per lane and use:
  // 1 / unit_v and 1 / dx (bit-identical to the device division), and unit_v unit_t / 2 / unit_p -- the factor of the position
  // defect's right-hand side and of its t columns (lib/con_dynamics.py:146-152,196-210), which the reference multiplies out per
  // element: v kpt differs from ((v unit_v) unit_t / 2) / unit_p by <= 2 ulp
  double inv_uv, inv_dx, kpt;
  double hT;   // unit_t / 2 (exact)
  // COO-direct output of the latency form (gel_eval_kernel.h): the full COO value array (pinned host memory, constants in place)
  // and, per phase, the first entry of the eight groups of runs in it; null: the compact layout
  double* coo_full;
  const int32_t* coo;   // [8 * S]
  int32_t split_vel;    // latency form, whole evaluations only: velocity sweep k runs in the wavefront of position sweep k
  // fused aero rows (AERO instantiation of the cooperative form only; null otherwise): per-phase records, the batch's output
  // records [B][aero_ld]
  const AeroPhaseDev* aero_ph;
  double* aero_out;
  int64_t aero_ld;
};

}  // namespace gel
