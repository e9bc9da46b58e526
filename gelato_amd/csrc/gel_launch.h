// gel_launch.h -- host-callable launchers of the kernels in gel_kernels.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

#include "gel_device.h"

namespace gel {

// the form of the fused kernel a launch takes: <JAC, MFMA, SPLIT, PACK> and its wavefront count
struct EvalForm { bool jac, mfma, split, pack; long long waves; };
EvalForm eval_form(const ProblemDev& P, int B, bool want_res, bool want_jac);
hipError_t launch_eval(const ProblemDev& P, int B, const double* d_x, double* d_res, double* d_jvar, hipStream_t s);
// defect groups + the aero rows of the aerodynamic phases' nodes 1 .. n in ONE launch (gel_eval_kernel.h, AERO instantiation)
bool eval_aero_fusable(const ProblemDev& P, int B);
hipError_t launch_eval_aero(const ProblemDev& P, int B, const double* d_x, double* d_res, double* d_jvar, hipStream_t s);
hipError_t launch_expand(long long nnz, long long V, int B, const double* cval, const int32_t* src,
                         const double* d_jvar, double* d_full, hipStream_t s);
// x-dependent entries only, into a full COO buffer that holds the constants (launch_fill_full lays them down)
hipError_t launch_update_full(long long nnz, long long V, int nvar, int B, const int32_t* vdst, const int32_t* vsrc,
                              int nlines, const int32_t* vline, const int32_t* src, const double* cval,
                              const double* d_jvar, double* d_full, hipStream_t s);
hipError_t launch_fill_full(long long nnz, int B, const double* cval, double* d_full, hipStream_t s);
// packed unit-shard exchange buffer [nranks][B][width] -> the ordinary res [B][nres] / jvar [B][V] layouts (either may be null)
hipError_t launch_shard_unpack(long long nres, long long V, long long width, int B, const int64_t* pos, const double* out,
                               double* d_res, double* d_jvar, hipStream_t s);
hipError_t launch_perturb_local(int nloc, double dx, const double* d_x, const int32_t* d_colmap, double* d_Xp, hipStream_t s);
hipError_t launch_quotient_local(int nloc, int nres, int roff, int nrows, double dx, const double* d_res, double* d_J,
                                 long long ldJ, int row0, const int32_t* d_colmap, hipStream_t s);
hipError_t launch_rhs_vel(bool air, int n, const double* mass_e, const double* pos_e, const double* vel_e,
                          const double* quat, const double* t, const double* tables, int Kw, int Kc, double thrust,
                          double area, double nozzle, double um, double up, double uv, double barC20, double* out,
                          hipStream_t s);
hipError_t launch_rhs_quat(int n, const double* quat, const double* u_e, double unit_u, double* out, hipStream_t s);
hipError_t launch_point(int kind, int n, const double* in, const double* aux, int aux_rows, double* out,
                        hipStream_t s);

// outputs of one aero launch: per kind (0 alpha, 1 q, 2 q-alpha) the constraint vector and the COO values, or null
struct AeroLaunchOut { double* con[3]; double* jac[3]; int32_t nrows[3]; };
hipError_t launch_aero(const ProblemDev& P, int nnodes, const AeroNodeDev* nodes, int B, const double* d_x,
                       const AeroLaunchOut& out, hipStream_t s, long long ld = 0, bool spec_major = false);

hipError_t launch_aero_wide(const ProblemDev& P, int nnodes, const AeroNodeDev* nodes, int B, const double* d_x,
                            const AeroLaunchOut& out, long long ld, hipStream_t s);

// one callback = one launch: defect groups (split form) + aero kinds + row table as workgroup ranges of one grid; aero / d_con
// may be null (that part is left out)
hipError_t launch_callback(const ProblemDev& P, bool want_jac, const double* d_x, double* d_res, double* d_jvar,
                           int nnodes, const AeroNodeDev* nodes, const AeroLaunchOut* aero,
                           int nlin, const LinRowDev* lin, int nfn, const FnRowDev* fr, double* d_con, double* d_jfn, hipStream_t s);
hipError_t launch_rows(const ProblemDev& P, int nlin, const LinRowDev* lin, int nfn, const FnRowDev* fr, int B,
                       const double* d_x, double* d_con, double* d_jfn, hipStream_t s);

// post-processing table (output_result.py): kOutputColumns values per state node, see include/gelato_amd.h
constexpr int kOutputColumns = 34;
hipError_t launch_output(const ProblemDev& P, int M, const double* d_x, const double* d_tx, const int32_t* d_node_sec,
                         double lat0, double lon0, double* d_out, hipStream_t s);

// US-1976 layer table as the kernels expect it: Lmb[11] | Tmb[11] | Pb[11] | R[11] | pexp[11] | gR[11] | Hb[11] | 1/Tmb[11]
constexpr int kAtmTableDoubles = 88;  // must equal kAtmDoubles of gel_physics.h (static_assert in gel_kernels.hip)
void fill_atmosphere_table(double* atm);
const char* check_tables(const double* wind, int Kw, const double* ca, int Kc);
std::vector<double> build_tables(const double* wind, int Kw, const double* ca, int Kc);
void append_rows_and_slopes(std::vector<double>& rows_out, std::vector<double>& slopes_out, const double* tab, int K, int w);
// doubles of the staged tables (atmosphere | wind rows | CA rows | wind slopes | CA slopes), as gel_physics.h table_doubles()
inline size_t staged_table_doubles(int Kw, int Kc) { return (size_t)kAtmTableDoubles + 3 * (size_t)Kw + 2 * (size_t)Kc + 2 * (size_t)(Kw - 1) + (size_t)(Kc - 1); }

}  // namespace gel
