// gel_kernels_aero.hip -- the AERO instantiation of the fused kernel (gel_eval_kernel.h) in a translation unit of its own: the
// defect groups of a batch AND the aero path constraints' rows of the aerodynamic phases' nodes in ONE launch
// (gel_eval_batch_aero_device; lib/con_aero.py:89-248,311-371 riding on lib/con_dynamics.py:216-496).
// Its own unit because it is compiled with its own scheduling strategy (Makefile, KFLAGS_AERO): under max-ilp, which the other
// instantiations are built with (-0.8 % launch time), this one spills 4..16 registers to scratch; under the default strategy it
// fits the 128 registers of four wavefronts per SIMD.
#include "gel_tables.h"
#include "gel_eval_kernel.h"

namespace gel {

// The fused launch with the aero rows riding along (P.aero_ph / aero_out / aero_ld set by the caller): only the cooperative form
// with derivatives and one vector per wavefront has that instantiation -- false: the caller launches the defect and the aero
// kernels separately (a handful of vectors, meshes of phases of at most 32 nodes, GEL_FLAG_FD_RECOMPUTE / GEL_FLAG_DX_VALU problems).
bool eval_aero_fusable(const ProblemDev& P, int B) {
  const EvalForm f = eval_form(P, B, true, true);
  return f.mfma && !f.split && !f.pack && !P.fd_recompute;
}
hipError_t launch_eval_aero(const ProblemDev& P, int B, const double* d_x, double* d_res, double* d_jvar, hipStream_t s) {
  if (B <= 0) return hipSuccess;
  if (!eval_aero_fusable(P, B) || !P.aero_ph || !P.aero_out || !d_res || !d_jvar) return hipErrorInvalidValue;
  // grid and LDS as launch_coop() of gel_kernels.hip (one vector per wavefront, four per workgroup)
  const unsigned nb = (unsigned)((B + 3) / 4);
  const unsigned grid = P.vmajor ? (unsigned)P.nchunks * 8u * ((nb + 7u) / 8u) : (unsigned)P.nchunks * nb;
  const size_t lds = sizeof(double) * ((size_t)P.park_off + (size_t)wave_lds_doubles(true, true, false, false, true) * (kBlock / 64));
  hipLaunchKernelGGL((eval_kernel<true, true, false, false, true, true, true>), dim3(grid), dim3(kBlock), lds, s, P, B, d_x, d_res, d_jvar);
  return hipGetLastError();
}


}  // namespace gel
