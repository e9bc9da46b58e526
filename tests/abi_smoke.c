/* Plain-C consumer of the C-ABI (include/gelato_amd.h): proves the boundary needs nothing but a C compiler.
 * Builds a small 2-phase problem, checks dims / pattern on a host-only handle, and -- when a GPU is
 * present (argv[1] == "gpu") -- evaluates residual + Jacobian and checks a few structural invariants.
 *   gcc -std=c11 -I include tests/abi_smoke.c -o abi_smoke -L gelato_amd -lgelato_amd -Wl,-rpath,$PWD/gelato_amd -lm
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "gelato_amd.h"

#define CHECK(c)                                                                     \
  do {                                                                               \
    if (!(c)) { fprintf(stderr, "FAIL %s:%d: %s (%s)\n", __FILE__, __LINE__, #c, gel_last_error()); return 1; } \
  } while (0)

int main(int argc, char** argv) {
  const int gpu = argc > 1 && strcmp(argv[1], "gpu") == 0;
  int32_t nodes[2] = {5, 70};
  double thrust[2] = {420000.0, 30700.0}, mdot[2] = {140.9, 9.88}, area[2] = {2.21, 0.0}, nozzle[2] = {0.68, 0.0};
  int32_t eng[2] = {1, 1}, hold[2] = {1, 0};
  double wind[3][3] = {{-1e8, 0, 0}, {3000.0, 0.0, 10.0}, {1e10, 0, 0}};
  double ca[3][2] = {{0.0, 0.3}, {1.0, 0.65}, {100.0, 0.3}};
  gel_problem_desc d;
  memset(&d, 0, sizeof d);
  d.num_sections = 2; d.num_nodes = nodes; d.thrust = thrust; d.massflow = mdot; d.reference_area = area;
  d.nozzle_area = nozzle; d.engine_on = eng; d.attitude_hold = hold;
  d.unit_mass = 27442.0; d.unit_position = 6378137.0; d.unit_velocity = 1000.0; d.unit_u = 1.0; d.unit_t = 600.0;
  d.dx = 1e-8; d.wind_rows = 3; d.wind_table = &wind[0][0]; d.ca_rows = 3; d.ca_table = &ca[0][0];
  d.device = gpu ? 0 : GEL_DEVICE_NONE;

  gel_problem* p = NULL;
  CHECK(gel_problem_create(&d, &p) == GEL_OK);
  gel_dims dm;
  CHECK(gel_problem_dims(p, &dm) == GEL_OK);
  CHECK(dm.S == 2 && dm.N == 75 && dm.M == 77 && dm.num_vars == 11 * 77 + 2 * 75 + 3);
  /* distinct x-dependent values: phase 0 air + hold = 39 per node (the tf column of vel / t is the negative of its t0 column:
     closed form, gel_eval_kernel.h), phase 1 NoAir + free = 30 + 10 per node (quaternion kinematics in closed form: six values
     and the t0 column instead of twenty), + 1 scalar per phase */
  CHECK(dm.num_var_entries == 39 * 5 + 1 + 40 * 70 + 1);
  CHECK(dm.stored_bytes == 8 * ((int64_t)11 * 75 + dm.num_var_entries));
  /* SURVEY.md 8(d) A_min: every x-dependent value the reference computes: 48 resp. 71 per node */
  CHECK(dm.algorithmic_bytes == 8 * ((int64_t)dm.num_vars + 11 * 75 + 48 * 5 + 71 * 70));
  {
    /* the gather map covers exactly those 48 * 5 + 71 * 70 entries */
    int32_t* src = malloc(sizeof(int32_t) * dm.total_nnz);
    int64_t nvar = 0;
    CHECK(gel_full_source(p, src) == GEL_OK);
    for (int64_t k = 0; k < dm.total_nnz; k++) {
      CHECK(src[k] >= -2 - (int32_t)dm.num_var_entries + 1 && src[k] < (int32_t)dm.num_var_entries);
      nvar += src[k] != -1;
    }
    /* 8 of the 16 diagonal-block quaternion entries per node of the free-attitude phase are constants of the pattern */
    CHECK(nvar == 48 * 5 + (71 - 8) * 70);
    free(src);
  }
  int32_t nch = 0;
  CHECK(gel_num_chunks(p, &nch) == GEL_OK && nch == 3); /* 5 -> 1 item, 70 -> 2 items */

  /* pattern: in-range, no duplicate (row, col) inside a block */
  for (int b = 0; b < GEL_NUM_BLOCKS; b++) {
    int64_t nnz = dm.block_nnz[b];
    int32_t* r = malloc(sizeof(int32_t) * (nnz ? nnz : 1));
    int32_t* c = malloc(sizeof(int32_t) * (nnz ? nnz : 1));
    CHECK(gel_pattern(p, b, r, c) == GEL_OK);
    for (int64_t k = 0; k < nnz; k++)
      CHECK(r[k] >= 0 && r[k] < dm.block_shape[b][0] && c[k] >= 0 && c[k] < dm.block_shape[b][1]);
    free(r); free(c);
  }
  {
    /* the packed unit-shard exchange layout is a host computation: three ranks over the 12 units (3 work items x 4 parts) */
    const int32_t ub[4] = {0, 3, 6, 12};
    int64_t width = 0;
    int64_t* rpos = malloc(sizeof(int64_t) * 11 * dm.N);
    int64_t* jpos = malloc(sizeof(int64_t) * dm.num_var_entries);
    int32_t* rown = malloc(sizeof(int32_t) * 11 * dm.N);
    int32_t* jown = malloc(sizeof(int32_t) * dm.num_var_entries);
    CHECK(gel_shard_plan(p, 3, ub, &width, rpos, jpos) == GEL_OK && width > 0 && width % 2 == 0);
    CHECK(gel_unit_owner(p, rown, jown) == GEL_OK);
    char* seen = calloc((size_t)(3 * width), 1);
    for (int64_t i = 0; i < 11 * dm.N + dm.num_var_entries; i++) {
      const int64_t ps = i < 11 * dm.N ? rpos[i] : jpos[i - 11 * dm.N];
      const int32_t un = i < 11 * dm.N ? rown[i] : jown[i - 11 * dm.N];
      CHECK(ps >= 0 && ps < 3 * width && !seen[ps]);            /* every entry has a place of its own ... */
      seen[ps] = 1;
      CHECK(ps / width == (un < 3 ? 0 : (un < 6 ? 1 : 2)));       /* ... in the slice of the rank that holds its unit */
    }
    const int32_t bad[4] = {0, 3, 2, 12};
    CHECK(gel_shard_plan(p, 3, bad, &width, NULL, NULL) == GEL_ERR_ARG);          /* ranges must be ordered and cover every unit */
    CHECK(gel_shard_plan(p, 3, ub, &width, NULL, NULL) == GEL_OK);
    free(seen); free(rpos); free(jpos); free(rown); free(jown);
  }
  double D5[5 * 6], tau5[5];
  CHECK(gel_problem_D(p, 0, D5) == GEL_OK && gel_problem_tau(p, 0, tau5) == GEL_OK && tau5[4] == 1.0);
  for (int j = 0; j < 5; j++) { double s = 0; for (int i = 0; i < 6; i++) s += D5[j * 6 + i]; CHECK(fabs(s) < 1e-12); }

  double* x = malloc(sizeof(double) * dm.num_vars);
  double* res = malloc(sizeof(double) * 11 * dm.N);
  double* vals = malloc(sizeof(double) * dm.total_nnz);
  /* the forward-difference blocks: described by any handle */
  int64_t brow[2], bcol[2], brow0[2], boff[3];
  CHECK(gel_jac_fd_block_dims(p, 2, brow, bcol, brow0, boff) == GEL_OK && brow[0] == 15 && bcol[0] == 78 && brow0[1] == 15 &&
        boff[2] == boff[1] + brow[1] * bcol[1]);
  {
    int32_t bc[78];
    CHECK(gel_jac_fd_block_cols(p, 0, bc) == GEL_OK && bc[0] == 0 && bc[6] == dm.M && bc[77] == dm.num_vars - 2);
    CHECK(gel_jac_fd_block_cols(p, 2, bc) == GEL_ERR_ARG);
  }
  {
    /* aero path constraints on phase 0, all nodes: the per-vector record of gel_eval_batch_aero_device and its gather map [r6] */
    const int32_t aph[1] = {0}, aall[1] = {1};
    const double alim[1] = {0.2};
    int32_t nrow = 0;
    int64_t nnz4[4], width = 0, oc[6], oj[6];
    CHECK(gel_aero_configure(p, 0, 1, aph, aall, alim) == GEL_OK && gel_aero_dims(p, 0, &nrow, nnz4) == GEL_OK && nrow > 1);
    CHECK(gel_aero_record_layout(p, &width, oc, oj) == GEL_OK && width >= nrow * 11 && width % 8 == 0);
    int64_t* ci = malloc(sizeof(int64_t) * nrow);
    int64_t* ji = malloc(sizeof(int64_t) * nnz4[0]);
    CHECK(gel_aero_record_map(p, 0, -1, ci) == GEL_OK && gel_aero_record_map(p, 0, 0, ji) == GEL_OK);
    for (int k = 0; k < nrow; k++) CHECK(ci[k] >= 0 && ci[k] < width);
    for (int64_t k = 0; k < nnz4[0]; k++) CHECK(ji[k] >= 0 && ji[k] < width);
    CHECK(gel_aero_record_map(p, 3, 0, ji) == GEL_ERR_ARG);
    CHECK(gel_aero_configure(p, 0, 0, NULL, NULL, NULL) == GEL_OK);   /* and off again */
    free(ci); free(ji);
  }
  if (!gpu) {
    CHECK(gel_eval_batch_aero_device(p, 1, x, res, vals, vals, NULL) == GEL_ERR_HIP);
    CHECK(gel_eval_residual(p, x, res) == GEL_ERR_HIP); /* host-only handles never evaluate */
    CHECK(gel_jac_fd_blocks(p, 2, x, vals) == GEL_ERR_HIP && gel_jac_fd_device(p, 2, x, vals, 1, NULL) == GEL_ERR_HIP);
    CHECK(gel_eval_shard_packed_device(p, 1, x, res, 0, 1, 2, NULL) == GEL_ERR_HIP);
    CHECK(gel_eval_full_device(p, 1, x, res, vals, vals, NULL) == GEL_ERR_HIP);
    {
      static double tx[77], table[77 * GEL_OUTPUT_COLUMNS];
      CHECK(gel_output_table(p, x, tx, 42.5, 143.4, table) == GEL_ERR_HIP);
      CHECK(gel_output_table(p, x, NULL, 42.5, 143.4, table) == GEL_ERR_ARG);
      CHECK(GEL_OUT_MACH == GEL_OUTPUT_COLUMNS - 1);
    }
    CHECK(gel_problem_destroy(p) == GEL_OK);
    printf("abi_smoke host OK\n");
    return 0;
  }
  /* a plausible state: 200 km circular-ish positions, unit quaternion, 25 t mass */
  int M = dm.M, N = dm.N;
  for (int i = 0; i < M; i++) {
    double th = 0.7 + 0.001 * i;
    x[i] = 0.9 - 0.005 * i;
    x[M + 3 * i] = 1.03 * cos(th); x[M + 3 * i + 1] = 1.03 * sin(th) * 0.8; x[M + 3 * i + 2] = 1.03 * sin(th) * 0.6;
    x[4 * M + 3 * i] = -7.0 * sin(th); x[4 * M + 3 * i + 1] = 7.0 * cos(th) * 0.8; x[4 * M + 3 * i + 2] = 7.0 * cos(th) * 0.6;
    x[7 * M + 4 * i] = 0.5; x[7 * M + 4 * i + 1] = 0.5; x[7 * M + 4 * i + 2] = -0.5; x[7 * M + 4 * i + 3] = 0.5;
  }
  for (int i = 0; i < 2 * N; i++) x[11 * M + i] = 0.01 * (i % 7);
  x[11 * M + 2 * N] = 0.0; x[11 * M + 2 * N + 1] = 0.3; x[11 * M + 2 * N + 2] = 1.0;
  CHECK(gel_eval(p, x, res, vals, 1) == GEL_OK);
  for (int i = 0; i < 11 * N; i++) CHECK(isfinite(res[i]));
  for (int64_t k = 0; k < dm.total_nnz; k++) CHECK(isfinite(vals[k]));
  /* phase 0 is a hold phase: its quaternion defect is q[j+1] - q[0] = 0 for this constant attitude */
  for (int j = 0; j < 5 * 4; j++) CHECK(res[7 * N + j] == 0.0);
  {
    /* velocity rows by forward difference: the blocks are the dense matrix without its zeros */
    double* J = calloc((size_t)3 * N * dm.num_vars, sizeof(double));
    double* blk = malloc(sizeof(double) * boff[2]);
    int32_t bc[13 * 70 + 13];
    CHECK(gel_jac_fd(p, 2, x, J) == GEL_OK && gel_jac_fd_blocks(p, 2, x, blk) == GEL_OK);
    for (int ph = 0; ph < 2; ph++) {
      CHECK(gel_jac_fd_block_cols(p, ph, bc) == GEL_OK);
      for (int64_t r = 0; r < brow[ph]; r++)
        for (int64_t c = 0; c < bcol[ph]; c++) CHECK(blk[boff[ph] + r * bcol[ph] + c] == J[(brow0[ph] + r) * dm.num_vars + bc[c]]);
    }
    free(J); free(blk);
  }
  /* a NaN in x is reported, not swallowed */
  x[M + 4] = NAN;
  CHECK(gel_eval_residual(p, x, res) == GEL_NONFINITE);
  CHECK(gel_problem_destroy(p) == GEL_OK);
  printf("abi_smoke gpu OK\n");
  return 0;
}
