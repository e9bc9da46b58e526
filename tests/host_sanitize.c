/* Host-side walk of the C-ABI under AddressSanitizer + UBSan (CPU build only: GPU ASan is not available on
 * the pool).  Exercises everything a host-only handle (device = GEL_DEVICE_NONE) can do -- LGR generator,
 * problem creation with ragged phases, dims, the 13 pattern blocks, constants, compact index map, work-item
 * list, aero pattern -- plus the argument-error paths.  Prints HOST_SANITIZE_OK on success. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/gelato_amd.h"

#define REQUIRE(c)                                                          \
  do {                                                                      \
    if (!(c)) { fprintf(stderr, "REQUIRE failed %s:%d: %s (%s)\n", __FILE__, __LINE__, #c, gel_last_error()); return 1; } \
  } while (0)

int main(void) {
  /* LGR nodes / differentiation matrices: row sums 0, D . tau_x = 1 */
  const int ns[] = {2, 3, 5, 16, 64, 100};
  for (unsigned t = 0; t < sizeof ns / sizeof *ns; t++) {
    const int n = ns[t];
    double* tau = malloc(sizeof(double) * n);
    double* D = malloc(sizeof(double) * n * (n + 1));
    REQUIRE(gel_lgr_nodes(n, tau) == GEL_OK && gel_lgr_diffmat(n, D) == GEL_OK);
    REQUIRE(fabs(tau[n - 1] - 1.0) < 1e-15);
    for (int k = 0; k < n; k++) {
      double s0 = 0.0, s1 = 0.0;
      for (int i = 0; i <= n; i++) {
        const double tx = (i == 0) ? -1.0 : tau[i - 1];
        s0 += D[k * (n + 1) + i];
        s1 += D[k * (n + 1) + i] * tx;
      }
      REQUIRE(fabs(s0) < 1e-10 && fabs(s1 - 1.0) < 1e-10);
    }
    free(tau); free(D);
  }
  REQUIRE(gel_lgr_nodes(1, NULL) != GEL_OK && gel_lgr_nodes(0, NULL) != GEL_OK);

  /* a ragged 7-phase problem touching every phase type */
  enum { S = 7 };
  const int32_t nn[S] = {2, 3, 100, 2, 17, 64, 5};
  const double thrust[S] = {420000.0, 0.0, 420000.0, 30700.0, 0.0, 30700.0, 1000.0};
  const double mdot[S] = {140.0, 0.0, 140.0, 9.8, 0.0, 9.8, 0.3};
  const double area[S] = {2.21, 2.21, 2.21, 0.0, 0.0, 2.21, 0.0};
  const double nozzle[S] = {0.68, 0.0, 0.68, 0.0, 0.0, 0.1, 0.0};
  const int32_t on[S] = {1, 0, 1, 1, 0, 1, 1};
  const int32_t hold[S] = {1, 0, 0, 1, 1, 0, 0};
  const double wind[3][3] = {{0.0, 1.0, 2.0}, {5000.0, 3.0, -1.0}, {20000.0, 10.0, 4.0}};
  const double ca[2][2] = {{0.0, 0.3}, {5.0, 0.5}};
  gel_problem_desc d;
  memset(&d, 0, sizeof d);
  d.num_sections = S; d.num_nodes = nn; d.thrust = thrust; d.massflow = mdot; d.reference_area = area;
  d.nozzle_area = nozzle; d.engine_on = on; d.attitude_hold = hold;
  d.unit_mass = 27442.0; d.unit_position = 6378137.0; d.unit_velocity = 1000.0; d.unit_u = 1.0; d.unit_t = 597.0;
  d.dx = 1e-8; d.wind_rows = 3; d.wind_table = &wind[0][0]; d.ca_rows = 2; d.ca_table = &ca[0][0];
  d.device = GEL_DEVICE_NONE;
  gel_problem* p = NULL;
  REQUIRE(gel_problem_create(&d, &p) == GEL_OK && p);
  gel_dims dm;
  REQUIRE(gel_problem_dims(p, &dm) == GEL_OK);
  int N = 0;
  for (int i = 0; i < S; i++) N += nn[i];
  REQUIRE(dm.S == S && dm.N == N && dm.M == N + S && dm.num_vars == 11 * (N + S) + 2 * N + S + 1);
  REQUIRE(dm.stored_bytes == 8 * ((int64_t)11 * N + dm.num_var_entries) && dm.algorithmic_bytes > dm.stored_bytes);
  int64_t tot = 0;
  for (int b = 0; b < GEL_NUM_BLOCKS; b++) {
    int32_t* r = malloc(sizeof(int32_t) * (size_t)(dm.block_nnz[b] + 1));
    int32_t* c = malloc(sizeof(int32_t) * (size_t)(dm.block_nnz[b] + 1));
    REQUIRE(gel_pattern(p, b, r, c) == GEL_OK);
    for (int64_t k = 0; k < dm.block_nnz[b]; k++)
      REQUIRE(r[k] >= 0 && r[k] < dm.block_shape[b][0] && c[k] >= 0 && c[k] < dm.block_shape[b][1]);
    tot += dm.block_nnz[b];
    free(r); free(c);
  }
  REQUIRE(tot == dm.total_nnz);
  REQUIRE(gel_pattern(p, GEL_NUM_BLOCKS, NULL, NULL) != GEL_OK);
  double* cv = malloc(sizeof(double) * (size_t)dm.total_nnz);
  int64_t* vi = malloc(sizeof(int64_t) * (size_t)dm.num_var_entries);
  REQUIRE(gel_const_values(p, cv) == GEL_OK && gel_var_index(p, vi) == GEL_OK);
  for (int64_t s = 0; s < dm.num_var_entries; s++) REQUIRE(vi[s] >= 0 && vi[s] < dm.total_nnz && cv[vi[s]] == 0.0);
  free(cv); free(vi);
  for (int i = 0; i < S; i++) {
    double* D = malloc(sizeof(double) * nn[i] * (nn[i] + 1));
    double* tau = malloc(sizeof(double) * nn[i]);
    REQUIRE(gel_problem_D(p, i, D) == GEL_OK && gel_problem_tau(p, i, tau) == GEL_OK);
    free(D); free(tau);
  }
  REQUIRE(gel_problem_D(p, S, NULL) != GEL_OK);
  int32_t nch = 0;
  REQUIRE(gel_num_chunks(p, &nch) == GEL_OK && nch == 1 + 1 + 2 + 1 + 1 + 1 + 1);
  int32_t* cp = malloc(sizeof(int32_t) * nch);
  REQUIRE(gel_chunk_phase(p, cp) == GEL_OK && cp[2] == 2 && cp[3] == 2 && cp[nch - 1] == S - 1);
  free(cp);
  /* aero rows: "all" on phase 2 (101 rows), "initial" on phase 5 (1 row) */
  const int32_t aph[2] = {2, 5}, aall[2] = {1, 0};
  const double alim[2] = {0.2, 0.1};
  for (int kind = 0; kind < 3; kind++) {
    REQUIRE(gel_aero_configure(p, kind, 2, aph, aall, alim) == GEL_OK);
    int32_t nrow = 0;
    int64_t nnz4[4];
    REQUIRE(gel_aero_dims(p, kind, &nrow, nnz4) == GEL_OK && nrow == 102);
    for (int v = 0; v < 4; v++) {
      int32_t* r = malloc(sizeof(int32_t) * (size_t)(nnz4[v] + 1));
      int32_t* c = malloc(sizeof(int32_t) * (size_t)(nnz4[v] + 1));
      REQUIRE(gel_aero_pattern(p, kind, v, r, c) == GEL_OK);
      for (int64_t k = 0; k < nnz4[v]; k++) REQUIRE(r[k] >= 0 && r[k] < nrow && c[k] >= 0);
      free(r); free(c);
    }
  }
  REQUIRE(gel_aero_configure(p, 3, 0, NULL, NULL, NULL) != GEL_OK);
  /* a host-only handle never evaluates (no CPU fallback) */
  double* x = calloc((size_t)dm.num_vars, sizeof(double));
  double* res = calloc((size_t)11 * N, sizeof(double));
  REQUIRE(gel_eval_residual(p, x, res) == GEL_ERR_HIP && strstr(gel_last_error(), "host-only"));
  REQUIRE(gel_eval_batch(p, 4, x, res, NULL) == GEL_ERR_HIP);
  free(x); free(res);
  REQUIRE(gel_problem_destroy(p) == GEL_OK);
  /* invalid descriptors are refused, not dereferenced */
  gel_problem* q = NULL;
  gel_problem_desc bad = d;
  bad.num_sections = 0;
  REQUIRE(gel_problem_create(&bad, &q) != GEL_OK && q == NULL);
  bad = d; bad.num_nodes = NULL;
  REQUIRE(gel_problem_create(&bad, &q) != GEL_OK);
  bad = d; bad.wind_rows = 1;
  REQUIRE(gel_problem_create(&bad, &q) != GEL_OK);
  const double dup[3][3] = {{0.0, 1.0, 2.0}, {5000.0, 3.0, -1.0}, {5000.0, 10.0, 4.0}};   /* repeated knot */
  bad = d; bad.wind_table = &dup[0][0];
  REQUIRE(gel_problem_create(&bad, &q) != GEL_OK && strstr(gel_last_error(), "strictly"));
  REQUIRE(gel_problem_create(NULL, &q) != GEL_OK);
  printf("HOST_SANITIZE_OK\n");
  return 0;
}
