/* The CPU oracle (oracle/gelato_oracle.c, compiled into this program) under AddressSanitizer + UBSan: residuals,
 * structured Jacobians, generic FD, batch evaluation and the aero constraints on a ragged problem with exactly
 * sized heap buffers, so that any out-of-range index in the restatement is caught.  Prints ORACLE_SANITIZE_OK. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../oracle/gelato_oracle.h"

#define REQUIRE(c) do { if (!(c)) { fprintf(stderr, "REQUIRE failed %s:%d: %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)

static double urand(unsigned* s) { *s = *s * 1664525u + 1013904223u; return (*s >> 8) / 16777216.0; }

int main(void) {
  enum { S = 5 };
  const int32_t nn[S] = {2, 3, 9, 2, 6};
  const double thrust[S] = {420000.0, 0.0, 420000.0, 30700.0, 30700.0};
  const double mdot[S] = {140.0, 0.0, 140.0, 9.8, 9.8};
  const double area[S] = {2.21, 2.21, 2.21, 0.0, 0.0};
  const double nozzle[S] = {0.68, 0.0, 0.68, 0.0, 0.0};
  const int32_t on[S] = {1, 0, 1, 1, 1};
  const int32_t hold[S] = {1, 0, 0, 1, 0};
  const double units[5] = {27442.0, 6378137.0, 1000.0, 1.0, 597.0};
  const double wind[3][3] = {{0.0, 1.0, 2.0}, {5000.0, 3.0, -1.0}, {20000.0, 10.0, 4.0}};
  const double ca[2][2] = {{0.0, 0.3}, {5.0, 0.5}};
  orc_problem* p = orc_problem_create(S, nn, thrust, mdot, area, nozzle, on, hold, units, 1e-8, -0.484165371736e-3,
                                      &wind[0][0], 3, &ca[0][0], 2, NULL, NULL);
  REQUIRE(p);
  int N = 0;
  for (int i = 0; i < S; i++) N += nn[i];
  const int M = N + S, nv = orc_num_vars(p);
  REQUIRE(nv == 11 * M + 2 * N + S + 1);
  double* x = malloc(sizeof(double) * nv);
  unsigned seed = 12345u;
  for (int i = 0; i < nv; i++) x[i] = 0.5 + urand(&seed);
  for (int i = 0; i < 3 * M; i++) x[M + i] = 0.6 + 0.1 * urand(&seed);          /* positions ~ 1.1 Earth radii */
  for (int i = 0; i <= S; i++) x[nv - S - 1 + i] = 0.1 * i;                      /* increasing knot times */
  double checksum = 0.0;
  for (int g = 0; g < 4; g++) {
    const int nr = orc_num_rows(p, g);
    double* r = malloc(sizeof(double) * nr);
    orc_residual(p, g, x, r);
    for (int i = 0; i < nr; i++) checksum += r[i];
    int64_t nnz = 0;
    for (int b = 0; b < orc_num_blocks(g); b++) nnz += orc_block_nnz(p, g, b);
    int32_t* rows = malloc(sizeof(int32_t) * (size_t)nnz);
    int32_t* cols = malloc(sizeof(int32_t) * (size_t)nnz);
    double* vals = malloc(sizeof(double) * (size_t)nnz);
    orc_jacobian(p, g, x, rows, cols, vals);
    int64_t off = 0;
    for (int b = 0; b < orc_num_blocks(g); b++) {
      int64_t shp[2];
      orc_block_shape(p, g, b, shp);
      for (int64_t k = 0; k < orc_block_nnz(p, g, b); k++, off++)
        REQUIRE(rows[off] >= 0 && rows[off] < shp[0] && cols[off] >= 0 && cols[off] < shp[1]);
    }
    double* J = malloc(sizeof(double) * (size_t)nr * nv);
    orc_jac_fd(p, g, x, J);
    for (size_t i = 0; i < (size_t)nr * nv; i++) checksum += J[i] * 1e-9;
    free(r); free(rows); free(cols); free(vals); free(J);
  }
  /* batch entry point, two threads */
  const int B = 3;
  const int64_t tot = orc_total_nnz(p);
  double* X = malloc(sizeof(double) * (size_t)B * nv);
  for (int b = 0; b < B; b++) memcpy(X + (size_t)b * nv, x, sizeof(double) * nv);
  double* R = malloc(sizeof(double) * (size_t)B * 11 * N);
  double* V = malloc(sizeof(double) * (size_t)B * (size_t)tot);
  orc_eval_batch(p, B, X, R, V, 2);
  REQUIRE(memcmp(R, R + 11 * N, sizeof(double) * 11 * N) == 0 && memcmp(V, V + tot, sizeof(double) * (size_t)tot) == 0);
  free(X); free(R); free(V);
  /* aero constraints */
  const int32_t aph[2] = {2, 0}, aall[2] = {1, 0};
  const double alim[2] = {0.2, 0.1};
  for (int kind = 0; kind < 3; kind++) {
    REQUIRE(orc_aero_configure(p, kind, 2, aph, aall, alim) == 0);
    const int nr = orc_aero_rows(p, kind);
    REQUIRE(nr == nn[2] + 1 + 1);
    double* c = malloc(sizeof(double) * nr);
    orc_aero_residual(p, kind, x, c);
    int64_t nnz = 0;
    for (int v = 0; v < 4; v++) nnz += orc_aero_nnz(p, kind, v);
    int32_t* rows = malloc(sizeof(int32_t) * (size_t)(nnz + 1));
    int32_t* cols = malloc(sizeof(int32_t) * (size_t)(nnz + 1));
    double* vals = malloc(sizeof(double) * (size_t)(nnz + 1));
    orc_aero_jacobian(p, kind, x, rows, cols, vals);
    for (int64_t k = 0; k < nnz; k++) REQUIRE(rows[k] >= 0 && rows[k] < nr);
    free(c); free(rows); free(cols); free(vals);
  }
  double gm[64];
  REQUIRE(M <= 64);
  orc_cost_jac(p, x, 1, gm);
  REQUIRE(orc_cost(p, x, 1) == -x[0] && orc_cost(p, x, 0) == x[nv - 1]);
  orc_problem_destroy(p);
  free(x);
  printf("ORACLE_SANITIZE_OK %g\n", checksum);
  return 0;
}
