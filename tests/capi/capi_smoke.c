/* capi_smoke.c -- the C ABI of include/hymls_mi.h used from plain C (no Python, no torch): what a C / Fortran / cgo
 * host would do.  Laplace 16^3, one level (exact inverse): x = P^{-1} (K x_ex) must reproduce x_ex.
 * Build: gcc -O2 -I include tests/capi/capi_smoke.c -o capi_smoke -L hymls_amd -lhymls_mi -Wl,-rpath,$PWD/hymls_amd -lm */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include "hymls_mi.h"

int main(void) {
  const int n = 16;
  int64_t nrows = 0, nnz = 0;
  if (hymls_mi_generate_matrix(0, n, n, n, 0.0, 0.0, &nrows, &nnz, NULL, NULL, NULL)) return 2;
  int32_t* rp = malloc((nrows + 1) * sizeof *rp);
  int32_t* ci = malloc(nnz * sizeof *ci);
  double* va = malloc(nnz * sizeof *va);
  double *tv = malloc(nrows * sizeof *tv), *xe = malloc(nrows * sizeof *xe), *b = calloc(nrows, sizeof *b), *x = malloc(nrows * sizeof *x);
  hymls_mi_generate_matrix(0, n, n, n, 0.0, 0.0, &nrows, &nnz, rp, ci, va);
  hymls_mi_generate_testvector(nrows, rp, ci, va, tv);
  hymls_mi_params p;
  hymls_mi_default_params(&p);
  p.nx = p.ny = p.nz = n; p.dim = 3; p.equations = 0; p.sx = 4; p.levels = 0;
  hymls_mi_t* h = NULL;
  int ierr = hymls_mi_create(&h, &p, 0);
  if (ierr) { printf("create: %d %s\n", ierr, hymls_mi_last_error(h)); return 3; }
  if ((ierr = hymls_mi_set_matrix_csr(h, nrows, rp, ci, va)) || (ierr = hymls_mi_set_testvector(h, tv))) { printf("set: %d %s\n", ierr, hymls_mi_last_error(h)); return 4; }
  if (hymls_mi_apply_inverse(h, b, nrows, x, nrows, 1, 0) != -1) { printf("ApplyInverse before Compute must fail with -1\n"); return 5; }
  if ((ierr = hymls_mi_compute(h))) { printf("compute: %d %s\n", ierr, hymls_mi_last_error(h)); return 6; }
  unsigned s = 12345u;
  for (int64_t i = 0; i < nrows; i++) { s = s * 1664525u + 1013904223u; xe[i] = (double)(s >> 8) / (double)(1u << 24) * 2.0 - 1.0; }
  for (int64_t i = 0; i < nrows; i++) for (int32_t e = rp[i]; e < rp[i + 1]; e++) b[i] += va[e] * xe[ci[e]];
  if ((ierr = hymls_mi_apply_inverse(h, b, nrows, x, nrows, 1, 0))) { printf("apply: %d %s\n", ierr, hymls_mi_last_error(h)); return 7; }
  double err = 0.0;
  for (int64_t i = 0; i < nrows; i++) err = fmax(err, fabs(x[i] - xe[i]));
  printf("CAPI_SMOKE levels %d size %lld schur %lld max error %.3e\n", hymls_mi_num_levels(h), (long long)hymls_mi_level_size(h, 0),
         (long long)hymls_mi_level_schur_size(h, 0), err);
  hymls_mi_destroy(h);
  return err < 1e-10 ? 0 : 1;
}
