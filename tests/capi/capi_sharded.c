/* Plain-C caller of the SHARDED entry points of include/hymls_mi.h over the MPI transport of include/hymls_mi_mpi.h
 * (no C++, no Python, no torch in the process): what an MPI application that is not built on Epetra binds to.
 *   mpiexec -n P capi_sharded [MPI|RCCL]
 * Every rank: create -> set_comm -> required_rows -> generate its rows -> set_matrix_rows -> Compute -> owned_rows ->
 * ApplyInverse on its part of a global vector.  Rank 0 gathers the parts and compares with a one-rank handle on the
 * full problem (Laplace 16^3, separator length 4, two-level; reference lifecycle src/HYMLS_Preconditioner.hpp:98-127). */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "hymls_mi_mpi.h"

#define CHECK(c) do { if (!(c)) { printf("[rank %d] FAILED %s:%d: %s (%s)\n", rank, __FILE__, __LINE__, #c, h ? hymls_mi_last_error(h) : ""); fflush(stdout); MPI_Abort(MPI_COMM_WORLD, 1); } } while (0)

static double val(int gid) { unsigned s = 2654435761u * (unsigned)(gid + 1); s ^= s >> 13; s *= 1274126177u; s ^= s >> 16; return (double)(s & 0xffffff) / (1 << 24) * 2.0 - 1.0; }

int main(int argc, char** argv) {
  int rank = 0, size = 1, px, py, pz, q, i;
  const int n = 16, N = n * n * n;
  const int use_rccl = argc > 1 && strcmp(argv[1], "RCCL") == 0;
  hymls_mi_t* h = NULL;
  hymls_mi_params p;
  hymls_mi_mpi_transport* t = NULL;
  void* nccl = NULL;
  int64_t nreq = 0, nown = 0, nnz = 0;
  int32_t *req, *own, *rp, *ci;
  double *va, *tv, *b, *x;
  MPI_Init(&argc, &argv);
  MPI_Comm_rank(MPI_COMM_WORLD, &rank);
  MPI_Comm_size(MPI_COMM_WORLD, &size);
  hymls_mi_default_params(&p);
  p.nx = p.ny = p.nz = n; p.dim = 3; p.equations = 0; p.sx = 4; p.levels = 1;
  CHECK(hymls_mi_rank_grid(size, &px, &py, &pz) == 0 && px * py * pz == size);
  CHECK(hymls_mi_create(&h, &p, 0) == 0);
  if (use_rccl) CHECK(hymls_mi_set_comm_rccl_mpi(h, MPI_COMM_WORLD, 0, px, py, pz, &nccl) == 0);
  else CHECK(hymls_mi_set_comm_mpi(h, MPI_COMM_WORLD, px, py, pz, &t) == 0);
  CHECK(hymls_mi_comm_selftest(h) == 0);
  CHECK(hymls_mi_required_rows(h, &nreq, NULL) == 0);
  req = (int32_t*)malloc((nreq + 1) * sizeof *req);
  CHECK(hymls_mi_required_rows(h, &nreq, req) == 0);
  CHECK(hymls_mi_generate_rows(0, n, n, n, (double)n * n, 1.0, nreq, req, &nnz, NULL, NULL, NULL) == 0);
  rp = (int32_t*)malloc((nreq + 1) * sizeof *rp); ci = (int32_t*)malloc((nnz + 1) * sizeof *ci); va = (double*)malloc((nnz + 1) * sizeof *va);
  CHECK(hymls_mi_generate_rows(0, n, n, n, (double)n * n, 1.0, nreq, req, &nnz, rp, ci, va) == 0);
  CHECK(hymls_mi_set_matrix_rows(h, nreq, req, rp, ci, va) == 0);
  tv = (double*)malloc((nreq + 1) * sizeof *tv);
  for (i = 0; i < nreq; i++) tv[i] = 1.0;
  CHECK(hymls_mi_set_testvector(h, tv) == 0);
  CHECK(hymls_mi_apply_inverse(h, tv, nreq, tv, nreq, 1, 0) == -1);      /* before Compute */
  CHECK(hymls_mi_compute(h) == 0 && hymls_mi_is_initialized(h) && hymls_mi_is_computed(h));
  CHECK(hymls_mi_owned_rows(h, &nown, NULL) == 0);
  own = (int32_t*)malloc((nown + 1) * sizeof *own);
  CHECK(hymls_mi_owned_rows(h, &nown, own) == 0);
  b = (double*)malloc((nown + 1) * sizeof *b); x = (double*)malloc((nown + 1) * sizeof *x);
  for (i = 0; i < nown; i++) b[i] = val(own[i]);
  CHECK(hymls_mi_apply_inverse(h, b, nown, x, nown, 1, 0) == 0);
  {
    /* gather (gid, x) on rank 0 */
    int *cnt = (int*)malloc(size * sizeof(int)), *dsp = (int*)malloc(size * sizeof(int)), mine = (int)nown, tot = 0;
    int32_t* gall; double* xall;
    MPI_Gather(&mine, 1, MPI_INT, cnt, 1, MPI_INT, 0, MPI_COMM_WORLD);
    if (rank == 0) for (q = 0; q < size; q++) { dsp[q] = tot; tot += cnt[q]; }
    gall = (int32_t*)malloc((tot + 1) * sizeof *gall); xall = (double*)malloc((tot + 1) * sizeof *xall);
    MPI_Gatherv(own, mine, MPI_INT32_T, gall, cnt, dsp, MPI_INT32_T, 0, MPI_COMM_WORLD);
    MPI_Gatherv(x, mine, MPI_DOUBLE, xall, cnt, dsp, MPI_DOUBLE, 0, MPI_COMM_WORLD);
    if (rank == 0) {
      hymls_mi_t* h1 = NULL;
      int64_t nr = 0, nz = 0;
      int32_t *rp1, *ci1; double *va1, *b1, *x1, *xs; char* seen;
      double err = 0, nrm = 0;
      CHECK(tot == N);                                       /* every row owned by exactly one rank */
      CHECK(hymls_mi_generate_matrix(0, n, n, n, (double)n * n, 1.0, &nr, &nz, NULL, NULL, NULL) == 0 && nr == N);
      rp1 = (int32_t*)malloc((N + 1) * sizeof *rp1); ci1 = (int32_t*)malloc(nz * sizeof *ci1); va1 = (double*)malloc(nz * sizeof *va1);
      CHECK(hymls_mi_generate_matrix(0, n, n, n, (double)n * n, 1.0, &nr, &nz, rp1, ci1, va1) == 0);
      CHECK(hymls_mi_create(&h1, &p, 0) == 0 && hymls_mi_set_matrix_csr(h1, N, rp1, ci1, va1) == 0 && hymls_mi_compute(h1) == 0);
      b1 = (double*)malloc(N * sizeof *b1); x1 = (double*)malloc(N * sizeof *x1); xs = (double*)malloc(N * sizeof *xs); seen = (char*)calloc(N, 1);
      for (i = 0; i < N; i++) b1[i] = val(i);
      CHECK(hymls_mi_apply_inverse(h1, b1, N, x1, N, 1, 0) == 0);
      for (i = 0; i < tot; i++) { CHECK(gall[i] >= 0 && gall[i] < N && !seen[gall[i]]); seen[gall[i]] = 1; xs[gall[i]] = xall[i]; }
      for (i = 0; i < N; i++) { if (fabs(xs[i] - x1[i]) > err) err = fabs(xs[i] - x1[i]); if (fabs(x1[i]) > nrm) nrm = fabs(x1[i]); }
      printf("%d ranks (%s transport), plain C: sharded ApplyInverse vs one rank: max diff %.2e (max |x| %.2e)\n", size, use_rccl ? "RCCL" : "MPI", err, nrm);
      CHECK(err <= 1e-12 * nrm);
      hymls_mi_destroy(h1);
      printf("CAPI_SHARDED_OK\n");
    }
  }
  if (t) hymls_mi_mpi_transport_free(t);
  hymls_mi_destroy(h);
  if (nccl) hymls_mi_rccl_comm_destroy(nccl);
  MPI_Finalize();
  return 0;
}
