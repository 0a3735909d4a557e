/* hymls_mi_mpi.h -- MPI-side transports of a sharded hymls_mi handle (header-only, plain C, needs <mpi.h>).
 *
 * An MPI application (one rank per GPU: what the reference is, src/main.cpp:48-67 MPI_Init + Epetra_MpiComm) attaches
 * its communicator to a handle with one of
 *
 *   hymls_mi_set_comm_rccl_mpi(h, comm, device, px, py, pz, &nccl)   the fast path: the library's built-in RCCL
 *       transport (hymls_mi_set_comm_rccl); MPI only carries the 128 bytes of the RCCL id (MPI_Bcast) -- every halo /
 *       separator / V-sum exchange of Compute and ApplyInverse is then an ncclSend/ncclRecv group on the handle's stream
 *       over xGMI, replacing the Epetra_Import / Epetra_Export traffic of the reference
 *       (src/HYMLS_Preconditioner.cpp:304-336,978-979,1050-1052, src/HYMLS_SchurPreconditioner.cpp:1076-1078);
 *   hymls_mi_set_comm_mpi(h, comm, px, py, pz, &t)                   the portable path: the two callbacks of
 *       hymls_mi_comm on MPI_Alltoallv.  Host-side (setup) exchanges go straight through MPI; device segments are
 *       staged through host buffers (hymls_mi_copy_to_host / _to_device), so it works with any MPI library, with
 *       several ranks on one GPU, and with the test-only host simulator -- at the cost of two PCIe copies per exchange.
 *
 * Both are collective over `comm`; rank r of `comm` owns box (r % px, (r / px) % py, r / (px py)) of the grid
 * (hymls_mi_rank_grid gives the reference's CreatePIDMap layout for a number of ranks).
 */
#ifndef HYMLS_MI_MPI_H
#define HYMLS_MI_MPI_H

#include <mpi.h>
#include <stdlib.h>
#include <string.h>

#include "hymls_mi.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hymls_mi_mpi_transport {
  MPI_Comm comm;
  hymls_mi_t* h;
  int rank, size;
  char *hs, *hr;               /* host staging of device segments */
  int64_t cap_s, cap_r;
  void** arenas;               /* device memory handed to the library (freed by hymls_mi_mpi_transport_free) */
  int n_arenas;
  int64_t chunk;               /* bytes per peer and round (MPI counts are int) */
  hymls_mi_comm cb;
} hymls_mi_mpi_transport;

static inline int hymls_mi_mpi__grow(char** buf, int64_t* cap, int64_t need) {
  if (need <= *cap) return 0;
  free(*buf);
  *cap = need + need / 4 + 4096;
  *buf = (char*)malloc((size_t)*cap);
  if (!*buf) { *cap = 0; return -1; }
  return 0;
}

/* all-to-all of byte segments in rounds of at most t->chunk bytes per peer (the same number of rounds on every rank) */
static inline int hymls_mi_mpi__a2a_bytes(hymls_mi_mpi_transport* t, const char* send, const int64_t* sb, char* recv, const int64_t* rb) {
  const int P = t->size;
  int64_t big = 0, gbig = 0;
  int q, rounds, r, ierr = 0;
  int *sc, *sd, *rc, *rd;
  for (q = 0; q < P; q++) { if (sb[q] > big) big = sb[q]; if (rb[q] > big) big = rb[q]; }
  if (MPI_Allreduce(&big, &gbig, 1, MPI_INT64_T, MPI_MAX, t->comm) != MPI_SUCCESS) return -1;
  if (gbig == 0) return 0;
  rounds = (int)((gbig + t->chunk - 1) / t->chunk);
  if (rounds == 1) {
    int64_t so = 0, ro = 0;
    sc = (int*)malloc(4 * (size_t)P * sizeof(int));
    if (!sc) return -1;
    sd = sc + P; rc = sd + P; rd = rc + P;
    for (q = 0; q < P; q++) {
      if (so > 2147483647 || ro > 2147483647) { rounds = 2; break; }   /* displacements do not fit: peer-wise rounds below */
      sc[q] = (int)sb[q]; sd[q] = (int)so; rc[q] = (int)rb[q]; rd[q] = (int)ro;
      so += sb[q]; ro += rb[q];
    }
    if (rounds == 1) {
      ierr = MPI_Alltoallv((void*)send, sc, sd, MPI_BYTE, recv, rc, rd, MPI_BYTE, t->comm) != MPI_SUCCESS;
      free(sc);
      return ierr ? -1 : 0;
    }
    free(sc);
  }
  {
    /* large exchanges (the reduced matrix of a 256^3 run is GBs): point-to-point pieces of bounded size */
    MPI_Request* req = (MPI_Request*)malloc(2 * (size_t)P * sizeof(MPI_Request));
    MPI_Status* sta = (MPI_Status*)malloc(2 * (size_t)P * sizeof(MPI_Status));
    int64_t *so = (int64_t*)malloc(2 * (size_t)P * sizeof(int64_t)), *ro;
    if (!req || !sta || !so) { free(req); free(sta); free(so); return -1; }
    ro = so + P;
    { int64_t a = 0, b = 0; for (q = 0; q < P; q++) { so[q] = a; ro[q] = b; a += sb[q]; b += rb[q]; } }
    rounds = (int)((gbig + t->chunk - 1) / t->chunk);
    for (r = 0; r < rounds && !ierr; r++) {
      int nreq = 0;
      for (q = 0; q < P; q++) {
        const int64_t off = (int64_t)r * t->chunk;
        int64_t ns = sb[q] - off, nr = rb[q] - off;
        if (ns > t->chunk) ns = t->chunk;
        if (nr > t->chunk) nr = t->chunk;
        if (nr > 0) ierr |= MPI_Irecv(recv + ro[q] + off, (int)nr, MPI_BYTE, q, 7700 + (r & 63), t->comm, &req[nreq++]) != MPI_SUCCESS;
        if (ns > 0) ierr |= MPI_Isend((void*)(send + so[q] + off), (int)ns, MPI_BYTE, q, 7700 + (r & 63), t->comm, &req[nreq++]) != MPI_SUCCESS;
      }
      ierr |= MPI_Waitall(nreq, req, sta) != MPI_SUCCESS;
    }
    free(req); free(sta); free(so);
  }
  return ierr ? -1 : 0;
}

static inline int hymls_mi_mpi__alltoallv(void* ctx, const void* send, const int64_t* scnt, void* recv, const int64_t* rcnt,
                                   int32_t elem_bytes, int32_t on_device) {
  hymls_mi_mpi_transport* t = (hymls_mi_mpi_transport*)ctx;
  const int P = t->size;
  int64_t ns = 0, nr = 0;
  int q, ierr;
  int64_t* sb = (int64_t*)malloc(2 * (size_t)P * sizeof(int64_t));
  int64_t* rb = sb + P;
  if (!sb) return -1;
  for (q = 0; q < P; q++) { sb[q] = scnt[q] * elem_bytes; rb[q] = rcnt[q] * elem_bytes; ns += sb[q]; nr += rb[q]; }
  if (!on_device) {
    ierr = hymls_mi_mpi__a2a_bytes(t, (const char*)send, sb, (char*)recv, rb);
  } else {
    /* device segments: down to the host (ordered on the handle's stream, complete on return), MPI, up again */
    ierr = hymls_mi_mpi__grow(&t->hs, &t->cap_s, ns) || hymls_mi_mpi__grow(&t->hr, &t->cap_r, nr);
    if (!ierr && ns > 0) ierr = hymls_mi_copy_to_host(t->h, t->hs, send, ns);
    if (!ierr) ierr = hymls_mi_mpi__a2a_bytes(t, t->hs, sb, t->hr, rb);
    if (!ierr && nr > 0) ierr = hymls_mi_copy_to_device(t->h, recv, t->hr, nr);
  }
  free(sb);
  return ierr ? -1 : 0;
}

static inline void* hymls_mi_mpi__alloc(void* ctx, int64_t bytes) {
  hymls_mi_mpi_transport* t = (hymls_mi_mpi_transport*)ctx;
  void* p = hymls_mi_device_alloc(t->h, bytes);
  void** a;
  if (!p) return NULL;
  a = (void**)realloc(t->arenas, (size_t)(t->n_arenas + 1) * sizeof(void*));
  if (!a) { hymls_mi_device_free(t->h, p); return NULL; }
  t->arenas = a;
  t->arenas[t->n_arenas++] = p;
  return p;
}

/* attach `comm` to the handle through the MPI_Alltoallv transport; *out has to outlive the handle's use of it and is
 * released with hymls_mi_mpi_transport_free BEFORE hymls_mi_destroy (it frees device memory through the handle). */
static inline int hymls_mi_set_comm_mpi(hymls_mi_t* h, MPI_Comm comm, int px, int py, int pz, hymls_mi_mpi_transport** out) {
  hymls_mi_mpi_transport* t;
  int ierr;
  if (!h || !out) return -2;
  t = (hymls_mi_mpi_transport*)calloc(1, sizeof *t);
  if (!t) return -3;
  t->comm = comm; t->h = h;
  MPI_Comm_rank(comm, &t->rank);
  MPI_Comm_size(comm, &t->size);
  t->chunk = (int64_t)1 << 30;
  if (getenv("HYMLS_MI_HOST_CHUNK_BYTES")) t->chunk = atoll(getenv("HYMLS_MI_HOST_CHUNK_BYTES"));   /* (tests: small rounds) */
  if (t->chunk < 8) t->chunk = 8;
  t->cb.ctx = t; t->cb.rank = t->rank; t->cb.size = t->size;
  t->cb.alltoallv = hymls_mi_mpi__alltoallv;
  t->cb.alloc = hymls_mi_mpi__alloc;
  ierr = hymls_mi_set_comm(h, &t->cb, px, py, pz);
  if (ierr) { free(t); return ierr; }
  *out = t;
  return 0;
}

static inline void hymls_mi_mpi_transport_free(hymls_mi_mpi_transport* t) {
  int i;
  if (!t) return;
  for (i = 0; i < t->n_arenas; i++) hymls_mi_device_free(t->h, t->arenas[i]);
  free(t->arenas); free(t->hs); free(t->hr);
  free(t);
}

/* attach `comm` through the library's built-in RCCL transport: rank 0 draws the RCCL id, MPI_Bcast hands it out, every rank
 * joins (ncclCommInitRank on `device`).  *nccl_comm is released with hymls_mi_rccl_comm_destroy after hymls_mi_destroy.
 * -3 when librccl cannot be loaded, -99 from the test-only host simulator. */
static inline int hymls_mi_set_comm_rccl_mpi(hymls_mi_t* h, MPI_Comm comm, int device, int px, int py, int pz, void** nccl_comm) {
  char id[128];
  int rank = 0, size = 1, ierr = 0, worst = 0;
  if (!h || !nccl_comm) return -2;
  MPI_Comm_rank(comm, &rank);
  MPI_Comm_size(comm, &size);
  memset(id, 0, sizeof id);
  if (rank == 0) ierr = hymls_mi_rccl_unique_id(id);
  MPI_Bcast(&ierr, 1, MPI_INT, 0, comm);
  if (ierr) return ierr;
  MPI_Bcast(id, 128, MPI_BYTE, 0, comm);
  ierr = hymls_mi_rccl_comm_init(id, rank, size, device, nccl_comm);
  MPI_Allreduce(&ierr, &worst, 1, MPI_INT, MPI_MIN, comm);
  if (worst) return ierr ? ierr : worst;
  return hymls_mi_set_comm_rccl(h, *nccl_comm, px, py, pz);
}

#ifdef __cplusplus
}
#endif
#endif /* HYMLS_MI_MPI_H */
