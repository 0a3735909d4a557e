// capi.cpp -- the C ABI of include/hymls_mi.h on top of LevelSolver.
#include "../../include/hymls_mi.h"
#include "precond.hpp"
#include <chrono>
#include <cstring>

namespace hymls {
int64_t generate_laplace3d(int nx, int ny, int nz, int32_t* rowptr, int32_t* col, double* val);
int64_t generate_stokes3d(int nx, int ny, int nz, double a, double b, int32_t* rowptr, int32_t* col, double* val);
int64_t generate_rows(int equations, int nx, int ny, int nz, double a, double b, int64_t nrows, const int32_t* gids,
                      int32_t* rowptr, int32_t* col, double* val, double re, const bool* per);
}

using namespace hymls;

struct hymls_mi {
  Params p;
  int device = 0;
  dev::Context* ctx = nullptr;   // this handle's streams, arenas, timers (bound at every entry below)
  Csr K;
  bool have_matrix = false;
  Comm comm;                 // one rank unless hymls_mi_set_comm was called
  ivec gids;                 // local nodes of K (sharded: rows given + ghost columns)
  int32_t nrows = 0;         // local nodes with a row
  ivec rows_rowptr, rows_colgid;   // sharded: the rows as they were given (pattern check of SetMatrix)
  int64_t ngid() const { return (int64_t)p.nx * p.ny * p.nz * p.dof; }
  dvec tv;
  std::unique_ptr<LevelSolver> top;
  bool initialized = false, computed = false;
  int n_init = 0, n_comp = 0, n_apply = 0, n_init_top = 0;
  double t_init = 0, t_comp = 0, t_apply = 0;
  std::string err;
  double *d_b = nullptr, *d_x = nullptr;
  int64_t buf_n = 0;
  bool profiling = false;
  double prof_sum[8] = {0};
  int prof_cnt[8] = {0};
};

static double now() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

#define API_BEGIN try { dev::bind(h->ctx);
#define API_END(h)                                                   \
  }                                                                  \
  catch (const hymls::Error& e) { if (h) (h)->err = e.what(); return e.code; } \
  catch (const std::exception& e) { if (h) (h)->err = e.what(); return -3; }   \
  return 0;

extern "C" {

void hymls_mi_default_params(hymls_mi_params* p) {
  std::memset(p, 0, sizeof *p);
  p->nx = p->ny = p->nz = -1;
  p->dim = 3; p->equations = 0; p->dof = -1;
  p->sx = 4; p->sy = p->sz = -1; p->cx = p->cy = p->cz = -1;
  p->levels = 1; p->partitioner = 0; p->retain_nodes = -1; p->retain_pressures = -1;
  p->link_velocities = 1; p->link_retained = 1; p->fix_pressure_level = 1; p->nfix = 0;
  for (int d = 0; d < 3; d++) p->retain_xyz[d] = -1;
  for (int l = 0; l < 8; l++) p->retain_at_level[l] = -1;
  for (int l = 0; l < 8; l++) for (int d = 0; d < 3; d++) p->retain_at_level_xyz[l][d] = -1;
  for (int d = 0; d < 3; d++) p->periodic[d] = 0;
}

static Params convert(const hymls_mi_params* q) {
  // defaults as BasePartitioner::SetParameters (reference src/HYMLS_BasePartitioner.cpp:31-252)
  Params p;
  p.dim = q->dim;
  p.nx = q->nx; p.ny = q->ny < 0 ? q->nx : q->ny;
  p.nz = q->nz < 0 ? (q->dim > 2 ? q->nx : 1) : q->nz;
  HYMLS_CHECK(p.nx > 0, -2, "You must presently specify nx, ny (and possibly nz)");
  p.sx = q->sx; p.sy = q->sy < 0 ? p.sx : q->sy; p.sz = p.nz > 1 ? (q->sz < 0 ? p.sx : q->sz) : 1;
  HYMLS_CHECK(p.sx > 1, -2, "Separator Length not set correctly");
  p.cx = q->cx < 0 ? p.sx : q->cx; p.cy = q->cy < 0 ? p.cx : q->cy; p.cz = p.nz > 1 ? (q->cz < 0 ? p.cx : q->cz) : 1;
  HYMLS_CHECK(p.cx > 1, -2, "Coarsening Factor not set correctly");
  p.retain = q->retain_nodes;
  for (int d = 0; d < 3; d++) p.retain_xyz[d] = q->retain_xyz[d];
  for (int l = 0; l < 8; l++) p.retain_at_level[l] = q->retain_at_level[l];
  for (int l = 0; l < 8; l++) for (int d = 0; d < 3; d++) p.retain_at_level_xyz[l][d] = q->retain_at_level_xyz[l][d];
  for (int d = 0; d < 3; d++) p.perio[d] = q->periodic[d] != 0 && d < p.dim;
  p.level = 0;
  p.set_retain();
  p.levels = q->levels;
  p.partitioner = q->partitioner;
  p.link_velocities = q->link_velocities != 0; p.link_retained = q->link_retained != 0;
  p.retain_pressures = q->retain_pressures < 0 ? 1 : q->retain_pressures;
  if (q->equations == 0) {
    p.dof = 1; p.vtype = {VT_LAPLACE};
  } else if (q->equations == 1) {
    p.dof = p.dim + 1;
    p.vtype = p.dim == 2 ? std::vector<int32_t>{VT_U, VT_V, VT_P} : std::vector<int32_t>{VT_U, VT_V, VT_W, VT_P};
    if (q->fix_pressure_level) p.fix_gid = {p.dim};
  } else {
    HYMLS_CHECK(q->dof > 0 && q->dof <= 8, -2, "'Equations' parameter not recognized and no 'Degrees of Freedom' given");
    p.dof = q->dof;
    p.vtype.assign(q->variable_type, q->variable_type + q->dof);
  }
  if (q->nfix > 0) p.fix_gid.assign(q->fix_gid, q->fix_gid + std::min(q->nfix, 4));
  return p;
}

int hymls_mi_create(hymls_mi_t** out, const hymls_mi_params* q, int device) {
  if (!out || !q) return -2;
  hymls_mi* h = new hymls_mi();
  *out = h;
  API_BEGIN
  h->p = convert(q);
  h->device = device;
  h->ctx = dev::create_context(device);
  API_END(h)
}

int hymls_mi_set_matrix_csr(hymls_mi_t* h, int64_t n, const int32_t* rowptr, const int32_t* colind, const double* val) {
  if (!h) return -2;
  API_BEGIN
  const int64_t N = h->ngid();
  HYMLS_CHECK(!h->comm.distributed(), -2, "sharded handle: pass this rank's rows with hymls_mi_set_matrix_rows");
  HYMLS_CHECK(n == N, -2, "matrix size does not match nx*ny*nz*dof");
  const int64_t nnz = rowptr[n];
  const bool same = h->have_matrix && h->K.n == n && (int64_t)h->K.col.size() == nnz &&
                    parallel_equal(h->K.rowptr.data(), rowptr, (size_t)(n + 1) * 4) &&
                    parallel_equal(h->K.col.data(), colind, (size_t)nnz * 4);
  h->K.n = (int32_t)n;
  if (!same) {
    parallel_assign(h->K.rowptr, rowptr, (size_t)n + 1);
    parallel_assign(h->K.col, colind, (size_t)nnz);
    h->gids.resize(n);
    parallel_for(n, [&](int64_t i) { h->gids[i] = (int32_t)i; }, 1 << 16);
  }
  parallel_assign(h->K.val, val, (size_t)nnz);
  h->nrows = (int32_t)n;
  h->have_matrix = true;
  h->computed = false;
  if (same && h->top) h->top->set_values(h->K.val);   // SetMatrix: pattern reused
  else { h->top.reset(); h->initialized = false; }
  API_END(h)
}

int hymls_mi_set_comm(hymls_mi_t* h, const hymls_mi_comm* c, int px, int py, int pz) {
  if (!h || !c) return -2;
  API_BEGIN
  HYMLS_CHECK(c->size >= 1 && c->rank >= 0 && c->rank < c->size && px * py * pz == c->size, -2,
              "rank grid px*py*pz must equal the number of ranks");
  HYMLS_CHECK(c->size == 1 || (c->alltoallv && c->alloc), -2, "a sharded run needs both transport callbacks");
  h->comm.release();   // (a built-in transport attached earlier)
  h->comm.rank = c->rank; h->comm.size = c->size; h->comm.ctx = c->ctx;
  h->comm.px = px; h->comm.py = py; h->comm.pz = pz;
  h->comm.alltoallv = c->alltoallv; h->comm.alloc = c->alloc;
  h->comm.force = c->size == 1 && c->alltoallv && c->alloc && std::getenv("HYMLS_MI_FORCE_SHARDED") != nullptr;
  h->top.reset(); h->initialized = false; h->computed = false; h->have_matrix = false;
  API_END(h)
}

int hymls_mi_rank_grid(int size, int* px, int* py, int* pz) {
  if (size < 1 || !px || !py || !pz) return -2;
  std::vector<int> f;
  for (int w = size, p = 2; w > 1; p++) while (w % p == 0) { f.push_back(p); w /= p; }
  std::sort(f.rbegin(), f.rend());
  int g[3] = {1, 1, 1};
  for (int v : f) {
    int d = 0;
    for (int k = 1; k < 3; k++) if (g[k] < g[d]) d = k;
    g[d] *= v;
  }
  *px = g[0]; *py = g[1]; *pz = g[2];
  return 0;
}

void* hymls_mi_device_alloc(hymls_mi_t* h, int64_t bytes) {
  if (!h || bytes < 0) return nullptr;
  try { dev::bind(h->ctx); return dev::alloc((size_t)bytes); } catch (const std::exception& e) { h->err = e.what(); return nullptr; }
}
void hymls_mi_device_free(hymls_mi_t* h, void* p) {
  if (!h || !p) return;
  try { dev::bind(h->ctx); dev::free(p); } catch (...) {}
}
int hymls_mi_copy_to_host(hymls_mi_t* h, void* host_dst, const void* device_src, int64_t bytes) {
  if (!h || bytes < 0 || (bytes && (!host_dst || !device_src))) return -2;
  API_BEGIN
  dev::d2h(host_dst, device_src, (size_t)bytes);
  API_END(h)
}
int hymls_mi_copy_to_device(hymls_mi_t* h, void* device_dst, const void* host_src, int64_t bytes) {
  if (!h || bytes < 0 || (bytes && (!device_dst || !host_src))) return -2;
  API_BEGIN
  dev::h2d(device_dst, host_src, (size_t)bytes);
  API_END(h)
}

int hymls_mi_rccl_unique_id(char* id128) {
  if (!id128) return -2;
  try { rccl_unique_id(id128); } catch (const hymls::Error& e) { return e.code; } catch (...) { return -3; }
  return 0;
}

int hymls_mi_rccl_comm_init(const char* id128, int rank, int size, int device, void** nccl_comm) {
  if (!id128 || !nccl_comm || size < 1 || rank < 0 || rank >= size) return -2;
  try { *nccl_comm = rccl_init(id128, rank, size, device); } catch (const hymls::Error& e) { return e.code; } catch (...) { return -3; }
  return 0;
}

void hymls_mi_rccl_comm_destroy(void* nccl_comm) { try { rccl_destroy(nccl_comm); } catch (...) {} }

int hymls_mi_set_comm_rccl(hymls_mi_t* h, void* nccl_comm, int px, int py, int pz) {
  if (!h || !nccl_comm) return -2;
  API_BEGIN
  rccl_attach(h->comm, nccl_comm, /*owns=*/false);
  HYMLS_CHECK(px * py * pz == h->comm.size, -2, "rank grid px*py*pz must equal the number of ranks of the communicator");
  h->comm.px = px; h->comm.py = py; h->comm.pz = pz;
  h->comm.force = h->comm.size == 1 && std::getenv("HYMLS_MI_FORCE_SHARDED") != nullptr;
  h->top.reset(); h->initialized = false; h->computed = false; h->have_matrix = false;
  API_END(h)
}

int hymls_mi_comm_selftest(hymls_mi_t* h) {
  if (!h) return -2;
  API_BEGIN
  const Comm& c = h->comm;
  HYMLS_CHECK(c.alltoallv && c.alloc, -2, "no transport set (hymls_mi_set_comm / hymls_mi_set_comm_rccl)");
  // uneven device all-to-all of stamped doubles: rank r sends (r + q) % 3 + 1 values r * 1000 + q + 0.25 k to rank q
  std::vector<int64_t> sc(c.size), rc(c.size);
  int64_t ns = 0, nr = 0;
  for (int q = 0; q < c.size; q++) { sc[q] = (c.rank + q) % 3 + 1; rc[q] = (q + c.rank) % 3 + 1; ns += sc[q]; nr += rc[q]; }
  dvec hs((size_t)ns), hr((size_t)nr, -1.0);
  { int64_t o = 0; for (int q = 0; q < c.size; q++) for (int64_t k = 0; k < sc[q]; k++) hs[o++] = c.rank * 1000.0 + q + 0.25 * k; }
  double* sb = c.send_arena(ns);
  double* rb = c.recv_arena(nr);
  dev::h2d(sb, hs.data(), (size_t)ns * sizeof(double));
  const int ierr = c.alltoallv(c.ctx, sb, sc.data(), rb, rc.data(), (int32_t)sizeof(double), 1);
  HYMLS_CHECK(ierr == 0, -3, std::string("transport self-test: device all-to-all failed ") + rccl_last_error(c));
  dev::d2h(hr.data(), rb, (size_t)nr * sizeof(double));
  { int64_t o = 0; for (int q = 0; q < c.size; q++) for (int64_t k = 0; k < rc[q]; k++) HYMLS_CHECK(hr[o++] == q * 1000.0 + c.rank + 0.25 * k, -3, "transport self-test: wrong data"); }
  // host-side exchange (setup path): counts round trip
  std::vector<int64_t> back = c.exchange_counts(sc);
  for (int q = 0; q < c.size; q++) HYMLS_CHECK(back[q] == rc[q], -3, "transport self-test: wrong host data");
  API_END(h)
}

int hymls_mi_invert_blocks(hymls_mi_t* h, int32_t nb, int32_t nblk, double* blocks) {
  if (!h || (!blocks && nb > 0 && nblk > 0)) return -2;
  API_BEGIN
  HYMLS_CHECK(nb >= 0 && nblk >= 0, -2, "negative block order or count");
  const size_t bytes = (size_t)nb * nb * nblk * sizeof(double);
  if (bytes) {
    double* d = (double*)dev::alloc(bytes);
    int32_t* d_flag = (int32_t*)dev::alloc(sizeof(int32_t));
    int32_t flag = 0;
    try {
      dev::h2d(d, blocks, bytes);
      dev::zero(d_flag, sizeof(int32_t));
      dev::dense_invert(nb, nblk, d, d_flag);
      dev::d2h(&flag, d_flag, sizeof flag);
      dev::d2h(blocks, d, bytes);
    } catch (...) { dev::free(d); dev::free(d_flag); throw; }
    dev::free(d); dev::free(d_flag);
    HYMLS_CHECK(flag == 0, -4, "singular block");
  }
  API_END(h)
}

int hymls_mi_required_rows(hymls_mi_t* h, int64_t* n, int32_t* gids) {
  if (!h || !n) return -2;
  API_BEGIN
  if (!h->top) { h->top.reset(new LevelSolver(h->p, 0, h->ngid(), &h->comm)); h->initialized = false; h->n_init_top = 0; }
  h->top->partition(nullptr);
  ivec r = h->top->required_gids();
  *n = (int64_t)r.size();
  if (gids) std::copy(r.begin(), r.end(), gids);
  API_END(h)
}

int hymls_mi_set_matrix_rows(hymls_mi_t* h, int64_t nrows, const int32_t* gids, const int32_t* rowptr,
                             const int32_t* colgid, const double* val) {
  if (!h || !gids || !rowptr) return -2;
  API_BEGIN
  const int64_t nnz = rowptr[nrows];
  const bool same = h->have_matrix && h->nrows == nrows && (int64_t)h->rows_colgid.size() == nnz &&
                    parallel_equal(h->gids.data(), gids, (size_t)nrows * 4) &&
                    parallel_equal(h->rows_rowptr.data(), rowptr, (size_t)(nrows + 1) * 4) &&
                    parallel_equal(h->rows_colgid.data(), colgid, (size_t)nnz * 4);
  h->computed = false;
  if (same && h->top && h->initialized) {
    parallel_assign(h->K.val, val, (size_t)nnz);
    h->top->set_values(h->K.val);
  } else {
    make_local_csr(nrows, gids, rowptr, colgid, val, h->ngid(), h->K, h->gids);
    h->nrows = (int32_t)nrows;
    h->rows_rowptr.assign(rowptr, rowptr + nrows + 1);
    h->rows_colgid.assign(colgid, colgid + nnz);
    h->have_matrix = true;
    h->initialized = false;
    h->tv.clear();
  }
  API_END(h)
}

int hymls_mi_owned_rows(const hymls_mi_t* hc, int64_t* n, int32_t* gids) {
  hymls_mi_t* h = const_cast<hymls_mi_t*>(hc);
  if (!h || !n) return -2;
  API_BEGIN
  HYMLS_CHECK(h->initialized && h->top, -1, "not initialized");
  const ivec& o = h->top->owned_gids();
  *n = (int64_t)o.size();
  if (gids) std::copy(o.begin(), o.end(), gids);
  API_END(h)
}

int hymls_mi_set_testvector(hymls_mi_t* h, const double* v) {
  if (!h) return -2;
  API_BEGIN
  HYMLS_CHECK(h->have_matrix, -1, "set the matrix first");
  h->tv.assign(v, v + h->nrows);
  if (!h->comm.distributed()) h->top.reset();
  h->initialized = false; h->computed = false;
  API_END(h)
}

int hymls_mi_initialize(hymls_mi_t* h) {
  if (!h) return -2;
  API_BEGIN
  HYMLS_CHECK(h->have_matrix, -1, "no matrix set");
  const double t0 = now();
  if (h->tv.empty()) h->tv.assign(h->nrows, 1.0);
  // sharded: keep the partition hymls_mi_required_rows already made, unless it was used up by an earlier Initialize
  if (!h->top || h->n_init_top) h->top.reset(new LevelSolver(h->p, 0, h->ngid(), &h->comm));
  h->n_init_top = 1;
  h->top->set_rows(h->K, h->gids, h->tv, h->nrows);
  h->top->initialize();
  h->top->profiling = h->profiling;
  h->initialized = true; h->computed = false;
  h->n_init++;
  h->t_init += now() - t0;
  API_END(h)
}

int hymls_mi_compute(hymls_mi_t* h) {
  if (!h) return -2;
  if (!h->initialized) {  // Preconditioner.cpp:403-409: "I'll do it for you"
    int ierr = hymls_mi_initialize(h);
    if (ierr) return ierr;
  }
  API_BEGIN
  const double t0 = now();
  h->top->compute();
  dev::sync();
  h->computed = true;
  h->n_comp++;
  h->t_comp += now() - t0;
  API_END(h)
}

int hymls_mi_apply_inverse(hymls_mi_t* h, const double* B, int64_t ldb, double* X, int64_t ldx, int nvec, int on_device) {
  if (!h) return -2;
  API_BEGIN
  HYMLS_CHECK(h->computed, -1, "The preconditioner has not yet been computed.");
  const double t0 = now();
  const int64_t n = h->top->num_owned();
  // several right-hand sides go through the multi-vector path in groups (the factors are streamed once per group)
  const int group = std::max(1, std::min(nvec, 8));
  if (!on_device && h->buf_n < n * group) {
    dev::free(h->d_b); dev::free(h->d_x);
    h->d_b = (double*)dev::alloc(n * group * sizeof(double));
    h->d_x = (double*)dev::alloc(n * group * sizeof(double));
    h->buf_n = n * group;
  }
  const int bm = h->top->border_size();
  dvec tz(std::max(bm, 1), 0.0), s0(std::max(bm, 1), 0.0);
  for (int k = 0; k < nvec; k += group) {
    const int g = std::min(group, nvec - k);
    if (on_device) {
      if (bm) for (int v = 0; v < g; v++) h->top->apply_inverse_bordered(B + (k + v) * ldb, tz.data(), X + (k + v) * ldx, s0.data());
      else h->top->apply_inverse_mv(B + k * ldb, ldb, X + k * ldx, ldx, g);
    } else {
      for (int v = 0; v < g; v++) dev::h2d(h->d_b + v * n, B + (k + v) * ldb, n * sizeof(double));
      if (bm) for (int v = 0; v < g; v++) h->top->apply_inverse_bordered(h->d_b + v * n, tz.data(), h->d_x + v * n, s0.data());
      else h->top->apply_inverse_mv(h->d_b, n, h->d_x, n, g);
      for (int v = 0; v < g; v++) dev::d2h(X + (k + v) * ldx, h->d_x + v * n, n * sizeof(double));
    }
  }
  h->n_apply++;
  if (!on_device) h->t_apply += now() - t0;
  API_END(h)
}

int hymls_mi_set_border(hymls_mi_t* h, int m, const double* V, int64_t ldv, const double* W, int64_t ldw, const double* C) {
  if (!h) return -2;
  if (!h->initialized) {
    int ierr = hymls_mi_initialize(h);
    if (ierr) return ierr;
  }
  API_BEGIN
  h->computed = false;
  if (m <= 0 || !V) { h->top->set_border(0, nullptr, nullptr, nullptr); return 0; }
  const int64_t n = h->top->num_owned();
  HYMLS_CHECK(ldv >= n && (!W || ldw >= n), -2, "SetBorder: leading dimension smaller than the number of rows");
  dvec v((size_t)n * m), w((size_t)n * m);
  for (int j = 0; j < m; j++) {
    std::copy(V + j * ldv, V + j * ldv + n, v.begin() + (size_t)j * n);
    const double* wj = W ? W + j * ldw : V + j * ldv;
    std::copy(wj, wj + n, w.begin() + (size_t)j * n);
  }
  double* dv = dev::upload(v);
  double* dw = dev::upload(w);
  try { h->top->set_border(m, dv, dw, C); } catch (...) { dev::free(dv); dev::free(dw); throw; }
  dev::free(dv); dev::free(dw);
  API_END(h)
}

int hymls_mi_apply_inverse_bordered(hymls_mi_t* h, const double* B, const double* T, double* X, double* S, int on_device) {
  if (!h) return -2;
  API_BEGIN
  HYMLS_CHECK(h->computed, -1, "The preconditioner has not yet been computed.");
  const int64_t n = h->top->num_owned();
  const int m = h->top->border_size();
  dvec t0(std::max(m, 1), 0.0), s0(std::max(m, 1), 0.0);
  const double* Tp = T ? T : t0.data();
  double* Sp = S ? S : s0.data();
  if (on_device) {
    h->top->apply_inverse_bordered(B, Tp, X, Sp);
  } else {
    if (h->buf_n < n) {
      dev::free(h->d_b); dev::free(h->d_x);
      h->d_b = (double*)dev::alloc(n * sizeof(double)); h->d_x = (double*)dev::alloc(n * sizeof(double)); h->buf_n = n;
    }
    dev::h2d(h->d_b, B, n * sizeof(double));
    h->top->apply_inverse_bordered(h->d_b, Tp, h->d_x, Sp);
    dev::d2h(X, h->d_x, n * sizeof(double));
  }
  h->n_apply++;
  API_END(h)
}

int hymls_mi_apply(hymls_mi_t*, const double*, double*) { return -1; }

int hymls_mi_matvec(hymls_mi_t* h, const double* X, double* Y, int on_device) {
  if (!h) return -2;
  API_BEGIN
  HYMLS_CHECK(h->computed, -1, "matvec needs a computed preconditioner (device copy of K)");
  const int64_t n = h->top->num_owned();
  if (on_device) { h->top->matvec(X, Y); }
  else {
    if (h->buf_n < n) {
      dev::free(h->d_b); dev::free(h->d_x);
      h->d_b = (double*)dev::alloc(n * sizeof(double)); h->d_x = (double*)dev::alloc(n * sizeof(double)); h->buf_n = n;
    }
    dev::h2d(h->d_b, X, n * sizeof(double));
    h->top->matvec(h->d_b, h->d_x);
    dev::d2h(Y, h->d_x, n * sizeof(double));
  }
  API_END(h)
}

int hymls_mi_is_initialized(const hymls_mi_t* h) { return h && h->initialized; }
int hymls_mi_is_computed(const hymls_mi_t* h) { return h && h->computed; }
int hymls_mi_num_initialize(const hymls_mi_t* h) { return h ? h->n_init : 0; }
int hymls_mi_num_compute(const hymls_mi_t* h) { return h ? h->n_comp : 0; }
int hymls_mi_num_apply_inverse(const hymls_mi_t* h) { return h ? h->n_apply : 0; }
double hymls_mi_initialize_time(const hymls_mi_t* h) { return h ? h->t_init : 0; }
double hymls_mi_compute_time(const hymls_mi_t* h) { return h ? h->t_comp : 0; }
double hymls_mi_apply_inverse_time(const hymls_mi_t* h) { return h ? h->t_apply : 0; }

static const Operator* level_op(const hymls_mi_t* h, int level, const LevelSolver** ls) {
  const LevelSolver* L = h->top.get();
  *ls = L;
  const Operator* op = L;
  for (int l = 0; l < level && op; l++) {
    if (!*ls) return nullptr;
    op = (*ls)->next();
    *ls = (*ls)->next_level();
  }
  return op;
}

int hymls_mi_num_levels(const hymls_mi_t* h) {
  if (!h || !h->top) return 0;
  int n = 1;
  const LevelSolver* L = h->top.get();
  while (L && L->next()) { n++; L = L->next_level(); }
  return n;
}
int64_t hymls_mi_level_size(const hymls_mi_t* h, int level) {
  if (!h || !h->top) return -1;
  const LevelSolver* ls; const Operator* op = level_op(h, level, &ls);
  return op ? op->size() : -1;
}
int64_t hymls_mi_level_schur_size(const hymls_mi_t* h, int level) {
  if (!h || !h->top) return -1;
  const LevelSolver* ls; const Operator* op = level_op(h, level, &ls);
  return (op && ls) ? ls->schur_size() : 0;
}
int64_t hymls_mi_level_num_subdomains(const hymls_mi_t* h, int level) {
  if (!h || !h->top) return -1;
  const LevelSolver* ls; const Operator* op = level_op(h, level, &ls);
  return (op && ls) ? (int64_t)ls->hiermap().sd.size() : 0;
}

double hymls_mi_apply_bytes(const hymls_mi_t* h, int which) {
  if (!h || !h->top) return 0;
  ApplyStats st;
  h->top->add_stats(st, false);
  switch (which) {
    case 1: return st.bytes_factor;
    case 2: return st.bytes_spmv;
    case 3: return st.bytes_sep;
    case 4: return st.bytes_coarse;
    case 5: return st.bytes_vec;
    case 6: return st.bytes_factor_sparse;
    case 7: return st.bytes_coarse_sparse;
    case 8: return std::min(st.bytes_factor, st.bytes_factor_sparse) + st.bytes_spmv + st.bytes_sep +
                   std::min(st.bytes_coarse, st.bytes_coarse_sparse) + st.bytes_vec;
    default: return st.bytes_factor + st.bytes_spmv + st.bytes_sep + st.bytes_coarse + st.bytes_vec;
  }
}
double hymls_mi_setup_flops(const hymls_mi_t* h, int which) {
  if (!h || !h->top) return 0;
  ApplyStats st;
  h->top->add_stats(st, false);
  switch (which) {
    case 1: return st.flops_factor;
    case 2: return st.flops_blocks;
    case 3: return st.flops_transform;
    default: return st.flops_factor + st.flops_blocks + st.flops_transform;
  }
}
double hymls_mi_last_apply_seconds(const hymls_mi_t* hc, int which) {
  hymls_mi_t* h = const_cast<hymls_mi_t*>(hc);
  if (!h || !h->top || which < 0 || which > 4) return 0;
  // average seconds per ApplyInverse of phase `which` since profiling was switched on
  try {
    double sum[8] = {0}; int cnt[8] = {0};
    dev::bind(h->ctx);
    dev::profile_collect(sum, cnt);
    for (int i = 0; i < 8; i++) { h->prof_sum[i] += sum[i]; h->prof_cnt[i] += cnt[i]; }
  } catch (...) { return 0; }
  return h->prof_cnt[0] ? h->prof_sum[which] / h->prof_cnt[0] : 0.0;
}
int hymls_mi_set_profiling(hymls_mi_t* h, int on) {
  if (!h) return -2;
  h->profiling = on != 0;
  for (int i = 0; i < 8; i++) { h->prof_sum[i] = 0; h->prof_cnt[i] = 0; }
  try { double s[8] = {0}; int c[8] = {0}; dev::bind(h->ctx); dev::profile_collect(s, c); } catch (...) {}
  if (h->top) h->top->profiling = h->profiling;
  return 0;
}
void* hymls_mi_stream(const hymls_mi_t* h) {
  if (!h || !h->ctx) return nullptr;
  try { dev::bind(h->ctx); return dev::stream(); } catch (...) { return nullptr; }
}

int hymls_mi_get_interior(const hymls_mi_t* h, int level, int sd, int32_t* n, int32_t* nodes) {
  if (!h || !h->top) return -1;
  const LevelSolver* ls; const Operator* op = level_op(h, level, &ls);
  if (!op || !ls || sd < 0 || sd >= (int)ls->hiermap().sd.size()) return -2;
  const Subdomain& S = ls->hiermap().sd[sd];
  *n = (int32_t)S.interior.size();
  if (nodes) std::copy(S.interior.begin(), S.interior.end(), nodes);
  return 0;
}
int hymls_mi_get_separator_groups(const hymls_mi_t* h, int level, int sd, int32_t* ng, int32_t* gptr, int32_t* gtype,
                                  int32_t* owned, int32_t* nodes) {
  if (!h || !h->top) return -1;
  const LevelSolver* ls; const Operator* op = level_op(h, level, &ls);
  if (!op || !ls || sd < 0 || sd >= (int)ls->hiermap().sd.size()) return -2;
  const Subdomain& S = ls->hiermap().sd[sd];
  *ng = (int32_t)S.groups.size();
  if (gptr) {
    int32_t off = 0;
    for (size_t g = 0; g < S.groups.size(); g++) {
      gptr[g] = off;
      if (gtype) gtype[g] = S.groups[g].type;
      if (owned) owned[g] = std::find(S.owned.begin(), S.owned.end(), (int32_t)g) != S.owned.end();
      if (nodes) std::copy(S.groups[g].nodes.begin(), S.groups[g].nodes.end(), nodes + off);
      off += (int32_t)S.groups[g].nodes.size();
    }
    gptr[S.groups.size()] = off;
  }
  return 0;
}

int hymls_mi_generate_matrix(int equations, int nx, int ny, int nz, double a, double b, int64_t* nrows, int64_t* nnz,
                             int32_t* rowptr, int32_t* colind, double* val) {
  if (!nrows || !nnz) return -2;
  if (equations == 0) {
    *nrows = (int64_t)nx * ny * nz;
    *nnz = generate_laplace3d(nx, ny, nz, rowptr, colind, val);
  } else if (equations == 1) {
    *nrows = (int64_t)nx * ny * nz * 4;
    *nnz = generate_stokes3d(nx, ny, nz, a, b, rowptr, colind, val);
  } else {
    return -99;
  }
  return 0;
}

int hymls_mi_generate_rows(int equations, int nx, int ny, int nz, double a, double b, int64_t nrows, const int32_t* gids,
                           int64_t* nnz, int32_t* rowptr, int32_t* colgid, double* val) {
  if (!nnz || !gids || (equations != 0 && equations != 1)) return -2;
  try { *nnz = generate_rows(equations, nx, ny, nz, a, b, nrows, gids, rowptr, colgid, val, 0.0, nullptr); }
  catch (...) { return -2; }
  return 0;
}

int hymls_mi_generate_problem_periodic(int problem, int nx, int ny, int nz, double a, double b, double re, int periodicity,
                                       int64_t nrows, const int32_t* gids, int64_t* nnz, int32_t* rowptr, int32_t* colgid, double* val) {
  if (!nnz || problem < 0 || problem > 3 || nx <= 0 || ny <= 0 || nz <= 0 || periodicity < 0 || periodicity > 7) return -2;
  const int64_t N = (int64_t)nx * ny * nz * (problem == 0 ? 1 : 4);
  if (!gids && nrows != N) return -2;
  if (gids) for (int64_t t = 0; t < nrows; t++) if (gids[t] < 0 || gids[t] >= N) return -2;
  const bool per[3] = {(periodicity & 4) != 0, (periodicity & 2) != 0, (periodicity & 1) != 0};   // GaleriExt::PERIO_Flag: X 4, Y 2, Z 1
  try { *nnz = generate_rows(problem, nx, ny, nz, a, b, nrows, gids, rowptr, colgid, val, re, per); }
  catch (const hymls::Error& e) { return e.code; }
  catch (...) { return -2; }
  return 0;
}

int hymls_mi_generate_problem(int problem, int nx, int ny, int nz, double a, double b, double re, int64_t nrows,
                              const int32_t* gids, int64_t* nnz, int32_t* rowptr, int32_t* colgid, double* val) {
  return hymls_mi_generate_problem_periodic(problem, nx, ny, nz, a, b, re, 0, nrows, gids, nnz, rowptr, colgid, val);
}

int hymls_mi_generate_testvector(int64_t n, const int32_t* rowptr, const int32_t* colind, const double* val, double* tv) {
  // create_testvector (reference src/HYMLS_MainUtils.cpp:208-258), Laplace / Stokes-C branch
  for (int64_t i = 0; i < n; i++) {
    bool is_diag = true;
    for (int32_t e = rowptr[i]; e < rowptr[i + 1]; e++)
      if (val[e] != 0.0 && colind[e] != i) { is_diag = false; break; }
    tv[i] = is_diag ? 0.0 : 1.0;
  }
  return 0;
}

int hymls_mi_drop_by_value(int64_t n, const int32_t* rowptr, const int32_t* col, const double* val, double tol, int kind,
                           int64_t* nnz_out, int32_t* rowptr_out, int32_t* col_out, double* val_out) {
  if (n < 0 || !rowptr || !nnz_out || kind < 0 || kind > 2 || (rowptr[n] > 0 && (!col || !val))) return -2;
  try {
    Csr A;
    A.n = (int32_t)n;
    A.rowptr.assign(rowptr, rowptr + n + 1);
    A.col.assign(col, col + rowptr[n]);
    A.val.assign(val, val + rowptr[n]);
    const Csr R = drop_by_value(A, tol, kind);
    *nnz_out = R.nnz();
    if (rowptr_out) {
      std::copy(R.rowptr.begin(), R.rowptr.end(), rowptr_out);
      if (col_out) std::copy(R.col.begin(), R.col.end(), col_out);
      if (val_out) std::copy(R.val.begin(), R.val.end(), val_out);
    }
  } catch (...) { return -3; }
  return 0;
}

const char* hymls_mi_last_error(const hymls_mi_t* h) { return h ? h->err.c_str() : "null handle"; }

void hymls_mi_destroy(hymls_mi_t* h) {
  if (!h) return;
  try {
    dev::bind(h->ctx);
    h->top.reset();
    dev::free(h->d_b); dev::free(h->d_x);
    h->comm.release();
    dev::destroy_context(h->ctx);
  } catch (...) {}
  delete h;
}

}  // extern "C"
