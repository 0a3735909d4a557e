// device_sim.cpp -- TEST-ONLY host simulator of the device boundary (hymls::dev).
//
// Implements every launcher of hymls_amd/csrc/device.hpp with plain sequential loops on
// host memory so that the host-side index plans (partition, ordering, assembly tree,
// gather/pull lists) can be validated end-to-end on a machine without a GPU.  It is linked
// ONLY into tests/hostsim/libhymls_mi_hostsim.so; the product library libhymls_mi.so
// contains the HIP implementation (device_hip.hip) and no CPU path.
#include "device.hpp"
#include "comm.hpp"
#include <algorithm>
#include <cstring>
#include <chrono>

namespace hymls {
// the built-in RCCL transport does not exist in the simulator
void rccl_unique_id(char*) { throw Error(-99, "the host simulator has no RCCL transport"); }
void* rccl_init(const char*, int, int, int) { throw Error(-99, "the host simulator has no RCCL transport"); }
void rccl_destroy(void*) {}
void rccl_attach(Comm&, void*, bool) { throw Error(-99, "the host simulator has no RCCL transport"); }
const char* rccl_last_error(const Comm&) { return ""; }
namespace dev {

struct Context { int device; };
Context* create_context(int device) { return new Context{device}; }
void bind(Context*) {}
Context* current() { return nullptr; }
void destroy_context(Context* c) { delete c; }
void* stream() { return nullptr; }
const double* zeros16() { static const double z[16] = {0}; return z; }
void* alloc(size_t bytes) { return std::calloc(std::max<size_t>(bytes, 8), 1); }
void free(void* p) { std::free(p); }
void h2d(void* d, const void* s, size_t n) { std::memcpy(d, s, n); }
void d2h(void* d, const void* s, size_t n) { std::memcpy(d, s, n); }
void d2d(void* d, const void* s, size_t n) { std::memmove(d, s, n); }
void zero(void* d, size_t n) { std::memset(d, 0, n); }
void sync() {}
void fork_streams() {}
void use_stream(int) {}
void join_streams() {}
static void* g_arena = nullptr;
static size_t g_arena_cap = 0;
void* shared_scratch(size_t bytes) {
  if (bytes > g_arena_cap) { std::free(g_arena); g_arena = std::malloc(bytes); g_arena_cap = bytes; }
  return g_arena;
}
size_t mem_free() { return (size_t)1 << 40; }
// ranges: no profiler on the host; the log file is what the tests read
static std::FILE* range_log() {
  static std::FILE* f = std::getenv("HYMLS_MI_RANGE_LOG") ? std::fopen(std::getenv("HYMLS_MI_RANGE_LOG"), "a") : nullptr;
  return f;
}
void range_push(const char* label) { if (std::FILE* f = range_log()) { std::fprintf(f, "push %s\n", label); std::fflush(f); } }
void range_pop() { if (std::FILE* f = range_log()) { std::fprintf(f, "pop\n"); std::fflush(f); } }
static std::chrono::steady_clock::time_point t0[16];
void timer_start(int id) { t0[id] = std::chrono::steady_clock::now(); }
static std::vector<std::pair<int, std::chrono::steady_clock::time_point>> g_log;
void mark(int phase, bool begin) { g_log.emplace_back(begin ? phase : phase + 8, std::chrono::steady_clock::now()); }
void profile_collect(double* sum, int* cnt) {
  std::chrono::steady_clock::time_point open[8];
  for (auto& e : g_log) {
    if (e.first < 8) open[e.first] = e.second;
    else { sum[e.first - 8] += std::chrono::duration<double>(e.second - open[e.first - 8]).count(); cnt[e.first - 8]++; }
  }
  g_log.clear();
}
double timer_stop(int id) { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0[id]).count(); }

void gather(int64_t n, const int32_t* idx, const double* src, double* dst) { for (int64_t i = 0; i < n; i++) dst[i] = src[idx[i]]; }
void scatter(int64_t n, const int32_t* idx, const double* src, double* dst) { for (int64_t i = 0; i < n; i++) dst[idx[i]] = src[i]; }
void scatter_add(int64_t n, const int32_t* idx, const double* src, double* dst) { for (int64_t i = 0; i < n; i++) dst[idx[i]] += src[i]; }
void axpby(int64_t n, double a, const double* x, double b, double* y) { for (int64_t i = 0; i < n; i++) y[i] = a * x[i] + b * y[i]; }
void scale_copy(int64_t n, double a, const double* x, double* y) { for (int64_t i = 0; i < n; i++) y[i] = a * x[i]; }
void spmv(int32_t nrows, const int32_t* rp, const int32_t* col, const double* val, const double* x, double* y,
          double alpha, double beta, int64_t) {
  for (int i = 0; i < nrows; i++) {
    double s = 0;
    for (int e = rp[i]; e < rp[i + 1]; e++) s += val[e] * x[col[e]];
    y[i] = alpha * s + (beta == 0.0 ? 0.0 : beta * y[i]);
  }
}
void pull_sum(int64_t n, const int64_t* ptr, const int64_t* idx, const double* in, double* out) {
  for (int64_t e = 0; e < n; e++) { double s = 0; for (int64_t t = ptr[e]; t < ptr[e + 1]; t++) s += in[idx[t]]; out[e] = s; }
}
static inline int32_t offdiag_target(int32_t c, const int32_t* ta, const int32_t* tb, const int32_t* excl) {
  return ta[c] >= 0 ? ta[c] : ((tb && excl[c] < 0) ? tb[c] : -1);
}
void offdiag_count(int64_t nrows, const int32_t* rows, const int32_t* krow, const int32_t* kcol, const int32_t* ta, const int32_t* tb,
                   const int32_t* excl, int32_t* count) {
  for (int64_t t = 0; t < nrows; t++) {
    int32_t o = 0;
    for (int32_t e = krow[rows[t]]; e < krow[rows[t] + 1]; e++) o += offdiag_target(kcol[e], ta, tb, excl) >= 0;
    count[t + 1] = o;
  }
}
void offdiag_fill(int64_t nrows, const int32_t* rows, const int32_t* krow, const int32_t* kcol, const int32_t* ta, const int32_t* tb,
                  const int32_t* excl, const int32_t* rowptr, int32_t* col, int32_t* src) {
  for (int64_t t = 0; t < nrows; t++) {
    int32_t o = rowptr[t];
    for (int32_t e = krow[rows[t]]; e < krow[rows[t] + 1]; e++) {
      const int32_t tg = offdiag_target(kcol[e], ta, tb, excl);
      if (tg >= 0) { col[o] = tg; src[o] = e; o++; }
    }
  }
}
void member_sources(int32_t nb, int32_t next, int32_t nent, const int32_t* ext, const int32_t* ent_row, const int32_t* ent_col,
                    const int32_t* krow, const int32_t* kcol, int32_t* src, int32_t* flag) {
  for (int b = 0; b < nb; b++) {
    const int32_t* nodes = ext + (int64_t)b * next;
    for (int q = 0; q < nent; q++) {
      const int32_t r = nodes[ent_row[q]], c = nodes[ent_col[q]];
      const int32_t* f = std::lower_bound(kcol + krow[r], kcol + krow[r + 1], c);
      const bool found = f < kcol + krow[r + 1] && *f == c;
      if (!found) *flag |= 1;
      src[(int64_t)b * nent + q] = found ? (int32_t)(f - kcol) : 0;
    }
  }
}
void build_pull_tables(int64_t nrows, const int64_t* rcount, const int32_t* rowptr, const uint64_t* keys, int64_t* ptr, int64_t* idx) {
  ptr[0] = 0;
  for (int64_t r = 0; r < nrows; r++) {
    int64_t e = (int64_t)rowptr[r] - 1;
    for (int64_t k = rcount[r]; k < rcount[r + 1]; k++) {
      if (k == rcount[r] || (keys[k] >> 33) != (keys[k - 1] >> 33)) e++;
      idx[k] = (int64_t)(keys[k] & (((uint64_t)1 << 33) - 1));
      ptr[e + 1] = k + 1;
    }
  }
}
void pull_sum_blocks(int64_t blen, int32_t nblk, const int64_t* ptr, const int64_t* base, const double* in, double* out) {
  for (int B = 0; B < nblk; B++)
    for (int64_t k = 0; k < blen; k++) { double s = 0; for (int64_t t = ptr[B]; t < ptr[B + 1]; t++) s += in[base[t] + k]; out[(int64_t)B * blen + k] = s; }
}

void sblock_init(const PlanD& P, const BatchD& B, int32_t b0, int32_t nbc, const double* kval) {
  const int64_t n2 = (int64_t)P.nS * P.nS;
  for (int s = 0; s < nbc; s++) {
    double* S = B.sblock + (int64_t)s * n2;
    std::memset(S, 0, n2 * sizeof(double));
    const int32_t* src = B.src + (int64_t)(b0 + s) * P.nent;
    for (int e = P.s_ent_begin; e < P.s_ent_end; e++) S[P.ent_pos[e]] += P.ent_w[e] * kval[src[P.ent_id[e]]];
  }
}

static void sim_factor_front(const PlanD& P, const BatchD& B, const FrontD& F, int slot, int b, const double* kval) {
    {
      const int w = F.w, ri = F.ri, rs = F.rs, m = w + ri + rs;
      double* sc = B.scratch + (int64_t)slot * P.scratch_size;
      double* A = sc + F.f_off;
      std::memset(A, 0, (size_t)m * m * sizeof(double));
      const int32_t* src = B.src + (int64_t)b * P.nent;
      for (int e = F.ent_begin; e < F.ent_end; e++) A[P.ent_pos[e]] += P.ent_w[e] * kval[src[P.ent_id[e]]];
      for (int ce = F.child_begin; ce < F.child_end; ce++) {
        const FrontD& C = P.fronts[P.children[ce]];
        const int mc = C.w + C.ri + C.rs, rc = C.ri + C.rs;
        const double* Ac = sc + C.f_off;
        const int32_t* rel = P.rel + C.rel_off;
        for (int bb = 0; bb < rc; bb++)
          for (int a = 0; a < rc; a++) A[rel[a] + (int64_t)m * rel[bb]] += Ac[(C.w + a) + (int64_t)mc * (C.w + bb)];
      }
      for (int k = 0; k < w; k++) {
        const double piv = A[k + (int64_t)m * k];
        if (piv == 0.0 || !std::isfinite(piv)) *B.flag = 1;
        const double ip = 1.0 / piv;
        for (int i = k + 1; i < m; i++) A[i + (int64_t)m * k] *= ip;
        for (int j = k + 1; j < m; j++) {
          const double u = A[k + (int64_t)m * j];
          if (u != 0.0) for (int i = k + 1; i < m; i++) A[i + (int64_t)m * j] -= A[i + (int64_t)m * k] * u;
        }
      }
      double* fac = B.factor + (int64_t)b * P.factor_size;
      double* Lp = fac + F.lp_off;
      double* Q = fac + F.q_off;
      const int ld = w + ri;
      auto Lk = [&](int i, int j) { return A[i + (int64_t)m * j]; };
      for (int t = 0; t < w; t++) {  // Linv column t (strictly lower part stored)
        for (int i = t + 1; i < w; i++) {
          double s = Lk(i, t);  // j = t term: L[i,t] * z_t (z_t = 1)
          for (int j = t + 1; j < i; j++) s += Lk(i, j) * Lp[j + (int64_t)ld * t];
          Lp[i + (int64_t)ld * t] = -s;
        }
      }
      for (int t = 0; t < w; t++) {  // Uinv column t (upper incl. diagonal)
        Lp[t + (int64_t)ld * t] = 1.0 / Lk(t, t);
        for (int i = t - 1; i >= 0; i--) {
          double s = 0;
          for (int j = i + 1; j <= t; j++) s += Lk(i, j) * Lp[j + (int64_t)ld * t];
          Lp[i + (int64_t)ld * t] = -s / Lk(i, i);
        }
      }
      for (int i = 0; i < ri; i++)  // PL row i = L21[i,:] L11^{-1}
        for (int k = w - 1; k >= 0; k--) {
          double s = Lk(w + i, k);
          for (int j = k + 1; j < w; j++) s -= Lp[(w + i) + (int64_t)ld * j] * Lk(j, k);
          Lp[(w + i) + (int64_t)ld * k] = s;
        }
      for (int j = 0; j < ri; j++)  // QU column j = U11^{-1} U12[:,j]
        for (int i = w - 1; i >= 0; i--) {
          double s = Lk(i, w + j);
          for (int k = i + 1; k < w; k++) s -= Lk(i, k) * Q[k + (int64_t)w * j];
          Q[i + (int64_t)w * j] = s / Lk(i, i);
        }
      (void)rs;   // (root fronts: root_update())
    }
}

void root_update(const PlanD& P, const BatchD& B, const FrontD& F, int32_t nbc) {
  const int w = F.w, ri = F.ri, rs = F.rs, m = w + ri + rs;
  const int32_t* rel = P.rel + F.rel_off;
  for (int slot = 0; slot < nbc; slot++) {
    const double* A = B.scratch + (int64_t)slot * P.scratch_size + F.f_off;
    double* S = B.sblock + (int64_t)slot * P.nS * P.nS;
    for (int bb = 0; bb < rs; bb++)
      for (int a = 0; a < rs; a++) S[rel[ri + a] + (int64_t)P.nS * rel[ri + bb]] += A[(w + ri + a) + (int64_t)m * (w + ri + bb)];
  }
}

void factor_level(const PlanD& P, const BatchD& B, const int32_t* list, int32_t count, int32_t b0, int32_t nbc,
                  const double* kval, int32_t) {
  for (int slot = 0; slot < nbc; slot++)
    for (int q = 0; q < count; q++) sim_factor_front(P, B, P.fronts[list[q]], slot, b0 + slot, kval);
}
void factor_big_front(const PlanD& P, const BatchD& B, const FrontD& F, const FrontD*, int32_t, int32_t b0, int32_t nbc,
                      const double* kval) {
  for (int slot = 0; slot < nbc; slot++) sim_factor_front(P, B, F, slot, b0 + slot, kval);
  if (F.parent < 0 && F.rs > 0) root_update(P, B, F, nbc);   // (as the HIP launcher does)
}

static void sim_fwd_front(const PlanD& P, const BatchD& B, const FrontD& F, int b, double* x) {
  std::vector<double> f;
    {
      const int w = F.w, ri = F.ri, ld = w + ri;
      f.assign(ld, 0.0);
      double* xb = x + B.xoff[b];
      double* cb = B.contrib + (int64_t)b * P.contrib_size;
      for (int j = 0; j < w; j++) f[j] = xb[F.c0 + j];
      for (int j = 0; j < ld; j++)   // assembly pull lists
        for (int t = P.asm_ptr[F.a_off + j]; t < P.asm_ptr[F.a_off + j + 1]; t++) f[j] += cb[P.asm_src[t]];
      const double* Lp = B.factor + (int64_t)b * P.factor_size + F.lp_off;
      for (int i = 0; i < w; i++) {
        double s = f[i];
        for (int k = 0; k < i; k++) s += Lp[i + (int64_t)ld * k] * f[k];
        xb[F.c0 + i] = s;
      }
      for (int i = 0; i < ri; i++) {
        double s = f[w + i];
        for (int k = 0; k < w; k++) s -= Lp[(w + i) + (int64_t)ld * k] * f[k];
        cb[F.c_off + i] = s;
      }
    }
}

void solve_fwd_level(const PlanD& P, const BatchD& B, const int32_t* list, int32_t count, double* x) {
  for (int b = 0; b < B.nb; b++) for (int q = 0; q < count; q++) sim_fwd_front(P, B, P.fronts[list[q]], b, x);
}
void solve_fwd_big(const PlanD& P, const BatchD& B, const int32_t* list, const FrontD*, const int64_t*, int32_t count, double* x) {
  for (int b = 0; b < B.nb; b++) for (int q = 0; q < count; q++) sim_fwd_front(P, B, P.fronts[list[q]], b, x);
}

static void sim_bwd_front(const PlanD& P, const BatchD& B, const FrontD& F, int b, double* x) {
  std::vector<double> g, out;
    {
      const int w = F.w, ri = F.ri, ld = w + ri;
      double* xb = x + B.xoff[b];
      g.resize(ld); out.resize(w);
      for (int k = 0; k < w; k++) g[k] = xb[F.c0 + k];
      for (int k = 0; k < ri; k++) g[w + k] = xb[P.fidx[F.idx_off + w + k]];
      const double* fac = B.factor + (int64_t)b * P.factor_size;
      const double* Lp = fac + F.lp_off;
      const double* Q = fac + F.q_off;
      for (int i = 0; i < w; i++) {
        double s = 0;
        for (int k = i; k < w; k++) s += Lp[i + (int64_t)ld * k] * g[k];
        for (int k = 0; k < ri; k++) s -= Q[i + (int64_t)w * k] * g[w + k];
        out[i] = s;
      }
      for (int i = 0; i < w; i++) xb[F.c0 + i] = out[i];
    }
}

void solve_bwd_level(const PlanD& P, const BatchD& B, const int32_t* list, int32_t count, double* x) {
  for (int b = 0; b < B.nb; b++) for (int q = 0; q < count; q++) sim_bwd_front(P, B, P.fronts[list[q]], b, x);
}
void solve_bwd_big(const PlanD& P, const BatchD& B, const int32_t* list, const FrontD*, const int64_t*, int32_t count, double* x) {
  for (int b = 0; b < B.nb; b++) for (int q = 0; q < count; q++) sim_bwd_front(P, B, P.fronts[list[q]], b, x);
}

void solve_fwd_tasks(const LvlTask* tasks, int32_t ntasks, const LvlSub* subs, const PlanD* plans, int32_t,
                     const double* x, double* y) {
  // tasks of one level are independent: results first, then the writes (as the workgroups of one launch)
  std::vector<std::pair<double*, double>> writes;
  for (int t = 0; t < ntasks; t++) {
    const LvlTask& T = tasks[t];
    const LvlSub& S = subs[T.sub];
    const PlanD& P = plans[S.cls];
    const FrontD& F = P.fronts[T.front];
    const int w = F.w, rows = F.w + F.ri;
    const double* xb = x + S.xoff;
    double* yb = y + S.xoff;
    std::vector<double> a(rows);
    for (int j = 0; j < rows; j++) {
      double v = j < w ? xb[F.c0 + j] : 0.0;
      for (int q = P.asm_ptr[F.a_off + j]; q < P.asm_ptr[F.a_off + j + 1]; q++) v += S.contrib[P.asm_src[q]];
      a[j] = v;
    }
    const double* Lp = S.fac + F.lp_off;
    const int i0 = T.r0 < 0 ? 0 : T.r0, i1 = T.r0 < 0 ? rows : std::min(rows, T.r0 + 64);
    for (int i = i0; i < i1; i++) {
      double s = 0;
      for (int k = 0; k < std::min(i, w); k++) s += Lp[i + (int64_t)rows * k] * a[k];
      if (i < w) writes.emplace_back(&yb[F.c0 + i], a[i] + s);
      else writes.emplace_back(&S.contrib[F.c_off + i - w], a[i] - s);
    }
  }
  for (auto& wv : writes) *wv.first = wv.second;
}
void solve_bwd_tasks(const LvlTask* tasks, int32_t ntasks, const LvlSub* subs, const PlanD* plans, int32_t,
                     const double* y, double* x) {
  std::vector<std::pair<double*, double>> writes;
  for (int t = 0; t < ntasks; t++) {
    const LvlTask& T = tasks[t];
    const LvlSub& S = subs[T.sub];
    const PlanD& P = plans[S.cls];
    const FrontD& F = P.fronts[T.front];
    const int w = F.w, ri = F.ri, ld = w + ri;
    double* xb = x + S.xoff;
    const double* yb = y + S.xoff;
    const double* Lp = S.fac + F.lp_off;
    const double* Q = S.fac + F.q_off;
    const int i0 = T.r0 < 0 ? 0 : T.r0, i1 = T.r0 < 0 ? w : std::min(w, T.r0 + 64);
    for (int i = i0; i < i1; i++) {
      double s = 0;
      for (int k = i; k < w; k++) s += Lp[i + (int64_t)ld * k] * yb[F.c0 + k];
      for (int k = 0; k < ri; k++) s -= Q[i + (int64_t)w * k] * xb[P.fidx[F.idx_off + w + k]];
      writes.emplace_back(&xb[F.c0 + i], s);
    }
  }
  for (auto& wv : writes) *wv.first = wv.second;
}

void repack_fronts(const PlanD& P, const BatchD& B, int32_t b0, int32_t nbc) {
  std::vector<double> tmp;
  for (int b = b0; b < b0 + nbc; b++)
    for (int s = 0; s < P.nfronts; s++) {
      const FrontD& F = P.fronts[s];
      const int64_t w = F.w, ri = F.ri, ld = w + ri;
      double* Lp = B.factor + (int64_t)b * P.factor_size + F.lp_off;
      tmp.assign(Lp, Lp + ld * w);
      for (int64_t k = 0; k < w; k++)
        for (int64_t i = 0; i < ld; i++) {
          const double v = tmp[i + ld * k];
          if (i >= w) Lp[packed_l21(w, ri, i - w, k)] = v;
          else if (i > k) Lp[packed_lower(w, i, k)] = v;
          else Lp[packed_upper(w, ri, i, k)] = v;
        }
    }
}

void interior_solve_fused(int32_t nsub, const FusedSub* subs, const PlanD* plans, int32_t, double* x, const FusedIO* iop) {
  // level-synchronous walk with the same work-item tables the HIP kernel uses
  const FusedIO io = iop ? *iop : FusedIO();
  std::vector<double> C, Fv, out, Xl;
  for (int b = 0; b < nsub; b++) {
    const PlanD& P = plans[subs[b].cls];
    const int xoff = subs[b].xoff;
    Xl.assign(std::max(P.nI, 1), 0.0);
    for (int i = 0; i < P.nI; i++) {
      if (io.in == 0) Xl[i] = x[xoff + i];
      else if (io.in == 1) Xl[i] = io.b[io.perm[xoff + i]];
      else { double v = 0; for (int e = io.a_row[xoff + i]; e < io.a_row[xoff + i + 1]; e++) v += io.a_val[e] * io.x2[io.a_col[e]]; Xl[i] = v; }
    }
    double* X = Xl.data();
    const double* fac = subs[b].fac;
    C.assign(std::max(P.contrib_size, 1), 0.0);
    Fv.assign(std::max(P.max_level_rows, 1), 0.0);
    for (int lev = 0; lev < P.nlev; lev++) {
      for (int it = P.fw_ptr[lev]; it < P.fw_ptr[lev + 1]; it++) {
        const FrontD& F = P.fronts[P.fw_items[it] >> 16];
        const int r = P.fw_items[it] & 0xffff;
        double v = r < F.w ? X[F.c0 + r] : 0.0;
        for (int t = P.asm_ptr[F.a_off + r]; t < P.asm_ptr[F.a_off + r + 1]; t++) v += C[P.asm_src[t]];
        if (r < F.w) Fv[F.lf_off + r] = v; else C[F.c_off + r - F.w] = v;   // update rows are assembled in place
      }
      for (int it = P.fw_ptr[lev]; it < P.fw_ptr[lev + 1]; it++) {
        const FrontD& F = P.fronts[P.fw_items[it] >> 16];
        const int r = P.fw_items[it] & 0xffff, w = F.w, ld = F.w + F.ri;
        const double* Lp = fac + F.lp_off;
        const int kmax = r < w ? r : w;
        double a = 0;
        for (int k = 0; k < kmax; k++) {
          const double l = !P.packed ? Lp[r + (int64_t)ld * k] : (r < w ? Lp[packed_lower(w, r, k)] : Lp[packed_l21(w, F.ri, r - w, k)]);
          a += l * Fv[F.lf_off + k];
        }
        if (r < w) X[F.c0 + r] = Fv[F.lf_off + r] + a; else C[F.c_off + r - w] -= a;
      }
    }
    for (int lev = P.nlev - 1; lev >= 0; lev--) {
      out.assign(P.bw_ptr[lev + 1] - P.bw_ptr[lev], 0.0);
      for (int it = P.bw_ptr[lev]; it < P.bw_ptr[lev + 1]; it++) {
        const FrontD& F = P.fronts[P.bw_items[it] >> 16];
        const int i = P.bw_items[it] & 0xffff, w = F.w, ri = F.ri, ld = w + ri;
        const double* Lp = fac + F.lp_off;
        const double* Q = fac + F.q_off;
        double a = 0;
        for (int k = i; k < w; k++) a += (P.packed ? Lp[packed_upper(w, ri, i, k)] : Lp[i + (int64_t)ld * k]) * X[F.c0 + k];
        for (int k = 0; k < ri; k++) a -= Q[i + (int64_t)w * k] * X[P.fidx[F.idx_off + w + k]];
        out[it - P.bw_ptr[lev]] = a;
      }
      for (int it = P.bw_ptr[lev]; it < P.bw_ptr[lev + 1]; it++) {
        const FrontD& F = P.fronts[P.bw_items[it] >> 16];
        X[F.c0 + (P.bw_items[it] & 0xffff)] = out[it - P.bw_ptr[lev]];
      }
    }
    for (int i = 0; i < P.nI; i++) {
      if (io.out == 0) x[xoff + i] = X[i];
      else io.user[io.perm[xoff + i]] = io.z[xoff + i] - X[i];
    }
  }
}

void ot_apply(int32_t ng, const int32_t* gptr, const double* w, double* x) {
  for (int g = 0; g < ng; g++) {
    double s = 0;
    for (int i = gptr[g]; i < gptr[g + 1]; i++) s += w[i] * x[i];
    for (int i = gptr[g]; i < gptr[g + 1]; i++) x[i] = 2.0 * w[i] * s - x[i];
  }
}

static void hh_rows(double* X, int64_t ld, int ncols, int pos, int n, const double* vin) {
  // Householder::Apply semantics (reference src/HYMLS_Householder.cpp:38-80) on rows pos..pos+n
  std::vector<double> v(vin, vin + n);
  const double sg = v[0] < 0 ? -1.0 : (v[0] > 0 ? 1.0 : 0.0);
  double nrm = 0;
  for (double& t : v) { t *= sg; nrm += t * t; }
  nrm = std::sqrt(nrm);
  const double v1 = v[0] + nrm;
  if (std::abs(v1) < SMALL_ENTRY || nrm < SMALL_ENTRY) return;
  const double fac1 = 1.0 / (nrm * v1);
  for (int k = 0; k < ncols; k++) {
    double* c = X + (int64_t)ld * k + pos;
    double fac2 = nrm * c[0];
    for (int i = 0; i < n; i++) fac2 += c[i] * v[i];
    const double fac = fac1 * fac2;
    c[0] = v1 * fac - c[0];
    for (int i = 1; i < n; i++) c[i] = v[i] * fac - c[i];
  }
}

void sblock_transform(int32_t nS, int32_t ng, const int32_t* gptr, const double* tv, double* sblock, int32_t nbc) {
  std::vector<double> T((size_t)nS * nS);
  for (int b = 0; b < nbc; b++) {
    double* S = sblock + (int64_t)b * nS * nS;
    const double* v = tv + (int64_t)b * nS;
    for (int g = 0; g < ng; g++) hh_rows(S, nS, nS, gptr[g], gptr[g + 1] - gptr[g], v + gptr[g]);
    for (int i = 0; i < nS; i++) for (int j = 0; j < nS; j++) T[j + (size_t)nS * i] = S[i + (size_t)nS * j];
    for (int g = 0; g < ng; g++) hh_rows(T.data(), nS, nS, gptr[g], gptr[g + 1] - gptr[g], v + gptr[g]);
    for (int i = 0; i < nS; i++) for (int j = 0; j < nS; j++) S[i + (size_t)nS * j] = T[j + (size_t)nS * i];
  }
}

// kept entries of the transformed block: the simulator transforms a copy and reads the kept positions
bool sblock_kept_fits(int32_t, int32_t) { return true; }
void sblock_kept(const KeptD& K, const double* tv, const double* sblock, double* out, int64_t out_stride, int32_t nbc) {
  const int nS = K.nS, ngl = K.ngl;
  std::vector<double> T((size_t)nS * nS);
  for (int s = 0; s < nbc; s++) {
    std::memcpy(T.data(), sblock + (int64_t)s * nS * nS, (size_t)nS * nS * sizeof(double));
    sblock_transform(nS, ngl, K.gptr, tv + (int64_t)s * nS, T.data(), 1);
    double* rec = out + (int64_t)s * out_stride;
    for (int J = 0; J < ngl; J++)
      for (int I = 0; I < ngl; I++) {
        rec[I + (int64_t)ngl * J] = T[K.gptr[I] + (int64_t)nS * K.gptr[J]];
        if (K.glink[I] >= 0 && K.glink[I] == K.glink[J]) {
          const int L = K.glink[I], nI = K.gptr[I + 1] - K.gptr[I], nJ = K.gptr[J + 1] - K.gptr[J];
          for (int b = 1; b < nJ; b++)
            for (int a = 1; a < nI; a++)
              rec[K.lboff[L] + (K.goff[I] + a - 1) + (int64_t)K.lblen[L] * (K.goff[J] + b - 1)] = T[(K.gptr[I] + a) + (int64_t)nS * (K.gptr[J] + b)];
        }
      }
  }
}

void sblock_extract(int32_t nS, int64_t npick, const int32_t* pick, const double* sblock, double* out,
                    int64_t out_stride, int32_t nbc) {
  for (int b = 0; b < nbc; b++)
    for (int64_t k = 0; k < npick; k++) out[(int64_t)b * out_stride + k] = sblock[(int64_t)b * nS * nS + pick[k]];
}

bool dense_invert_blocked_order(int32_t) { return false; }
void dense_invert(int32_t nb, int32_t nblk, double* blocks, int32_t* flag) {
  std::vector<double> W((size_t)nb * 2 * nb);
  for (int B = 0; B < nblk; B++) {
    double* A = blocks + (int64_t)B * nb * nb;
    // Gauss-Jordan with partial pivoting on [A | I], column-major with ld = nb, 2nb columns
    for (int j = 0; j < nb; j++) for (int i = 0; i < nb; i++) { W[i + (size_t)nb * j] = A[i + (size_t)nb * j]; W[i + (size_t)nb * (nb + j)] = (i == j); }
    for (int k = 0; k < nb; k++) {
      int p = k; double best = std::abs(W[k + (size_t)nb * k]);
      for (int i = k + 1; i < nb; i++) if (std::abs(W[i + (size_t)nb * k]) > best) { best = std::abs(W[i + (size_t)nb * k]); p = i; }
      if (best == 0.0 || !std::isfinite(best)) { *flag = 1; break; }
      if (p != k) for (int j = 0; j < 2 * nb; j++) std::swap(W[k + (size_t)nb * j], W[p + (size_t)nb * j]);
      const double ip = 1.0 / W[k + (size_t)nb * k];
      for (int j = 0; j < 2 * nb; j++) W[k + (size_t)nb * j] *= ip;
      for (int i = 0; i < nb; i++) {
        if (i == k) continue;
        const double l = W[i + (size_t)nb * k];
        if (l != 0.0) for (int j = 0; j < 2 * nb; j++) W[i + (size_t)nb * j] -= l * W[k + (size_t)nb * j];
      }
    }
    for (int j = 0; j < nb; j++) for (int i = 0; i < nb; i++) A[i + (size_t)nb * j] = W[i + (size_t)nb * (nb + j)];
  }
}

void blocks_apply(int32_t nb, int32_t nblk, const double* binv, const int32_t* ids, const double* x, double* y) {
  for (int B = 0; B < nblk; B++) {
    const double* M = binv + (int64_t)B * nb * nb;
    const int32_t* id = ids + (int64_t)B * nb;
    for (int i = 0; i < nb; i++) {
      double s = 0;
      for (int j = 0; j < nb; j++) s += M[i + (size_t)nb * j] * x[id[j]];
      y[id[i]] = s;
    }
  }
}

double dot(int64_t n, const double* x, const double* y) {
  double s = 0;
  for (int64_t i = 0; i < n; i++) s += x[i] * y[i];
  return s;
}

void solve_transposed(const PlanD& P, const BatchD& B, const int32_t* order, int32_t nfronts, int32_t, double* x) {
  std::vector<double> a;
  for (int b = 0; b < B.nb; b++) {
    double* xb = x + B.xoff[b];
    double* cb = B.contrib + (int64_t)b * P.contrib_size;
    const double* fac = B.factor + (int64_t)b * P.factor_size;
    auto linv = [&](const FrontD& F, const double* lp, int64_t i, int64_t k) { return P.packed ? lp[packed_lower(F.w, i, k)] : lp[i + (int64_t)(F.w + F.ri) * k]; };
    auto uinv = [&](const FrontD& F, const double* lp, int64_t i, int64_t k) { return P.packed ? lp[packed_upper(F.w, F.ri, i, k)] : lp[i + (int64_t)(F.w + F.ri) * k]; };
    auto pl = [&](const FrontD& F, const double* lp, int64_t j, int64_t k) { return P.packed ? lp[packed_l21(F.w, F.ri, j, k)] : lp[(F.w + j) + (int64_t)(F.w + F.ri) * k]; };
    for (int q = 0; q < nfronts; q++) {
      const FrontD& F = P.fronts[order[q]];
      const int w = F.w, ri = F.ri;
      const double* lp = fac + F.lp_off; const double* Q = fac + F.q_off;
      a.assign(w + ri, 0.0);
      for (int j = 0; j < w + ri; j++) {
        double v = j < w ? xb[F.c0 + j] : 0.0;
        for (int t = P.asm_ptr[F.a_off + j]; t < P.asm_ptr[F.a_off + j + 1]; t++) v += cb[P.asm_src[t]];
        a[j] = v;
      }
      for (int i = 0; i < w; i++) { double s = 0; for (int k = 0; k <= i; k++) s += uinv(F, lp, k, i) * a[k]; xb[F.c0 + i] = s; }
      for (int j = 0; j < ri; j++) { double s = 0; for (int k = 0; k < w; k++) s += Q[k + (int64_t)w * j] * a[k]; cb[F.c_off + j] = a[w + j] - s; }
    }
    for (int q = nfronts - 1; q >= 0; q--) {
      const FrontD& F = P.fronts[order[q]];
      const int w = F.w, ri = F.ri;
      const double* lp = fac + F.lp_off;
      a.assign(w + ri, 0.0);
      for (int k = 0; k < w + ri; k++) a[k] = k < w ? xb[F.c0 + k] : xb[P.fidx[F.idx_off + k]];
      std::vector<double> out(w);
      for (int i = 0; i < w; i++) {
        double s = a[i];
        for (int k = i + 1; k < w; k++) s += linv(F, lp, k, i) * a[k];
        for (int j = 0; j < ri; j++) s -= pl(F, lp, j, i) * a[w + j];
        out[i] = s;
      }
      for (int i = 0; i < w; i++) xb[F.c0 + i] = out[i];
    }
  }
}

void dense_invert_all(int32_t nblk, const BlkD* blocks, int32_t, int32_t* flag) {
  for (int b = 0; b < nblk; b++) dense_invert(blocks[b].nb, 1, const_cast<double*>(blocks[b].binv), flag);
}

void blocks_apply_all(int32_t nblk, const BlkD* blocks, int32_t, const double* x, double* y) {
  std::vector<double> xs, out;
  for (int b = 0; b < nblk; b++) {
    const BlkD& D = blocks[b];
    xs.resize(D.nb); out.resize(D.nb);
    for (int j = 0; j < D.nb; j++) xs[j] = x[D.ids[j]];
    const int i0 = D.r0 < 0 ? 0 : D.r0, i1 = D.r0 < 0 ? D.nb : std::min(D.nb, D.r0 + 64);
    for (int i = i0; i < i1; i++) {
      double s = 0;
      for (int j = 0; j < D.nb; j++) s += D.binv[i + (int64_t)D.nb * j] * xs[j];
      out[i] = s;
    }
    for (int i = i0; i < i1; i++) y[D.ids[i]] = out[i];
  }
}


// several right-hand sides: the simulator walks over the columns; column v uses its own contribution scratch
// (cstride doubles behind that of column v - 1, as in the HIP kernels)
static std::vector<LvlSub> column_subs(const LvlTask* tasks, int32_t ntasks, const LvlSub* subs, int v) {
  int32_t ns = 0;
  for (int32_t t = 0; t < ntasks; t++) ns = std::max(ns, tasks[t].sub + 1);
  std::vector<LvlSub> out(subs, subs + ns);
  for (auto& s : out) s.contrib += (int64_t)v * s.cstride;
  return out;
}
void interior_solve_fused_mv(int32_t nsub, const FusedSub* subs, const PlanD* plans, int32_t lds_doubles, int32_t, double* x, int64_t ldx, int nv) {
  for (int v = 0; v < nv; v++) interior_solve_fused(nsub, subs, plans, lds_doubles, x + v * ldx, nullptr);
}
void solve_fwd_tasks_mv(const LvlTask* tasks, int32_t ntasks, const LvlSub* subs, const PlanD* plans, int32_t lds, const double* x, double* y, int64_t ld, int nv) {
  for (int v = 0; v < nv; v++) { auto sv = column_subs(tasks, ntasks, subs, v); solve_fwd_tasks(tasks, ntasks, sv.data(), plans, lds, x + v * ld, y + v * ld); }
}
void solve_bwd_tasks_mv(const LvlTask* tasks, int32_t ntasks, const LvlSub* subs, const PlanD* plans, int32_t lds, const double* y, double* x, int64_t ld, int nv) {
  for (int v = 0; v < nv; v++) { auto sv = column_subs(tasks, ntasks, subs, v); solve_bwd_tasks(tasks, ntasks, sv.data(), plans, lds, y + v * ld, x + v * ld); }
}
void blocks_apply_all_mv(int32_t nblk, const BlkD* blocks, int32_t max_nb, const double* x, int64_t ldx, double* y, int64_t ldy, int nv) {
  for (int v = 0; v < nv; v++) blocks_apply_all(nblk, blocks, max_nb, x + v * ldx, y + v * ldy);
}

}  // namespace dev
}  // namespace hymls
