// device_hip.hip -- HIP (gfx950 / MI355X) implementation of the device boundary (device.hpp).
//
// Every floating-point operation of Compute() and ApplyInverse() runs in the kernels of this
// file, on one stream (plus side streams for independent setup work).  Kernel overview (FP64 throughout):
//   apply path (HBM-bound):
//     k_interior_fused           one workgroup per subdomain walks its whole assembly tree level by level with
//                                the solution, the contribution vectors and the front descriptors in LDS; the
//                                factor panels (column-major or packed, rows on consecutive lanes) are streamed once
//     k_lvl_fwd / k_lvl_bwd      subdomains too large for LDS: one launch per tree level for all classes, a task is a
//                                whole small front or a 64-row tile of a large one
//     k_solve_* / k_panel_*      the coarse direct solver (one class, one member): small fronts per workgroup,
//                                wide supernodes as tiled panel-times-vector products with a fixed-order finalize
//     k_spmv                     CSR SpMV, 1-8 lanes per row + sub-wave shuffle reduction
//     k_ot                       per-group Householder: 8 lanes per group, dot + axpy
//     k_blocks_apply_all         dense block inverse times vector for every separator block of a level
//     k_gather/k_scatter/k_axpby vector glue and exchange packing
//   setup path:
//     k_factor_level             multifrontal front: assemble, LU of the pivot block, triangular inverses, panel
//                                products and the Schur update as workgroup-level tiled GEMMs (LDS staged)
//     k_big_* / k_gemm_f64       wide supernodes spread over many workgroups; FP64 MFMA GEMMs with triangular masks
//     k_gemm_f64_big             the same products with 128 x 128 tiles and a register prefetch of the next K slab (rank >= 256
//                                updates, or rank >= 64 when the launch fills the chip); factor_big_front works in outer blocks
//                                of 512 columns so that the trailing matrix is read and written once per rank-512 update;
//                                panel_trmm: the panel products of a pivot piece in place on the same kernels
//     k_big_pivot_blk            pivot pieces (<= 128 x 128): 32 x 32 diagonal blocks factored and inverted in registers,
//                                everything else as MFMA tile products between LDS operands (k_big_pivot: scalar variant)
//     k_repack                   packed L-side panels for the classes of the fused solve
//     k_sblock_*                 separator block init / transformation + dropping in one pass (k_sblock_kept) / extraction
//     k_gj_* / k_dense_invert    separator blocks: blocked Gauss-Jordan with partial pivoting (32 pivots per panel in registers,
//                                rank-32 update on the matrix cores) / per-block Gauss-Jordan in LDS or global memory
//     k_pull_sum*                deterministic assembly of the kept Schur entries
//   tables of Initialize built on the device (integers):
//     k_member_sources           entry of the level matrix behind every entry of a member's extended local CSR (binary search)
//     k_offdiag<count / fill>    the blocks A12 / A21 cut out of the level matrix
//     k_build_pull_tables        pull pointers / indices of the reduced matrix from its sorted keys
//     k_solve_transposed, k_dot  bordered systems
#include <hip/hip_runtime.h>
#include <type_traits>
#include <dlfcn.h>
#include <chrono>
#include <cstring>
#include <array>
#include <map>
#include <string>
#include <cstdio>
#include "device.hpp"

namespace hymls {
namespace dev {

#define HIP_CHECK(call)                                                                    \
  do {                                                                                     \
    hipError_t e_ = (call);                                                                \
    if (e_ != hipSuccess)                                                                  \
      throw ::hymls::Error(-3, std::string("HIP error: ") + hipGetErrorString(e_) + " at " + \
                                   __FILE__ + ":" + std::to_string(__LINE__));             \
  } while (0)

// ---- per-handle device context: device ordinal, the stream every launcher uses (main or one of the side streams),
// side streams and their events, setup arenas (one per stream), timers, profiling marks and small scratch buffers.
// A handle binds its context at every API entry (capi.cpp); nothing device-related is process-global, so any number
// of preconditioners (on the same or on different devices) can live in one process, as in the reference.
struct MarkRec { int phase; bool begin; hipEvent_t ev; };
struct Context {
  int device = 0;
  hipStream_t cur = nullptr, main = nullptr, side[NSIDE] = {};
  hipEvent_t fork_ev = nullptr, join_ev[NSIDE] = {};
  int cur_idx = 0;
  bool side_init = false;
  void* arena[NSIDE + 1] = {};
  size_t arena_cap[NSIDE + 1] = {};
  hipEvent_t ev[16][2] = {};
  bool ev_init = false;
  std::vector<MarkRec> marks;
  std::vector<hipEvent_t> pool;
  double* dpart = nullptr;     // partial sums of dot()
  double* zeros = nullptr;     // 16 zeros
  bool big_attr_set = false;
};
static thread_local Context* t_ctx = nullptr;
static inline Context& ctx() {
  if (!t_ctx) throw Error(-3, "no device context bound (internal error: API entry without dev::bind)");
  return *t_ctx;
}
#define g_stream (ctx().cur)

Context* create_context(int device) {
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count == 0)
    throw Error(-3, "no HIP device available: hymls_amd needs an AMD GPU (there is no CPU fallback)");
  if (device < 0 || device >= count) throw Error(-2, "HIP device ordinal out of range");
  HIP_CHECK(hipSetDevice(device));
  Context* c = new Context();
  c->device = device;
  hipError_t es = hipStreamCreate(&c->main);
  if (es != hipSuccess) { delete c; HIP_CHECK(es); }
  c->cur = c->main;
  return c;
}
Context* current() { return t_ctx; }
void bind(Context* c) {
  t_ctx = c;
  if (c) HIP_CHECK(hipSetDevice(c->device));
}
void destroy_context(Context* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->main);
  for (int k = 0; k <= NSIDE; k++) if (c->arena[k]) (void)hipFree(c->arena[k]);
  if (c->dpart) (void)hipFree(c->dpart);
  if (c->zeros) (void)hipFree(c->zeros);
  for (auto& m : c->marks) (void)hipEventDestroy(m.ev);
  for (auto& e : c->pool) (void)hipEventDestroy(e);
  if (c->ev_init) for (auto& e : c->ev) { (void)hipEventDestroy(e[0]); (void)hipEventDestroy(e[1]); }
  if (c->side_init) {
    for (int k = 0; k < NSIDE; k++) { (void)hipStreamSynchronize(c->side[k]); (void)hipStreamDestroy(c->side[k]); (void)hipEventDestroy(c->join_ev[k]); }
    (void)hipEventDestroy(c->fork_ev);
  }
  (void)hipStreamDestroy(c->main);
  if (t_ctx == c) t_ctx = nullptr;
  delete c;
}
void* stream() { return (void*)ctx().main; }
static void init_side_streams(Context& c) {
  if (c.side_init) return;
  for (int k = 0; k < NSIDE; k++) { HIP_CHECK(hipStreamCreate(&c.side[k])); HIP_CHECK(hipEventCreateWithFlags(&c.join_ev[k], hipEventDisableTiming)); }
  HIP_CHECK(hipEventCreateWithFlags(&c.fork_ev, hipEventDisableTiming));
  c.side_init = true;
}
void fork_streams() {
  Context& c = ctx();
  init_side_streams(c);
  HIP_CHECK(hipEventRecord(c.fork_ev, c.main));
  for (int k = 0; k < NSIDE; k++) HIP_CHECK(hipStreamWaitEvent(c.side[k], c.fork_ev, 0));
}
void use_stream(int k) { Context& c = ctx(); c.cur_idx = k; c.cur = k == 0 ? c.main : c.side[k - 1]; }
void join_streams() {
  Context& c = ctx();
  for (int k = 0; k < NSIDE; k++) { HIP_CHECK(hipEventRecord(c.join_ev[k], c.side[k])); HIP_CHECK(hipStreamWaitEvent(c.main, c.join_ev[k], 0)); }
  use_stream(0);   // (the side arenas stay allocated: re-allocating them costs about a second per Compute at 256^3)
}
void* alloc(size_t bytes) {
  void* p = nullptr;
  static const bool fine = std::getenv("HYMLS_MI_VERBOSE") && std::atoi(std::getenv("HYMLS_MI_VERBOSE")) >= 2;
  const auto t0 = std::chrono::steady_clock::now();
  HIP_CHECK(hipMalloc(&p, std::max<size_t>(bytes, 8)));
  if (fine && bytes >= ((size_t)1 << 30))
    std::fprintf(stderr, "[hymls_mi]       . hipMalloc of %.1f GiB %.3f s\n", bytes / 1073741824.0,
                 std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
  return p;
}
void free(void* p) { if (p) (void)hipFree(p); }
// (Measured, tools/copy_rate.py: a plain hipMemcpy from / to pageable memory runs at 54 - 56 GB/s on this platform, the same as
// a copy staged through two pinned buffers by all setup threads: no staging here.)
void h2d(void* d, const void* s, size_t n) {
  if (!n) return;
  HIP_CHECK(hipMemcpyAsync(d, s, n, hipMemcpyHostToDevice, g_stream));
  HIP_CHECK(hipStreamSynchronize(g_stream));
}
void d2h(void* d, const void* s, size_t n) {
  if (!n) return;
  HIP_CHECK(hipMemcpyAsync(d, s, n, hipMemcpyDeviceToHost, g_stream));
  HIP_CHECK(hipStreamSynchronize(g_stream));
}
void d2d(void* d, const void* s, size_t n) {
  if (!n) return;
  HIP_CHECK(hipMemcpyAsync(d, s, n, hipMemcpyDeviceToDevice, g_stream));
}
void zero(void* d, size_t n) { if (n) HIP_CHECK(hipMemsetAsync(d, 0, n, g_stream)); }
void sync() { HIP_CHECK(hipStreamSynchronize(g_stream)); }
void* shared_scratch(size_t bytes) {
  Context& c = ctx();
  const int k = c.cur_idx;
  if (bytes > c.arena_cap[k]) {
    sync();
    if (c.arena[k]) (void)hipFree(c.arena[k]);
    c.arena[k] = nullptr; c.arena_cap[k] = 0;
    c.arena[k] = alloc(bytes);
    c.arena_cap[k] = bytes;
  }
  return c.arena[k];
}
const double* zeros16() {
  Context& c = ctx();
  if (!c.zeros) { c.zeros = (double*)alloc(16 * sizeof(double)); zero(c.zeros, 16 * sizeof(double)); }
  return c.zeros;
}
size_t mem_free() { size_t f = 0, t = 0; HIP_CHECK(hipMemGetInfo(&f, &t)); return f; }
namespace {
struct Roctx {
  int (*push)(const char*) = nullptr;
  int (*pop)() = nullptr;
  std::FILE* log = nullptr;
  Roctx() {
    for (const char* name : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"}) {
      void* h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (!h) continue;
      push = (int (*)(const char*))dlsym(h, "roctxRangePushA");
      pop = (int (*)())dlsym(h, "roctxRangePop");
      if (push && pop) break;
      push = nullptr; pop = nullptr;
    }
    if (const char* f = std::getenv("HYMLS_MI_RANGE_LOG")) log = std::fopen(f, "a");
  }
};
Roctx& roctx() { static Roctx r; return r; }
}  // namespace
void range_push(const char* label) {
  Roctx& r = roctx();
  if (r.push) r.push(label);
  if (r.log) { std::fprintf(r.log, "push %s\n", label); std::fflush(r.log); }
}
void range_pop() {
  Roctx& r = roctx();
  if (r.pop) r.pop();
  if (r.log) { std::fprintf(r.log, "pop\n"); std::fflush(r.log); }
}
void timer_start(int id) {
  Context& c = ctx();
  if (!c.ev_init) {
    for (auto& e : c.ev) { HIP_CHECK(hipEventCreate(&e[0])); HIP_CHECK(hipEventCreate(&e[1])); }
    c.ev_init = true;
  }
  HIP_CHECK(hipEventRecord(c.ev[id][0], g_stream));
}
double timer_stop(int id) {
  Context& c = ctx();
  HIP_CHECK(hipEventRecord(c.ev[id][1], g_stream));
  HIP_CHECK(hipEventSynchronize(c.ev[id][1]));
  float ms = 0;
  HIP_CHECK(hipEventElapsedTime(&ms, c.ev[id][0], c.ev[id][1]));
  return 1e-3 * ms;
}

void mark(int phase, bool begin) {
  Context& c = ctx();
  hipEvent_t e;
  if (!c.pool.empty()) { e = c.pool.back(); c.pool.pop_back(); }
  else HIP_CHECK(hipEventCreate(&e));
  HIP_CHECK(hipEventRecord(e, g_stream));
  c.marks.push_back({phase, begin, e});
}
void profile_collect(double* sum, int* cnt) {
  Context& c = ctx();
  HIP_CHECK(hipStreamSynchronize(g_stream));
  hipEvent_t open[8] = {};
  for (auto& m : c.marks) {
    if (m.begin) open[m.phase] = m.ev;
    else if (open[m.phase]) {
      float ms = 0;
      HIP_CHECK(hipEventElapsedTime(&ms, open[m.phase], m.ev));
      sum[m.phase] += 1e-3 * ms;
      cnt[m.phase]++;
    }
  }
  for (auto& m : c.marks) c.pool.push_back(m.ev);
  c.marks.clear();
}

static inline void launch_check() { HIP_CHECK(hipGetLastError()); }

// HYMLS_MI_SETUP_PROF=1: the launches of the big-front path are timed one by one (events + a synchronisation per scope, so
// the run itself is slower) and a table of shapes, time and rate goes to stderr when the library is unloaded
// (tools/setup_prof.py reads it).
struct SetupProf {
  struct Row { double ms = 0, flop = 0; long calls = 0; };
  std::map<std::pair<std::string, std::array<int, 5>>, Row> rows;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  ~SetupProf() {
    for (auto& kv : rows)
      std::fprintf(stderr, "SETUPPROF %s %d %d %d %d %d calls %ld ms %.4f tflops %.3f\n", kv.first.first.c_str(), kv.first.second[0],
                   kv.first.second[1], kv.first.second[2], kv.first.second[3], kv.first.second[4], kv.second.calls, kv.second.ms,
                   kv.second.ms > 0 ? kv.second.flop / (kv.second.ms * 1e9) : 0.0);
  }
  static SetupProf* get() {
    static const bool on = std::getenv("HYMLS_MI_SETUP_PROF") && std::atoi(std::getenv("HYMLS_MI_SETUP_PROF")) > 0;
    static SetupProf prof;
    return on ? &prof : nullptr;
  }
  struct Scope {
    SetupProf* p;
    std::pair<std::string, std::array<int, 5>> key;
    double flop;
    Scope(const char* name, int a, int b, int c, int d, int e, double flop_ = 0) : p(get()), flop(flop_) {
      if (!p) return;
      key = {name, {a, b, c, d, e}};
      if (!p->e0) { HIP_CHECK(hipEventCreate(&p->e0)); HIP_CHECK(hipEventCreate(&p->e1)); }
      HIP_CHECK(hipEventRecord(p->e0, g_stream));
    }
    ~Scope() {
      if (!p) return;
      float ms = 0;
      if (hipEventRecord(p->e1, g_stream) != hipSuccess || hipEventSynchronize(p->e1) != hipSuccess ||
          hipEventElapsedTime(&ms, p->e0, p->e1) != hipSuccess) return;
      Row& r = p->rows[key];
      r.ms += ms; r.calls++; r.flop += flop;
    }
  };
};

// Pointers that reach a kernel inside a structure (plan tables, per-subdomain slabs) are generic to the compiler, which
// then emits FLAT loads: they count on lgkmcnt as well as vmcnt and go through the LDS issue path, so every wait for an
// LDS value also waits for the panel entries in flight.  as_global() states what they are -- global memory -- and the
// loads become global_load (vmcnt only).
template <class T> using gptr = const T __attribute__((address_space(1)))*;
template <class T> __device__ inline gptr<T> as_global(const T* q) { return (gptr<T>)q; }
template <class T> using gmptr = T __attribute__((address_space(1)))*;        // writable
template <class T> __device__ inline gmptr<T> as_global_rw(T* q) { return (gmptr<T>)q; }

static inline int nblocks(int64_t n, int bs, int cap = 1 << 20) { return (int)std::max<int64_t>(1, std::min<int64_t>((n + bs - 1) / bs, cap)); }

// ------------------------------------------------------------------ vector kernels
__global__ void k_gather(int64_t n, const int32_t* __restrict__ idx, const double* __restrict__ src, double* __restrict__ dst) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] = src[idx[i]];
}
__global__ void k_scatter(int64_t n, const int32_t* __restrict__ idx, const double* __restrict__ src, double* __restrict__ dst) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[idx[i]] = src[i];
}
__global__ void k_axpby(int64_t n, double a, const double* __restrict__ x, double b, double* __restrict__ y) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) y[i] = a * x[i] + b * y[i];
}
__global__ void k_scale_copy(int64_t n, double a, const double* __restrict__ x, double* __restrict__ y) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) y[i] = a * x[i];
}
void gather(int64_t n, const int32_t* idx, const double* src, double* dst) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_gather, dim3(nblocks(n, 256, 8192)), dim3(256), 0, g_stream, n, idx, src, dst); launch_check();
}
void scatter(int64_t n, const int32_t* idx, const double* src, double* dst) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_scatter, dim3(nblocks(n, 256, 8192)), dim3(256), 0, g_stream, n, idx, src, dst); launch_check();
}
__global__ void k_scatter_add(int64_t n, const int32_t* __restrict__ idx, const double* __restrict__ src, double* __restrict__ dst) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) atomicAdd(&dst[idx[i]], src[i]);
}
void scatter_add(int64_t n, const int32_t* idx, const double* src, double* dst) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_scatter_add, dim3(nblocks(n, 256, 65536)), dim3(256), 0, g_stream, n, idx, src, dst); launch_check();
}
void axpby(int64_t n, double a, const double* x, double b, double* y) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_axpby, dim3(nblocks(n, 256, 8192)), dim3(256), 0, g_stream, n, a, x, b, y); launch_check();
}
void scale_copy(int64_t n, double a, const double* x, double* y) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_scale_copy, dim3(nblocks(n, 256, 8192)), dim3(256), 0, g_stream, n, a, x, y); launch_check();
}

// CSR SpMV: L consecutive lanes share one row (L = 1,4,16,64), values/indices of a row are
// read by consecutive lanes, partial sums combined with a sub-wave shuffle reduction.
template <int L>
__global__ void k_spmv(int32_t nrows, const int32_t* __restrict__ rp, const int32_t* __restrict__ col,
                       const double* __restrict__ val, const double* __restrict__ x, double* __restrict__ y,
                       double alpha, double beta) {
  const int lane = (int)(threadIdx.x % L);
  const int rpb = (int)blockDim.x / L;   // rows per block and pass
  // grid-stride over blocks of rows: the trip count is uniform inside a workgroup, so every lane of a sub-wave takes
  // part in the shuffles (the grid is capped at 2^20 workgroups)
  for (int64_t base = (int64_t)blockIdx.x * rpb; base < nrows; base += (int64_t)gridDim.x * rpb) {
    const int64_t row = base + threadIdx.x / L;
    double s = 0.0;
    if (row < nrows) {
      const int b = rp[row], e = rp[row + 1];
      for (int k = b + lane; k < e; k += L) s += val[k] * x[col[k]];
    }
#pragma unroll
    for (int off = L / 2; off > 0; off >>= 1) s += __shfl_down(s, off, L);
    if (row < nrows && lane == 0) y[row] = alpha * s + (beta == 0.0 ? 0.0 : beta * y[row]);
  }
}
void spmv(int32_t nrows, const int32_t* rp, const int32_t* col, const double* val, const double* x, double* y,
          double alpha, double beta, int64_t nnz_hint) {
  if (nrows <= 0) return;
  // lanes per row from the average row length (the CSR arrays live on the device: the caller passes nnz if it
  // knows it; 4 lanes otherwise, rows here have 1-33 nnz)
  const double avg = nnz_hint >= 0 ? (double)nnz_hint / nrows : 8.0;
  const int L = avg < 2.5 ? 1 : (avg < 5.0 ? 2 : (avg < 20.0 ? 4 : 8));
  const int64_t nt = (int64_t)nrows * L;
  const dim3 grid(nblocks(nt, 256));
  switch (L) {
    case 1: hipLaunchKernelGGL(k_spmv<1>, grid, dim3(256), 0, g_stream, nrows, rp, col, val, x, y, alpha, beta); break;
    case 2: hipLaunchKernelGGL(k_spmv<2>, grid, dim3(256), 0, g_stream, nrows, rp, col, val, x, y, alpha, beta); break;
    case 4: hipLaunchKernelGGL(k_spmv<4>, grid, dim3(256), 0, g_stream, nrows, rp, col, val, x, y, alpha, beta); break;
    default: hipLaunchKernelGGL(k_spmv<8>, grid, dim3(256), 0, g_stream, nrows, rp, col, val, x, y, alpha, beta); break;
  }
  launch_check();
}

__global__ void k_pull_sum(int64_t n, const int64_t* __restrict__ ptr, const int64_t* __restrict__ idx,
                           const double* __restrict__ in, double* __restrict__ out) {
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    double s = 0.0;
    for (int64_t t = ptr[e]; t < ptr[e + 1]; t++) s += in[idx[t]];
    out[e] = s;
  }
}
void pull_sum(int64_t n, const int64_t* ptr, const int64_t* idx, const double* in, double* out) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_pull_sum, dim3(nblocks(n, 256, 65536)), dim3(256), 0, g_stream, n, ptr, idx, in, out); launch_check();
}
template <bool FILL>
__global__ void __launch_bounds__(256) k_offdiag(int64_t nrows, const int32_t* __restrict__ rows, const int32_t* __restrict__ krow,
                                                 const int32_t* __restrict__ kcol, const int32_t* __restrict__ ta, const int32_t* __restrict__ tb,
                                                 const int32_t* __restrict__ excl, const int32_t* __restrict__ rowptr, int32_t* __restrict__ out,
                                                 int32_t* __restrict__ src) {
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < nrows; t += (int64_t)gridDim.x * blockDim.x) {
    const int32_t r = rows[t];
    int32_t o = FILL ? rowptr[t] : 0;
    for (int32_t e = krow[r]; e < krow[r + 1]; e++) {
      const int32_t c = kcol[e];
      int32_t tg = ta[c];
      if (tg < 0 && tb && excl[c] < 0) tg = tb[c];
      if (tg < 0) continue;
      if (FILL) { out[o] = tg; src[o] = e; }
      o++;
    }
    if (!FILL) out[t + 1] = o;
  }
}
void offdiag_count(int64_t nrows, const int32_t* rows, const int32_t* krow, const int32_t* kcol, const int32_t* ta, const int32_t* tb,
                   const int32_t* excl, int32_t* count) {
  if (nrows <= 0) return;
  hipLaunchKernelGGL(k_offdiag<false>, dim3(nblocks(nrows, 256, 1 << 20)), dim3(256), 0, g_stream, nrows, rows, krow, kcol, ta, tb, excl,
                     (const int32_t*)nullptr, count, (int32_t*)nullptr);
  launch_check();
}
void offdiag_fill(int64_t nrows, const int32_t* rows, const int32_t* krow, const int32_t* kcol, const int32_t* ta, const int32_t* tb,
                  const int32_t* excl, const int32_t* rowptr, int32_t* col, int32_t* src) {
  if (nrows <= 0) return;
  hipLaunchKernelGGL(k_offdiag<true>, dim3(nblocks(nrows, 256, 1 << 20)), dim3(256), 0, g_stream, nrows, rows, krow, kcol, ta, tb, excl,
                     rowptr, col, src);
  launch_check();
}
__global__ void __launch_bounds__(256) k_member_sources(int32_t next, int32_t nent, const int32_t* __restrict__ ext,
                                                        const int32_t* __restrict__ ent_row, const int32_t* __restrict__ ent_col,
                                                        const int32_t* __restrict__ krow, const int32_t* __restrict__ kcol,
                                                        int32_t* __restrict__ src, int32_t* __restrict__ flag) {
  const int b = blockIdx.y;
  const int32_t* nodes = ext + (int64_t)b * next;
  for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < nent; q += gridDim.x * blockDim.x) {
    const int32_t r = nodes[ent_row[q]], c = nodes[ent_col[q]];
    int32_t lo = krow[r], hi = krow[r + 1];          // first entry of the row with column >= c
    while (lo < hi) { const int32_t mid = (lo + hi) >> 1; if (kcol[mid] < c) lo = mid + 1; else hi = mid; }
    const bool found = lo < krow[r + 1] && kcol[lo] == c;
    if (!found) atomicOr(flag, 1);
    src[(int64_t)b * nent + q] = found ? lo : 0;
  }
}
void member_sources(int32_t nb, int32_t next, int32_t nent, const int32_t* ext, const int32_t* ent_row, const int32_t* ent_col,
                    const int32_t* krow, const int32_t* kcol, int32_t* src, int32_t* flag) {
  if (nb <= 0 || nent <= 0) return;
  for (int b0 = 0; b0 < nb; b0 += 65535) {
    const int n = std::min(65535, nb - b0);
    hipLaunchKernelGGL(k_member_sources, dim3(nblocks(nent, 256, 64), n), dim3(256), 0, g_stream, next, nent, ext + (int64_t)b0 * next, ent_row,
                       ent_col, krow, kcol, src + (int64_t)b0 * nent, flag);
    launch_check();
  }
}
__global__ void k_build_pull_tables(int64_t nrows, const int64_t* __restrict__ rcount, const int32_t* __restrict__ rowptr,
                                    const uint64_t* __restrict__ keys, int64_t* __restrict__ ptr, int64_t* __restrict__ idx) {
  for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < nrows; r += (int64_t)gridDim.x * blockDim.x) {
    int64_t e = (int64_t)rowptr[r] - 1;
    const int64_t k0 = rcount[r], k1 = rcount[r + 1];
    uint64_t prev = 0;
    for (int64_t k = k0; k < k1; k++) {
      const uint64_t key = keys[k], cg = key >> 33;
      if (k == k0 || cg != prev) { e++; prev = cg; }
      idx[k] = (int64_t)(key & (((uint64_t)1 << 33) - 1));
      ptr[e + 1] = k + 1;
    }
    if (r == 0) ptr[0] = 0;
  }
}
void build_pull_tables(int64_t nrows, const int64_t* rcount, const int32_t* rowptr, const uint64_t* keys, int64_t* ptr, int64_t* idx) {
  if (nrows <= 0) { zero(ptr, sizeof(int64_t)); return; }
  hipLaunchKernelGGL(k_build_pull_tables, dim3(nblocks(nrows, 64, 65536)), dim3(64), 0, g_stream, nrows, rcount, rowptr, keys, ptr, idx);
  launch_check();
}
__global__ void k_pull_sum_blocks(int64_t blen, const int64_t* __restrict__ ptr, const int64_t* __restrict__ base,
                                  const double* __restrict__ in, double* __restrict__ out) {
  const int B = blockIdx.y;
  const int64_t t0 = ptr[B], t1 = ptr[B + 1];
  for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < blen; k += (int64_t)gridDim.x * blockDim.x) {
    double s = 0.0;
    for (int64_t t = t0; t < t1; t++) s += in[base[t] + k];
    out[(int64_t)B * blen + k] = s;
  }
}
void pull_sum_blocks(int64_t blen, int32_t nblk, const int64_t* ptr, const int64_t* base, const double* in, double* out) {
  if (nblk <= 0 || blen <= 0) return;
  for (int b0 = 0; b0 < nblk; b0 += 65535) {
    const int nb = std::min(65535, nblk - b0);
    hipLaunchKernelGGL(k_pull_sum_blocks, dim3(nblocks(blen, 256, 64), nb), dim3(256), 0, g_stream, blen, ptr + b0, base, in,
                       out + (int64_t)b0 * blen);
    launch_check();
  }
}

// ------------------------------------------------------------------ multifrontal factorisation
constexpr int FT = 256;          // threads per front workgroup
// (i, j) of element t = tid, tid + step, ... of an n-row column-major grid without a division per element
struct Idx2 {
  int i, j, di, dj, n;
  __device__ Idx2(int t0, int nrows, int step) : i(t0 % nrows), j(t0 / nrows), di(step % nrows), dj(step / nrows), n(nrows) {}
  __device__ void next() { i += di; j += dj; if (i >= n) { i -= n; j++; } }
};
constexpr int GEMM_KB = 16;

// C(MxN) = beta*C - or + A(MxK) B(KxN), column-major, executed by the whole workgroup.
// 64x64 tile per pass, 4x4 micro-tile per thread, K staged through LDS in slabs of GEMM_KB.
template <bool SUBTRACT>
__device__ void wg_gemm(double* __restrict__ C, int64_t ldc, const double* __restrict__ A, int64_t lda,
                        const double* __restrict__ B, int64_t ldb, int M, int N, int K, double* lds) {
  double* As = lds;                    // [GEMM_KB][64]
  double* Bs = lds + GEMM_KB * 64;     // [GEMM_KB][64]
  const int tid = threadIdx.x;
  const int tx = tid & 15, ty = tid >> 4;  // row group tx (rows tx*4..), col group ty
  for (int tn = 0; tn < N; tn += 64)
    for (int tm = 0; tm < M; tm += 64) {
      double acc[4][4];
#pragma unroll
      for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) acc[a][b] = 0.0;
      for (int k0 = 0; k0 < K; k0 += GEMM_KB) {
        __syncthreads();
        for (int t = tid; t < GEMM_KB * 64; t += FT) {
          const int r = t & 63, kk = t >> 6;
          const int gm = tm + r, gk = k0 + kk;
          As[kk * 64 + r] = (gm < M && gk < K) ? A[gm + lda * gk] : 0.0;
        }
        for (int t = tid; t < GEMM_KB * 64; t += FT) {
          const int kk = t & (GEMM_KB - 1), r = t / GEMM_KB;
          const int gn = tn + r, gk = k0 + kk;
          Bs[kk * 64 + r] = (gn < N && gk < K) ? B[gk + ldb * gn] : 0.0;
        }
        __syncthreads();
#pragma unroll 2   // (fully unrolled this loop alone took k_factor_level to 248 VGPRs = ONE workgroup per CU; now 98: four)
        for (int kk = 0; kk < GEMM_KB; kk++) {
          double a[4], b[4];
#pragma unroll
          for (int q = 0; q < 4; q++) { a[q] = As[kk * 64 + tx * 4 + q]; b[q] = Bs[kk * 64 + ty * 4 + q]; }
#pragma unroll
          for (int p = 0; p < 4; p++)
#pragma unroll
            for (int q = 0; q < 4; q++) acc[p][q] += a[p] * b[q];
        }
      }
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int gn = tn + ty * 4 + q;
        if (gn >= N) continue;
#pragma unroll
        for (int p = 0; p < 4; p++) {
          const int gm = tm + tx * 4 + p;
          if (gm >= M) continue;
          if (SUBTRACT) C[gm + ldc * gn] -= acc[p][q];
          else C[gm + ldc * gn] = acc[p][q];
        }
      }
    }
  __syncthreads();
}

// element growth of an LU without pivoting (w x w block, column-major with leading dimension ld, factored in place):
// flags 2 in *flag when a multiplier or an entry of U exceeds GROWTH_LIMIT times the largest entry of the block
// before the factorisation -- the factor would be inaccurate without anybody noticing (high Reynolds numbers: the
// reference's KLU would pivot there).  m0: max |entry| before, filled by growth_before().
constexpr double GROWTH_LIMIT = 1e8;
__device__ inline void block_absmax(const double* S, int64_t ld, int w, bool lower, bool upper, unsigned long long* acc) {
  double m = 0.0;
  for (Idx2 q(threadIdx.x, w, blockDim.x); q.j < w; q.next())
    if ((q.i > q.j && lower) || (q.i <= q.j && upper)) m = fmax(m, fabs(S[q.i + ld * q.j]));
  atomicMax(acc, (unsigned long long)__double_as_longlong(m));
}

// LU without pivoting of the w x w block S (LDS, column-major, leading dimension w) followed by the in-place inverses of
// both factors: strictly lower part <- L^{-1} (unit diagonal implied), upper part incl. diagonal <- U^{-1}.  Executed by
// the whole workgroup with ONE barrier per elimination / inversion step (the multipliers are scaled after the loop; the
// column an inversion step needs is copied into the other half of xv during the step before).  xv: 2 w doubles.
// Element growth is measured on the way (see block_absmax); s_bad |= 1 for a zero / non-finite pivot, |= 2 for growth.
// LPR lanes share a row of an inversion step (2 for the 256-thread front kernels, 8 for the 1024-thread pivot pieces).
template <int LPR>
__device__ inline void lds_lu_and_inverses(double* S, int w, double* xv, int* s_bad, unsigned long long* s_m0,
                                           unsigned long long* s_ml, unsigned long long* s_mu) {
  const int tid = threadIdx.x, nt = blockDim.x;
  block_absmax(S, w, w, true, true, s_m0);
  __syncthreads();   // every wave has finished scanning S before the first elimination step writes into it
  for (int k = 0; k < w; k++) {
    const double piv = S[k + w * k];
    if (tid == 0 && (piv == 0.0 || !isfinite(piv))) *s_bad |= 1;
    const double ip = 1.0 / piv;
    // a wave takes 64 consecutive rows of one column (no bank conflicts, no integer division), the waves take the columns in turn
    const int ti = tid & 63, tj = tid >> 6, nw = nt >> 6;
    for (int j = k + 1 + tj; j < w; j += nw) {
      const double ukj = S[k + w * j];
      for (int i = k + 1 + ti; i < w; i += 64) S[i + w * j] -= (S[i + w * k] * ip) * ukj;
    }
    __syncthreads();
  }
  for (Idx2 q(tid, w, nt); q.j < w; q.next())
    if (q.i > q.j) S[q.i + w * q.j] *= 1.0 / S[q.j + w * q.j];
  __syncthreads();
  block_absmax(S, w, w, false, true, s_mu);
  __syncthreads();
  if (tid == 0) {
    // growth factor max |u_ij| / max |a_ij| of this block (the multipliers alone say nothing: in a saddle-point block
    // [a b; c 0] scaled like the reference's Stokes matrices they reach a / b^2 ~ 1e5 without any loss of accuracy)
    const double m0 = __longlong_as_double((long long)*s_m0), mu = __longlong_as_double((long long)*s_mu);
    const double rho = m0 > 0.0 ? mu / m0 : 0.0;
    if (rho > GROWTH_LIMIT) *s_bad |= 2;
    *s_ml = (unsigned long long)__double_as_longlong(rho);   // (handed to the caller: recorded next to the flag)
  }
  __syncthreads();
  // inverse of the unit lower factor, columns from the last to the first:
  // X[j+1:, j] = - X[j+1:, j+1:] * L[j+1:, j]   (X[j+1:, j+1:] already holds the inverse)
  double* xa = xv;
  double* xb = xv + w;
  if (w >= 2) for (int i = w - 1 + tid; i < w; i += nt) xa[i] = S[i + w * (w - 2)];
  __syncthreads();
  for (int j = w - 2; j >= 0; j--) {
    // two lanes per row split the dot product (both halves in four chains), combined by a lane shuffle
    for (int p = tid; p < LPR * (w - j - 1); p += nt) {
      const int i = j + 1 + p / LPR, h = p % LPR;
      const int len = i - (j + 1), kb = j + 1 + (len * h) / LPR, ke = j + 1 + (len * (h + 1)) / LPR;
      double s0 = h ? 0.0 : xa[i] /* unit diagonal of X */, s1 = 0.0, s2 = 0.0, s3 = 0.0;
      int k = kb;
      for (; k + 3 < ke; k += 4) {
        s0 += S[i + w * k] * xa[k]; s1 += S[i + w * (k + 1)] * xa[k + 1];
        s2 += S[i + w * (k + 2)] * xa[k + 2]; s3 += S[i + w * (k + 3)] * xa[k + 3];
      }
      for (; k < ke; k++) s0 += S[i + w * k] * xa[k];
      double sum = (s0 + s1) + (s2 + s3);
#pragma unroll
      for (int o = 1; o < LPR; o <<= 1) sum += __shfl_xor(sum, o, 64);
      if (!h) S[i + w * j] = -sum;
    }
    if (j > 0) for (int i = j + tid; i < w; i += nt) xb[i] = S[i + w * (j - 1)];   // the column of the next step
    __syncthreads();
    double* t = xa; xa = xb; xb = t;
  }
  // inverse of the upper factor, columns from the first to the last:
  // Y[j,j] = 1/U[j,j]; Y[0:j, j] = -Y[0:j, 0:j] * U[0:j, j] * Y[j,j]
  for (int i = tid; i < 1 && i < w; i += nt) xa[i] = S[i];
  __syncthreads();
  for (int j = 0; j < w; j++) {
    const double d = 1.0 / xa[j];
    for (int p = tid; p < LPR * j; p += nt) {
      const int i = p / LPR, h = p % LPR;
      const int len = j - i, kb = i + (len * h) / LPR, ke = i + (len * (h + 1)) / LPR;
      double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
      int k = kb;
      for (; k + 3 < ke; k += 4) {
        s0 += S[i + w * k] * xa[k]; s1 += S[i + w * (k + 1)] * xa[k + 1];
        s2 += S[i + w * (k + 2)] * xa[k + 2]; s3 += S[i + w * (k + 3)] * xa[k + 3];
      }
      for (; k < ke; k++) s0 += S[i + w * k] * xa[k];
      double sum = (s0 + s1) + (s2 + s3);
#pragma unroll
      for (int o = 1; o < LPR; o <<= 1) sum += __shfl_xor(sum, o, 64);
      if (!h) S[i + w * j] = -sum * d;
    }
    if (tid == 0) S[j + w * j] = d;
    if (j + 1 < w) for (int i = tid; i <= j + 1; i += nt) xb[i] = S[i + w * (j + 1)];
    __syncthreads();
    double* t = xa; xa = xb; xb = t;
  }
}

// LD: doubles of LDS (pivot block + work vector, then staging of the panel products, then the GEMM slabs: the three
// uses follow each other).  3072 (24 KiB, 6 workgroups per CU) for pivot blocks up to 54 wide, else 6144 (up to 77).
template <bool PROF, int LD>
__global__ void __launch_bounds__(FT) k_factor_level(PlanD P, BatchD B, const int32_t* __restrict__ list, int32_t b0,
                                                      const double* __restrict__ kval, unsigned long long* __restrict__ prof) {
  // PROF (HYMLS_MI_FACTOR_PROF): 100 MHz wall-clock ticks per phase, summed over the workgroups of the launch
  long long t_prev = 0;
  auto tick = [&](int phase) {
    if (PROF) { __syncthreads(); if (threadIdx.x == 0) { const long long t = wall_clock64(); atomicAdd(&prof[phase], (unsigned long long)(t - t_prev)); t_prev = t; } }
  };
  if (PROF && threadIdx.x == 0) t_prev = wall_clock64();
  __shared__ double lds[LD];
  __shared__ int s_bad;
  __shared__ unsigned long long s_m0, s_ml, s_mu;
  const int tid = threadIdx.x;
  const int slot = blockIdx.y;
  const int b = b0 + slot;
  const FrontD F = P.fronts[list[blockIdx.x]];
  const int w = F.w, ri = F.ri, rs = F.rs, m = w + ri + rs, r = ri + rs;
  double* sc = B.scratch + (int64_t)slot * P.scratch_size;
  double* A = sc + F.f_off;
  const int64_t mm = (int64_t)m * m;
  if (tid == 0) { s_bad = 0; s_m0 = 0; s_ml = 0; s_mu = 0; }
  // 1. zero, assemble matrix entries and the children's update matrices
  for (int64_t t = tid; t < mm; t += FT) A[t] = 0.0;
  __syncthreads();
  const int32_t* src = B.src + (int64_t)b * P.nent;
  for (int e = F.ent_begin + tid; e < F.ent_end; e += FT) A[P.ent_pos[e]] += P.ent_w[e] * kval[src[P.ent_id[e]]];
  __syncthreads();
  tick(0);
  for (int ce = F.child_begin; ce < F.child_end; ce++) {
    const FrontD Cf = P.fronts[P.children[ce]];
    const int mc = Cf.w + Cf.ri + Cf.rs, rc = Cf.ri + Cf.rs;
    const double* Ac = sc + Cf.f_off;
    const int32_t* rel = P.rel + Cf.rel_off;
    if (rc > 0)
      for (Idx2 q(tid, rc, FT); q.j < rc; q.next())
        A[rel[q.i] + (int64_t)m * rel[q.j]] += Ac[(Cf.w + q.i) + (int64_t)mc * (Cf.w + q.j)];
    __syncthreads();
  }
  tick(1);
  double* fac = B.factor + (int64_t)b * P.factor_size;
  double* Lp = fac + F.lp_off;
  double* Q = fac + F.q_off;
  const int64_t ld = w + ri;
  if (w * w + 2 * w <= LD) {
    // 2+3 (LDS path): LU of the pivot block and in-place triangular inverses inside LDS
    double* S = lds;
    double* xv = lds + w * w;
    for (Idx2 q(tid, w, FT); q.j < w; q.next()) S[q.i + w * q.j] = A[q.i + (int64_t)m * q.j];
    __syncthreads();
    lds_lu_and_inverses<2>(S, w, xv, &s_bad, &s_m0, &s_ml, &s_mu);
    for (Idx2 q(tid, w, FT); q.j < w; q.next()) Lp[q.i + ld * q.j] = S[q.i + w * q.j];
  } else {
    // 2. LU (no pivoting) of the w x w pivot block in global memory (wide pivot blocks)
    block_absmax(A, m, w, true, true, &s_m0);
    for (int k = 0; k < w; k++) {
      const double piv = A[k + (int64_t)m * k];
      if (tid == 0 && (piv == 0.0 || !isfinite(piv))) s_bad = 1;
      const double ip = 1.0 / piv;
      __syncthreads();
      for (int i = k + 1 + tid; i < w; i += FT) A[i + (int64_t)m * k] *= ip;
      __syncthreads();
      const int rem = w - k - 1;
      for (int t = tid; t < rem * rem; t += FT) {
        const int i = k + 1 + t % rem, j = k + 1 + t / rem;
        A[i + (int64_t)m * j] -= A[i + (int64_t)m * k] * A[k + (int64_t)m * j];
      }
      __syncthreads();
    }
    block_absmax(A, m, w, false, true, &s_mu);
    __syncthreads();
    if (tid == 0) {
      const double m0 = __longlong_as_double((long long)s_m0), mu = __longlong_as_double((long long)s_mu);
      const double rho = m0 > 0.0 ? mu / m0 : 0.0;
      if (rho > GROWTH_LIMIT) s_bad |= 2;
      s_ml = (unsigned long long)__double_as_longlong(rho);
    }
    __syncthreads();
    // 3. triangular inverses into the factor slab: strictly lower = L11^{-1}, upper = U11^{-1}
    for (int t = tid; t < w; t += FT) {
      for (int i = t + 1; i < w; i++) {
        double s = A[i + (int64_t)m * t];
        for (int j = t + 1; j < i; j++) s += A[i + (int64_t)m * j] * Lp[j + ld * t];
        Lp[i + ld * t] = -s;
      }
      Lp[t + ld * t] = 1.0 / A[t + (int64_t)m * t];
      for (int i = t - 1; i >= 0; i--) {
        double s = 0.0;
        for (int j = i + 1; j <= t; j++) s += A[i + (int64_t)m * j] * Lp[j + ld * t];
        Lp[i + ld * t] = -s / A[i + (int64_t)m * i];
      }
    }
  }
  if (tid == 0) {
    if (s_bad) atomicOr(B.flag, s_bad);
    atomicMax((unsigned long long*)(B.flag + 2), s_ml);   // largest growth factor seen by this batch
  }
  __threadfence_block();
  __syncthreads();
  tick(2);
  if (r > 0) {
    // 4. U12 = L11^{-1} F12 in place (column blocks staged in LDS)
    {
      const int cb = max(1, LD / w);
      for (int j0 = 0; j0 < r; j0 += cb) {
        const int nc = min(cb, r - j0);
        for (Idx2 q(tid, w, FT); q.j < nc; q.next()) lds[q.i + w * q.j] = A[q.i + (int64_t)m * (w + j0 + q.j)];
        __syncthreads();
        for (Idx2 q(tid, w, FT); q.j < nc; q.next()) {
          const int i = q.i, j = q.j;
          double s0 = lds[i + w * j], s1 = 0.0, s2 = 0.0, s3 = 0.0;
          int k = 0;
          for (; k + 3 < i; k += 4) {
            s0 += Lp[i + ld * k] * lds[k + w * j]; s1 += Lp[i + ld * (k + 1)] * lds[k + 1 + w * j];
            s2 += Lp[i + ld * (k + 2)] * lds[k + 2 + w * j]; s3 += Lp[i + ld * (k + 3)] * lds[k + 3 + w * j];
          }
          for (; k < i; k++) s0 += Lp[i + ld * k] * lds[k + w * j];
          A[i + (int64_t)m * (w + j0 + j)] = (s0 + s1) + (s2 + s3);
        }
        __syncthreads();
      }
    }
    tick(3);
    // 5. L21 = F21 U11^{-1} in place (row blocks staged in LDS)
    {
      const int rb = max(1, LD / w);
      for (int i0 = 0; i0 < r; i0 += rb) {
        const int nr = min(rb, r - i0);
        for (Idx2 q(tid, nr, FT); q.j < w; q.next()) lds[q.i + nr * q.j] = A[(w + i0 + q.i) + (int64_t)m * q.j];
        __syncthreads();
        for (Idx2 q(tid, nr, FT); q.j < w; q.next()) {
          const int i = q.i, j = q.j;
          double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
          int k = 0;
          for (; k + 3 <= j; k += 4) {
            s0 += lds[i + nr * k] * Lp[k + ld * j]; s1 += lds[i + nr * (k + 1)] * Lp[k + 1 + ld * j];
            s2 += lds[i + nr * (k + 2)] * Lp[k + 2 + ld * j]; s3 += lds[i + nr * (k + 3)] * Lp[k + 3 + ld * j];
          }
          for (; k <= j; k++) s0 += lds[i + nr * k] * Lp[k + ld * j];
          A[(w + i0 + i) + (int64_t)m * j] = (s0 + s1) + (s2 + s3);
        }
        __syncthreads();
      }
    }
    tick(4);
    // 6. Schur update F22 -= L21 U12
    wg_gemm<true>(A + w + (int64_t)m * w, m, A + w, m, A + (int64_t)m * w, m, r, r, w, lds);   // (2 x 16 x 64 doubles <= LD)
    tick(5);
    // 7. solve panels: PL = L21_int L11^{-1}, QU = U11^{-1} U12_int
    if (ri > 0) {
      for (Idx2 q(tid, ri, FT); q.j < w; q.next()) {
        const int i = q.i, k = q.j;
        double s0 = A[(w + i) + (int64_t)m * k] /* unit diagonal of L11^{-1} */, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        int j = k + 1;
        for (; j + 3 < w; j += 4) {
          s0 += A[(w + i) + (int64_t)m * j] * Lp[j + ld * k]; s1 += A[(w + i) + (int64_t)m * (j + 1)] * Lp[j + 1 + ld * k];
          s2 += A[(w + i) + (int64_t)m * (j + 2)] * Lp[j + 2 + ld * k]; s3 += A[(w + i) + (int64_t)m * (j + 3)] * Lp[j + 3 + ld * k];
        }
        for (; j < w; j++) s0 += A[(w + i) + (int64_t)m * j] * Lp[j + ld * k];
        Lp[(w + i) + ld * k] = (s0 + s1) + (s2 + s3);
      }
      for (Idx2 q(tid, w, FT); q.j < ri; q.next()) {
        const int i = q.i, j = q.j;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        int k = i;
        for (; k + 3 < w; k += 4) {
          s0 += Lp[i + ld * k] * A[k + (int64_t)m * (w + j)]; s1 += Lp[i + ld * (k + 1)] * A[k + 1 + (int64_t)m * (w + j)];
          s2 += Lp[i + ld * (k + 2)] * A[k + 2 + (int64_t)m * (w + j)]; s3 += Lp[i + ld * (k + 3)] * A[k + 3 + (int64_t)m * (w + j)];
        }
        for (; k < w; k++) s0 += Lp[i + ld * k] * A[k + (int64_t)m * (w + j)];
        Q[i + (int64_t)w * j] = (s0 + s1) + (s2 + s3);
      }
    }
    tick(6);
    // 8. (root fronts add their update to the separator block in root_update(), one launch per root in a fixed order:
    //     concurrent atomic adds of several roots made Compute irreproducible in the last bits)
  }
  tick(7);
}

void factor_level(const PlanD& P, const BatchD& B, const int32_t* list, int32_t count, int32_t b0, int32_t nbc,
                  const double* kval, int32_t max_w) {
  static const bool force_big = std::getenv("HYMLS_MI_FACTOR_LDS") && std::atoi(std::getenv("HYMLS_MI_FACTOR_LDS")) > 3072;   // (A/B switch)
  const bool small = !force_big && max_w * max_w + 2 * max_w <= 3072;
  if (count <= 0 || nbc <= 0) return;
  for (int s0 = 0; s0 < nbc; s0 += 65535) {
    const int ns = std::min(65535, nbc - s0);
    BatchD B2 = B;
    B2.scratch = B.scratch + (int64_t)s0 * P.scratch_size;
    if (B.sblock) B2.sblock = B.sblock + (int64_t)s0 * P.nS * P.nS;
    static const bool prof = std::getenv("HYMLS_MI_FACTOR_PROF") != nullptr;
    if (!prof) {
      if (small) hipLaunchKernelGGL((k_factor_level<false, 3072>), dim3(count, ns), dim3(FT), 0, g_stream, P, B2, list, b0 + s0, kval, (unsigned long long*)nullptr);
      else hipLaunchKernelGGL((k_factor_level<false, 6144>), dim3(count, ns), dim3(FT), 0, g_stream, P, B2, list, b0 + s0, kval, (unsigned long long*)nullptr);
      launch_check();
      continue;
    }
    // development aid: per-phase ticks of this launch on stderr (synchronises)
    unsigned long long* dprof = (unsigned long long*)alloc(8 * sizeof(unsigned long long));
    zero(dprof, 8 * sizeof(unsigned long long));
    timer_start(15);
    if (small) hipLaunchKernelGGL((k_factor_level<true, 3072>), dim3(count, ns), dim3(FT), 0, g_stream, P, B2, list, b0 + s0, kval, dprof);
    else hipLaunchKernelGGL((k_factor_level<true, 6144>), dim3(count, ns), dim3(FT), 0, g_stream, P, B2, list, b0 + s0, kval, dprof);
    launch_check();
    const double sec = timer_stop(15);
    unsigned long long h[8];
    d2h(h, dprof, sizeof h);
    free(dprof);
    double tot = 0;
    for (int q = 0; q < 8; q++) tot += (double)h[q];
    std::fprintf(stderr, "[hymls_mi] k_factor_level: %d fronts x %d members, %.3f ms; share of workgroup time: entries %.1f%% extend-add %.1f%% LU+inverses %.1f%% "
                         "U12 %.1f%% L21 %.1f%% Schur GEMM %.1f%% solve panels %.1f%% root update %.1f%%; mean workgroup %.1f us\n",
                 count, ns, 1e3 * sec, 100 * h[0] / tot, 100 * h[1] / tot, 100 * h[2] / tot, 100 * h[3] / tot, 100 * h[4] / tot, 100 * h[5] / tot,
                 100 * h[6] / tot, 100 * h[7] / tot, tot / 100.0 / ((double)count * ns));
  }
}

__global__ void k_sblock_entries(PlanD P, BatchD B, int32_t b0, const double* __restrict__ kval) {
  const int slot = blockIdx.y;
  const int32_t* src = B.src + (int64_t)(b0 + slot) * P.nent;
  double* S = B.sblock + (int64_t)slot * P.nS * P.nS;
  for (int e = P.s_ent_begin + blockIdx.x * blockDim.x + threadIdx.x; e < P.s_ent_end; e += gridDim.x * blockDim.x)
    S[P.ent_pos[e]] += P.ent_w[e] * kval[src[P.ent_id[e]]];  // positions are unique
}
void sblock_init(const PlanD& P, const BatchD& B, int32_t b0, int32_t nbc, const double* kval) {
  if (nbc <= 0 || P.nS == 0) return;
  zero(B.sblock, (size_t)nbc * P.nS * P.nS * sizeof(double));
  const int ne = P.s_ent_end - P.s_ent_begin;
  if (ne <= 0) return;
  for (int s0 = 0; s0 < nbc; s0 += 65535) {
    const int ns = std::min(65535, nbc - s0);
    BatchD B2 = B;
    B2.sblock = B.sblock + (int64_t)s0 * P.nS * P.nS;
    hipLaunchKernelGGL(k_sblock_entries, dim3(nblocks(ne, 256, 64), ns), dim3(256), 0, g_stream, P, B2, b0 + s0, kval);
    launch_check();
  }
}

// ------------------------------------------------------------------ big fronts (many workgroups per front)
// The same algebra as k_factor_level, cut into grid-wide steps: assemble, LU + inverses of the
// pivot block (one workgroup), panel products, Schur update on the FP64 matrix cores.
__global__ void k_big_zero(double* __restrict__ A, int64_t mm, int64_t stride) {
  double* p = A + (int64_t)blockIdx.y * stride;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < mm; t += (int64_t)gridDim.x * blockDim.x) p[t] = 0.0;
}
__global__ void k_big_entries(PlanD P, BatchD B, FrontD F, int32_t b0, const double* __restrict__ kval) {
  const int slot = blockIdx.y;
  double* A = B.scratch + (int64_t)slot * P.scratch_size + F.f_off;
  const int32_t* src = B.src + (int64_t)(b0 + slot) * P.nent;
  for (int e = F.ent_begin + blockIdx.x * blockDim.x + threadIdx.x; e < F.ent_end; e += gridDim.x * blockDim.x)
    A[P.ent_pos[e]] += P.ent_w[e] * kval[src[P.ent_id[e]]];
}
__global__ void k_big_extend_add(PlanD P, BatchD B, FrontD F, FrontD Cf) {
  const int slot = blockIdx.y;
  double* sc = B.scratch + (int64_t)slot * P.scratch_size;
  double* A = sc + F.f_off;
  const double* Ac = sc + Cf.f_off;
  const int m = F.w + F.ri + F.rs, mc = Cf.w + Cf.ri + Cf.rs, rc = Cf.ri + Cf.rs;
  const int32_t* rel = P.rel + Cf.rel_off;
  if (rc <= 0) return;
  for (Idx2 q(blockIdx.x * blockDim.x + threadIdx.x, rc, gridDim.x * blockDim.x); q.j < rc; q.next())   // (rc^2 and the stride fit an int)
    A[rel[q.i] + (int64_t)m * rel[q.j]] += Ac[(Cf.w + q.i) + (int64_t)mc * (Cf.w + q.j)];
}
constexpr int PIVOT_T = 1024;   // threads of a pivot-piece workgroup: the LDS block allows one workgroup per CU anyway
// LU (no pivoting) of one wk x wk pivot piece (wk <= PIECE = 128), entirely in LDS, followed by the
// in-place triangular inversions (unit-lower L and upper U share the block).  Results: packed into
// the supernode's slab block (strictly lower = L^{-1}, upper = U^{-1}) and as dense copies Lf
// (unit lower) / Uf (upper) for the panel products.  One workgroup per batch slot.
__global__ void __launch_bounds__(PIVOT_T) k_big_pivot(double* __restrict__ A0, int64_t ld, int64_t strideA, int wk,
                                                  double* __restrict__ slab0, int64_t lds, int64_t strideS,
                                                  double* __restrict__ tmp0, int64_t strideT, int32_t* flag) {
  extern __shared__ double S[];          // wk x wk block (column-major) + 2 wk work vector
  __shared__ int s_bad;
  __shared__ unsigned long long s_m0, s_ml, s_mu;
  const int tid = threadIdx.x, slot = blockIdx.x, w = wk;
  double* A = A0 + (int64_t)slot * strideA;
  double* Sb = slab0 + (int64_t)slot * strideS;
  double* Lf = tmp0 + (int64_t)slot * strideT;
  double* Uf = Lf + (int64_t)PIECE * PIECE;
  double* xv = S + w * w;
  if (tid == 0) { s_bad = 0; s_m0 = 0; s_ml = 0; s_mu = 0; }
  for (Idx2 q(tid, w, PIVOT_T); q.j < w; q.next()) S[q.i + w * q.j] = A[q.i + ld * q.j];
  __syncthreads();
  lds_lu_and_inverses<PIVOT_T / 128>(S, w, xv, &s_bad, &s_m0, &s_ml, &s_mu);
  if (tid == 0) {
    if (s_bad) atomicOr(flag, s_bad);
    atomicMax((unsigned long long*)(flag + 2), s_ml);
  }
  for (Idx2 q(tid, w, PIVOT_T); q.j < w; q.next()) {
    const int i = q.i, j = q.j;
    const double v = S[i + w * j];
    Sb[i + lds * j] = v;
    Lf[i + (int64_t)w * j] = i > j ? v : (i == j ? 1.0 : 0.0);
    Uf[i + (int64_t)w * j] = i <= j ? v : 0.0;
  }
}
// ---- the same pivot piece, blocked: 32 x 32 diagonal blocks are factored and inverted IN REGISTERS by one wave (a row per
// lane, rows broadcast with v_readlane: no LDS traffic, no barriers), everything else is 16 x 16 x 4 FP64 MFMA tile products
// between LDS operands.  The piece is padded with the identity to a multiple of 32.  Per diagonal block: LU (wave 0) |
// L^{-1} (wave 0) and U^{-1} (wave 1) | panels U12 = L11^{-1} A12, L21 = A21 U11^{-1} | trailing update; then the inverses of
// the block triangular factors from the last block column / row to the first through a 96 x 32 side buffer:
// X21 = -X22 (L21 L11^{-1}), Y12 = -(U11^{-1} U12) Y22.  About 25 barriers instead of 384.
typedef double d4 __attribute__((ext_vector_type(4)));
constexpr int PB = 32;
constexpr int PIVB_T = 512;
constexpr int PIVB_LT = 96;   // rows of the side buffer (PIECE - PB)
__device__ __forceinline__ double lane_bcast(double x, int src) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(x), src), hi = __builtin_amdgcn_readlane(__double2hiint(x), src);
  return __hiloint2double(hi, lo);
}
// one 16 x 16 tile sum_k fa(i, k) fb(k, j), k = kb .. ke-1 (multiple of 4 apart); lane l supplies fa(l & 15, k0 + (l >> 4)) and
// fb(k0 + (l >> 4), l & 15), result register r holds the entry (row (l >> 4) + 4 r, column l & 15)
template <class FA, class FB>
__device__ __forceinline__ d4 tile_mma(int kb, int ke, int lane, FA fa, FB fb) {
  d4 acc = (d4){0.0, 0.0, 0.0, 0.0};
  for (int k0 = kb; k0 < ke; k0 += 4) {
    const int k = k0 + (lane >> 4);
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(fa(lane & 15, k), fb(k, lane & 15), acc, 0, 0, 0);
  }
  return acc;
}
__global__ void __launch_bounds__(PIVB_T) k_big_pivot_blk(double* __restrict__ A0, int64_t ld, int64_t strideA, int wk,
                                                          double* __restrict__ slab0, int64_t lds, int64_t strideS,
                                                          double* __restrict__ tmp0, int64_t strideT, int32_t* flag) {
  extern __shared__ double S[];          // W x W block (column-major, identity padded) + PIVB_LT x PB side buffer
  __shared__ int s_bad;
  __shared__ unsigned long long s_m0, s_mu;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwv = PIVB_T / 64, slot = blockIdx.x, w = wk;
  const int W = (w + PB - 1) / PB * PB, nbk = W / PB;
  double* A = A0 + (int64_t)slot * strideA;
  double* Sb = slab0 + (int64_t)slot * strideS;
  double* Lf = tmp0 + (int64_t)slot * strideT;
  double* Uf = Lf + (int64_t)PIECE * PIECE;
  double* T = S + W * W;
  if (tid == 0) { s_bad = 0; s_m0 = 0; s_mu = 0; }
  for (Idx2 q(tid, W, PIVB_T); q.j < W; q.next()) S[q.i + W * q.j] = (q.i < w && q.j < w) ? A[q.i + ld * q.j] : (q.i == q.j ? 1.0 : 0.0);
  __syncthreads();
  block_absmax(S, W, w, true, true, &s_m0);
  __syncthreads();   // (the scan of S is complete before wave 0 overwrites the first diagonal block)
  double umax = 0.0;                     // largest |u_ij| this thread has produced
  const int r = lane & 31;
  for (int kblk = 0; kblk < nbk; kblk++) {
    const int k0 = PB * kblk, rem = W - k0 - PB, ns = rem / 16;
    double* D = S + k0 + W * k0;
    // LU of the diagonal block: row r in the registers of lane r (both halves of the wave hold the same rows)
    if (wave == 0) {
      double a[PB];
#pragma unroll
      for (int c = 0; c < PB; c++) a[c] = D[r + W * c];
#pragma unroll
      for (int t = 0; t < PB; t++) {
        const double piv = lane_bcast(a[t], t);
        if (lane == 0 && (piv == 0.0 || !isfinite(piv))) s_bad |= 1;
        const double l = a[t] * (1.0 / piv);
        const bool below = r > t;
        if (below) a[t] = l;
#pragma unroll
        for (int c = t + 1; c < PB; c++) { const double u = lane_bcast(a[c], t); if (below) a[c] -= l * u; }
      }
#pragma unroll
      for (int c = 0; c < PB; c++) { if (c >= r && k0 + c < w) umax = fmax(umax, fabs(a[c])); /* (not the identity padding) */ if (lane < PB) D[r + W * c] = a[c]; }
    }
    __syncthreads();
    if (wave == 0) {
      // X = L^{-1} (unit lower): rows of X in lanes, row t is final when its turn comes
      double a[PB], x[PB];
#pragma unroll
      for (int c = 0; c < PB; c++) { a[c] = c < r ? D[r + W * c] : 0.0; x[c] = c == r ? 1.0 : 0.0; }
#pragma unroll
      for (int t = 0; t < PB - 1; t++) {
        const bool below = r > t;
        const double l = a[t];
#pragma unroll
        for (int c = 0; c <= t; c++) { const double xt = lane_bcast(x[c], t); if (below) x[c] -= l * xt; }
      }
#pragma unroll
      for (int c = 0; c < PB; c++) if (lane < PB && c < r) D[r + W * c] = x[c];
    } else if (wave == 1) {
      // Y = U^{-1}: from the last row to the first
      double a[PB], y[PB];
#pragma unroll
      for (int c = 0; c < PB; c++) { a[c] = c >= r ? D[r + W * c] : 0.0; y[c] = c == r ? 1.0 : 0.0; }
#pragma unroll
      for (int t = PB - 1; t >= 0; t--) {
        const double d = 1.0 / lane_bcast(a[t], t);
        if (r == t) {
#pragma unroll
          for (int c = t; c < PB; c++) y[c] *= d;
        }
        const bool above = r < t;
        const double u = a[t];
#pragma unroll
        for (int c = t; c < PB; c++) { const double yt = lane_bcast(y[c], t); if (above) y[c] -= u * yt; }
      }
#pragma unroll
      for (int c = 0; c < PB; c++) if (lane < PB && c >= r) D[r + W * c] = y[c];
    }
    __syncthreads();
    if (rem > 0) {
      // panels, in place: a wave owns a strip of 16 columns of A12 (both row tiles) or 16 rows of A21 (both column tiles)
      for (int task = wave; task < 2 * ns; task += nwv) {
        if (task < ns) {
          const int jc = k0 + PB + 16 * task;
          auto fb = [&](int k, int j) { return S[k + W * (jc + j)]; };
          d4 acc[2];
#pragma unroll
          for (int h = 0; h < 2; h++) {
            const int ib = k0 + 16 * h;
            auto fa = [&](int i, int k) { const int gi = ib + i; const double v = S[gi + W * k]; return k > gi ? 0.0 : (k == gi ? 1.0 : v); };
            acc[h] = tile_mma(k0, ib + 16, lane, fa, fb);
          }
#pragma unroll
          for (int h = 0; h < 2; h++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
              S[(k0 + 16 * h + (lane >> 4) + 4 * q) + W * (jc + (lane & 15))] = acc[h][q];
              umax = fmax(umax, fabs(acc[h][q]));
            }
        } else {
          const int ir = k0 + PB + 16 * (task - ns);
          auto fa = [&](int i, int k) { return S[(ir + i) + W * k]; };
          d4 acc[2];
#pragma unroll
          for (int h = 0; h < 2; h++) {
            const int jb = k0 + 16 * h;
            auto fb = [&](int k, int j) { const int gj = jb + j; return k > gj ? 0.0 : S[k + W * gj]; };
            acc[h] = tile_mma(k0, jb + 16, lane, fa, fb);
          }
#pragma unroll
          for (int h = 0; h < 2; h++)
#pragma unroll
            for (int q = 0; q < 4; q++) S[(ir + (lane >> 4) + 4 * q) + W * (k0 + 16 * h + (lane & 15))] = acc[h][q];
        }
      }
      __syncthreads();
      // trailing update A22 -= L21 U12
      for (int task = wave; task < ns * ns; task += nwv) {
        const int i0 = k0 + PB + 16 * (task % ns), j0 = k0 + PB + 16 * (task / ns);
        auto fa = [&](int i, int k) { return S[(i0 + i) + W * k]; };
        auto fb = [&](int k, int j) { return S[k + W * (j0 + j)]; };
        const d4 acc = tile_mma(k0, k0 + PB, lane, fa, fb);
#pragma unroll
        for (int q = 0; q < 4; q++) S[(i0 + (lane >> 4) + 4 * q) + W * (j0 + (lane & 15))] -= acc[q];
      }
      __syncthreads();
    }
  }
  // growth factor max |u_ij| / max |a_ij| of the piece (see lds_lu_and_inverses)
  atomicMax(&s_mu, (unsigned long long)__double_as_longlong(umax));
  __syncthreads();
  if (tid == 0) {
    const double m0 = __longlong_as_double((long long)s_m0), mu = __longlong_as_double((long long)s_mu);
    const double rho = m0 > 0.0 ? mu / m0 : 0.0;
    if (rho > GROWTH_LIMIT) s_bad |= 2;
    if (s_bad) atomicOr(flag, s_bad);
    atomicMax((unsigned long long*)(flag + 2), (unsigned long long)__double_as_longlong(rho));
  }
  // inverse of the block lower triangular L: block columns from the last but one to the first
  for (int jb = nbk - 2; jb >= 0; jb--) {
    const int c0 = PB * jb, r0 = c0 + PB, rem = W - r0, ns = rem / 16;
    for (int task = wave; task < 2 * ns; task += nwv) {          // T = L21 L11^{-1}
      const int i0 = r0 + 16 * (task % ns), tc = task / ns, jc = c0 + 16 * tc;
      auto fa = [&](int i, int k) { return S[(i0 + i) + W * k]; };
      auto fb = [&](int k, int j) { const int gj = jc + j; const double v = S[k + W * gj]; return k < gj ? 0.0 : (k == gj ? 1.0 : v); };
      const d4 acc = tile_mma(jc, c0 + PB, lane, fa, fb);
#pragma unroll
      for (int q = 0; q < 4; q++) T[(i0 - r0 + (lane >> 4) + 4 * q) + PIVB_LT * (16 * tc + (lane & 15))] = acc[q];
    }
    __syncthreads();
    for (int task = wave; task < 2 * ns; task += nwv) {          // X21 = -X22 T
      const int i0 = r0 + 16 * (task % ns), tc = task / ns;
      auto fa = [&](int i, int k) { const int gi = i0 + i; const double v = S[gi + W * k]; return k > gi ? 0.0 : (k == gi ? 1.0 : v); };
      auto fb = [&](int k, int j) { return T[(k - r0) + PIVB_LT * (16 * tc + j)]; };
      const d4 acc = tile_mma(r0, i0 + 16, lane, fa, fb);
#pragma unroll
      for (int q = 0; q < 4; q++) S[(i0 + (lane >> 4) + 4 * q) + W * (c0 + 16 * tc + (lane & 15))] = -acc[q];
    }
    __syncthreads();
  }
  // inverse of the block upper triangular U: block rows from the last but one to the first (side buffer 32 x rem, ld 32)
  for (int jb = nbk - 2; jb >= 0; jb--) {
    const int r0 = PB * jb, c0 = r0 + PB, rem = W - c0, ns = rem / 16;
    for (int task = wave; task < 2 * ns; task += nwv) {          // T = U11^{-1} U12
      const int tr = task % 2, j0 = c0 + 16 * (task / 2), ib = r0 + 16 * tr;
      auto fa = [&](int i, int k) { const int gi = ib + i; return k < gi ? 0.0 : S[gi + W * k]; };
      auto fb = [&](int k, int j) { return S[k + W * (j0 + j)]; };
      const d4 acc = tile_mma(ib, r0 + PB, lane, fa, fb);
#pragma unroll
      for (int q = 0; q < 4; q++) T[(16 * tr + (lane >> 4) + 4 * q) + PB * (j0 - c0 + (lane & 15))] = acc[q];
    }
    __syncthreads();
    for (int task = wave; task < 2 * ns; task += nwv) {          // Y12 = -T Y22
      const int tr = task % 2, j0 = c0 + 16 * (task / 2);
      auto fa = [&](int i, int k) { return T[(16 * tr + i) + PB * (k - c0)]; };
      auto fb = [&](int k, int j) { const int gj = j0 + j; return k > gj ? 0.0 : S[k + W * gj]; };
      const d4 acc = tile_mma(c0, j0 + 16, lane, fa, fb);
#pragma unroll
      for (int q = 0; q < 4; q++) S[(r0 + 16 * tr + (lane >> 4) + 4 * q) + W * (j0 + (lane & 15))] = -acc[q];
    }
    __syncthreads();
  }
  for (Idx2 q(tid, w, PIVB_T); q.j < w; q.next()) {
    const int i = q.i, j = q.j;
    const double v = S[i + W * j];
    Sb[i + lds * j] = v;
    Lf[i + (int64_t)w * j] = i > j ? v : (i == j ? 1.0 : 0.0);
    Uf[i + (int64_t)w * j] = i <= j ? v : 0.0;
  }
}
// row panel right of a pivot piece: U12 = L^{-1} F12 in place, one workgroup per 8 columns
__global__ void __launch_bounds__(256) k_big_trmm_u(double* __restrict__ A0, int64_t ld, int64_t strideA, int wk, int rk,
                                                    const double* __restrict__ tmp0, int64_t strideT) {
  extern __shared__ double st[];
  const int slot = blockIdx.y, tid = threadIdx.x, w = wk;
  double* A = A0 + (int64_t)slot * strideA;                 // points at the piece's diagonal block
  const double* Lf = tmp0 + (int64_t)slot * strideT;
  const int j0 = blockIdx.x * 8, nc = min(8, rk - j0);
  for (Idx2 q(tid, w, 256); q.j < nc; q.next()) st[q.i + w * q.j] = A[q.i + ld * (w + j0 + q.j)];
  __syncthreads();
  for (Idx2 q(tid, w, 256); q.j < nc; q.next()) {
    const int i = q.i, j = q.j;
    double s0 = st[i + w * j], s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int k = 0;
    for (; k + 3 < i; k += 4) {
      s0 += Lf[i + (int64_t)w * k] * st[k + w * j]; s1 += Lf[i + (int64_t)w * (k + 1)] * st[k + 1 + w * j];
      s2 += Lf[i + (int64_t)w * (k + 2)] * st[k + 2 + w * j]; s3 += Lf[i + (int64_t)w * (k + 3)] * st[k + 3 + w * j];
    }
    for (; k < i; k++) s0 += Lf[i + (int64_t)w * k] * st[k + w * j];
    A[i + ld * (w + j0 + j)] = (s0 + s1) + (s2 + s3);
  }
}
// column panel below a pivot piece: L21 = F21 U^{-1} in place, one workgroup per 8 rows
__global__ void __launch_bounds__(256) k_big_trmm_l(double* __restrict__ A0, int64_t ld, int64_t strideA, int wk, int rk,
                                                    const double* __restrict__ tmp0, int64_t strideT) {
  extern __shared__ double st[];
  const int slot = blockIdx.y, tid = threadIdx.x, w = wk;
  double* A = A0 + (int64_t)slot * strideA;
  const double* Uf = tmp0 + (int64_t)slot * strideT + (int64_t)PIECE * PIECE;
  const int i0 = blockIdx.x * 8, nr = min(8, rk - i0);
  for (Idx2 q(tid, nr, 256); q.j < w; q.next()) st[q.i + nr * q.j] = A[(w + i0 + q.i) + ld * q.j];
  __syncthreads();
  for (Idx2 q(tid, nr, 256); q.j < w; q.next()) {
    const int i = q.i, j = q.j;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int k = 0;
    for (; k + 3 <= j; k += 4) {
      s0 += st[i + nr * k] * Uf[k + (int64_t)w * j]; s1 += st[i + nr * (k + 1)] * Uf[k + 1 + (int64_t)w * j];
      s2 += st[i + nr * (k + 2)] * Uf[k + 2 + (int64_t)w * j]; s3 += st[i + nr * (k + 3)] * Uf[k + 3 + (int64_t)w * j];
    }
    for (; k <= j; k++) s0 += st[i + nr * k] * Uf[k + (int64_t)w * j];
    A[(w + i0 + i) + ld * j] = (s0 + s1) + (s2 + s3);
  }
}

// C (MxN) = A B | C - A B | -(A B) on the FP64 matrix cores (v_mfma_f64_16x16x4_f64), column-major.
// 64x64 tile per workgroup, each of the 4 waves owns a 32x32 quadrant = 2x2 MFMA tiles; slabs of
// 16 in K are staged through LDS k-major.  Lane l holds A[i = l & 15][k = l >> 4] and
// B[k = l >> 4][j = l & 15]; result register r holds D[row = (l >> 4) + 4 r][col = l & 15]
// (f64 layout, cdna_hip_programming.md section 3).  MA / MB mask a triangular operand that is
// stored packed with its sibling triangle: 1 = unit lower (above diagonal 0, diagonal 1),
// 2 = upper (below diagonal 0).
template <int MODE, int MA, int MB, int VAR = 0>
__global__ void __launch_bounds__(256) k_gemm_f64(double* C /* may alias an operand: panel_trmm */, int64_t ldc, int64_t strideC,
                                                   const double* A, int64_t lda, int64_t strideA,
                                                   const double* Bm, int64_t ldb, int64_t strideB,
                                                   int M, int N, int K) {
  __shared__ double As[16 * 64];
  __shared__ double Bs[16 * 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = (wave & 1) * 32, wn = (wave >> 1) * 32;
  const int tm = blockIdx.x * 64, tn = blockIdx.y * 64;
  C += (int64_t)blockIdx.z * strideC; A += (int64_t)blockIdx.z * strideA; Bm += (int64_t)blockIdx.z * strideB;
  d4 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int b = 0; b < 2; b++) acc[a][b] = (d4){0.0, 0.0, 0.0, 0.0};
  // triangular operands: skip the K slabs that are entirely zero for this tile
  int kbeg = 0, kend = K;
  if (MA == 1) kend = min(K, tm + 64);          // A lower: columns beyond the tile's last row are zero
  if (MA == 2) kbeg = max(0, (tm / 16) * 16);   // A upper: columns before the tile's first row are zero
  if (MB == 1) kbeg = max(kbeg, (tn / 16) * 16);  // B lower: rows above the tile's first column are zero
  if (MB == 2) kend = min(kend, tn + 64);         // B upper: rows below the tile's last column are zero
  for (int k0 = kbeg; k0 < kend; k0 += 16) {
    __syncthreads();
    for (int t = tid; t < 16 * 64; t += 256) {
      const int r = t & 63, kk = t >> 6;
      const int gm = tm + r, gk = k0 + kk;
      double v = (gm < M && gk < K) ? A[gm + lda * gk] : 0.0;
      if (MA == 1) v = gk > gm ? 0.0 : (gk == gm ? 1.0 : v);
      if (MA == 2) v = gk < gm ? 0.0 : v;
      As[kk * 64 + r] = v;
    }
    for (int t = tid; t < 16 * 64; t += 256) {
      const int kk = t & 15, c = t >> 4;
      const int gn = tn + c, gk = k0 + kk;
      double v = (gn < N && gk < K) ? Bm[gk + ldb * gn] : 0.0;
      if (MB == 1) v = gn > gk ? 0.0 : (gn == gk ? 1.0 : v);
      if (MB == 2) v = gn < gk ? 0.0 : v;
      Bs[kk * 64 + c] = v;
    }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < 16; ks += 4) {
      const int kk = ks + (lane >> 4);
      double a[2], b[2];
#pragma unroll
      for (int q = 0; q < 2; q++) {
        a[q] = As[kk * 64 + wm + q * 16 + (lane & 15)];
        b[q] = Bs[kk * 64 + wn + q * 16 + (lane & 15)];
      }
#pragma unroll
      for (int p = 0; p < 2; p++)
#pragma unroll
        for (int q = 0; q < 2; q++)
          acc[p][q] = VAR == 0 ? __builtin_amdgcn_mfma_f64_16x16x4f64(a[p], b[q], acc[p][q], 0, 0, 0)
                               : __builtin_amdgcn_mfma_f64_16x16x4f64(b[q], a[p], acc[p][q], 0, 0, 0);   // (transposed tile, see below)
    }
  }
#pragma unroll
  for (int p = 0; p < 2; p++)
#pragma unroll
    for (int q = 0; q < 2; q++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int gm = tm + wm + p * 16 + (VAR == 0 ? (lane >> 4) + 4 * r : (lane & 15));
        const int gn = tn + wn + q * 16 + (VAR == 0 ? (lane & 15) : (lane >> 4) + 4 * r);
        if (gm < M && gn < N) {
          if (MODE == 1) C[gm + ldc * gn] -= acc[p][q][r];   // (requesting the C tile before the K loop changed nothing: measured)
          else if (MODE == 2) C[gm + ldc * gn] = -acc[p][q][r];
          else C[gm + ldc * gn] = acc[p][q][r];
        }
      }
}
// The same product with a 128 x 128 tile per workgroup: every wave owns a 64 x 64 quadrant = 4 x 4 MFMA tiles, so one K step
// of four costs 8 LDS reads per lane for 16 matrix instructions (the 64 x 64 kernel above: 4 reads for 4 -- it is bound by
// LDS bandwidth at half the matrix rate), and the next K slab travels from global memory to registers while the current one
// is multiplied.  Rows of the LDS slabs are 144 doubles apart: the four groups of 16 lanes of a read (four K values) then
// fall on two disjoint halves of the banks.  Every element of C receives exactly the sequence of matrix instructions the
// 64 x 64 kernel gives it (K slabs of 16 in ascending order, four K values per instruction): the results are the same bits.
constexpr int GB_LD = 144;
// VAR (development switch, tools/gemm_check.hip): 0 = C rows on lane >> 4 (a store touches 16 columns x 32 bytes),
// 1 = operands exchanged in the matrix instruction, which computes the transposed tile: lane & 15 runs along the rows of C,
// a load / store of C touches 4 columns x 128 contiguous bytes; 2 = 1 + the B slab fetched with K fastest (128 contiguous
// bytes per column).  The sums are the same in all three (a b = b a exactly, same K order).
template <int MODE, int MA, int MB, int VAR = 0>
__global__ void __launch_bounds__(256, 2) k_gemm_f64_big(double* C /* may alias an operand: panel_trmm */, int64_t ldc, int64_t strideC,
                                                       const double* A, int64_t lda, int64_t strideA,
                                                       const double* Bm, int64_t ldb, int64_t strideB,
                                                       int M, int N, int K) {
  __shared__ double As[16 * GB_LD];
  __shared__ double Bs[16 * GB_LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = (wave & 1) * 64, wn = (wave >> 1) * 64;
  const int tm = blockIdx.x * 128, tn = blockIdx.y * 128;
  C += (int64_t)blockIdx.z * strideC; A += (int64_t)blockIdx.z * strideA; Bm += (int64_t)blockIdx.z * strideB;
  d4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; a++)
#pragma unroll
    for (int b = 0; b < 4; b++) acc[a][b] = (d4){0.0, 0.0, 0.0, 0.0};
  int kbeg = 0, kend = K;
  if (MA == 1) kend = min(K, tm + 128);
  if (MA == 2) kbeg = max(0, (tm / 16) * 16);
  if (MB == 1) kbeg = max(kbeg, (tn / 16) * 16);
  if (MB == 2) kend = min(kend, tn + 128);
  // this thread's share of a slab: A rows r = tid & 127 at k = (tid >> 7) + 2 t; B columns c = tid >> 1 at k = (tid & 1) * 8 + t
  // (VAR 2: B at k = tid & 15, columns c = (tid >> 4) + 16 t)
  const int ar = tid & 127, ak = tid >> 7;
  const int bc = VAR == 2 ? (tid >> 4) : (tid >> 1), bk = VAR == 2 ? (tid & 15) : (tid & 1) * 8;
  double ra[8], rb[8];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int t = 0; t < 8; t++) {
      const int gm = tm + ar, gk = k0 + ak + 2 * t;
      double v = (gm < M && gk < K) ? A[gm + lda * gk] : 0.0;
      if (MA == 1) v = gk > gm ? 0.0 : (gk == gm ? 1.0 : v);
      if (MA == 2) v = gk < gm ? 0.0 : v;
      ra[t] = v;
    }
#pragma unroll
    for (int t = 0; t < 8; t++) {
      const int gn = tn + bc + (VAR == 2 ? 16 * t : 0), gk = k0 + bk + (VAR == 2 ? 0 : t);
      double v = (gn < N && gk < K) ? Bm[gk + ldb * gn] : 0.0;
      if (MB == 1) v = gn > gk ? 0.0 : (gn == gk ? 1.0 : v);
      if (MB == 2) v = gn < gk ? 0.0 : v;
      rb[t] = v;
    }
  };
  if (kbeg < kend) fetch(kbeg);
  for (int k0 = kbeg; k0 < kend; k0 += 16) {
    __syncthreads();                                     // the previous slab has been read by every wave
#pragma unroll
    for (int t = 0; t < 8; t++) {
      As[(ak + 2 * t) * GB_LD + ar] = ra[t];
      if (VAR == 2) Bs[(bc + 16 * t) * 17 + bk] = rb[t];   // (column-major slab, 17 doubles apart: writes and reads without conflicts)
      else Bs[(bk + t) * GB_LD + bc] = rb[t];
    }
    __syncthreads();
    if (k0 + 16 < kend) fetch(k0 + 16);                  // in flight during the products below
#pragma unroll
    for (int ks = 0; ks < 16; ks += 4) {
      const int kk = ks + (lane >> 4);
      double a[4], b[4];
#pragma unroll
      for (int q = 0; q < 4; q++) {
        a[q] = As[kk * GB_LD + wm + q * 16 + (lane & 15)];
        b[q] = VAR == 2 ? Bs[(wn + q * 16 + (lane & 15)) * 17 + kk] : Bs[kk * GB_LD + wn + q * 16 + (lane & 15)];
      }
#pragma unroll
      for (int p = 0; p < 4; p++)
#pragma unroll
        for (int q = 0; q < 4; q++)
          acc[p][q] = VAR == 0 ? __builtin_amdgcn_mfma_f64_16x16x4f64(a[p], b[q], acc[p][q], 0, 0, 0)
                               : __builtin_amdgcn_mfma_f64_16x16x4f64(b[q], a[p], acc[p][q], 0, 0, 0);
    }
  }
#pragma unroll
  for (int p = 0; p < 4; p++)
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int gm = tm + wm + p * 16 + (VAR == 0 ? (lane >> 4) + 4 * r : (lane & 15));
        const int gn = tn + wn + q * 16 + (VAR == 0 ? (lane & 15) : (lane >> 4) + 4 * r);
        if (gm < M && gn < N) {
          if (MODE == 1) C[gm + ldc * gn] -= acc[p][q][r];
          else if (MODE == 2) C[gm + ldc * gn] = -acc[p][q][r];
          else C[gm + ldc * gn] = acc[p][q][r];
        }
      }
}
template <int MODE, int MA, int MB>
static void gemm_f64_launch(double* C, int64_t ldc, int64_t sC, const double* A, int64_t lda, int64_t sA, const double* Bm, int64_t ldb,
                            int64_t sB, int M, int N, int K, int batch);
template <int MODE, int MA, int MB>
static void gemm_f64(double* C, int64_t ldc, int64_t sC, const double* A, int64_t lda, int64_t sA, const double* Bm, int64_t ldb,
                     int64_t sB, int M, int N, int K, int batch) {
  if (M <= 0 || N <= 0 || batch <= 0) return;
  SetupProf::Scope prof("gemm", MODE * 100 + MA * 10 + MB, M, N, K, batch, 2.0 * M * N * K * batch);
  gemm_f64_launch<MODE, MA, MB>(C, ldc, sC, A, lda, sA, Bm, ldb, sB, M, N, K, batch);
}
template <int MODE, int MA, int MB>
static void gemm_f64_launch(double* C, int64_t ldc, int64_t sC, const double* A, int64_t lda, int64_t sA, const double* Bm, int64_t ldb,
                            int64_t sB, int M, int N, int K, int batch) {
  // Tile choice, measured on the setup's shapes (tools/gemm_check.hip, profiles/r03_v_gemm_variants.txt): the 128 x 128 tiles win
  // from K = 256 on, and from K = 64 on when the launch fills the chip; thin updates (K = 32 .. 49 of the level-0 root fronts) are bound by the
  // read-modify-write of C and run faster with four times as many, smaller workgroups.  Every variant gives the same bits.
  // (HYMLS_MI_GEMM_TILE=64 / 128 forces one kernel: A/B measurements)
  static const int forced = std::getenv("HYMLS_MI_GEMM_TILE") ? std::atoi(std::getenv("HYMLS_MI_GEMM_TILE")) : 0;
  const int64_t tiles = (int64_t)((M + 127) / 128) * ((N + 127) / 128) * batch;
  const bool big = forced == 64 ? false : forced == 128 ? (M > 96 && N > 96) : (M > 96 && N > 96 && (K >= 256 || (K >= 64 && tiles >= 768)));
  if (big) {
    hipLaunchKernelGGL((k_gemm_f64_big<MODE, MA, MB, 2>), dim3((M + 127) / 128, (N + 127) / 128, batch), dim3(256), 0, g_stream, C, ldc,
                       sC, A, lda, sA, Bm, ldb, sB, M, N, K);
  } else {
    hipLaunchKernelGGL((k_gemm_f64<MODE, MA, MB, 1>), dim3((M + 63) / 64, (N + 63) / 64, batch), dim3(256), 0, g_stream, C, ldc, sC, A,
                       lda, sA, Bm, ldb, sB, M, N, K);
  }
  launch_check();
}
// The panels of a pivot piece on the matrix cores, IN PLACE: U12 = L^{-1} F12 (wk x rk, right of the piece) and L21 = F21 U^{-1}
// (rk x wk, below it), with the piece's explicit triangular inverses from the pivot kernel.  In place is safe because one
// workgroup owns everything it reads of the panel: its tile spans all wk <= 128 rows (columns) of the panel, the K loop has
// taken the whole operand in before the first store, and no other workgroup reads that tile.  (Before: k_big_trmm_u / _l,
// one thread per entry with a dot product from global memory, 1.8 TFLOP/s, 11 % of the recompute's kernel time.)
static void panel_trmm(double* Ak, int64_t ld, int64_t sA, const double* tk, int64_t sT, int wk, int rk, int nbc) {
  const double* Lf = tk;
  const double* Uf = tk + (int64_t)PIECE * PIECE;
  double* U12 = Ak + ld * wk;
  double* L21 = Ak + wk;
  if (wk <= 64) {
    hipLaunchKernelGGL((k_gemm_f64<0, 0, 0, 1>), dim3(1, (rk + 63) / 64, nbc), dim3(256), 0, g_stream, U12, ld, sA, Lf, (int64_t)wk, sT, U12, ld, sA, wk, rk, wk);
    launch_check();
    hipLaunchKernelGGL((k_gemm_f64<0, 0, 0, 1>), dim3((rk + 63) / 64, 1, nbc), dim3(256), 0, g_stream, L21, ld, sA, L21, ld, sA, Uf, (int64_t)wk, sT, rk, wk, wk);
    launch_check();
  } else {
    hipLaunchKernelGGL((k_gemm_f64_big<0, 0, 0, 2>), dim3(1, (rk + 127) / 128, nbc), dim3(256), 0, g_stream, U12, ld, sA, Lf, (int64_t)wk, sT, U12, ld, sA, wk, rk, wk);
    launch_check();
    hipLaunchKernelGGL((k_gemm_f64_big<0, 0, 0, 2>), dim3((rk + 127) / 128, 1, nbc), dim3(256), 0, g_stream, L21, ld, sA, L21, ld, sA, Uf, (int64_t)wk, sT, rk, wk, wk);
    launch_check();
  }
}
__global__ void k_big_root_update(PlanD P, BatchD B, FrontD F) {
  const int slot = blockIdx.y;
  const int w = F.w, ri = F.ri, rs = F.rs, m = w + ri + rs;
  const double* A = B.scratch + (int64_t)slot * P.scratch_size + F.f_off;
  double* S = B.sblock + (int64_t)slot * P.nS * P.nS;
  const int32_t* rel = P.rel + F.rel_off;
  if (rs <= 0) return;
  for (Idx2 q(blockIdx.x * blockDim.x + threadIdx.x, rs, gridDim.x * blockDim.x); q.j < rs; q.next())
    S[rel[ri + q.i] + (int64_t)P.nS * rel[ri + q.j]] += A[(w + ri + q.i) + (int64_t)m * (w + ri + q.j)];   // (targets are unique within a root)
}
void root_update(const PlanD& P, const BatchD& B, const FrontD& F, int32_t nbc) {
  if (nbc <= 0 || F.rs <= 0) return;
  for (int s0 = 0; s0 < nbc; s0 += 65535) {
    const int ns = std::min(65535, nbc - s0);
    BatchD B2 = B;
    B2.scratch = B.scratch + (int64_t)s0 * P.scratch_size;
    B2.sblock = B.sblock + (int64_t)s0 * P.nS * P.nS;
    hipLaunchKernelGGL(k_big_root_update, dim3(nblocks((int64_t)F.rs * F.rs, 256, 8192), ns), dim3(256), 0, g_stream, P, B2, F);
    launch_check();
  }
}

void factor_big_front(const PlanD& P, const BatchD& B, const FrontD& F, const FrontD* kids, int32_t nkids, int32_t b0,
                      int32_t nbc, const double* kval) {
  if (nbc <= 0) return;
  if (!ctx().big_attr_set) {
    HIP_CHECK(hipFuncSetAttribute((const void*)k_big_pivot, hipFuncAttributeMaxDynamicSharedMemorySize, (int)((PIECE * PIECE + 2 * PIECE) * sizeof(double))));
    HIP_CHECK(hipFuncSetAttribute((const void*)k_big_pivot_blk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)((PIECE * PIECE + PIVB_LT * PB) * sizeof(double))));
    ctx().big_attr_set = true;
  }
  if (nbc > 65535) throw Error(-3, "too many batch members for the big-front path");
  const int w = F.w, ri = F.ri, rs = F.rs, m = w + ri + rs;
  const int64_t ld = m, mm = (int64_t)m * m, sA = P.scratch_size, sS = P.factor_size, sT = B.tmp_stride;
  const int64_t lds = w + ri;
  double* A0 = B.scratch + F.f_off;
  double* slab = B.factor + (int64_t)b0 * P.factor_size + F.lp_off;
  double* Qs = B.factor + (int64_t)b0 * P.factor_size + F.q_off;
  { SetupProf::Scope prof("zero", m, 0, 0, 0, nbc);
  hipLaunchKernelGGL(k_big_zero, dim3(nblocks(mm, 256, 8192), nbc), dim3(256), 0, g_stream, A0, mm, P.scratch_size); launch_check(); }
  if (F.ent_end > F.ent_begin) {
    SetupProf::Scope prof("entries", m, 0, 0, 0, nbc);
    hipLaunchKernelGGL(k_big_entries, dim3(nblocks(F.ent_end - F.ent_begin, 256, 256), nbc), dim3(256), 0, g_stream, P, B, F, b0, kval);
    launch_check();
  }
  for (int c = 0; c < nkids; c++) {
    const int64_t rc = kids[c].ri + kids[c].rs;
    if (rc <= 0) continue;
    SetupProf::Scope prof("extend_add", m, (int)rc, 0, 0, nbc);
    hipLaunchKernelGGL(k_big_extend_add, dim3(nblocks(rc * rc, 256, 8192), nbc), dim3(256), 0, g_stream, P, B, F, kids[c]);
    launch_check();
  }
  // ---- blocked LU of the pivot block, in place, pieces of PIECE columns inside outer blocks of OUTER columns.  Inside an
  // outer block the pieces are updated lazily (a piece's column and row panels receive the products of the block's earlier
  // pieces just before the piece is factored), and everything beyond the outer block receives ONE update of rank OUTER:
  // the read-modify-write of the trailing matrix, which bounds a rank-128 update at about a quarter of the matrix peak,
  // is paid once per 512 columns (tools/gemm_check.hip: 3793 x 3793, rank 128 / 256 / 512 = 24 / 34 / 43 TFLOP/s).
  // HYMLS_MI_OUTER_BLOCK=128 gives the plain right-looking order (A/B measurements).
  static const int OUTER = [] {
    const char* e = std::getenv("HYMLS_MI_OUTER_BLOCK");
    const int v = e ? std::atoi(e) : 512;
    return std::max(PIECE, v / PIECE * PIECE);
  }();
  const int np = (w + PIECE - 1) / PIECE;
  const int ppo = OUTER / PIECE;                       // pieces per outer block
  for (int k = 0; k < np; k++) {
    const int off = k * PIECE, wk = std::min(PIECE, w - off), rk = m - off - wk;
    const int o0 = (k / ppo) * OUTER;                   // first column of this piece's outer block
    const int o1 = std::min(w, o0 + OUTER);             // one past its last pivot column
    double* Ak = A0 + off * (ld + 1);
    double* tk = B.tmp + (int64_t)k * 2 * PIECE * PIECE;
    if (off > o0) {
      // products of the earlier pieces of this outer block: columns of this piece from its diagonal block down, then
      // its rows to the right of the diagonal block
      gemm_f64<1, 0, 0>(Ak, ld, sA, A0 + off + ld * o0, ld, sA, A0 + o0 + ld * off, ld, sA, m - off, wk, off - o0, nbc);
      if (rk > 0)
        gemm_f64<1, 0, 0>(Ak + ld * wk, ld, sA, A0 + off + ld * o0, ld, sA, A0 + o0 + ld * (off + wk), ld, sA, wk, rk, off - o0, nbc);
    }
    static const bool scalar_pivot = std::getenv("HYMLS_MI_PIVOT_BLOCKED") && std::atoi(std::getenv("HYMLS_MI_PIVOT_BLOCKED")) == 0;   // (A/B switch)
    const int Wk = (wk + PB - 1) / PB * PB;
    {
    SetupProf::Scope prof("pivot", wk, 0, 0, 0, nbc, 4.0 / 3 * wk * wk * wk * nbc);
    // (measured, tools/pivot_check.hip: the blocked kernel wins unless the identity padding to a multiple of 32 is large)
    if (scalar_pivot || !(wk > 96 || (wk > 16 && Wk - wk < 24))) {
      hipLaunchKernelGGL(k_big_pivot, dim3(nbc), dim3(PIVOT_T), (size_t)(wk * wk + 2 * wk) * sizeof(double), g_stream, Ak, ld, sA, wk, slab + off * (lds + 1), lds, sS, tk, sT, B.flag);
    } else {
      hipLaunchKernelGGL(k_big_pivot_blk, dim3(nbc), dim3(PIVB_T), (size_t)(Wk * Wk + PIVB_LT * PB) * sizeof(double), g_stream, Ak, ld, sA, wk, slab + off * (lds + 1), lds, sS, tk, sT, B.flag);
    }
    launch_check();
    }
    if (rk > 0) {
      { SetupProf::Scope prof("trmm", wk, rk, 0, 0, nbc, 2.0 * wk * wk * rk * nbc);
      static const bool scalar_trmm = std::getenv("HYMLS_MI_TRMM_SCALAR") != nullptr;   // (A/B switch)
      if (!scalar_trmm) panel_trmm(Ak, ld, sA, tk, sT, wk, rk, nbc);
      else {
      hipLaunchKernelGGL(k_big_trmm_u, dim3((rk + 7) / 8, nbc), dim3(256), (size_t)wk * 8 * sizeof(double), g_stream, Ak, ld, sA, wk, rk, tk, sT);
      launch_check();
      hipLaunchKernelGGL(k_big_trmm_l, dim3((rk + 7) / 8, nbc), dim3(256), (size_t)wk * 8 * sizeof(double), g_stream, Ak, ld, sA, wk, rk, tk, sT);
      launch_check(); } }
      // the last piece of an outer block: everything beyond the block, one update of rank o1 - o0
      const int rt = m - o1;
      if (off + wk == o1 && rt > 0)
        gemm_f64<1, 0, 0>(A0 + o1 * (ld + 1), ld, sA, A0 + o1 + ld * o0, ld, sA, A0 + o0 + ld * o1, ld, sA, rt, rt, o1 - o0, nbc);
    }
  }
  // ---- explicit inverse of the whole pivot block (block columns / rows from the last to the first)
  double* T = B.tmp + (int64_t)np * 2 * PIECE * PIECE;   // (w x PIECE) panel scratch
  for (int j = np - 2; j >= 0; j--) {
    const int off = j * PIECE, wj = PIECE, rj = w - off - wj;
    const double* Lf = B.tmp + (int64_t)j * 2 * PIECE * PIECE;
    const double* Uf = Lf + (int64_t)PIECE * PIECE;
    // L^{-1}[J+, j] = - L^{-1}[J+, J+] (L[J+, j] L_jj^{-1})
    gemm_f64<0, 0, 0>(T, rj, sT, A0 + (off + wj) + ld * off, ld, sA, Lf, wj, sT, rj, wj, wj, nbc);
    gemm_f64<2, 1, 0>(slab + (off + wj) + lds * off, lds, sS, slab + (off + wj) * (lds + 1), lds, sS, T, rj, sT, rj, wj, rj, nbc);
    // U^{-1}[j, J+] = - (U_jj^{-1} U[j, J+]) U^{-1}[J+, J+]
    gemm_f64<0, 0, 0>(T, wj, sT, Uf, wj, sT, A0 + off + ld * (off + wj), ld, sA, wj, rj, wj, nbc);
    gemm_f64<2, 0, 2>(slab + off + lds * (off + wj), lds, sS, T, wj, sT, slab + (off + wj) * (lds + 1), lds, sS, wj, rj, rj, nbc);
  }
  if (ri > 0) {
    // solve panels: PL = L21_int L^{-1}, QU = U^{-1} U12_int
    gemm_f64<0, 0, 1>(slab + w, lds, sS, A0 + w, ld, sA, slab, lds, sS, ri, w, w, nbc);
    gemm_f64<0, 2, 0>(Qs, w, sS, slab, lds, sS, A0 + ld * w, ld, sA, w, ri, w, nbc);
  }
  if (F.parent < 0 && rs > 0) {
    SetupProf::Scope prof("root_update", rs, 0, 0, 0, nbc);
    root_update(P, B, F, nbc);
  }
}

// ---- solves of the big fronts, all big fronts of one tree level per launch
// assembled vector a = [x pivot slice ; 0] + pulled child contributions, into the workspace
__global__ void __launch_bounds__(256) k_asm_big(PlanD P, BatchD B, const int32_t* __restrict__ list, const double* __restrict__ x) {
  const FrontD F = P.fronts[list[blockIdx.y]];
  const int b = blockIdx.z, rows = F.w + F.ri;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= rows) return;
  const double* cb = B.contrib + (int64_t)b * P.contrib_size;
  double v = j < F.w ? x[B.xoff[b] + F.c0 + j] : 0.0;
  for (int t = P.asm_ptr[F.a_off + j]; t < P.asm_ptr[F.a_off + j + 1]; t++) v += cb[P.asm_src[t]];
  B.swork[(int64_t)b * B.swork_stride + F.a_off + j] = v;
}
// Panel-times-vector products of the big fronts.  A workgroup (4 waves) owns a tile of 64 rows x
// KT = 1024 (or 256) columns; lane = row (every panel load is a coalesced 512-byte line), wave g takes 256
// of the columns, 8 independent loads in flight per lane.  Partial sums go to the workspace and a
// finalize kernel adds the column tiles in a fixed order (bitwise reproducible).  Tiles that lie
// entirely outside the triangular part of the pivot block are skipped by both kernels.
__device__ inline double wave4_reduce_store(double v, double (*red)[64], int g, int lane) {
  red[g][lane] = v;
  __syncthreads();
  return red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
}
template <int KT>
__global__ void __launch_bounds__(256) k_panel_fwd(PlanD P, BatchD B, const int32_t* __restrict__ list, const int64_t* __restrict__ poff,
                                                    int32_t count) {
  __shared__ double red[4][64];
  // (measured and not kept, profiles/r03_f_ab_*, r03_q_*: XCD-aware block orders meant to let the two workgroups that share
  // a cache line -- vertically adjacent 64-row tiles; FETCH_SIZE shows 1.25 x the panel bytes here -- share an L2.  All
  // consecutive tiles on one XCD: slower (the largest fronts land on one XCD); chunks of four, balanced: no difference)
  const int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  const int q = bz % count, b = bz / count;
  const FrontD F = P.fronts[list[q]];
  const int w = F.w, rows = F.w + F.ri;
  const int r0 = bx * 64, c0k = by * KT;
  if (r0 >= rows || c0k >= w) return;
  const int kchunk = (r0 + 63 < w) ? r0 + 63 : w;      // columns >= kchunk are zero for every row of the chunk
  if (c0k >= kchunk) return;
  const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int i = r0 + lane;
  const int64_t ld = rows;
  const double* __restrict__ a = B.swork + (int64_t)b * B.swork_stride + F.a_off;
  const double* __restrict__ Lp = B.factor + (int64_t)b * P.factor_size + F.lp_off + (i < rows ? i : 0);
  const int krow = i < rows ? (i < w ? i : w) : 0;    // this row uses columns k < krow
  const int kb = c0k + g * (KT / 4), ke = min(min(kb + KT / 4, kchunk), c0k + KT);
  double acc[8];
#pragma unroll
  for (int u = 0; u < 8; u++) acc[u] = 0.0;
  int k = kb;
  for (; k + 7 < ke; k += 8) {
    double l[8], t[8];
#pragma unroll
    for (int u = 0; u < 8; u++) { l[u] = Lp[ld * (k + u)]; t[u] = a[k + u]; }
#pragma unroll
    for (int u = 0; u < 8; u++) if (k + u < krow) acc[u] += l[u] * t[u];
  }
  for (; k < ke; k++) if (k < krow) acc[0] += Lp[ld * k] * a[k];
  const double v = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
  const double sum = wave4_reduce_store(v, red, g, lane);
  if (g == 0 && i < rows) B.swork[(int64_t)b * B.swork_stride + P.asm_rows + poff[q] + (int64_t)by * rows + i] = sum;
}
template <int KT>
__global__ void __launch_bounds__(256) k_final_fwd(PlanD P, BatchD B, const int32_t* __restrict__ list, const int64_t* __restrict__ poff,
                                                    int32_t count, double* __restrict__ x) {
  const int q = blockIdx.y % count, b = blockIdx.y / count;
  const FrontD F = P.fronts[list[q]];
  const int w = F.w, rows = F.w + F.ri;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= rows) return;
  const int r0 = (i / 64) * 64;
  const int kchunk = (r0 + 63 < w) ? r0 + 63 : w;
  const double* part = B.swork + (int64_t)b * B.swork_stride + P.asm_rows + poff[q];
  double sum = 0.0;
  for (int ct = 0; ct * KT < kchunk; ct++) sum += part[(int64_t)ct * rows + i];
  const double ai = B.swork[(int64_t)b * B.swork_stride + F.a_off + i];
  if (i < w) x[B.xoff[b] + F.c0 + i] = ai + sum;
  else B.contrib[(int64_t)b * P.contrib_size + F.c_off + i - w] = ai - sum;
}
void solve_fwd_big(const PlanD& P, const BatchD& B, const int32_t* list, const FrontD* hf, const int64_t* poff, int32_t count,
                   double* x) {
  if (count <= 0 || B.nb <= 0) return;
  if ((int64_t)count * B.nb > 65535) throw Error(-3, "too many (front, member) pairs for the big-front path");
  int maxrows = 0, maxw = 0;
  for (int q = 0; q < count; q++) { maxrows = std::max(maxrows, hf[q].w + hf[q].ri); maxw = std::max(maxw, hf[q].w); }
  hipLaunchKernelGGL(k_asm_big, dim3((maxrows + 255) / 256, count, B.nb), dim3(256), 0, g_stream, P, B, list, x); launch_check();
  if (solve_kt() == 256) {
    hipLaunchKernelGGL(k_panel_fwd<256>, dim3((maxrows + 63) / 64, (maxw + 255) / 256, count * B.nb), dim3(256), 0, g_stream, P, B, list, poff, count);
    launch_check();
    hipLaunchKernelGGL(k_final_fwd<256>, dim3((maxrows + 255) / 256, count * B.nb), dim3(256), 0, g_stream, P, B, list, poff, count, x);
  } else {
    hipLaunchKernelGGL(k_panel_fwd<1024>, dim3((maxrows + 63) / 64, (maxw + 1023) / 1024, count * B.nb), dim3(256), 0, g_stream, P, B, list, poff, count);
    launch_check();
    hipLaunchKernelGGL(k_final_fwd<1024>, dim3((maxrows + 255) / 256, count * B.nb), dim3(256), 0, g_stream, P, B, list, poff, count, x);
  }
  launch_check();
}
// backward: x_s = U^{-1} y_s - (U^{-1} U12) x_ancestors; column tiles [0, ctU) run over the upper
// triangle of the pivot block, the following ones over the U-side panel
template <int KT>
__global__ void __launch_bounds__(256) k_panel_bwd(PlanD P, BatchD B, const int32_t* __restrict__ list, const int64_t* __restrict__ poff,
                                                    int32_t count, const double* __restrict__ x) {
  __shared__ double red[4][64];
  const int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  const int q = bz % count, b = bz / count;
  const FrontD F = P.fronts[list[q]];
  const int w = F.w, ri = F.ri;
  const int r0 = bx * 64;
  if (r0 >= w) return;
  const int ctU = (w + KT - 1) / KT, ct = by;
  const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int i = r0 + lane;
  const double* xb = x + B.xoff[b];
  const double* fac = B.factor + (int64_t)b * P.factor_size;
  double acc[8];
#pragma unroll
  for (int u = 0; u < 8; u++) acc[u] = 0.0;
  if (ct < ctU) {
    const int c0k = ct * KT;
    if (c0k + KT <= r0) return;                       // tile left of the diagonal: all zero
    const int64_t ld = w + ri;
    const double* __restrict__ Lp = fac + F.lp_off + (i < w ? i : 0);
    const double* __restrict__ y = xb + F.c0;
    const int kb = max(c0k + g * (KT / 4), (r0 / 8) * 8), ke = min(c0k + g * (KT / 4) + KT / 4, w);
    int k = kb;
    for (; k + 7 < ke; k += 8) {
      double l[8], t[8];
#pragma unroll
      for (int u = 0; u < 8; u++) { l[u] = Lp[ld * (k + u)]; t[u] = y[k + u]; }
#pragma unroll
      for (int u = 0; u < 8; u++) if (k + u >= i && i < w) acc[u] += l[u] * t[u];
    }
    for (; k < ke; k++) if (k >= i && i < w) acc[0] += Lp[ld * k] * y[k];
  } else {
    const int c0k = (ct - ctU) * KT;
    if (c0k >= ri) return;
    const double* __restrict__ Q = fac + F.q_off + (i < w ? i : 0);
    const int32_t* __restrict__ idx = P.fidx + F.idx_off + w;
    const int kb = c0k + g * (KT / 4), ke = min(kb + KT / 4, ri);
    int k = kb;
    for (; k + 7 < ke; k += 8) {
      double l[8], t[8];
#pragma unroll
      for (int u = 0; u < 8; u++) { l[u] = Q[(int64_t)w * (k + u)]; t[u] = xb[idx[k + u]]; }
#pragma unroll
      for (int u = 0; u < 8; u++) acc[u] -= l[u] * t[u];
    }
    for (; k < ke; k++) acc[0] -= Q[(int64_t)w * k] * xb[idx[k]];
  }
  const double v = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
  const double sum = wave4_reduce_store(v, red, g, lane);
  if (g == 0 && i < w) B.swork[(int64_t)b * B.swork_stride + P.asm_rows + poff[q] + (int64_t)ct * w + i] = sum;
}
template <int KT>
__global__ void __launch_bounds__(256) k_final_bwd(PlanD P, BatchD B, const int32_t* __restrict__ list, const int64_t* __restrict__ poff,
                                                    int32_t count, double* __restrict__ x) {
  const int q = blockIdx.y % count, b = blockIdx.y / count;
  const FrontD F = P.fronts[list[q]];
  const int w = F.w, ri = F.ri;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= w) return;
  const int r0 = (i / 64) * 64;
  const int ctU = (w + KT - 1) / KT, ctQ = (ri + KT - 1) / KT;
  const double* part = B.swork + (int64_t)b * B.swork_stride + P.asm_rows + poff[q];
  double sum = 0.0;
  for (int ct = 0; ct < ctU; ct++) if (ct * KT + KT > r0) sum += part[(int64_t)ct * w + i];
  for (int ct = 0; ct < ctQ; ct++) sum += part[(int64_t)(ctU + ct) * w + i];
  x[B.xoff[b] + F.c0 + i] = sum;
}
void solve_bwd_big(const PlanD& P, const BatchD& B, const int32_t* list, const FrontD* hf, const int64_t* poff, int32_t count,
                   double* x) {
  if (count <= 0 || B.nb <= 0) return;
  if ((int64_t)count * B.nb > 65535) throw Error(-3, "too many (front, member) pairs for the big-front path");
  const int KT = solve_kt();
  int maxw = 0, maxct = 0;
  for (int q = 0; q < count; q++) {
    maxw = std::max(maxw, hf[q].w);
    maxct = std::max(maxct, (hf[q].w + KT - 1) / KT + (hf[q].ri + KT - 1) / KT);
  }
  if (KT == 256) {
    hipLaunchKernelGGL(k_panel_bwd<256>, dim3((maxw + 63) / 64, maxct, count * B.nb), dim3(256), 0, g_stream, P, B, list, poff, count, x);
    launch_check();
    hipLaunchKernelGGL(k_final_bwd<256>, dim3((maxw + 255) / 256, count * B.nb), dim3(256), 0, g_stream, P, B, list, poff, count, x);
  } else {
    hipLaunchKernelGGL(k_panel_bwd<1024>, dim3((maxw + 63) / 64, maxct, count * B.nb), dim3(256), 0, g_stream, P, B, list, poff, count, x);
    launch_check();
    hipLaunchKernelGGL(k_final_bwd<1024>, dim3((maxw + 255) / 256, count * B.nb), dim3(256), 0, g_stream, P, B, list, poff, count, x);
  }
  launch_check();
}

// ------------------------------------------------------------------ solves
// forward: [y ; contrib] = [L11^{-1} ; -L21 L11^{-1}] * t, t assembled from x and the children
__global__ void __launch_bounds__(256) k_solve_fwd(PlanD P, BatchD B, const int32_t* __restrict__ list, double* __restrict__ x) {
  extern __shared__ double f[];
  const int tid = threadIdx.x;
  const int b = blockIdx.y;
  const FrontD F = P.fronts[list[blockIdx.x]];
  const int w = F.w, ri = F.ri, ld = w + ri;
  double* xb = x + B.xoff[b];
  double* cb = B.contrib + (int64_t)b * P.contrib_size;
  for (int j = tid; j < ld; j += blockDim.x) {
    double v = j < w ? xb[F.c0 + j] : 0.0;
    for (int t = P.asm_ptr[F.a_off + j]; t < P.asm_ptr[F.a_off + j + 1]; t++) v += cb[P.asm_src[t]];
    f[j] = v;
  }
  __syncthreads();
  const double* __restrict__ Lp = B.factor + (int64_t)b * P.factor_size + F.lp_off;
  for (int i = tid; i < ld; i += blockDim.x) {
    double s = f[i];
    if (i < w) {
      for (int k = 0; k < i; k++) s += Lp[i + (int64_t)ld * k] * f[k];
      xb[F.c0 + i] = s;
    } else {
      double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
      int k = 0;
      for (; k + 3 < w; k += 4) {
        s0 += Lp[i + (int64_t)ld * k] * f[k];
        s1 += Lp[i + (int64_t)ld * (k + 1)] * f[k + 1];
        s2 += Lp[i + (int64_t)ld * (k + 2)] * f[k + 2];
        s3 += Lp[i + (int64_t)ld * (k + 3)] * f[k + 3];
      }
      for (; k < w; k++) s0 += Lp[i + (int64_t)ld * k] * f[k];
      cb[F.c_off + i - w] = s - ((s0 + s1) + (s2 + s3));
    }
  }
}
// backward: x_s = U11^{-1} y_s - (U11^{-1} U12) x_ancestors
__global__ void __launch_bounds__(256) k_solve_bwd(PlanD P, BatchD B, const int32_t* __restrict__ list, double* __restrict__ x) {
  extern __shared__ double g[];
  const int tid = threadIdx.x;
  const int b = blockIdx.y;
  const FrontD F = P.fronts[list[blockIdx.x]];
  const int w = F.w, ri = F.ri, ld = w + ri;
  double* xb = x + B.xoff[b];
  const int32_t* idx = P.fidx + F.idx_off;
  for (int k = tid; k < ld; k += blockDim.x) g[k] = k < w ? xb[F.c0 + k] : xb[idx[k]];
  __syncthreads();
  const double* fac = B.factor + (int64_t)b * P.factor_size;
  const double* __restrict__ Lp = fac + F.lp_off;
  const double* __restrict__ Q = fac + F.q_off;
  for (int i = tid; i < w; i += blockDim.x) {
    double s = 0.0;
    for (int k = i; k < w; k++) s += Lp[i + (int64_t)ld * k] * g[k];
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int k = 0;
    for (; k + 3 < ri; k += 4) {
      s0 += Q[i + (int64_t)w * k] * g[w + k];
      s1 += Q[i + (int64_t)w * (k + 1)] * g[w + k + 1];
      s2 += Q[i + (int64_t)w * (k + 2)] * g[w + k + 2];
      s3 += Q[i + (int64_t)w * (k + 3)] * g[w + k + 3];
    }
    for (; k < ri; k++) s0 += Q[i + (int64_t)w * k] * g[w + k];
    xb[F.c0 + i] = s - ((s0 + s1) + (s2 + s3));
  }
}

static int solve_block_size(int rows) { return rows <= 64 ? 64 : (rows <= 128 ? 128 : 256); }

void solve_fwd_level(const PlanD& P, const BatchD& B, const int32_t* list, int32_t count, double* x) {
  if (count <= 0 || B.nb <= 0) return;
  // LDS: largest front of the class (upper bound for every level)
  const int rows = P.max_solve_rows;
  const size_t shm = (size_t)rows * sizeof(double);
  if (shm > 64 * 1024) throw Error(-3, "front too large for the single-workgroup solve kernel");
  for (int s0 = 0; s0 < B.nb; s0 += 65535) {
    const int ns = std::min(65535, B.nb - s0);
    BatchD B2 = B;
    B2.xoff = B.xoff + s0; B2.factor = B.factor + (int64_t)s0 * P.factor_size; B2.contrib = B.contrib + (int64_t)s0 * P.contrib_size;
    hipLaunchKernelGGL(k_solve_fwd, dim3(count, ns), dim3(solve_block_size(rows)), shm, g_stream, P, B2, list, x);
    launch_check();
  }
}
void solve_bwd_level(const PlanD& P, const BatchD& B, const int32_t* list, int32_t count, double* x) {
  if (count <= 0 || B.nb <= 0) return;
  const int rows = P.max_solve_rows;
  const size_t shm = (size_t)rows * sizeof(double);
  if (shm > 64 * 1024) throw Error(-3, "front too large for the single-workgroup solve kernel");
  for (int s0 = 0; s0 < B.nb; s0 += 65535) {
    const int ns = std::min(65535, B.nb - s0);
    BatchD B2 = B;
    B2.xoff = B.xoff + s0; B2.factor = B.factor + (int64_t)s0 * P.factor_size; B2.contrib = B.contrib + (int64_t)s0 * P.contrib_size;
    hipLaunchKernelGGL(k_solve_bwd, dim3(count, ns), dim3(solve_block_size(rows)), shm, g_stream, P, B2, list, x);
    launch_check();
  }
}

// ------------------------------------------------------------------ merged level solve
// (see device.hpp) one workgroup per task, 256 threads.  Tile tasks: lane = row (coalesced 512-byte panel
// loads), the four waves split the columns, fixed-order reduction through LDS (bitwise reproducible).
__global__ void __launch_bounds__(256) k_lvl_fwd(const LvlTask* __restrict__ tasks, const LvlSub* __restrict__ subs,
                                                  const PlanD* __restrict__ plans, const double* __restrict__ x,
                                                  double* __restrict__ y) {
  extern __shared__ double f[];
  const LvlTask T = tasks[blockIdx.x];
  const LvlSub S = subs[T.sub];
  const PlanD* P = plans + S.cls;
  const FrontD F = P->fronts[T.front];
  const int tid = threadIdx.x, w = F.w, rows = F.w + F.ri;
  const int64_t ld = rows;
  const double* xb = x + S.xoff;
  double* yb = y + S.xoff;
  const gmptr<double> cb = as_global_rw(S.contrib);
  const gptr<int32_t> aptr = as_global(P->asm_ptr) + F.a_off;
  const gptr<int32_t> asrc = as_global(P->asm_src);
  const gptr<double> Lp = as_global(S.fac) + F.lp_off;
  if (T.r0 < 0) {
    for (int j = tid; j < rows; j += 256) {
      double v = j < w ? xb[F.c0 + j] : 0.0;
      for (int t = aptr[j]; t < aptr[j + 1]; t++) v += cb[asrc[t]];
      f[j] = v;
    }
    __syncthreads();
    for (int i = tid; i < rows; i += 256) {
      const int kmax = i < w ? i : w;
      double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
      int k = 0;
      for (; k + 3 < kmax; k += 4) {
        s0 += Lp[i + ld * k] * f[k];
        s1 += Lp[i + ld * (k + 1)] * f[k + 1];
        s2 += Lp[i + ld * (k + 2)] * f[k + 2];
        s3 += Lp[i + ld * (k + 3)] * f[k + 3];
      }
      for (; k < kmax; k++) s0 += Lp[i + ld * k] * f[k];
      const double s = (s0 + s1) + (s2 + s3);
      if (i < w) yb[F.c0 + i] = f[i] + s;
      else cb[F.c_off + i - w] = f[i] - s;
    }
    return;
  }
  const int r0 = T.r0, lane = tid & 63, g = tid >> 6, i = r0 + lane;
  const int kneed = min(w, r0 + 63);                 // row i uses columns k < min(i, w)
  double* own = f + ((kneed + 7) & ~7);
  double (*red)[64] = (double (*)[64])(own + 64);
  for (int j = tid; j < kneed; j += 256) {
    double v = xb[F.c0 + j];
    for (int t = aptr[j]; t < aptr[j + 1]; t++) v += cb[asrc[t]];
    f[j] = v;
  }
  if (tid < 64 && i < rows) {
    double v = i < w ? xb[F.c0 + i] : 0.0;
    for (int t = aptr[i]; t < aptr[i + 1]; t++) v += cb[asrc[t]];
    own[lane] = v;
  }
  __syncthreads();
  const int krow = i < rows ? (i < w ? i : w) : 0;
  const gptr<double> Lr = Lp + (i < rows ? i : 0);
  const int chunk = ((kneed + 31) / 32) * 8;
  const int kb = g * chunk, ke = min(kb + chunk, kneed);
  double acc[8];
#pragma unroll
  for (int u = 0; u < 8; u++) acc[u] = 0.0;
  int k = kb;
  for (; k + 7 < ke; k += 8) {
    double l[8];
#pragma unroll
    for (int u = 0; u < 8; u++) l[u] = Lr[ld * (k + u)];
#pragma unroll
    for (int u = 0; u < 8; u++) if (k + u < krow) acc[u] += l[u] * f[k + u];
  }
  for (; k < ke; k++) if (k < krow) acc[0] += Lr[ld * k] * f[k];
  red[g][lane] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
  __syncthreads();
  if (g == 0 && i < rows) {
    const double sum = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
    if (i < w) yb[F.c0 + i] = own[lane] + sum;
    else cb[F.c_off + i - w] = own[lane] - sum;
  }
}

__global__ void __launch_bounds__(256) k_lvl_bwd(const LvlTask* __restrict__ tasks, const LvlSub* __restrict__ subs,
                                                  const PlanD* __restrict__ plans, const double* __restrict__ y,
                                                  double* __restrict__ x) {
  extern __shared__ double f[];
  const LvlTask T = tasks[blockIdx.x];
  const LvlSub S = subs[T.sub];
  const PlanD* P = plans + S.cls;
  const FrontD F = P->fronts[T.front];
  const int tid = threadIdx.x, w = F.w, ri = F.ri;
  const int64_t ld = w + ri;
  double* xb = x + S.xoff;
  const double* yb = y + S.xoff;
  const gptr<int32_t> idx = as_global(P->fidx) + F.idx_off + w;
  const gptr<double> Lp = as_global(S.fac) + F.lp_off;
  const gptr<double> Q = as_global(S.fac) + F.q_off;
  if (T.r0 < 0) {
    for (int k = tid; k < w + ri; k += 256) f[k] = k < w ? yb[F.c0 + k] : xb[idx[k - w]];
    __syncthreads();
    for (int i = tid; i < w; i += 256) {
      double s = 0.0;
      for (int k = i; k < w; k++) s += Lp[i + ld * k] * f[k];
      double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
      int k = 0;
      for (; k + 3 < ri; k += 4) {
        s0 += Q[i + (int64_t)w * k] * f[w + k];
        s1 += Q[i + (int64_t)w * (k + 1)] * f[w + k + 1];
        s2 += Q[i + (int64_t)w * (k + 2)] * f[w + k + 2];
        s3 += Q[i + (int64_t)w * (k + 3)] * f[w + k + 3];
      }
      for (; k < ri; k++) s0 += Q[i + (int64_t)w * k] * f[w + k];
      xb[F.c0 + i] = s - ((s0 + s1) + (s2 + s3));
    }
    return;
  }
  const int r0 = T.r0, lane = tid & 63, g = tid >> 6, i = r0 + lane;
  const int nU = w - r0, total = nU + ri;            // columns r0..w-1 of the pivot block, then the U-side panel
  double (*red)[64] = (double (*)[64])(f + ((total + 7) & ~7));
  for (int k = tid; k < total; k += 256) f[k] = k < nU ? yb[F.c0 + r0 + k] : xb[idx[k - nU]];
  __syncthreads();
  const int iv = i < w ? i : r0;
  const gptr<double> Lr = Lp + iv + ld * r0;   // column r0 + kk
  const gptr<double> Qr = Q + iv;
  const int chunk = ((total + 31) / 32) * 8;
  const int kb = g * chunk, ke = min(kb + chunk, total);
  double acc[8];
#pragma unroll
  for (int u = 0; u < 8; u++) acc[u] = 0.0;
  // pivot-block part of this wave's range
  {
    const int e = min(ke, nU);
    int k = kb;
    for (; k + 7 < e; k += 8) {
      double l[8];
#pragma unroll
      for (int u = 0; u < 8; u++) l[u] = Lr[ld * (k + u)];
#pragma unroll
      for (int u = 0; u < 8; u++) if (k + u >= lane) acc[u] += l[u] * f[k + u];
    }
    for (; k < e; k++) if (k >= lane) acc[0] += Lr[ld * k] * f[k];
  }
  // U-side panel part
  {
    int k = max(kb, nU);
    for (; k + 7 < ke; k += 8) {
      double l[8];
#pragma unroll
      for (int u = 0; u < 8; u++) l[u] = Qr[(int64_t)w * (k - nU + u)];
#pragma unroll
      for (int u = 0; u < 8; u++) acc[u] -= l[u] * f[k + u];
    }
    for (; k < ke; k++) acc[0] -= Qr[(int64_t)w * (k - nU)] * f[k];
  }
  red[g][lane] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
  __syncthreads();
  if (g == 0 && i < w) xb[F.c0 + i] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
}

void solve_fwd_tasks(const LvlTask* tasks, int32_t ntasks, const LvlSub* subs, const PlanD* plans, int32_t lds_doubles,
                     const double* x, double* y) {
  if (ntasks <= 0) return;
  if ((size_t)lds_doubles * sizeof(double) > 64 * 1024)
    HIP_CHECK(hipFuncSetAttribute((const void*)k_lvl_fwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds_doubles * sizeof(double))));
  hipLaunchKernelGGL(k_lvl_fwd, dim3(ntasks), dim3(256), (size_t)lds_doubles * sizeof(double), g_stream, tasks, subs, plans, x, y);
  launch_check();
}
void solve_bwd_tasks(const LvlTask* tasks, int32_t ntasks, const LvlSub* subs, const PlanD* plans, int32_t lds_doubles,
                     const double* y, double* x) {
  if (ntasks <= 0) return;
  if ((size_t)lds_doubles * sizeof(double) > 64 * 1024)
    HIP_CHECK(hipFuncSetAttribute((const void*)k_lvl_bwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds_doubles * sizeof(double))));
  hipLaunchKernelGGL(k_lvl_bwd, dim3(ntasks), dim3(256), (size_t)lds_doubles * sizeof(double), g_stream, tasks, subs, plans, y, x);
  launch_check();
}

// ------------------------------------------------------------------ bordered systems
__global__ void __launch_bounds__(256) k_dot_partial(int64_t n, const double* __restrict__ x, const double* __restrict__ y, double* __restrict__ part) {
  __shared__ double red[256];
  double s = 0.0;
  for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) s += x[i] * y[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) { if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st]; __syncthreads(); }
  if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}
double dot(int64_t n, const double* x, const double* y) {
  if (n <= 0) return 0.0;
  double*& dpart = ctx().dpart;
  if (!dpart) dpart = (double*)alloc(1024 * sizeof(double));
  const int nb = (int)std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, 1024));
  hipLaunchKernelGGL(k_dot_partial, dim3(nb), dim3(256), 0, g_stream, n, x, y, dpart);
  launch_check();
  double h[1024];
  d2h(h, dpart, nb * sizeof(double));
  double s = 0.0;
  for (int i = 0; i < nb; i++) s += h[i];
  return s;
}

// panel entries of a front in either layout (device.hpp)
struct PanelRef {
  const double* lp; const double* q; int64_t w, ri, ld; int packed;
  __device__ double linv(int i, int k) const { return packed ? lp[packed_lower(w, i, k)] : lp[i + ld * k]; }            // i > k
  __device__ double uinv(int i, int k) const { return packed ? lp[packed_upper(w, ri, i, k)] : lp[i + ld * k]; }        // i <= k
  __device__ double pl(int j, int k) const { return packed ? lp[packed_l21(w, ri, j, k)] : lp[(w + j) + ld * k]; }       // row j of L21 L11^{-1}
  __device__ double qu(int i, int j) const { return q[i + w * j]; }                                                     // U11^{-1} U12
};
__global__ void __launch_bounds__(256) k_solve_transposed(PlanD P, BatchD B, const int32_t* __restrict__ order, int32_t nfronts,
                                                           double* __restrict__ x) {
  extern __shared__ double a[];
  const int b = blockIdx.x, tid = threadIdx.x;
  double* xb = x + B.xoff[b];
  double* cb = B.contrib + (int64_t)b * P.contrib_size;
  const double* fac = B.factor + (int64_t)b * P.factor_size;
  // forward with U^T: z = U11^{-T} a, contributions a_upd - (U11^{-1} U12)^T a
  for (int q = 0; q < nfronts; q++) {
    const FrontD F = P.fronts[order[q]];
    const int w = F.w, ri = F.ri;
    const PanelRef R{fac + F.lp_off, fac + F.q_off, w, ri, w + ri, P.packed};
    for (int j = tid; j < w + ri; j += 256) {
      double v = j < w ? xb[F.c0 + j] : 0.0;
      for (int t = P.asm_ptr[F.a_off + j]; t < P.asm_ptr[F.a_off + j + 1]; t++) v += cb[P.asm_src[t]];
      a[j] = v;
    }
    __syncthreads();
    for (int i = tid; i < w + ri; i += 256) {
      double s = 0.0;
      if (i < w) { for (int k = 0; k <= i; k++) s += R.uinv(k, i) * a[k]; xb[F.c0 + i] = s; }
      else { for (int k = 0; k < w; k++) s += R.qu(k, i - w) * a[k]; cb[F.c_off + i - w] = a[i] - s; }
    }
    __syncthreads();
  }
  // backward with L^T: x = L11^{-T} z - (L21 L11^{-1})^T x_ancestors
  for (int q = nfronts - 1; q >= 0; q--) {
    const FrontD F = P.fronts[order[q]];
    const int w = F.w, ri = F.ri;
    const PanelRef R{fac + F.lp_off, fac + F.q_off, w, ri, w + ri, P.packed};
    const int32_t* idx = P.fidx + F.idx_off;
    for (int k = tid; k < w + ri; k += 256) a[k] = k < w ? xb[F.c0 + k] : xb[idx[k]];
    __syncthreads();
    for (int i = tid; i < w; i += 256) {
      double s = a[i];
      for (int k = i + 1; k < w; k++) s += R.linv(k, i) * a[k];
      for (int j = 0; j < ri; j++) s -= R.pl(j, i) * a[w + j];
      xb[F.c0 + i] = s;
    }
    __syncthreads();
  }
}
void solve_transposed(const PlanD& P, const BatchD& B, const int32_t* order, int32_t nfronts, int32_t max_rows, double* x) {
  if (B.nb <= 0 || nfronts <= 0) return;
  const size_t shm = (size_t)max_rows * sizeof(double);
  if (shm > 64 * 1024) throw Error(-99, "front too large for the transposed solve (bordered systems)");
  hipLaunchKernelGGL(k_solve_transposed, dim3(B.nb), dim3(256), shm, g_stream, P, B, order, nfronts, x);
  launch_check();
}

// ------------------------------------------------------------------ packed panels
// one workgroup per (front, member): copy the (w+ri) x w panel to the frontal scratch of the member's slot,
// write it back in the packed order (device.hpp)
__global__ void __launch_bounds__(256) k_repack(PlanD P, BatchD B, int32_t b0) {
  const FrontD F = P.fronts[blockIdx.x];
  const int slot = blockIdx.y, b = b0 + slot;
  const int64_t w = F.w, ri = F.ri, ld = w + ri, n = ld * w;
  double* Lp = B.factor + (int64_t)b * P.factor_size + F.lp_off;
  double* tmp = B.scratch + (int64_t)slot * P.scratch_size + F.f_off;   // this front's own frontal matrix: (w+ri+rs)^2 >= (w+ri) w
  for (int64_t t = threadIdx.x; t < n; t += 256) tmp[t] = Lp[t];
  __syncthreads();
  for (int64_t t = threadIdx.x; t < n; t += 256) {
    const int64_t i = t % ld, k = t / ld;
    const int64_t dst = i >= w ? packed_l21(w, ri, i - w, k) : (i > k ? packed_lower(w, i, k) : packed_upper(w, ri, i, k));
    Lp[dst] = tmp[t];
  }
}
void repack_fronts(const PlanD& P, const BatchD& B, int32_t b0, int32_t nbc) {
  if (P.nfronts <= 0 || nbc <= 0) return;
  hipLaunchKernelGGL(k_repack, dim3(P.nfronts, nbc), dim3(256), 0, g_stream, P, B, b0);
  launch_check();
}

// ------------------------------------------------------------------ fused interior solve
// One workgroup per subdomain, level-synchronous: all fronts of one tree level are processed
// together, one work item per row of [pivot | update rows] (forward) or per pivot row (backward),
// so the many small leaf fronts cost ONE memory round trip per level instead of one per front.
// LDS: X[nI] solution | C[contrib_size] contribution vectors | F[max_level_rows] assembled
// vectors of the current level | R[256] reduction scratch.  Assembly is a pull (fixed order),
// hence no write conflicts and bitwise reproducible results.  A panel ((w+ri) x w, column-major)
// is read with consecutive rows on consecutive lanes; with few items the k range is split over
// 2 or 4 thread groups so that small levels still keep many loads in flight.
struct FusedFront { int32_t c0, w, ri, c_off, a_off, lf_off, idx_off, pad; int64_t lp_off, q_off; };
// addressing of one row of the L-side / U-side panel of a front (plain or packed, device.hpp):
// L-side entry (r, k) = p[c1 k - tri k (k + 3) / 2], U11^{-1} entry (i, k) = p[c1 k + tri k (k + 1) / 2]
template <class PTR>
__device__ inline PTR lside(PTR base, int packed, int r, int w, int ri, int& c1, int& tri) {
  if (!packed) { c1 = w + ri; tri = 0; return base + r; }
  if (r < w) { c1 = w; tri = 1; return base + (r - 1); }        // k (2w - k - 1) / 2 + r - k - 1 = (r - 1) + w k - k (k + 3) / 2
  c1 = ri; tri = 0;
  return base + ((w * (w - 1)) >> 1) + (r - w);
}
template <class PTR>
__device__ inline PTR uside(PTR base, int packed, int i, int w, int ri, int& c1, int& tri) {
  if (!packed) { c1 = w + ri; tri = 0; return base + i; }
  c1 = 0; tri = 1;
  return base + ((w * (w - 1)) >> 1) + ri * w + i;
}
// (measured and not kept, round 3, profiles/r03_e_*, r03_f_*: requesting the first panel entries of the next phase before
// the level barrier -- the extra live registers spill at 8 waves per SIMD, 9.7 / 10.9 ms per launch instead of 8.8;
// non-temporal panel loads -- 13.0 ms: the hint defeats the L2 reuse of the lines neighbouring columns share)
template <bool PROF>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) k_interior_fused(const FusedSub* __restrict__ subs, const PlanD* __restrict__ plans,
                                                         double* __restrict__ x, long long* __restrict__ prof, FusedIO io) {
  extern __shared__ double lds[];
  auto ldp = [](gptr<double> q) { return *q; };
  long long tp[6] = {0, 0, 0, 0, 0, 0}, t0 = 0, tstart = 0;
  auto tick = [&](int bucket) { if (PROF) { const long long t = wall_clock64(); tp[bucket] += t - t0; t0 = t; } };
  if (PROF) { t0 = wall_clock64(); tstart = t0; }
  const FusedSub S = subs[blockIdx.x];
  const PlanD P = plans[S.cls];
  double* X = lds;
  double* C = lds + P.nI;
  double* Fv = C + P.contrib_size;
  // reduction scratch of the k-split levels (<= 128 items, so F uses at most its first 128 entries there): inside F
  double* R = Fv + 128;
  FusedFront* LF = (FusedFront*)(Fv + (P.max_level_rows > 384 ? P.max_level_rows : 384));   // compact front descriptors, cached in LDS
  const int tid = threadIdx.x;
  for (int i = tid; i < P.nfronts; i += 256) {
    const FrontD G = P.fronts[i];
    FusedFront f;
    f.c0 = G.c0; f.w = G.w; f.ri = G.ri; f.c_off = G.c_off; f.a_off = G.a_off; f.lf_off = G.lf_off; f.idx_off = G.idx_off; f.pad = 0;
    f.lp_off = G.lp_off; f.q_off = G.q_off;
    LF[i] = f;
  }
  double* xg = x + S.xoff;
  if (io.in == 0) {
    for (int i = tid; i < P.nI; i += 256) X[i] = xg[i];
  } else if (io.in == 1) {
    for (int i = tid; i < P.nI; i += 256) X[i] = io.b[io.perm[S.xoff + i]];
  } else {
    for (int i = tid; i < P.nI; i += 256) {
      double v = 0.0;
      for (int e = io.a_row[S.xoff + i]; e < io.a_row[S.xoff + i + 1]; e++) v += io.a_val[e] * io.x2[io.a_col[e]];
      X[i] = v;
    }
  }
  __syncthreads();
  tick(0);
  const gptr<double> fac = as_global(S.fac);
  const gptr<int32_t> fw_items = as_global(P.fw_items), bw_items = as_global(P.bw_items), fidx = as_global(P.fidx),
                      asm_ptr = as_global(P.asm_ptr), asm_src = as_global(P.asm_src), fw_ptr = as_global(P.fw_ptr), bw_ptr = as_global(P.bw_ptr);
  typedef int v4i __attribute__((ext_vector_type(4)));
  const gptr<v4i> fw_rec = (gptr<v4i>)P.fw_rec;          // (FwRec is 16 bytes: one dwordx4 load, unpacked below)
  // ---------------- forward (leaves to root)
  for (int lev = 0; lev < P.nlev; lev++) {
    const int ib = fw_ptr[lev], ni = fw_ptr[lev + 1] - ib;
    for (int it = tid; it < ni; it += 256) {
      FwRec rc;                                          // item + inline assembly sources: one round trip
      { const v4i q = fw_rec[ib + it]; static_assert(sizeof(FwRec) == 16, "FwRec is one dwordx4"); __builtin_memcpy(&rc, &q, 16); }
      const int item = rc.item;
      const FusedFront& F = LF[item >> 16];
      const int r = item & 0xffff;
      double v = r < F.w ? X[F.c0 + r] : 0.0;
      if (rc.n != 0xffff) {
#pragma unroll
        for (int q = 0; q < 5; q++) if (q < rc.n) v += C[rc.s[q]];
      } else {
        for (int t = asm_ptr[F.a_off + r]; t < asm_ptr[F.a_off + r + 1]; t++) v += C[asm_src[t]];
      }
      if (r < F.w) Fv[F.lf_off + r] = v; else C[F.c_off + r - F.w] = v;   // update rows are assembled in place
    }
    __syncthreads();
    tick(1);
    if (ni > 128) {
      for (int it = tid; it < ni; it += 256) {
        const int item = fw_items[ib + it];
        const FusedFront& F = LF[item >> 16];
        const int r = item & 0xffff, w = F.w;
        // entry (r, k) of the L-side panel sits at p[c1 * k - tri * k (k + 3) / 2]: plain columns (tri = 0), or the
        // packed strictly-lower triangle for a pivot row of a packed panel (tri = 1)
        int c1, tri;
        const gptr<double> p = lside(fac + F.lp_off, P.packed, r, w, F.ri, c1, tri);
        const double* f = Fv + F.lf_off;
        const int kmax = r < w ? r : w;
        double a[4];
#pragma unroll
        for (int u = 0; u < 4; u++) a[u] = 0.0;
        int k = 0;
        for (; k + 3 < kmax; k += 4) {
          double l[4];
#pragma unroll
          for (int u = 0; u < 4; u++) l[u] = ldp(p + (c1 * (k + u) - tri * (((k + u) * (k + u + 3)) >> 1)));
#pragma unroll
          for (int u = 0; u < 4; u++) a[u] += l[u] * f[k + u];
        }
        for (; k < kmax; k++) a[0] += ldp(p + (c1 * k - tri * ((k * (k + 3)) >> 1))) * f[k];
        const double sum = (a[0] + a[1]) + (a[2] + a[3]);
        if (r < w) X[F.c0 + r] = f[r] + sum; else C[F.c_off + r - w] -= sum;
      }
    } else {
      const int RT = ni > 64 ? 128 : 64, KG = 256 / RT;
      const int it = tid % RT, kg = tid / RT;
      double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
      if (it < ni) {
        const int item = fw_items[ib + it];
        const FusedFront& F = LF[item >> 16];
        const int r = item & 0xffff, w = F.w;
        int c1, tri;
        const gptr<double> p = lside(fac + F.lp_off, P.packed, r, w, F.ri, c1, tri);
        const double* f = Fv + F.lf_off;
        const int kmax = r < w ? r : w;
        auto at = [&](int kk) { return ldp(p + (c1 * kk - tri * ((kk * (kk + 3)) >> 1))); };
        int k = kg;
        for (; k + 3 * KG < kmax; k += 4 * KG) {
          const double l0 = at(k), l1 = at(k + KG), l2 = at(k + 2 * KG), l3 = at(k + 3 * KG);
          a0 += l0 * f[k]; a1 += l1 * f[k + KG]; a2 += l2 * f[k + 2 * KG]; a3 += l3 * f[k + 3 * KG];
        }
        for (; k < kmax; k += KG) a0 += at(k) * f[k];
      }
      R[kg * RT + it] = (a0 + a1) + (a2 + a3);
      __syncthreads();
      if (tid < ni) {
        const int item = fw_items[ib + tid];
        const FusedFront& F = LF[item >> 16];
        const int r = item & 0xffff;
        double sum = 0.0;
        for (int g = 0; g < KG; g++) sum += R[g * RT + tid];
        if (r < F.w) X[F.c0 + r] = Fv[F.lf_off + r] + sum; else C[F.c_off + r - F.w] -= sum;
      }
    }
    __syncthreads();
    tick(2);
  }
  // ---------------- backward (root to leaves)
  for (int lev = P.nlev - 1; lev >= 0; lev--) {
    const int ib = bw_ptr[lev], ni = bw_ptr[lev + 1] - ib;
    if (ni > 128) {
      for (int it = tid; it < ni; it += 256) {
        const int item = bw_items[ib + it];
        const FusedFront& F = LF[item >> 16];
        const int i = item & 0xffff, w = F.w, ri = F.ri;
        // entry (i, k), k >= i, of U11^{-1}: column k of the tall panel, or of the packed upper triangle
        int c1, tri;
        const gptr<double> p = uside(fac + F.lp_off, P.packed, i, w, ri, c1, tri);
        const double* Xs = X + F.c0;
        double a[4];
#pragma unroll
        for (int u = 0; u < 4; u++) a[u] = 0.0;
        int k = i;
        for (; k + 3 < w; k += 4) {
          double l[4];
#pragma unroll
          for (int u = 0; u < 4; u++) l[u] = ldp(p + (c1 * (k + u) + tri * (((k + u) * (k + u + 1)) >> 1)));
#pragma unroll
          for (int u = 0; u < 4; u++) a[u] += l[u] * Xs[k + u];
        }
        for (; k < w; k++) a[0] += ldp(p + (c1 * k + tri * ((k * (k + 1)) >> 1))) * Xs[k];
        const gptr<double> qv = fac + F.q_off + i;
        const gptr<int32_t> idx = fidx + F.idx_off + w;
        k = 0;
        for (; k + 3 < ri; k += 4) {
          double l[4]; int id[4];
#pragma unroll
          for (int u = 0; u < 4; u++) { l[u] = ldp(qv + (int64_t)w * (k + u)); id[u] = idx[k + u]; }
#pragma unroll
          for (int u = 0; u < 4; u++) a[u] -= l[u] * X[id[u]];
        }
        for (; k < ri; k++) a[0] -= ldp(qv + (int64_t)w * k) * X[idx[k]];
        Fv[it] = (a[0] + a[1]) + (a[2] + a[3]);     // F is free during the backward sweep
      }
      __syncthreads();   // every read of this level's pivot values is done
      for (int it = tid; it < ni; it += 256) {
        const int item = bw_items[ib + it];
        X[LF[item >> 16].c0 + (item & 0xffff)] = Fv[it];
      }
    } else {
      const int RT = ni > 64 ? 128 : 64, KG = 256 / RT;
      const int it = tid % RT, kg = tid / RT;
      double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
      if (it < ni) {
        const int item = bw_items[ib + it];
        const FusedFront& F = LF[item >> 16];
        const int i = item & 0xffff, w = F.w, ri = F.ri;
        int c1, tri;
        const gptr<double> p = uside(fac + F.lp_off, P.packed, i, w, ri, c1, tri);
        const double* Xs = X + F.c0;
        auto at = [&](int kk) { return ldp(p + (c1 * kk + tri * ((kk * (kk + 1)) >> 1))); };
        int k = i + kg;
        for (; k + 3 * KG < w; k += 4 * KG) {
          const double l0 = at(k), l1 = at(k + KG), l2 = at(k + 2 * KG), l3 = at(k + 3 * KG);
          a0 += l0 * Xs[k]; a1 += l1 * Xs[k + KG]; a2 += l2 * Xs[k + 2 * KG]; a3 += l3 * Xs[k + 3 * KG];
        }
        for (; k < w; k += KG) a0 += at(k) * Xs[k];
        const gptr<double> qv = fac + F.q_off + i;
        const gptr<int32_t> idx = fidx + F.idx_off + w;
        k = kg;
        for (; k + 3 * KG < ri; k += 4 * KG) {
          const double q0 = ldp(qv + (int64_t)w * k), q1 = ldp(qv + (int64_t)w * (k + KG)), q2 = ldp(qv + (int64_t)w * (k + 2 * KG)), q3 = ldp(qv + (int64_t)w * (k + 3 * KG));
          const int i0 = idx[k], i1 = idx[k + KG], i2 = idx[k + 2 * KG], i3 = idx[k + 3 * KG];
          a0 -= q0 * X[i0]; a1 -= q1 * X[i1]; a2 -= q2 * X[i2]; a3 -= q3 * X[i3];
        }
        for (; k < ri; k += KG) a0 -= ldp(qv + (int64_t)w * k) * X[idx[k]];
      }
      R[kg * RT + it] = (a0 + a1) + (a2 + a3);
      __syncthreads();
      if (tid < ni) {
        const int item = bw_items[ib + tid];
        double sum = 0.0;
        for (int g = 0; g < KG; g++) sum += R[g * RT + tid];
        X[LF[item >> 16].c0 + (item & 0xffff)] = sum;
      }
    }
    __syncthreads();
    tick(ni > 128 ? 3 : 4);
  }
  if (io.out == 0) {
    for (int i = tid; i < P.nI; i += 256) xg[i] = X[i];
  } else {
    for (int i = tid; i < P.nI; i += 256) io.user[io.perm[S.xoff + i]] = io.z[S.xoff + i] - X[i];
  }
  if (PROF && tid == 0) {
    tp[5] = wall_clock64() - tstart;
    for (int q = 0; q < 6; q++) prof[(int64_t)blockIdx.x * 8 + q] = tp[q];
    prof[(int64_t)blockIdx.x * 8 + 6] = tstart;
    prof[(int64_t)blockIdx.x * 8 + 7] = P.nI;
  }
}

void interior_solve_fused(int32_t nsub, const FusedSub* subs, const PlanD* plans, int32_t lds_doubles, double* x, const FusedIO* iop) {
  const FusedIO io = iop ? *iop : FusedIO();
  if (nsub <= 0) return;
  const size_t shm = (size_t)lds_doubles * sizeof(double);
  if (std::getenv("HYMLS_MI_FUSED_PROF")) {
    // development aid: per-phase wall-clock ticks (100 MHz) of every workgroup, averaged, on stderr
    long long* dprof = (long long*)alloc((size_t)nsub * 8 * sizeof(long long));
    if (shm > 64 * 1024) HIP_CHECK(hipFuncSetAttribute((const void*)k_interior_fused<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    hipLaunchKernelGGL((k_interior_fused<true>), dim3(nsub), dim3(256), shm, g_stream, subs, plans, x, dprof, io);
    launch_check();
    std::vector<long long> h((size_t)nsub * 8);
    d2h(h.data(), dprof, h.size() * sizeof(long long));
    free(dprof);
    double sum[6] = {0, 0, 0, 0, 0, 0};
    long long tmin = h[6], tmax = 0;
    for (int b = 0; b < nsub; b++) { for (int q = 0; q < 6; q++) sum[q] += (double)h[(size_t)b * 8 + q]; tmin = std::min(tmin, h[(size_t)b * 8 + 6]); tmax = std::max(tmax, h[(size_t)b * 8 + 6] + h[(size_t)b * 8 + 5]); }
    std::fprintf(stderr, "[hymls_mi] fused solve: %d workgroups, kernel span %.1f us; mean per workgroup (us): setup %.1f | fwd assembly %.1f | fwd panels %.1f | bwd panels (wide levels) %.1f | bwd panels (k-split levels) %.1f | total %.1f\n",
                 nsub, (tmax - tmin) / 100.0, sum[0] / nsub / 100, sum[1] / nsub / 100, sum[2] / nsub / 100, sum[3] / nsub / 100, sum[4] / nsub / 100, sum[5] / nsub / 100);
    return;
  }
  if (shm > 64 * 1024) HIP_CHECK(hipFuncSetAttribute((const void*)k_interior_fused<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
  hipLaunchKernelGGL(k_interior_fused<false>, dim3(nsub), dim3(256), shm, g_stream, subs, plans, x, (long long*)nullptr, io);
  launch_check();
}

// ------------------------------------------------------------------ multi-vector variants (several right-hand sides)
// ApplyInverse with nvec > 1 (Epetra_MultiVector; the reference sizes its containers for numvec, src/HYMLS_MatrixBlock.cpp:
// 335-344): the factor panels, the dominant bytes of a solve, are streamed ONCE for a group of NV columns -- every panel
// entry loaded is used for NV multiply-adds.  Vectors are column-major with a leading dimension; the kernels are the
// single-vector ones with the per-row state (accumulators, LDS vectors) replicated NV times.  A launcher picks the
// widest group (4, 2, 1) whose LDS fits and walks over the columns.  (Measured and not kept: 16 / 8 panel entries requested per
// thread ahead of their use instead of 4 -- the 4-column kernel got 10 % slower; it is not short of bytes in flight.)
constexpr size_t LDS_LIMIT_BYTES = 160 * 1024;
// widest column group of a launcher (development switch: HYMLS_MI_MV_GROUP_<FUSED|LVL|BLK> = 1, 2 or 4)
static int mv_group_cap(const char* which) {
  const char* e = std::getenv((std::string("HYMLS_MI_MV_GROUP_") + which).c_str());
  return e ? std::max(1, std::atoi(e)) : NV_MAX;
}

template <int NV>
__global__ void __launch_bounds__(256) k_interior_fused_mv(const FusedSub* __restrict__ subs, const PlanD* __restrict__ plans,
                                                            double* __restrict__ x, int64_t ldx) {
  extern __shared__ double lds[];
  const FusedSub S = subs[blockIdx.x];
  const PlanD P = plans[S.cls];
  const int nI = P.nI, CS = P.contrib_size, FS = P.max_level_rows > 384 ? P.max_level_rows : 384;
  double* X = lds;                       // [NV][nI]
  double* C = X + NV * nI;               // [NV][CS]
  double* Fv = C + NV * CS;              // [NV][FS]   (reduction scratch of the k-split levels inside, at +128)
  FusedFront* LF = (FusedFront*)(Fv + NV * FS);
  const int tid = threadIdx.x;
  for (int i = tid; i < P.nfronts; i += 256) {
    const FrontD G = P.fronts[i];
    FusedFront f;
    f.c0 = G.c0; f.w = G.w; f.ri = G.ri; f.c_off = G.c_off; f.a_off = G.a_off; f.lf_off = G.lf_off; f.idx_off = G.idx_off; f.pad = 0;
    f.lp_off = G.lp_off; f.q_off = G.q_off;
    LF[i] = f;
  }
  double* xg = x + S.xoff;
#pragma unroll
  for (int v = 0; v < NV; v++)
    for (int i = tid; i < nI; i += 256) X[v * nI + i] = xg[v * ldx + i];
  __syncthreads();
  const gptr<double> fac = as_global(S.fac);            // (pointers out of structures: global memory, see as_global)
  const gptr<int32_t> fw_items = as_global(P.fw_items), bw_items = as_global(P.bw_items), fidx = as_global(P.fidx),
                      asm_ptr = as_global(P.asm_ptr), asm_src = as_global(P.asm_src), fw_ptr = as_global(P.fw_ptr), bw_ptr = as_global(P.bw_ptr);
  typedef int v4i __attribute__((ext_vector_type(4)));
  const gptr<v4i> fw_rec = (gptr<v4i>)P.fw_rec;
  // ---------------- forward
  for (int lev = 0; lev < P.nlev; lev++) {
    const int ib = fw_ptr[lev], ni = fw_ptr[lev + 1] - ib;
    for (int it = tid; it < ni; it += 256) {
      FwRec rc;
      { const v4i q = fw_rec[ib + it]; __builtin_memcpy(&rc, &q, 16); }
      const int item = rc.item;
      const FusedFront& F = LF[item >> 16];
      const int r = item & 0xffff;
#pragma unroll
      for (int v = 0; v < NV; v++) {
        const double* Cv = C + v * CS;
        double val = r < F.w ? X[v * nI + F.c0 + r] : 0.0;
        if (rc.n != 0xffff) {
#pragma unroll
          for (int q = 0; q < 5; q++) if (q < rc.n) val += Cv[rc.s[q]];
        } else {
          for (int t = asm_ptr[F.a_off + r]; t < asm_ptr[F.a_off + r + 1]; t++) val += Cv[asm_src[t]];
        }
        if (r < F.w) Fv[v * FS + F.lf_off + r] = val; else C[v * CS + F.c_off + r - F.w] = val;
      }
    }
    __syncthreads();
    if (ni > 128) {
      for (int it = tid; it < ni; it += 256) {
        const int item = fw_items[ib + it];
        const FusedFront& F = LF[item >> 16];
        const int r = item & 0xffff, w = F.w;
        int c1, tri;
        const gptr<double> p = lside(fac + F.lp_off, P.packed, r, w, F.ri, c1, tri);
        const double* f = Fv + F.lf_off;
        const int kmax = r < w ? r : w;
        double a[4][NV];    // (the accumulation order of the single-vector kernel: bitwise the same column)
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
          for (int v = 0; v < NV; v++) a[u][v] = 0.0;
        int k = 0;
        for (; k + 3 < kmax; k += 4) {
          double l[4];
#pragma unroll
          for (int u = 0; u < 4; u++) l[u] = p[c1 * (k + u) - tri * (((k + u) * (k + u + 3)) >> 1)];
#pragma unroll
          for (int u = 0; u < 4; u++)
#pragma unroll
            for (int v = 0; v < NV; v++) a[u][v] += l[u] * f[v * FS + k + u];
        }
        for (; k < kmax; k++) {
          const double l = p[c1 * k - tri * ((k * (k + 3)) >> 1)];
#pragma unroll
          for (int v = 0; v < NV; v++) a[0][v] += l * f[v * FS + k];
        }
#pragma unroll
        for (int v = 0; v < NV; v++) {
          const double sum = (a[0][v] + a[1][v]) + (a[2][v] + a[3][v]);
          if (r < w) X[v * nI + F.c0 + r] = f[v * FS + r] + sum; else C[v * CS + F.c_off + r - w] -= sum;
        }
      }
    } else {
      const int RT = ni > 64 ? 128 : 64, KG = 256 / RT;
      const int it = tid % RT, kg = tid / RT;
      double a0[NV], a1[NV], a2[NV], a3[NV];
#pragma unroll
      for (int v = 0; v < NV; v++) { a0[v] = 0.0; a1[v] = 0.0; a2[v] = 0.0; a3[v] = 0.0; }
      if (it < ni) {
        const int item = fw_items[ib + it];
        const FusedFront& F = LF[item >> 16];
        const int r = item & 0xffff, w = F.w;
        int c1, tri;
        const gptr<double> p = lside(fac + F.lp_off, P.packed, r, w, F.ri, c1, tri);
        const double* f = Fv + F.lf_off;
        const int kmax = r < w ? r : w;
        auto at = [&](int kk) { return p[c1 * kk - tri * ((kk * (kk + 3)) >> 1)]; };
        int k = kg;
        for (; k + 3 * KG < kmax; k += 4 * KG) {
          const double l0 = at(k), l1 = at(k + KG), l2 = at(k + 2 * KG), l3 = at(k + 3 * KG);
#pragma unroll
          for (int v = 0; v < NV; v++) {
            a0[v] += l0 * f[v * FS + k]; a1[v] += l1 * f[v * FS + k + KG];
            a2[v] += l2 * f[v * FS + k + 2 * KG]; a3[v] += l3 * f[v * FS + k + 3 * KG];
          }
        }
        for (; k < kmax; k += KG) {
          const double l = at(k);
#pragma unroll
          for (int v = 0; v < NV; v++) a0[v] += l * f[v * FS + k];
        }
      }
#pragma unroll
      for (int v = 0; v < NV; v++) Fv[v * FS + 128 + kg * RT + it] = (a0[v] + a1[v]) + (a2[v] + a3[v]);
      __syncthreads();
      if (tid < ni) {
        const int item = fw_items[ib + tid];
        const FusedFront& F = LF[item >> 16];
        const int r = item & 0xffff;
#pragma unroll
        for (int v = 0; v < NV; v++) {
          double sum = 0.0;
          for (int g = 0; g < KG; g++) sum += Fv[v * FS + 128 + g * RT + tid];
          if (r < F.w) X[v * nI + F.c0 + r] = Fv[v * FS + F.lf_off + r] + sum; else C[v * CS + F.c_off + r - F.w] -= sum;
        }
      }
    }
    __syncthreads();
  }
  // ---------------- backward
  for (int lev = P.nlev - 1; lev >= 0; lev--) {
    const int ib = bw_ptr[lev], ni = bw_ptr[lev + 1] - ib;
    if (ni > 128) {
      for (int it = tid; it < ni; it += 256) {
        const int item = bw_items[ib + it];
        const FusedFront& F = LF[item >> 16];
        const int i = item & 0xffff, w = F.w, ri = F.ri;
        int c1, tri;
        const gptr<double> p = uside(fac + F.lp_off, P.packed, i, w, ri, c1, tri);
        const double* Xs = X + F.c0;
        double a[4][NV];
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
          for (int v = 0; v < NV; v++) a[u][v] = 0.0;
        int k = i;
        for (; k + 3 < w; k += 4) {
          double l[4];
#pragma unroll
          for (int u = 0; u < 4; u++) l[u] = p[c1 * (k + u) + tri * (((k + u) * (k + u + 1)) >> 1)];
#pragma unroll
          for (int u = 0; u < 4; u++)
#pragma unroll
            for (int v = 0; v < NV; v++) a[u][v] += l[u] * Xs[v * nI + k + u];
        }
        for (; k < w; k++) {
          const double l = p[c1 * k + tri * ((k * (k + 1)) >> 1)];
#pragma unroll
          for (int v = 0; v < NV; v++) a[0][v] += l * Xs[v * nI + k];
        }
        const gptr<double> qv = fac + F.q_off + i;
        const gptr<int32_t> idx = fidx + F.idx_off + w;
        k = 0;
        for (; k + 3 < ri; k += 4) {
          double l[4]; int id[4];
#pragma unroll
          for (int u = 0; u < 4; u++) { l[u] = qv[(int64_t)w * (k + u)]; id[u] = idx[k + u]; }
#pragma unroll
          for (int u = 0; u < 4; u++)
#pragma unroll
            for (int v = 0; v < NV; v++) a[u][v] -= l[u] * X[v * nI + id[u]];
        }
        for (; k < ri; k++) {
          const double l = qv[(int64_t)w * k]; const int id = idx[k];
#pragma unroll
          for (int v = 0; v < NV; v++) a[0][v] -= l * X[v * nI + id];
        }
#pragma unroll
        for (int v = 0; v < NV; v++) Fv[v * FS + it] = (a[0][v] + a[1][v]) + (a[2][v] + a[3][v]);
      }
      __syncthreads();
      for (int it = tid; it < ni; it += 256) {
        const int item = bw_items[ib + it];
        const int dst = LF[item >> 16].c0 + (item & 0xffff);
#pragma unroll
        for (int v = 0; v < NV; v++) X[v * nI + dst] = Fv[v * FS + it];
      }
    } else {
      const int RT = ni > 64 ? 128 : 64, KG = 256 / RT;
      const int it = tid % RT, kg = tid / RT;
      double a0[NV], a1[NV], a2[NV], a3[NV];
#pragma unroll
      for (int v = 0; v < NV; v++) { a0[v] = 0.0; a1[v] = 0.0; a2[v] = 0.0; a3[v] = 0.0; }
      if (it < ni) {
        const int item = bw_items[ib + it];
        const FusedFront& F = LF[item >> 16];
        const int i = item & 0xffff, w = F.w, ri = F.ri;
        int c1, tri;
        const gptr<double> p = uside(fac + F.lp_off, P.packed, i, w, ri, c1, tri);
        const double* Xs = X + F.c0;
        auto at = [&](int kk) { return p[c1 * kk + tri * ((kk * (kk + 1)) >> 1)]; };
        int k = i + kg;
        for (; k + 3 * KG < w; k += 4 * KG) {
          const double l0 = at(k), l1 = at(k + KG), l2 = at(k + 2 * KG), l3 = at(k + 3 * KG);
#pragma unroll
          for (int v = 0; v < NV; v++) {
            a0[v] += l0 * Xs[v * nI + k]; a1[v] += l1 * Xs[v * nI + k + KG];
            a2[v] += l2 * Xs[v * nI + k + 2 * KG]; a3[v] += l3 * Xs[v * nI + k + 3 * KG];
          }
        }
        for (; k < w; k += KG) {
          const double l = at(k);
#pragma unroll
          for (int v = 0; v < NV; v++) a0[v] += l * Xs[v * nI + k];
        }
        const gptr<double> qv = fac + F.q_off + i;
        const gptr<int32_t> idx = fidx + F.idx_off + w;
        k = kg;
        for (; k + 3 * KG < ri; k += 4 * KG) {
          const double q0 = qv[(int64_t)w * k], q1 = qv[(int64_t)w * (k + KG)], q2 = qv[(int64_t)w * (k + 2 * KG)], q3 = qv[(int64_t)w * (k + 3 * KG)];
          const int i0 = idx[k], i1 = idx[k + KG], i2 = idx[k + 2 * KG], i3 = idx[k + 3 * KG];
#pragma unroll
          for (int v = 0; v < NV; v++) {
            a0[v] -= q0 * X[v * nI + i0]; a1[v] -= q1 * X[v * nI + i1]; a2[v] -= q2 * X[v * nI + i2]; a3[v] -= q3 * X[v * nI + i3];
          }
        }
        for (; k < ri; k += KG) {
          const double q0 = qv[(int64_t)w * k]; const int i0 = idx[k];
#pragma unroll
          for (int v = 0; v < NV; v++) a0[v] -= q0 * X[v * nI + i0];
        }
      }
#pragma unroll
      for (int v = 0; v < NV; v++) Fv[v * FS + 128 + kg * RT + it] = (a0[v] + a1[v]) + (a2[v] + a3[v]);
      __syncthreads();
      if (tid < ni) {
        const int item = bw_items[ib + tid];
        const int dst = LF[item >> 16].c0 + (item & 0xffff);
#pragma unroll
        for (int v = 0; v < NV; v++) {
          double sum = 0.0;
          for (int g = 0; g < KG; g++) sum += Fv[v * FS + 128 + g * RT + tid];
          X[v * nI + dst] = sum;
        }
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int v = 0; v < NV; v++)
    for (int i = tid; i < nI; i += 256) xg[v * ldx + i] = X[v * nI + i];
}

template <int NV>
static void launch_fused_mv(int32_t nsub, const FusedSub* subs, const PlanD* plans, size_t shm, double* x, int64_t ldx) {
  if (shm > 64 * 1024) HIP_CHECK(hipFuncSetAttribute((const void*)k_interior_fused_mv<NV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
  hipLaunchKernelGGL(k_interior_fused_mv<NV>, dim3(nsub), dim3(256), shm, g_stream, subs, plans, x, ldx);
  launch_check();
}
// lds_doubles: LDS need of one vector; front_doubles: the part of it that holds the front descriptors (not replicated)
void interior_solve_fused_mv(int32_t nsub, const FusedSub* subs, const PlanD* plans, int32_t lds_doubles, int32_t front_doubles,
                             double* x, int64_t ldx, int nv) {
  if (nsub <= 0 || nv <= 0) return;
  const size_t per = (size_t)(lds_doubles - front_doubles) * sizeof(double), fixed = (size_t)front_doubles * sizeof(double);
  int v = 0;
  while (v < nv) {
    int g = nv - v >= 4 ? 4 : (nv - v >= 2 ? 2 : 1);
    while (g > 1 && (per * g + fixed > LDS_LIMIT_BYTES || g > mv_group_cap("FUSED"))) g >>= 1;
    double* xv = x + (int64_t)v * ldx;
    if (g == 4) launch_fused_mv<4>(nsub, subs, plans, per * 4 + fixed, xv, ldx);
    else if (g == 2) launch_fused_mv<2>(nsub, subs, plans, per * 2 + fixed, xv, ldx);
    else interior_solve_fused(nsub, subs, plans, lds_doubles, xv, nullptr);
    v += g;
  }
}

// merged level-synchronous solve, NV columns (k_lvl_fwd / k_lvl_bwd with the LDS vectors and accumulators replicated;
// the contribution vectors of column v live cstride doubles behind those of column v - 1)
template <int NV>
__global__ void __launch_bounds__(256) k_lvl_fwd_mv(const LvlTask* __restrict__ tasks, const LvlSub* __restrict__ subs,
                                                     const PlanD* __restrict__ plans, const double* __restrict__ x,
                                                     double* __restrict__ y, int64_t ld, int32_t LS, int32_t c0) {
  extern __shared__ double f[];     // [NV][LS]
  const LvlTask T = tasks[blockIdx.x];
  const LvlSub S = subs[T.sub];
  const PlanD* P = plans + S.cls;
  const FrontD F = P->fronts[T.front];
  const int tid = threadIdx.x, w = F.w, rows = F.w + F.ri;
  const int64_t ldp = rows;
  const double* xb = x + S.xoff;
  double* yb = y + S.xoff;
  const int64_t cs = S.cstride;
  const gmptr<double> cb = as_global_rw(S.contrib) + (int64_t)c0 * cs;   // contribution vectors of the columns c0 .. c0 + NV - 1 of the whole block
  const gptr<int32_t> aptr = as_global(P->asm_ptr) + F.a_off;
  const gptr<int32_t> asrc = as_global(P->asm_src);
  const gptr<double> Lp = as_global(S.fac) + F.lp_off;
  if (T.r0 < 0) {
    for (int j = tid; j < rows; j += 256) {
#pragma unroll
      for (int v = 0; v < NV; v++) {
        double val = j < w ? xb[v * ld + F.c0 + j] : 0.0;
        for (int t = aptr[j]; t < aptr[j + 1]; t++) val += cb[v * cs + asrc[t]];
        f[v * LS + j] = val;
      }
    }
    __syncthreads();
    for (int i = tid; i < rows; i += 256) {
      const int kmax = i < w ? i : w;
      double s0[NV], s1[NV], s2[NV], s3[NV];   // (the accumulation order of the single-vector kernel)
#pragma unroll
      for (int v = 0; v < NV; v++) { s0[v] = 0.0; s1[v] = 0.0; s2[v] = 0.0; s3[v] = 0.0; }
      int k = 0;
      for (; k + 3 < kmax; k += 4) {
        const double l0 = Lp[i + ldp * k], l1 = Lp[i + ldp * (k + 1)], l2 = Lp[i + ldp * (k + 2)], l3 = Lp[i + ldp * (k + 3)];
#pragma unroll
        for (int v = 0; v < NV; v++) {
          s0[v] += l0 * f[v * LS + k]; s1[v] += l1 * f[v * LS + k + 1]; s2[v] += l2 * f[v * LS + k + 2]; s3[v] += l3 * f[v * LS + k + 3];
        }
      }
      for (; k < kmax; k++) {
        const double l0 = Lp[i + ldp * k];
#pragma unroll
        for (int v = 0; v < NV; v++) s0[v] += l0 * f[v * LS + k];
      }
#pragma unroll
      for (int v = 0; v < NV; v++) {
        const double s = (s0[v] + s1[v]) + (s2[v] + s3[v]);
        if (i < w) yb[v * ld + F.c0 + i] = f[v * LS + i] + s;
        else cb[v * cs + F.c_off + i - w] = f[v * LS + i] - s;
      }
    }
    return;
  }
  const int r0 = T.r0, lane = tid & 63, g = tid >> 6, i = r0 + lane;
  const int kneed = min(w, r0 + 63);
  const int KP = (kneed + 7) & ~7;
  // per vector: [KP assembled pivots | 64 own rows | 4 x 64 reduction]
  for (int j = tid; j < kneed; j += 256) {
#pragma unroll
    for (int v = 0; v < NV; v++) {
      double val = xb[v * ld + F.c0 + j];
      for (int t = aptr[j]; t < aptr[j + 1]; t++) val += cb[v * cs + asrc[t]];
      f[v * LS + j] = val;
    }
  }
  if (tid < 64 && i < rows) {
#pragma unroll
    for (int v = 0; v < NV; v++) {
      double val = i < w ? xb[v * ld + F.c0 + i] : 0.0;
      for (int t = aptr[i]; t < aptr[i + 1]; t++) val += cb[v * cs + asrc[t]];
      f[v * LS + KP + lane] = val;
    }
  }
  __syncthreads();
  const int krow = i < rows ? (i < w ? i : w) : 0;
  const gptr<double> Lr = Lp + (i < rows ? i : 0);
  const int chunk = ((kneed + 31) / 32) * 8;
  const int kb = g * chunk, ke = min(kb + chunk, kneed);
  double acc[8][NV];
#pragma unroll
  for (int u = 0; u < 8; u++)
#pragma unroll
    for (int v = 0; v < NV; v++) acc[u][v] = 0.0;
  int k = kb;
  for (; k + 7 < ke; k += 8) {
    double l[8];
#pragma unroll
    for (int u = 0; u < 8; u++) l[u] = Lr[ldp * (k + u)];
#pragma unroll
    for (int u = 0; u < 8; u++)
      if (k + u < krow) {
#pragma unroll
        for (int v = 0; v < NV; v++) acc[u][v] += l[u] * f[v * LS + k + u];
      }
  }
  for (; k < ke; k++) if (k < krow) {
    const double l = Lr[ldp * k];
#pragma unroll
    for (int v = 0; v < NV; v++) acc[0][v] += l * f[v * LS + k];
  }
#pragma unroll
  for (int v = 0; v < NV; v++)
    f[v * LS + KP + 64 + g * 64 + lane] = ((acc[0][v] + acc[1][v]) + (acc[2][v] + acc[3][v])) + ((acc[4][v] + acc[5][v]) + (acc[6][v] + acc[7][v]));
  __syncthreads();
  if (g == 0 && i < rows) {
#pragma unroll
    for (int v = 0; v < NV; v++) {
      const double* red = f + v * LS + KP + 64;
      const double sum = (red[lane] + red[64 + lane]) + (red[128 + lane] + red[192 + lane]);
      const double own = f[v * LS + KP + lane];
      if (i < w) yb[v * ld + F.c0 + i] = own + sum;
      else cb[v * cs + F.c_off + i - w] = own - sum;
    }
  }
}

template <int NV>
__global__ void __launch_bounds__(256) k_lvl_bwd_mv(const LvlTask* __restrict__ tasks, const LvlSub* __restrict__ subs,
                                                     const PlanD* __restrict__ plans, const double* __restrict__ y,
                                                     double* __restrict__ x, int64_t ld, int32_t LS) {
  extern __shared__ double f[];
  const LvlTask T = tasks[blockIdx.x];
  const LvlSub S = subs[T.sub];
  const PlanD* P = plans + S.cls;
  const FrontD F = P->fronts[T.front];
  const int tid = threadIdx.x, w = F.w, ri = F.ri;
  const int64_t ldp = w + ri;
  double* xb = x + S.xoff;
  const double* yb = y + S.xoff;
  const gptr<int32_t> idx = as_global(P->fidx) + F.idx_off + w;
  const gptr<double> Lp = as_global(S.fac) + F.lp_off;
  const gptr<double> Q = as_global(S.fac) + F.q_off;
  if (T.r0 < 0) {
    for (int k = tid; k < w + ri; k += 256) {
#pragma unroll
      for (int v = 0; v < NV; v++) f[v * LS + k] = k < w ? yb[v * ld + F.c0 + k] : xb[v * ld + idx[k - w]];
    }
    __syncthreads();
    for (int i = tid; i < w; i += 256) {
      double s[NV], s0[NV], s1[NV], s2[NV], s3[NV];
#pragma unroll
      for (int v = 0; v < NV; v++) { s[v] = 0.0; s0[v] = 0.0; s1[v] = 0.0; s2[v] = 0.0; s3[v] = 0.0; }
      for (int k = i; k < w; k++) {
        const double l = Lp[i + ldp * k];
#pragma unroll
        for (int v = 0; v < NV; v++) s[v] += l * f[v * LS + k];
      }
      int k = 0;
      for (; k + 3 < ri; k += 4) {
        const double q0 = Q[i + (int64_t)w * k], q1 = Q[i + (int64_t)w * (k + 1)], q2 = Q[i + (int64_t)w * (k + 2)], q3 = Q[i + (int64_t)w * (k + 3)];
#pragma unroll
        for (int v = 0; v < NV; v++) {
          s0[v] += q0 * f[v * LS + w + k]; s1[v] += q1 * f[v * LS + w + k + 1]; s2[v] += q2 * f[v * LS + w + k + 2]; s3[v] += q3 * f[v * LS + w + k + 3];
        }
      }
      for (; k < ri; k++) {
        const double q0 = Q[i + (int64_t)w * k];
#pragma unroll
        for (int v = 0; v < NV; v++) s0[v] += q0 * f[v * LS + w + k];
      }
#pragma unroll
      for (int v = 0; v < NV; v++) xb[v * ld + F.c0 + i] = s[v] - ((s0[v] + s1[v]) + (s2[v] + s3[v]));
    }
    return;
  }
  const int r0 = T.r0, lane = tid & 63, g = tid >> 6, i = r0 + lane;
  const int nU = w - r0, total = nU + ri;
  const int TP = (total + 7) & ~7;
  for (int k = tid; k < total; k += 256) {
#pragma unroll
    for (int v = 0; v < NV; v++) f[v * LS + k] = k < nU ? yb[v * ld + F.c0 + r0 + k] : xb[v * ld + idx[k - nU]];
  }
  __syncthreads();
  const int iv = i < w ? i : r0;
  const gptr<double> Lr = Lp + iv + ldp * r0;
  const gptr<double> Qr = Q + iv;
  const int chunk = ((total + 31) / 32) * 8;
  const int kb = g * chunk, ke = min(kb + chunk, total);
  double acc[8][NV];
#pragma unroll
  for (int u = 0; u < 8; u++)
#pragma unroll
    for (int v = 0; v < NV; v++) acc[u][v] = 0.0;
  {
    const int e = min(ke, nU);
    int k = kb;
    for (; k + 7 < e; k += 8) {
      double l[8];
#pragma unroll
      for (int u = 0; u < 8; u++) l[u] = Lr[ldp * (k + u)];
#pragma unroll
      for (int u = 0; u < 8; u++)
        if (k + u >= lane) {
#pragma unroll
          for (int v = 0; v < NV; v++) acc[u][v] += l[u] * f[v * LS + k + u];
        }
    }
    for (; k < e; k++) if (k >= lane) {
      const double l = Lr[ldp * k];
#pragma unroll
      for (int v = 0; v < NV; v++) acc[0][v] += l * f[v * LS + k];
    }
  }
  {
    int k = max(kb, nU);
    for (; k + 7 < ke; k += 8) {
      double l[8];
#pragma unroll
      for (int u = 0; u < 8; u++) l[u] = Qr[(int64_t)w * (k - nU + u)];
#pragma unroll
      for (int u = 0; u < 8; u++)
#pragma unroll
        for (int v = 0; v < NV; v++) acc[u][v] -= l[u] * f[v * LS + k + u];
    }
    for (; k < ke; k++) {
      const double l = Qr[(int64_t)w * (k - nU)];
#pragma unroll
      for (int v = 0; v < NV; v++) acc[0][v] -= l * f[v * LS + k];
    }
  }
#pragma unroll
  for (int v = 0; v < NV; v++)
    f[v * LS + TP + g * 64 + lane] = ((acc[0][v] + acc[1][v]) + (acc[2][v] + acc[3][v])) + ((acc[4][v] + acc[5][v]) + (acc[6][v] + acc[7][v]));
  __syncthreads();
  if (g == 0 && i < w) {
#pragma unroll
    for (int v = 0; v < NV; v++) {
      const double* red = f + v * LS + TP;
      xb[v * ld + F.c0 + i] = (red[lane] + red[64 + lane]) + (red[128 + lane] + red[192 + lane]);
    }
  }
}

template <int NV>
static void launch_lvl_mv(bool fwd, const LvlTask* tasks, int32_t ntasks, const LvlSub* subs, const PlanD* plans, int32_t ls,
                          const double* a, double* b, int64_t ld, int32_t c0) {
  const size_t shm = (size_t)ls * NV * sizeof(double);
  if (fwd) {
    if (shm > 64 * 1024) HIP_CHECK(hipFuncSetAttribute((const void*)k_lvl_fwd_mv<NV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    hipLaunchKernelGGL(k_lvl_fwd_mv<NV>, dim3(ntasks), dim3(256), shm, g_stream, tasks, subs, plans, a, b, ld, ls, c0);
  } else {
    if (shm > 64 * 1024) HIP_CHECK(hipFuncSetAttribute((const void*)k_lvl_bwd_mv<NV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    hipLaunchKernelGGL(k_lvl_bwd_mv<NV>, dim3(ntasks), dim3(256), shm, g_stream, tasks, subs, plans, a, b, ld, ls);
  }
  launch_check();
}
static void lvl_tasks_mv(bool fwd, const LvlTask* tasks, int32_t ntasks, const LvlSub* subs, const PlanD* plans, int32_t lds_doubles,
                         const double* a, double* b, int64_t ld, int nv) {
  if (ntasks <= 0 || nv <= 0) return;
  int v = 0;
  while (v < nv) {
    int g = nv - v >= 4 ? 4 : (nv - v >= 2 ? 2 : 1);
    while (g > 1 && ((size_t)lds_doubles * g * sizeof(double) > LDS_LIMIT_BYTES || g > mv_group_cap("LVL"))) g >>= 1;
    const double* av = a + (int64_t)v * ld;
    double* bv = b + (int64_t)v * ld;
    // (every column of the block keeps its own contribution vectors between the tree levels: slot = column index)
    if (g == 4) launch_lvl_mv<4>(fwd, tasks, ntasks, subs, plans, lds_doubles, av, bv, ld, v);
    else if (g == 2) launch_lvl_mv<2>(fwd, tasks, ntasks, subs, plans, lds_doubles, av, bv, ld, v);
    else if (!fwd) solve_bwd_tasks(tasks, ntasks, subs, plans, lds_doubles, av, bv);    // (the backward sweep reads no contributions)
    else if (v == 0) solve_fwd_tasks(tasks, ntasks, subs, plans, lds_doubles, av, bv);
    else launch_lvl_mv<1>(fwd, tasks, ntasks, subs, plans, lds_doubles, av, bv, ld, v);
    v += g;
  }
}
void solve_fwd_tasks_mv(const LvlTask* tasks, int32_t ntasks, const LvlSub* subs, const PlanD* plans, int32_t lds_doubles,
                        const double* x, double* y, int64_t ld, int nv) {
  lvl_tasks_mv(true, tasks, ntasks, subs, plans, lds_doubles, x, y, ld, nv);
}
void solve_bwd_tasks_mv(const LvlTask* tasks, int32_t ntasks, const LvlSub* subs, const PlanD* plans, int32_t lds_doubles,
                        const double* y, double* x, int64_t ld, int nv) {
  lvl_tasks_mv(false, tasks, ntasks, subs, plans, lds_doubles, y, x, ld, nv);
}

// ------------------------------------------------------------------ separator-side kernels
// eight lanes per group (the groups have 8 nodes on average; consecutive groups are contiguous, so a wave still
// reads one contiguous stretch): dot product by shuffle reduction inside the 8 lanes, then the axpy
__global__ void __launch_bounds__(256) k_ot(int32_t ng, const int32_t* __restrict__ gptr, const double* __restrict__ w, double* __restrict__ x) {
  const int g = blockIdx.x * 32 + (threadIdx.x >> 3);
  const int lane = threadIdx.x & 7;
  const bool on = g < ng;
  const int b = on ? gptr[g] : 0, e = on ? gptr[g + 1] : 0;
  double s = 0.0;
  for (int i = b + lane; i < e; i += 8) s += w[i] * x[i];
#pragma unroll
  for (int off = 4; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  for (int i = b + lane; i < e; i += 8) x[i] = 2.0 * w[i] * s - x[i];
}
void ot_apply(int32_t ng, const int32_t* gptr, const double* w, double* x) {
  if (ng <= 0) return;
  hipLaunchKernelGGL(k_ot, dim3((ng + 31) / 32), dim3(256), 0, g_stream, ng, gptr, w, x); launch_check();
}

// Householder data of one group from the test vector slice (reference Householder::Apply):
// returns false if the transform is the identity.
__device__ inline bool hh_setup(const double* v, int n, double& sg, double& nrm, double& v1, double& fac1) {
  sg = v[0] < 0 ? -1.0 : (v[0] > 0 ? 1.0 : 0.0);
  double s = 0.0;
  for (int i = 0; i < n; i++) s += v[i] * v[i];
  nrm = sqrt(s) * fabs(sg);
  v1 = sg * v[0] + nrm;
  if (fabs(v1) < 1e-14 || nrm < 1e-14) return false;
  fac1 = 1.0 / (nrm * v1);
  return true;
}
// pass 0: rows of every group (one thread per column); pass 1: columns (one thread per row)
template <int PASS>
__global__ void __launch_bounds__(256) k_sblock_hh(int32_t nS, int32_t ng, const int32_t* __restrict__ gptr,
                                                    const double* __restrict__ tv, double* __restrict__ sblock) {
  const int slot = blockIdx.y;
  double* S = sblock + (int64_t)slot * nS * nS;
  const double* v = tv + (int64_t)slot * nS;
  const int c = blockIdx.x * blockDim.x + threadIdx.x;  // column (pass 0) or row (pass 1)
  if (c >= nS) return;
  const int64_t inc = PASS == 0 ? 1 : nS;               // stride along the transformed index
  double* base = PASS == 0 ? S + (int64_t)nS * c : S + c;
  for (int g = 0; g < ng; g++) {
    const int pos = gptr[g], n = gptr[g + 1] - pos;
    if (n <= 0) continue;
    double sg, nrm, v1, fac1;
    if (!hh_setup(v + pos, n, sg, nrm, v1, fac1)) continue;
    double* p = base + inc * pos;
    double fac2 = nrm * p[0];
    for (int i = 0; i < n; i++) fac2 += p[inc * i] * (sg * v[pos + i]);
    const double fac = fac1 * fac2;
    p[0] = v1 * fac - p[0];
    for (int i = 1; i < n; i++) p[inc * i] = (sg * v[pos + i]) * fac - p[inc * i];
  }
}
// pass 0 with the lanes along the rows of a group: task = (group, column), eight lanes per task (groups have about 8
// nodes), consecutive tasks = consecutive columns; every column segment is read and written once, 64 contiguous bytes
// at a time (the thread-per-column version walked down the columns: one cache line per lane and load)
__global__ void __launch_bounds__(256) k_sblock_hh_rows(int32_t nS, int32_t ng, const int32_t* __restrict__ gptr,
                                                         const double* __restrict__ tv, double* __restrict__ sblock) {
  const int slot = blockIdx.y;
  double* S = sblock + (int64_t)slot * nS * nS;
  const double* v = tv + (int64_t)slot * nS;
  const int64_t t = (int64_t)blockIdx.x * 32 + (threadIdx.x >> 3);
  const int lane = threadIdx.x & 7;
  const bool on = t < (int64_t)ng * nS;
  const int g = on ? (int)(t / nS) : 0, c = on ? (int)(t % nS) : 0;
  const int pos = gptr[g], n = on ? gptr[g + 1] - pos : 0;
  // Householder data of the group (hh_setup), reduced over the eight lanes
  const double v0 = n > 0 ? v[pos] : 0.0;
  const double sg = v0 < 0 ? -1.0 : (v0 > 0 ? 1.0 : 0.0);
  double ss = 0.0;
  for (int i = lane; i < n; i += 8) ss += v[pos + i] * v[pos + i];
#pragma unroll
  for (int off = 4; off > 0; off >>= 1) ss += __shfl_xor(ss, off, 64);
  const double nrm = sqrt(ss) * fabs(sg);
  const double v1 = sg * v0 + nrm;
  const bool act = n > 0 && !(fabs(v1) < 1e-14 || nrm < 1e-14);
  double* p = S + (int64_t)nS * c + pos;
  double f2 = 0.0;
  if (act) for (int i = lane; i < n; i += 8) f2 += p[i] * (i == 0 ? nrm + sg * v0 : sg * v[pos + i]);
#pragma unroll
  for (int off = 4; off > 0; off >>= 1) f2 += __shfl_xor(f2, off, 64);
  if (!act) return;
  const double fac = f2 / (nrm * v1);
  for (int i = lane; i < n; i += 8) p[i] = (i == 0 ? v1 : sg * v[pos + i]) * fac - p[i];
}
void sblock_transform(int32_t nS, int32_t ng, const int32_t* gptr, const double* tv, double* sblock, int32_t nbc) {
  if (nS <= 0 || nbc <= 0) return;
  for (int s0 = 0; s0 < nbc; s0 += 65535) {
    const int ns = std::min(65535, nbc - s0);
    hipLaunchKernelGGL(k_sblock_hh_rows, dim3((unsigned)(((int64_t)ng * nS + 31) / 32), ns), dim3(256), 0, g_stream, nS, ng, gptr,
                       tv + (int64_t)s0 * nS, sblock + (int64_t)s0 * nS * nS);
    launch_check();
    hipLaunchKernelGGL(k_sblock_hh<1>, dim3((nS + 255) / 256, ns), dim3(256), 0, g_stream, nS, ng, gptr,
                       tv + (int64_t)s0 * nS, sblock + (int64_t)s0 * nS * nS);
    launch_check();
  }
}

__global__ void k_sblock_extract(int32_t nS, int64_t npick, const int32_t* __restrict__ pick, const double* __restrict__ sblock,
                                 double* __restrict__ out, int64_t out_stride) {
  const int slot = blockIdx.y;
  const double* S = sblock + (int64_t)slot * nS * nS;
  for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < npick; k += (int64_t)gridDim.x * blockDim.x)
    out[(int64_t)slot * out_stride + k] = S[pick[k]];
}
void sblock_extract(int32_t nS, int64_t npick, const int32_t* pick, const double* sblock, double* out, int64_t out_stride, int32_t nbc) {
  if (npick <= 0 || nbc <= 0) return;
  for (int s0 = 0; s0 < nbc; s0 += 65535) {
    const int ns = std::min(65535, nbc - s0);
    hipLaunchKernelGGL(k_sblock_extract, dim3(nblocks(npick, 256, 256), ns), dim3(256), 0, g_stream, nS, npick, pick,
                       sblock + (int64_t)s0 * nS * nS, out + (int64_t)s0 * out_stride, out_stride);
    launch_check();
  }
}

// ---- kept entries of the transformed separator block in ONE read pass (AssembleTransformAndDrop / ConstructSCPart of the
// reference, src/HYMLS_SchurPreconditioner.cpp:698-986: two-sided Householder per group on the dense matrix, then only the
// V-sum x V-sum entries and the non-V-sum block of every linked set are kept).  With H_g = gam_g u_g u_g' + bet_g I
// (bet = -1, gam = 1 / (nrm v1) for an active transform; bet = +1, gam = 0 for the identity, reference
// src/HYMLS_Householder.cpp:38-80) the kept entries of the block of row group I and column group J are
//   T[a,b] = gam_I gam_J u_I[a] t u_J[b] + gam_I bet_J u_I[a] r_b + bet_I gam_J q_a u_J[b] + bet_I bet_J S[a,b]
//   q = S[I,J] u_J,   r = u_I' S[I,J],   t = u_I' S[I,J] u_J,
// so the transformed matrix never has to be written: a workgroup owns one (slot, column group J), streams the columns
// of J once with the rows on consecutive lanes (q for every row at once), and finishes with segmented sums over the row
// groups.  The separate transform + extract passes read and wrote the nS^2 block four times.
__global__ void __launch_bounds__(256) k_sblock_kept(KeptD K, const double* __restrict__ tv, const double* __restrict__ sblock,
                                                     double* __restrict__ out, int64_t out_stride) {
  extern __shared__ double sh[];           // u[nS] | q[nS] | c0[nS] | gam[ngl] | bet[ngl] | tI[ngl] | r0[ngl]
  const int nS = K.nS, ngl = K.ngl, tid = threadIdx.x;
  const int slot = blockIdx.y, J = blockIdx.x;
  double* u = sh; double* q = u + nS; double* c0 = q + nS;
  double* gam = c0 + nS; double* bet = gam + ngl; double* tI = bet + ngl; double* r0 = tI + ngl;
  const double* S = sblock + (int64_t)slot * nS * nS;
  const double* v = tv + (int64_t)slot * nS;
  double* rec = out + (int64_t)slot * out_stride;
  // Householder data of every group of this slot
  for (int g = tid; g < ngl; g += 256) {
    const int pos = K.gptr[g], n = K.gptr[g + 1] - pos;
    double sg, nrm, v1, fac1;
    const bool act = n > 0 && hh_setup(v + pos, n, sg, nrm, v1, fac1);
    gam[g] = act ? fac1 : 0.0;
    bet[g] = act ? -1.0 : 1.0;
    for (int i = 0; i < n; i++) u[pos + i] = act ? (i == 0 ? v1 : sg * v[pos + i]) : 0.0;
  }
  __syncthreads();
  const int jb = K.gptr[J], nJ = K.gptr[J + 1] - jb;
  // q_i = sum_j S[i, j] u_J[j] for every row i, c0_i = S[i, first column of J]
  for (int i = tid; i < nS; i += 256) {
    const double* __restrict__ p = S + i + (int64_t)nS * jb;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    const double first = p[0];
    int j = 0;
    for (; j + 3 < nJ; j += 4) {
      const double s0 = p[(int64_t)nS * j], s1 = p[(int64_t)nS * (j + 1)], s2 = p[(int64_t)nS * (j + 2)], s3 = p[(int64_t)nS * (j + 3)];
      a0 += s0 * u[jb + j]; a1 += s1 * u[jb + j + 1]; a2 += s2 * u[jb + j + 2]; a3 += s3 * u[jb + j + 3];
    }
    for (; j < nJ; j++) a0 += p[(int64_t)nS * j] * u[jb + j];
    q[i] = (a0 + a1) + (a2 + a3);
    c0[i] = first;
  }
  __syncthreads();
  // segmented sums over the row groups: t_I = u_I' q_I, r0_I = u_I' c0_I
  for (int I = tid; I < ngl; I += 256) {
    const int ib = K.gptr[I], nI = K.gptr[I + 1] - ib;
    double t = 0.0, r = 0.0;
    for (int i = 0; i < nI; i++) { t += u[ib + i] * q[ib + i]; r += u[ib + i] * c0[ib + i]; }
    tI[I] = t; r0[I] = r;
  }
  __syncthreads();
  const double gJ = gam[J], bJ = bet[J], uJ0 = nJ > 0 ? u[jb] : 0.0;
  // V-sum x V-sum entries of column J
  for (int I = tid; I < ngl; I += 256) {
    const int ib = K.gptr[I];
    const double gI = gam[I], bI = bet[I], uI0 = u[ib];
    rec[I + (int64_t)ngl * J] = gI * gJ * uI0 * tI[I] * uJ0 + gI * bJ * uI0 * r0[I] + bI * gJ * q[ib] * uJ0 + bI * bJ * c0[ib];
  }
  // the non-V-sum block of the linked set of J: row groups I of the same set
  const int L = K.glink[J];
  if (L < 0 || nJ < 2) return;
  const int64_t boff = K.lboff[L];
  const int blen = K.lblen[L];
  for (int I = 0; I < ngl; I++) {
    if (K.glink[I] != L) continue;
    const int ib = K.gptr[I], nI = K.gptr[I + 1] - ib;
    if (nI < 2) continue;
    const double gI = gam[I], bI = bet[I];
    __syncthreads();                       // (q of the previous partner group is done with c0 as scratch)
    // r_b = u_I' S[I, b] for the columns b of J (scratch: c0[0 .. nJ))
    for (int b = tid; b < nJ; b += 256) {
      const double* __restrict__ p = S + ib + (int64_t)nS * (jb + b);
      double r = 0.0;
      for (int i = 0; i < nI; i++) r += u[ib + i] * p[i];
      c0[b] = r;
    }
    __syncthreads();
    const double t = tI[I];
    for (int e = tid; e < (nI - 1) * (nJ - 1); e += 256) {
      const int a = 1 + e % (nI - 1), b = 1 + e / (nI - 1);
      const double sab = S[(ib + a) + (int64_t)nS * (jb + b)];
      const double val = gI * gJ * u[ib + a] * t * u[jb + b] + gI * bJ * u[ib + a] * c0[b] + bI * gJ * q[ib + a] * u[jb + b] + bI * bJ * sab;
      rec[boff + (K.goff[I] + a - 1) + (int64_t)blen * (K.goff[J] + b - 1)] = val;
    }
  }
}
bool sblock_kept_fits(int32_t nS, int32_t ngl) {
  static const bool two_pass = std::getenv("HYMLS_MI_SBLOCK_TWO_PASS") != nullptr;   // (tests: force the fallback on small problems)
  return !two_pass && (size_t)(3 * (int64_t)nS + 4 * (int64_t)ngl) * sizeof(double) + 1024 <= LDS_LIMIT_BYTES;
}
void sblock_kept(const KeptD& K, const double* tv, const double* sblock, double* out, int64_t out_stride, int32_t nbc) {
  if (K.nS <= 0 || K.ngl <= 0 || nbc <= 0) return;
  const size_t shm = (size_t)(3 * K.nS + 4 * K.ngl) * sizeof(double);
  if (shm > 64 * 1024) HIP_CHECK(hipFuncSetAttribute((const void*)k_sblock_kept, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
  for (int s0 = 0; s0 < nbc; s0 += 65535) {
    const int ns = std::min(65535, nbc - s0);
    hipLaunchKernelGGL(k_sblock_kept, dim3(K.ngl, ns), dim3(256), shm, g_stream, K, tv + (int64_t)s0 * K.nS,
                       sblock + (int64_t)s0 * K.nS * K.nS, out + (int64_t)s0 * out_stride, out_stride);
    launch_check();
  }
}

// in-place Gauss-Jordan inversion with partial pivoting, one workgroup per block.
// `all` != nullptr: blocks of any order from a descriptor table (one launch for a whole level), else nblk blocks of
// order nb0.  INLDS: blocks of order <= lds_nb are copied into LDS (odd leading dimension) and inverted there, larger
// ones are left to the !INLDS launch over the same table (which in turn skips the small ones when lds_nb > 0).
// Three barriers per pivot: search (wave shuffles, then one LDS stage) | row interchange + scaling of the pivot row,
// with the multiplier column copied to LDS | elimination, a wave per column, lanes down the rows.
constexpr int GJ_LDS_NB = 88;   // 63.7 KiB of dynamic LDS at most
template <int BS, bool INLDS>
__global__ void __launch_bounds__(BS) k_dense_invert(int32_t nb0, double* __restrict__ blocks, const BlkD* __restrict__ all, int32_t* flag,
                                                      int32_t lds_nb) {
  extern __shared__ double dyn[];   // colk[nb] | piv[nb] (ints) | INLDS: the block
  __shared__ double red_v[BS / 64];
  __shared__ int red_i[BS / 64];
  const int nb = all ? all[blockIdx.x].nb : nb0;
  if (INLDS ? nb > lds_nb : nb <= lds_nb) return;
  double* G = all ? const_cast<double*>(all[blockIdx.x].binv) : blocks + (int64_t)blockIdx.x * nb * nb;
  double* colk = dyn;
  int* piv = (int*)(dyn + nb);
  const int ld = INLDS ? (nb | 1) : nb;
  double* A = INLDS ? dyn + nb + (nb + 1) / 2 : G;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nt = blockDim.x, nw = nt >> 6;
  if (INLDS) {
    for (int t = tid; t < nb * nb; t += nt) A[(t % nb) + ld * (t / nb)] = G[t];
    __syncthreads();
  }
  for (int k = 0; k < nb; k++) {
    // pivot search in column k, rows k..nb-1: largest modulus, the first of equals
    double best = -1.0; int bi = k;
    for (int i = k + tid; i < nb; i += nt) { const double a = fabs(A[i + (int64_t)ld * k]); if (a > best) { best = a; bi = i; } }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const double ov = __shfl_xor(best, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    if (lane == 0) { red_v[wave] = best; red_i[wave] = bi; }
    __syncthreads();
    double pv = red_v[0]; int p = red_i[0];
    for (int q = 1; q < nw; q++) {
      const double ov = red_v[q]; const int oi = red_i[q];
      if (ov > pv || (ov == pv && oi < p)) { pv = ov; p = oi; }
    }
    if (!(pv > 0.0) || !isfinite(pv)) { if (tid == 0) atomicExch(flag, 1); return; }   // (the same for every thread)
    const double ip = 1.0 / A[p + (int64_t)ld * k];
    // multiplier column (after the interchange, pivot row excluded); column k itself is rewritten in the elimination
    for (int i = tid; i < nb; i += nt) colk[i] = i == k ? 0.0 : A[(i == p ? k : i) + (int64_t)ld * k];
    if (tid == 0) piv[k] = p;
    // rows k and p change places, the new row k is scaled by 1 / pivot
    for (int j = tid; j < nb; j += nt)
      if (j != k) {
        const double akj = A[k + (int64_t)ld * j], apj = A[p + (int64_t)ld * j];
        A[k + (int64_t)ld * j] = apj * ip;
        if (p != k) A[p + (int64_t)ld * j] = akj;
      }
    __syncthreads();
    for (int j = wave; j < nb; j += nw) {
      double* cj = A + (int64_t)ld * j;
      if (j == k) {
        for (int i = lane; i < nb; i += 64) cj[i] = i == k ? ip : -colk[i] * ip;
      } else {
        const double r = cj[k];
        for (int i = lane; i < nb; i += 64) cj[i] -= colk[i] * r;     // (colk[k] = 0: the pivot row stays)
      }
    }
    __syncthreads();
  }
  for (int k = nb - 1; k >= 0; k--) {
    const int p = piv[k];
    if (p != k)
      for (int i = tid; i < nb; i += nt) { const double t = A[i + (int64_t)ld * k]; A[i + (int64_t)ld * k] = A[i + (int64_t)ld * p]; A[i + (int64_t)ld * p] = t; }
    __syncthreads();
  }
  if (INLDS)
    for (int t = tid; t < nb * nb; t += nt) G[t] = A[(t % nb) + ld * (t / nb)];
}
static size_t gj_dyn_bytes(int nb, bool inlds) {
  return ((size_t)nb + (nb + 1) / 2 + (inlds ? (size_t)(nb | 1) * nb : 0)) * sizeof(double);
}
// ---- blocked Gauss-Jordan inversion of the large separator blocks (orders of several hundred on the coarser levels) ----
// The scalar kernel above streams a whole block through one workgroup once per pivot: nb^3 x 16 B of traffic, HBM bound
// for the whole chip.  Here GJB = 32 pivots are taken at a time:
//   k_gj_panel       partial pivoting LU of the panel (rows >= kb of columns kb .. kb+31), one row per thread IN REGISTERS,
//                    pivot row broadcast through LDS; leaves the pivot list, Inv = (A_KK)^{-1} of the permuted pivot rows,
//                    W = the permuted panel with its pivot rows zeroed, and zeroes the panel columns of the block;
//   k_gj_swap_scale  one thread per column: the panel's row interchanges, then the pivot rows A_K <- R = Inv A_K
//                    (for the panel columns themselves: A_KK <- Inv);
//   gemm_f64         A <- A - W R on the matrix cores (every row except the pivot rows, all columns: the panel columns
//                    become -A_OK Inv, as the in-place Gauss-Jordan step does one column at a time);
//   k_gj_unpermute   the column interchanges in reverse order at the end.
// Same pivot candidates as the scalar kernel (rows not yet used as pivots), chosen panel by panel.
constexpr int GJB = 32;
constexpr int GJ_MAX = 1024;    // one row per thread
// elimination step T of the panel LU (compile-time T: the row stays in registers), then step T + 1
template <int T>
__device__ __forceinline__ void gj_panel_steps(double (&a)[GJB], int bw, int kb, int i, bool active, int lane, int wave, int nw, int tid,
                                               double* prow, double* krow, double* red_v, int* red_i, int* s_perm, int32_t* piv, int32_t* flag) {
  if constexpr (T < GJB) {
    constexpr int t = T;
    if (t < bw) {
      const int kr = kb + t;
      double v = (active && i >= kr) ? fabs(a[t]) : -1.0;
      if (v != v) v = INFINITY;   // a NaN must not hide behind the comparisons
      int idx = i;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const double ov = __shfl_xor(v, o, 64);
        const int oi = __shfl_xor(idx, o, 64);
        if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
      }
      if (lane == 0) { red_v[wave] = v; red_i[wave] = idx; }
      __syncthreads();
      double pv = red_v[0];
      int p = red_i[0];
      for (int q = 1; q < nw; q++) {
        const double ov = red_v[q];
        const int oi = red_i[q];
        if (ov > pv || (ov == pv && oi < p)) { pv = ov; p = oi; }
      }
      if (tid == 0 && (!(pv > 0.0) || !isfinite(pv))) atomicExch(flag, 1);   // (the caller discards the blocks then)
      if (i == p) {
#pragma unroll
        for (int c = 0; c < GJB; c++) prow[c] = a[c];
      }
      if (i == kr) {
#pragma unroll
        for (int c = 0; c < GJB; c++) krow[c] = a[c];
      }
      if (tid == 0) { piv[kr] = p; const int tmp = s_perm[kr]; s_perm[kr] = s_perm[p]; s_perm[p] = tmp; }
      __syncthreads();
      if (p != kr) {
        if (i == p) {
#pragma unroll
          for (int c = 0; c < GJB; c++) a[c] = krow[c];
        } else if (i == kr) {
#pragma unroll
          for (int c = 0; c < GJB; c++) a[c] = prow[c];
        }
      }
      if (active && i > kr) {
        const double l = a[t] / prow[t];
        a[t] = l;
#pragma unroll
        for (int c = t + 1; c < GJB; c++) a[c] -= l * prow[c];
      }
    }
    gj_panel_steps<T + 1>(a, bw, kb, i, active, lane, wave, nw, tid, prow, krow, red_v, red_i, s_perm, piv, flag);
  }
}
__global__ void __launch_bounds__(1024) k_gj_panel(int nb, int kb, double* __restrict__ blocks, double* __restrict__ W0,
                                                   double* __restrict__ Inv0, int32_t* __restrict__ piv0, int32_t* flag) {
  __shared__ double prow[GJB], krow[GJB], red_v[16], LU[GJB * GJB], X[GJB * GJB];
  __shared__ int red_i[16], s_perm[GJ_MAX];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6, i = tid;
  double* A = blocks + (int64_t)blockIdx.x * nb * nb;
  double* W = W0 + (int64_t)blockIdx.x * nb * GJB;
  double* Inv = Inv0 + (int64_t)blockIdx.x * GJB * GJB;
  int32_t* piv = piv0 + (int64_t)blockIdx.x * nb;
  const int bw = min(GJB, nb - kb);
  const bool have = i < nb, active = have && i >= kb;
  double a[GJB];
#pragma unroll
  for (int c = 0; c < GJB; c++) a[c] = (have && c < bw) ? A[i + (int64_t)nb * (kb + c)] : 0.0;
  if (have) s_perm[i] = i;
  for (int t = tid; t < GJB * GJB; t += blockDim.x) LU[t] = 0.0;
  __syncthreads();
  gj_panel_steps<0>(a, bw, kb, i, active, lane, wave, nw, tid, prow, krow, red_v, red_i, s_perm, piv, flag);
  // pivot rows: L11 \ U11 of the permuted panel -> Inv = U11^{-1} L11^{-1}, one column per thread
  if (active && i < kb + bw) {
#pragma unroll
    for (int c = 0; c < GJB; c++) if (c < bw) LU[(i - kb) + GJB * c] = a[c];
  }
  __syncthreads();
  if (tid < GJB) {
    const int j = tid;
    if (j < bw) {
      for (int r = 0; r < bw; r++) {
        double sum = r == j ? 1.0 : 0.0;
        for (int q = 0; q < r; q++) sum -= LU[r + GJB * q] * X[q * GJB + j];
        X[r * GJB + j] = sum;
      }
      for (int r = bw - 1; r >= 0; r--) {
        double sum = X[r * GJB + j];
        for (int q = r + 1; q < bw; q++) sum -= LU[r + GJB * q] * X[q * GJB + j];
        X[r * GJB + j] = sum / LU[r + GJB * r];
      }
    }
    for (int r = 0; r < GJB; r++) Inv[r + GJB * j] = (r < bw && j < bw) ? X[r * GJB + j] : 0.0;
  }
  // W: the panel in its permuted row order (the block itself still holds the original panel), pivot rows zero
  if (have) {
    const int src = s_perm[i];
#pragma unroll
    for (int c = 0; c < GJB; c++) a[c] = c < bw ? A[src + (int64_t)nb * (kb + c)] : 0.0;
  }
  __syncthreads();
  if (have) {
    const bool pivot_row = i >= kb && i < kb + bw;
#pragma unroll
    for (int c = 0; c < GJB; c++)
      if (c < bw) { W[i + (int64_t)nb * c] = pivot_row ? 0.0 : a[c]; A[i + (int64_t)nb * (kb + c)] = 0.0; }
  }
}
// the panel step for orders above GJ_MAX (up to GJ_BIG_MAX): the same result as k_gj_panel, with the panel LU done in a
// global scratch copy (Lw = this block's part of R, free until k_gj_swap_scale writes it) instead of registers: 32 pivots x
// three barriers, rows spread over the 1024 threads, the pivot row staged in LDS
constexpr int GJ_BIG_MAX = 4096;
__global__ void __launch_bounds__(1024) k_gj_panel_big(int nb, int kb, double* __restrict__ blocks, double* __restrict__ W0,
                                                       double* __restrict__ Lw0, double* __restrict__ Inv0, int32_t* __restrict__ piv0, int32_t* flag) {
  __shared__ double prow[GJB], red_v[16], LU[GJB * GJB], X[GJB * GJB];
  __shared__ int red_i[16], s_perm[GJ_BIG_MAX];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nt = blockDim.x, nw = nt >> 6;
  double* A = blocks + (int64_t)blockIdx.x * nb * nb;
  double* W = W0 + (int64_t)blockIdx.x * nb * GJB;
  double* Lw = Lw0 + (int64_t)blockIdx.x * nb * GJB;      // panel copy, column-major with leading dimension nb
  double* Inv = Inv0 + (int64_t)blockIdx.x * GJB * GJB;
  int32_t* piv = piv0 + (int64_t)blockIdx.x * nb;
  const int bw = min(GJB, nb - kb);
  for (int c = 0; c < bw; c++)
    for (int i = tid; i < nb; i += nt) Lw[i + (int64_t)nb * c] = A[i + (int64_t)nb * (kb + c)];
  for (int i = tid; i < nb; i += nt) s_perm[i] = i;
  for (int t = tid; t < GJB * GJB; t += nt) LU[t] = 0.0;
  __syncthreads();
  for (int t = 0; t < bw; t++) {
    const int kr = kb + t;
    double* col = Lw + (int64_t)nb * t;
    double v = -1.0; int idx = kr;
    for (int i = kr + tid; i < nb; i += nt) { double a = fabs(col[i]); if (a != a) a = INFINITY; if (a > v) { v = a; idx = i; } }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const double ov = __shfl_xor(v, o, 64);
      const int oi = __shfl_xor(idx, o, 64);
      if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
    }
    if (lane == 0) { red_v[wave] = v; red_i[wave] = idx; }
    __syncthreads();
    double pv = red_v[0]; int p = red_i[0];
    for (int q = 1; q < nw; q++) {
      const double ov = red_v[q]; const int oi = red_i[q];
      if (ov > pv || (ov == pv && oi < p)) { pv = ov; p = oi; }
    }
    if (tid == 0 && (!(pv > 0.0) || !isfinite(pv))) atomicExch(flag, 1);
    // rows kr and p of the panel copy change places; the new pivot row goes to LDS
    if (tid < bw) {
      const double a = Lw[kr + (int64_t)nb * tid], b = Lw[p + (int64_t)nb * tid];
      Lw[kr + (int64_t)nb * tid] = b; Lw[p + (int64_t)nb * tid] = a;
      prow[tid] = b;
    }
    if (tid == 0) { piv[kr] = p; const int tmp = s_perm[kr]; s_perm[kr] = s_perm[p]; s_perm[p] = tmp; }
    __syncthreads();
    const double ip = 1.0 / prow[t];
    for (int i = kr + 1 + tid; i < nb; i += nt) {
      const double l = col[i] * ip;
      col[i] = l;
      for (int c = t + 1; c < bw; c++) Lw[i + (int64_t)nb * c] -= l * prow[c];
    }
    __syncthreads();
  }
  for (int q = tid; q < bw * bw; q += nt) { const int r = q % bw, c = q / bw; LU[r + GJB * c] = Lw[(kb + r) + (int64_t)nb * c]; }
  __syncthreads();
  if (tid < GJB) {
    const int j = tid;
    if (j < bw) {
      for (int r = 0; r < bw; r++) {
        double sum = r == j ? 1.0 : 0.0;
        for (int q = 0; q < r; q++) sum -= LU[r + GJB * q] * X[q * GJB + j];
        X[r * GJB + j] = sum;
      }
      for (int r = bw - 1; r >= 0; r--) {
        double sum = X[r * GJB + j];
        for (int q = r + 1; q < bw; q++) sum -= LU[r + GJB * q] * X[q * GJB + j];
        X[r * GJB + j] = sum / LU[r + GJB * r];
      }
    }
    for (int r = 0; r < GJB; r++) Inv[r + GJB * j] = (r < bw && j < bw) ? X[r * GJB + j] : 0.0;
  }
  // W: the ORIGINAL panel (still in the block) in its permuted row order, pivot rows zero; then the panel columns are zeroed
  for (int c = 0; c < bw; c++)
    for (int i = tid; i < nb; i += nt) W[i + (int64_t)nb * c] = (i >= kb && i < kb + bw) ? 0.0 : A[s_perm[i] + (int64_t)nb * (kb + c)];
  __syncthreads();
  for (int c = 0; c < bw; c++)
    for (int i = tid; i < nb; i += nt) A[i + (int64_t)nb * (kb + c)] = 0.0;
}
__global__ void __launch_bounds__(256) k_gj_swap_scale(int nb, int kb, double* __restrict__ blocks, const double* __restrict__ Inv0,
                                                       const int32_t* __restrict__ piv0, double* __restrict__ R0) {
  __shared__ double sInv[GJB * GJB];
  __shared__ int sp[GJB];
  const int tid = threadIdx.x, j = blockIdx.x * 256 + tid, bw = min(GJB, nb - kb);
  double* A = blocks + (int64_t)blockIdx.y * nb * nb;
  const double* Inv = Inv0 + (int64_t)blockIdx.y * GJB * GJB;
  double* R = R0 + (int64_t)blockIdx.y * GJB * nb;
  for (int t = tid; t < GJB * GJB; t += 256) sInv[t] = Inv[t];
  if (tid < GJB) sp[tid] = tid < bw ? piv0[(int64_t)blockIdx.y * nb + kb + tid] : 0;
  __syncthreads();
  if (j >= nb) return;
  double* col = A + (int64_t)nb * j;
  double x[GJB];
  if (j >= kb && j < kb + bw) {
    // a panel column: the identity in the pivot rows before the scaling, so R = Inv e_(j - kb)
#pragma unroll
    for (int r = 0; r < GJB; r++) x[r] = sInv[r + GJB * (j - kb)];
  } else {
    for (int t = 0; t < bw; t++) {
      const int p = sp[t];
      if (p != kb + t) { const double tmp = col[kb + t]; col[kb + t] = col[p]; col[p] = tmp; }
    }
    double y[GJB];
#pragma unroll
    for (int t = 0; t < GJB; t++) y[t] = t < bw ? col[kb + t] : 0.0;
#pragma unroll
    for (int r = 0; r < GJB; r++) {
      double s0 = 0.0, s1 = 0.0;
#pragma unroll
      for (int t = 0; t < GJB; t += 2) { s0 += sInv[r + GJB * t] * y[t]; s1 += sInv[r + GJB * (t + 1)] * y[t + 1]; }
      x[r] = s0 + s1;
    }
  }
#pragma unroll
  for (int r = 0; r < GJB; r++) {
    R[r + GJB * j] = x[r];
    if (r < bw) col[kb + r] = x[r];
  }
}
__global__ void __launch_bounds__(256) k_gj_unpermute(int nb, double* __restrict__ blocks, const int32_t* __restrict__ piv0) {
  double* A = blocks + (int64_t)blockIdx.x * nb * nb;
  const int32_t* piv = piv0 + (int64_t)blockIdx.x * nb;
  for (int k = nb - 1; k >= 0; k--) {
    const int p = piv[k];
    if (p != k)
      for (int i = threadIdx.x; i < nb; i += blockDim.x) { const double t = A[i + (int64_t)nb * k]; A[i + (int64_t)nb * k] = A[i + (int64_t)nb * p]; A[i + (int64_t)nb * p] = t; }
    __syncthreads();
  }
}
static int gj_blocked_min() {
  static const int v = std::getenv("HYMLS_MI_INVERT_BLOCKED_MIN") ? std::atoi(std::getenv("HYMLS_MI_INVERT_BLOCKED_MIN")) : 160;
  return v;
}
static void dense_invert_blocked(int32_t nb, int32_t nblk, double* blocks, int32_t* flag) {
  const size_t wbytes = (size_t)nblk * nb * GJB * sizeof(double), ibytes = (size_t)nblk * GJB * GJB * sizeof(double);
  const size_t pbytes = (((size_t)nblk * nb * sizeof(int32_t)) + 255) / 256 * 256;
  char* ws = (char*)shared_scratch(2 * wbytes + ibytes + pbytes);
  double* W = (double*)ws;
  double* R = (double*)(ws + wbytes);
  double* Inv = (double*)(ws + 2 * wbytes);
  int32_t* piv = (int32_t*)(ws + 2 * wbytes + ibytes);
  const int threads = std::min(1024, (nb + 63) / 64 * 64);
  for (int kb = 0; kb < nb; kb += GJB) {
    const int bw = std::min(GJB, nb - kb);
    if (nb <= GJ_MAX) hipLaunchKernelGGL(k_gj_panel, dim3(nblk), dim3(threads), 0, g_stream, nb, kb, blocks, W, Inv, piv, flag);
    else hipLaunchKernelGGL(k_gj_panel_big, dim3(nblk), dim3(1024), 0, g_stream, nb, kb, blocks, W, R, Inv, piv, flag);
    hipLaunchKernelGGL(k_gj_swap_scale, dim3((nb + 255) / 256, nblk), dim3(256), 0, g_stream, nb, kb, blocks, Inv, piv, R);
    launch_check();
    gemm_f64<1, 0, 0>(blocks, nb, (int64_t)nb * nb, W, nb, (int64_t)nb * GJB, R, GJB, (int64_t)GJB * nb, nb, nb, bw, nblk);
  }
  hipLaunchKernelGGL(k_gj_unpermute, dim3(nblk), dim3(256), 0, g_stream, nb, blocks, piv);
  launch_check();
}
bool dense_invert_blocked_order(int32_t nb) { return nb >= gj_blocked_min() && nb <= GJ_BIG_MAX; }
void dense_invert(int32_t nb, int32_t nblk, double* blocks, int32_t* flag) {
  if (nb <= 0 || nblk <= 0) return;
  if (dense_invert_blocked_order(nb)) {
    // (chunks: the panel workspace W | R of a call stays below 1 GiB, the grid below 65536 blocks)
    const int32_t per = (int32_t)std::max<int64_t>(1, std::min<int64_t>(65535, ((int64_t)1 << 30) / ((int64_t)2 * nb * GJB * sizeof(double))));
    for (int32_t b0 = 0; b0 < nblk; b0 += per) dense_invert_blocked(nb, std::min(per, nblk - b0), blocks + (int64_t)b0 * nb * nb, flag);
    return;
  }
  if (nb <= GJ_LDS_NB)
    hipLaunchKernelGGL((k_dense_invert<256, true>), dim3(nblk), dim3(nb <= 16 ? 64 : 256), gj_dyn_bytes(nb, true), g_stream, nb, blocks, (const BlkD*)nullptr, flag, GJ_LDS_NB);
  else if (nb > 256)
    hipLaunchKernelGGL((k_dense_invert<1024, false>), dim3(nblk), dim3(1024), gj_dyn_bytes(nb, false), g_stream, nb, blocks, (const BlkD*)nullptr, flag, 0);
  else
    hipLaunchKernelGGL((k_dense_invert<256, false>), dim3(nblk), dim3(256), gj_dyn_bytes(nb, false), g_stream, nb, blocks, (const BlkD*)nullptr, flag, 0);
  launch_check();
}
void dense_invert_all(int32_t nblk, const BlkD* blocks, int32_t max_nb, int32_t* flag) {
  if (nblk <= 0) return;
  // two launches over the same table: orders up to GJ_LDS_NB inside LDS, the others in global memory (1024 threads where
  // blocks are large: the rank-1 updates take the time)
  hipLaunchKernelGGL((k_dense_invert<256, true>), dim3(nblk), dim3(256), gj_dyn_bytes(std::min(max_nb, GJ_LDS_NB), true), g_stream, 0, (double*)nullptr,
                     blocks, flag, GJ_LDS_NB);
  if (max_nb > 256)
    hipLaunchKernelGGL((k_dense_invert<1024, false>), dim3(nblk), dim3(1024), gj_dyn_bytes(max_nb, false), g_stream, 0, (double*)nullptr, blocks, flag, GJ_LDS_NB);
  else if (max_nb > GJ_LDS_NB)
    hipLaunchKernelGGL((k_dense_invert<256, false>), dim3(nblk), dim3(256), gj_dyn_bytes(max_nb, false), g_stream, 0, (double*)nullptr, blocks, flag, GJ_LDS_NB);
  launch_check();
}

__global__ void __launch_bounds__(256) k_blocks_apply(int32_t nb, const double* __restrict__ binv, const int32_t* __restrict__ ids,
                                                       const double* __restrict__ x, double* __restrict__ y) {
  extern __shared__ double xs[];
  const double* M = binv + (int64_t)blockIdx.x * nb * nb;
  const int32_t* id = ids + (int64_t)blockIdx.x * nb;
  for (int j = threadIdx.x; j < nb; j += blockDim.x) xs[j] = x[id[j]];
  __syncthreads();
  for (int i = threadIdx.x; i < nb; i += blockDim.x) {
    double s0 = 0.0, s1 = 0.0;
    int j = 0;
    for (; j + 1 < nb; j += 2) { s0 += M[i + (int64_t)nb * j] * xs[j]; s1 += M[i + (int64_t)nb * (j + 1)] * xs[j + 1]; }
    if (j < nb) s0 += M[i + (int64_t)nb * j] * xs[j];
    y[id[i]] = s0 + s1;
  }
}
// one wave per block, lane = row, eight columns in flight
__global__ void __launch_bounds__(64) k_blocks_apply_all(const BlkD* __restrict__ blocks, const double* __restrict__ x, double* __restrict__ y) {
  extern __shared__ double xs[];
  const BlkD D = blocks[blockIdx.x];
  const int nb = D.nb;
  const gptr<int32_t> ids = as_global(D.ids);          // (pointers out of the descriptor: global memory, see as_global)
  for (int j = threadIdx.x; j < nb; j += 64) xs[j] = x[ids[j]];
  __syncthreads();
  const int i0 = D.r0 < 0 ? 0 : D.r0, i1 = D.r0 < 0 ? nb : min(nb, D.r0 + 64);
  for (int i = i0 + threadIdx.x; i < i1; i += 64) {
    const gptr<double> M = as_global(D.binv) + i;
    double a[8];
#pragma unroll
    for (int u = 0; u < 8; u++) a[u] = 0.0;
    int j = 0;
    for (; j + 7 < nb; j += 8) {
      double l[8];
#pragma unroll
      for (int u = 0; u < 8; u++) l[u] = M[(int64_t)nb * (j + u)];
#pragma unroll
      for (int u = 0; u < 8; u++) a[u] += l[u] * xs[j + u];
    }
    for (; j < nb; j++) a[0] += M[(int64_t)nb * j] * xs[j];
    y[ids[i]] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
  }
}
void blocks_apply_all(int32_t nblk, const BlkD* blocks, int32_t max_nb, const double* x, double* y) {
  if (nblk <= 0) return;
  hipLaunchKernelGGL(k_blocks_apply_all, dim3(nblk), dim3(64), (size_t)max_nb * sizeof(double), g_stream, blocks, x, y);
  launch_check();
}

void blocks_apply(int32_t nb, int32_t nblk, const double* binv, const int32_t* ids, const double* x, double* y) {
  if (nb <= 0 || nblk <= 0) return;
  const int bs = nb <= 64 ? 64 : (nb <= 128 ? 128 : 256);
  hipLaunchKernelGGL(k_blocks_apply, dim3(nblk), dim3(bs), (size_t)nb * sizeof(double), g_stream, nb, binv, ids, x, y);
  launch_check();
}

template <int NV>
__global__ void __launch_bounds__(64) k_blocks_apply_all_mv(const BlkD* __restrict__ blocks, const double* __restrict__ x, int64_t ldx,
                                                             double* __restrict__ y, int64_t ldy, int32_t XS) {
  extern __shared__ double xs[];   // [NV][XS]
  const BlkD D = blocks[blockIdx.x];
  const int nb = D.nb;
  const gptr<int32_t> ids = as_global(D.ids);
  for (int j = threadIdx.x; j < nb; j += 64) {
    const int id = ids[j];
#pragma unroll
    for (int v = 0; v < NV; v++) xs[v * XS + j] = x[v * ldx + id];
  }
  __syncthreads();
  const int i0 = D.r0 < 0 ? 0 : D.r0, i1 = D.r0 < 0 ? nb : min(nb, D.r0 + 64);
  for (int i = i0 + threadIdx.x; i < i1; i += 64) {
    const gptr<double> M = as_global(D.binv) + i;
    double a[8][NV];
#pragma unroll
    for (int u = 0; u < 8; u++)
#pragma unroll
      for (int v = 0; v < NV; v++) a[u][v] = 0.0;
    int j = 0;
    for (; j + 7 < nb; j += 8) {
      double l[8];
#pragma unroll
      for (int u = 0; u < 8; u++) l[u] = M[(int64_t)nb * (j + u)];
#pragma unroll
      for (int u = 0; u < 8; u++)
#pragma unroll
        for (int v = 0; v < NV; v++) a[u][v] += l[u] * xs[v * XS + j + u];
    }
    for (; j < nb; j++) {
      const double l = M[(int64_t)nb * j];
#pragma unroll
      for (int v = 0; v < NV; v++) a[0][v] += l * xs[v * XS + j];
    }
    const int id = ids[i];
#pragma unroll
    for (int v = 0; v < NV; v++) y[v * ldy + id] = ((a[0][v] + a[1][v]) + (a[2][v] + a[3][v])) + ((a[4][v] + a[5][v]) + (a[6][v] + a[7][v]));
  }
}
void blocks_apply_all_mv(int32_t nblk, const BlkD* blocks, int32_t max_nb, const double* x, int64_t ldx, double* y, int64_t ldy, int nv) {
  if (nblk <= 0 || nv <= 0) return;
  int v = 0;
  while (v < nv) {
    int g = nv - v >= 4 ? 4 : (nv - v >= 2 ? 2 : 1);
    while (g > 1 && ((size_t)max_nb * g * sizeof(double) > 64 * 1024 || g > mv_group_cap("BLK"))) g >>= 1;
    const double* xv = x + (int64_t)v * ldx;
    double* yv = y + (int64_t)v * ldy;
    const size_t shm = (size_t)max_nb * g * sizeof(double);
    if (g == 4) { hipLaunchKernelGGL(k_blocks_apply_all_mv<4>, dim3(nblk), dim3(64), shm, g_stream, blocks, xv, ldx, yv, ldy, max_nb); launch_check(); }
    else if (g == 2) { hipLaunchKernelGGL(k_blocks_apply_all_mv<2>, dim3(nblk), dim3(64), shm, g_stream, blocks, xv, ldx, yv, ldy, max_nb); launch_check(); }
    else blocks_apply_all(nblk, blocks, max_nb, xv, yv);
    v += g;
  }
}

}  // namespace dev
}  // namespace hymls
