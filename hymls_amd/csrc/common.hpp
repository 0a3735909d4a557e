// common.hpp -- shared host-side types of the MI355X HYMLS hot path.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <vector>
#include <algorithm>
#include <numeric>
#include <cmath>
#include <thread>
#include <memory>
#include <cstring>
#include <atomic>
#include <mutex>
#include <exception>

namespace hymls {

// mirrors HYMLS_SMALL_ENTRY (reference src/HYMLS_Macros.hpp:26-30)
constexpr double SMALL_ENTRY = 1e-14;

struct Error : std::runtime_error {
  int code;
  Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

#define HYMLS_CHECK(cond, code, msg)                                              \
  do {                                                                            \
    if (!(cond)) throw ::hymls::Error((code), std::string(msg) + " [" + __FILE__ + \
                                               ":" + std::to_string(__LINE__) + "]"); \
  } while (0)

using ivec = std::vector<int32_t>;
using dvec = std::vector<double>;
// vectors of plain numbers whose resize() leaves the new entries uninitialised: for the gigabyte-sized tables of the setup
// that are filled in parallel right after (a value-initialising resize is a serial memset: 10 GB of it at 256^3)
template <class T>
struct no_init_allocator : std::allocator<T> {
  template <class U> struct rebind { using other = no_init_allocator<U>; };
  no_init_allocator() = default;
  template <class U> no_init_allocator(const no_init_allocator<U>&) {}
  template <class U> void construct(U* p) { ::new ((void*)p) U; }
  template <class U, class A0, class... A> void construct(U* p, A0&& a0, A&&... a) { ::new ((void*)p) U(std::forward<A0>(a0), std::forward<A>(a)...); }
};
template <class T> using rawvec = std::vector<T, no_init_allocator<T>>;
using cvec = rawvec<int32_t>;   // column indices of a Csr
using vvec = rawvec<double>;    // values of a Csr

// host CSR matrix with an explicit gid per local row/col (square, same map)
struct Csr {
  int32_t n = 0;
  ivec rowptr;
  cvec col;  // local column indices (resize() does not initialise: filled by whoever sizes it)
  vvec val;
  int64_t nnz() const { return (int64_t)col.size(); }
};

enum VarType : int32_t { VT_LAPLACE = 0, VT_U = 1, VT_V = 2, VT_W = 3, VT_P = 4, VT_INTERIOR = 5 };

// the reference's "Problem"/"Preconditioner" keys (BasePartitioner.cpp:31-252)
struct Params {
  int nx = -1, ny = -1, nz = -1, dim = 3, dof = 1;
  int sx = 4, sy = -1, sz = -1, cx = -1, cy = -1, cz = -1;
  int rx = -1, ry = -1, rz = -1;   // "Retain Nodes" of THIS level (see set_retain)
  int retain = -1, retain_xyz[3] = {-1, -1, -1}, retain_at_level[8] = {-1, -1, -1, -1, -1, -1, -1, -1};
  int retain_at_level_xyz[8][3] = {{-1, -1, -1}, {-1, -1, -1}, {-1, -1, -1}, {-1, -1, -1}, {-1, -1, -1}, {-1, -1, -1}, {-1, -1, -1}, {-1, -1, -1}};
  bool perio[3] = {false, false, false};   // "x-periodic", "y-periodic", "z-periodic"
  int level = 0;                   // partitioner level (0 = finest)
  int levels = 1;
  int partitioner = 0;  // 0 Cartesian, 1 Skew Cartesian
  int retain_pressures = 1;
  bool link_velocities = true, link_retained = true;
  std::vector<int32_t> vtype;    // per dof
  std::vector<int32_t> fix_gid;  // "Fix GID n"
  // rx, ry, rz of this level: "Retain Nodes at Level <level> (x|y|z)" wins, then "Retain Nodes (x|y|z)", then
  // "Retain Nodes at Level <level>", then "Retain Nodes" (reference src/HYMLS_BasePartitioner.cpp:108-137)
  void set_retain() {
    const int at = level < 8 ? retain_at_level[level] : -1;
    int* r[3] = {&rx, &ry, &rz};
    for (int d = 0; d < 3; d++) {
      const int atd = level < 8 ? retain_at_level_xyz[level][d] : -1;
      *r[d] = atd != -1 ? atd : (retain_xyz[d] != -1 ? retain_xyz[d] : (at != -1 ? at : retain));
    }
  }
  Params next_level() const {  // SetNextLevelParameters (BasePartitioner.cpp:321-346)
    Params q = *this;
    q.sx = sx * cx; q.sy = sy * cy; q.sz = sz * cz;
    q.level = level + 1;
    q.set_retain();
    return q;
  }
};

// static chunks over [0, n) on up to 16 host threads (setup-time integer work only)
template <class Fn>
void parallel_for(int64_t n, Fn fn, int64_t grain = 256) {
  // host threads of the setup: 16 by default (the share of a GPU on an 8-GPU node), HYMLS_MI_HOST_THREADS overrides
  static const int64_t cap = std::getenv("HYMLS_MI_HOST_THREADS") ? std::max(1, std::atoi(std::getenv("HYMLS_MI_HOST_THREADS"))) : 16;
  const int nt = (int)std::max<int64_t>(1, std::min<int64_t>({(int64_t)std::thread::hardware_concurrency(), cap, n / grain}));
  if (nt <= 1) { for (int64_t i = 0; i < n; i++) fn(i); return; }
  std::vector<std::thread> th;
  std::exception_ptr err = nullptr;
  std::mutex mu;
  for (int t = 0; t < nt; t++)
    th.emplace_back([&, t] {
      try { for (int64_t i = n * t / nt; i < n * (t + 1) / nt; i++) fn(i); }
      catch (...) { std::lock_guard<std::mutex> lk(mu); err = std::current_exception(); }
    });
  for (auto& x : th) x.join();
  if (err) std::rethrow_exception(err);
}

// memcpy / memcmp of gigabytes on the setup threads (the matrix of a 256^3 run is 12 GB: a serial copy takes two seconds)
inline void parallel_memcpy(void* dst, const void* src, size_t bytes) {
  constexpr size_t CH = (size_t)8 << 20;
  if (bytes < 4 * CH) { if (bytes) std::memcpy(dst, src, bytes); return; }
  parallel_for((int64_t)((bytes + CH - 1) / CH), [&](int64_t q) {
    const size_t o = (size_t)q * CH;
    std::memcpy((char*)dst + o, (const char*)src + o, std::min(CH, bytes - o));
  }, 1);
}
inline bool parallel_equal(const void* a, const void* b, size_t bytes) {
  constexpr size_t CH = (size_t)8 << 20;
  if (bytes < 4 * CH) return bytes == 0 || std::memcmp(a, b, bytes) == 0;
  std::atomic<int> diff{0};
  parallel_for((int64_t)((bytes + CH - 1) / CH), [&](int64_t q) {
    const size_t o = (size_t)q * CH;
    if (!diff && std::memcmp((const char*)a + o, (const char*)b + o, std::min(CH, bytes - o)) != 0) diff = 1;
  }, 1);
  return diff == 0;
}
// vector <- array, copied on the setup threads (resize() of a rawvec does not touch the memory first)
template <class V, class T>
void parallel_assign(V& v, const T* src, size_t n) {
  v.resize(n);
  parallel_memcpy(v.data(), src, n * sizeof(T));
}
// in-place inclusive prefix sum of a[0 .. n) on the setup threads (chunk sums, their serial scan, offsets added in parallel)
template <class T>
void parallel_inclusive_scan(T* a, int64_t n) {
  constexpr int64_t CH = 1 << 20;
  if (n < 4 * CH) { for (int64_t i = 1; i < n; i++) a[i] += a[i - 1]; return; }
  const int64_t nc = (n + CH - 1) / CH;
  std::vector<T> sum((size_t)nc);
  parallel_for(nc, [&](int64_t c) {
    const int64_t e = std::min(n, (c + 1) * CH);
    for (int64_t i = c * CH + 1; i < e; i++) a[i] += a[i - 1];
    sum[(size_t)c] = a[e - 1];
  }, 1);
  for (int64_t c = 1; c < nc; c++) sum[(size_t)c] += sum[(size_t)c - 1];
  parallel_for(nc - 1, [&](int64_t c) {
    const T off = sum[(size_t)c];
    const int64_t e = std::min(n, (c + 2) * CH);
    for (int64_t i = (c + 1) * CH; i < e; i++) a[i] += off;
  }, 1);
}
// stream compaction in index order: emit(i, position among the kept) for every i in [0, n) with keep(i); returns the count
template <class Keep, class Emit>
int64_t parallel_compact(int64_t n, Keep keep, Emit emit) {
  constexpr int64_t CH = 1 << 20;
  const int64_t nc = std::max<int64_t>(1, (n + CH - 1) / CH);
  std::vector<int64_t> cnt((size_t)nc + 1, 0);
  parallel_for(nc, [&](int64_t c) {
    int64_t k = 0;
    for (int64_t i = c * CH, e = std::min(n, (c + 1) * CH); i < e; i++) k += keep(i) ? 1 : 0;
    cnt[(size_t)c + 1] = k;
  }, 1);
  for (int64_t c = 0; c < nc; c++) cnt[(size_t)c + 1] += cnt[(size_t)c];
  parallel_for(nc, [&](int64_t c) {
    int64_t k = cnt[(size_t)c];
    for (int64_t i = c * CH, e = std::min(n, (c + 1) * CH); i < e; i++) if (keep(i)) emit(i, k++);
  }, 1);
  return cnt[(size_t)nc];
}

}  // namespace hymls
