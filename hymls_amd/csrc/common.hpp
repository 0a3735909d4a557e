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

// host CSR matrix with an explicit gid per local row/col (square, same map)
struct Csr {
  int32_t n = 0;
  ivec rowptr, col;  // local column indices
  dvec val;
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

}  // namespace hymls
