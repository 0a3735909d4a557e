// precond.cpp -- see precond.hpp
#include "precond.hpp"
#include <cstring>
#include <chrono>
#include <thread>
#include <mutex>
#include <atomic>
#include <map>
#include <unordered_map>

namespace hymls {

namespace {

struct Hasher {
  uint64_t h = 1469598103934665603ULL;
  // 64 bits per step (the patterns of 70 k subdomains, ~100 kB each, are hashed at Initialize; equal hashes are confirmed by
  // a full comparison afterwards)
  void add(const void* p, size_t n) {
    const unsigned char* c = (const unsigned char*)p;
    size_t i = 0;
    for (; i + 8 <= n; i += 8) {
      uint64_t w;
      std::memcpy(&w, c + i, 8);
      h = (h ^ w) * 0x9E3779B97F4A7C15ULL;
      h ^= h >> 29;
    }
    for (; i < n; i++) { h ^= c[i]; h *= 1099511628211ULL; }
  }
  template <class T> void addv(const std::vector<T>& v) { size_t n = v.size(); add(&n, sizeof n); if (n) add(v.data(), n * sizeof(T)); }
};

inline void gid_coord(const Params& p, int32_t gid, int32_t* c) {
  const int var = gid % p.dof, cell = gid / p.dof;
  const int i = cell % p.nx, j = (cell / p.nx) % p.ny, k = cell / (p.nx * p.ny);
  const int32_t vt = p.vtype[var];
  c[0] = 2 * i + (vt == VT_U); c[1] = 2 * j + (vt == VT_V); c[2] = 2 * k + (vt == VT_W);
}

constexpr int LEAF_SIZE = 24;
constexpr int LEAF_SIZE_UPPER = 64;   // levels >= 1 (measured at 256^3: coarse phase 6.2 -> 5.7 ms; 48: 5.8, 96: 5.6; end of round 3, gpurun_out/r3at: 64 / 96 / 128 / 192 = 5.58 / 5.52 / 5.65 / 5.83 ms)
constexpr int MAX_WIDTH = 256;
constexpr int MAX_WIDTH_COARSE = 256;  // wider supernodes take the piece-wise big-front path
constexpr int LEAF_SIZE_COARSE = 512; // last-level solver (measured on configs[1], 216 k unknowns, gpurun_out/r3aj + r3ak: coarse phase 2.40 ms at 64,
                                      // 2.29 / 2.17 / 2.14 / 2.00 / 2.11 ms at 32 / 128 / 256 / 512 / 1024; the numeric Compute 0.98 -> 1.07 s)
// doubles of frontal scratch per factorisation pass: 8 GiB (HYMLS_MI_SCRATCH_GIB: other values).  Measured at 256^3
// (gpurun_out/r3ar): recompute 2.69 - 2.81 s at 8 GiB, 2.92 - 2.99 s at 4 GiB, 4.1 s at 2 GiB (smaller batches per launch);
// the first Compute, which allocates the arenas, 6.1 - 7.2 / 5.9 - 6.1 / 6.9 s.
static const int64_t SCRATCH_BUDGET = [] {
  const char* e = std::getenv("HYMLS_MI_SCRATCH_GIB");
  const double gib = e ? std::max(0.25, std::atof(e)) : 8.0;
  return (int64_t)(gib * (double)(1LL << 27));
}();

}  // namespace

// ------------------------------------------------------------------ DropByValue
Csr drop_by_value(const Csr& A, double tol, int kind) {
  Csr R;
  drop_by_value(A, tol, kind, R);
  return R;
}

// the same into an existing matrix: its arrays are reused when the sizes have not changed (a recompute with the same
// pattern: no allocation, no zero-filling of gigabytes)
void drop_by_value(const Csr& A, double tol, int kind, Csr& R) {
  const bool zero_diag = kind == 1, full_diag = kind == 2;
  R.n = A.n;
  R.rowptr.resize(A.n + 1);
  R.rowptr[0] = 0;
  // A diagonal entry at rounding level (<= tol x the largest entry of its row) counts as the structural zero it is: the
  // pressure diagonals of a reduced matrix cancel exactly on paper, and the ordering of the next level / the last-level
  // solver tells pressures from velocities by a zero diagonal (as the reference does, MatrixUtils.cpp:1344-1352; its
  // ComputeScaling uses the same 1e-14 relative threshold, SparseDirectSolver.cpp:632-664).  Summed on the GPU such a
  // diagonal came out as 1.05e-14 next to entries of 1e3 and passed the absolute test: Darcy3D with separator length 16
  // then paired velocities with the wrong nodes and the pivot-free factorisation grew by 1e13.
  dvec diag(A.n, 0.0);
  parallel_for(A.n, [&](int64_t i) {
    double d = 0.0, rmax = 0.0;
    for (int e = A.rowptr[i]; e < A.rowptr[i + 1]; e++) {
      rmax = std::max(rmax, std::abs(A.val[e]));
      if (A.col[e] == i) d = A.val[e];
    }
    diag[i] = std::abs(d) <= tol * rmax ? 0.0 : d;
  });
  // two passes over the rows (count, fill), rows in parallel
  auto row = [&](int64_t i, int32_t* col, double* val) {
    int32_t n = 0;
    auto put = [&](int32_t j, double v) { if (col) { col[n] = j; val[n] = v; } n++; };
    if (full_diag) put((int32_t)i, std::abs(diag[i]) > tol ? diag[i] : 0.0);
    for (int e = A.rowptr[i]; e < A.rowptr[i + 1]; e++) {
      const int j = A.col[e];
      const bool isd = j == i;
      if (isd && full_diag) continue;
      const double scal = isd ? 1.0 : std::max(std::abs(diag[i]), std::abs(diag[j]));
      const double v = isd ? diag[i] : A.val[e];
      if (std::abs(v) > scal * tol && std::abs(v) > tol) put(j, v);
      else if (isd && zero_diag) put(j, 0.0);
    }
    return n;
  };
  parallel_for(A.n, [&](int64_t i) { R.rowptr[i + 1] = row(i, nullptr, nullptr); });
  for (int i = 0; i < A.n; i++) R.rowptr[i + 1] += R.rowptr[i];
  if (R.col.size() != (size_t)R.rowptr[A.n]) { cvec().swap(R.col); R.col.resize((size_t)R.rowptr[A.n]); }
  if (R.val.size() != (size_t)R.rowptr[A.n]) { vvec().swap(R.val); R.val.resize((size_t)R.rowptr[A.n]); }
  parallel_for(A.n, [&](int64_t i) { row(i, R.col.data() + R.rowptr[i], R.val.data() + R.rowptr[i]); });
}

// ------------------------------------------------------------------ BatchedLU
BatchedLU::~BatchedLU() {
  for (void* p : owned) dev::free(p);
}

// members per factorisation pass and the doubles of frontal scratch / separator blocks / pivot-piece workspace a pass borrows
// from the stream's arena (plan-only: known right after the symbolic analysis, so that the arenas can be requested early)
void BatchedLU::plan_scratch(int64_t budget, bool with_sblock) {
  const int nb = (int)members.size();
  const int64_t per = plan.scratch_size + (with_sblock ? (int64_t)plan.nS * plan.nS : 0);
  chunk = (int32_t)std::max<int64_t>(1, std::min<int64_t>(nb, budget / std::max<int64_t>(per, 1)));
  scratch_need_ = std::max<int64_t>(1, (int64_t)chunk * plan.scratch_size);
  sblock_need_ = with_sblock ? std::max<int64_t>(1, (int64_t)chunk * plan.nS * plan.nS) : 0;
  bool any_wide = false;
  for (auto& L : plan.fwide_levels) any_wide |= !L.empty();
  batch.tmp_stride = 0; tmp_need_ = 0;
  if (any_wide) {
    // pivot-piece workspace of the multi-workgroup factorisation
    const int64_t np = (plan.max_w + dev::PIECE - 1) / dev::PIECE;
    batch.tmp_stride = np * 2 * dev::PIECE * dev::PIECE + 2LL * plan.max_w * dev::PIECE;
    tmp_need_ = (int64_t)chunk * batch.tmp_stride;
  }
}

void BatchedLU::upload(int64_t budget, bool with_sblock) {
  const int nb = (int)members.size();
  nent = (int32_t)plan.ent_id.size();
  std::vector<dev::FrontD> fd(plan.fronts.size());
  for (size_t s = 0; s < fd.size(); s++) {
    const Front& F = plan.fronts[s];
    fd[s] = dev::FrontD{F.c0, F.w, F.ri, F.rs, F.parent, F.idx_off, F.rel_off, F.c_off, F.a_off, F.lf_off,
                        F.ent_begin, F.ent_end, F.child_begin, F.child_end, F.f_off, F.lp_off, F.q_off};
  }
  auto keep = [&](auto* p) { owned.push_back((void*)p); return p; };
  dplan.nI = plan.nI; dplan.nS = plan.nS; dplan.nfronts = (int32_t)fd.size(); dplan.nent = nent;
  dplan.fronts = keep(dev::upload(fd));
  dplan.fidx = keep(dev::upload(plan.fidx));
  dplan.rel = keep(dev::upload(plan.rel));
  dplan.children = keep(dev::upload(plan.children));
  dplan.ent_id = keep(dev::upload(plan.ent_id));
  dplan.ent_pos = keep(dev::upload(plan.ent_pos));
  dplan.ent_w = keep(dev::upload(plan.ent_w));
  dplan.asm_ptr = keep(dev::upload(plan.asm_ptr));
  dplan.asm_src = keep(dev::upload(plan.asm_src));
  dplan.asm_rows = plan.asm_rows;
  dplan.fw_ptr = keep(dev::upload(plan.fw_ptr)); dplan.fw_items = keep(dev::upload(plan.fw_items));
  dplan.bw_ptr = keep(dev::upload(plan.bw_ptr)); dplan.bw_items = keep(dev::upload(plan.bw_items));
  {
    std::vector<dev::FwRec> rec(plan.fw_items.size());
    for (size_t t = 0; t < rec.size(); t++) {
      const int32_t item = plan.fw_items[t];
      const Front& F = plan.fronts[item >> 16];
      const int32_t a = F.a_off + (item & 0xffff);
      const int32_t nb = plan.asm_ptr[a], ne = plan.asm_ptr[a + 1];
      dev::FwRec r{item, (uint16_t)(ne - nb), {0, 0, 0, 0, 0}};
      if (ne - nb > 5 || plan.contrib_size > 65535) r.n = 0xffff;
      else for (int32_t q = nb; q < ne; q++) r.s[q - nb] = (uint16_t)plan.asm_src[q];
      rec[t] = r;
    }
    dplan.fw_rec = keep(dev::upload(rec));
  }
  dplan.nlev = (int32_t)plan.levels.size(); dplan.max_level_rows = plan.max_level_rows;
  dplan.s_ent_begin = plan.s_ent_begin; dplan.s_ent_end = nent;
  dplan.scratch_size = plan.scratch_size; dplan.factor_size = plan.factor_size;
  dplan.contrib_size = plan.contrib_size;
  dplan.packed = packed ? 1 : 0;
  {
    int32_t msr = 1;  // LDS vector of the one-workgroup-per-front solve kernels: small fronts only
    for (auto& F : plan.fronts) if (!F.big) msr = std::max(msr, F.w + F.ri);
    dplan.max_solve_rows = msr;
  }
  for (auto& L : plan.levels) d_lists.push_back(keep(dev::upload(L)));
  for (auto& L : plan.flevels) d_flists.push_back(keep(dev::upload(L)));
  plan_scratch(budget, with_sblock);
  batch.nb = nb;
  if (!h_ext.empty()) {
    HYMLS_CHECK(d_krow && d_kcol && (int64_t)h_ent_row.size() == nent && (int64_t)h_ext.size() == (int64_t)nb * n_ext, -3,
                "entry source lists: inconsistent tables");
    int32_t* src = (int32_t*)keep(dev::alloc(std::max<int64_t>(1, (int64_t)nb * nent) * sizeof(int32_t)));
    int32_t* d_ext = dev::upload(h_ext);
    int32_t* d_er = dev::upload(h_ent_row);
    int32_t* d_ec = dev::upload(h_ent_col);
    int32_t* d_fl = (int32_t*)dev::alloc(sizeof(int32_t));
    dev::zero(d_fl, sizeof(int32_t));
    dev::member_sources(nb, n_ext, nent, d_ext, d_er, d_ec, d_krow, d_kcol, src, d_fl);
    int32_t fl = 0;
    dev::d2h(&fl, d_fl, sizeof fl);
    dev::free(d_ext); dev::free(d_er); dev::free(d_ec); dev::free(d_fl);
    HYMLS_CHECK(fl == 0, -3, "entry of a subdomain pattern not found in the level matrix");
    batch.src = src;
    { rawvec<int32_t>().swap(h_ext); }
  } else {
    batch.src = keep(dev::upload(h_src));
  }
  batch.xoff = keep(dev::upload(h_xoff));
  const size_t fbytes = factor_bytes(nb, plan.factor_size);
  batch.factor = (double*)keep(pre_factor && pre_factor->bytes == fbytes ? pre_factor->take() : dev::alloc(fbytes));
  pre_factor.reset();
  // frontal scratch / separator blocks / pivot workspace are borrowed from the shared arena at factor time (plan_scratch)
  batch.scratch = nullptr; batch.sblock = nullptr;
  batch.contrib = (double*)keep(dev::alloc(std::max<int64_t>(1, (int64_t)contrib_nv * nb * plan.contrib_size) * sizeof(double)));
  batch.flag = (int32_t*)keep(dev::alloc(4 * sizeof(int32_t)));   // [flag bits | pad | largest growth factor (double bits)]
  dev::zero(batch.flag, 4 * sizeof(int32_t));
  h_fronts = fd;
  bool any_big = false, any_wide = false;
  for (auto& L : plan.big_levels) any_big |= !L.empty();
  for (auto& L : plan.fwide_levels) any_wide |= !L.empty();
  batch.tmp = nullptr; batch.swork = nullptr; batch.swork_stride = 0;
  (void)any_wide;
  if (any_big) {
    int64_t part_max = 0;
    for (auto& L : plan.big_levels) {
      d_big_lists.push_back(keep(dev::upload(L)));
      h_big_fronts.emplace_back();
      std::vector<int64_t> poff;
      int64_t off = 0;
      for (int s : L) {
        h_big_fronts.back().push_back(fd[s]);
        const int64_t w = fd[s].w, ri = fd[s].ri, kt = dev::solve_kt();
        const int64_t ctw = (w + kt - 1) / kt, ctr = (ri + kt - 1) / kt;
        poff.push_back(off);
        off += std::max(ctw * (w + ri), (ctw + ctr) * w);
      }
      part_max = std::max(part_max, off);
      d_big_poff.push_back(keep(dev::upload(poff)));
    }
    batch.swork_stride = (int64_t)plan.asm_rows + part_max;
    batch.swork = (double*)keep(dev::alloc((size_t)nb * batch.swork_stride * sizeof(double)));
  }
}

std::vector<dev::FrontD> BatchedLU::kids_of(int s) const {
  std::vector<dev::FrontD> k;
  for (int e = plan.fronts[s].child_begin; e < plan.fronts[s].child_end; e++) k.push_back(h_fronts[plan.children[e]]);
  return k;
}

void BatchedLU::bind_scratch() {
  double* a = (double*)dev::shared_scratch((size_t)(scratch_need_ + sblock_need_ + tmp_need_) * sizeof(double));
  batch.scratch = a;
  batch.sblock = sblock_need_ ? a + scratch_need_ : nullptr;
  batch.tmp = tmp_need_ ? a + scratch_need_ + sblock_need_ : nullptr;
}

void BatchedLU::factor_chunk(const double* kval, int32_t b0, int32_t nbc, bool spread_wide) {
  bind_scratch();
  if (batch.sblock) dev::sblock_init(dplan, batch, b0, nbc, kval);
  // spread_wide (one large system factored from the main stream: the last-level solver): the wide fronts of a tree level
  // are independent launch chains of small grids -- they go to the side streams, each with pivot-piece workspace of its own
  const bool spread = spread_wide && dev::side_streams() > 1 && !std::getenv("HYMLS_MI_NO_SIDE_STREAMS");
  struct MainStreamGuard { bool on; ~MainStreamGuard() { if (on) { try { dev::use_stream(0); } catch (...) {} } } } back_to_main{spread};
  for (size_t l = 0; l < plan.flevels.size(); l++) {
    int32_t mw = 1;
    for (int s : plan.flevels[l]) mw = std::max(mw, plan.fronts[s].w);
    dev::factor_level(dplan, batch, d_flists[l], (int32_t)plan.flevels[l].size(), b0, nbc, kval, mw);
    const auto& wide = plan.fwide_levels[l];
    if (spread && wide.size() > 1) {
      dev::fork_streams();
      for (size_t q = 0; q < wide.size(); q++) {
        dev::use_stream(1 + (int)(q % dev::side_streams()));
        dev::BatchD bq = batch;
        bq.tmp = tmp_need_ ? (double*)dev::shared_scratch((size_t)tmp_need_ * sizeof(double)) : nullptr;
        auto k = kids_of(wide[q]);
        dev::factor_big_front(dplan, bq, h_fronts[wide[q]], k.data(), (int32_t)k.size(), b0, nbc, kval);
      }
      dev::join_streams();
      continue;
    }
    for (int s : wide) {
      auto k = kids_of(s);
      dev::factor_big_front(dplan, batch, h_fronts[s], k.data(), (int32_t)k.size(), b0, nbc, kval);
    }
  }
  // the roots factored by k_factor_level add their updates to the separator block now, one after the other
  if (batch.sblock)
    for (size_t s = 0; s < plan.fronts.size(); s++)
      if (!plan.fronts[s].wide && plan.fronts[s].parent < 0 && plan.fronts[s].rs > 0) dev::root_update(dplan, batch, h_fronts[s], nbc);
}

void BatchedLU::repack_chunk(int32_t b0, int32_t nbc) {
  if (packed) dev::repack_fronts(dplan, batch, b0, nbc);
}

void BatchedLU::solve(double* x) const {
  const int nl = (int)plan.levels.size();
  for (int l = 0; l < nl; l++) {
    dev::solve_fwd_level(dplan, batch, d_lists[l], (int32_t)plan.levels[l].size(), x);
    if (!plan.big_levels[l].empty())
      dev::solve_fwd_big(dplan, batch, d_big_lists[l], h_big_fronts[l].data(), d_big_poff[l], (int32_t)plan.big_levels[l].size(), x);
  }
  for (int l = nl - 1; l >= 0; l--) {
    if (!plan.big_levels[l].empty())
      dev::solve_bwd_big(dplan, batch, d_big_lists[l], h_big_fronts[l].data(), d_big_poff[l], (int32_t)plan.big_levels[l].size(), x);
    dev::solve_bwd_level(dplan, batch, d_lists[l], (int32_t)plan.levels[l].size(), x);
  }
}

int32_t BatchedLU::check_flag(double* growth) const {
  int32_t f[4] = {0, 0, 0, 0};
  dev::d2h(f, batch.flag, sizeof f);
  if (growth) std::memcpy(growth, f + 2, sizeof(double));
  return f[0];
}

// ------------------------------------------------------------------ merged level-synchronous solve tables
// (device.hpp: solve_fwd_tasks / solve_bwd_tasks) for a set of (plan index, batch): one launch per tree level and sweep
// covers every (class, member, front) of that level; a task is a whole small front or a 64-row tile of a large one
bool merged_solve_fits(const ClassPlan& plan) {
  for (auto& F : plan.fronts) if (F.w + F.ri > dev::LVL_MAX_ROWS) return false;
  return true;
}

MergedSolve::~MergedSolve() { dev::free(d_subs); dev::free(d_fw); dev::free(d_bw); }

void MergedSolve::build(const std::vector<std::pair<int32_t, const BatchedLU*>>& classes) {
  dev::free(d_subs); dev::free(d_fw); dev::free(d_bw);
  d_subs = nullptr; d_fw = d_bw = nullptr;
  const int small_rows = std::getenv("HYMLS_MI_LVL_SMALL_ROWS") ? std::atoi(std::getenv("HYMLS_MI_LVL_SMALL_ROWS")) : dev::LVL_SMALL_ROWS;
  std::vector<dev::LvlSub> lsubs;
  std::vector<const ClassPlan*> sub_plan;
  max_nv = dev::NV_MAX;
  std::vector<std::vector<std::pair<int64_t, dev::LvlTask>>> fw, bw;   // per tree level: (cost, task)
  for (auto& cb : classes) {
    const BatchedLU& lu = *cb.second;
    const size_t nl = lu.plan.levels.size();
    if (fw.size() < nl) { fw.resize(nl); bw.resize(nl); }
    for (size_t b = 0; b < lu.members.size(); b++) {
      const int32_t sub = (int32_t)lsubs.size();
      lsubs.push_back(dev::LvlSub{lu.batch.factor + (int64_t)b * lu.plan.factor_size,
                                  lu.batch.contrib + (int64_t)b * lu.plan.contrib_size, lu.h_xoff[b], cb.first,
                                  (int64_t)lu.members.size() * lu.plan.contrib_size});
      max_nv = std::min(max_nv, lu.contrib_nv);
      sub_plan.push_back(&lu.plan);
      for (size_t l = 0; l < nl; l++) {
        auto add = [&](int s) {
          const Front& F = lu.plan.fronts[s];
          const int rows = F.w + F.ri;
          if (rows <= small_rows) {
            fw[l].push_back({(int64_t)rows * F.w, dev::LvlTask{sub, s, -1, 0}});
            bw[l].push_back({(int64_t)rows * F.w, dev::LvlTask{sub, s, -1, 0}});
          } else {
            for (int r0 = 0; r0 < rows; r0 += 64) fw[l].push_back({64LL * std::min(F.w, r0 + 63), dev::LvlTask{sub, s, r0, 0}});
            for (int r0 = 0; r0 < F.w; r0 += 64) bw[l].push_back({64LL * (F.w - r0 + F.ri), dev::LvlTask{sub, s, r0, 0}});
          }
        };
        for (int s : lu.plan.levels[l]) add(s);
        for (int s : lu.plan.big_levels[l]) add(s);
      }
    }
  }
  nsubs = (int32_t)lsubs.size();
  fw_off.assign(1, 0); bw_off.assign(1, 0); fw_lds.clear(); bw_lds.clear();
  if (!nsubs) return;
  std::vector<dev::LvlTask> tf, tb;
  auto heavy_first = [](const std::pair<int64_t, dev::LvlTask>& a, const std::pair<int64_t, dev::LvlTask>& b) { return a.first > b.first; };
  for (size_t l = 0; l < fw.size(); l++) {
    std::stable_sort(fw[l].begin(), fw[l].end(), heavy_first);
    std::stable_sort(bw[l].begin(), bw[l].end(), heavy_first);
    int32_t lf = 0, lb = 0;
    for (auto& t : fw[l]) {
      const Front& F = sub_plan[t.second.sub]->fronts[t.second.front];
      lf = std::max(lf, t.second.r0 < 0 ? F.w + F.ri : ((std::min(F.w, t.second.r0 + 63) + 7) & ~7) + 64 + 256);
      tf.push_back(t.second);
    }
    for (auto& t : bw[l]) {
      const Front& F = sub_plan[t.second.sub]->fronts[t.second.front];
      lb = std::max(lb, t.second.r0 < 0 ? F.w + F.ri : ((F.w - t.second.r0 + F.ri + 7) & ~7) + 256);
      tb.push_back(t.second);
    }
    fw_off.push_back((int32_t)tf.size()); bw_off.push_back((int32_t)tb.size());
    fw_lds.push_back(lf); bw_lds.push_back(lb);
    if (std::getenv("HYMLS_MI_VERBOSE")) {
      int64_t cf = 0, cbk = 0, mf = 0, mb = 0;
      for (auto& t : fw[l]) { cf += t.first; mf = std::max(mf, t.first); }
      for (auto& t : bw[l]) { cbk += t.first; mb = std::max(mb, t.first); }
      std::fprintf(stderr, "[hymls_mi] merged solve tree level %zu: forward %zu tasks %.1f MB (largest %.2f MB), backward %zu tasks %.1f MB (largest %.2f MB)\n",
                   l, fw[l].size(), 8e-6 * cf, 8e-6 * mf, bw[l].size(), 8e-6 * cbk, 8e-6 * mb);
    }
  }
  d_subs = dev::upload(lsubs);
  d_fw = dev::upload(tf); d_bw = dev::upload(tb);
}

// x <- A^{-1} x for every member of every class of the tables, nv columns with leading dimension ld; y: scratch (same shape)
void MergedSolve::solve(const dev::PlanD* d_plans, double* x, double* y, int64_t ld, int nv) const {
  const int nl = (int)fw_lds.size();
  for (int v0 = 0; v0 < nv; v0 += max_nv) {   // (column groups no wider than the contribution scratch of the batches)
    const int g = std::min(max_nv, nv - v0);
    double* xv = x + (int64_t)v0 * ld;
    double* yv = y + (int64_t)v0 * ld;
    for (int l = 0; l < nl; l++)
      dev::solve_fwd_tasks_mv(d_fw + fw_off[l], fw_off[l + 1] - fw_off[l], d_subs, d_plans, fw_lds[l], xv, yv, ld, g);
    for (int l = nl - 1; l >= 0; l--)
      dev::solve_bwd_tasks_mv(d_bw + bw_off[l], bw_off[l + 1] - bw_off[l], d_subs, d_plans, bw_lds[l], yv, xv, ld, g);
  }
}

// ------------------------------------------------------------------ DirectSolver
static double wall();
// CoarseSolver::Compute (reference src/HYMLS_CoarseSolver.cpp:131-152), value part: dropping, Dirichlet rows of the fixed
// gids, the pressure node that joins a pending border.  fix_rows: local rows of the fixed gids.
Csr DirectSolver::prepare(const Csr& A0, const ivec& gids, const ivec& fix_gids, const Params& cp, bool border_pending, ivec& fix_rows) {
  Csr A = drop_by_value(A0, SMALL_ENTRY, 2);
  const int32_t n = A.n;
  fix_rows.clear();
  for (int32_t g : fix_gids) {
    int lid = -1;
    for (int i = 0; i < n; i++) if (gids[i] == g) { lid = i; break; }
    HYMLS_CHECK(lid >= 0, -2, "fix GID " + std::to_string(g) + " not in matrix row map");
    // PutDirichlet (reference src/HYMLS_MatrixUtils.cpp:1229-1309)
    for (int i = 0; i < n; i++)
      for (int e = A.rowptr[i]; e < A.rowptr[i + 1]; e++) {
        if (i == lid) A.val[e] = (A.col[e] == lid) ? 1.0 : 0.0;
        else if (A.col[e] == lid) A.val[e] = 0.0;
      }
    fix_rows.push_back(lid);
  }
  tail_z_.clear();
  if (border_pending && fix_gids.empty() && n > 0) {
    // the last pressure node (if the problem has pressures) joins the border: its row and column are kept aside and
    // replaced by a Dirichlet row in the matrix that is factored.  On a grid with periodic directions the constant of
    // every velocity component can be a null vector as well (the reference's periodic Stokes3D has Neumann velocity
    // Laplacians, GaleriExt_Stokes3D.h:77-80; its driver borders with the dof constants, "Null Space Type" = "Constant",
    // HYMLS_MainUtils.cpp:361-376): the last node of each velocity component joins the border too.
    auto last_of = [&](int32_t vt) { for (int i = n - 1; i >= 0; i--) if (cp.vtype[gids[i] % cp.dof] == vt) return i; return -1; };
    std::vector<int32_t> kinds = {VT_P};
    if (cp.perio[0] || cp.perio[1] || cp.perio[2]) for (int32_t vt : {VT_U, VT_V, VT_W, VT_LAPLACE}) kinds.push_back(vt);
    for (int32_t vt : kinds) { const int z = last_of(vt); if (z >= 0) tail_z_.push_back(z); }
    const int tl = (int)tail_z_.size();
    if (tl) {
      ivec pos(n, -1);
      for (int t = 0; t < tl; t++) pos[tail_z_[t]] = t;
      tail_col_.assign((size_t)n * tl, 0.0); tail_row_.assign((size_t)n * tl, 0.0); tail_d_.assign((size_t)tl * tl, 0.0);
      for (int i = 0; i < n; i++)
        for (int e = A.rowptr[i]; e < A.rowptr[i + 1]; e++) {
          const int c = A.col[e], ti = pos[i], tc = pos[c];
          if (ti >= 0 && tc >= 0) { tail_d_[ti + (size_t)tl * tc] = A.val[e]; A.val[e] = (i == c) ? 1.0 : 0.0; }
          else if (ti >= 0) { tail_row_[c + (size_t)n * ti] = A.val[e]; A.val[e] = 0.0; }
          else if (tc >= 0) { tail_col_[i + (size_t)n * tc] = A.val[e]; A.val[e] = 0.0; }
        }
    }
  }
  return A;
}

static std::vector<char> zero_diagonal(const Csr& A) {
  std::vector<char> zd((size_t)A.n, 1);
  for (int i = 0; i < A.n; i++)
    for (int e = A.rowptr[i]; e < A.rowptr[i + 1]; e++)
      if (A.col[e] == i && A.val[e] != 0.0) zd[i] = 0;
  return zd;
}

void DirectSolver::numeric(const vvec& val) {
  if (!d_val_) d_val_ = dev::upload(val);
  else dev::h2d(d_val_, val.data(), val.size() * sizeof(double));
  lu_->factor_chunk(d_val_, 0, 1, true);
  double g = 0.0;
  const int32_t f = lu_->check_flag(&g);
  if (std::getenv("HYMLS_MI_VERBOSE")) std::fprintf(stderr, "[hymls_mi] coarse solver: largest element growth of a pivot block %.3g\n", g);
  HYMLS_CHECK((f & 1) == 0, -4, "coarse factorisation hit a zero or non-finite pivot");
  HYMLS_CHECK((f & 2) == 0, -4, "coarse factorisation without pivoting is unstable for this matrix: element growth " + std::to_string(g) + " > 1e8");
}

DirectSolver::DirectSolver(const Csr& A0, const ivec& gids, const ivec& fix_gids, int64_t ngid,
                           const Params& cp, const ivec* clu_ptr, const ivec* clu, const ivec* clu_coord, bool border_pending) {
  label_level_ = cp.level + 2;   // (cp: the parameters of the level above; the reference counts its levels from 1)
  dev::Range range("CoarseSolver", label_level_, "Compute");
  ivec fix_rows;
  Csr A = prepare(A0, gids, fix_gids, cp, border_pending, fix_rows);
  n_ = A.n;
  border_pending_ = border_pending;
  if (n_ == 0) return;
  LocalPattern lp;
  lp.nI = n_; lp.nS = 0;
  lp.rowptr = A.rowptr; lp.col.assign(A.col.begin(), A.col.end());
  pat_zero_diag_ = zero_diagonal(A);
  lp.zero_diag.assign(pat_zero_diag_.begin(), pat_zero_diag_.end());
  lp.coord.resize(3 * (size_t)n_);
  for (int i = 0; i < n_; i++) gid_coord(cp, gids[i], &lp.coord[3 * (size_t)i]);
  (void)ngid;
  if (clu_ptr && !clu_ptr->empty() && !std::getenv("HYMLS_MI_NO_CLUSTER_ND")) { lp.clu_ptr = *clu_ptr; lp.clu = *clu; lp.clu_coord = *clu_coord; }
  lu_.reset(new BatchedLU());
  const double t_an = wall();
  static const int leaf_coarse = std::getenv("HYMLS_MI_LEAF_SIZE_COARSE") ? std::max(1, std::atoi(std::getenv("HYMLS_MI_LEAF_SIZE_COARSE"))) : LEAF_SIZE_COARSE;
  lu_->plan = analyse_class(lp, leaf_coarse, MAX_WIDTH_COARSE, 65536);
  if (std::getenv("HYMLS_MI_VERBOSE")) {
    std::fprintf(stderr, "[hymls_mi] coarse solver: ordering + symbolic factorisation + plan %.2f s (host)\n", wall() - t_an);
    print_plan_stats(lu_->plan, "coarse solver", 1);
  }
  if (const char* dump = std::getenv("HYMLS_MI_DUMP_COARSE")) {
    // development aid: the matrix that is factored (CSR) and the elimination order, as raw little-endian arrays
    if (FILE* f = std::fopen(dump, "wb")) {
      const int64_t hdr[3] = {n_, (int64_t)A.col.size(), (int64_t)lu_->plan.fronts.size()};
      std::fwrite(hdr, sizeof(int64_t), 3, f);
      std::fwrite(A.rowptr.data(), sizeof(int32_t), A.rowptr.size(), f);
      std::fwrite(A.col.data(), sizeof(int32_t), A.col.size(), f);
      std::fwrite(A.val.data(), sizeof(double), A.val.size(), f);
      std::fwrite(lu_->plan.perm.data(), sizeof(int32_t), lu_->plan.perm.size(), f);
      for (auto& F : lu_->plan.fronts) { const int32_t t[4] = {F.c0, F.w, F.ri, F.rs}; std::fwrite(t, sizeof(int32_t), 4, f); }
      std::fclose(f);
    }
  }
  lu_->members = {0};
  lu_->h_xoff = {0};
  lu_->contrib_nv = dev::NV_MAX;
  // entry e of the extended CSR is entry e of A
  lu_->h_src.resize(A.col.size());
  std::iota(lu_->h_src.begin(), lu_->h_src.end(), 0);
  const double t_up = wall();
  lu_->upload(SCRATCH_BUDGET, false);
  if (std::getenv("HYMLS_MI_VERBOSE")) std::fprintf(stderr, "[hymls_mi] coarse solver: plan upload + allocations %.2f s\n", wall() - t_up);
  numeric(A.val);
  d_perm_ = dev::upload(lu_->plan.perm);
  // the tree levels of one large system are launch-latency bound with one launch chain per level (assemble, panels,
  // finalize, small fronts: about 100 launches per solve at 216 k unknowns); the merged task kernels need one launch
  // per tree level and sweep
  if (merged_solve_fits(lu_->plan) && !std::getenv("HYMLS_MI_NO_MERGED_SOLVE")) {
    merged_.build({{0, lu_.get()}});
    d_plan_ = dev::upload(std::vector<dev::PlanD>{lu_->dplan});
  }
  for (int lid : fix_rows)
    if (lid > 0) fix_lids_.push_back(lu_->plan.iperm[lid]);  // CoarseSolver.cpp:288-289: lid 0 is not zeroed
  d_fix_ = dev::upload(fix_lids_);
  pat_rowptr_ = std::move(A.rowptr); pat_col_ = std::move(A.col); pat_gids_ = gids; pat_fix_ = fix_gids;
}

// Compute with an unchanged pattern (a Newton step: same rows, same kept entries, same zero diagonals): the ordering, the
// symbolic factorisation and every index table on the device stay, only the numeric factorisation is redone.  (The
// reference's CoarseSolver keeps its Amesos solver across Compute calls in the same way, CoarseSolver.cpp:131-152.)
bool DirectSolver::refactor(const Csr& A0, const ivec& gids, const ivec& fix_gids, const Params& cp, bool border_pending) {
  if (n_ == 0 || !lu_ || border_pending != border_pending_ || gids != pat_gids_ || fix_gids != pat_fix_) return false;
  dev::Range range("CoarseSolver", label_level_, "Compute");
  const ivec tail_before = tail_z_;
  ivec fix_rows;
  Csr A = prepare(A0, gids, fix_gids, cp, border_pending, fix_rows);
  if (A.n != n_ || tail_z_ != tail_before || A.rowptr != pat_rowptr_ || A.col != pat_col_ || zero_diagonal(A) != pat_zero_diag_) return false;
  if (std::getenv("HYMLS_MI_VERBOSE")) std::fprintf(stderr, "[hymls_mi] coarse solver: pattern unchanged, numeric refactorisation with the existing plan\n");
  numeric(A.val);
  return true;
}

DirectSolver::~DirectSolver() {
  dev::free(d_val_); dev::free(d_z_); dev::free(d_perm_); dev::free(d_fix_); dev::free(d_bZ_); dev::free(d_bW_);
  dev::free(d_plan_); dev::free(d_y_);
}

void DirectSolver::solve(const double* b, double* x, bool zero_fixed) { solve_mv(b, n_, x, n_, 1, zero_fixed); }

void DirectSolver::solve_mv(const double* b, int64_t ldb, double* x, int64_t ldx, int nv, bool zero_fixed) {
  if (n_ == 0) return;
  dev::Range range("CoarseSolver", label_level_, "ApplyInverse");
  if (nv > nv_alloc_) {
    dev::sync();
    dev::free(d_z_); dev::free(d_y_);
    d_z_ = (double*)dev::alloc((size_t)n_ * nv * sizeof(double));
    d_y_ = merged_.nsubs > 0 ? (double*)dev::alloc((size_t)n_ * nv * sizeof(double)) : nullptr;
    nv_alloc_ = nv;
  }
  for (int v = 0; v < nv; v++) {
    dev::gather(n_, d_perm_, b + v * ldb, d_z_ + (size_t)v * n_);
    if (zero_fixed && !fix_lids_.empty())   // zero the Dirichlet right-hand sides: scatter zeros
      dev::scatter((int64_t)fix_lids_.size(), d_fix_, dev::zeros16(), d_z_ + (size_t)v * n_);
  }
  if (merged_.nsubs > 0) merged_.solve(d_plan_, d_z_, d_y_, n_, nv);
  else for (int v = 0; v < nv; v++) lu_->solve(d_z_ + (size_t)v * n_);
  for (int v = 0; v < nv; v++) dev::scatter(n_, d_perm_, d_z_ + (size_t)v * n_, x + v * ldx);
}

void DirectSolver::apply_inverse(const double* b, double* x) { solve(b, x, true); }
void DirectSolver::apply_inverse_mv(const double* b, int64_t ldb, double* x, int64_t ldx, int nv) { solve_mv(b, ldb, x, ldx, nv, true); }

// CoarseSolver with a border (reference src/HYMLS_CoarseSolver.cpp:196-260,454-560): the reference factors the
// AugmentedMatrix [A V; W' C]; here A (with its Dirichlet fixes) is already factored, so the same solution is
// obtained by block elimination: Z = A^{-1} V once, then y = (C - W' Z)^{-1} (T - W' A^{-1} b), x = A^{-1} b - Z y.
// (A has to be nonsingular, i.e. "Fix Pressure Level" stays on for Stokes problems.)
void DirectSolver::set_border(int m, const double* dV, const double* dW, const double* C) {
  dev::free(d_bZ_); dev::free(d_bW_);
  d_bZ_ = d_bW_ = nullptr;
  bm_ = 0; bMinv_.clear();
  if (m <= 0 || n_ == 0) return;
  bm_ = m;
  const int tl = (int)tail_z_.size(), mt = m + tl;
  const size_t n = (size_t)n_;
  // border columns [tail columns | V], border rows [tail rows | W]; entries at the tail nodes go to the small block D
  dvec D((size_t)mt * mt, 0.0);
  d_bZ_ = (double*)dev::alloc(n * mt * sizeof(double));
  d_bW_ = (double*)dev::alloc(n * mt * sizeof(double));
  double* d_U = (double*)dev::alloc(n * mt * sizeof(double));
  if (tl) {
    dev::h2d(d_U, tail_col_.data(), n * tl * sizeof(double));
    dev::h2d(d_bW_, tail_row_.data(), n * tl * sizeof(double));
    for (int j = 0; j < tl; j++) for (int i = 0; i < tl; i++) D[i + (size_t)mt * j] = tail_d_[i + (size_t)tl * j];
  }
  dev::d2d(d_U + n * tl, dV, n * m * sizeof(double));
  dev::d2d(d_bW_ + n * tl, dW, n * m * sizeof(double));
  for (int j = 0; j < m; j++)
    for (int i = 0; i < m; i++) D[(i + tl) + (size_t)mt * (j + tl)] = C[i + (size_t)m * j];
  const double zero = 0.0;
  for (int t = 0; t < tl; t++)
    for (int j = 0; j < m; j++) {
      dev::d2h(&D[t + (size_t)mt * (j + tl)], d_U + n * (j + tl) + tail_z_[t], sizeof(double));      // V[z_t, j]
      dev::d2h(&D[(j + tl) + (size_t)mt * t], d_bW_ + n * (j + tl) + tail_z_[t], sizeof(double));    // W[z_t, j]
      dev::h2d(d_U + n * (j + tl) + tail_z_[t], &zero, sizeof(double));
      dev::h2d(d_bW_ + n * (j + tl) + tail_z_[t], &zero, sizeof(double));
    }
  for (int j = 0; j < mt; j++) solve(d_U + n * j, d_bZ_ + n * j, false);                  // Z = A^{-1} U
  dev::free(d_U);
  // M = D - W' Z, inverted on the host with partial pivoting
  dvec M = D;
  for (int j = 0; j < mt; j++)
    for (int i = 0; i < mt; i++) M[i + (size_t)mt * j] -= dev::dot(n_, d_bW_ + n * i, d_bZ_ + n * j);
  bMinv_.assign((size_t)mt * mt, 0.0);
  for (int i = 0; i < mt; i++) bMinv_[i + (size_t)mt * i] = 1.0;
  for (int k = 0; k < mt; k++) {
    int p = k;
    for (int i = k + 1; i < mt; i++) if (std::abs(M[i + (size_t)mt * k]) > std::abs(M[p + (size_t)mt * k])) p = i;
    HYMLS_CHECK(M[p + (size_t)mt * k] != 0.0 && std::isfinite(M[p + (size_t)mt * k]), -4, "singular bordered coarse system");
    if (p != k) for (int j = 0; j < mt; j++) { std::swap(M[k + (size_t)mt * j], M[p + (size_t)mt * j]); std::swap(bMinv_[k + (size_t)mt * j], bMinv_[p + (size_t)mt * j]); }
    const double ip = 1.0 / M[k + (size_t)mt * k];
    for (int j = 0; j < mt; j++) { M[k + (size_t)mt * j] *= ip; bMinv_[k + (size_t)mt * j] *= ip; }
    for (int i = 0; i < mt; i++) {
      if (i == k) continue;
      const double f = M[i + (size_t)mt * k];
      if (f == 0.0) continue;
      for (int j = 0; j < mt; j++) { M[i + (size_t)mt * j] -= f * M[k + (size_t)mt * j]; bMinv_[i + (size_t)mt * j] -= f * bMinv_[k + (size_t)mt * j]; }
    }
  }
}

void DirectSolver::apply_inverse_bordered(const double* b, const double* T, double* x, double* S) {
  if (bm_ == 0) { apply_inverse(b, x); return; }
  const int tl = (int)tail_z_.size(), mt = bm_ + tl;
  const size_t n = (size_t)n_;
  dvec r(mt), y(mt);
  if (tl) {
    // the equations of the tail nodes belong to the border: their right-hand side entries move there
    dev::d2d(x, b, n * sizeof(double));          // (x doubles as the modified right-hand side)
    const double zero = 0.0;
    for (int t = 0; t < tl; t++) { dev::d2h(&r[t], b + tail_z_[t], sizeof(double)); dev::h2d(x + tail_z_[t], &zero, sizeof(double)); }
    solve(x, x, false);
  } else {
    solve(b, x, false);                           // (the augmented system of the reference does not zero the fixed rows)
  }
  for (int i = 0; i < bm_; i++) r[i + tl] = T[i];
  for (int i = 0; i < mt; i++) r[i] -= dev::dot(n_, d_bW_ + n * i, x);
  for (int i = 0; i < mt; i++) { y[i] = 0.0; for (int j = 0; j < mt; j++) y[i] += bMinv_[i + (size_t)mt * j] * r[j]; }
  for (int j = 0; j < mt; j++) dev::axpby(n_, -y[j], d_bZ_ + n * j, 1.0, x);
  for (int t = 0; t < tl; t++) dev::h2d(x + tail_z_[t], &y[t], sizeof(double));
  for (int i = 0; i < bm_; i++) S[i] = y[i + tl];
}

void DirectSolver::add_stats(ApplyStats& st, bool) const {
  if (!lu_) return;
  st.bytes_coarse += 8.0 * lu_->plan.nnz_factor + 8.0 * 4 * n_;
  st.bytes_coarse_sparse += 12.0 * lu_->plan.nnz_sparse + 24.0 * n_ + 8.0 * 4 * n_;
  st.flops_factor += (double)lu_->plan.flops_factor;
}

// ------------------------------------------------------------------ LevelSolver
struct LevelSolver::Cls {
  BatchedLU lu;
  LocalPattern pat;
  ivec lgptr;                 // offsets of every group in the subdomain's separator list
  std::vector<ivec> llinked;  // linked sets over all groups (indices)
  ivec key_extra;             // group types etc. for the class key
  ivec mult;                  // per entry multiplicity (class key)
  ivec pick;                  // kept entries of the separator block
  std::vector<int64_t> blk_off;  // per linked set: offset in the extraction record (-1: no rows)
  ivec blk_len;
  int64_t ext_size = 0, ext_base = 0;
  int32_t ngl = 0;
  int32_t* d_pick = nullptr;
  int32_t* d_lgptr = nullptr;
  int32_t *d_glink = nullptr, *d_goff = nullptr, *d_lblen = nullptr;   // linked-set tables of the kept-entries kernel
  int64_t* d_lboff = nullptr;
  dvec tvloc;
  double* d_tvloc = nullptr;
  ~Cls() { dev::free(d_pick); dev::free(d_lgptr); dev::free(d_tvloc); dev::free(d_glink); dev::free(d_goff); dev::free(d_lblen); dev::free(d_lboff); }
};

void make_local_csr(int64_t nrows, const int32_t* row_gids, const int32_t* rowptr, const int32_t* col_gids,
                    const double* val, int64_t ngid, Csr& K, ivec& gids) {
  ivec g2l(ngid, -1);
  for (int64_t i = 0; i < nrows; i++) {
    HYMLS_CHECK(row_gids[i] >= 0 && row_gids[i] < ngid && g2l[row_gids[i]] < 0, -2, "row gid out of range or given twice");
    g2l[row_gids[i]] = (int32_t)i;
  }
  const int64_t nnz = rowptr[nrows];
  ivec ghosts;
  for (int64_t e = 0; e < nnz; e++) {
    const int32_t c = col_gids[e];
    HYMLS_CHECK(c >= 0 && c < ngid, -2, "column gid out of range");
    if (g2l[c] == -1) { g2l[c] = -2; ghosts.push_back(c); }
  }
  std::sort(ghosts.begin(), ghosts.end());
  gids.assign(row_gids, row_gids + nrows);
  for (int32_t g : ghosts) { g2l[g] = (int32_t)gids.size(); gids.push_back(g); }
  K.n = (int32_t)gids.size();
  K.rowptr.assign(K.n + 1, (int32_t)nnz);
  std::copy(rowptr, rowptr + nrows + 1, K.rowptr.begin());
  K.col.resize(nnz);
  for (int64_t e = 0; e < nnz; e++) K.col[e] = g2l[col_gids[e]];
  K.val.assign(val, val + nnz);
}

LevelSolver::LevelSolver(const Params& p, int level, int64_t ngid, const Comm* comm)
    : p_(p), level_(level), ngid_(ngid), comm_(comm) {}

LevelSolver::~LevelSolver() {
  void* ptrs[] = {d_kval_, d_krow_, d_kcol_, d_inperm_, d_z_, d_t1_, d_t2_, d_y2_, d_a12_row_, d_a12_col_,
                  d_a12_src_, d_a21_row_, d_a21_col_, d_a21_src_, d_a12_val_, d_a21_val_, d_gptr_, d_otw_,
                  d_vs_, d_red_pull_ptr_, d_red_pull_idx_, d_red_val_, d_ext_, d_vrhs_, d_vsol_, d_yb_, d_flag_,
                  d_nrhs_, d_nsol_};
  for (void* q : ptrs) dev::free(q);
  dev::free(d_fsubs_); dev::free(d_fplans_);
  dev::free(d_blkd_); dev::free(d_blka_);
  { void* bp[] = {d_bVu_, d_bWu_, d_bW1_, d_bQ1_, d_bSV_, d_bSW_, d_bNV_, d_bNW_, d_btmp_, d_a12t_row_, d_a12t_col_, d_a12t_src_, d_a12t_val_};
    for (void* q : bp) dev::free(q);
    for (int32_t* q : d_orders_) dev::free(q); }
  dev::free(d_mv_row_); dev::free(d_mv_col_); dev::free(d_mv_src_); dev::free(d_mv_node_); dev::free(d_mv_val_); dev::free(d_mv_x_);
  dev::free(d_ytmp_);
  for (auto& b : blocks_) { dev::free(b.d_binv); dev::free(b.d_ids); dev::free(b.d_pull_ptr); dev::free(b.d_pull_base); }
}

void LevelSolver::set_rows(const Csr& K, const ivec& gids, const dvec& tv, int32_t nrows) {
  // (copies on the setup threads: 12 GB at the finest level of a 256^3 run)
  K_.n = K.n;
  parallel_assign(K_.rowptr, K.rowptr.data(), K.rowptr.size());
  parallel_assign(K_.col, K.col.data(), K.col.size());
  parallel_assign(K_.val, K.val.data(), K.val.size());
  parallel_assign(gids_, gids.data(), gids.size());
  parallel_assign(tv_, tv.data(), tv.size());
  nrows_ = nrows;
  HYMLS_CHECK((int)gids_.size() == K_.n && nrows_ <= K_.n, -2, "level: inconsistent sizes");
  tv_.resize(K_.n, 1.0);
}

// which subdomains live here.  One rank: all of them.  Sharded: the subdomains whose reference
// corner lies in this rank's box of the grid (the boxes of CreatePIDMap, reference
// src/HYMLS_BasePartitioner.cpp:361-586), plus the halo of subdomains of other ranks that share a
// separator node with them (needed for the ownership rule "first subdomain that lists the group",
// src/HYMLS_HierarchicalMap.cpp:261-271, for the A22 multiplicities and for the Schur contributions).
void LevelSolver::partition(const ivec* level_gids) {
  if (partitioned_) return;
  const int nsd = num_subdomains(p_);
  const bool dist = comm_->distributed();
  std::vector<char> present;
  if (level_gids) { present.assign(ngid_, 0); for (int32_t g : *level_gids) present[g] = 1; }
  sd_rank_.assign(nsd, 0);
  std::vector<char> cand;
  if (dist) {
    cand.assign(nsd, 0);
    const int bx = (p_.nx + comm_->px - 1) / comm_->px, by = (p_.ny + comm_->py - 1) / comm_->py,
              bz = (p_.nz + comm_->pz - 1) / comm_->pz;
    const int rx = comm_->rank % comm_->px, ry = (comm_->rank / comm_->px) % comm_->py, rz = comm_->rank / (comm_->px * comm_->py);
    const int mx = 2 * p_.sx + 3, my = 2 * p_.sy + 3, mz = 2 * p_.sz + 3;
    // few subdomains (coarser levels): no geometric prefilter, every subdomain is a halo candidate
    const int all_below = std::getenv("HYMLS_MI_HALO_ALL_BELOW") ? std::atoi(std::getenv("HYMLS_MI_HALO_ALL_BELOW")) : 4096;
    for (int s = 0; s < nsd; s++) {
      int x, y, z;
      sd_position(p_, s, x, y, z);
      const int cx = std::min(std::max(x, 0), p_.nx - 1), cy = std::min(std::max(y, 0), p_.ny - 1),
                cz = std::min(std::max(z, 0), p_.nz - 1);
      sd_rank_[s] = ((cz / bz) * comm_->py + cy / by) * comm_->px + cx / bx;
      // (a periodic direction: the window around the rank's box wraps around the grid)
      auto near = [](int v, int lo, int hi, int n, bool per) {
        return (v >= lo && v < hi) || (per && ((v + n >= lo && v + n < hi) || (v - n >= lo && v - n < hi)));
      };
      cand[s] = nsd <= all_below || (near(x, rx * bx - mx, (rx + 1) * bx + mx, p_.nx, p_.perio[0]) &&
                                     near(y, ry * by - my, (ry + 1) * by + my, p_.ny, p_.perio[1]) &&
                                     near(z, rz * bz - mz, (rz + 1) * bz + mz, p_.nz, p_.perio[2]));
    }
  }
  hm_ = build_hiermap(p_, level_gids ? &present : nullptr, dist ? &cand : nullptr);
  my_sds_.clear(); halo_sds_.clear();
  for (int s = 0; s < nsd; s++) if (sd_rank_[s] == comm_->rank) my_sds_.push_back(s);
  if (dist) {
    std::vector<char> mark(ngid_, 0);
    for (int s : my_sds_) for (auto& g : hm_.sd[s].groups) for (int32_t x : g.nodes) mark[x] = 1;
    for (int s = 0; s < nsd; s++) {
      if (!cand[s] || sd_rank_[s] == comm_->rank) continue;
      bool touches = false;
      for (auto& g : hm_.sd[s].groups) { for (int32_t x : g.nodes) if (mark[x]) { touches = true; break; } if (touches) break; }
      if (touches) halo_sds_.push_back(s);
      else hm_.sd[s] = Subdomain();
    }
  }
  partitioned_ = true;
}

ivec LevelSolver::required_gids() const {
  ivec r;
  for (int s : my_sds_) {
    const Subdomain& S = hm_.sd[s];
    r.insert(r.end(), S.interior.begin(), S.interior.end());
    for (auto& g : S.groups) r.insert(r.end(), g.nodes.begin(), g.nodes.end());
  }
  std::sort(r.begin(), r.end());
  r.erase(std::unique(r.begin(), r.end()), r.end());
  return r;
}

// keep the rows this rank needs, turn everything else it references into ghost columns
void LevelSolver::localize() {
  ivec req = required_gids();
  std::vector<char> need(ngid_, 0);
  for (int32_t g : req) need[g] = 1;
  ivec rows, rp(1, 0), cg;
  dvec va, tv;
  keep_entries_.clear();
  given_nnz_ = K_.val.size();
  for (int i = 0; i < nrows_; i++) {
    if (!need[gids_[i]]) continue;
    need[gids_[i]] = 2;
    rows.push_back(gids_[i]);
    tv.push_back(tv_[i]);
    for (int e = K_.rowptr[i]; e < K_.rowptr[i + 1]; e++) { cg.push_back(gids_[K_.col[e]]); va.push_back(K_.val[e]); keep_entries_.push_back(e); }
    rp.push_back((int32_t)cg.size());
  }
  for (int32_t g : req) HYMLS_CHECK(need[g] == 2, -2, "rank " + std::to_string(comm_->rank) + " was not given the row of gid " +
                                                          std::to_string(g) + " (see hymls_mi_required_rows)");
  Csr K; ivec gids;
  make_local_csr((int64_t)rows.size(), rows.data(), rp.data(), cg.data(), va.data(), ngid_, K, gids);
  nrows_ = (int32_t)rows.size();
  K_ = std::move(K); gids_ = std::move(gids);
  tv.resize(K_.n, 1.0);
  tv_ = std::move(tv);
}

static double wall() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// HYMLS_MI_VERBOSE=2: wall time of the steps inside the phases of Initialize (host work; development aid)
struct FineLap {
  double t;
  bool on;
  FineLap() : t(wall()), on(std::getenv("HYMLS_MI_VERBOSE") && std::atoi(std::getenv("HYMLS_MI_VERBOSE")) >= 2) {}
  void operator()(const char* what) {
    if (!on) return;
    const double n = wall();
    std::fprintf(stderr, "[hymls_mi]       . %-40s %.3f s\n", what, n - t);
    t = n;
  }
};

void LevelSolver::initialize() {
  dev::Range range("Preconditioner", level_ + 1, "Initialize");
  const bool dist = comm_->distributed();
  const bool verbose = std::getenv("HYMLS_MI_VERBOSE") != nullptr;
  double t0 = wall();
  auto lap = [&](const char* what) {
    if (verbose) std::fprintf(stderr, "[hymls_mi] rank %d level %d: %-28s %.2f s\n", comm_->rank, level_, what, wall() - t0);
    t0 = wall();
  };
  if (level_ == 0) partition(nullptr);
  else { ivec lg(gids_.begin(), gids_.begin() + nrows_); partition(&lg); }
  lap("partition");
  if (dist) localize();
  FineLap fine;
  const int n = K_.n;
  HYMLS_CHECK((int)gids_.size() == n && (int)tv_.size() == n, -2, "level: inconsistent sizes");
  g2l_.assign(ngid_, -1);
  parallel_for(n, [&](int64_t i) { g2l_[gids_[i]] = (int32_t)i; }, 1 << 16);
  // separator numbering: owned groups of this rank's subdomains (sd 0, sd 1, ...: map2 of the reference),
  // then the groups of its subdomains that another rank owns (ghosts)
  pos2_.assign(n, -1);
  intidx_.assign(n, -1);
  sep_row_.clear();
  gptr_.assign(1, 0);
  fine("g2l, pos2, intidx tables");
  {
    // group pointers first (sequential, one entry per owned group), then the nodes of every subdomain in parallel
    std::vector<int64_t> g0(my_sds_.size() + 1, 0);
    for (size_t k = 0; k < my_sds_.size(); k++) {
      const Subdomain& S = hm_.sd[my_sds_[k]];
      for (int gi : S.owned) gptr_.push_back(gptr_.back() + (int32_t)S.groups[gi].nodes.size());
      g0[k + 1] = (int64_t)gptr_.size() - 1;
    }
    n2_ = gptr_.back();
    sep_row_.assign((size_t)n2_, -1);
    std::atomic<int> bad{0};
    parallel_for((int64_t)my_sds_.size(), [&](int64_t k) {
      const Subdomain& S = hm_.sd[my_sds_[k]];
      int64_t g = g0[k];
      for (int gi : S.owned) {
        int32_t o = gptr_[g++];
        for (int32_t x : S.groups[gi].nodes) {
          const int r = g2l_[x];
          if (!(r >= 0 && r < nrows_)) { bad = 1; continue; }
          sep_row_[o] = r;
          // (a node listed by two owned groups would be written twice: caught below)
          pos2_[r] = o++;
        }
      }
    }, 16);
    HYMLS_CHECK(bad == 0, -3, "separator node listed twice or missing");
    parallel_for(n2_, [&](int64_t k) { if (sep_row_[k] < 0 || pos2_[sep_row_[k]] != (int32_t)k) bad = 1; }, 1 << 16);
    HYMLS_CHECK(bad == 0, -3, "separator node listed twice or missing");
  }
  // ghost separators and who owns them
  std::vector<std::vector<int64_t>> want_sep(comm_->size);
  std::vector<ivec> dst_sep(comm_->size);
  if (dist) {
    std::unordered_map<int32_t, int32_t> first_lister;   // first node of a group -> first subdomain listing it
    for (int s = 0; s < (int)hm_.sd.size(); s++)
      for (int gi : hm_.sd[s].owned) first_lister.emplace(hm_.sd[s].groups[gi].nodes[0], s);
    for (int s : my_sds_)
      for (auto& g : hm_.sd[s].groups) {
        const int r0 = g2l_[g.nodes[0]];
        HYMLS_CHECK(r0 >= 0, -3, "separator node without a row");
        if (pos2_[r0] >= 0) continue;
        auto it = first_lister.find(g.nodes[0]);
        HYMLS_CHECK(it != first_lister.end() && sd_rank_[it->second] != comm_->rank, -3, "separator group without owner");
        const int q = sd_rank_[it->second];
        for (int32_t x : g.nodes) {
          const int r = g2l_[x];
          HYMLS_CHECK(r >= 0 && r < nrows_ && pos2_[r] < 0, -3, "ghost separator node listed twice or missing");
          pos2_[r] = (int32_t)sep_row_.size();
          want_sep[q].push_back(x);
          dst_sep[q].push_back((int32_t)sep_row_.size());
          sep_row_.push_back(r);
        }
      }
  }
  ngs_ = (int32_t)sep_row_.size() - n2_;
  direct_schur_ = level_ >= p_.levels;
  lap("separator numbering");
  build_classes();
  lap("pattern classes + plans");
  fine.t = wall();
  // owned rows = layout of the vectors handed to apply_inverse (order of the rows as they were given)
  {
    ivec user(n, -1);
    owned_gids_.assign((size_t)n1_ + n2_, 0);
    const int64_t nowned = n1_ + n2_;
    const int64_t found = parallel_compact(nrows_, [&](int64_t i) { return intidx_[i] >= 0 || (pos2_[i] >= 0 && pos2_[i] < n2_); },
                                           [&](int64_t i, int64_t k) { user[i] = (int32_t)k; if (k < nowned) owned_gids_[k] = gids_[i]; });
    HYMLS_CHECK(found == nowned, -3, "partition does not cover the map exactly once");
    if (!dist) HYMLS_CHECK(n1_ + n2_ == n, -3, "partition does not cover the map exactly once");
    parallel_for(nowned, [&](int64_t t) { in_perm_[t] = user[in_perm_[t]]; }, 1 << 16);
    global_n_ = comm_->allsum(n1_ + n2_);
    global_n2_ = comm_->allsum(n2_);
  }
  fine("owned rows / user order");
  // A12 (interior rows x separators incl. ghosts) / A21 (owned separator rows x interiors incl. ghosts)
  ivec node_sd;     // local node -> halo subdomain holding it as interior
  if (dist) {
    node_sd.assign(n, -1);
    for (int s : halo_sds_) for (int32_t x : hm_.sd[s].interior) if (g2l_[x] >= 0) node_sd[g2l_[x]] = s;
  }
  ivec row_of_internal(n1_);
  parallel_for(n, [&](int64_t i) { if (intidx_[i] >= 0) row_of_internal[intidx_[i]] = (int32_t)i; }, 1 << 16);
  a12_row_.assign(n1_ + 1, 0); a21_row_.assign(n2_ + 1, 0);
  fine("row_of_internal");
  // ghost interior columns of A21 are numbered first (sequential over the few boundary rows)
  std::vector<std::vector<int64_t>> want_int(comm_->size);
  std::vector<ivec> dst_int(comm_->size);
  ivec ghost_idx(dist ? n : 0, -1);
  ngi_ = 0;
  if (dist)
    for (int k = 0; k < n2_; k++) {
      const int r = sep_row_[k];
      for (int e = K_.rowptr[r]; e < K_.rowptr[r + 1]; e++) {
        const int c = K_.col[e];
        if (intidx_[c] < 0 && pos2_[c] < 0 && node_sd[c] >= 0 && ghost_idx[c] < 0) {
          ghost_idx[c] = n1_ + ngi_++;
          const int q = sd_rank_[node_sd[c]];
          want_int[q].push_back(gids_[c]);
          dst_int[q].push_back(ghost_idx[c]);
        }
      }
    }
  // The two blocks are cut out of the level matrix on the device (its pattern is there already): count per row, prefix sum
  // on the host threads, fill.  The tables stay on the device; the host keeps copies (border set-up, sizes).
  //   A12: interior rows, columns = separators incl. ghosts (target pos2)
  //   A21: owned separator rows, columns = interiors (target intidx), sharded: + the neighbours' interior layer (ghost_idx)
  d_krow_ = dev::upload(K_.rowptr); d_kcol_ = dev::upload(K_.col);
  {
    int32_t* d_pos2 = dev::upload(pos2_);
    int32_t* d_int = dev::upload(intidx_);
    int32_t* d_ghost = dist ? dev::upload(ghost_idx) : nullptr;
    int32_t* d_rows1 = dev::upload(row_of_internal);
    ivec sep_rows(sep_row_.begin(), sep_row_.begin() + n2_);
    int32_t* d_rows2 = dev::upload(sep_rows);
    auto cut = [&](int64_t nr, const int32_t* d_rows, const int32_t* ta, const int32_t* tb, const int32_t* excl, ivec& row, cvec& col,
                   cvec& src, int32_t*& d_row, int32_t*& d_col, int32_t*& d_src) {
      d_row = (int32_t*)dev::alloc((size_t)(nr + 1) * sizeof(int32_t));
      dev::zero(d_row, sizeof(int32_t));
      dev::offdiag_count(nr, d_rows, d_krow_, d_kcol_, ta, tb, excl, d_row);
      dev::d2h(row.data(), d_row, (size_t)(nr + 1) * sizeof(int32_t));
      parallel_inclusive_scan(row.data(), nr + 1);
      dev::h2d(d_row, row.data(), (size_t)(nr + 1) * sizeof(int32_t));
      const size_t nz = (size_t)row[nr];
      d_col = (int32_t*)dev::alloc(std::max<size_t>(1, nz) * sizeof(int32_t));
      d_src = (int32_t*)dev::alloc(std::max<size_t>(1, nz) * sizeof(int32_t));
      dev::offdiag_fill(nr, d_rows, d_krow_, d_kcol_, ta, tb, excl, d_row, d_col, d_src);
      col.resize(nz); src.resize(nz);
      if (nz) { dev::d2h(col.data(), d_col, nz * sizeof(int32_t)); dev::d2h(src.data(), d_src, nz * sizeof(int32_t)); }
    };
    cut(n1_, d_rows1, d_pos2, nullptr, nullptr, a12_row_, a12_col_, a12_src_, d_a12_row_, d_a12_col_, d_a12_src_);
    fine("A12 count / prefix / fill (device)");
    cut(n2_, d_rows2, d_int, d_ghost, d_pos2, a21_row_, a21_col_, a21_src_, d_a21_row_, d_a21_col_, d_a21_src_);
    dev::sync();
    dev::free(d_pos2); dev::free(d_int); dev::free(d_ghost); dev::free(d_rows1); dev::free(d_rows2);
  }
  if (dist) {
    // x1 of the neighbours' interiors next to separators owned here; x2 of the separators owned elsewhere
    xch_int_.build(*comm_, want_int, dst_int, [&](int64_t g) { const int l = g2l_[g]; return l >= 0 ? intidx_[l] : -1; });
    xch_sep_.build(*comm_, want_sep, dst_sep, [&](int64_t g) { const int l = g2l_[g]; return (l >= 0 && pos2_[l] < n2_) ? pos2_[l] : -1; });
  }
  fine("A21 count / prefix / fill + halo plans");
  lap("A12/A21 + halo plans");
  // the large factor arrays are requested now, on helper threads: the allocations pass while the host builds the tables of
  // the Schur set-up (not earlier: the device calls of the A12 / A21 step above would queue behind them)
  for (auto& cp : cls_) {
    const size_t fb = BatchedLU::factor_bytes((int64_t)cp->lu.members.size(), cp->lu.plan.factor_size);
    if (fb >= ((size_t)1 << 30) && !cp->lu.pre_factor) cp->lu.pre_factor.reset(new AsyncAlloc(fb));
  }
  // (the pattern of the level matrix is on the device already: the classes build their entry source lists from it there)
  for (auto& cp : cls_) { cp->lu.d_krow = d_krow_; cp->lu.d_kcol = d_kcol_; }
  build_schur_setup();
  lap("Schur setup + uploads");
  // device residents
  d_kval_ = (double*)dev::alloc(std::max<size_t>(1, K_.val.size()) * sizeof(double));
  d_inperm_ = dev::upload(in_perm_);
  d_z_ = (double*)dev::alloc((size_t)std::max(n1_ + ngi_ + n2_, 1) * sizeof(double));
  d_t1_ = (double*)dev::alloc((size_t)std::max(n1_, 1) * sizeof(double));
  d_t2_ = (double*)dev::alloc((size_t)std::max(n2_ + ngs_, 1) * sizeof(double));
  d_y2_ = (double*)dev::alloc((size_t)std::max(n2_, 1) * sizeof(double));
  d_yb_ = (double*)dev::alloc((size_t)std::max(n2_, 1) * sizeof(double));
  d_a12_val_ = (double*)dev::alloc(std::max<size_t>(1, a12_col_.size()) * sizeof(double));
  d_a21_val_ = (double*)dev::alloc(std::max<size_t>(1, a21_col_.size()) * sizeof(double));
  d_flag_ = (int32_t*)dev::alloc(sizeof(int32_t));
  initialized_ = true;
}

void LevelSolver::build_classes() {
  const bool verbose_bc = std::getenv("HYMLS_MI_VERBOSE") != nullptr && (level_ == 0 || std::atoi(std::getenv("HYMLS_MI_VERBOSE")) >= 2);
  double t_bc = wall();
  auto lap_bc = [&](const char* what) {
    if (!verbose_bc) return;
    std::fprintf(stderr, "[hymls_mi] rank %d level %d:   classes / %-22s %.2f s\n", comm_->rank, level_, what, wall() - t_bc);
    t_bc = wall();
  };
  const int n = K_.n;
  const int nsd = (int)hm_.sd.size();
  const int nsep = n2_ + ngs_;
  auto sidx = [&](int32_t gid) { const int l = g2l_[gid]; return l >= 0 ? pos2_[l] : -1; };
  // subdomains (of any rank) listing each separator node (for the A22 multiplicities)
  ivec cnt(nsep + 1, 0);
  for (auto& S : hm_.sd) for (auto& g : S.groups) for (int32_t x : g.nodes) { const int k = sidx(x); if (k >= 0) cnt[k + 1]++; }
  for (int i = 0; i < nsep; i++) cnt[i + 1] += cnt[i];
  ivec sdl(cnt[nsep]), fill(cnt.begin(), cnt.end() - 1);
  for (int s = 0; s < nsd; s++)
    for (auto& g : hm_.sd[s].groups) for (int32_t x : g.nodes) { const int k = sidx(x); if (k >= 0) sdl[fill[k]++] = s; }
  auto common = [&](int a, int b) {   // |L(a) and L(b)|: both lists ascend (filled in subdomain order)
    int c = 0, i = cnt[a], j = cnt[b];
    const int ie = cnt[a + 1], je = cnt[b + 1];
    while (i < ie && j < je) {
      const int x = sdl[i], y = sdl[j];
      c += x == y; i += x <= y; j += y <= x;
    }
    return c;
  };
  FineLap fine;
  sep_sd_ptr_ = cnt; sep_sd_ = sdl;
  fine("(classes) subdomains listing each separator node");
  sd_center_.assign(3 * (size_t)nsd, 0);
  for (int s = 0; s < nsd; s++) {
    int64_t acc[3] = {0, 0, 0}, m = 0;
    int32_t cc[3];
    for (auto& g : hm_.sd[s].groups) { gid_coord(p_, g.nodes[0], cc); for (int a = 0; a < 3; a++) acc[a] += cc[a]; m++; }
    for (int32_t x : hm_.sd[s].interior) { gid_coord(p_, x, cc); for (int a = 0; a < 3; a++) acc[a] += cc[a]; m++; if (m > 64) break; }
    for (int a = 0; a < 3; a++) sd_center_[3 * (size_t)s + a] = m ? (int32_t)(acc[a] / m) : 0;
  }
  sd_xoff_.assign(nsd, 0); sd_cls_.assign(nsd, -1); sd_bidx_.assign(nsd, -1);
  fine("(classes) subdomain centres");
  (void)n;
  // ---- pass 1: extended local pattern of every subdomain (in parallel, chunk by chunk), then classification
  struct SdPat { LocalPattern lp; ivec src, ext, mult, lgptr, key_extra; uint64_t hash = 0; std::string err; };
  // rows of the level matrix with strictly ascending columns: the device finds the entries of every member itself
  // (dev::member_sources) from the members' node lists; otherwise the lists are built here, entry by entry
  // (one pass over the matrix answers the question and leaves the row hashes of the class signatures below)
  struct SigMix {
    uint64_t a = 0x243F6A8885A308D3ULL, b = 0x13198A2E03707344ULL;
    void add(uint64_t w) {
      a = (a ^ w) * 0x9E3779B97F4A7C15ULL; a ^= a >> 29;
      b ^= w * 0xC2B2AE3D27D4EB4FULL; b = (b << 31) | (b >> 33); b *= 0x165667B19E3779F9ULL;
    }
  };
  const bool want_fast = !std::getenv("HYMLS_MI_NO_FAST_CLASSES") && !std::getenv("HYMLS_MI_HOST_SOURCE_LISTS");
  std::vector<uint64_t> rowh;
  if (want_fast) rowh.resize((size_t)K_.n);
  std::atomic<int> unsorted{0};
  parallel_for(K_.n, [&](int64_t r) {
    SigMix m;
    m.add((uint64_t)(K_.rowptr[r + 1] - K_.rowptr[r]));
    uint64_t dz = 1;    // no non-zero diagonal entry
    bool bad = false;
    for (int e = K_.rowptr[r]; e < K_.rowptr[r + 1]; e++) {
      if (e > K_.rowptr[r] && K_.col[e] <= K_.col[e - 1]) bad = true;
      if (want_fast) {
        m.add((uint64_t)(uint32_t)(K_.col[e] - (int32_t)r));
        if (K_.col[e] == r && K_.val[e] != 0.0) dz = 0;
      }
    }
    if (bad) unsorted = 1;
    if (want_fast) { m.add(dz); rowh[(size_t)r] = m.a ^ (m.b << 1); }
  }, 1 << 12);
  const bool device_src = unsorted == 0 && !std::getenv("HYMLS_MI_HOST_SOURCE_LISTS");
  static std::atomic<long long> tprof[6];
  auto tnow = []() { return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  // coordinates of the interior nodes relative to the subdomain (parity of the corner kept: staggered grids)
  auto rel_coords = [&](int s, ivec& coord) {
    const Subdomain& S = hm_.sd[s];
    const int nI = (int)S.interior.size();
    coord.resize(3 * (size_t)nI);
    int32_t mn[3] = {INT32_MAX, INT32_MAX, INT32_MAX};
    int spos[3] = {0, 0, 0};
    const bool wraps = p_.perio[0] || p_.perio[1] || p_.perio[2];
    if (wraps) sd_position(p_, s, spos[0], spos[1], spos[2]);
    const int nn[3] = {p_.nx, p_.ny, p_.nz}, ss[3] = {p_.sx, p_.sy, p_.sz};
    for (int i = 0; i < nI; i++) {
      int32_t* cc = &coord[3 * (size_t)i];
      gid_coord(p_, S.interior[i], cc);
      // a subdomain that reaches across a periodic boundary: the image next to the subdomain's reference corner (every
      // node lies within [pos - s - 1, pos + s + 1]), so that relative coordinates are those of an inner subdomain
      if (wraps) for (int a = 0; a < 3; a++) if (p_.perio[a]) {
        const int lo = 2 * (spos[a] - ss[a] - 1), per2 = 2 * nn[a];
        cc[a] = lo + (((cc[a] - lo) % per2) + per2) % per2;
      }
      for (int a = 0; a < 3; a++) mn[a] = std::min(mn[a], cc[a]);
    }
    for (int i = 0; i < nI; i++) for (int a = 0; a < 3; a++) coord[3 * (size_t)i + a] -= mn[a] & ~1;
  };
  // ---- Shortcut for the thousands of subdomains that are translates of one another (69 696 subdomains, 54 classes at 256^3):
  // a 128-bit signature of everything the extended local pattern is a function of -- the node list relative to its first
  // node, a hash of every row of the level matrix relative to its row number (columns - row, diagonal zero or not), the
  // subdomains listing every separator node relative to this subdomain (the A22 multiplicities are intersections of those
  // lists), group sizes / types / links, relative coordinates.  A subdomain whose signature is known takes the class without
  // its pattern being built (the scan of its 1 335 matrix rows with a hash probe per entry); every class still has a fully
  // built, fully compared representative.  HYMLS_MI_VERIFY_CLASSES=1 builds every pattern nevertheless and checks that the
  // shortcut agrees (tests); HYMLS_MI_NO_FAST_CLASSES switches it off.
  struct Sig { uint64_t a, b; bool operator==(const Sig& o) const { return a == o.a && b == o.b; } };
  struct SigHash { size_t operator()(const Sig& q) const { return (size_t)(q.a ^ (q.b * 0x9E3779B97F4A7C15ULL)); } };
  const bool fast_classes = device_src && want_fast;
  const bool verify_classes = std::getenv("HYMLS_MI_VERIFY_CLASSES") != nullptr;
  // relative coordinates follow from the relative node numbers when the grid is much wider than a subdomain (a difference of
  // two gids then has one decomposition into (dx, dy, dz, dvar)); on small grids they are hashed explicitly
  const bool coords_implied = p_.nx >= 4 * p_.sx + 8 && p_.ny >= 4 * p_.sy + 8 && (p_.nz == 1 || p_.nz >= 4 * p_.sz + 8) &&
                              !(p_.perio[0] || p_.perio[1] || p_.perio[2]);
  std::unordered_map<Sig, int, SigHash> sigtable;
  std::atomic<long long> n_fast{0};
  auto signature = [&](int s, ivec& ext, Sig& sig) -> bool {
    const Subdomain& S = hm_.sd[s];
    const int nI = (int)S.interior.size();
    ext.clear();
    ext.reserve(nI + S.num_sep());
    // (the look-ups below run at memory latency: the lines a few nodes ahead are requested early)
    for (int i = 0; i < nI; i++) {
      if (i + 16 < nI) __builtin_prefetch(&g2l_[S.interior[i + 16]]);
      ext.push_back(g2l_[S.interior[i]]);
    }
    SigMix m;
    m.add((uint64_t)nI); m.add((uint64_t)S.num_sep()); m.add((uint64_t)S.groups.size());
    for (auto& g : S.groups) {
      for (int32_t x : g.nodes) ext.push_back(g2l_[x]);
      m.add(((uint64_t)(uint32_t)g.type << 32) | (uint64_t)g.nodes.size());
    }
    for (auto& L : S.linked) { m.add(0xFFFFFFFFFFFFFFF9ULL); for (int gi : L) m.add((uint64_t)gi); }
    if (ext.empty()) return false;
    const int ne = (int)ext.size();
    for (int i = 0; i < ne; i++) if (!(ext[i] >= 0 && ext[i] < nrows_)) return false;     // (the full build reports what is wrong)
    const int32_t e0 = ext[0];
    for (int i = 0; i < ne; i++) {
      if (i + 16 < ne) { __builtin_prefetch(&rowh[(size_t)ext[i + 16]]); if (i + 16 >= nI) __builtin_prefetch(&pos2_[ext[i + 16]]); }
      m.add((uint64_t)(uint32_t)(ext[i] - e0)); m.add(rowh[(size_t)ext[i]]);
    }
    for (int i = nI; i < ne; i++) {
      const int k = pos2_[ext[i]];
      if (k < 0) return false;
      m.add((uint64_t)(cnt[k + 1] - cnt[k]));
      for (int t = cnt[k]; t < cnt[k + 1]; t++) m.add((uint64_t)(uint32_t)(sdl[t] - s));
    }
    if (coords_implied && nI > 0) {
      int32_t cc[3];
      gid_coord(p_, S.interior[0], cc);      // parity of the corner (staggered grid)
      m.add((uint64_t)((cc[0] & 1) | ((cc[1] & 1) << 1) | ((cc[2] & 1) << 2)));
      m.add((uint64_t)(uint32_t)(S.interior[0] % p_.dof));
    } else {
      ivec coord;
      rel_coords(s, coord);
      for (int32_t c : coord) m.add((uint64_t)(uint32_t)c);
    }
    sig = Sig{m.a, m.b};
    return true;
  };
  auto build_pattern = [&](int s, SdPat& out) {
    long long tq = tnow();
    auto lapq = [&](int k) { const long long t = tnow(); tprof[k] += t - tq; tq = t; };
    const Subdomain& S = hm_.sd[s];
    LocalPattern& lp = out.lp;
    lp.nI = (int32_t)S.interior.size();
    lp.nS = S.num_sep();
    ivec ext_rows;  // local nodes of the extended local numbering
    ext_rows.reserve(lp.nI + lp.nS);
    for (int32_t g : S.interior) ext_rows.push_back(g2l_[g]);
    out.lgptr.assign(1, 0);
    for (auto& g : S.groups) {
      for (int32_t x : g.nodes) ext_rows.push_back(g2l_[x]);
      out.lgptr.push_back((int32_t)ext_rows.size() - lp.nI);
      out.key_extra.push_back(g.type);
    }
    for (auto& L : S.linked) { out.key_extra.push_back(-7); out.key_extra.insert(out.key_extra.end(), L.begin(), L.end()); }
    const int ne = lp.nI + lp.nS;
    // node -> position in this subdomain: small open-addressing table (thread-private)
    int cap = 16;
    while (cap < 2 * ne + 2) cap <<= 1;
    ivec hk(cap, -1), hv(cap, -1);
    auto find = [&](int32_t node) {
      uint32_t h = ((uint32_t)node * 2654435761u) & (cap - 1);
      while (hk[h] != -1) { if (hk[h] == node) return hv[h]; h = (h + 1) & (cap - 1); }
      return -1;
    };
    for (int i = 0; i < ne; i++) {
      if (!(ext_rows[i] >= 0 && ext_rows[i] < nrows_) || find(ext_rows[i]) >= 0) { out.err = "node listed twice in a subdomain or without a row"; return; }
      uint32_t h = ((uint32_t)ext_rows[i] * 2654435761u) & (cap - 1);
      while (hk[h] != -1) h = (h + 1) & (cap - 1);
      hk[h] = ext_rows[i]; hv[h] = i;
    }
    lapq(0);
    lp.rowptr.assign(ne + 1, 0);
    lp.zero_diag.assign(lp.nI, 1);
    for (int i = 0; i < ne; i++) {
      const int r = ext_rows[i];
      for (int e = K_.rowptr[r]; e < K_.rowptr[r + 1]; e++) {
        const int c = K_.col[e], lc = find(c);
        if (lc < 0) {
          if (i < lp.nI && pos2_[c] < 0) { out.err = "partitioning does not decouple the interiors of different subdomains"; return; }
          continue;
        }
        int m = 1;
        if (i >= lp.nI && lc >= lp.nI) {
          m = common(pos2_[r], pos2_[c]);
          if (m <= 0) { out.err = "separator coupling not inside any subdomain"; return; }
        }
        lp.col.push_back(lc); out.mult.push_back(m);
        if (!device_src) out.src.push_back(e);
        lp.weight.push_back(1.0 / m);
        if (lc == i && i < lp.nI && K_.val[e] != 0.0) lp.zero_diag[i] = 0;
      }
      lp.rowptr[i + 1] = (int32_t)lp.col.size();
    }
    lapq(1);
    rel_coords(s, lp.coord);
    Hasher H;
    H.add(&lp.nI, 4); H.add(&lp.nS, 4);
    H.addv(lp.rowptr); H.addv(lp.col); H.addv(lp.zero_diag); H.addv(lp.coord); H.addv(out.mult);
    H.addv(out.lgptr); H.addv(out.key_extra);
    out.hash = H.h;
    if (device_src) out.ext.swap(ext_rows);
    lapq(2);
  };
  std::unordered_map<uint64_t, std::vector<int>> table;
  std::vector<ivec> sd_src(nsd);
  const size_t first_new = cls_.size();
  {
    // Every host thread takes the next subdomain, builds its pattern and looks its class up through the 64-bit hash of
    // everything that defines it (and a size check) under a lock; the full comparison with the class representative runs
    // outside the lock.  Classes are renumbered afterwards in the order of their first member, members are listed in
    // subdomain order: the result does not depend on which thread came first.  (Before: chunks of 512 patterns with a
    // sequential classification step between them -- 2.6 s of wall time for 16.5 thread-seconds at 256^3.)
    const int64_t nmine = (int64_t)my_sds_.size();
    std::atomic<int64_t> next{0};
    std::atomic<int> mismatch{0};
    std::mutex mu;
    int64_t err_k = -1;
    std::string err_msg;
    parallel_for(64, [&](int64_t) {
      for (int64_t k = next.fetch_add(1); k < nmine; k = next.fetch_add(1)) {
        const int s = my_sds_[k];
        Sig sig{0, 0};
        ivec ext_fast;
        int cid_fast = -1;
        const bool have_sig = fast_classes && signature(s, ext_fast, sig);
        if (have_sig) {
          std::lock_guard<std::mutex> lk(mu);
          auto it = sigtable.find(sig);
          if (it != sigtable.end()) cid_fast = it->second;
        }
        if (cid_fast >= 0 && !verify_classes) {
          sd_cls_[s] = cid_fast;
          sd_src[s].swap(ext_fast);
          n_fast++;
          continue;
        }
        SdPat Pt;
        build_pattern(s, Pt);
        if (!Pt.err.empty()) {
          std::lock_guard<std::mutex> lk(mu);
          if (err_k < 0 || k < err_k) { err_k = k; err_msg = Pt.err; }
          continue;
        }
        const Subdomain& S = hm_.sd[s];
        LocalPattern& lp = Pt.lp;
        int cid = -1;
        bool is_new = false;
        const Cls* rep = nullptr;
        {
          std::lock_guard<std::mutex> lk(mu);
          for (int c : table[Pt.hash]) {
            const Cls& C = *cls_[c];
            if (C.pat.nI == lp.nI && C.pat.nS == lp.nS && C.pat.col.size() == lp.col.size() && C.lgptr.size() == Pt.lgptr.size()) { cid = c; break; }
          }
          if (cid < 0) {
            cid = (int)cls_.size();
            cls_.emplace_back(new Cls());
            Cls& C = *cls_.back();
            C.pat = std::move(lp);
            C.mult = Pt.mult; C.lgptr = Pt.lgptr; C.key_extra = Pt.key_extra;
            C.llinked = S.linked;
            C.ngl = (int32_t)S.groups.size();
            table[Pt.hash].push_back(cid);
            is_new = true;
          }
          rep = cls_[cid].get();     // (the vector may grow under another thread's hands: the object itself stays put)
          if (have_sig) {
            auto ins = sigtable.emplace(sig, cid);
            if (ins.first->second != cid || (cid_fast >= 0 && cid_fast != cid)) mismatch = 2;   // (verification mode, or two builders of one signature)
          }
        }
        if (!is_new) {
          const Cls& C = *rep;
          if (!(C.pat.rowptr == lp.rowptr && C.pat.col == lp.col && C.pat.zero_diag == lp.zero_diag && C.pat.coord == lp.coord &&
                C.mult == Pt.mult && C.lgptr == Pt.lgptr && C.key_extra == Pt.key_extra)) mismatch = 1;
        }
        sd_cls_[s] = cid;
        // the class's entry numbering (plan.ent_id refers to the extended CSR) mapped onto this member: kept per
        // subdomain and copied into the class arrays below, in parallel and without reallocation
        sd_src[s].swap(device_src ? Pt.ext : Pt.src);
      }
    }, 1);
    HYMLS_CHECK(err_k < 0, err_msg.find("decouple") != std::string::npos ? -2 : -3, err_msg);
    HYMLS_CHECK(mismatch != 2, -3, "pattern class shortcut: one signature, two different extended patterns");
    HYMLS_CHECK(mismatch == 0, -3, "two different subdomain patterns share one 64-bit hash");
    if (std::getenv("HYMLS_MI_PATTERN_PROF"))
      std::fprintf(stderr, "[hymls_mi] pattern classes: %lld of %lld subdomains classified by signature\n", (long long)n_fast, (long long)nmine);
    // new classes in the order of their first member; members in subdomain order
    const int nnew = (int)(cls_.size() - first_new);
    ivec new_id(nnew, -1);
    int seen = 0;
    for (int s : my_sds_) { int& id = new_id[sd_cls_[s] - (int)first_new]; if (id < 0) id = seen++; }
    HYMLS_CHECK(seen == nnew, -3, "pattern class without members");
    std::vector<std::unique_ptr<Cls>> ordered((size_t)nnew);
    for (int c = 0; c < nnew; c++) ordered[new_id[c]] = std::move(cls_[first_new + c]);
    for (int c = 0; c < nnew; c++) cls_[first_new + c] = std::move(ordered[c]);
    for (int s : my_sds_) {
      sd_cls_[s] = (int)first_new + new_id[sd_cls_[s] - (int)first_new];
      Cls& C = *cls_[sd_cls_[s]];
      sd_bidx_[s] = (int32_t)C.lu.members.size();
      C.lu.members.push_back(s);
    }
  }
  if (std::getenv("HYMLS_MI_PATTERN_PROF")) std::fprintf(stderr, "[hymls_mi] build_pattern thread-seconds: ext rows + table %.2f | K scan + find + mult %.2f | coords + hash %.2f\n", tprof[0] / 1e9, tprof[1] / 1e9, tprof[2] / 1e9);
  lap_bc("patterns + class lookup");
  for (size_t c = first_new; c < cls_.size(); c++) {
    Cls& C = *cls_[c];
    const size_t ne = device_src ? (size_t)(C.pat.nI + C.pat.nS) : C.pat.col.size();
    rawvec<int32_t>& dst = device_src ? C.lu.h_ext : C.lu.h_src;
    dst.resize(ne * C.lu.members.size());
    parallel_for((int64_t)C.lu.members.size(), [&](int64_t b) {
      ivec& v = sd_src[C.lu.members[b]];
      std::copy(v.begin(), v.end(), dst.begin() + (size_t)b * ne);
      ivec().swap(v);
    }, 64);
    if (device_src) {
      C.lu.n_ext = (int32_t)ne;
      C.lu.h_ent_col.assign(C.pat.col.begin(), C.pat.col.end());
      C.lu.h_ent_row.resize(C.pat.col.size());
      for (int i = 0; i < C.pat.nI + C.pat.nS; i++) for (int q = C.pat.rowptr[i]; q < C.pat.rowptr[i + 1]; q++) C.lu.h_ent_row[q] = i;
    }
  }
  lap_bc("entry source lists");
  // ---- pass 2: symbolic analysis of every class (independent: in parallel)
  // (the classes of a coarser level differ in size by an order of magnitude: the threads take them from a counter, largest first)
  const int64_t ncls_new = (int64_t)(cls_.size() - first_new);
  std::vector<int64_t> by_size((size_t)ncls_new);
  std::iota(by_size.begin(), by_size.end(), 0);
  std::stable_sort(by_size.begin(), by_size.end(), [&](int64_t a, int64_t b) { return cls_[first_new + a]->pat.col.size() > cls_[first_new + b]->pat.col.size(); });
  std::atomic<int64_t> next_cls{0};
  parallel_for(std::min<int64_t>(ncls_new, 64), [&](int64_t) {
    for (int64_t q = next_cls.fetch_add(1); q < ncls_new; q = next_cls.fetch_add(1)) {
    const int64_t k = by_size[(size_t)q];
    Cls& C = *cls_[first_new + k];
    // leaf size of the nested dissection: the finest level (LDS-fused solve, one workgroup walks the whole tree) is fastest at
    // 24 -- smaller leaves store up to 26 % fewer panel entries but add tree levels, each a barrier-bound step of the
    // workgroup; larger ones only add bytes (profiles/r03_g_leaf_size_sweep.txt).  The large subdomains of the coarser
    // levels are solved with one launch per tree level, so fewer levels pay there.
    static const int leaf0 = std::getenv("HYMLS_MI_LEAF_SIZE") ? std::atoi(std::getenv("HYMLS_MI_LEAF_SIZE")) : LEAF_SIZE;
    static const int leaf1 = std::getenv("HYMLS_MI_LEAF_SIZE_UPPER") ? std::atoi(std::getenv("HYMLS_MI_LEAF_SIZE_UPPER")) : LEAF_SIZE_UPPER;
    C.lu.plan = analyse_class(C.pat, level_ == 0 ? leaf0 : leaf1, MAX_WIDTH);
    }
  }, 1);
  lap_bc("symbolic analysis");
  // ---- pass 3: interior numbering in elimination order, subdomain by subdomain
  n1_ = 0;
  for (int s : my_sds_) {      // offsets first (sequential, cheap), then the entries of every subdomain in parallel
    Cls& C = *cls_[sd_cls_[s]];
    sd_xoff_[s] = n1_;
    C.lu.h_xoff.push_back(n1_);
    n1_ += C.pat.nI;
  }
  in_perm_.assign((size_t)n1_ + n2_, 0);
  parallel_for((int64_t)my_sds_.size(), [&](int64_t k) {
    const int s = my_sds_[k];
    const Subdomain& S = hm_.sd[s];
    const Cls& C = *cls_[sd_cls_[s]];
    const int32_t off = sd_xoff_[s];
    for (int t = 0; t < C.pat.nI; t++) {
      const int r = g2l_[S.interior[C.lu.plan.perm[t]]];
      in_perm_[off + t] = r;
      intidx_[r] = off + t;
    }
  }, 16);
  parallel_for(n2_, [&](int64_t k) { in_perm_[(size_t)n1_ + k] = sep_row_[k]; }, 1 << 16);
  lap_bc("interior numbering");
  if (std::getenv("HYMLS_MI_VERBOSE")) {
    std::fprintf(stderr, "[hymls_mi] rank %d level %d: local nodes %d subdomains %zu (+%zu halo) classes %zu n1 %d n2 %d ghost sep %d\n",
                 comm_->rank, level_, n, my_sds_.size(), halo_sds_.size(), cls_.size(), n1_, n2_, ngs_);
    size_t shown = 0;
    for (auto& c : cls_) if (shown++ < 4 || c->lu.members.size() > 50) print_plan_stats(c->lu.plan, "  class", (int)c->lu.members.size());
  }
}

// what is kept of one subdomain's (transformed) separator block, as laid out in its extraction
// record: the V-sum x V-sum part (ngl x ngl, column-major) followed by one dense block per linked
// set with at least one non-V-sum row.  Depends on the group structure only, so a rank can lay out
// the records it receives from its neighbours.
LevelSolver::ExtLayout LevelSolver::ext_layout(const Subdomain& S) const {
  ExtLayout L;
  if (direct_schur_) { const int64_t nS = S.num_sep(); L.size = nS * nS; return L; }
  L.ngl = (int32_t)S.groups.size();
  int64_t off = (int64_t)L.ngl * L.ngl;
  for (auto& ls : S.linked) {
    int32_t len = 0;
    for (int gi : ls) len += (int32_t)S.groups[gi].nodes.size() - 1;
    L.blk_len.push_back(len);
    L.blk_off.push_back(len ? off : -1);
    off += (int64_t)len * len;
  }
  L.size = off;
  return L;
}

void LevelSolver::build_schur_setup() {
  const bool dist = comm_->distributed();
  const int ng_owned = (int)gptr_.size() - 1;
  const bool verbose = std::getenv("HYMLS_MI_VERBOSE") != nullptr;
  double t0 = wall();
  auto lap = [&](const char* what) {
    if (verbose) std::fprintf(stderr, "[hymls_mi] rank %d level %d:   schur setup / %-18s %.2f s\n", comm_->rank, level_, what, wall() - t0);
    t0 = wall();
  };
  // ---- per class: what to keep of the (transformed) separator block
  ext_total_ = 0;
  for (auto& cp : cls_) {
    Cls& C = *cp;
    const int nS = C.pat.nS;
    C.pick.clear(); C.blk_off.clear(); C.blk_len.clear();
    if (direct_schur_) {
      C.pick.resize((size_t)nS * nS);
      std::iota(C.pick.begin(), C.pick.end(), 0);
    } else {
      for (int b = 0; b < C.ngl; b++) for (int a = 0; a < C.ngl; a++) C.pick.push_back(C.lgptr[a] + nS * C.lgptr[b]);
      for (auto& L : C.llinked) {
        ivec locs;
        for (int gi : L) for (int t = C.lgptr[gi] + 1; t < C.lgptr[gi + 1]; t++) locs.push_back(t);
        C.blk_len.push_back((int32_t)locs.size());
        C.blk_off.push_back(locs.empty() ? -1 : (int64_t)C.pick.size());
        for (int b : locs) for (int a : locs) C.pick.push_back(a + nS * b);
      }
    }
    C.ext_size = (int64_t)C.pick.size();
    C.ext_base = ext_total_;
    ext_total_ += C.ext_size * (int64_t)C.lu.members.size();
    // test vector at the separators of every member
    C.tvloc.assign((size_t)nS * C.lu.members.size(), 1.0);
    parallel_for((int64_t)C.lu.members.size(), [&](int64_t b) {
      const Subdomain& S = hm_.sd[C.lu.members[b]];
      size_t t = 0;
      for (auto& g : S.groups) for (int32_t x : g.nodes) C.tvloc[(size_t)b * nS + t++] = tv_[g2l_[x]];
    }, 64);
  }
  ext_recv_base_ = ext_total_;
  // owned group lookup by first gid
  // (a table over the gids: a hash map of the 1.7 M groups of a 256^3 run took 0.3 s to fill and is probed 5 M times)
  ivec gidx_of_first(ngid_, -1);
  {
    int g = 0;
    for (int s : my_sds_) for (int gi : hm_.sd[s].owned) gidx_of_first[hm_.sd[s].groups[gi].nodes[0]] = g++;
  }
  auto owned_sep = [&](int32_t gid) { const int l = g2l_[gid]; return (l >= 0 && pos2_[l] >= 0 && pos2_[l] < n2_) ? pos2_[l] : -1; };
  lap("kept entries");
  FineLap fine;
  // ---- records of the neighbours' subdomains that touch separators owned here
  std::vector<std::pair<int, int64_t>> contributors;   // (subdomain, base of its record in the extraction buffer)
  for (int s : my_sds_) contributors.emplace_back(s, cls_[sd_cls_[s]]->ext_base + (int64_t)sd_bidx_[s] * cls_[sd_cls_[s]]->ext_size);
  if (dist) {
    std::vector<std::vector<int64_t>> want(comm_->size);
    std::vector<std::vector<int64_t>> want_len(comm_->size);
    std::vector<std::pair<int, int64_t>> halo_list;
    for (int q = 0; q < comm_->size; q++) {
      for (int t : halo_sds_) {
        if (sd_rank_[t] != q) continue;
        const Subdomain& T = hm_.sd[t];
        bool mine = false;
        for (auto& g : T.groups) {
          if (direct_schur_) { for (int32_t x : g.nodes) if (owned_sep(x) >= 0) { mine = true; break; } }
          else mine = gidx_of_first[g.nodes[0]] >= 0;
          if (mine) break;
        }
        if (!mine) continue;
        const int64_t len = ext_layout(T).size;
        want[q].push_back(t);
        want_len[q].push_back(len);
        halo_list.emplace_back(t, ext_total_);
        ext_total_ += len;
      }
    }
    auto asked = comm_->exchange_lists(want);
    auto asked_len = comm_->exchange_lists(want_len);
    rec_send_.clear();
    rec_scnt_.assign(comm_->size, 0); rec_rcnt_.assign(comm_->size, 0);
    rec_nsend_ = 0; rec_nrecv_ = ext_total_ - ext_recv_base_;
    for (int q = 0; q < comm_->size; q++) {
      for (int64_t l : want_len[q]) rec_rcnt_[q] += l;
      for (size_t k = 0; k < asked[q].size(); k++) {
        const int s = (int)asked[q][k];
        HYMLS_CHECK(s >= 0 && s < (int)hm_.sd.size() && sd_rank_[s] == comm_->rank && sd_cls_[s] >= 0, -3,
                    "a neighbour asked for the Schur record of a subdomain that does not live here");
        const Cls& C = *cls_[sd_cls_[s]];
        HYMLS_CHECK(C.ext_size == asked_len[q][k], -3, "Schur record laid out differently on two ranks");
        rec_send_.push_back({C.ext_base + (int64_t)sd_bidx_[s] * C.ext_size, C.ext_size});
        rec_scnt_[q] += C.ext_size;
        rec_nsend_ += C.ext_size;
      }
    }
    rec_any_ = comm_->allsum(rec_nsend_) > 0;
    if (rec_any_) { comm_->send_arena(rec_nsend_); comm_->recv_arena(rec_nrecv_); }
    std::sort(halo_list.begin(), halo_list.end());
    contributors.insert(contributors.end(), halo_list.begin(), halo_list.end());
    std::sort(contributors.begin(), contributors.end());   // subdomain order, as on one rank
  }
  // ---- pull lists.  Entries (row, column gid, source position) are bucketed by row with a counting pass,
  // packed as (column gid << 33 | source) and sorted row by row in parallel.
  HYMLS_CHECK(ext_total_ < ((int64_t)1 << 33), -2, "extraction buffer too large for the packed pull keys");
  const size_t ext_bytes = (size_t)std::max<int64_t>(1, ext_total_) * sizeof(double);
  std::unique_ptr<AsyncAlloc> pre_ext(ext_bytes >= ((size_t)1 << 30) ? new AsyncAlloc(ext_bytes) : nullptr);   // (see AsyncAlloc)
  std::vector<int64_t> rcount;
  rawvec<uint64_t> keys;
  std::vector<int64_t> rfill;
  int pass = 0;
  auto emit = [&](int64_t row, int64_t colgid, int64_t src) {
    if (pass == 0) rcount[row + 1]++;
    else keys[rfill[row]++] = ((uint64_t)colgid << 33) | (uint64_t)src;
  };
  red_.n = direct_schur_ ? n2_ : ng_owned;
  rcount.assign((size_t)red_.n + 1, 0);
  std::map<int32_t, int> bc_of_size;
  std::vector<std::pair<int32_t, int32_t>> block_of_group;   // owned group that comes first in a linked set -> (block class, index)
  std::vector<std::vector<int64_t>> blk_cnt;                  // per block class: contributions per block, then their offsets
  struct BlkContrib { int cls, blk; int64_t src; };
  std::vector<int64_t> ct_off;
  ivec ct_vg;
  std::vector<std::vector<BlkContrib>> ct_blk;
  for (pass = 0; pass < 2; pass++) {
  fine(pass == 0 ? "(contributions) contributor list" : "(contributions) pass 0 incl. blocks");
  if (pass == 1) {
    for (int64_t r = 0; r < red_.n; r++) rcount[r + 1] += rcount[r];
    keys.resize((size_t)rcount[red_.n]);
    rfill.assign(rcount.begin(), rcount.end() - 1);
  }
  if (direct_schur_) {
    for (auto& ct : contributors) {
      const Subdomain& S = hm_.sd[ct.first];
      const int nS = S.num_sep();
      ivec gp, gg;
      for (auto& g : S.groups) for (int32_t x : g.nodes) { gp.push_back(owned_sep(x)); gg.push_back(x); }
      for (int b = 0; b < nS; b++)
        for (int a = 0; a < nS; a++)
          if (gp[a] >= 0) emit(gp[a], gg[b], ct.second + a + (int64_t)nS * b);
    }
  } else if (pass == 0) {
    // Householder rows (InitializeOT, reference src/HYMLS_SchurPreconditioner.cpp:384-467 +
    // Householder::Construct, src/HYMLS_Householder.cpp:128-163)
    otw_.assign(n2_, 0.0);
    vs_.resize(ng_owned);
    parallel_for(ng_owned, [&](int64_t g) {
      const int b = gptr_[g], e = gptr_[g + 1];
      vs_[g] = b;
      double* v = &otw_[b];             // (built in place: the entries of a group are this group's alone)
      for (int i = b; i < e; i++) v[i - b] = tv_[sep_row_[i]];
      const double sg = v[0] < 0 ? -1.0 : (v[0] > 0 ? 1.0 : 0.0);
      double nrm = 0;
      for (int i = 0; i < e - b; i++) { v[i] *= sg; nrm += v[i] * v[i]; }
      nrm = std::sqrt(nrm);
      v[0] += nrm;
      double nrm2 = 0;
      for (int i = 0; i < e - b; i++) nrm2 += v[i] * v[i];
      nrm2 = std::sqrt(nrm2);
      if (nrm2 < SMALL_ENTRY) { for (int i = 0; i < e - b; i++) v[i] = 0.0; return; }  // no row in T: the transform acts as -I (reference quirk)
      for (int i = 0; i < e - b; i++) v[i] = v[i] / nrm2;
    }, 4096);
    // dense blocks: one per owned linked set with at least one non-V-sum row.  Sizes per subdomain in parallel, class and
    // index of every block in subdomain order (sequential, integers only), node lists in parallel.
    const int64_t nmine = (int64_t)my_sds_.size();
    std::vector<ivec> sd_nb((size_t)nmine), sd_bc((size_t)nmine), sd_bi((size_t)nmine);
    parallel_for(nmine, [&](int64_t k) {
      const Subdomain& S = hm_.sd[my_sds_[k]];
      for (auto& L : S.owned_linked) {
        int32_t nb = 0;
        for (int gi : L) nb += (int32_t)S.groups[gi].nodes.size() - 1;
        sd_nb[k].push_back(nb);
      }
    }, 64);
    block_of_group.assign((size_t)ng_owned, {-1, -1});
    for (int64_t k = 0; k < nmine; k++) {
      const Subdomain& S = hm_.sd[my_sds_[k]];
      sd_bc[k].assign(sd_nb[k].size(), -1); sd_bi[k].assign(sd_nb[k].size(), -1);
      for (size_t li = 0; li < sd_nb[k].size(); li++) {
        const int nb = sd_nb[k][li];
        if (nb == 0) continue;
        if (!bc_of_size.count(nb)) { bc_of_size[nb] = (int)blocks_.size(); blocks_.emplace_back(); blocks_.back().nb = nb; }
        const int bc = bc_of_size[nb];
        sd_bc[k][li] = bc; sd_bi[k][li] = blocks_[bc].nblk;
        const int g = gidx_of_first[S.groups[S.owned_linked[li][0]].nodes[0]];
        HYMLS_CHECK(g >= 0, -3, "owned linked separator set does not start with an owned group");
        block_of_group[g] = {bc, blocks_[bc].nblk};
        blocks_[bc].nblk++;
      }
    }
    for (auto& B : blocks_) B.ids.assign((size_t)B.nb * B.nblk, 0);
    parallel_for(nmine, [&](int64_t k) {
      const Subdomain& S = hm_.sd[my_sds_[k]];
      for (size_t li = 0; li < sd_nb[k].size(); li++) {
        if (sd_bc[k][li] < 0) continue;
        BlockClass& B = blocks_[sd_bc[k][li]];
        int32_t* ids = B.ids.data() + (size_t)sd_bi[k][li] * B.nb;
        for (int gi : S.owned_linked[li]) for (size_t t = 1; t < S.groups[gi].nodes.size(); t++) *ids++ = pos2_[g2l_[S.groups[gi].nodes[t]]];
      }
    }, 64);
    blk_cnt.resize(blocks_.size());
    for (size_t c = 0; c < blocks_.size(); c++) blk_cnt[c].assign((size_t)blocks_[c].nblk + 1, 0);
  }
  if (!direct_schur_) {
    const int64_t nct = (int64_t)contributors.size();
    if (pass == 0) {
      // per contributor (in parallel): which owned group every one of its groups is, and its block contributions
      ct_off.assign(nct + 1, 0);
      for (int64_t c = 0; c < nct; c++) ct_off[c + 1] = ct_off[c] + (int64_t)hm_.sd[contributors[c].first].groups.size();
      ct_vg.assign((size_t)ct_off[nct], -1);
      ct_blk.assign((size_t)nct, {});
      parallel_for(nct, [&](int64_t c) {
        const Subdomain& S = hm_.sd[contributors[c].first];
        const ExtLayout L = ext_layout(S);
        const int64_t base = contributors[c].second;
        int32_t* vg = ct_vg.data() + ct_off[c];
        for (int a = 0; a < L.ngl; a++) {
          vg[a] = gidx_of_first[S.groups[a].nodes[0]];
          HYMLS_CHECK(dist || vg[a] >= 0, -3, "separator group without owner");
          if (vg[a] >= 0)
            HYMLS_CHECK(gptr_[vg[a] + 1] - gptr_[vg[a]] == (int)S.groups[a].nodes.size(), -3, "group differs between subdomains");
        }
        for (size_t li = 0; li < S.linked.size(); li++) {
          if (L.blk_off[li] < 0) continue;
          const int kg = gidx_of_first[S.groups[S.linked[li][0]].nodes[0]];
          const std::pair<int32_t, int32_t>* it = kg >= 0 && block_of_group[kg].first >= 0 ? &block_of_group[kg] : nullptr;
          HYMLS_CHECK(dist || it, -3, "linked separator set without owner");
          if (!it) continue;   // eliminated on another rank
          const BlockClass& B = blocks_[it->first];
          HYMLS_CHECK(B.nb == L.blk_len[li], -3, "linked separator set differs between subdomains");
          // same node order as the owner's block?
          size_t t = 0;
          for (int gi : S.linked[li])
            for (size_t q = 1; q < S.groups[gi].nodes.size(); q++, t++)
              HYMLS_CHECK(B.ids[(size_t)it->second * B.nb + t] == pos2_[g2l_[S.groups[gi].nodes[q]]], -3,
                          "linked separator set ordered differently between subdomains");
          ct_blk[c].push_back({it->first, it->second, base + L.blk_off[li]});
        }
      }, 16);
      for (int64_t c = 0; c < nct; c++) {
        const int ngl = (int)(ct_off[c + 1] - ct_off[c]);
        for (int a = 0; a < ngl; a++) if (ct_vg[ct_off[c] + a] >= 0) rcount[ct_vg[ct_off[c] + a] + 1] += ngl;
        for (auto& bc : ct_blk[c]) blk_cnt[bc.cls][(size_t)bc.blk + 1]++;
      }
    } else {
      // fill (in parallel): every (contributor, owned row group) reserves its ngl slots of the row with one atomic add;
      // the rows are sorted afterwards, so the order of arrival does not matter
      parallel_for(nct, [&](int64_t c) {
        const Subdomain& S = hm_.sd[contributors[c].first];
        const int ngl = (int)(ct_off[c + 1] - ct_off[c]);
        const int64_t base = contributors[c].second;
        const int32_t* vg = ct_vg.data() + ct_off[c];
        for (int a = 0; a < ngl; a++) {
          if (vg[a] < 0) continue;
          int64_t o = __atomic_fetch_add(&rfill[vg[a]], (int64_t)ngl, __ATOMIC_RELAXED);
          for (int b = 0; b < ngl; b++) keys[o++] = ((uint64_t)S.groups[b].nodes[0] << 33) | (uint64_t)(base + a + (int64_t)ngl * b);
        }
      }, 16);
    }
  }
  }  // passes
  fine("(contributions) pass 1 (key fill)");
  if (!direct_schur_) {
    for (size_t c = 0; c < blocks_.size(); c++) {
      BlockClass& B = blocks_[c];
      for (int q = 0; q < B.nblk; q++) blk_cnt[c][(size_t)q + 1] += blk_cnt[c][q];
      B.pull_ptr.assign(blk_cnt[c].begin(), blk_cnt[c].end());
      B.pull_base.assign((size_t)blk_cnt[c][B.nblk], 0);
    }
    // the summands of a block in contributor (= subdomain) order: their order is fixed
    for (auto& list : ct_blk)
      for (auto& bc : list) blocks_[bc.cls].pull_base[(size_t)blk_cnt[bc.cls][bc.blk]++] = bc.src;
    for (size_t c = 0; c < blocks_.size(); c++) {
      BlockClass& B = blocks_[c];
      B.d_ids = dev::upload(B.ids);
      B.d_pull_ptr = dev::upload(B.pull_ptr); B.d_pull_base = dev::upload(B.pull_base);
      B.d_binv = (double*)dev::alloc((size_t)B.nb * B.nb * B.nblk * sizeof(double));
    }
    {
      // one descriptor per block, largest first, for the single-launch inversion; for the apply the large blocks
      // (coarser levels: orders of 1000+) are cut into 64-row tiles so that many waves share one block
      std::vector<dev::BlkD> bd, ba;
      blk_max_nb_ = 0; blk_max_nb_inv_ = 0;
      const int tile_min = std::getenv("HYMLS_MI_BLOCK_TILE_MIN") ? std::atoi(std::getenv("HYMLS_MI_BLOCK_TILE_MIN")) : 128;
      for (auto& B : blocks_) {
        for (int q = 0; q < B.nblk; q++) {
          const dev::BlkD D{B.d_binv + (int64_t)q * B.nb * B.nb, B.d_ids + (int64_t)q * B.nb, B.nb, -1};
          if (!dev::dense_invert_blocked_order(B.nb)) { bd.push_back(D); blk_max_nb_inv_ = std::max(blk_max_nb_inv_, B.nb); }
          if (B.nb <= tile_min) ba.push_back(D);
          else for (int r0 = 0; r0 < B.nb; r0 += 64) { dev::BlkD T = D; T.r0 = r0; ba.push_back(T); }
        }
        blk_max_nb_ = std::max(blk_max_nb_, B.nb);
      }
      auto by_size = [](const dev::BlkD& a, const dev::BlkD& b) { return a.nb > b.nb; };
      std::stable_sort(bd.begin(), bd.end(), by_size);
      std::stable_sort(ba.begin(), ba.end(), by_size);
      n_blk_ = (int32_t)bd.size();
      d_blkd_ = dev::upload(bd);
      n_blk_apply_ = (int32_t)ba.size();
      d_blka_ = dev::upload(ba);
    }
    d_gptr_ = dev::upload(gptr_); d_otw_ = dev::upload(otw_); d_vs_ = dev::upload(vs_);
    d_vrhs_ = (double*)dev::alloc((size_t)std::max(ng_owned, 1) * sizeof(double));
    d_vsol_ = (double*)dev::alloc((size_t)std::max(ng_owned, 1) * sizeof(double));
  }
  fine("(contributions) block tables + uploads");
  lap("contributions");
  // buckets -> CSR pattern (columns = gids) with pull lists: sort every row (column gid, then source position)
  const int64_t nr = red_.n;
  red_.rowptr.assign(nr + 1, 0);
  parallel_for(nr, [&](int64_t r) {
    std::sort(keys.begin() + rcount[r], keys.begin() + rcount[r + 1]);
    int32_t nu = 0;
    for (int64_t k = rcount[r]; k < rcount[r + 1]; k++) nu += k == rcount[r] || (keys[k] >> 33) != (keys[k - 1] >> 33);
    red_.rowptr[r + 1] = nu;
  });
  fine("(pull lists) row sorts");
  for (int64_t r = 0; r < nr; r++) red_.rowptr[r + 1] += red_.rowptr[r];
  red_.col.resize((size_t)red_.rowptr[nr]);     // (not initialised: every entry is written below)
  parallel_for(nr, [&](int64_t r) {
    int64_t e = red_.rowptr[r] - 1;
    for (int64_t k = rcount[r]; k < rcount[r + 1]; k++)
      if (k == rcount[r] || (keys[k] >> 33) != (keys[k - 1] >> 33)) red_.col[++e] = (int32_t)(keys[k] >> 33);
  });
  fine("(pull lists) columns");
  if (fine.on) std::fprintf(stderr, "[hymls_mi]       . reduced matrix: %lld rows, %lld entries, %lld pulls; extraction buffer %lld doubles\n", (long long)nr,
                            (long long)red_.col.size(), (long long)keys.size(), (long long)ext_total_);
  // the pull tables (per entry: where its summands lie in the extraction buffer) are only ever read by the device: the
  // sorted keys go up and a kernel unpacks them there (at 256^3: 3 GB of keys instead of 5.4 GB of tables written on the
  // host and copied)
  {
    uint64_t* d_keys = dev::upload(keys);
    { rawvec<uint64_t>().swap(keys); }
    int64_t* d_rcount = dev::upload(rcount);
    int32_t* d_rowptr = dev::upload(red_.rowptr);
    n_pulls_ = rcount[nr];
    d_red_pull_ptr_ = (int64_t*)dev::alloc(((size_t)red_.rowptr[nr] + 1) * sizeof(int64_t));
    d_red_pull_idx_ = (int64_t*)dev::alloc(std::max<size_t>(1, (size_t)n_pulls_) * sizeof(int64_t));
    dev::build_pull_tables(nr, d_rcount, d_rowptr, d_keys, d_red_pull_ptr_, d_red_pull_idx_);
    dev::sync();
    dev::free(d_keys); dev::free(d_rcount); dev::free(d_rowptr);
  }
  red_.val.resize(red_.col.size());             // (written by every Compute before it is read)
  d_red_val_ = (double*)dev::alloc(std::max<size_t>(1, red_.col.size()) * sizeof(double));
  d_ext_ = (double*)(pre_ext ? pre_ext->take() : dev::alloc(ext_bytes));
  for (auto& cp : cls_) {
    Cls& C = *cp;
    C.d_pick = dev::upload(C.pick);
    C.d_lgptr = dev::upload(C.lgptr);
    C.d_tvloc = dev::upload(C.tvloc);
    if (!direct_schur_) {
      ivec glink(C.ngl, -1), goff(C.ngl, 0), lblen(C.blk_len.begin(), C.blk_len.end());
      std::vector<int64_t> lboff(C.blk_off.begin(), C.blk_off.end());
      for (size_t li = 0; li < C.llinked.size(); li++) {
        int32_t off = 0;
        for (int gi : C.llinked[li]) {
          if (C.blk_len[li] > 0) { glink[gi] = (int32_t)li; goff[gi] = off; }
          off += C.lgptr[gi + 1] - C.lgptr[gi] - 1;
        }
      }
      C.d_glink = dev::upload(glink); C.d_goff = dev::upload(goff); C.d_lblen = dev::upload(lblen); C.d_lboff = dev::upload(lboff);
    }
  }
  fine("(pull lists) uploads + allocations");
  lap("pull lists");
  // ---- tables of the fused interior solve (classes whose vectors fit in LDS)
  constexpr int32_t LDS_CAP = 12288;  // doubles (96 KiB)
  std::vector<dev::PlanD> plans;
  std::vector<dev::FusedSub> subs;
  cls_fused_.assign(cls_.size(), 0);
  fused_lds_ = 0; fused_front_lds_ = 0; fused_vec_lds_ = 0;
  for (size_t c = 0; c < cls_.size(); c++) {
    Cls& C = *cls_[c];
    const int32_t need = C.lu.plan.nI + C.lu.plan.contrib_size + std::max(C.lu.plan.max_level_rows, 384) +
                         (int32_t)(C.lu.plan.fronts.size() * 6 + 1);   // X | C | F (+ R inside) | compact front descriptors (48 B)
    bool any_big = false;
    for (auto& L : C.lu.plan.big_levels) any_big |= !L.empty();
    const bool fused = !(any_big || C.lu.plan.max_level_rows > dev::FUSED_MAX_ITEMS || need > LDS_CAP || C.lu.plan.nI == 0 ||
                         C.lu.plan.fw_items.empty() || std::getenv("HYMLS_MI_NO_FUSED_SOLVE"));
    C.lu.packed = fused && !std::getenv("HYMLS_MI_NO_PACKED_PANELS");
    C.lu.contrib_nv = fused ? 1 : dev::NV_MAX;   // (the fused kernel keeps its contribution vectors in LDS)
    C.lu.upload(SCRATCH_BUDGET, true);
    plans.push_back(C.lu.dplan);
    if (!fused) continue;
    cls_fused_[c] = 1;
    fused_lds_ = std::max(fused_lds_, need);
    fused_front_lds_ = std::max(fused_front_lds_, (int32_t)(C.lu.plan.fronts.size() * 6 + 1));
    fused_vec_lds_ = std::max(fused_vec_lds_, need - (int32_t)(C.lu.plan.fronts.size() * 6 + 1));
    for (size_t b = 0; b < C.lu.members.size(); b++)
      subs.push_back(dev::FusedSub{C.lu.batch.factor + (int64_t)b * C.lu.plan.factor_size, C.lu.h_xoff[b], (int32_t)c});
  }
  // (Measured and dropped, gpurun_out/r3ao: the workgroups of the heaviest classes first, so that the tail of the launch consists
  // of the light ones -- 8.89 against 8.81 ms per launch, averaged over both set-up orders of the A/B harness.)
  n_fsubs_ = (int32_t)subs.size();
  if (std::getenv("HYMLS_MI_VERBOSE"))
    std::fprintf(stderr, "[hymls_mi] rank %d level %d: fused interior solve for %d of %zu subdomains, LDS %d doubles (%.1f KiB)\n",
                 comm_->rank, level_, n_fsubs_, my_sds_.size(), fused_lds_, fused_lds_ * 8.0 / 1024);
  d_fplans_ = dev::upload(plans);
  d_fsubs_ = dev::upload(subs);
  lap("class uploads");
  // ---- merged level solve for the classes that do not fit (large subdomains of the coarser levels):
  // one launch per tree level and sweep for all of them together
  cls_merged_.assign(cls_.size(), 0);
  std::vector<std::pair<int32_t, const BatchedLU*>> merged;
  for (size_t c = 0; c < cls_.size(); c++) {
    Cls& C = *cls_[c];
    if (cls_fused_[c] || C.lu.plan.nI == 0 || std::getenv("HYMLS_MI_NO_MERGED_SOLVE") || !merged_solve_fits(C.lu.plan)) continue;
    cls_merged_[c] = 1;
    merged.emplace_back((int32_t)c, &C.lu);
  }
  merged_.build(merged);
  if (merged_.nsubs) {
    d_ytmp_ = (double*)dev::alloc((size_t)std::max(n1_ + ngi_ + n2_, 1) * sizeof(double));
    if (std::getenv("HYMLS_MI_VERBOSE"))
      std::fprintf(stderr, "[hymls_mi] rank %d level %d: merged level solve for %d subdomains, %zu tree levels\n",
                   comm_->rank, level_, merged_.nsubs, merged_.fw_lds.size());
  }
}

// the extraction records of boundary subdomains travel to the ranks that own some of their separators
// (the Export with SumInto of the reference's SchurComplement::Construct, src/HYMLS_SchurComplement.cpp:195-260;
// here the owner pulls, in subdomain order, so the sums are formed in the same order as on one rank)
void LevelSolver::exchange_records() {
  if (!rec_any_) return;
  double* sb = comm_->send_arena(rec_nsend_);
  double* rb = comm_->recv_arena(rec_nrecv_);
  int64_t off = 0;
  for (auto& sg : rec_send_) { dev::d2d(sb + off, d_ext_ + sg.off, (size_t)sg.len * sizeof(double)); off += sg.len; }
  const int ierr = comm_->alltoallv(comm_->ctx, sb, rec_scnt_.data(), rb, rec_rcnt_.data(), (int32_t)sizeof(double), 1);
  HYMLS_CHECK(ierr == 0, -3, std::string("device all-to-all failed in the transport ") + rccl_last_error(*comm_));
  if (rec_nrecv_) dev::d2d(d_ext_ + ext_recv_base_, rb, (size_t)rec_nrecv_ * sizeof(double));
}

// set_values without a copy: the level takes the array and hands back its old one (same size)
void LevelSolver::swap_values(vvec& val) {
  if (comm_->distributed() && initialized_) { set_values(val); return; }
  HYMLS_CHECK(val.size() == K_.val.size(), -2, "SetMatrix: pattern changed");
  K_.val.swap(val);
}

void LevelSolver::set_values(const vvec& val) {
  if (comm_->distributed() && initialized_) {   // values of the rows as they were given; keep the local ones
    HYMLS_CHECK(val.size() == given_nnz_, -2, "SetMatrix: pattern changed");
    parallel_for((int64_t)keep_entries_.size(), [&](int64_t i) { K_.val[i] = val[keep_entries_[i]]; }, 1 << 16);
    return;
  }
  HYMLS_CHECK(val.size() == K_.val.size(), -2, "SetMatrix: pattern changed");
  parallel_memcpy(K_.val.data(), val.data(), val.size() * sizeof(double));
}

// the reduced matrix of all ranks: every rank contributes the rows it owns; rows in rank order,
// columns turned into global row numbers and sorted.  On one rank this is just red_.
// the pattern part of assemble_reduced: gathered gids, clusters, global columns and the permutation of the values.  Depends on
// Initialize-time data only, so that the first Compute of an unsharded handle runs it on a helper thread next to the
// factorisation (compute()).
void LevelSolver::prepare_reduced_pattern(const dvec* tvn) {
  FineLap fine;
  if (!glob_ready_) {
    ivec my_gids(red_.n);
    if (direct_schur_) for (int k = 0; k < n2_; k++) my_gids[k] = gids_[sep_row_[k]];
    else for (int g = 0; g < red_.n; g++) my_gids[g] = gids_[sep_row_[vs_[g]]];
    ivec my_len(red_.n);
    for (int r = 0; r < red_.n; r++) my_len[r] = red_.rowptr[r + 1] - red_.rowptr[r];
    std::vector<int64_t> cnt;
    glob_gids_ = comm_->allgather(my_gids, &cnt);
    glob_row_off_.assign(1, 0);
    for (int64_t c : cnt) glob_row_off_.push_back(glob_row_off_.back() + c);
    ivec len = comm_->allgather(my_len);
    cvec colg_all;
    if (comm_->distributed()) colg_all = comm_->allgather(red_.col);
    const cvec& colg = comm_->distributed() ? colg_all : red_.col;      // (one rank: no copy of 0.3 G columns)
    if (tvn) glob_tv_ = comm_->allgather(*tvn);
    fine("(reduced) gather of gids / lengths / columns");
    {
      // clusters of the rows = the subdomains that list the node (the last-level direct solver dissects along them);
      // every rank contributes the lists of its rows and the centres of its subdomains, so that a sharded run orders
      // the coarse system exactly like a one-rank run
      ivec my_cnt(red_.n), my_clu;
      for (int r = 0; r < red_.n; r++) {
        const int k = direct_schur_ ? r : vs_[r];
        for (int t = sep_sd_ptr_[k]; t < sep_sd_ptr_[k + 1]; t++) my_clu.push_back(sep_sd_[t]);
        my_cnt[r] = sep_sd_ptr_[k + 1] - sep_sd_ptr_[k];
      }
      ivec cntg = comm_->allgather(my_cnt);
      glob_clu_ = comm_->allgather(my_clu);
      glob_clu_ptr_.assign(1, 0);
      for (int32_t c : cntg) glob_clu_ptr_.push_back(glob_clu_ptr_.back() + c);
      ivec my_ctr;
      for (int sd : my_sds_) { my_ctr.push_back(sd); for (int a = 0; a < 3; a++) my_ctr.push_back(sd_center_[3 * (size_t)sd + a]); }
      ivec ctr = comm_->allgather(my_ctr);
      glob_sd_center_.assign(sd_center_.size(), 0);
      for (size_t t = 0; t + 3 < ctr.size(); t += 4) for (int a = 0; a < 3; a++) glob_sd_center_[3 * (size_t)ctr[t] + a] = ctr[t + 1 + a];
    }
    fine("(reduced) clusters of the rows");
    const int64_t N = (int64_t)glob_gids_.size();
    HYMLS_CHECK(N < (int64_t)1 << 31, -2, "reduced matrix too large for 32-bit row numbers");
    ivec row_of(ngid_, -1);     // gid -> global row
    for (int64_t i = 0; i < N; i++) {
      HYMLS_CHECK(row_of[glob_gids_[i]] < 0, -3, "separator owned by two ranks");
      row_of[glob_gids_[i]] = (int32_t)i;
    }
    glob_.n = (int32_t)N;
    glob_.rowptr.assign(N + 1, 0);
    for (int64_t i = 0; i < N; i++) glob_.rowptr[i + 1] = glob_.rowptr[i] + len[i];
    glob_.col.resize(colg.size());
    glob_perm_.resize(colg.size());
    std::atomic<int> missing{0};
    parallel_for(N, [&](int64_t i) {
      std::vector<std::pair<int32_t, int64_t>> row;
      for (int64_t e = glob_.rowptr[i]; e < glob_.rowptr[i + 1]; e++) {
        if (row_of[colg[e]] < 0) missing = 1;
        row.emplace_back(row_of[colg[e]], e);
      }
      std::sort(row.begin(), row.end());
      for (size_t k = 0; k < row.size(); k++) { glob_.col[glob_.rowptr[i] + k] = row[k].first; glob_perm_[row[k].second] = glob_.rowptr[i] + (int64_t)k; }
    });
    HYMLS_CHECK(missing == 0, -3, "reduced matrix refers to a node nobody owns");
    fine("(reduced) global columns, sorted rows");
    glob_.val.resize(glob_.col.size());
    glob_ready_ = true;
  }
}

const Csr& LevelSolver::assemble_reduced(ivec& row_gids, dvec* tvn) {
  FineLap fine;
  prepare_reduced_pattern(tvn);
  {
    // (one rank: no copy of the 0.2 G values of a 256^3 run)
    vvec gathered;
    if (comm_->distributed()) gathered = comm_->allgather(red_.val);
    const vvec& vals = comm_->distributed() ? gathered : red_.val;
    HYMLS_CHECK(vals.size() == glob_.val.size(), -3, "reduced matrix changed its pattern between two Compute calls");
    parallel_for((int64_t)vals.size(), [&](int64_t e) { glob_.val[glob_perm_[e]] = vals[e]; }, 1 << 16);
  }
  fine("(reduced) values into place");
  row_gids = glob_gids_;
  return glob_;
}

// vectors between this level's owners of the V-sum nodes and the next level's layout
void LevelSolver::build_handoff(const ivec& next_owned) {
  if (!comm_->distributed()) return;
  std::unordered_map<int32_t, int32_t> row_of;
  row_of.reserve(glob_gids_.size() * 2);
  for (size_t i = 0; i < glob_gids_.size(); i++) row_of[glob_gids_[i]] = (int32_t)i;
  std::vector<std::vector<int64_t>> want(comm_->size);
  std::vector<ivec> dst(comm_->size);
  for (size_t k = 0; k < next_owned.size(); k++) {
    const int64_t i = row_of.at(next_owned[k]);
    const int q = (int)(std::upper_bound(glob_row_off_.begin(), glob_row_off_.end(), i) - glob_row_off_.begin()) - 1;
    want[q].push_back(i - glob_row_off_[q]);
    dst[q].push_back((int32_t)k);
  }
  const int64_t nown = red_.n;
  xch_down_.build(*comm_, want, dst, [nown](int64_t k) { return k < nown ? (int32_t)k : -1; });
  dev::free(d_nrhs_); dev::free(d_nsol_);
  n_next_owned_ = (int64_t)next_owned.size();
  d_nrhs_ = (double*)dev::alloc(std::max<size_t>(1, next_owned.size()) * nvec_alloc_ * sizeof(double));
  d_nsol_ = (double*)dev::alloc(std::max<size_t>(1, next_owned.size()) * nvec_alloc_ * sizeof(double));
}

// which streams the factorisation of this level uses: the classes of the coarser levels round robin over the side streams,
// the chunks of the finest level alternating between chunk_streams of them
void LevelSolver::stream_plan(bool& side, int& chunk_streams) const {
  const bool no_side = std::getenv("HYMLS_MI_NO_SIDE_STREAMS") != nullptr;
  side = level_ >= 1 && cls_.size() > 1 && !no_side;
  static const int chunk_streams_env = std::getenv("HYMLS_MI_CHUNK_STREAMS") ? std::atoi(std::getenv("HYMLS_MI_CHUNK_STREAMS")) : 2;
  chunk_streams = level_ == 0 && !no_side ? std::max(0, std::min(chunk_streams_env, dev::side_streams())) : 0;
}

// (Measured and dropped, gpurun_out/r3ag: requesting the scratch arenas of the first Compute -- 6 - 8 GiB per stream, 1.4 s of
// hipMalloc -- on helper threads during Initialize as well.  Together with the factor arrays that is 50 GiB for the driver to
// clear, and every HIP call of the setup thread queues behind it: A12 / A21 0.46 -> 1.7 s, pull lists 0.63 -> 1.0 s.)
// next test vector = V-sum part of H * testvector (reference src/HYMLS_SchurPreconditioner.cpp:569-573)
void LevelSolver::next_test_vector(dvec& tvn) const {
  const int ng = (int)vs_.size();
  tvn.resize(ng);
  for (int g = 0; g < ng; g++) {
    double dot = 0;
    for (int i = gptr_[g]; i < gptr_[g + 1]; i++) dot += otw_[i] * tv_[sep_row_[i]];
    tvn[g] = 2.0 * otw_[gptr_[g]] * dot - tv_[sep_row_[gptr_[g]]];
  }
}

void LevelSolver::compute() {
  HYMLS_CHECK(initialized_, -1, "level not initialized");
  dev::Range range("Preconditioner", level_ + 1, "Compute");
  const bool verbose = std::getenv("HYMLS_MI_VERBOSE") != nullptr;
  double t0 = wall();
  auto lap = [&](const char* what) {
    if (!verbose) return;
    dev::sync();
    std::fprintf(stderr, "[hymls_mi] rank %d level %d compute: %-28s %.2f s\n", comm_->rank, level_, what, wall() - t0);
    t0 = wall();
  };
  dev::h2d(d_kval_, K_.val.data(), K_.val.size() * sizeof(double));
  mv_stale_ = true;
  dev::gather((int64_t)a12_col_.size(), d_a12_src_, d_kval_, d_a12_val_);
  dev::gather((int64_t)a21_col_.size(), d_a21_src_, d_kval_, d_a21_val_);
  // ---- interior factorisations + separator blocks, class by class, chunk by chunk.  The coarser levels have many
  // classes with one or two large subdomains each: their launch chains are independent and run on side streams
  // The finest level has a few classes with thousands of members: there the CHUNKS of a class alternate between two side
  // streams, so that the narrow launches at the top of one chunk's elimination tree overlap the leaf launches of the next
  // (HYMLS_MI_CHUNK_STREAMS = 0 switches that off, 1..NSIDE sets the number of streams).
  bool side = false;
  int chunk_streams = 0;
  stream_plan(side, chunk_streams);
  struct MainStreamGuard { ~MainStreamGuard() { try { dev::use_stream(0); } catch (...) {} } } back_to_main;   // also when a launch throws
  for (auto& cp : cls_) dev::zero(cp->lu.batch.flag, 4 * sizeof(int32_t));   // (on the main stream, before any side stream starts)
  // First Compute of an unsharded handle: the pattern of the reduced matrix (0.5 s of host work at 256^3) is put together on a
  // helper thread while this thread launches the factorisation.  (Sharded: the gathers are collective, they stay in line.)
  dvec tvn_early;
  std::thread pattern_thread;
  std::exception_ptr pattern_err;
  struct JoinGuard { std::thread& t; ~JoinGuard() { if (t.joinable()) t.join(); } } pattern_join{pattern_thread};
  if (!glob_ready_ && !direct_schur_ && !comm_->distributed() && !std::getenv("HYMLS_MI_NO_PATTERN_THREAD")) {
    next_test_vector(tvn_early);
    pattern_thread = std::thread([&] { try { prepare_reduced_pattern(&tvn_early); } catch (...) { pattern_err = std::current_exception(); } });
  }
  // (the subdomain factorisations and what is kept of the Schur complement come out of the same launches here: one range)
  auto range_mb = std::make_unique<dev::Range>("MatrixBlock", level_ + 1, "Compute");
  if (side || chunk_streams) dev::fork_streams();
  int64_t chunk_id = 0;
  // (coarser levels: classes in their order, round robin over the side streams.  Measured and not kept: classes in descending
  // order of work on the least loaded stream -- 1.36 s instead of 1.13 s for level 2 of the 256^3 run with four streams, the
  // largest classes then run at the same time; six or eight streams: 1.12 s and 15-27 GiB more memory.)
  for (size_t c = 0; c < cls_.size(); c++) {
    Cls& C = *cls_[c];
    if (side) dev::use_stream(1 + (int)(c % dev::side_streams()));
    const int nb = (int)C.lu.members.size();
    for (int b0 = 0; b0 < nb; b0 += C.lu.chunk) {
      const int nbc = std::min(C.lu.chunk, nb - b0);
      if (chunk_streams) dev::use_stream(1 + (int)(chunk_id++ % chunk_streams));
      C.lu.factor_chunk(d_kval_, b0, nbc);
      C.lu.repack_chunk(b0, nbc);
      if (C.pat.nS == 0) continue;
      if (!direct_schur_) {
        // orthogonal transformation + dropping: only the kept entries are formed, in one read pass over the block
        if (dev::sblock_kept_fits(C.pat.nS, C.ngl)) {
          const dev::KeptD K{C.pat.nS, C.ngl, C.d_lgptr, C.d_glink, C.d_goff, C.d_lboff, C.d_lblen};
          dev::sblock_kept(K, C.d_tvloc + (size_t)b0 * C.pat.nS, C.lu.batch.sblock, d_ext_ + C.ext_base + (int64_t)b0 * C.ext_size,
                           C.ext_size, nbc);
        } else {
          // separator blocks too large for the fused kernel's LDS vectors: two-sided Householder in place, then extraction
          dev::sblock_transform(C.pat.nS, C.ngl, C.d_lgptr, C.d_tvloc + (size_t)b0 * C.pat.nS, C.lu.batch.sblock, nbc);
          dev::sblock_extract(C.pat.nS, C.ext_size, C.d_pick, C.lu.batch.sblock, d_ext_ + C.ext_base + (int64_t)b0 * C.ext_size, C.ext_size, nbc);
        }
      } else {
        dev::sblock_extract(C.pat.nS, C.ext_size, C.d_pick, C.lu.batch.sblock,
                            d_ext_ + C.ext_base + (int64_t)b0 * C.ext_size, C.ext_size, nbc);
      }
    }
  }
  if (side || chunk_streams) dev::join_streams();
  range_mb.reset();
  dev::Range range_sp("SchurPreconditioner", level_ + 1, "Compute");
  int32_t bad = 0, grown = 0;
  double growth = 0.0;
  for (auto& cp : cls_) { double g = 0.0; const int32_t f = cp->lu.check_flag(&g); bad |= (f & 1); grown |= (f & 2) >> 1; growth = std::max(growth, g); }
  if (verbose) std::fprintf(stderr, "[hymls_mi] rank %d level %d compute: largest element growth of a pivot block %.3g\n", comm_->rank, level_, growth);
  // (collective: every rank has to reach the exchanges below, so errors are agreed on first)
  HYMLS_CHECK(comm_->allsum(bad) == 0, -4, "subdomain factorisation hit a zero or non-finite pivot (level " +
                                               std::to_string(level_) + ")");
  HYMLS_CHECK(comm_->allsum(grown) == 0, -4, "subdomain factorisation without pivoting is unstable for this matrix: element growth " +
                                             std::to_string(growth) + " > 1e8 (level " + std::to_string(level_) + "); the factor would be inaccurate");
  lap("factor + transform + extract");
  compute_border();
  exchange_records();
  // ---- assemble what is kept of the Schur complement
  dev::pull_sum((int64_t)red_.col.size(), d_red_pull_ptr_, d_red_pull_idx_, d_ext_, d_red_val_);
  dev::d2h(red_.val.data(), d_red_val_, red_.val.size() * sizeof(double));
  ivec next_gids;
  if (direct_schur_) {
    // Preconditioner.cpp:485-500: S assembled, DropByValue (RelZeroDiag), CoarseSolver
    const Csr& G = assemble_reduced(next_gids, nullptr);
    Csr S = drop_by_value(G, SMALL_ENTRY, 1);
    DirectSolver* ds = next_is_direct_ && !next_level_ ? dynamic_cast<DirectSolver*>(next_.get()) : nullptr;
    if (!(ds && ds->refactor(S, next_gids, p_.fix_gid, p_, bm_ > 0))) {
      next_.reset();
      next_level_ = nullptr;
      next_is_direct_ = true;
      next_.reset(new DirectSolver(S, next_gids, p_.fix_gid, ngid_, p_, &glob_clu_ptr_, &glob_clu_, &glob_sd_center_, bm_ > 0));
      build_handoff(next_gids);
    }
    set_next_border();
    return;
  }
  dev::zero(d_flag_, sizeof(int32_t));
  auto range_fb = std::make_unique<dev::Range>("SchurPreconditioner", level_ + 1, "factor blocks");
  for (auto& B : blocks_) {
    // block values = sum of the contributions of every adjacent subdomain, then LU + inverse
    // (Ifpack_DenseContainer::Compute -> dgetrf in the reference, SchurPreconditioner.cpp:284-291)
    const int64_t bl = (int64_t)B.nb * B.nb;
    dev::pull_sum_blocks(bl, B.nblk, B.d_pull_ptr, B.d_pull_base, d_ext_, B.d_binv);
  }
  if (getenv("HYMLS_MI_BLOCK_STATS")) {
    std::map<int, int> hist;
    double cube = 0;
    for (auto& B : blocks_) { hist[B.nb] += B.nblk; cube += (double)B.nblk * B.nb * B.nb * B.nb; }
    fprintf(stderr, "level %d separator blocks: %d, max order %d, sum nb^3 %.3e;", level_, n_blk_, blk_max_nb_, cube);
    for (auto& h : hist) fprintf(stderr, " %dx%d", h.second, h.first);
    fprintf(stderr, "\n");
  }
  {
    // the large blocks of the coarser levels group by group (blocked, matrix cores), spread over the side streams;
    // everything else in one launch on the main stream meanwhile
    bool any_big = false;
    for (auto& B : blocks_) any_big = any_big || dev::dense_invert_blocked_order(B.nb);
    struct MainStreamGuard { ~MainStreamGuard() { try { dev::use_stream(0); } catch (...) {} } } back_to_main;
    if (any_big) {
      dev::fork_streams();
      int k = 0;
      for (auto& B : blocks_)
        if (dev::dense_invert_blocked_order(B.nb)) {
          dev::use_stream(1 + (k++ % dev::side_streams()));
          dev::dense_invert(B.nb, B.nblk, B.d_binv, d_flag_);
        }
      dev::use_stream(0);
    }
    dev::dense_invert_all(n_blk_, d_blkd_, blk_max_nb_inv_, d_flag_);
    if (any_big) dev::join_streams();
  }
  int32_t flag = 0;
  dev::d2h(&flag, d_flag_, sizeof flag);
  range_fb.reset();
  HYMLS_CHECK(comm_->allsum(flag != 0) == 0, -4,
              "singular separator block on level " + std::to_string(level_) +
                  " (3D Stokes-C needs the Skew Cartesian partitioner: isolated pressure 'tubes' on subdomain edges)");
  // ---- next level (ComputeNextLevel, reference src/HYMLS_SchurPreconditioner.cpp:520-629)
  const int ng = (int)vs_.size();
  (void)ng;
  if (pattern_thread.joinable()) pattern_thread.join();
  if (pattern_err) std::rethrow_exception(pattern_err);
  dvec tvn;
  if (!glob_ready_) next_test_vector(tvn);
  lap("pull + separator blocks");
  auto range_nl = std::make_unique<dev::Range>("SchurPreconditioner", level_ + 1, "ComputeNextLevel");
  const Csr& G = assemble_reduced(next_gids, &tvn);
  Csr& R = next_R_;
  drop_by_value(G, SMALL_ENTRY, 0, R);
  { FineLap f2; f2.t = t0; f2("(reduced) assemble + drop, total"); }
  lap("reduced matrix (host)");
  if (level_ + 1 < p_.levels) {
    next_is_direct_ = false;
    if (next_level_ && next_pattern_key_rowptr_.size() == R.rowptr.size() && next_pattern_key_col_.size() == R.col.size() &&
        parallel_equal(next_pattern_key_rowptr_.data(), R.rowptr.data(), R.rowptr.size() * sizeof(int32_t)) &&
        parallel_equal(next_pattern_key_col_.data(), R.col.data(), R.col.size() * sizeof(int32_t))) {
      next_level_->swap_values(R.val);   // (sharded: the next level keeps the rows it needs; R keeps a buffer of the right size)
    } else {
      FineLap fnl;
      parallel_assign(next_pattern_key_rowptr_, R.rowptr.data(), R.rowptr.size());
      parallel_assign(next_pattern_key_col_, R.col.data(), R.col.size());
      fnl("(next level) pattern key copy");
      next_level_ = new LevelSolver(p_.next_level(), level_ + 1, ngid_, comm_);
      next_.reset(next_level_);
      next_level_->set_rows(R, next_gids, glob_tv_, R.n);
      fnl("(next level) set_rows");
      next_level_->initialize();
      fnl("(next level) initialize");
      build_handoff(next_level_->owned_gids());
      fnl("(next level) hand-off");
    }
    next_level_->profiling = false;  // phases are reported for the top level only
    lap("next level initialize");
    set_next_border();
    next_level_->compute();
    lap("next level compute");
  } else {
    DirectSolver* ds = next_is_direct_ && !next_level_ ? dynamic_cast<DirectSolver*>(next_.get()) : nullptr;
    if (!(ds && ds->refactor(R, next_gids, p_.fix_gid, p_, bm_ > 0))) {
      next_.reset();            // (the old factor goes before the new one is built)
      next_level_ = nullptr;
      next_is_direct_ = true;
      next_.reset(new DirectSolver(R, next_gids, p_.fix_gid, ngid_, p_, &glob_clu_ptr_, &glob_clu_, &glob_sd_center_, bm_ > 0));
      build_handoff(next_gids);
    }
    set_next_border();
    lap("coarse solver");
  }
}

void LevelSolver::interior_solve(double* x1) { interior_solve_mv(x1, n1_ + ngi_, 1); }

// x1 <- A11^{-1} x1 for nv columns (leading dimension ld): the factor panels are streamed once per group of columns
void LevelSolver::interior_solve_mv(double* x1, int64_t ld, int nv) {
  dev::Range range("MatrixBlock", level_ + 1, "ApplyInverse");
  if (n_fsubs_ > 0) {
    if (nv == 1) dev::interior_solve_fused(n_fsubs_, d_fsubs_, d_fplans_, fused_lds_, x1);
    else dev::interior_solve_fused_mv(n_fsubs_, d_fsubs_, d_fplans_, fused_vec_lds_ + fused_front_lds_, fused_front_lds_, x1, ld, nv);
  }
  if (merged_.nsubs > 0) {
    ensure_nvec(nv);   // (the scratch shares the leading dimension of x: ld <= n1 + ngi + n2)
    merged_.solve(d_fplans_, x1, d_ytmp_, ld, nv);
  }
  for (size_t c = 0; c < cls_.size(); c++)
    if (!cls_fused_[c] && !cls_merged_[c])
      for (int v = 0; v < nv; v++) cls_[c]->lu.solve(x1 + v * ld);
}

// buffers of the apply for nv right-hand sides (column-major, every vector with the single-vector layout)
void LevelSolver::ensure_nvec(int nv) {
  if (nv <= nvec_alloc_) return;
  dev::sync();
  auto grow = [&](double*& p, int64_t len) { dev::free(p); p = (double*)dev::alloc((size_t)std::max<int64_t>(len, 1) * nv * sizeof(double)); };
  grow(d_z_, n1_ + ngi_ + n2_); grow(d_t1_, n1_); grow(d_t2_, n2_ + ngs_);
  if (!direct_schur_) { const int64_t ng = (int64_t)vs_.size(); grow(d_vrhs_, ng); grow(d_vsol_, ng); }
  if (d_nrhs_) { grow(d_nrhs_, n_next_owned_); grow(d_nsol_, n_next_owned_); }
  if (d_ytmp_) grow(d_ytmp_, n1_ + ngi_ + n2_);
  nvec_alloc_ = nv;
}

// rhs/sol: entries of the nodes this rank owns on this level that go on to the next one
void LevelSolver::next_apply(const double* rhs, double* sol, int64_t ld, int nv) {
  if (!comm_->distributed()) { next_->apply_inverse_mv(rhs, ld, sol, ld, nv); return; }
  const int64_t ldn = std::max<int64_t>(n_next_owned_, 1);
  for (int v = 0; v < nv; v++) xch_down_.forward(rhs + v * ld, d_nrhs_ + v * ldn);
  next_->apply_inverse_mv(d_nrhs_, ldn, d_nsol_, ldn, nv);
  for (int v = 0; v < nv; v++) {
    if (next_is_direct_) {
      // every rank solved the whole coarse system: keep the entries owned here (rows are in rank order)
      if (red_.n) dev::d2d(sol + v * ld, d_nsol_ + v * ldn + glob_row_off_[comm_->rank], (size_t)red_.n * sizeof(double));
    } else {
      xch_down_.backward(d_nsol_ + v * ldn, sol + v * ld);
    }
  }
}

void LevelSolver::schur_apply(double* rhs2, int64_t ldr, double* x2, int64_t ldx, int nv) {
  dev::Range range("SchurPreconditioner", level_ + 1, "ApplyInverse");
  if (global_n2_ == 0) return;
  if (direct_schur_) {
    // (the direct solver of the whole Schur complement takes any leading dimensions)
    if (!comm_->distributed()) { next_->apply_inverse_mv(rhs2, ldr, x2, ldx, nv); return; }
    for (int v = 0; v < nv; v++) next_apply(rhs2 + v * ldr, x2 + v * ldx, 0, 1);
    return;
  }
  // SchurPreconditioner::ApplyInverse (reference src/HYMLS_SchurPreconditioner.cpp:1010-1093)
  const int ng = (int)vs_.size();
  const int64_t ldv = std::max(ng, 1);
  for (int v = 0; v < nv; v++) dev::ot_apply(ng, d_gptr_, d_otw_, rhs2 + v * ldr);                   // B' = H rhs
  // (measured and not kept, profiles/r03_f_ab_*: the separator-block products on a side stream while the next level runs on
  // the main one -- both are bandwidth-bound, the Schur phase took 8.60 ms with and 8.56 ms without the overlap)
  dev::blocks_apply_all_mv(n_blk_apply_, d_blka_, blk_max_nb_, rhs2, ldr, x2, ldx, nv);
  for (int v = 0; v < nv; v++) dev::gather(ng, d_vs_, rhs2 + v * ldr, d_vrhs_ + v * ldv);
  if (profiling && level_ == 0) dev::mark(4, true);
  next_apply(d_vrhs_, d_vsol_, ldv, nv);
  if (profiling && level_ == 0) dev::mark(4, false);
  for (int v = 0; v < nv; v++) {
    dev::scatter(ng, d_vs_, d_vsol_ + v * ldv, x2 + v * ldx);
    dev::ot_apply(ng, d_gptr_, d_otw_, x2 + v * ldx);                                              // Y = H Y
  }
}

void LevelSolver::apply_inverse(const double* b, double* x) { apply_inverse_mv(b, 0, x, 0, 1); }

// Preconditioner::ApplyInverse (reference src/HYMLS_Preconditioner.cpp:930-1070) for nv right-hand sides; the two
// interior solves, the separator blocks and the coarser levels read their factors once per group of up to NV_MAX columns
void LevelSolver::apply_inverse_mv(const double* b, int64_t ldb, double* x, int64_t ldx, int nv) {
  dev::Range range("Preconditioner", level_ + 1, "ApplyInverse");
  HYMLS_CHECK(next_ != nullptr || global_n2_ == 0, -1, "The preconditioner has not yet been computed.");
  ensure_nvec(nv);
  const int64_t ldz = n1_ + ngi_ + n2_, ld1 = std::max(n1_, 1), ld2 = std::max(n2_ + ngs_, 1);
  double* z1 = d_z_;                    // per column: [x1 | x1 of the neighbours next to my separators | x2]
  double* z2 = d_z_ + n1_ + ngi_;
  if (profiling) dev::mark(0, true);
  for (int v = 0; v < nv; v++) {
    dev::gather(n1_, d_inperm_, b + v * ldb, z1 + v * ldz);        // b1
    dev::gather(n2_, d_inperm_ + n1_, b + v * ldb, z2 + v * ldz);  // b2
  }
  if (profiling) dev::mark(1, true);
  interior_solve_mv(z1, ldz, nv);                                  // x1 = A11 \ b1
  if (profiling) { dev::mark(1, false); dev::mark(2, true); }
  for (int v = 0; v < nv; v++) {
    xch_int_.forward(z1 + v * ldz, z1 + v * ldz);                  // halo: interior layer of the neighbouring ranks
    dev::spmv(n2_, d_a21_row_, d_a21_col_, d_a21_val_, z1 + v * ldz, z2 + v * ldz, -1.0, 1.0, (int64_t)a21_col_.size());  // b2 - A21 x1
  }
  if (profiling) { dev::mark(2, false); dev::mark(3, true); }
  schur_apply(z2, ldz, d_t2_, ld2, nv);                            // x2
  if (profiling) { dev::mark(3, false); dev::mark(2, true); }
  for (int v = 0; v < nv; v++) {
    xch_sep_.forward(d_t2_ + v * ld2, d_t2_ + v * ld2);            // halo: separators owned by the neighbouring ranks
    dev::spmv(n1_, d_a12_row_, d_a12_col_, d_a12_val_, d_t2_ + v * ld2, d_t1_ + v * ld1, 1.0, 0.0, (int64_t)a12_col_.size());  // y1 = A12 x2
  }
  if (profiling) { dev::mark(2, false); dev::mark(1, true); }
  interior_solve_mv(d_t1_, ld1, nv);                               // A11 \ y1
  if (profiling) dev::mark(1, false);
  for (int v = 0; v < nv; v++) {
    dev::axpby(n1_, -1.0, d_t1_ + v * ld1, 1.0, z1 + v * ldz);     // x1 -= ...
    dev::scatter(n1_, d_inperm_, z1 + v * ldz, x + v * ldx);
    dev::scatter(n2_, d_inperm_ + n1_, d_t2_ + v * ld2, x + v * ldx);
  }
  if (profiling) dev::mark(0, false);
}

// plan of the sharded K x: which columns of my owned rows live elsewhere, and who owns them.  The owners are found by
// publishing the wanted gids to everybody (boundary layers only) and letting every rank claim what it owns.
void LevelSolver::build_matvec() {
  const int n = K_.n, nown = (int)owned_gids_.size();
  ivec user(n, -1);
  {
    int u = 0;
    for (int i = 0; i < nrows_; i++) if (intidx_[i] >= 0 || (pos2_[i] >= 0 && pos2_[i] < n2_)) user[i] = u++;
  }
  ivec row(1, 0), col, src, node(nown);
  std::vector<char> wanted(n, 0);
  ivec want;
  for (int i = 0; i < nrows_; i++) {
    if (user[i] < 0) continue;
    node[user[i]] = i;
    for (int e = K_.rowptr[i]; e < K_.rowptr[i + 1]; e++) {
      const int c = K_.col[e];
      col.push_back(c); src.push_back(e);
      if (user[c] < 0 && !wanted[c]) { wanted[c] = 1; want.push_back(gids_[c]); }
    }
    row.push_back((int32_t)col.size());
  }
  // (rows were visited in local order = user order)
  std::vector<int64_t> cnt;
  ivec all = comm_->allgather(want, &cnt);
  std::vector<ivec> claim(comm_->size);
  {
    int64_t off = 0;
    for (int q = 0; q < comm_->size; q++) {
      for (int64_t t = off; t < off + cnt[q]; t++) {
        const int l = g2l_[all[t]];
        if (q != comm_->rank && l >= 0 && user[l] >= 0) claim[q].push_back(all[t]);
      }
      off += cnt[q];
    }
  }
  auto offered = comm_->exchange_lists(claim);   // offered[q]: the gids of my list that rank q owns
  std::vector<std::vector<int64_t>> keys(comm_->size);
  std::vector<ivec> dst(comm_->size);
  int64_t found = 0;
  for (int q = 0; q < comm_->size; q++)
    for (int32_t g : offered[q]) { keys[q].push_back(g); dst[q].push_back(g2l_[g]); found++; }
  HYMLS_CHECK(found == (int64_t)want.size(), -3, "sharded matvec: a column of an owned row has no (or more than one) owner");
  xch_mv_.build(*comm_, keys, dst, [&](int64_t g) { const int l = g2l_[g]; return l >= 0 ? user[l] : -1; });
  d_mv_row_ = dev::upload(row); d_mv_col_ = dev::upload(col); d_mv_src_ = dev::upload(src); d_mv_node_ = dev::upload(node);
  mv_nnz_ = (int64_t)col.size();
  d_mv_val_ = (double*)dev::alloc(std::max<size_t>(1, col.size()) * sizeof(double));
  d_mv_x_ = (double*)dev::alloc((size_t)std::max(n, 1) * sizeof(double));
  mv_ready_ = true;
}

// ------------------------------------------------------------------ bordered systems [K V; W' C]
// (BorderedOperator interface of HYMLS::Preconditioner, reference src/HYMLS_Preconditioner.cpp:519-588,844-918,
// 930-1070 and of the SchurPreconditioner, src/HYMLS_SchurPreconditioner.cpp:631-664,1517-1617)
void LevelSolver::set_border(int m, const double* dV, const double* dW, const double* C) {
  dev::Range range("Preconditioner", level_ + 1, "SetBorder");
  HYMLS_CHECK(initialized_, -1, "SetBorder needs an initialized preconditioner");
  void* ptrs[] = {d_bVu_, d_bWu_, d_bW1_, d_bQ1_, d_bSV_, d_bSW_, d_bNV_, d_bNW_, d_btmp_};
  for (void* q : ptrs) dev::free(q);
  d_bVu_ = d_bWu_ = d_bW1_ = d_bQ1_ = d_bSV_ = d_bSW_ = d_bNV_ = d_bNW_ = d_btmp_ = nullptr;
  bm_ = 0;
  if (m <= 0) return;
  bm_ = m;
  const size_t n = (size_t)(n1_ + n2_);
  d_bVu_ = (double*)dev::alloc(n * m * sizeof(double));
  d_bWu_ = (double*)dev::alloc(n * m * sizeof(double));
  dev::d2d(d_bVu_, dV, n * m * sizeof(double));
  dev::d2d(d_bWu_, dW ? dW : dV, n * m * sizeof(double));
  bC_.assign((size_t)m * m, 0.0);
  if (C) bC_.assign(C, C + (size_t)m * m);
  // (sharded: Q1 columns carry the halo of the neighbours' interior layer behind the n1 local entries)
  d_bW1_ = (double*)dev::alloc(std::max<size_t>(1, (size_t)n1_ * m) * sizeof(double));
  d_bQ1_ = (double*)dev::alloc(std::max<size_t>(1, (size_t)(n1_ + ngi_) * m) * sizeof(double));
  d_bSV_ = (double*)dev::alloc(std::max<size_t>(1, (size_t)n2_ * m) * sizeof(double));
  d_bSW_ = (double*)dev::alloc(std::max<size_t>(1, (size_t)n2_ * m) * sizeof(double));
  d_btmp_ = (double*)dev::alloc(std::max<size_t>(1, (size_t)std::max(n1_ + ngi_, n2_ + ngs_)) * sizeof(double));
  const size_t ng = vs_.size();
  d_bNV_ = (double*)dev::alloc(std::max<size_t>(1, ng * m) * sizeof(double));
  d_bNW_ = (double*)dev::alloc(std::max<size_t>(1, ng * m) * sizeof(double));
  if (!d_a12t_row_) {
    // A12^T (separators x interiors) for W2 - A12' (A11' \ W1)
    const int ns = n2_ + ngs_;
    ivec row(ns + 1, 0), col(a12_col_.size()), src(a12_col_.size());
    for (int32_t c : a12_col_) row[c + 1]++;
    for (int i = 0; i < ns; i++) row[i + 1] += row[i];
    ivec fill(row.begin(), row.end() - 1);
    for (int t = 0; t < n1_; t++)
      for (int e = a12_row_[t]; e < a12_row_[t + 1]; e++) { const int o = fill[a12_col_[e]]++; col[o] = t; src[o] = a12_src_[e]; }
    d_a12t_row_ = dev::upload(row); d_a12t_col_ = dev::upload(col); d_a12t_src_ = dev::upload(src);
    a12t_nnz_ = (int64_t)col.size();
    d_a12t_val_ = (double*)dev::alloc(std::max<size_t>(1, col.size()) * sizeof(double));
    for (auto& cp : cls_) {
      ivec order;
      int32_t rows = 1;
      for (size_t l = 0; l < cp->lu.plan.levels.size(); l++) {
        order.insert(order.end(), cp->lu.plan.levels[l].begin(), cp->lu.plan.levels[l].end());
        order.insert(order.end(), cp->lu.plan.big_levels[l].begin(), cp->lu.plan.big_levels[l].end());
      }
      for (auto& F : cp->lu.plan.fronts) rows = std::max(rows, F.w + F.ri);
      d_orders_.push_back(dev::upload(order));
      order_rows_.push_back(rows);
    }
  }
}

void LevelSolver::interior_solve_transposed(double* x1) {
  for (size_t c = 0; c < cls_.size(); c++) {
    Cls& C = *cls_[c];
    if (C.lu.plan.nI == 0) continue;
    dev::solve_transposed(C.lu.dplan, C.lu.batch, d_orders_[c], (int32_t)C.lu.plan.fronts.size(), order_rows_[c], x1);
  }
}

// ComputeBorder: Q1 = A11 \ V1, border of the Schur system SV = V2 - A21 Q1, SW = W2 - A12' (A11' \ W1),
// SC = C - W1' Q1; the SchurPreconditioner transforms SV and SW with the OT and hands their V-sum rows to the next level
void LevelSolver::compute_border() {
  if (bm_ == 0) return;
  const int m = bm_;
  dev::gather(a12t_nnz_, d_a12t_src_, d_kval_, d_a12t_val_);
  for (int j = 0; j < m; j++) {
    const double* vu = d_bVu_ + (size_t)j * (n1_ + n2_);
    const double* wu = d_bWu_ + (size_t)j * (n1_ + n2_);
    double* q1 = d_bQ1_ + (size_t)j * (n1_ + ngi_);
    double* w1 = d_bW1_ + (size_t)j * n1_;
    double* sv = d_bSV_ + (size_t)j * n2_;
    double* sw = d_bSW_ + (size_t)j * n2_;
    dev::gather(n1_, d_inperm_, vu, q1);
    dev::gather(n2_, d_inperm_ + n1_, vu, sv);
    dev::gather(n1_, d_inperm_, wu, w1);
    dev::gather(n2_, d_inperm_ + n1_, wu, sw);
    interior_solve(q1);                                                               // Q1 = A11 \ V1
    xch_int_.forward(q1, q1);                                                         // (halo of the neighbours' interior layer)
    dev::spmv(n2_, d_a21_row_, d_a21_col_, d_a21_val_, q1, sv, -1.0, 1.0);            // SV = V2 - A21 Q1
    dev::d2d(d_btmp_, w1, (size_t)n1_ * sizeof(double));
    interior_solve_transposed(d_btmp_);                                               // A11' \ W1
    if (!comm_->distributed()) {
      dev::spmv(n2_, d_a12t_row_, d_a12t_col_, d_a12t_val_, d_btmp_, sw, -1.0, 1.0);  // SW = W2 - A12' (...)
    } else {
      // rows of A12' for separators owned elsewhere are partial sums that belong to their owners
      double* t2 = d_t2_;                      // n2 + ngs entries, free during Compute
      dev::spmv(n2_ + ngs_, d_a12t_row_, d_a12t_col_, d_a12t_val_, d_btmp_, t2, 1.0, 0.0);
      xch_sep_.backward(t2, t2, true);
      dev::axpby(n2_, -1.0, t2, 1.0, sw);
    }
  }
  bSC_ = bC_;
  {
    dvec d((size_t)m * m);
    for (int j = 0; j < m; j++)
      for (int i = 0; i < m; i++) d[i + (size_t)m * j] = dev::dot(n1_, d_bW1_ + (size_t)i * n1_, d_bQ1_ + (size_t)j * (n1_ + ngi_));
    comm_->allsum(d);
    for (size_t k = 0; k < d.size(); k++) bSC_[k] -= d[k];
  }
  if (!direct_schur_) {
    const int ng = (int)vs_.size();
    for (int j = 0; j < m; j++) {
      dev::ot_apply(ng, d_gptr_, d_otw_, d_bSV_ + (size_t)j * n2_);
      dev::ot_apply(ng, d_gptr_, d_otw_, d_bSW_ + (size_t)j * n2_);
      dev::gather(ng, d_vs_, d_bSV_ + (size_t)j * n2_, d_bNV_ + (size_t)j * ng);
      dev::gather(ng, d_vs_, d_bSW_ + (size_t)j * n2_, d_bNW_ + (size_t)j * ng);
    }
  }
}

// the next level (or the coarse / direct Schur solver) gets its border before it is computed
void LevelSolver::set_next_border() {
  if (!next_) return;
  if (bm_ == 0) { next_->set_border(0, nullptr, nullptr, nullptr); return; }
  const double* v = direct_schur_ ? d_bSV_ : d_bNV_;
  const double* w = direct_schur_ ? d_bSW_ : d_bNW_;
  if (!comm_->distributed()) { next_->set_border(bm_, v, w, bSC_.data()); return; }
  // sharded: the border rows travel to the next level's layout like every right-hand side (hand-off exchange)
  const size_t nown = (size_t)red_.n, nn = (size_t)std::max<int64_t>(n_next_owned_, 1);
  double* nv = (double*)dev::alloc(nn * bm_ * sizeof(double));
  double* nw = (double*)dev::alloc(nn * bm_ * sizeof(double));
  for (int j = 0; j < bm_; j++) {
    xch_down_.forward(v + nown * j, nv + (size_t)n_next_owned_ * j);
    xch_down_.forward(w + nown * j, nw + (size_t)n_next_owned_ * j);
  }
  next_->set_border(bm_, nv, nw, bSC_.data());
  dev::free(nv); dev::free(nw);
}

void LevelSolver::next_apply_bordered(const double* rhs, const double* T, double* sol, double* S) {
  if (!comm_->distributed()) { next_->apply_inverse_bordered(rhs, T, sol, S); return; }
  ensure_nvec(1);
  xch_down_.forward(rhs, d_nrhs_);
  next_->apply_inverse_bordered(d_nrhs_, T, d_nsol_, S);
  if (next_is_direct_) {
    if (red_.n) dev::d2d(sol, d_nsol_ + glob_row_off_[comm_->rank], (size_t)red_.n * sizeof(double));
  } else {
    xch_down_.backward(d_nsol_, sol);
  }
}

void LevelSolver::schur_apply_bordered(double* rhs2, const double* q, double* x2, double* S) {
  dev::Range range("SchurPreconditioner", level_ + 1, "ApplyInverse (bordered)");
  if (direct_schur_) { next_apply_bordered(rhs2, q, x2, S); return; }
  const int ng = (int)vs_.size(), m = bm_;
  dev::ot_apply(ng, d_gptr_, d_otw_, rhs2);
  dev::zero(x2, (size_t)n2_ * sizeof(double));                          // V-sum entries are zero in W'(M11 \ f1)
  dev::blocks_apply_all(n_blk_apply_, d_blka_, blk_max_nb_, rhs2, x2);
  dvec tc(m);
  for (int j = 0; j < m; j++) tc[j] = dev::dot(n2_, d_bSW_ + (size_t)j * n2_, x2);
  comm_->allsum(tc);
  for (int j = 0; j < m; j++) tc[j] = q[j] - tc[j];
  dev::gather(ng, d_vs_, rhs2, d_vrhs_);
  next_apply_bordered(d_vrhs_, tc.data(), d_vsol_, S);
  dev::scatter(ng, d_vs_, d_vsol_, x2);
  dev::ot_apply(ng, d_gptr_, d_otw_, x2);
}

void LevelSolver::apply_inverse_bordered(const double* b, const double* T, double* x, double* S) {
  dev::Range range("Preconditioner", level_ + 1, "ApplyInverse (bordered)");
  if (bm_ == 0) { apply_inverse(b, x); return; }
  HYMLS_CHECK(next_ != nullptr, -1, "The preconditioner has not yet been computed.");
  const int m = bm_;
  double* z1 = d_z_;
  double* z2 = d_z_ + n1_ + ngi_;
  dev::gather(n1_, d_inperm_, b, z1);
  dev::gather(n2_, d_inperm_ + n1_, b, z2);
  interior_solve(z1);                                                               // x1 = A11 \ b1
  xch_int_.forward(z1, z1);
  dev::spmv(n2_, d_a21_row_, d_a21_col_, d_a21_val_, z1, z2, -1.0, 1.0);            // b2 - A21 x1
  dvec q(m);
  for (int j = 0; j < m; j++) q[j] = dev::dot(n1_, d_bW1_ + (size_t)j * n1_, z1);
  comm_->allsum(q);
  for (int j = 0; j < m; j++) q[j] = T[j] - q[j];                                   // T - W1' x1
  schur_apply_bordered(z2, q.data(), d_t2_, S);
  xch_sep_.forward(d_t2_, d_t2_);
  dev::spmv(n1_, d_a12_row_, d_a12_col_, d_a12_val_, d_t2_, d_t1_, 1.0, 0.0);
  interior_solve(d_t1_);
  dev::axpby(n1_, -1.0, d_t1_, 1.0, z1);                                            // x1 -= A11 \ (A12 x2)
  for (int j = 0; j < m; j++) dev::axpby(n1_, -S[j], d_bQ1_ + (size_t)j * (n1_ + ngi_), 1.0, z1);   // x1 -= Q1 S
  dev::scatter(n1_, d_inperm_, z1, x);
  dev::scatter(n2_, d_inperm_ + n1_, d_t2_, x);
}

void LevelSolver::matvec(const double* x, double* y) {
  dev::Range range("MatrixBlock", level_ + 1, "Apply");
  if (!comm_->distributed()) { dev::spmv(K_.n, d_krow_, d_kcol_, d_kval_, x, y, 1.0, 0.0); return; }
  if (!mv_ready_) build_matvec();
  if (mv_stale_) { dev::gather(mv_nnz_, d_mv_src_, d_kval_, d_mv_val_); mv_stale_ = false; }   // new values since the last Compute
  const int nown = (int)owned_gids_.size();
  dev::scatter(nown, d_mv_node_, x, d_mv_x_);                    // owned entries to their local nodes
  xch_mv_.forward(x, d_mv_x_);                                   // columns owned elsewhere
  dev::spmv(nown, d_mv_row_, d_mv_col_, d_mv_val_, d_mv_x_, y, 1.0, 0.0);
}

void LevelSolver::add_stats(ApplyStats& st, bool as_coarse) const {
  double f = 0, fs = 0, sp = 0, sep = 0, vec = 0;
  for (auto& cp : cls_) {
    f += 2.0 * 8.0 * (double)cp->lu.plan.nnz_factor * (double)cp->lu.members.size();
    fs += 2.0 * (12.0 * (double)cp->lu.plan.nnz_sparse + 24.0 * cp->lu.plan.nI) * (double)cp->lu.members.size();
  }
  sp = 12.0 * (double)(a12_col_.size() + a21_col_.size()) + 4.0 * (n1_ + n2_ + 2);
  if (!direct_schur_) {
    sep = 2.0 * 12.0 * n2_;
    for (auto& B : blocks_) sep += (8.0 * B.nb * B.nb + 4.0 * B.nb) * B.nblk;
  }
  const double N = (double)(n1_ + n2_);
  vec = 8.0 * (4.0 * N + 7.0 * n1_ + 12.0 * n2_);
  for (auto& cp : cls_) {
    st.flops_factor += (double)cp->lu.plan.flops_factor * (double)cp->lu.members.size();
    if (!direct_schur_) st.flops_transform += 4.0 * (double)cp->pat.nS * (double)cp->pat.nS * (double)cp->lu.members.size();
  }
  if (!direct_schur_) for (auto& B : blocks_) st.flops_blocks += 2.0 * (double)B.nb * B.nb * B.nb * B.nblk;
  if (as_coarse) { st.bytes_coarse += f + sp + sep + vec; st.bytes_coarse_sparse += fs + sp + sep + vec; }
  else { st.bytes_factor += f; st.bytes_factor_sparse += fs; st.bytes_spmv += sp; st.bytes_sep += sep; st.bytes_vec += vec; }
  if (next_) next_->add_stats(st, true);
}

}  // namespace hymls
