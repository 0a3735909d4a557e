// precond.hpp -- host orchestration of the multilevel preconditioner on one GPU.
//
// LevelSolver is one HYMLS::Preconditioner level (reference src/HYMLS_Preconditioner.cpp)
// together with its SchurPreconditioner (src/HYMLS_SchurPreconditioner.cpp); DirectSolver is
// the CoarseSolver (src/HYMLS_CoarseSolver.cpp) and also the engine behind every batched
// interior solve.  All integer work happens here on the host, once; all floating point
// work of Compute/ApplyInverse runs in the kernels behind device.hpp.
#pragma once
#include <memory>
#include <thread>
#include <exception>
#include "common.hpp"
#include "partition.hpp"
#include "symbolic.hpp"
#include "device.hpp"
#include "comm.hpp"

namespace hymls {

struct ApplyStats {
  double bytes_factor = 0, bytes_spmv = 0, bytes_sep = 0, bytes_coarse = 0, bytes_vec = 0;
  // the same with every factor counted as a sparse solver would stream it: nnz(L + U) x (8 B value + 4 B index) +
  // 24 B of permutation / scaling data per unknown (reference KluSolve, src/HYMLS_SparseDirectSolver.cpp:788-856)
  double bytes_factor_sparse = 0, bytes_coarse_sparse = 0;
  // floating point operations of one numeric Compute (all levels), counted from the symbolic plans:
  // flops_factor: multifrontal LU of every subdomain + last-level solver, per front 2/3 w^3 + 2 w^2 r + 2 w r^2
  // (r = rows below the pivot block incl. the separator rows) + 2/3 w^3 for the two triangular inverses;
  // flops_blocks: inversion of the separator blocks, 2 nb^3 per block (dgetrf + dgetri in the reference, K10);
  // flops_transform: orthogonal transformation + dropping of the separator blocks, 4 nS^2 per subdomain (one fused pass,
  // K8; the reference's two-sided Householder per group costs 4 nS^2 x #groups)
  double flops_factor = 0, flops_blocks = 0, flops_transform = 0;
};

// A device allocation made on a helper thread while the setup thread goes on: a hipMalloc of tens of GiB takes about 30 ms per
// GiB (the driver clears the memory) -- 1.2 s for the factors and the extraction buffer of a 256^3 run, which now passes
// while the host builds its tables.
struct AsyncAlloc {
  std::thread th;
  void* p = nullptr;
  size_t bytes = 0;
  std::exception_ptr err;
  explicit AsyncAlloc(size_t n) : bytes(n) {
    dev::Context* c = dev::current();
    th = std::thread([this, c] {
      try { dev::bind(c); p = dev::alloc(bytes); dev::bind(nullptr); } catch (...) { err = std::current_exception(); }
    });
  }
  void* take() {      // the allocation (the caller owns it from here on)
    if (th.joinable()) th.join();
    if (err) std::rethrow_exception(err);
    void* q = p;
    p = nullptr;
    return q;
  }
  ~AsyncAlloc() {
    if (th.joinable()) th.join();
    if (p) dev::free(p);
  }
  AsyncAlloc(const AsyncAlloc&) = delete;
  AsyncAlloc& operator=(const AsyncAlloc&) = delete;
};

// batched multifrontal LU of one pattern class, resident on the device
struct BatchedLU {
  std::unique_ptr<AsyncAlloc> pre_factor;   // factor storage requested ahead of upload() (see AsyncAlloc)
  static size_t factor_bytes(int64_t nb, int64_t factor_size) { return (size_t)std::max<int64_t>(1, nb * factor_size) * sizeof(double); }
  ClassPlan plan;
  ivec members;                 // caller-defined ids (level subdomain ids)
  ivec h_xoff;
  rawvec<int32_t> h_src;        // [nb][nent] entry of the level matrix behind every entry of the extended local CSR ...
  // ... or, when the rows of the level matrix have ascending columns, what the device needs to find them itself
  // (dev::member_sources): the nodes of every member in the order of the extended local numbering, and row / column of
  // every entry of the class's extended CSR
  rawvec<int32_t> h_ext;        // [nb][n_ext]
  int32_t n_ext = 0;
  ivec h_ent_row, h_ent_col;    // [nent]
  const int32_t *d_krow = nullptr, *d_kcol = nullptr;   // the level matrix on the device (not owned)
  // device
  dev::PlanD dplan{};
  dev::BatchD batch{};
  std::vector<int32_t*> d_lists;  // per tree level (solve)
  std::vector<int32_t*> d_flists; // per tree level (factorisation: the fronts one workgroup factors)
  std::vector<dev::FrontD> h_fronts;  // host copies (grid setup of the big-front kernels)
  std::vector<int32_t*> d_big_lists;  // per tree level: ids of the big fronts
  std::vector<std::vector<dev::FrontD>> h_big_fronts;
  std::vector<int64_t*> d_big_poff;   // per level: partial-sum offsets of the big fronts
  std::vector<dev::FrontD> kids_of(int s) const;
  std::vector<void*> owned;       // device allocations to free
  int32_t nent = 0;
  int32_t chunk = 0;              // members factored per pass
  bool packed = false;            // panels repacked after the factorisation (classes solved by the fused kernel)
  int contrib_nv = 1;             // columns of contribution scratch (several right-hand sides in the task kernels)
  ~BatchedLU();
  void plan_scratch(int64_t scratch_budget_doubles, bool with_sblock);
  void upload(int64_t scratch_budget_doubles, bool with_sblock);
  // numeric factorisation of members [b0,b0+nbc) (scratch slots 0..nbc-1)
  void factor_chunk(const double* kval, int32_t b0, int32_t nbc, bool spread_wide = false);
  void repack_chunk(int32_t b0, int32_t nbc);
  void bind_scratch();            // point batch.scratch / sblock / tmp into the shared setup arena
  int64_t scratch_need_ = 0, sblock_need_ = 0, tmp_need_ = 0;   // doubles   // after factor_chunk and after the separator block was read
  void solve(double* x) const;    // forward + backward, all members, in place
  int32_t check_flag(double* growth = nullptr) const;   // flag bits (1 zero pivot, 2 growth) and the largest growth factor
};

// tables of the merged level-synchronous solve (device.hpp: solve_fwd_tasks / solve_bwd_tasks) for a set of batches
struct MergedSolve {
  dev::LvlSub* d_subs = nullptr;
  dev::LvlTask *d_fw = nullptr, *d_bw = nullptr;
  int32_t nsubs = 0;
  ivec fw_off, bw_off, fw_lds, bw_lds;
  MergedSolve() = default;
  MergedSolve(const MergedSolve&) = delete;
  MergedSolve& operator=(const MergedSolve&) = delete;
  ~MergedSolve();
  // (index of the batch's plan in the PlanD table handed to solve(), batch)
  void build(const std::vector<std::pair<int32_t, const BatchedLU*>>& classes);
  void solve(const dev::PlanD* d_plans, double* x, double* y, int64_t ld, int nv) const;
  int max_nv = 1;   // widest column group the contribution scratch of the batches holds
};
bool merged_solve_fits(const ClassPlan& plan);   // every front within the LDS limits of the task kernels

class Operator {  // something with ApplyInverse on device vectors in its own row ordering
 public:
  virtual ~Operator() {}
  virtual void apply_inverse(const double* b, double* x) = 0;
  // nv right-hand sides, column-major with leading dimensions (Epetra_MultiVector); default: column by column
  virtual void apply_inverse_mv(const double* b, int64_t ldb, double* x, int64_t ldx, int nv) {
    for (int v = 0; v < nv; v++) apply_inverse(b + v * ldb, x + v * ldx);
  }
  // BorderedOperator (reference src/HYMLS_BorderedOperator.hpp): [K V; W' C]; V, W: device arrays (size() x m,
  // column-major, the operator's vector layout), C: host m x m column-major; m = 0 removes the border.  Takes effect
  // with the next compute of the owner.  apply_inverse_bordered: T and S are host vectors of length m.
  virtual void set_border(int m, const double* dV, const double* dW, const double* C) = 0;
  virtual void apply_inverse_bordered(const double* b, const double* T, double* x, double* S) = 0;
  virtual int64_t size() const = 0;
  virtual void add_stats(ApplyStats& st, bool as_coarse) const = 0;
};

// CoarseSolver: exact sparse LU of one matrix (after dropping / Dirichlet fixes)
class DirectSolver : public Operator {
 public:
  // clu_ptr/clu/clu_coord: optional clusters of the rows (see LocalPattern::clu)
  // border_pending: a border will be set (set_border) before the first solve.  If no Dirichlet node is configured the
  // matrix may be singular with its null space only fixed by the border (Stokes without "Fix Pressure Level"): then one
  // pressure row/column is moved into the border ("tail") and the remaining nonsingular matrix is factored.
  DirectSolver(const Csr& A, const ivec& gids, const ivec& fix_gids, int64_t ngid, const Params& coord_params,
               const ivec* clu_ptr = nullptr, const ivec* clu = nullptr, const ivec* clu_coord = nullptr,
               bool border_pending = false);
  ~DirectSolver() override;
  // numeric factorisation only, if A0 (same gids, fixes, border state) still has the pattern this solver was analysed for;
  // false (nothing changed) otherwise: the caller builds a new solver then
  bool refactor(const Csr& A0, const ivec& gids, const ivec& fix_gids, const Params& coord_params, bool border_pending);
  void apply_inverse(const double* b, double* x) override;
  void apply_inverse_mv(const double* b, int64_t ldb, double* x, int64_t ldx, int nv) override;
  void set_border(int m, const double* dV, const double* dW, const double* C) override;
  void apply_inverse_bordered(const double* b, const double* T, double* x, double* S) override;
  int64_t size() const override { return n_; }
  void add_stats(ApplyStats& st, bool as_coarse) const override;

 private:
  void solve(const double* b, double* x, bool zero_fixed);
  void solve_mv(const double* b, int64_t ldb, double* x, int64_t ldx, int nv, bool zero_fixed);
  Csr prepare(const Csr& A0, const ivec& gids, const ivec& fix_gids, const Params& cp, bool border_pending, ivec& fix_rows);
  void numeric(const vvec& val);
  ivec pat_rowptr_, pat_gids_, pat_fix_;
  cvec pat_col_;   // what the plan was built for
  std::vector<char> pat_zero_diag_;
  bool border_pending_ = false;
  int32_t n_ = 0;
  int label_level_ = 1;   // level in the reference's numbering (labels of the profiler ranges)
  int nv_alloc_ = 0;
  // border: x = A^{-1} b - Z y, y = (C - W' Z)^{-1} (T - W' A^{-1} b), Z = A^{-1} V
  int bm_ = 0;
  double *d_bZ_ = nullptr, *d_bW_ = nullptr;
  dvec bMinv_;
  // tail: the rows/columns tail_z_ of the matrix live in the border (see prepare): n x tl columns / rows, tl x tl block
  ivec tail_z_;
  dvec tail_col_, tail_row_, tail_d_;
  std::unique_ptr<BatchedLU> lu_;
  MergedSolve merged_;
  dev::PlanD* d_plan_ = nullptr;
  double* d_y_ = nullptr;
  double* d_val_ = nullptr;
  double* d_z_ = nullptr;
  int32_t* d_perm_ = nullptr;   // elimination position -> row
  ivec fix_lids_;
  int32_t* d_fix_ = nullptr;
};

// local CSR of a sharded level: `row_gids` rows with global column ids become a square local matrix over
// the nodes [rows in the given order | ghost columns in ascending gid order] (ghost nodes have empty rows)
void make_local_csr(int64_t nrows, const int32_t* row_gids, const int32_t* rowptr, const int32_t* col_gids,
                    const double* val, int64_t ngid, Csr& K, ivec& gids);

class LevelSolver : public Operator {
 public:
  LevelSolver(const Params& p, int level, int64_t ngid, const Comm* comm);
  ~LevelSolver() override;
  // subdomains of this rank (+ halo); level_gids: the gids that exist on this level (nullptr: all)
  void partition(const ivec* level_gids);
  ivec required_gids() const;           // rows this rank must be given: interiors + separators of its subdomains
  // K over local nodes: the first nrows nodes have rows (any superset of required_gids()), the others are
  // ghost columns; on one rank simply the whole matrix
  void set_rows(const Csr& K, const ivec& gids, const dvec& tv, int32_t nrows);
  void stream_plan(bool& side, int& chunk_streams) const;
  void prepare_reduced_pattern(const dvec* tvn);
  void next_test_vector(dvec& tvn) const;
  void initialize();
  void compute();                       // uses the host values of K (uploads them)
  void set_values(const vvec& val);     // SetMatrix with unchanged pattern
  void swap_values(vvec& val);          // the same without a copy (one rank): takes the array, hands back the old one
  // b, x: this rank's owned rows (interiors of its subdomains + separators it owns) in the order of owned_gids()
  void apply_inverse(const double* b, double* x) override;
  void apply_inverse_mv(const double* b, int64_t ldb, double* x, int64_t ldx, int nv) override;
  void set_border(int m, const double* dV, const double* dW, const double* C) override;
  void apply_inverse_bordered(const double* b, const double* T, double* x, double* S) override;
  bool have_border() const { return bm_ > 0; }
  int border_size() const { return bm_; }
  int64_t size() const override { return global_n_; }
  void add_stats(ApplyStats& st, bool as_coarse) const override;
  // y = K x on the rows this rank owns (vectors in the layout of apply_inverse); sharded: collective, the values
  // of the columns owned elsewhere are imported first (the Epetra_CrsMatrix::Apply of the reference's Krylov loop)
  void matvec(const double* x, double* y);

  // introspection
  const HierMap& hiermap() const { return hm_; }
  int level() const { return level_; }
  int64_t schur_size() const { return global_n2_; }
  int64_t num_subdomains_global() const { return (int64_t)hm_.sd.size(); }
  const ivec& owned_gids() const { return owned_gids_; }
  int64_t num_owned() const { return (int64_t)owned_gids_.size(); }
  int32_t num_rows() const { return nrows_; }
  Operator* next() const { return next_.get(); }
  LevelSolver* next_level() const { return next_level_; }
  double phase_seconds[5] = {0, 0, 0, 0, 0};
  bool profiling = false;

 private:
  struct Cls;
  struct ExtLayout { int32_t ngl = 0; std::vector<int64_t> blk_off; ivec blk_len; int64_t size = 0; };
  ExtLayout ext_layout(const Subdomain& S) const;
  void localize();
  void build_classes();
  void build_schur_setup();
  void exchange_records();
  const Csr& assemble_reduced(ivec& row_gids, dvec* tvn);
  void schur_apply(double* rhs2, int64_t ldr, double* x2, int64_t ldx, int nv);
  void next_apply(const double* rhs, double* sol, int64_t ld, int nv);
  void interior_solve_mv(double* x1, int64_t ld, int nv);
  void ensure_nvec(int nv);
  int nvec_alloc_ = 1;
  void build_handoff(const ivec& next_owned);
  void interior_solve(double* x1);
  void interior_solve_transposed(double* x1);
  void compute_border();
  void set_next_border();
  void schur_apply_bordered(double* rhs2, const double* q, double* x2, double* S);
  void next_apply_bordered(const double* rhs, const double* T, double* sol, double* S);
  int64_t n_next_owned_ = 0;   // rows of the next level's layout on this rank (sharded hand-off)
  // border (one rank only): user-layout copies, their interior / separator parts, A11^{-1} V1, transformed Schur border
  int bm_ = 0;
  dvec bC_, bSC_;
  double *d_bVu_ = nullptr, *d_bWu_ = nullptr;
  double *d_bW1_ = nullptr, *d_bQ1_ = nullptr, *d_bSV_ = nullptr, *d_bSW_ = nullptr, *d_bNV_ = nullptr, *d_bNW_ = nullptr, *d_btmp_ = nullptr;
  int32_t *d_a12t_row_ = nullptr, *d_a12t_col_ = nullptr, *d_a12t_src_ = nullptr;
  double* d_a12t_val_ = nullptr;
  int64_t a12t_nnz_ = 0;
  std::vector<int32_t*> d_orders_;     // per class: fronts in elimination order
  ivec order_rows_;                    // per class: max w + ri

  Params p_;
  int level_;
  int64_t ngid_;
  const Comm* comm_;
  Csr K_;
  int32_t nrows_ = 0;        // local nodes with a row
  ivec keep_entries_;        // sharded: entries of the rows as given that were kept (set_values)
  size_t given_nnz_ = 0;
  ivec gids_;
  dvec tv_;
  HierMap hm_;
  bool partitioned_ = false;
  ivec sd_rank_;             // per subdomain: owning rank
  ivec my_sds_, halo_sds_;   // ascending
  int64_t global_n_ = 0, global_n2_ = 0;
  int32_t n1_ = 0, n2_ = 0, ngs_ = 0, ngi_ = 0;   // interiors, owned separators, ghost separators, ghost interiors
  ivec g2l_;                 // gid -> local node (-1)
  ivec sep_row_;             // separator index (owned, then ghost) -> local node
  ivec pos2_;                // local node -> separator index (-1)
  ivec intidx_;              // local node -> internal interior index (-1)
  ivec in_perm_;             // internal index (interiors, owned separators) -> position in the user vector
  ivec owned_gids_;
  ivec sd_xoff_, sd_cls_, sd_bidx_;
  ivec sep_sd_ptr_, sep_sd_;   // per separator index: the subdomains whose groups contain it
  ivec sd_center_;             // 3 ints per subdomain (mean coordinate of its separator nodes)
  std::vector<std::unique_ptr<Cls>> cls_;
  // halo exchanges of the apply (empty on one rank)
  Exchange xch_int_, xch_sep_, xch_down_, xch_mv_;
  // sharded K x: owned rows in user order over the local nodes, built on first use
  bool mv_ready_ = false, mv_stale_ = true;
  int32_t *d_mv_row_ = nullptr, *d_mv_col_ = nullptr, *d_mv_src_ = nullptr, *d_mv_node_ = nullptr;
  double *d_mv_val_ = nullptr, *d_mv_x_ = nullptr;
  int64_t mv_nnz_ = 0;
  void build_matvec();
  // records of neighbouring ranks' subdomains that touch separators owned here
  struct RecSeg { int64_t off, len; };
  std::vector<RecSeg> rec_send_;              // in send order (peer-major)
  std::vector<int64_t> rec_scnt_, rec_rcnt_;
  int64_t rec_nsend_ = 0, rec_nrecv_ = 0, ext_recv_base_ = 0;
  bool rec_any_ = false;
  // fused interior solve tables
  dev::FusedSub* d_fsubs_ = nullptr;
  dev::PlanD* d_fplans_ = nullptr;
  int32_t n_fsubs_ = 0, fused_lds_ = 0, fused_front_lds_ = 0, fused_vec_lds_ = 0;
  std::vector<char> cls_fused_;
  // merged level solve tables (classes too large for the fused kernel)
  std::vector<char> cls_merged_;
  MergedSolve merged_;
  double* d_ytmp_ = nullptr;
  // device
  double* d_kval_ = nullptr;
  int32_t *d_krow_ = nullptr, *d_kcol_ = nullptr;
  int32_t* d_inperm_ = nullptr;
  double *d_z_ = nullptr, *d_t1_ = nullptr, *d_t2_ = nullptr, *d_y2_ = nullptr;
  // A12 / A21
  ivec a12_row_, a21_row_;
  cvec a12_col_, a12_src_, a21_col_, a21_src_;   // (host copies of the device tables)
  int32_t *d_a12_row_ = nullptr, *d_a12_col_ = nullptr, *d_a12_src_ = nullptr;
  int32_t *d_a21_row_ = nullptr, *d_a21_col_ = nullptr, *d_a21_src_ = nullptr;
  double *d_a12_val_ = nullptr, *d_a21_val_ = nullptr;
  // Schur preconditioner data
  bool direct_schur_ = false;
  ivec gptr_;                // owned groups: offsets into the separator vector
  dvec otw_;                 // Householder rows (n2)
  ivec vs_;                  // V-sum separator indices (one per owned group)
  int32_t* d_gptr_ = nullptr; double* d_otw_ = nullptr; int32_t* d_vs_ = nullptr;
  struct BlockClass { int32_t nb = 0, nblk = 0; ivec ids; ivec owner_key; double* d_binv = nullptr; int32_t* d_ids = nullptr;
                      std::vector<int64_t> pull_ptr, pull_base; int64_t* d_pull_ptr = nullptr; int64_t* d_pull_base = nullptr; };
  std::vector<BlockClass> blocks_;
  dev::BlkD* d_blkd_ = nullptr;   // all blocks (single-launch inversion)
  dev::BlkD* d_blka_ = nullptr;   // apply tasks: small blocks whole, large blocks in 64-row tiles
  int32_t n_blk_ = 0, n_blk_apply_ = 0, blk_max_nb_ = 0, blk_max_nb_inv_ = 0;   // (_inv_: largest order in the inversion table)
  // rows of the reduced (V-sum) matrix or of the full Schur complement owned here: pattern + pull lists
  Csr red_;                  // col = gid of the column node
  int64_t n_pulls_ = 0;   // summands of the reduced matrix (length of the device-side pull index table)
  int64_t *d_red_pull_ptr_ = nullptr, *d_red_pull_idx_ = nullptr;
  double* d_red_val_ = nullptr;
  double* d_ext_ = nullptr; int64_t ext_total_ = 0;
  double *d_vrhs_ = nullptr, *d_vsol_ = nullptr, *d_yb_ = nullptr;
  double *d_nrhs_ = nullptr, *d_nsol_ = nullptr;   // next level's right-hand side / solution in its own layout
  int32_t* d_flag_ = nullptr;
  // the reduced matrix of all ranks (rows in rank order), rebuilt every Compute from the gathered values
  Csr glob_;
  ivec glob_clu_ptr_, glob_clu_, glob_sd_center_;   // clusters (subdomains) of the rows of glob_, centres of all subdomains
  ivec glob_gids_;
  dvec glob_tv_;
  rawvec<int64_t> glob_perm_;           // gathered entry -> entry of glob_
  std::vector<int64_t> glob_row_off_;   // first global row of every rank
  bool glob_ready_ = false;
  Csr next_R_;                          // reduced matrix after DropByValue (kept: a recompute reuses its arrays)
  std::unique_ptr<Operator> next_;
  LevelSolver* next_level_ = nullptr;
  bool next_is_direct_ = false;
  ivec next_pattern_key_rowptr_;
  cvec next_pattern_key_col_;
  bool initialized_ = false;
};

// MatrixUtils::DropByValue (reference src/HYMLS_MatrixUtils.cpp:1011-1212); kind:
// 0 RelDropDiag, 1 RelZeroDiag, 2 RelFullDiag
Csr drop_by_value(const Csr& A, double tol, int kind);
void drop_by_value(const Csr& A, double tol, int kind, Csr& R);   // into an existing matrix (arrays reused)

}  // namespace hymls
