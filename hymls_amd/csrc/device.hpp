// device.hpp -- the device boundary of the host code: memory, stream, and one launcher
// per kernel.  The product library implements this with HIP (device_hip.hip); the
// host-logic simulator under tests/hostsim implements the same functions with plain
// loops so that the index plans can be validated on a machine without a GPU.
// Nothing in the product links the simulator.
#pragma once
#include <cstdio>
#include "common.hpp"
#include "symbolic.hpp"

namespace hymls {
namespace dev {

// ---- runtime
// per-handle device context (device ordinal, streams, setup arenas, timers, profiling marks).  Every C-ABI entry binds
// the handle's context on the calling thread before it touches the device; nothing device-related is process-global.
struct Context;
Context* create_context(int device);   // throws -3 without a HIP device, -2 for a bad ordinal
void bind(Context* c);                 // make c current on this thread (selects its device); nullptr: unbind
Context* current();                    // the context bound to this thread (nullptr: none)
void destroy_context(Context* c);
void* stream();                        // hipStream_t of the bound context (nullptr in the simulator)
const double* zeros16();               // 16 device zeros (bound context)
void* alloc(size_t bytes);
void free(void* p);
void h2d(void* dst, const void* src, size_t bytes);
void d2h(void* dst, const void* src, size_t bytes);
void d2d(void* dst, const void* src, size_t bytes);
void zero(void* dst, size_t bytes);
void sync();
size_t mem_free();
// process-wide setup scratch (frontal matrices, separator blocks, pivot-piece workspace): factorisations run one
// after the other on the stream, so every batch borrows the same arena; grown on demand (synchronises when it grows)
void* shared_scratch(size_t bytes);
// side streams for independent setup work (the many small batches of the coarser levels would otherwise run one tiny
// launch after the other): fork_streams() makes the side streams wait for the main one, use_stream(k) directs every
// following launch / copy / scratch request to stream k (0 = main, 1..NSIDE), join_streams() makes the main stream wait
// for all of them and switches back to it
constexpr int NSIDE = 8;   // streams that exist; side_streams() of them are used (HYMLS_MI_SIDE_STREAMS, default 4)
inline int side_streams() {
  static const int n = std::getenv("HYMLS_MI_SIDE_STREAMS") ? std::max(1, std::min(NSIDE, std::atoi(std::getenv("HYMLS_MI_SIDE_STREAMS")))) : 4;
  return n;
}
void fork_streams();
void use_stream(int k);
void join_streams();
// named ranges for profilers: roctx ranges (`rocprofv3 --marker-trace`), labelled as the reference labels its timers,
// "<class>_L<level>: <function>" (HYMLS_LPROF, reference src/HYMLS_Macros.hpp:86-137).  The roctx library is looked up at run
// time (librocprofiler-sdk-roctx.so, then libroctx64.so); without it the calls do nothing.  HYMLS_MI_RANGE_LOG=<file>
// additionally appends "push <label>" / "pop" lines to a file (tests).
void range_push(const char* label);
void range_pop();
struct Range {
  Range(const char* cls, int level, const char* fn) {
    char buf[96];
    std::snprintf(buf, sizeof buf, "%s_L%d: %s", cls, level, fn);
    range_push(buf);
  }
  ~Range() { range_pop(); }
  Range(const Range&) = delete;
  Range& operator=(const Range&) = delete;
};
// event timing on the stream (seconds); ids are small integers
void timer_start(int id);
double timer_stop(int id);  // synchronises
// non-intrusive phase profiling: mark() records an event on the stream (no synchronisation);
// profile_collect() synchronises once, adds the elapsed seconds of every begin/end pair to
// sum[phase], counts the pairs in cnt[phase] and clears the log.  phase < 8.
void mark(int phase, bool begin);
void profile_collect(double* sum, int* cnt);

template <class T, class A>
T* upload(const std::vector<T, A>& v) {
  T* p = (T*)alloc(std::max<size_t>(v.size(), 1) * sizeof(T));
  if (!v.empty()) h2d(p, v.data(), v.size() * sizeof(T));
  return p;
}

// ---- device-side plan of one pattern class (POD, arrays live on the device)
struct FrontD {
  int32_t c0, w, ri, rs;
  int32_t parent, idx_off, rel_off, c_off, a_off, lf_off;
  int32_t ent_begin, ent_end, child_begin, child_end;
  int64_t f_off, lp_off, q_off;
};

// forward work item of the fused solve with its assembly sources inline: one 16-byte load replaces the chain
// item -> asm_ptr -> asm_src (n == 0xffff: more than 5 sources, use the lists)
struct alignas(16) FwRec { int32_t item; uint16_t n; uint16_t s[5]; };

struct PlanD {
  int32_t nI, nS, nfronts, nent;   // nent: entries of the extended local CSR
  const FrontD* fronts;
  const int32_t* fidx;
  const int32_t* rel;
  const int32_t* children;
  const int32_t* ent_id;
  const int32_t* ent_pos;
  const double* ent_w;
  const int32_t* asm_ptr;   // [asm_rows + 1]
  const int32_t* asm_src;
  int32_t asm_rows;
  const int32_t* fw_ptr;    // [nlev + 1] level-synchronous fused solve: forward work items
  const int32_t* fw_items;  // front << 16 | row
  const struct FwRec* fw_rec;   // per forward item: the item and its (few) assembly sources inline
  const int32_t* bw_ptr;
  const int32_t* bw_items;
  int32_t nlev, max_level_rows;
  int32_t s_ent_begin, s_ent_end;
  int64_t scratch_size, factor_size;
  int32_t contrib_size;
  int32_t max_solve_rows;   // max over fronts of w + ri (LDS vector of the solve kernels)
  int32_t packed;           // L-side panels of this class are repacked (see repack_fronts)
};

// a batch of subdomains of one class
struct BatchD {
  int32_t nb;
  const int32_t* src;    // [nb][plan.nent] index into the level matrix values
  const int32_t* xoff;   // [nb] offset of the subdomain's interior block in the level vector
  double* factor;        // [nb][factor_size]
  double* scratch;       // [chunk][scratch_size]   frontal matrices
  double* sblock;        // [chunk][nS*nS]          separator (Schur) block, col-major
  double* contrib;       // [nb][contrib_size]      solve scratch
  int32_t* flag;         // device int: set != 0 on zero / non-finite pivot
  double* tmp;           // [chunk][tmp_stride] dense pivot-piece inverses + panel scratch (big fronts)
  int64_t tmp_stride;
  double* swork;         // [nb][swork_stride] solve workspace: assembled rows, then partial sums
  int64_t swork_stride;
};

// ---- vector kernels
void gather(int64_t n, const int32_t* idx, const double* src, double* dst);      // dst[i] = src[idx[i]]
void scatter(int64_t n, const int32_t* idx, const double* src, double* dst);     // dst[idx[i]] = src[i]
void scatter_add(int64_t n, const int32_t* idx, const double* src, double* dst); // dst[idx[i]] += src[i] (atomic: idx may repeat)
void axpby(int64_t n, double a, const double* x, double b, double* y);           // y = a x + b y
void scale_copy(int64_t n, double a, const double* x, double* y);                // y = a x
// y = alpha * A x + beta * y, CSR with 32-bit indices
void spmv(int32_t nrows, const int32_t* rowptr, const int32_t* col, const double* val,
          const double* x, double* y, double alpha, double beta, int64_t nnz_hint = -1);
// out[e] = sum_{t in [ptr[e],ptr[e+1])} in[idx[t]]   (deterministic pull-assembly)
void pull_sum(int64_t n, const int64_t* ptr, const int64_t* idx, const double* in, double* out);
// Off-diagonal blocks of the level matrix (A12, A21) cut out on the device: for the matrix rows rows[0 .. nrows) keep the
// entries whose column c has a target, target(c) = ta[c] if ta[c] >= 0, else tb[c] if tb is given and excl[c] < 0, else none.
// offdiag_count writes the kept entries of row t to count[t + 1] (count[0] untouched); offdiag_fill writes their targets and
// their positions in the level matrix behind rowptr[t].
void offdiag_count(int64_t nrows, const int32_t* rows, const int32_t* krow, const int32_t* kcol, const int32_t* ta, const int32_t* tb,
                   const int32_t* excl, int32_t* count);
void offdiag_fill(int64_t nrows, const int32_t* rows, const int32_t* krow, const int32_t* kcol, const int32_t* ta, const int32_t* tb,
                  const int32_t* excl, const int32_t* rowptr, int32_t* col, int32_t* src);
// entry source lists of the members of a pattern class, built on the device: src[b][q] = index in the level matrix (CSR krow /
// kcol, rows with ascending columns) of entry q of member b's extended local CSR, whose row and column are positions ent_row[q]
// and ent_col[q] in the member's node list ext[b][0 .. next).  *flag |= 1 when an entry is not found.
void member_sources(int32_t nb, int32_t next, int32_t nent, const int32_t* ext, const int32_t* ent_row, const int32_t* ent_col,
                    const int32_t* krow, const int32_t* kcol, int32_t* src, int32_t* flag);
// pull tables of the reduced matrix from its sorted keys (column gid << 33 | source position), rows [rcount[r], rcount[r+1])
// of the key array, entries [rowptr[r], rowptr[r+1]) of the matrix: idx[k] = source of key k, ptr[e] .. ptr[e+1] = the keys
// of entry e (a run of equal column gids)
void build_pull_tables(int64_t nrows, const int64_t* rcount, const int32_t* rowptr, const uint64_t* keys, int64_t* ptr, int64_t* idx);
// out[B*blen + k] = sum_{t in [ptr[B],ptr[B+1])} in[base[t] + k], k < blen  (whole dense blocks)
void pull_sum_blocks(int64_t blen, int32_t nblk, const int64_t* ptr, const int64_t* base,
                     const double* in, double* out);

// ---- multifrontal numeric factorisation, one tree level, fronts [first, first+count) of
// `list` (front ids), batch members [b0, b0+nbc) mapped to scratch slots 0..nbc-1
// max_w: widest pivot block among the listed fronts (selects the LDS size of the kernel)
void factor_level(const PlanD& P, const BatchD& B, const int32_t* list, int32_t count,
                  int32_t b0, int32_t nbc, const double* kval, int32_t max_w);
// separator block: S = weighted A22 entries (call before the tree), per batch member
void sblock_init(const PlanD& P, const BatchD& B, int32_t b0, int32_t nbc, const double* kval);

// one big front (supernode), spread over many workgroups: its pivot block is factored in pieces
// of `piece` columns in place, then inverted explicitly (kids: host copies of its children)
constexpr int PIECE = 128;   // pivot pieces are factored and inverted inside LDS (128 x 128 x 8 B = 128 KiB)
void factor_big_front(const PlanD& P, const BatchD& B, const FrontD& F, const FrontD* kids, int32_t nkids,
                      int32_t b0, int32_t nbc, const double* kval);
// root front F (no interior ancestors): add its update matrix to the separator block of every slot.  One launch per
// root, in a fixed order on the stream: the sum is formed in the same order on every Compute (bitwise reproducible)
void root_update(const PlanD& P, const BatchD& B, const FrontD& F, int32_t nbc);
// all big fronts of one tree level (device list of front ids, host copies for the grid sizes)
// poff: device array (per listed front) of offsets into the partial-sum area of the workspace
// columns per workgroup tile of the big-front panel products: 1024, or 256 (HYMLS_MI_SOLVE_KT; more workgroups for the
// fronts of a few thousand columns near the root of one large system)
inline int solve_kt() {
  // (measured on configs[1], 128^3 2-level, 216 k-unknown coarse system: 2.36 ms with 256 against 2.85 ms with 1024)
  static const int kt = (std::getenv("HYMLS_MI_SOLVE_KT") && std::atoi(std::getenv("HYMLS_MI_SOLVE_KT")) == 1024) ? 1024 : 256;
  return kt;
}
void solve_fwd_big(const PlanD& P, const BatchD& B, const int32_t* list, const FrontD* hfronts, const int64_t* poff,
                   int32_t count, double* x);
void solve_bwd_big(const PlanD& P, const BatchD& B, const int32_t* list, const FrontD* hfronts, const int64_t* poff,
                   int32_t count, double* x);

// ---- solves with the factor panels; x is the level vector (interior part), in place
void solve_fwd_level(const PlanD& P, const BatchD& B, const int32_t* list, int32_t count, double* x);
void solve_bwd_level(const PlanD& P, const BatchD& B, const int32_t* list, int32_t count, double* x);

// ---- packed panels (classes solved by the fused kernel).  The factorisation leaves per front the
// column-major (w+ri) x w panel [L11^{-1} strictly lower \ U11^{-1} upper ; L21 L11^{-1}]; a sweep only ever
// needs one of the two triangles, and reading a triangle out of the tall columns drags the other one's
// cache lines along.  repack_fronts rearranges the same (w+ri) w entries in place into three contiguous
// pieces, so that each sweep streams only bytes it uses:
//   [ strictly lower triangle, packed by columns : w(w-1)/2 | L21 L11^{-1}, ri x w, ld = ri | upper
//     triangle incl. diagonal, packed by columns : w(w+1)/2 ]            (the U-side panel Q stays where it is)
#if defined(__HIPCC__)
#define HYMLS_HD __host__ __device__
#else
#define HYMLS_HD
#endif
HYMLS_HD inline int64_t packed_lower(int64_t w, int64_t i, int64_t k) { return k * (2 * w - k - 1) / 2 + (i - k - 1); }   // i > k
HYMLS_HD inline int64_t packed_l21(int64_t w, int64_t ri, int64_t i, int64_t k) { return w * (w - 1) / 2 + i + ri * k; }      // row i of L21
HYMLS_HD inline int64_t packed_upper(int64_t w, int64_t ri, int64_t i, int64_t k) { return w * (w - 1) / 2 + ri * w + k * (k + 1) / 2 + i; }  // i <= k
// members [b0, b0+nbc) of the batch; uses the frontal scratch of the chunk as temporary
void repack_fronts(const PlanD& P, const BatchD& B, int32_t b0, int32_t nbc);

// ---- fused interior solve: one workgroup per subdomain walks its whole assembly tree
// (forward then backward) with the solution vector and all contribution vectors in LDS;
// one launch covers every subdomain of every pattern class of a level.
struct FusedSub {
  const double* fac;   // factor slab of this subdomain
  int32_t xoff;        // offset of its interior block in the level vector
  int32_t cls;         // index into the PlanD table
};
constexpr int FUSED_MAX_ITEMS = 2048;  // max work items (rows) of one tree level handled by the fused kernel
// optional fusion of the neighbouring vector kernels into the load / store of the fused solve:
//   in  = 0: right-hand side = x (in place)      1: x_rhs[i] = b[perm[i]] (the entry gather of ApplyInverse)
//         2: x_rhs[i] = (A x2)[i], A in CSR over the interior rows (y1 = A12 x2 of the second solve)
//   out = 0: x[i] = solution                     1: user[perm[i]] = z[i] - solution (x1 -= A11 \ y1 and the exit scatter)
struct FusedIO {
  int32_t in = 0, out = 0;
  const double* b = nullptr; const int32_t* perm = nullptr;
  const int32_t* a_row = nullptr; const int32_t* a_col = nullptr; const double* a_val = nullptr; const double* x2 = nullptr;
  const double* z = nullptr; double* user = nullptr;
};
void interior_solve_fused(int32_t nsub, const FusedSub* subs, const PlanD* plans, int32_t lds_doubles, double* x,
                          const FusedIO* io = nullptr);

// ---- merged level-synchronous solve of the classes that do not fit the fused kernel (large subdomains
// of the coarser levels): ONE launch per tree level and sweep covers every (class, member, front) of that
// level.  A task is one workgroup: a whole small front, or a 64-row tile of a large one (all its columns;
// the assembly of the pivot entries it needs is fused in, which is why the forward sweep reads x and
// writes y instead of working in place).
struct LvlSub { const double* fac; double* contrib; int32_t xoff, cls; int64_t cstride; };   // one (class, member); cstride: distance of
                                                                                             // the contribution vectors of two columns
struct LvlTask { int32_t sub, front, r0, pad; };   // r0 < 0: whole front; else rows [r0, r0 + 64)
constexpr int LVL_MAX_ROWS = 6144;                 // w + ri limit of a front on this path (LDS vector).  Measured: with wider
                                                   // fronts (one 216 k-unknown system, root front 7 k wide) a 64-row tile task
                                                   // streams megabytes through ONE workgroup and the level takes longer than
                                                   // the column-split panel kernels of the big-front path, launch chain included
constexpr int LVL_SMALL_ROWS = 256;                // fronts up to this many rows are one task
// y[pivots] = forward-substituted values, contributions pushed to the parents' assembly
void solve_fwd_tasks(const LvlTask* tasks, int32_t ntasks, const LvlSub* subs, const PlanD* plans, int32_t lds_doubles,
                     const double* x, double* y);
// x[pivots] = U^{-1} y - (U^{-1} U12) x[ancestors]
void solve_bwd_tasks(const LvlTask* tasks, int32_t ntasks, const LvlSub* subs, const PlanD* plans, int32_t lds_doubles,
                     const double* y, double* x);

// ---- several right-hand sides (column-major multivectors, leading dimension ld, nv columns): the factor panels are
// streamed once for groups of up to 4 columns (the single-vector kernels with the per-row state replicated)
constexpr int NV_MAX = 4;
// lds_doubles: LDS need for one vector, front_doubles: the share of it that is not replicated per vector
void interior_solve_fused_mv(int32_t nsub, const FusedSub* subs, const PlanD* plans, int32_t lds_doubles, int32_t front_doubles,
                             double* x, int64_t ldx, int nv);
void solve_fwd_tasks_mv(const LvlTask* tasks, int32_t ntasks, const LvlSub* subs, const PlanD* plans, int32_t lds_doubles,
                        const double* x, double* y, int64_t ld, int nv);
void solve_bwd_tasks_mv(const LvlTask* tasks, int32_t ntasks, const LvlSub* subs, const PlanD* plans, int32_t lds_doubles,
                        const double* y, double* x, int64_t ld, int nv);

// ---- separator-side kernels
// Householder per owned group on a level separator vector: x <- 2 w (w.x) - x
// gptr[ng+1] offsets into the separator vector; w = 0 rows mean "x <- -x" (reference quirk)
void ot_apply(int32_t ng, const int32_t* gptr, const double* w, double* x);
// two-sided Householder of every group on the dense separator blocks of a chunk:
// S_b <- H S_b H ; gptr: local group offsets (shared by the class), tv: [nbc][nS] test vector
void sblock_transform(int32_t nS, int32_t ng, const int32_t* gptr, const double* tv,
                      double* sblock, int32_t nbc);
// out[b][k] = sblock[b][pick[k]]  (extraction of the kept entries)
void sblock_extract(int32_t nS, int64_t npick, const int32_t* pick, const double* sblock,
                    double* out, int64_t out_stride, int32_t nbc);
// transform + extraction in one read pass (device_hip.hip: k_sblock_kept): the record of every slot = V-sum x V-sum
// entries (ngl x ngl, column-major) followed by the non-V-sum block of every linked set
struct KeptD {
  int32_t nS, ngl;
  const int32_t* gptr; const int32_t* glink; const int32_t* goff; const int64_t* lboff; const int32_t* lblen;
};
void sblock_kept(const KeptD& K, const double* tv, const double* sblock, double* out, int64_t out_stride, int32_t nbc);
// false when the three separator-length vectors of the fused kernel exceed the LDS of a workgroup (separator blocks of order
// 6000+: separator length 32): the caller then takes the two-pass route, sblock_transform + sblock_extract
bool sblock_kept_fits(int32_t nS, int32_t ngl);
// batched dense inverse with partial pivoting: nblk blocks of order nb (col-major), in place
void dense_invert(int32_t nb, int32_t nblk, double* blocks, int32_t* flag);
// true where dense_invert() takes the blocked (32 pivots at a time, matrix-core update) route: such groups are worth a
// call of their own instead of a slot in the single-launch table of dense_invert_all()
bool dense_invert_blocked_order(int32_t nb);
// y[ids] = Binv x[ids] for nblk blocks of order nb; ids: [nblk][nb]
void blocks_apply(int32_t nb, int32_t nblk, const double* binv, const int32_t* ids,
                  const double* x, double* y);

// ---- bordered systems (setup-time helpers; not on the ApplyInverse fast path)
// x . y with a fixed-order two-stage reduction; synchronises the stream and returns the value to the host
double dot(int64_t n, const double* x, const double* y);
// x <- A11^{-T} x for every member of the batch (transposed solve with the same panels: forward with U^T,
// backward with L^T); order[nfronts]: fronts in elimination order (children before parents); one workgroup per
// member walks the tree sequentially -- used m times per Compute for the border W
void solve_transposed(const PlanD& P, const BatchD& B, const int32_t* order, int32_t nfronts, int32_t max_rows, double* x);

// all separator blocks of a level in one launch (blocks of any order; heavy ones first)
struct BlkD { const double* binv; const int32_t* ids; int32_t nb, r0; };   // r0 < 0: all rows; else rows [r0, r0 + 64) (tiles of a large block)
void blocks_apply_all(int32_t nblk, const BlkD* blocks, int32_t max_nb, const double* x, double* y);
void blocks_apply_all_mv(int32_t nblk, const BlkD* blocks, int32_t max_nb, const double* x, int64_t ldx, double* y, int64_t ldy, int nv);
// in-place inverse (partial pivoting) of every block of the table, one launch
void dense_invert_all(int32_t nblk, const BlkD* blocks, int32_t max_nb, int32_t* flag);

}  // namespace dev
}  // namespace hymls
