// symbolic.hpp -- host-side analysis of one "pattern class" of local problems.
//
// A local problem is the extended matrix of one subdomain: nI interior unknowns
// (eliminated) followed by nS separator unknowns around the subdomain (kept).
// Thousands of subdomains share one sparsity pattern, so ordering, supernodes,
// the assembly tree and every index map are computed ONCE per class here and
// the GPU runs the same plan for the whole batch (multifrontal, no pivoting).
//
// This replaces, for the GPU, what the reference does per subdomain with
// SparseDirectSolver::Initialize/Compute (src/HYMLS_SparseDirectSolver.cpp:300-418),
// MatrixUtils::FillReducingOrdering (src/HYMLS_MatrixUtils.cpp:1311-1755: V-nodes
// ordered on the graph of A+BB', every P-node eliminated right after a V-node
// that still connects it to an uneliminated pressure or to the boundary) and,
// through the kept separator rows, SchurComplement::Construct11
// (src/HYMLS_SchurComplement.cpp:131-256): the update matrix that reaches the
// separator block IS  -A21 A11^{-1} A12.
#pragma once
#include "common.hpp"

namespace hymls {

struct LocalPattern {
  int32_t nI = 0, nS = 0;
  ivec rowptr, col;          // extended local CSR pattern, n = nI + nS rows
  std::vector<char> zero_diag;  // size nI: row has no / a zero diagonal ("P-node")
  ivec coord;                // 3 * nI integer coordinates (for nested dissection)
  dvec weight;               // per entry: 1, or 1/multiplicity for separator-separator entries
  // optional clusters (V-sum graphs: the finer-level subdomains every node belongs to).  Two nodes
  // are adjacent only if they share a cluster, so a vertex separator = nodes whose clusters lie
  // on both sides of a cut through the cluster centres (thin, unlike a cut through node positions).
  ivec clu_ptr, clu;         // CSR node -> cluster ids (interior nodes only; empty = unused)
  ivec clu_coord;            // 3 ints per cluster
};

struct Front {
  int32_t c0 = 0, w = 0;      // eliminated columns: elimination positions [c0, c0+w)
  int32_t ri = 0, rs = 0;     // rows of the update: interior ancestors / separators
  int32_t parent = -1;        // front id, or -1: update goes to the separator block
  int32_t level = 0;
  int32_t idx_off = 0;        // into ClassPlan::fidx  (w + ri + rs entries)
  int32_t rel_off = 0;        // into ClassPlan::rel   (ri + rs entries): position in parent's index list
  int64_t f_off = 0;          // frontal matrix (m x m, col-major) in the per-subdomain scratch
  int64_t lp_off = 0;         // L-side panel ((w+ri) x w, col-major) in the factor slab
  int64_t q_off = 0;          // U-side panel (w x ri, col-major) in the factor slab
  int32_t c_off = 0;          // contribution vector (ri) in the per-subdomain solve scratch
  int32_t a_off = 0;          // first row of this front in the assembled-row numbering (w + ri rows)
  int32_t lf_off = 0;         // offset of its assembled vector inside its tree level's LDS region (fused solve)
  int32_t ent_begin = 0, ent_end = 0;  // matrix entries assembled into this front
  int32_t child_begin = 0, child_end = 0;  // into ClassPlan::children
  bool big = false;            // factored AND solved by the multi-workgroup kernels
  bool wide = false;           // factored by the multi-workgroup kernels (big, or a Schur update too large for one workgroup)
  int32_t m() const { return w + ri + rs; }
};

struct ClassPlan {
  int32_t nI = 0, nS = 0;
  ivec perm;                  // elimination position -> local interior index
  ivec iperm;                 // local interior index -> elimination position
  std::vector<Front> fronts;  // in elimination (post) order
  ivec fidx;                  // per front: index list [cols | interior rows | separator rows]
                              // interior as elimination position, separator as nI + local id
  ivec rel;                   // per front: its update rows located in the parent's index list
                              // (parent == -1: separator local id)
  ivec children;              // child front ids grouped per front
  std::vector<ivec> levels;   // front ids per tree level (leaves = 0): fronts handled by one workgroup each
  std::vector<ivec> big_levels;  // per tree level: fronts spread over many workgroups
  // scheduling of the factorisation only: fronts one workgroup factors / fronts the multi-workgroup kernels factor
  // (the same partition of the fronts as levels / big_levels, with more fronts on the multi-workgroup side)
  std::vector<ivec> flevels, fwide_levels;
  int32_t max_w = 0;
  int32_t asm_rows = 0;       // sum over fronts of (w + ri)
  ivec asm_ptr, asm_src;      // per assembled row: contribution entries (index into the contrib array) to add
  // level-synchronous fused solve: work items (front << 16 | row) per tree level
  ivec fw_ptr, fw_items;      // forward: every row of [pivot | update] of every front of the level
  ivec bw_ptr, bw_items;      // backward: every pivot row of every front of the level
  int32_t max_level_rows = 0; // max over levels of sum (w + ri)
  // matrix entry assembly: sorted by front; S-block entries last (front == nfronts)
  ivec ent_id;                // entry number in the extended CSR
  ivec ent_pos;               // position in the front (row + m*col) or in S (row + nS*col)
  dvec ent_w;                 // weight
  int32_t s_ent_begin = 0;    // first separator-block entry
  int64_t scratch_size = 0;   // doubles of frontal scratch per subdomain (excluding S)
  int64_t factor_size = 0;    // doubles of factor slab per subdomain
  int32_t contrib_size = 0;   // doubles of solve scratch per subdomain
  int32_t max_front = 0;      // max m
  int32_t max_solve_rows = 0; // max w + ri
  int64_t nnz_factor = 0;     // stored panel entries (dense supernodal panels incl. explicit triangular inverses)
  int64_t nnz_sparse = 0;     // nnz(L + U) of the scalar LU in the same order (sparse-equivalent footprint)
  int64_t flops_factor = 0;
};

// throws hymls::Error(-4, ...) if the pressure nodes cannot all be attached
// (structurally singular interior block).
ClassPlan analyse_class(const LocalPattern& lp, int leaf_size, int max_width, int64_t big_panel_entries = (int64_t)1 << 40);
void print_plan_stats(const ClassPlan& P, const char* label, int nmembers);

}  // namespace hymls
