// hymls_cpu.cpp -- compiled CPU restatement of one level of the HYMLS preconditioner.  TEST INFRASTRUCTURE ONLY:
// it is the checker of the parity tests at sizes the numpy oracle cannot reach and the CPU baseline that bench.py
// times beside the GPU path (cpu_baseline.kind = "port"); nothing under hymls_amd/ links, loads or calls it.
//
// It performs the arithmetic the way the reference does, per subdomain and with sparse factors:
//   * subdomain solvers: F-matrix ordering (V-nodes by minimum degree on A + B B', every P-node right behind a V-node
//     that grounds it: reference src/HYMLS_MatrixUtils.cpp:1311-1755) and a sparse LU without numerical pivoting,
//     row-wise L and U (src/HYMLS_SparseDirectSolver.cpp:244-254: KLU with pivot tolerance 0 on the given ordering;
//     :788-856: permute, triangular solves, unpermute);
//   * SchurComplement::Construct11 / Construct22 per subdomain as dense matrices (src/HYMLS_SchurComplement.cpp:131-306:
//     n_sep right-hand sides through the subdomain solver, then A21 times the result);
//   * two-sided Householder per separator group on the dense matrix, kept entries = V-sum x V-sum + the non-V-sum
//     block of every linked set (src/HYMLS_SchurPreconditioner.cpp:698-986, src/HYMLS_Householder.cpp:38-126,
//     src/HYMLS_RestrictedOT.hpp:21-37), Replace for the A22 part, SumInto for the other (:835-865);
//   * dense LU with partial pivoting of every block (:284-291 -> dgetrf) and dgetrs in the apply (:1311-1346);
//   * ApplyInverse steps 1-8 of src/HYMLS_Preconditioner.cpp:930-1070 and :1010-1093, split around the solve with
//     the reduced (V-sum) system, which the caller (oracle/cpu_oracle.py) does recursively or with a sparse direct solver.
// Parallelism = the reference's: over subdomains (there: one MPI rank per group of subdomains; here OpenMP threads).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <iterator>
#include <numeric>
#include <unordered_map>
#include <vector>
#include <omp.h>

namespace {

using ivec = std::vector<int>;
using dvec = std::vector<double>;
constexpr double SMALL = 1e-14;   // HYMLS_SMALL_ENTRY, reference src/HYMLS_Macros.hpp:26-30

struct Csr { int n = 0; ivec ptr, col; dvec val; };

// ---------------------------------------------------------------- dense LU with partial pivoting (dgetrf / dgetrs)
bool dense_lu(int n, double* A, int* piv) {   // column-major, in place
  bool ok = true;
  for (int k = 0; k < n; k++) {
    int p = k;
    double best = std::abs(A[k + (size_t)n * k]);
    for (int i = k + 1; i < n; i++) if (std::abs(A[i + (size_t)n * k]) > best) { best = std::abs(A[i + (size_t)n * k]); p = i; }
    piv[k] = p;
    if (best == 0.0) { ok = false; continue; }   // dgetrf: INFO > 0, the factorisation goes on
    if (p != k) for (int j = 0; j < n; j++) std::swap(A[k + (size_t)n * j], A[p + (size_t)n * j]);
    const double ip = 1.0 / A[k + (size_t)n * k];
    for (int i = k + 1; i < n; i++) A[i + (size_t)n * k] *= ip;
    for (int j = k + 1; j < n; j++) {
      const double u = A[k + (size_t)n * j];
      if (u == 0.0) continue;
      for (int i = k + 1; i < n; i++) A[i + (size_t)n * j] -= A[i + (size_t)n * k] * u;
    }
  }
  return ok;
}
void dense_solve(int n, const double* A, const int* piv, double* x) {
  for (int k = 0; k < n; k++) if (piv[k] != k) std::swap(x[k], x[piv[k]]);
  for (int k = 0; k < n; k++) { const double xk = x[k]; if (xk != 0.0) for (int i = k + 1; i < n; i++) x[i] -= A[i + (size_t)n * k] * xk; }
  for (int k = n - 1; k >= 0; k--) { x[k] /= A[k + (size_t)n * k]; const double xk = x[k]; for (int i = 0; i < k; i++) x[i] -= A[i + (size_t)n * k] * xk; }
}

// ---------------------------------------------------------------- subdomain solver
struct SubLU {
  int n = 0;
  ivec perm;            // elimination position -> local interior index
  Csr L, U;             // row-wise: strictly lower (unit diagonal implied) and upper incl. diagonal, in elimination order
  bool singular = false;
  // x <- A^{-1} x (x in local interior order), work: n doubles
  void solve(double* x, double* w) const {
    for (int i = 0; i < n; i++) w[i] = x[perm[i]];
    for (int i = 0; i < n; i++) { double s = w[i]; for (int e = L.ptr[i]; e < L.ptr[i + 1]; e++) s -= L.val[e] * w[L.col[e]]; w[i] = s; }
    for (int i = n - 1; i >= 0; i--) {
      double s = w[i];
      const int b = U.ptr[i];
      for (int e = b + 1; e < U.ptr[i + 1]; e++) s -= U.val[e] * w[U.col[e]];
      w[i] = s / U.val[b];
    }
    for (int i = 0; i < n; i++) x[perm[i]] = w[i];
  }
};

// minimum degree on an undirected graph (sorted adjacency lists, modified in place)
ivec min_degree(std::vector<ivec>& adj, const std::vector<char>& active) {
  const int n = (int)adj.size();
  ivec order;
  std::vector<char> done(n, 0);
  ivec tmp;
  int nact = 0;
  for (int i = 0; i < n; i++) nact += active[i];
  for (int step = 0; step < nact; step++) {
    int v = -1;
    size_t best = (size_t)-1;
    for (int i = 0; i < n; i++) if (active[i] && !done[i] && adj[i].size() < best) { best = adj[i].size(); v = i; }
    done[v] = 1;
    order.push_back(v);
    const ivec N = adj[v];
    for (int u : N) {
      tmp.clear();
      std::set_union(adj[u].begin(), adj[u].end(), N.begin(), N.end(), std::back_inserter(tmp));
      ivec& a = adj[u];
      a.clear();
      for (int t : tmp) if (t != u && t != v) a.push_back(t);
    }
    ivec().swap(adj[v]);
  }
  return order;
}

// F-matrix ordering of a local matrix (n x n, sorted rows): returns elimination order; false if a P-node stays unpaired
bool fmatrix_ordering(const Csr& A, ivec& perm) {
  const int n = A.n;
  std::vector<char> isP(n, 1);
  for (int i = 0; i < n; i++)
    for (int e = A.ptr[i]; e < A.ptr[i + 1]; e++) if (A.col[e] == i && A.val[e] != 0.0) isP[i] = 0;
  // symmetrised structure
  std::vector<ivec> sadj(n);
  for (int i = 0; i < n; i++) for (int e = A.ptr[i]; e < A.ptr[i + 1]; e++) { const int j = A.col[e]; if (j != i) { sadj[i].push_back(j); sadj[j].push_back(i); } }
  for (auto& a : sadj) { std::sort(a.begin(), a.end()); a.erase(std::unique(a.begin(), a.end()), a.end()); }
  std::vector<ivec> g(n), pn(n);   // V-graph of A + B B'; pn[v]: P neighbours of V-node v, pn[p]: V neighbours of P-node p
  for (int i = 0; i < n; i++)
    for (int j : sadj[i]) {
      if (isP[i] != isP[j]) pn[i].push_back(j);
      else if (!isP[i]) g[i].push_back(j);
    }
  for (int p = 0; p < n; p++) if (isP[p]) for (int a : pn[p]) for (int b : pn[p]) if (a != b) g[a].push_back(b);
  for (auto& a : g) { std::sort(a.begin(), a.end()); a.erase(std::unique(a.begin(), a.end()), a.end()); }
  std::vector<char> isV(n);
  for (int i = 0; i < n; i++) isV[i] = !isP[i];
  const ivec vorder = min_degree(g, isV);
  // pressures: union-find over P-nodes, id n = "grounded" (boundary / no second pressure)
  ivec uf(n + 1);
  std::iota(uf.begin(), uf.end(), 0);
  auto find = [&](int x) { while (uf[x] != x) { uf[x] = uf[uf[x]]; x = uf[x]; } return x; };
  std::vector<char> pdone(n, 0);
  perm.clear();
  for (int v : vorder) {
    perm.push_back(v);
    if (pn[v].size() > 2) return false;          // not an F-matrix
    const int g1 = pn[v].size() > 0 ? find(pn[v][0]) : n, g2 = pn[v].size() > 1 ? find(pn[v][1]) : n;
    if (g1 == g2) continue;
    int elim;
    if (g1 == n) { elim = g2; uf[g2] = n; }
    else if (g2 == n) { elim = g1; uf[g1] = n; }
    else { elim = g2; uf[g2] = g1; }
    // the union-find root of a component is the pressure that is still uneliminated
    pdone[elim] = 1;
    perm.push_back(elim);
  }
  for (int p = 0; p < n; p++) if (isP[p] && !pdone[p]) return false;
  return (int)perm.size() == n;
}

// sparse LU without pivoting in the given order; row-wise (IKJ) with a dense work row
void sparse_lu(const Csr& A, SubLU& F) {
  const int n = A.n;
  F.n = n;
  ivec iperm(n);
  for (int i = 0; i < n; i++) iperm[F.perm[i]] = i;
  F.L.n = F.U.n = n;
  F.L.ptr.assign(1, 0); F.U.ptr.assign(1, 0);
  dvec w(n, 0.0);
  std::vector<char> mark(n, 0);
  ivec pat;
  for (int i = 0; i < n; i++) {
    const int r = F.perm[i];
    pat.clear();
    for (int e = A.ptr[r]; e < A.ptr[r + 1]; e++) { const int c = iperm[A.col[e]]; w[c] = A.val[e]; if (!mark[c]) { mark[c] = 1; pat.push_back(c); } }
    if (!mark[i]) { mark[i] = 1; pat.push_back(i); }   // structural diagonal
    // eliminate columns k < i in increasing order (fill enters the pattern on the way)
    std::make_heap(pat.begin(), pat.end(), std::greater<int>());
    ivec lcols, ucols;
    while (!pat.empty()) {
      std::pop_heap(pat.begin(), pat.end(), std::greater<int>());
      const int k = pat.back();
      pat.pop_back();
      if (k >= i) { ucols.push_back(k); continue; }
      lcols.push_back(k);
      const double lik = w[k] / F.U.val[F.U.ptr[k]];
      w[k] = lik;
      for (int e = F.U.ptr[k] + 1; e < F.U.ptr[k + 1]; e++) {
        const int c = F.U.col[e];
        if (!mark[c]) { mark[c] = 1; w[c] = 0.0; pat.push_back(c); std::push_heap(pat.begin(), pat.end(), std::greater<int>()); }
        w[c] -= lik * F.U.val[e];
      }
    }
    for (int k : lcols) { F.L.col.push_back(k); F.L.val.push_back(w[k]); mark[k] = 0; w[k] = 0.0; }
    F.L.ptr.push_back((int)F.L.col.size());
    for (int k : ucols) { F.U.col.push_back(k); F.U.val.push_back(w[k]); mark[k] = 0; w[k] = 0.0; }
    F.U.ptr.push_back((int)F.U.col.size());
    const double d = F.U.val[F.U.ptr[i]];
    if (d == 0.0 || !std::isfinite(d)) F.singular = true;
  }
}

// ---------------------------------------------------------------- Householder (reference src/HYMLS_Householder.cpp)
inline double sgn(double x) { return x < 0 ? -1.0 : (x > 0 ? 1.0 : 0.0); }
// X <- (2 u u'/u'u - I) X on the n rows starting at `pos` of the column-major (ld x ncols) matrix, or on columns (trans)
void hh_apply(double* S, int ld, int ncols, int pos, int n, const double* v0, bool cols) {
  if (n <= 0) return;
  const double sg = sgn(v0[0]);
  double nrm = 0;
  for (int i = 0; i < n; i++) nrm += v0[i] * v0[i];
  nrm = std::sqrt(nrm) * std::abs(sg);
  const double v1 = sg * v0[0] + nrm;
  if (std::abs(v1) < SMALL || nrm < SMALL) return;
  const double fac1 = 1.0 / (nrm * v1);
  for (int c = 0; c < ncols; c++) {
    double* p = cols ? S + c + (size_t)ld * pos : S + pos + (size_t)ld * c;
    const size_t inc = cols ? (size_t)ld : 1;
    double fac2 = nrm * p[0];
    for (int i = 0; i < n; i++) fac2 += (sg * v0[i]) * p[inc * i];
    const double fac = fac1 * fac2;
    p[0] = v1 * fac - p[0];
    for (int i = 1; i < n; i++) p[inc * i] = (sg * v0[i]) * fac - p[inc * i];
  }
}

struct Entry { double v22 = 0.0, v11 = 0.0; bool has22 = false; };

struct Level {
  int n = 0, nsd = 0, n1 = 0, n2 = 0, ng = 0;
  bool direct = false;
  int nthreads = 1;
  Csr A;
  // partition
  ivec int_ptr, int_idx;                 // interiors per subdomain (local rows)
  ivec gsd_ptr, g_ptr, g_idx, g_owned, g_link, g_olink;
  // numbering
  ivec i1, i2;                           // interior / separator order -> local row
  ivec pos1, pos2;                       // local row -> interior / separator index (-1)
  ivec sd_off;                           // first interior index of a subdomain
  std::vector<SubLU> lu;
  Csr A12, A21;                          // n1 x n2, n2 x n1
  // Schur preconditioner
  ivec og_ptr;                           // owned groups: offsets into the separator numbering (consecutive)
  dvec otw;                              // Householder rows
  std::vector<char> ot_has;              // per owned group: row present in T
  ivec vs;                               // V-sum separator index per owned group
  struct Block { ivec ids; dvec lu; ivec piv; };
  std::vector<Block> blocks;
  Csr red;                               // reduced matrix (V-sum x V-sum over owned-group index) or the whole S (direct)
  dvec next_tv;
  bool singular_sd = false, singular_block = false;
  // apply scratch
  dvec x1, x2, t1, t2, y2;
};

void interior_solve(const Level& L, double* x1) {
#pragma omp parallel num_threads(L.nthreads)
  {
    dvec w;
#pragma omp for schedule(dynamic, 4)
    for (int s = 0; s < L.nsd; s++) {
      const SubLU& F = L.lu[s];
      if (F.n == 0) continue;
      if ((int)w.size() < F.n) w.resize(F.n);
      F.solve(x1 + L.sd_off[s], w.data());
    }
  }
}

void spmv(const Csr& A, const double* x, double* y, double alpha, double beta, int nthreads) {
#pragma omp parallel for num_threads(nthreads) schedule(static)
  for (int i = 0; i < A.n; i++) {
    double s = 0.0;
    for (int e = A.ptr[i]; e < A.ptr[i + 1]; e++) s += A.val[e] * x[A.col[e]];
    y[i] = alpha * s + (beta == 0.0 ? 0.0 : beta * y[i]);
  }
}

// 2 T'(T v) - v  (reference src/HYMLS_Householder.cpp:353-363; a group without a row in T gets -v)
void apply_ot(const Level& L, double* x) {
#pragma omp parallel for num_threads(L.nthreads) schedule(static)
  for (int g = 0; g < L.ng; g++) {
    const int b = L.og_ptr[g], e = L.og_ptr[g + 1];
    double s = 0.0;
    for (int i = b; i < e; i++) s += L.otw[i] * x[i];
    for (int i = b; i < e; i++) x[i] = 2.0 * L.otw[i] * s - x[i];
  }
}

}  // namespace

extern "C" {

void* hcpu_level_create(int n, const int* rowptr, const int* col, const double* val, int nsd, const int* int_ptr,
                        const int* int_idx, const int* gsd_ptr, const int* g_ptr, const int* g_idx, const int* g_owned,
                        const int* g_link, const int* g_olink, const double* tv, int direct_schur, int nthreads) {
  Level* Lp = new Level();
  Level& L = *Lp;
  L.n = n; L.nsd = nsd; L.direct = direct_schur != 0; L.nthreads = std::max(1, nthreads);
  L.A.n = n; L.A.ptr.assign(rowptr, rowptr + n + 1); L.A.col.assign(col, col + rowptr[n]); L.A.val.assign(val, val + rowptr[n]);
  L.int_ptr.assign(int_ptr, int_ptr + nsd + 1); L.int_idx.assign(int_idx, int_idx + int_ptr[nsd]);
  L.gsd_ptr.assign(gsd_ptr, gsd_ptr + nsd + 1);
  const int ngt = gsd_ptr[nsd];
  L.g_ptr.assign(g_ptr, g_ptr + ngt + 1); L.g_idx.assign(g_idx, g_idx + g_ptr[ngt]);
  L.g_owned.assign(g_owned, g_owned + ngt); L.g_link.assign(g_link, g_link + ngt); L.g_olink.assign(g_olink, g_olink + ngt);
  // ---- numbering: interiors subdomain by subdomain (map1), owned groups subdomain by subdomain (map2)
  L.pos1.assign(n, -1); L.pos2.assign(n, -1);
  L.sd_off.assign(nsd + 1, 0);
  for (int s = 0; s < nsd; s++) {
    L.sd_off[s] = (int)L.i1.size();
    for (int e = int_ptr[s]; e < int_ptr[s + 1]; e++) { L.pos1[int_idx[e]] = (int)L.i1.size(); L.i1.push_back(int_idx[e]); }
  }
  L.sd_off[nsd] = (int)L.i1.size();
  L.og_ptr.assign(1, 0);
  for (int s = 0; s < nsd; s++)
    for (int g = gsd_ptr[s]; g < gsd_ptr[s + 1]; g++) {
      if (!g_owned[g]) continue;
      for (int e = g_ptr[g]; e < g_ptr[g + 1]; e++) { L.pos2[g_idx[e]] = (int)L.i2.size(); L.i2.push_back(g_idx[e]); }
      L.og_ptr.push_back((int)L.i2.size());
    }
  L.n1 = (int)L.i1.size(); L.n2 = (int)L.i2.size(); L.ng = (int)L.og_ptr.size() - 1;
  if (L.n1 + L.n2 != n) { delete Lp; return nullptr; }   // partition does not cover the map exactly once
  // ---- A12, A21 (MatrixBlock::Compute, reference src/HYMLS_MatrixBlock.cpp:74-134)
  auto extract = [&](const ivec& rows, const ivec& cpos, Csr& B) {
    B.n = (int)rows.size();
    B.ptr.assign(1, 0);
    for (int r : rows) {
      for (int e = L.A.ptr[r]; e < L.A.ptr[r + 1]; e++) if (cpos[L.A.col[e]] >= 0) { B.col.push_back(cpos[L.A.col[e]]); B.val.push_back(L.A.val[e]); }
      B.ptr.push_back((int)B.col.size());
    }
  };
  extract(L.i1, L.pos2, L.A12);
  extract(L.i2, L.pos1, L.A21);
  // ---- subdomain solvers (MatrixBlock::ComputeSubdomainSolvers, :210-292) + dense Schur parts, in parallel
  L.lu.resize(nsd);
  struct Kept { std::vector<uint64_t> key; dvec v22, v11; };
  std::vector<Kept> kept(nsd);
  bool sing = false;
#pragma omp parallel num_threads(L.nthreads) reduction(|| : sing)
  {
    ivec loc(n, -1);
    dvec S22, S11, rhs, wk;
#pragma omp for schedule(dynamic, 1)
    for (int s = 0; s < nsd; s++) {
      const int nI = int_ptr[s + 1] - int_ptr[s];
      const int* ii = int_idx + int_ptr[s];
      SubLU& F = L.lu[s];
      Csr A11;
      A11.n = nI;
      for (int i = 0; i < nI; i++) loc[ii[i]] = i;
      A11.ptr.assign(1, 0);
      for (int i = 0; i < nI; i++) {
        const int r = ii[i];
        for (int e = L.A.ptr[r]; e < L.A.ptr[r + 1]; e++) if (loc[L.A.col[e]] >= 0) { A11.col.push_back(loc[L.A.col[e]]); A11.val.push_back(L.A.val[e]); }
        A11.ptr.push_back((int)A11.col.size());
      }
      for (int i = 0; i < nI; i++) loc[ii[i]] = -1;
      if (nI > 0) {
        if (!fmatrix_ordering(A11, F.perm)) { sing = true; F.perm.resize(nI); std::iota(F.perm.begin(), F.perm.end(), 0); }
        sparse_lu(A11, F);
        sing = sing || F.singular;
      }
      // separators around the subdomain, group by group (SpawnMap(sd, Separators))
      ivec seps, gstart;
      for (int g = gsd_ptr[s]; g < gsd_ptr[s + 1]; g++) { gstart.push_back((int)seps.size()); for (int e = g_ptr[g]; e < g_ptr[g + 1]; e++) seps.push_back(g_idx[e]); }
      const int nS = (int)seps.size(), ngl = (int)gstart.size();
      if (nS == 0) continue;
      S22.assign((size_t)nS * nS, 0.0); S11.assign((size_t)nS * nS, 0.0);
      for (int a = 0; a < nS; a++) loc[seps[a]] = a;
      // Construct22: A22 restricted to these separators
      for (int a = 0; a < nS; a++) {
        const int r = seps[a];
        for (int e = L.A.ptr[r]; e < L.A.ptr[r + 1]; e++) { const int b = loc[L.A.col[e]]; if (b >= 0) S22[a + (size_t)nS * b] = L.A.val[e]; }
      }
      for (int a = 0; a < nS; a++) loc[seps[a]] = -1;
      // Construct11: -A21 (A11 \ A12), one right-hand side per separator column
      if (nI > 0) {
        for (int i = 0; i < nI; i++) loc[ii[i]] = i;
        // columns of A12_sd: entry (i, b) for interior row i
        std::vector<std::vector<std::pair<int, double>>> colsB(nS);
        {
          ivec sloc(0);
          std::unordered_map<int, int> smap;
          for (int a = 0; a < nS; a++) smap[seps[a]] = a;
          for (int i = 0; i < nI; i++) {
            const int r = ii[i];
            for (int e = L.A.ptr[r]; e < L.A.ptr[r + 1]; e++) { auto it = smap.find(L.A.col[e]); if (it != smap.end()) colsB[it->second].emplace_back(i, L.A.val[e]); }
          }
        }
        rhs.assign(nI, 0.0); wk.assign(nI, 0.0);
        for (int b = 0; b < nS; b++) {
          if (colsB[b].empty()) continue;
          std::fill(rhs.begin(), rhs.end(), 0.0);
          for (auto& pr : colsB[b]) rhs[pr.first] = pr.second;
          F.solve(rhs.data(), wk.data());
          for (int a = 0; a < nS; a++) {
            const int r = seps[a];
            double sum = 0.0;
            for (int e = L.A.ptr[r]; e < L.A.ptr[r + 1]; e++) { const int i = loc[L.A.col[e]]; if (i >= 0) sum += L.A.val[e] * rhs[i]; }
            S11[a + (size_t)nS * b] = -sum;
          }
        }
        for (int i = 0; i < nI; i++) loc[ii[i]] = -1;
      }
      Kept& K = kept[s];
      if (L.direct) {
        for (int b = 0; b < nS; b++) for (int a = 0; a < nS; a++) {
          if (L.pos2[seps[a]] < 0) continue;
          K.key.push_back(((uint64_t)L.pos2[seps[a]] << 32) | (uint32_t)seps[b]);   // column kept as local row id (resolved later)
          K.v22.push_back(S22[a + (size_t)nS * b]); K.v11.push_back(S11[a + (size_t)nS * b]);
        }
        continue;
      }
      // ConstructSCPart: OT on both parts, rows then columns of every group (RestrictedOT::Apply)
      dvec v(nS);
      for (int a = 0; a < nS; a++) v[a] = tv[seps[a]];
      for (int g = 0; g < ngl; g++) {
        const int pos = gstart[g], len = (g + 1 < ngl ? gstart[g + 1] : nS) - pos;
        for (dvec* M : {&S22, &S11}) { hh_apply(M->data(), nS, nS, pos, len, v.data() + pos, false); hh_apply(M->data(), nS, nS, pos, len, v.data() + pos, true); }
      }
      auto keep = [&](int a, int b) {
        K.key.push_back(((uint64_t)seps[a] << 32) | (uint32_t)seps[b]);
        K.v22.push_back(S22[a + (size_t)nS * b]); K.v11.push_back(S11[a + (size_t)nS * b]);
      };
      for (int a = 0; a < ngl; a++) for (int b = 0; b < ngl; b++) keep(gstart[a], gstart[b]);
      int nlink = 0;
      for (int g = 0; g < ngl; g++) nlink = std::max(nlink, g_link[gsd_ptr[s] + g] + 1);
      for (int l = 0; l < nlink; l++) {
        ivec locs;
        for (int g = 0; g < ngl; g++) if (g_link[gsd_ptr[s] + g] == l) { const int e = g + 1 < ngl ? gstart[g + 1] : nS; for (int t = gstart[g] + 1; t < e; t++) locs.push_back(t); }
        for (int a : locs) for (int b : locs) keep(a, b);
      }
    }
  }
  L.singular_sd = sing;
  // ---- assembly in subdomain order: Replace for the A22 part (first value wins: they are identical), SumInto for the other
  std::unordered_map<uint64_t, Entry> M;
  for (int s = 0; s < nsd; s++) {
    Kept& K = kept[s];
    for (size_t t = 0; t < K.key.size(); t++) {
      Entry& E = M[K.key[t]];
      if (!E.has22) { E.v22 = K.v22[t]; E.has22 = true; }
      E.v11 += K.v11[t];
    }
    Kept().key.swap(K.key); dvec().swap(K.v22); dvec().swap(K.v11);
  }
  if (L.direct) {
    // the whole Schur complement over the separator numbering (Preconditioner.cpp:485-500)
    std::vector<std::vector<std::pair<int, double>>> rows(L.n2);
    for (auto& kv : M) { const int r = (int)(kv.first >> 32), c = L.pos2[(int)(uint32_t)kv.first]; if (c >= 0) rows[r].emplace_back(c, kv.second.v22 + kv.second.v11); }
    L.red.n = L.n2; L.red.ptr.assign(1, 0);
    for (auto& r : rows) { std::sort(r.begin(), r.end()); for (auto& e : r) { L.red.col.push_back(e.first); L.red.val.push_back(e.second); } L.red.ptr.push_back((int)L.red.col.size()); }
    return Lp;
  }
  auto value = [&](int ra, int rb) { auto it = M.find(((uint64_t)ra << 32) | (uint32_t)rb); return it == M.end() ? 0.0 : it->second.v22 + it->second.v11; };
  // ---- Householder rows (InitializeOT, reference src/HYMLS_SchurPreconditioner.cpp:384-467)
  L.otw.assign(L.n2, 0.0); L.vs.resize(L.ng); L.ot_has.assign(L.ng, 0);
  for (int g = 0; g < L.ng; g++) {
    const int b = L.og_ptr[g], e = L.og_ptr[g + 1];
    L.vs[g] = b;
    dvec v(e - b);
    for (int i = b; i < e; i++) v[i - b] = tv[L.i2[i]];
    const double sg = sgn(v[0]);
    double nrm = 0;
    for (double& x : v) { x *= sg; nrm += x * x; }
    v[0] += std::sqrt(nrm);
    double nrm2 = 0;
    for (double x : v) nrm2 += x * x;
    nrm2 = std::sqrt(nrm2);
    if (nrm2 < SMALL) continue;
    L.ot_has[g] = 1;
    for (int i = b; i < e; i++) L.otw[i] = v[i - b] / nrm2;
  }
  // ---- block solvers (InitializeBlocks :301-340, Compute :284-291)
  for (int s = 0; s < nsd; s++) {
    int nol = 0;
    for (int g = gsd_ptr[s]; g < gsd_ptr[s + 1]; g++) nol = std::max(nol, g_olink[g] + 1);
    for (int l = 0; l < nol; l++) {
      Level::Block B;
      for (int g = gsd_ptr[s]; g < gsd_ptr[s + 1]; g++) if (g_olink[g] == l) for (int e = g_ptr[g] + 1; e < g_ptr[g + 1]; e++) B.ids.push_back(L.pos2[g_idx[e]]);
      if (B.ids.empty()) continue;
      L.blocks.push_back(std::move(B));
    }
  }
  bool bsing = false;
#pragma omp parallel for num_threads(L.nthreads) schedule(dynamic, 8) reduction(|| : bsing)
  for (size_t q = 0; q < L.blocks.size(); q++) {
    Level::Block& B = L.blocks[q];
    const int nb = (int)B.ids.size();
    B.lu.resize((size_t)nb * nb); B.piv.resize(nb);
    for (int b = 0; b < nb; b++) for (int a = 0; a < nb; a++) B.lu[a + (size_t)nb * b] = value(L.i2[B.ids[a]], L.i2[B.ids[b]]);
    if (!dense_lu(nb, B.lu.data(), B.piv.data())) bsing = true;
  }
  L.singular_block = bsing;
  // ---- reduced matrix on the V-sum nodes (ComputeNextLevel :520-629), rows/cols = owned group index
  {
    ivec gof(n, -1);
    for (int g = 0; g < L.ng; g++) gof[L.i2[L.vs[g]]] = g;
    std::vector<std::vector<std::pair<int, double>>> rows(L.ng);
    for (auto& kv : M) {
      const int ra = (int)(kv.first >> 32), rb = (int)(uint32_t)kv.first;
      if (gof[ra] >= 0 && gof[rb] >= 0) rows[gof[ra]].emplace_back(gof[rb], kv.second.v22 + kv.second.v11);
    }
    L.red.n = L.ng; L.red.ptr.assign(1, 0);
    for (auto& r : rows) { std::sort(r.begin(), r.end()); for (auto& e : r) { L.red.col.push_back(e.first); L.red.val.push_back(e.second); } L.red.ptr.push_back((int)L.red.col.size()); }
  }
  // next test vector = V-sum part of H * testvector (:569-573)
  {
    dvec t(L.n2);
    for (int i = 0; i < L.n2; i++) t[i] = tv[L.i2[i]];
    apply_ot(L, t.data());
    L.next_tv.resize(L.ng);
    for (int g = 0; g < L.ng; g++) L.next_tv[g] = t[L.vs[g]];
  }
  return Lp;
}

void hcpu_level_destroy(void* h) { delete (Level*)h; }
int hcpu_level_flags(void* h) { Level& L = *(Level*)h; return (L.singular_sd ? 1 : 0) | (L.singular_block ? 2 : 0); }
void hcpu_level_sizes(void* h, int* n1, int* n2, int* nred, int64_t* red_nnz, int64_t* nnz_lu) {
  Level& L = *(Level*)h;
  *n1 = L.n1; *n2 = L.n2; *nred = L.red.n; *red_nnz = (int64_t)L.red.col.size();
  int64_t t = 0;
  for (auto& F : L.lu) t += (int64_t)F.L.col.size() + (int64_t)F.U.col.size();
  *nnz_lu = t;
}
// reduced matrix: rows = owned groups (or separator indices when direct); row_nodes = local row of the node behind each row
void hcpu_level_reduced(void* h, int* ptr, int* col, double* val, int* row_nodes, double* next_tv) {
  Level& L = *(Level*)h;
  std::copy(L.red.ptr.begin(), L.red.ptr.end(), ptr);
  std::copy(L.red.col.begin(), L.red.col.end(), col);
  std::copy(L.red.val.begin(), L.red.val.end(), val);
  for (int r = 0; r < L.red.n; r++) row_nodes[r] = L.direct ? L.i2[r] : L.i2[L.vs[r]];
  if (!L.direct && next_tv) std::copy(L.next_tv.begin(), L.next_tv.end(), next_tv);
}
void hcpu_level_set_threads(void* h, int nthreads) { ((Level*)h)->nthreads = std::max(1, nthreads); }

// ApplyInverse, first half: steps 1-4 and 5a-5c of SURVEY 3.2; vrhs = right-hand side of the reduced system
void hcpu_level_apply_pre(void* h, const double* b, double* vrhs) {
  Level& L = *(Level*)h;
  L.x1.resize(L.n1); L.x2.resize(L.n2); L.t1.resize(L.n1); L.t2.resize(L.n2);
  for (int i = 0; i < L.n1; i++) L.x1[i] = b[L.i1[i]];
  for (int i = 0; i < L.n2; i++) L.t2[i] = b[L.i2[i]];
  interior_solve(L, L.x1.data());                                   // x1 = A11 \ b1
  spmv(L.A21, L.x1.data(), L.t2.data(), -1.0, 1.0, L.nthreads);     // b2 - A21 x1
  if (L.direct) { std::copy(L.t2.begin(), L.t2.end(), vrhs); return; }
  apply_ot(L, L.t2.data());                                         // B' = H rhs
  std::fill(L.x2.begin(), L.x2.end(), 0.0);
#pragma omp parallel num_threads(L.nthreads)
  {
    dvec w;
#pragma omp for schedule(dynamic, 8)
    for (size_t q = 0; q < L.blocks.size(); q++) {                  // ApplyBlockDiagonal
      const Level::Block& B = L.blocks[q];
      const int nb = (int)B.ids.size();
      w.resize(nb);
      for (int a = 0; a < nb; a++) w[a] = L.t2[B.ids[a]];
      dense_solve(nb, B.lu.data(), B.piv.data(), w.data());
      for (int a = 0; a < nb; a++) L.x2[B.ids[a]] = w[a];
    }
  }
  for (int g = 0; g < L.ng; g++) vrhs[g] = L.t2[L.vs[g]];
}
// second half: 5d-5e, 6-8
void hcpu_level_apply_post(void* h, const double* vsol, double* x) {
  Level& L = *(Level*)h;
  if (L.direct) std::copy(vsol, vsol + L.n2, L.x2.begin());
  else { for (int g = 0; g < L.ng; g++) L.x2[L.vs[g]] = vsol[g]; apply_ot(L, L.x2.data()); }
  spmv(L.A12, L.x2.data(), L.t1.data(), 1.0, 0.0, L.nthreads);      // y1 = A12 x2
  interior_solve(L, L.t1.data());                                   // A11 \ y1
  for (int i = 0; i < L.n1; i++) x[L.i1[i]] = L.x1[i] - L.t1[i];
  for (int i = 0; i < L.n2; i++) x[L.i2[i]] = L.x2[i];
}

int hcpu_max_threads() { return omp_get_max_threads(); }

}  // extern "C"
