// symbolic.cpp -- see symbolic.hpp
#include "symbolic.hpp"
#include <iterator>
#include <chrono>
#include <cstdio>
#include <cstdlib>

namespace hymls {

namespace {

constexpr int NDIR = 9;
const int DIRS[NDIR][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}, {1, 1, 0}, {1, -1, 0}, {1, 0, 1}, {1, 0, -1}, {0, 1, 1}, {0, 1, -1}};

struct NdCtx {
  const std::vector<ivec>* vadj;  // V-graph adjacency (local interior ids)
  const ivec* coord;
  int leaf;
  std::vector<char> side;                  // scratch marks (0 = not in the current set)
  int reach[NDIR];                          // longest edge of the graph along every direction
  std::vector<std::pair<int, int>> snodes; // (begin, end) into order
  ivec order;                              // V nodes in elimination order
  inline int proj(int v, int d) const {
    const ivec& co = *coord;
    return co[3 * v] * DIRS[d][0] + co[3 * v + 1] * DIRS[d][1] + co[3 * v + 2] * DIRS[d][2];
  }
};

// Nested dissection by cutting planes.  Subdomains of the Skew Cartesian partitioner (and the
// V-sum graphs built on them) have their natural separators on diagonal planes, so the cut is
// chosen among 9 directions (axes and face diagonals) and a few offsets around the median by the
// size of the vertex separator it produces (evaluated only inside the slab of nodes an edge can
// reach across the plane), weighted by the imbalance.
void nd_recurse(NdCtx& c, ivec& nodes) {
  const int n = (int)nodes.size();
  if (n == 0) return;
  auto emit_leaf = [&](ivec& v) {
    const ivec& co = *c.coord;
    std::sort(v.begin(), v.end(), [&](int a, int b) {
      if (co[3 * a + 2] != co[3 * b + 2]) return co[3 * a + 2] < co[3 * b + 2];
      if (co[3 * a + 1] != co[3 * b + 1]) return co[3 * a + 1] < co[3 * b + 1];
      if (co[3 * a] != co[3 * b]) return co[3 * a] < co[3 * b];
      return a < b;
    });
    const int b = (int)c.order.size();
    c.order.insert(c.order.end(), v.begin(), v.end());
    c.snodes.emplace_back(b, (int)c.order.size());
  };
  if (n <= c.leaf) { emit_leaf(nodes); return; }
  for (int v : nodes) c.side[v] = 1;   // membership mark while candidates are evaluated
  const int ntry = n > 4000 ? 9 : 15;
  double best_score = 1e300;
  int best_d = -1, best_t = 0;
  ivec pr(n), sorted(n);
  // large sets: only the directions whose slab behind the median plane is least populated
  bool use_dir[NDIR];
  for (int d = 0; d < NDIR; d++) use_dir[d] = true;
  if (n > 20000) {
    std::vector<std::pair<int64_t, int>> pop;
    for (int d = 0; d < NDIR; d++) {
      for (int i = 0; i < n; i++) pr[i] = c.proj(nodes[i], d);
      sorted = pr;
      std::nth_element(sorted.begin(), sorted.begin() + n / 2, sorted.end());
      const int med = sorted[n / 2];
      int64_t cnt = 0;
      for (int i = 0; i < n; i++) cnt += pr[i] >= med && pr[i] <= med + c.reach[d];
      pop.emplace_back(cnt, d);
    }
    std::sort(pop.begin(), pop.end());
    for (int d = 0; d < NDIR; d++) use_dir[d] = false;
    for (int q = 0; q < 3; q++) use_dir[pop[q].second] = true;
  }
  ivec mn(n), mx(n);
  for (int d = 0; d < NDIR; d++) {
    if (!use_dir[d]) continue;
    for (int i = 0; i < n; i++) pr[i] = c.proj(nodes[i], d);
    sorted = pr;
    std::sort(sorted.begin(), sorted.end());
    if (sorted.front() == sorted.back()) continue;
    // candidate thresholds: distinct values whose "< t" count is closest to n/2
    std::vector<std::pair<int, int>> cand;  // (|count - n/2|, t)
    for (int i = 1; i < n; i++)
      if (sorted[i] != sorted[i - 1]) cand.emplace_back(std::abs(i - n / 2), sorted[i]);
    std::sort(cand.begin(), cand.end());
    int nc = std::min<int>(ntry, (int)cand.size());
    while (nc > 1 && cand[nc - 1].first > n / 4) nc--;   // too unbalanced to be worth it
    int tmin = INT32_MAX, tmax = INT32_MIN;
    for (int q = 0; q < nc; q++) { tmin = std::min(tmin, cand[q].second); tmax = std::max(tmax, cand[q].second); }
    // one pass over the slab an edge can reach across any candidate plane: range of the neighbours
    for (int i = 0; i < n; i++) {
      const int pv = pr[i];
      mn[i] = pv; mx[i] = pv;
      if (pv < tmin - c.reach[d] || pv > tmax + c.reach[d]) continue;
      int lo = pv, hi = pv;
      for (int u : (*c.vadj)[nodes[i]])
        if (c.side[u]) { const int pu = c.proj(u, d); lo = std::min(lo, pu); hi = std::max(hi, pu); }
      mn[i] = lo; mx[i] = hi;
    }
    for (int q = 0; q < nc; q++) {
      const int t = cand[q].second;
      int sepR = 0, sepL = 0, nL = 0;
      for (int i = 0; i < n; i++) {
        const int pv = pr[i];
        if (pv < t) { nL++; sepL += mx[i] >= t; }
        else sepR += mn[i] < t;
      }
      const int sep = std::min(sepR, sepL);
      const double imb = std::abs(nL - n / 2) / (double)n;
      const double score = (sep + 1.0) * (1.0 + 2.0 * imb);
      if (score < best_score) { best_score = score; best_d = d; best_t = t; }
    }
  }
  if (best_d < 0) {
    for (int v : nodes) c.side[v] = 0;
    emit_leaf(nodes);
    return;
  }
  ivec L, R;
  for (int v : nodes) {
    if (c.proj(v, best_d) < best_t) { L.push_back(v); c.side[v] = 1; }
    else { R.push_back(v); c.side[v] = 2; }
  }
  ivec sepR, sepL;
  for (int v : R)
    for (int u : (*c.vadj)[v]) if (c.side[u] == 1) { sepR.push_back(v); break; }
  for (int v : L)
    for (int u : (*c.vadj)[v]) if (c.side[u] == 2) { sepL.push_back(v); break; }
  const bool useR = sepR.size() <= sepL.size();
  ivec& sep = useR ? sepR : sepL;
  for (int v : sep) c.side[v] = 3;
  ivec L2, R2;
  for (int v : L) if (c.side[v] == 1) L2.push_back(v);
  for (int v : R) if (c.side[v] == 2) R2.push_back(v);
  for (int v : nodes) c.side[v] = 0;
  if (L2.empty() && R2.empty()) { emit_leaf(nodes); return; }
  nd_recurse(c, L2);
  nd_recurse(c, R2);
  if (!sep.empty()) emit_leaf(sep);
}

// Nested dissection on cluster centres (see LocalPattern::clu).  nlo/nhi per node and direction are
// recomputed per call from the clusters (<= 8 per node).
struct CluCtx {
  const ivec* cptr; const ivec* cids; const ivec* ccoord;
  const std::vector<ivec>* vclu;   // per V-node: its clusters incl. those of its P neighbours
  inline int cproj(int cl, int d) const {
    const ivec& co = *ccoord;
    return co[3 * cl] * DIRS[d][0] + co[3 * cl + 1] * DIRS[d][1] + co[3 * cl + 2] * DIRS[d][2];
  }
};

void nd_cluster_recurse(NdCtx& c, const CluCtx& k, ivec& nodes) {
  const int n = (int)nodes.size();
  if (n == 0) return;
  auto emit_leaf = [&](ivec& v) {
    const ivec& co = *c.coord;
    std::sort(v.begin(), v.end(), [&](int a, int b) {
      if (co[3 * a + 2] != co[3 * b + 2]) return co[3 * a + 2] < co[3 * b + 2];
      if (co[3 * a + 1] != co[3 * b + 1]) return co[3 * a + 1] < co[3 * b + 1];
      if (co[3 * a] != co[3 * b]) return co[3 * a] < co[3 * b];
      return a < b;
    });
    const int b = (int)c.order.size();
    c.order.insert(c.order.end(), v.begin(), v.end());
    c.snodes.emplace_back(b, (int)c.order.size());
  };
  if (n <= c.leaf) { emit_leaf(nodes); return; }
  double best_score = 1e300;
  int best_d = -1, best_t = 0;
  ivec lo(n), hi(n), slo, shi, cand;
  for (int d = 0; d < NDIR; d++) {
    cand.clear();
    for (int i = 0; i < n; i++) {
      int a = INT32_MAX, b = INT32_MIN;
      for (int cl : (*k.vclu)[nodes[i]]) { const int p = k.cproj(cl, d); a = std::min(a, p); b = std::max(b, p); cand.push_back(p); }
      lo[i] = a; hi[i] = b;
    }
    std::sort(cand.begin(), cand.end());
    cand.erase(std::unique(cand.begin(), cand.end()), cand.end());
    slo = lo; shi = hi;
    std::sort(slo.begin(), slo.end()); std::sort(shi.begin(), shi.end());
    for (int t : cand) {
      // clusters with projection < t are "left": L = nodes with hi < t, R = nodes with lo >= t
      const int nL = (int)(std::lower_bound(shi.begin(), shi.end(), t) - shi.begin());
      const int nR = n - (int)(std::lower_bound(slo.begin(), slo.end(), t) - slo.begin());
      if (nL == 0 || nR == 0) continue;
      const int sep = n - nL - nR;
      const double imb = std::abs(nL - nR) / (double)n;
      if (imb > 0.6) continue;
      const double score = (sep + 1.0) * (1.0 + 2.0 * imb);
      if (score < best_score) { best_score = score; best_d = d; best_t = t; }
    }
  }
  if (best_d < 0) { emit_leaf(nodes); return; }
  ivec L, R, S;
  for (int i = 0; i < n; i++) {
    int a = INT32_MAX, b = INT32_MIN;
    for (int cl : (*k.vclu)[nodes[i]]) { const int p = k.cproj(cl, best_d); a = std::min(a, p); b = std::max(b, p); }
    if (b < best_t) L.push_back(nodes[i]);
    else if (a >= best_t) R.push_back(nodes[i]);
    else S.push_back(nodes[i]);
  }
  nd_cluster_recurse(c, k, L);
  nd_cluster_recurse(c, k, R);
  if (!S.empty()) emit_leaf(S);
}

}  // namespace

void print_plan_stats(const ClassPlan& P, const char* label, int nmembers) {
  int nbig = 0, maxw = 0, maxr = 0;
  double wrs = 0;
  for (auto& F : P.fronts) { nbig += F.big; maxw = std::max(maxw, F.w); maxr = std::max(maxr, F.ri + F.rs); wrs += 2.0 * F.w * F.rs; }
  std::fprintf(stderr, "[hymls_mi] %s: members %d nI %d nS %d fronts %zu (big %d) levels %zu max_w %d max_r %d nnz_factor %.3g (separator panels 2 w rs: %.3g) scratch %.3g MB flops %.3g contrib %d level_rows %d\n",
               label, nmembers, P.nI, P.nS, P.fronts.size(), nbig, P.levels.size(), maxw, maxr, (double)P.nnz_factor, wrs,
               8e-6 * (double)P.scratch_size, (double)P.flops_factor, P.contrib_size, P.max_level_rows);
}

ClassPlan analyse_class(const LocalPattern& lp, int leaf_size, int max_width, int64_t big_panel_entries) {
  ClassPlan P;
  const int nI = lp.nI, nS = lp.nS, n = nI + nS;
  P.nI = nI; P.nS = nS;
  static const bool prof_an = std::getenv("HYMLS_MI_ANALYSE_PROF") != nullptr;   // development aid: seconds per section on stderr
  auto t_an = std::chrono::steady_clock::now();
  auto lap_an = [&](const char* what) {
    if (!prof_an || n < 10000) return;
    const auto t = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[hymls_mi] analyse_class section before mark %s: %.3f s\n", what, std::chrono::duration<double>(t - t_an).count());
    t_an = t;
  };
  lap_an("0");
  // --- symmetric adjacency of the extended pattern
  // (one large system -- the last-level solver -- spreads the row-wise passes over the host threads; the classes of a level are
  // analysed in parallel already)
  const bool big = n >= 20000;
  auto rows_for = [&](int64_t count, auto fn) { if (big) parallel_for(count, fn, 1024); else for (int64_t i = 0; i < count; i++) fn(i); };
  auto sort_unique = [](ivec& a) { std::sort(a.begin(), a.end()); a.erase(std::unique(a.begin(), a.end()), a.end()); };
  std::vector<ivec> adj(n);
  {
    ivec deg(n, 0);
    for (int i = 0; i < n; i++)
      for (int e = lp.rowptr[i]; e < lp.rowptr[i + 1]; e++) {
        const int j = lp.col[e];
        if (j != i) { deg[i]++; deg[j]++; }
      }
    rows_for(n, [&](int64_t i) { adj[i].reserve(deg[i]); });
  }
  for (int i = 0; i < n; i++)
    for (int e = lp.rowptr[i]; e < lp.rowptr[i + 1]; e++) {
      const int j = lp.col[e];
      if (j != i) { adj[i].push_back(j); adj[j].push_back(i); }
    }
  rows_for(n, [&](int64_t i) { sort_unique(adj[i]); });
  lap_an("1");
  // --- V-graph (A + B B^T) on interior V-nodes
  std::vector<ivec> vadj(nI);
  std::vector<ivec> pv(nI);  // for P-nodes: interior V neighbours; for V-nodes: interior P neighbours
  rows_for(nI, [&](int64_t i) {
    for (int j : adj[i]) {
      if (j >= nI) continue;
      if (lp.zero_diag[i] != lp.zero_diag[j]) pv[i].push_back(j);
      else if (!lp.zero_diag[i]) vadj[i].push_back(j);
    }
  });
  // a velocity block without couplings of its own (Darcy: A = a I, the velocities only meet through their pressures) fills
  // much less than its dense supernodes hold: small leaves keep the padding down -- measured on Darcy3D 256^3: 15.4 instead
  // of 22.8 GB of panels per launch and 5.3 instead of 6.3 ms at leaf 8 (profiles/r03_g_leaf_size_sweep.txt); leaves of
  // 4 / 6 / 8 / 12 / 16: 19.4 / 18.0 / 17.6 / 17.4 / 17.6 ms per ApplyInverse (gpurun_out/r3ak): 12.  With velocity
  // couplings (Stokes) the larger leaves win, see LevelSolver::build_classes
  bool diagonal_a = nI > 0;
  for (int i = 0; i < nI && diagonal_a; i++) if (!vadj[i].empty()) diagonal_a = false;
  bool any_p = false;
  for (int i = 0; i < nI && !any_p; i++) any_p = lp.zero_diag[i] != 0;
  static const int leaf_diag = std::getenv("HYMLS_MI_LEAF_SIZE_DIAG") ? std::max(1, std::atoi(std::getenv("HYMLS_MI_LEAF_SIZE_DIAG"))) : 12;
  if (diagonal_a && any_p) leaf_size = std::min(leaf_size, leaf_diag);
  // B B^T: the V-nodes of one pressure are connected to each other; row a collects through its own pressures (pv[a])
  rows_for(nI, [&](int64_t a) {
    if (lp.zero_diag[a]) return;
    for (int p : pv[a])
      for (int b : pv[p]) if (b != (int)a) vadj[a].push_back(b);
    sort_unique(vadj[a]);
  });
  lap_an("2");
  // --- nested dissection of the V-nodes
  NdCtx c;
  c.vadj = &vadj; c.coord = &lp.coord; c.leaf = std::max(leaf_size, 1);
  c.side.assign(nI, 0);
  for (int d = 0; d < NDIR; d++) {
    int r = 0;
    for (int v = 0; v < nI; v++)
      for (int u : vadj[v]) r = std::max(r, std::abs(c.proj(u, d) - c.proj(v, d)));
    c.reach[d] = r;
  }
  ivec vnodes;
  for (int i = 0; i < nI; i++) if (!lp.zero_diag[i]) vnodes.push_back(i);
  if (!lp.clu_ptr.empty()) {
    // cluster-based dissection: a V-node carries its own clusters and those of its P neighbours
    // (eliminating the pressure right after it connects all velocities of that pressure)
    std::vector<ivec> vclu(nI);
    for (int v : vnodes) {
      ivec& cl = vclu[v];
      cl.assign(lp.clu.begin() + lp.clu_ptr[v], lp.clu.begin() + lp.clu_ptr[v + 1]);
      for (int p : pv[v]) cl.insert(cl.end(), lp.clu.begin() + lp.clu_ptr[p], lp.clu.begin() + lp.clu_ptr[p + 1]);
      std::sort(cl.begin(), cl.end());
      cl.erase(std::unique(cl.begin(), cl.end()), cl.end());
      HYMLS_CHECK(!cl.empty(), -3, "node without cluster");
    }
    CluCtx k{&lp.clu_ptr, &lp.clu, &lp.clu_coord, &vclu};
    nd_cluster_recurse(c, k, vnodes);
  } else {
    nd_recurse(c, vnodes);
  }
  lap_an("3");
  // --- attach every P-node behind a V-node that grounds it (union-find over pressures;
  //     the id nI stands for "boundary / separator / no second pressure")
  ivec uf(nI + 1);
  std::iota(uf.begin(), uf.end(), 0);
  auto find = [&](int x) { while (uf[x] != x) { uf[x] = uf[uf[x]]; x = uf[x]; } return x; };
  ivec cont(nI + 1, 0);
  for (int p = 0; p < nI; p++) if (lp.zero_diag[p]) cont[p] = (int)pv[p].size();
  std::vector<char> pdone(nI, 0);
  P.perm.clear();
  std::vector<std::pair<int, int>> sn;  // supernodes as [begin,end) elimination positions
  for (auto& s : c.snodes) {
    const int b = (int)P.perm.size();
    for (int t = s.first; t < s.second; t++) {
      const int v = c.order[t];
      P.perm.push_back(v);
      HYMLS_CHECK(pv[v].size() <= 2, -4, "not an F-matrix: a velocity node couples to more than two pressures");
      int g1 = pv[v].size() > 0 ? find(pv[v][0]) : nI;
      int g2 = pv[v].size() > 1 ? find(pv[v][1]) : nI;
      if (g1 == g2) continue;
      int elim;
      if (g1 == nI) { elim = g2; uf[g2] = nI; }
      else if (g2 == nI) { elim = g1; uf[g1] = nI; }
      else if (cont[g2] > cont[g1]) { elim = g1; uf[g1] = g2; cont[g2] = cont[g1] + cont[g2] - 2; }
      else { elim = g2; uf[g2] = g1; cont[g1] = cont[g1] + cont[g2] - 2; }
      pdone[elim] = 1;
      P.perm.push_back(elim);
    }
    if ((int)P.perm.size() > b) sn.emplace_back(b, (int)P.perm.size());
  }
  for (int p = 0; p < nI; p++)
    HYMLS_CHECK(!lp.zero_diag[p] || pdone[p], -4,
                "structurally singular interior block: a zero-diagonal node cannot be paired with a velocity");
  HYMLS_CHECK((int)P.perm.size() == nI, -3, "ordering lost nodes");
  P.iperm.assign(nI, -1);
  for (int i = 0; i < nI; i++) P.iperm[P.perm[i]] = i;
  lap_an("4");
  // --- sparse-equivalent size of the factors: nnz(L + U) of a scalar (column by column) LU of the interior block in this
  //     elimination order, without supernode padding, dense leaves or explicit triangular inverses -- what a sparse
  //     solver such as the reference's KLU would stream per solve (the smaller of this and the stored
  //     footprint is the algorithmic figure).  Column structures are merged up the elimination tree.
  {
    std::vector<ivec> st(nI);        // struct(L_j) as elimination positions > j (interior only)
    std::vector<ivec> ekids(nI);
    int64_t nnz = 0;
    ivec tmp;
    for (int j = 0; j < nI; j++) {
      ivec& S = st[j];
      for (int u : adj[P.perm[j]]) if (u < nI && P.iperm[u] > j) S.push_back(P.iperm[u]);
      std::sort(S.begin(), S.end());
      for (int ch : ekids[j]) {
        tmp.clear();
        const ivec& C = st[ch];
        std::set_union(S.begin(), S.end(), std::upper_bound(C.begin(), C.end(), j), C.end(), std::back_inserter(tmp));
        S.swap(tmp);
        ivec().swap(st[ch]);
      }
      nnz += 1 + 2 * (int64_t)S.size();
      if (!S.empty()) ekids[S[0]].push_back(j);
    }
    P.nnz_sparse = nnz;
  }
  // wide supernodes stay whole (their pivot block is factored piece by piece in place and then
  // inverted explicitly, so that the solve needs one panel product per supernode)
  const int nf = (int)sn.size();
  ivec sn_of(nI);
  for (int s = 0; s < nf; s++) for (int t = sn[s].first; t < sn[s].second; t++) sn_of[t] = s;
  auto pos_of = [&](int node) { return node < nI ? P.iperm[node] : node; };  // separators keep nI + id
  lap_an("5");
  // --- symbolic factorisation on the given supernode partition
  P.fronts.resize(nf);
  std::vector<ivec> kids(nf);
  std::vector<ivec> rows(nf);
  ivec mark(n, -1);
  for (int s = 0; s < nf; s++) {
    Front& F = P.fronts[s];
    F.c0 = sn[s].first; F.w = sn[s].second - sn[s].first;
    const int cend = F.c0 + F.w;
    ivec& R = rows[s];
    for (int t = F.c0; t < cend; t++)
      for (int j : adj[P.perm[t]]) {
        const int pj = pos_of(j);
        if (pj >= cend && mark[pj] != s) { mark[pj] = s; R.push_back(pj); }
      }
    for (int ch : kids[s])
      for (int pj : rows[ch])
        if (pj >= cend && mark[pj] != s) { mark[pj] = s; R.push_back(pj); }
    std::sort(R.begin(), R.end());
    F.ri = (int)(std::lower_bound(R.begin(), R.end(), nI) - R.begin());
    F.rs = (int)R.size() - F.ri;
    F.parent = (F.ri > 0) ? sn_of[R[0]] : -1;
    if (F.parent >= 0) kids[F.parent].push_back(s);
  }
  lap_an("6");
  // --- levels
  int maxlev = 0;
  for (int s = 0; s < nf; s++) {
    int lv = 0;
    for (int ch : kids[s]) lv = std::max(lv, P.fronts[ch].level + 1);
    P.fronts[s].level = lv;
    maxlev = std::max(maxlev, lv);
  }
  P.levels.assign(nf ? maxlev + 1 : 0, ivec());
  P.big_levels.assign(nf ? maxlev + 1 : 0, ivec());
  P.flevels.assign(nf ? maxlev + 1 : 0, ivec());
  P.fwide_levels.assign(nf ? maxlev + 1 : 0, ivec());
  lap_an("7");
  // --- index lists, offsets
  int64_t foff = 0, fac = 0;
  int32_t coff = 0, aoff = 0;
  for (int s = 0; s < nf; s++) {
    Front& F = P.fronts[s];
    F.idx_off = (int32_t)P.fidx.size();
    for (int t = 0; t < F.w; t++) P.fidx.push_back(F.c0 + t);
    P.fidx.insert(P.fidx.end(), rows[s].begin(), rows[s].end());
    F.child_begin = (int32_t)P.children.size();
    P.children.insert(P.children.end(), kids[s].begin(), kids[s].end());
    F.child_end = (int32_t)P.children.size();
    const int64_t m = F.m();
    F.f_off = foff; foff += m * m;
    F.lp_off = fac; fac += (int64_t)(F.w + F.ri) * F.w;
    F.q_off = fac; fac += (int64_t)F.w * F.ri;
    F.c_off = 0;
    F.a_off = aoff; aoff += F.w + F.ri;
    P.max_front = std::max<int32_t>(P.max_front, (int32_t)m);
    P.max_w = std::max<int32_t>(P.max_w, F.w);
    {
      // a front whose Schur update is too much work for one workgroup goes to the multi-workgroup path
      const double r = F.ri + F.rs;
      static const double big_flops = std::getenv("HYMLS_MI_BIG_FLOPS") ? std::atof(std::getenv("HYMLS_MI_BIG_FLOPS")) : 2.5e7;
      F.big = r * r * F.w > big_flops || m > 2048 || (int64_t)(F.w + F.ri) * F.w > big_panel_entries || F.w > max_width;
      (F.big ? P.big_levels : P.levels)[F.level].push_back(s);
      // factorisation: a Schur update of a few MFLOP already takes one workgroup milliseconds (measured: the root front
      // of a level-1 subdomain, 546 x 546 x 49, 4.4 ms in k_factor_level -- 41 % in its VALU GEMM, 21 % extend-add); the
      // grid-wide kernels (MFMA GEMM batched over the members) do the same work for all members of a chunk in one go
      static const double wide_flops = std::getenv("HYMLS_MI_WIDE_FACTOR_FLOPS") ? std::atof(std::getenv("HYMLS_MI_WIDE_FACTOR_FLOPS")) : 3e6;
      F.wide = F.big || r * r * F.w > wide_flops;
      (F.wide ? P.fwide_levels : P.flevels)[F.level].push_back(s);
    }
    P.max_solve_rows = std::max<int32_t>(P.max_solve_rows, F.w + F.ri);
    P.nnz_factor += (int64_t)F.w * F.w + 2LL * F.w * F.ri;
    const double w = F.w, r = F.ri + F.rs;
    // LU of the pivot block + both triangular inverses (2/3 w^3 each pair), the two panels, the Schur update
    P.flops_factor += (int64_t)(4.0 / 3.0 * w * w * w + 2.0 * w * w * r + 2.0 * w * r * r);
  }
  // contribution vectors: a front's vector is written at its own tree level and read at its parent's;
  // its space is reused from the level after the parent's (first-fit free list)
  {
    const int nl = (int)P.levels.size();
    std::vector<ivec> by_level(nl), release(nl + 1);
    for (int s = 0; s < nf; s++) by_level[P.fronts[s].level].push_back(s);
    std::vector<std::pair<int32_t, int32_t>> freel;  // (offset, size)
    for (int l = 0; l < nl; l++) {
      for (int s : release[l]) {
        freel.emplace_back(P.fronts[s].c_off, P.fronts[s].ri);
      }
      // merge adjacent free blocks
      std::sort(freel.begin(), freel.end());
      std::vector<std::pair<int32_t, int32_t>> merged;
      for (auto& f : freel) {
        if (!merged.empty() && merged.back().first + merged.back().second == f.first) merged.back().second += f.second;
        else merged.push_back(f);
      }
      freel.swap(merged);
      for (int s : by_level[l]) {
        Front& F = P.fronts[s];
        if (F.ri == 0) continue;
        bool placed = false;
        for (auto& f : freel)
          if (f.second >= F.ri) { F.c_off = f.first; f.first += F.ri; f.second -= F.ri; placed = true; break; }
        if (!placed) { F.c_off = coff; coff += F.ri; }
        const int pl = P.fronts[F.parent].level;
        release[std::min(pl + 1, nl)].push_back(s);
      }
    }
  }
  P.scratch_size = foff; P.factor_size = fac; P.contrib_size = coff;
  lap_an("8");
  // --- relative maps (update rows -> position in the parent's index list / separator id)
  ivec where(n, -1);
  for (int s = 0; s < nf; s++) P.fronts[s].rel_off = -1;
  {
    int32_t roff = 0;
    for (int s = 0; s < nf; s++) { P.fronts[s].rel_off = roff; roff += P.fronts[s].ri + P.fronts[s].rs; }
    P.rel.assign(roff, -1);
  }
  for (int p = 0; p < nf; p++) {
    const Front& Fp = P.fronts[p];
    for (int t = 0; t < Fp.m(); t++) where[P.fidx[Fp.idx_off + t]] = t;
    for (int e = Fp.child_begin; e < Fp.child_end; e++) {
      const Front& Fc = P.fronts[P.children[e]];
      for (int t = 0; t < Fc.ri + Fc.rs; t++) {
        const int pj = P.fidx[Fc.idx_off + Fc.w + t];
        HYMLS_CHECK(where[pj] >= 0, -3, "symbolic: child row missing in parent front");
        P.rel[Fc.rel_off + t] = where[pj];
      }
    }
    for (int t = 0; t < Fp.m(); t++) where[P.fidx[Fp.idx_off + t]] = -1;
  }
  for (int s = 0; s < nf; s++) {
    const Front& F = P.fronts[s];
    if (F.parent >= 0) continue;
    for (int t = 0; t < F.rs; t++) P.rel[F.rel_off + t] = P.fidx[F.idx_off + F.w + t] - nI;
  }
  lap_an("9");
  // --- assembly pull lists: for every row of every front's solve vector [pivot | update rows]
  //     the contribution-vector entries of its children that land there (fixed order)
  P.asm_rows = aoff;
  {
    std::vector<ivec> src((size_t)aoff);
    for (int p = 0; p < nf; p++) {
      const Front& Fp = P.fronts[p];
      for (int e = Fp.child_begin; e < Fp.child_end; e++) {
        const Front& Fc = P.fronts[P.children[e]];
        for (int k = 0; k < Fc.ri; k++) src[(size_t)Fp.a_off + P.rel[Fc.rel_off + k]].push_back(Fc.c_off + k);
      }
    }
    P.asm_ptr.assign((size_t)aoff + 1, 0);
    for (int i = 0; i < aoff; i++) {
      P.asm_src.insert(P.asm_src.end(), src[i].begin(), src[i].end());
      P.asm_ptr[i + 1] = (int32_t)P.asm_src.size();
    }
  }
  lap_an("10");
  // --- work items of the level-synchronous fused solve
  {
    const int nl = (int)P.levels.size();
    P.fw_ptr.assign(1, 0); P.bw_ptr.assign(1, 0);
    for (int l = 0; l < nl; l++) {
      int32_t lf = 0;
      for (int pass = 0; pass < 2; pass++)
        for (int s : (pass == 0 ? P.levels[l] : P.big_levels[l])) {
          Front& F = P.fronts[s];
          F.lf_off = lf; lf += F.w;   // only the pivot part of the assembled vector needs its own space
          if (nf < 65536 && F.w + F.ri < 65536) {
            for (int r = 0; r < F.w + F.ri; r++) P.fw_items.push_back((s << 16) | r);
            for (int r = 0; r < F.w; r++) P.bw_items.push_back((s << 16) | r);
          }
        }
      P.max_level_rows = std::max(P.max_level_rows, lf);
      P.fw_ptr.push_back((int32_t)P.fw_items.size());
      P.bw_ptr.push_back((int32_t)P.bw_items.size());
    }
  }
  lap_an("11");
  // --- matrix entries -> fronts
  // counting sort of the entries by front (ascending entry index within a front), then positions front by front; one
  // large system (the coarse solver: 1e7 entries and more) spreads both passes over the host threads
  {
    const int64_t nnz = (int64_t)lp.col.size();
    const bool par = n >= 20000;
    ivec rowof((size_t)nnz), ent_front((size_t)nnz);
    auto classify = [&](int64_t i) {
      for (int e = lp.rowptr[i]; e < lp.rowptr[i + 1]; e++) {
        rowof[e] = (int)i;
        const int pm = std::min(pos_of((int)i), pos_of(lp.col[e]));
        ent_front[e] = pm < nI ? sn_of[pm] : nf;
      }
    };
    if (par) parallel_for(n, classify, 4096); else for (int i = 0; i < n; i++) classify(i);
    lap_an("12a classify");
    std::vector<int64_t> fptr((size_t)nf + 2, 0);
    P.ent_id.resize((size_t)nnz); P.ent_pos.resize((size_t)nnz); P.ent_w.resize((size_t)nnz);
    {
      // stable counting sort in NB contiguous blocks of entries: a histogram per block, offsets per (front, block), fill
      const int NB = par ? 16 : 1;
      std::vector<std::vector<int64_t>> hist((size_t)NB, std::vector<int64_t>((size_t)nf + 1, 0));
      auto blk = [&](int b) { return std::make_pair(nnz * b / NB, nnz * (b + 1) / NB); };
      auto count = [&](int64_t b) { const auto r = blk((int)b); for (int64_t e = r.first; e < r.second; e++) hist[b][ent_front[e]]++; };
      if (par) parallel_for(NB, count, 1); else count(0);
      int64_t run = 0;
      for (int s2 = 0; s2 <= nf; s2++) {
        fptr[s2] = run;
        for (int b = 0; b < NB; b++) { const int64_t c = hist[b][s2]; hist[b][s2] = run; run += c; }
      }
      fptr[(size_t)nf + 1] = run;
      auto fill = [&](int64_t b) { const auto r = blk((int)b); for (int64_t e = r.first; e < r.second; e++) P.ent_id[hist[b][ent_front[e]]++] = (int32_t)e; };
      if (par) parallel_for(NB, fill, 1); else fill(0);
    }
    lap_an("12b count + fill");
    auto place = [&](int64_t s2) {
      const int sf = (int)s2;
      if (sf < nf) {
        if (fptr[sf] == fptr[sf + 1]) return;
        static thread_local ivec wh;                 // node position -> row of the front, -1 outside (restored after use)
        if ((int)wh.size() < n) wh.resize(n, -1);
        const Front& F = P.fronts[sf];
        for (int t = 0; t < F.m(); t++) wh[P.fidx[F.idx_off + t]] = t;
        bool ok = true;
        for (int64_t k = fptr[sf]; k < fptr[sf + 1]; k++) {
          const int e = P.ent_id[k];
          const int a = wh[pos_of(rowof[e])], b = wh[pos_of(lp.col[e])];
          ok = ok && a >= 0 && b >= 0;
          P.ent_pos[k] = a + F.m() * b;
          P.ent_w[k] = lp.weight.empty() ? 1.0 : lp.weight[e];
        }
        for (int t = 0; t < F.m(); t++) wh[P.fidx[F.idx_off + t]] = -1;
        HYMLS_CHECK(ok, -3, "symbolic: matrix entry outside its front");
      } else {
        for (int64_t k = fptr[nf]; k < fptr[nf + 1]; k++) {
          const int e = P.ent_id[k];
          P.ent_pos[k] = (rowof[e] - nI) + nS * (lp.col[e] - nI);
          P.ent_w[k] = lp.weight.empty() ? 1.0 : lp.weight[e];
        }
      }
    };
    if (par) parallel_for(nf + 1, place, 8); else for (int s2 = 0; s2 <= nf; s2++) place(s2);
    lap_an("12c place");
    for (int s2 = 0; s2 < nf; s2++) {
      const bool any = fptr[s2] < fptr[s2 + 1];
      P.fronts[s2].ent_begin = any ? (int32_t)fptr[s2] : 0;
      P.fronts[s2].ent_end = any ? (int32_t)fptr[s2 + 1] : 0;
    }
    P.s_ent_begin = (int32_t)fptr[nf];
  }
  lap_an("end");
  return P;
}

}  // namespace hymls
