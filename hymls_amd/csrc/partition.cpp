// partition.cpp -- see partition.hpp
#include "partition.hpp"

namespace hymls {

namespace {

struct Range {
  bool skip;
  int type, start, end;
};

// one coordinate direction of a subdomain is cut into three pieces: the layer in
// front of it (type 0), its own cells (type 1) and its closing separator layer
// (type 2).  Behaviour of GetSubdomainStartAndEnd (CartesianPartitioner.cpp:224-263),
// including "Retain Nodes" > 1 (idx_max pieces of type 1); in a periodic direction the first
// subdomain keeps its front layer and the last one its closing layer (they wrap around).
Range piece(int pos, int idx, int idx_max, int dim, int mx, bool perio) {
  Range r{false, 0, 0, 0};
  const int len = std::max((mx + idx_max - 1) / idx_max, 1);
  r.type = (idx == idx_max) ? 2 : (idx >= 0 ? 1 : 0);
  r.start = idx;
  if (idx == idx_max) r.start = mx;
  else if (idx > 0) r.start = std::min(len * idx, mx);
  r.end = r.start + 1;
  if (r.type == 1) r.end = std::min(len * (idx + 1), mx);
  if (!perio) {
    if (pos == 0 && idx == -1) { r.skip = true; return r; }
    if (pos + mx + 1 == dim) {
      if (idx == idx_max) { r.skip = true; return r; }
      if (idx == idx_max - 1) r.end += 1;
    }
  }
  if (r.start == r.end) r.skip = true;
  return r;
}

inline bool is_velocity(int32_t t) { return t == VT_U || t == VT_V || t == VT_W || t == VT_LAPLACE; }

}  // namespace

int cartesian_num_subdomains(const Params& p) {
  return ((p.nx - 1) / p.sx + 1) * ((p.ny - 1) / p.sy + 1) * ((p.nz - 1) / p.sz + 1);
}

void cartesian_get_groups(const Params& p, int sd, ivec& interior, std::vector<Group>& groups) {
  interior.clear();
  groups.clear();
  const int npx = (p.nx - 1) / p.sx + 1, npy = (p.ny - 1) / p.sy + 1, npz = (p.nz - 1) / p.sz + 1;
  const int xpos = (sd % npx) * p.sx, ypos = ((sd / npx) % npy) * p.sy, zpos = ((sd / npx / npy) % npz) * p.sz;
  const int xmax = std::min(p.nx - xpos - 1, p.sx - 1);
  const int ymax = std::min(p.ny - ypos - 1, p.sy - 1);
  const int zmax = std::min(p.nz - zpos - 1, p.sz - 1);
  HYMLS_CHECK(!(xmax == 0 || ymax == 0 || (zmax == 0 && p.nz > 1)), -2, "Can't have a subdomain of size 1");
  const int imax = p.rx > 1 ? p.rx : 1, jmax = p.ry > 1 ? p.ry : 1, kmax = p.rz > 1 ? p.rz : 1;
  ivec retained;
  for (int kidx = -1; kidx <= kmax; kidx++) {
    const bool kint = kidx >= 0 && kidx < kmax;
    Range K = piece(zpos, kidx, kmax, p.nz, zmax, p.perio[2]);
    if (K.skip) continue;
    for (int jidx = -1; jidx <= jmax; jidx++) {
      const bool jint = jidx >= 0 && jidx < jmax;
      Range J = piece(ypos, jidx, jmax, p.ny, ymax, p.perio[1]);
      if (J.skip) continue;
      for (int iidx = -1; iidx <= imax; iidx++) {
        const bool iint = iidx >= 0 && iidx < imax;
        Range I = piece(xpos, iidx, imax, p.nx, xmax, p.perio[0]);
        if (I.skip) continue;
        for (int d = 0; d < p.dof; d++) {
          const int32_t vt = p.vtype[d];
          const bool is_p = vt == VT_P;
          if ((is_p || vt == VT_INTERIOR) && (iidx == -1 || jidx == -1 || kidx == -1)) continue;
          ivec* dst;
          const bool to_interior =
              (iint && jint && kint) || vt == VT_INTERIOR ||
              (is_p && ((iint && jint) || (iint && kint) || (jint && kint) || p.retain_pressures > 1));
          if (to_interior) {
            dst = &interior;
          } else {
            int type = -1000;
            if (p.link_retained) type = 2 * p.dof * (I.type + 3 * (J.type + 3 * K.type));
            if (!(p.link_velocities && is_velocity(vt))) type += 2 * d;
            groups.emplace_back();
            groups.back().type = type;
            dst = &groups.back().nodes;
          }
          for (int k = K.start; k < K.end; k++)
            for (int j = J.start; j < J.end; j++)
              for (int i = I.start; i < I.end; i++) {
                const int32_t gid = d + ((i + xpos + p.nx) % p.nx) * p.dof +
                                    ((j + ypos + p.ny) % p.ny) * p.nx * p.dof +
                                    ((k + zpos + p.nz) % p.nz) * p.nx * p.ny * p.dof;
                if (is_p && i >= 0 && j >= 0 && k >= 0 && (int)retained.size() < p.retain_pressures)
                  retained.push_back(gid);
                else
                  dst->push_back(gid);
              }
        }
      }
    }
  }
  groups.erase(std::remove_if(groups.begin(), groups.end(), [](const Group& g) { return g.nodes.empty(); }),
               groups.end());
  for (int32_t g : retained) {
    groups.emplace_back();
    groups.back().type = -1;
    groups.back().nodes.push_back(g);
  }
}

static void link_groups(const std::vector<Group>& groups, const ivec& idxs, std::vector<ivec>& linked) {
  linked.clear();
  for (int32_t gi : idxs) {
    const int t = groups[gi].type;
    bool found = false;
    if (t >= 0)
      for (auto& L : linked)
        if (groups[L[0]].type == t) { L.push_back(gi); found = true; break; }
    if (!found) linked.push_back(ivec{gi});
  }
}

void skew_get_groups(const Params& p, int sd, ivec& interior, std::vector<Group>& groups);
int skew_num_subdomains(const Params& p);
void skew_sd_position(const Params& p, int sd, int& x, int& y, int& z);

int num_subdomains(const Params& p) { return p.partitioner == 0 ? cartesian_num_subdomains(p) : skew_num_subdomains(p); }

void sd_position(const Params& p, int sd, int& x, int& y, int& z) {
  if (p.partitioner == 0) {
    const int npx = (p.nx - 1) / p.sx + 1, npy = (p.ny - 1) / p.sy + 1, npz = (p.nz - 1) / p.sz + 1;
    x = (sd % npx) * p.sx; y = ((sd / npx) % npy) * p.sy; z = ((sd / npx / npy) % npz) * p.sz;
  } else {
    skew_sd_position(p, sd, x, y, z);
  }
}

HierMap build_hiermap(const Params& p, const std::vector<char>* present, const std::vector<char>* cand) {
  HierMap h;
  h.ngid = (int64_t)p.nx * p.ny * p.nz * p.dof;
  HYMLS_CHECK(h.ngid < (int64_t)1 << 31, -2, "more than 2^31 unknowns need 64-bit GIDs");
  const int nsd = num_subdomains(p);
  h.sd.resize(nsd);
  // groups of every (candidate) subdomain: independent, in parallel; ownership ("first subdomain that lists the group")
  // afterwards in subdomain order
  if (nsd > 0) { ivec it; std::vector<Group> gr; if (p.partitioner != 0) skew_get_groups(p, 0, it, gr); }   // build the template once
  parallel_for(nsd, [&](int64_t s) {
    if (cand && !(*cand)[s]) return;
    Subdomain& S = h.sd[s];
    std::vector<Group> raw;
    if (p.partitioner == 0) cartesian_get_groups(p, (int)s, S.interior, raw);
    else skew_get_groups(p, (int)s, S.interior, raw);
    auto filt = [&](ivec& v) {
      std::sort(v.begin(), v.end());
      if (present) v.erase(std::remove_if(v.begin(), v.end(), [&](int32_t g) { return !(*present)[g]; }), v.end());
    };
    filt(S.interior);
    for (auto& g : raw) {
      filt(g.nodes);
      if (!g.nodes.empty()) S.groups.push_back(std::move(g));
    }
  }, 4);
  std::vector<char> seen(h.ngid, 0);  // first gid of every group already owned
  for (int s = 0; s < nsd; s++) {
    if (cand && !(*cand)[s]) continue;
    Subdomain& S = h.sd[s];
    for (int gi = 0; gi < (int)S.groups.size(); gi++) {
      const int32_t first = S.groups[gi].nodes[0];
      if (!seen[first]) { seen[first] = 1; S.owned.push_back(gi); }
    }
  }
  // (the linking needs nothing of the other subdomains: in parallel)
  parallel_for(nsd, [&](int64_t s) {
    if (cand && !(*cand)[s]) return;
    Subdomain& S = h.sd[s];
    ivec all(S.groups.size());
    std::iota(all.begin(), all.end(), 0);
    link_groups(S.groups, all, S.linked);
    link_groups(S.groups, S.owned, S.owned_linked);
  }, 16);
  return h;
}

}  // namespace hymls
