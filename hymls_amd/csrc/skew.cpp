// skew.cpp -- "Skew Cartesian" partitioner: subdomains are xy-diamonds that shear with z, so that
// no pressure node is left isolated on a subdomain edge (needed for 3D Stokes on a C grid).
//
// Behaviour of the reference's SkewCartesianPartitioner, periodic directions included (:154-159 duplicate
// subdomains, :199-206 ids across the periodic boundary, :686-688 wrap-around), for
// "Retain Nodes" <= 1 (src/HYMLS_SkewCartesianPartitioner.cpp: subdomain numbering :128-213,
// node template :349-563, grouping by the set of neighbouring subdomains :565-651, placement,
// clipping, retained pressure, ownership split and wall nodes :653-812).  Here the template is
// kept as integer (x,y,z,var) offsets and grouped through coordinate hashing instead of linear
// ids on a 4sx-wide auxiliary grid.
#include "partition.hpp"
#include <map>
#include <unordered_map>
#include <array>

namespace hymls {

namespace {

struct Pt { int x, y, z, var; };
inline int64_t key(const Pt& p) { return (((int64_t)(p.z + 512) * 4096 + (p.y + 512)) * 4096 + (p.x + 512)) * 8 + p.var; }

struct XY { int x, y; };
inline bool operator==(const XY& a, const XY& b) { return a.x == b.x && a.y == b.y; }

// rows of a 45-degree diamond: `length` widening rows, optionally one row of equal width, then
// narrowing rows; row r spans x in [xl, xr] at y = y0 + r.
struct Diamond { std::vector<std::vector<XY>> rows; };

Diamond make_diamond(int x0, int y0, int length, int kind) {
  int xl = x0, xr = x0, height = 2 * length;
  bool extra = false;
  if (kind == 0) { xl -= 1; height++; extra = true; }   // u: starts two wide
  else if (kind == 3) { height++; extra = true; }        // p
  Diamond d;
  for (int r = 0; r < height - 1; r++) {
    d.rows.emplace_back();
    for (int x = xl; x <= xr; x++) d.rows.back().push_back({x, y0 + r});
    if (r < length - 1) { xl--; xr++; }
    else if (extra && r == length - 1) {}
    else { xl++; xr--; }
  }
  return d;
}

struct SkewTemplate {
  int sx = 0, dof = 0;
  // groups[0] = interior candidates; groups[g>=1][var] = nodes shared with the same set of
  // neighbouring subdomain copies.  Offsets are relative to the auxiliary grid origin.
  std::vector<std::vector<std::vector<Pt>>> groups;
};

SkewTemplate build_template(const Params& p) {
  const int sx = p.sx, dof = p.dof;
  SkewTemplate T;
  T.sx = sx; T.dof = dof;
  const int first_y[4] = {1, 0, 1, 1};
  const int base_len[4] = {sx / 2, sx / 2 + 1, sx / 2 + 1, sx / 2};
  const int32_t kind_of[4] = {VT_U, VT_V, VT_W, VT_P};
  // layers[kind][z] : list of (x,y)
  std::vector<std::vector<std::vector<XY>>> layers(4);
  for (int kind = 0; kind < 4; kind++) {
    auto& L = layers[kind];
    L.assign(2 * sx + 1, {});
    Diamond D = make_diamond(sx / 2, first_y[kind], base_len[kind], kind);
    std::vector<XY> plane;
    for (auto& r : D.rows) plane.insert(plane.end(), r.begin(), r.end());
    L[sx] = plane;
    if (p.nz <= 1) continue;
    std::vector<XY> bottom, top = plane;
    std::vector<int> rowlen;
    for (auto& r : D.rows) rowlen.push_back((int)r.size() - 1);
    std::vector<int> active, offset;
    for (int i = 0; i < base_len[kind]; i++) { active.push_back(i); offset.push_back(rowlen[i]); }
    auto shifted = [&](const std::vector<XY>& src, int dy, std::vector<XY>& dst) {
      for (auto& q : src) dst.push_back({q.x, q.y + dy});
    };
    for (int i = 0; i < sx; i++) {
      for (size_t j = 0; j < active.size(); j++) {
        const XY v = D.rows[active[j]][offset[j]];
        bottom.push_back(v);
        top.erase(std::remove(top.begin(), top.end(), v), top.end());
      }
      if (kind == 2) {  // w layers alternate and are shifted by one row
        if (i % 2 == 1) { shifted(top, -1, L[sx + i]); shifted(top, 0, L[sx + 1 + i]); }
        else {
          shifted(bottom, 0, L[i]);
          if (i > 0) shifted(bottom, -1, L[i - 1]);
          else shifted(plane, -1, L[sx - 1]);
        }
      } else {
        const int isp = kind == 3;
        if (i < sx - isp) shifted(bottom, 0, L[i + isp]);
        shifted(top, 0, L[sx + 1 + i]);
      }
      if (i < sx - 1) {
        for (int& o : offset) o--;
        if (kind == 3) {
          if (offset[0] < 0) {
            active.push_back(active.back() + 1); active.erase(active.begin());
            offset.push_back(rowlen[active.back()]); offset.erase(offset.begin());
          }
        } else if (offset[0] < 0) { active.erase(active.begin()); offset.erase(offset.begin()); }
        else if (offset[0] == 0) { active.push_back(active.back() + 1); offset.push_back(rowlen[active.back()]); }
      }
    }
  }
  // trim the first/last layers exactly like the reference (:520-531) and merge per variable;
  // z index of layer l of kind k after trimming: u,v,p lose layer 0 => z = l+1; w keeps it => z = l
  std::vector<Pt> all;
  for (int var = 0; var < dof; var++)
    for (int kind = 0; kind < 4; kind++) {
      if (p.vtype[var] != kind_of[kind]) continue;
      const int lo = (kind == 2) ? 0 : 1, hi = 2 * sx - 1;  // inclusive layer range kept
      for (int l = lo; l <= hi; l++)
        for (auto& q : layers[kind][l]) all.push_back({q.x, q.y, l, var});
    }
  // group by the set of the 27 shifted copies that also contain the node
  std::unordered_map<int64_t, int> present;
  present.reserve(all.size() * 2);
  for (auto& q : all) present[key(q)] = 1;
  const int h = sx / 2;
  const int d1[3] = {h, h, 0}, d2[3] = {-h, h, sx}, d3[3] = {0, 0, sx};
  std::vector<std::array<int, 3>> shifts;
  {
    const int c1[3] = {0, -1, 1}, c2[3] = {0, -1, 1}, c3[3] = {0, -1, 1};
    for (int a : c1) for (int b : c2) for (int c : c3)
      shifts.push_back({a * d1[0] + b * d2[0] + c * d3[0], a * d1[1] + b * d2[1] + c * d3[1], a * d1[2] + b * d2[2] + c * d3[2]});
  }
  // order of first appearance follows the reference: nodes sorted by (z, y, x, var)
  std::sort(all.begin(), all.end(), [](const Pt& a, const Pt& b) {
    if (a.z != b.z) return a.z < b.z;
    if (a.y != b.y) return a.y < b.y;
    if (a.x != b.x) return a.x < b.x;
    return a.var < b.var;
  });
  std::vector<uint32_t> masks;
  masks.push_back(1);
  T.groups.emplace_back(1);
  for (auto& q : all) {
    uint32_t m = 0;
    for (int s = 0; s < 27; s++) {
      Pt r{q.x - shifts[s][0], q.y - shifts[s][1], q.z - shifts[s][2], q.var};
      if (present.count(key(r))) m |= 1u << s;
    }
    size_t g = 0;
    for (; g < masks.size(); g++) if (masks[g] == m) break;
    if (g == masks.size()) { masks.push_back(m); T.groups.emplace_back(dof); }
    if (g == 0) T.groups[0][0].push_back(q);
    else T.groups[g][q.var].push_back(q);
  }
  return T;
}

struct Layout {
  int npx, npy, npz;
};

inline void skew_position(const Params& p, int sd, int& x, int& y, int& z) {
  const int npx = p.nx / p.sx, npy = p.ny / p.sy;
  const int per_layer = 2 * npx * npy + npx + npy, per_row = 2 * npx + 1;
  const int Z = per_layer > 0 ? sd / per_layer : 0;
  int Y = ((sd - Z * per_layer) / per_row) * 2 - 1;
  int X = ((sd - Z * per_layer) % per_row) * 2;
  if (X >= npx * 2) { X -= npx * 2 + 1; Y += 1; }
  x = (X * p.sx) / 2;
  y = (Y * p.sx) / 2 + p.sx / 2;   // C++ truncation of negative odd products matches the reference
  z = Z * p.sx;
}

inline int skew_owner(const Params& p, int x, int y, int z) {
  const int sx = p.sx, npx = p.nx / sx, npy = p.ny / p.sy;
  const int dir1 = npx + 1, dir2 = npx, dir3 = 2 * npx * npy + npx + npy;
  const int xc = x / sx, yc = y / sx, zc = z / sx;
  int sd = zc * dir3 + yc * (dir2 + dir1) + xc;
  x -= xc * sx - 1; y -= yc * sx; z -= zc * sx;
  const bool front = y < sx - x, right = y < x;
  const bool below = right ? (z <= sx + y - x) : (z <= y - x);
  if (!front) sd += dir1;
  if (!right) sd += dir2;
  if (!below) sd += dir3;
  // across a periodic boundary the subdomain is the one at the other end (reference :199-206)
  if (!front && right && p.perio[0] && xc == npx - 1) sd -= dir2;
  if (!front && !right && p.perio[1] && yc == npy - 1) sd -= dir3 - dir2;
  if (!below && p.perio[2] && zc == p.nz / p.sz - 1) sd -= (p.nz / p.sz) * dir3;
  return sd;
}

const SkewTemplate& cached_template(const Params& p) {
  static std::map<std::vector<int>, SkewTemplate> cache;
  std::vector<int> k = {p.sx, p.dof, p.nz > 1};
  k.insert(k.end(), p.vtype.begin(), p.vtype.end());
  auto it = cache.find(k);
  if (it == cache.end()) it = cache.emplace(k, build_template(p)).first;
  return it->second;
}

}  // namespace

int skew_num_subdomains(const Params& p) {
  HYMLS_CHECK(p.sx == p.sy && (p.nz <= 1 || p.sx == p.sz), -2, "sx, sy and sz should be the same");
  HYMLS_CHECK(p.sx % 2 == 0, -2, "sx should be even");
  const int npx = p.nx / p.sx, npy = p.ny / p.sy, npz = p.nz / p.sz;
  HYMLS_CHECK(p.nx == npx * p.sx && p.ny == npy * p.sy && p.nz == npz * p.sz, -2,
              "the Skew Cartesian partitioner needs nx, ny, nz to be multiples of the separator length");
  const int per_layer = 2 * npx * npy + npx + npy;
  int n = per_layer;
  if (p.nz > 1) n += per_layer * npz;
  return std::max(n, 1);
}

void skew_sd_position(const Params& p, int sd, int& x, int& y, int& z) { skew_position(p, sd, x, y, z); }

void skew_get_groups(const Params& p, int sd, ivec& interior, std::vector<Group>& out) {
  interior.clear();
  out.clear();
  const SkewTemplate& T = cached_template(p);
  const int sx = p.sx, dof = p.dof;
  int sdx, sdy, sdz;
  skew_position(p, sd, sdx, sdy, sdz);
  // the periodic image of another subdomain (GetSubdomainPosition returns 1 and CreateSubdomainMap leaves it out,
  // reference :154-159,249-250): it stays in the numbering as an empty subdomain
  if ((sdx == p.nx - sx / 2 && p.perio[0]) || (sdy == p.ny && p.perio[1]) || (sdz == p.nz && p.perio[2])) return;
  // place the template: the auxiliary origin sits at (sdx - 1, sdy - 1 - sx/2, sdz - sx) after
  // removing the (sx, sx, sx) shift the reference applies before grouping (:585, :683-685)
  const int ox = sdx - 1, oy = sdy - 1 - sx / 2, oz = sdz - sx;
  auto place = [&](const Pt& q, int32_t& gid, int& x, int& y, int& z) {
    x = q.x + ox; y = q.y + oy; z = q.z + oz;
    if (p.perio[0]) x = (x + p.nx) % p.nx;
    if (p.perio[1]) y = (y + p.ny) % p.ny;
    if (p.perio[2]) z = (z + p.nz) % p.nz;
    if (x < 0 || x >= p.nx || y < 0 || y >= p.ny || z < 0 || z >= p.nz) return false;
    gid = ((z * p.ny + y) * p.nx + x) * dof + q.var;
    return true;
  };
  std::vector<std::vector<ivec>> groups;
  for (auto& cat : T.groups) {
    groups.emplace_back();
    for (auto& g : cat) {
      groups.back().emplace_back();
      for (auto& q : g) {
        int32_t gid; int x, y, z;
        if (place(q, gid, x, y, z)) groups.back().back().push_back(gid);
      }
    }
  }
  // first pressure(s) of the interior become single-node separator groups
  int retained = 0;
  for (size_t t = 0; t < groups[0][0].size() && retained < std::max(p.retain_pressures, 1); t++) {
    const int32_t node = groups[0][0][t];
    if (p.vtype[node % dof] == VT_P) {
      groups.push_back({ivec{node}});
      groups[0][0].erase(groups[0][0].begin() + t);
      t--;
      retained++;
    }
  }
  interior = groups[0][0];
  int type = 1;
  for (size_t i = 1; i < groups.size(); i++) {
    type++;
    for (auto& g : groups[i]) {
      std::map<int, ivec> by_owner;  // staggered w-groups can straddle two subdomains
      for (int32_t node : g) {
        const int c = node / dof;
        by_owner[skew_owner(p, c % p.nx, (c / p.nx) % p.ny, c / (p.nx * p.ny))].push_back(node);
      }
      for (auto& kv : by_owner) {
        out.emplace_back();
        out.back().type = p.link_velocities ? type : -1;
        out.back().nodes = kv.second;
      }
    }
  }
  // velocities on the closing walls of the domain do not border another subdomain
  for (auto& g : out) {
    ivec keep;
    for (int32_t node : g.nodes) {
      const int var = node % dof, c = node / dof;
      const int x = c % p.nx, y = (c / p.nx) % p.ny, z = c / (p.nx * p.ny);
      const int32_t vt = p.vtype[var];
      const bool wall = dof > 1 && ((x == p.nx - 1 && vt == VT_U && !p.perio[0]) || (y == p.ny - 1 && vt == VT_V && !p.perio[1]) ||
                                    (p.nz > 1 && z == p.nz - 1 && vt == VT_W && !p.perio[2]));
      if (!wall) { keep.push_back(node); continue; }
      if (skew_owner(p, x, y, z) == sd) interior.push_back(node);
    }
    g.nodes.swap(keep);
  }
  out.erase(std::remove_if(out.begin(), out.end(), [](const Group& g) { return g.nodes.empty(); }), out.end());
}

}  // namespace hymls
