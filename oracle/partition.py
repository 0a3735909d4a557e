"""Cartesian partitioner + hierarchical map (oracle side).  TEST INFRASTRUCTURE ONLY.

Restates, for the serial (one rank) case:
  src/HYMLS_BasePartitioner.cpp:31-346      SetParameters / SetNextLevelParameters
  src/HYMLS_CartesianPartitioner.cpp:80-121 subdomain id <-> position
  src/HYMLS_CartesianPartitioner.cpp:224-408 GetSubdomainStartAndEnd / GetGroups
  src/HYMLS_OverlappingPartitioner.cpp:121-147 DetectSeparators
  src/HYMLS_HierarchicalMap.cpp:120-285     LinkSeparators / FillComplete
"""
from dataclasses import dataclass, field, replace
import numpy as np

VEL_U, VEL_V, VEL_W, PRESSURE, INTERIOR = "U", "V", "W", "P", "I"


@dataclass
class Params:
    nx: int
    ny: int
    nz: int
    dof: int = 1
    dim: int = 3
    sx: int = 4
    sy: int = -1
    sz: int = -1
    cx: int = -1
    cy: int = -1
    cz: int = -1
    rx: int = -1  # "Retain Nodes" (finalize / next_level resolve them per level, BasePartitioner.cpp:108-137)
    ry: int = -1
    rz: int = -1
    retain_xyz: tuple = (-1, -1, -1)          # "Retain Nodes (x|y|z)"
    retain_at_level: dict = field(default_factory=dict)   # "Retain Nodes at Level k"
    level: int = 0
    retain: int = None
    levels: int = 1  # XML "Number of Levels" (0-based, SURVEY top)
    variable_types: list = field(default_factory=list)
    retain_pressures: int = 1
    link_velocities: bool = True  # BasePartitioner.cpp:141
    link_retained: bool = True    # BasePartitioner.cpp:139
    fix_gids: list = field(default_factory=list)
    equations: str = "Laplace"
    partitioner: str = "Cartesian"  # or "Skew Cartesian"
    perio: tuple = (False, False, False)   # "x-periodic", "y-periodic", "z-periodic" (BasePartitioner.cpp:49-62)

    def finalize(self):
        """BasePartitioner::SetParameters defaults (BasePartitioner.cpp:70-252)."""
        p = replace(self)
        if p.nz == 1 and p.dim > 2:
            p.dim = 2
        if p.sy == -1:
            p.sy = p.sx
        if p.sz == -1:
            p.sz = p.sx if p.nz > 1 else 1
        if p.cx == -1:
            p.cx = p.sx
        if p.cy == -1:
            p.cy = p.cx
        if p.cz == -1:
            p.cz = p.cx if p.nz > 1 else 1
        if p.retain is None:
            p.retain = p.rx            # "Retain Nodes"
        p._set_retain()
        if not p.variable_types:
            if p.equations == "Laplace":
                p.dof = 1
                p.variable_types = [VEL_V]  # "Laplace" maps to Velocity_V (:272)
            elif p.equations.startswith("Stokes"):
                p.dof = p.dim + 1
                p.variable_types = [VEL_U, VEL_V, VEL_W][: p.dim] + [PRESSURE]
                if not p.fix_gids:
                    p.fix_gids = [p.dim]  # "Fix GID 1" = pvar (:237-243)
            else:
                raise ValueError("'Equations' parameter not recognized")
        return p

    def _set_retain(self):
        at = self.retain_at_level.get(self.level, -1)
        r = [self.retain_xyz[d] if self.retain_xyz[d] != -1 else (at if at != -1 else self.retain) for d in range(3)]
        self.rx, self.ry, self.rz = r

    def next_level(self):
        """SetNextLevelParameters (BasePartitioner.cpp:321-346); the partitioner of the next level reads its own
        "Retain Nodes at Level k" (:112)."""
        q = replace(self, sx=self.sx * self.cx, sy=self.sy * self.cy, sz=self.sz * self.cz, level=self.level + 1)
        q._set_retain()
        return q


def _start_end(pos, idx, idx_max, dim, mx, perio=False):
    """GetSubdomainStartAndEnd (CartesianPartitioner.cpp:224-263). Returns (skip,type,start,end)."""
    ln = max((mx + idx_max - 1) // idx_max, 1)
    if idx == idx_max:
        typ = 2
    elif idx >= 0:
        typ = 1
    else:
        typ = 0
    start = idx
    if idx == idx_max:
        start = mx
    elif idx > 0:
        start = min(ln * idx, mx)
    end = start + 1
    if typ == 1:
        end = min(ln * (idx + 1), mx)
    if not perio:
        if pos == 0 and idx == -1:
            return True, typ, start, end
        if pos + mx + 1 == dim:
            if idx == idx_max:
                return True, typ, start, end
            if idx == idx_max - 1:
                end += 1
    if start == end:
        return True, typ, start, end
    return False, typ, start, end


class CartesianPartitioner:
    def __init__(self, params: Params):
        self.p = params
        p = params
        self.npx = (p.nx - 1) // p.sx + 1
        self.npy = (p.ny - 1) // p.sy + 1
        self.npz = (p.nz - 1) // p.sz + 1

    def num_subdomains(self):
        return self.npx * self.npy * self.npz

    def position(self, sd):
        p = self.p
        return ((sd % self.npx) * p.sx, ((sd // self.npx) % self.npy) * p.sy,
                ((sd // self.npx // self.npy) % self.npz) * p.sz)

    def get_groups(self, sd):
        """GetGroups (CartesianPartitioner.cpp:265-408).
        Returns (interior gids [unsorted], [(type, gids)] separator groups)."""
        p = self.p
        xpos, ypos, zpos = self.position(sd)
        xmax = min(p.nx - xpos - 1, p.sx - 1)
        ymax = min(p.ny - ypos - 1, p.sy - 1)
        zmax = min(p.nz - zpos - 1, p.sz - 1)
        if xmax == 0 or ymax == 0 or (zmax == 0 and p.nz > 1):
            raise ValueError("Can't have a subdomain of size 1")
        imax = p.rx if p.rx > 1 else 1
        jmax = p.ry if p.ry > 1 else 1
        kmax = p.rz if p.rz > 1 else 1
        interior = []
        groups = []
        retained = []
        for kidx in range(-1, kmax + 1):
            kint = 0 <= kidx < kmax
            skip, ktype, ks, ke = _start_end(zpos, kidx, kmax, p.nz, zmax, p.perio[2])
            if skip:
                continue
            for jidx in range(-1, jmax + 1):
                jint = 0 <= jidx < jmax
                skip, jtype, js, je = _start_end(ypos, jidx, jmax, p.ny, ymax, p.perio[1])
                if skip:
                    continue
                for iidx in range(-1, imax + 1):
                    iint = 0 <= iidx < imax
                    skip, itype, is_, ie = _start_end(xpos, iidx, imax, p.nx, xmax, p.perio[0])
                    if skip:
                        continue
                    kk, jj, ii = np.meshgrid(np.arange(ks, ke), np.arange(js, je),
                                             np.arange(is_, ie), indexing="ij")
                    kk, jj, ii = kk.ravel(), jj.ravel(), ii.ravel()
                    base = (((ii + xpos + p.nx) % p.nx) * p.dof
                            + ((jj + ypos + p.ny) % p.ny) * p.nx * p.dof
                            + ((kk + zpos + p.nz) % p.nz) * p.nx * p.ny * p.dof)
                    for d in range(p.dof):
                        vt = p.variable_types[d]
                        if vt in (PRESSURE, INTERIOR) and (iidx == -1 or jidx == -1 or kidx == -1):
                            continue
                        gids = (base + d).tolist()
                        if ((iint and jint and kint) or vt == INTERIOR or
                                (vt == PRESSURE and ((iint and jint) or (iint and kint) or
                                                      (jint and kint) or p.retain_pressures > 1))):
                            target = interior
                        else:
                            typ = -1000
                            if p.link_retained:
                                typ = 2 * p.dof * (itype + 3 * (jtype + 3 * ktype))
                            if not (p.link_velocities and vt in (VEL_U, VEL_V, VEL_W)):
                                typ += 2 * d
                            target = []
                            groups.append((typ, target))
                        if vt == PRESSURE:
                            for g, i_, j_, k_ in zip(gids, ii, jj, kk):
                                if i_ >= 0 and j_ >= 0 and k_ >= 0 and len(retained) < p.retain_pressures:
                                    retained.append(g)
                                else:
                                    target.append(g)
                        else:
                            target.extend(gids)
        groups = [(t, g) for (t, g) in groups if g]
        for g in retained:
            groups.append((-1, [g]))
        return interior, groups


class HierarchicalMap:
    """Result of OverlappingPartitioner::DetectSeparators + HierarchicalMap::FillComplete
    for one level, one rank.

    interior[sd]  : sorted np.array of gids
    groups[sd]    : list of (type, sorted np.array) -- all separator groups around sd
    owned[sd]     : indices into groups[sd] of the groups this subdomain owns
                    (first subdomain that lists a group, identified by its first GID,
                    HierarchicalMap.cpp:261-271)
    linked[sd]    : list of lists of group indices with equal type >= 0 (LinkSeparators :120-142)
    """

    def __init__(self, params: Params, present=None):
        """present: boolean mask over all N gids that exist in the level's base map
        (None = all)."""
        self.p = params
        if params.partitioner == "Skew Cartesian":
            from .skew import SkewPartitioner
            part = SkewPartitioner(params)
        else:
            part = CartesianPartitioner(params)
        self.part = part
        nsd = part.num_subdomains()
        self.interior, self.groups, self.owned, self.linked = [], [], [], []
        seen = set()
        for sd in range(nsd):
            if getattr(part, "skipped", lambda sd: False)(sd):
                # a Skew subdomain that is the periodic image of another one (GetSubdomainPosition returns 1, the
                # reference's CreateSubdomainMap leaves it out, SkewCartesianPartitioner.cpp:154-159,249-250); kept
                # here as an EMPTY subdomain so that ids stay what GetSubdomainID computes
                inter, grps = [], []
            else:
                inter, grps = part.get_groups(sd)
            inter = np.array(sorted(inter), dtype=np.int64)
            if present is not None:
                inter = inter[present[inter]]
            out = []
            for (t, g) in grps:
                g = np.array(sorted(g), dtype=np.int64)
                if present is not None:
                    g = g[present[g]]
                if g.size:
                    out.append((t, g))
            own = []
            for gi, (t, g) in enumerate(out):
                if int(g[0]) not in seen:
                    seen.add(int(g[0]))
                    own.append(gi)
            self.interior.append(inter)
            self.groups.append(out)
            self.owned.append(own)
            self.linked.append(self._link(out, range(len(out))))
        self.nsd = nsd

    @staticmethod
    def _link(groups, idxs):
        linked = []
        for gi in idxs:
            t = groups[gi][0]
            found = False
            if t >= 0:
                for L in linked:
                    if groups[L[0]][0] == t:
                        L.append(gi)
                        found = True
                        break
            if not found:
                linked.append([gi])
        return linked

    def owned_linked(self, sd):
        """LinkSeparators on the LocalSeparators object (HierarchicalMap.cpp:507-522)."""
        return self._link(self.groups[sd], self.owned[sd])

    def interior_map(self):
        return np.concatenate(self.interior) if self.nsd else np.zeros(0, np.int64)

    def separator_map(self):
        parts = [self.groups[sd][gi][1] for sd in range(self.nsd) for gi in self.owned[sd]]
        return np.concatenate(parts) if parts else np.zeros(0, np.int64)

    def sd_separators(self, sd):
        """SpawnMap(sd, Separators): all groups of sd concatenated (HierarchicalMap.cpp:575-592)."""
        if not self.groups[sd]:
            return np.zeros(0, np.int64)
        return np.concatenate([g for (_, g) in self.groups[sd]])

    def vsum_map(self):
        """CreateVSumMap on the LocalSeparators object (SchurPreconditioner.cpp:469-518)."""
        return np.array([self.groups[sd][gi][1][0] for sd in range(self.nsd)
                         for gi in self.owned[sd]], dtype=np.int64)
