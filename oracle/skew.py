"""Skew Cartesian partitioner (oracle side).  TEST INFRASTRUCTURE ONLY.

Procedural restatement of the reference's template construction, used to pin the
product's closed-form version:
  src/HYMLS_SkewCartesianPartitioner.cpp:27-76    buildPlane45
  src/HYMLS_SkewCartesianPartitioner.cpp:128-165  GetSubdomainPosition
  src/HYMLS_SkewCartesianPartitioner.cpp:167-213  GetSubdomainID
  src/HYMLS_SkewCartesianPartitioner.cpp:349-563  getTemplate
  src/HYMLS_SkewCartesianPartitioner.cpp:565-651  solveGroups
  src/HYMLS_SkewCartesianPartitioner.cpp:653-812  GetGroups
Periodic grids included (x/y/z-periodic: :154-159 duplicate subdomains, :199-206 ids across the periodic
boundary, :686-688 wrap-around, :783,791,799 no wall velocities); Retain Nodes <= 1.
"""
import numpy as np
from .partition import Params, VEL_U, VEL_V, VEL_W, PRESSURE


def _build_plane45(first, length, dirX, dirY, typ):
    left = right = first
    height = 2 * length
    extra = False
    dir1 = dirY + dirX
    dir2 = dirY - dirX
    if typ == 0:
        left -= dirX
        height += 1
        extra = True
    elif typ == 3:
        height += 1
        extra = True
    plane, ptr = [], [0]
    for i in range(height - 1):
        plane.extend(range(left, right + 1, dirX))
        ptr.append(len(plane))
        if i < length - 1:
            left += dir2
            right += dir1
        elif extra and i == length - 1:
            left += dirY
            right += dirY
        else:
            left += dir1
            right += dir2
    return plane, ptr


class SkewPartitioner:
    def __init__(self, params: Params):
        p = self.p = params
        if p.sx != p.sy or (p.nz > 1 and p.sx != p.sz):
            raise ValueError("sx, sy and sz should be the same")
        if p.sx % 2:
            raise ValueError("sx should be even")
        self.npx, self.npy, self.npz = p.nx // p.sx, p.ny // p.sy, p.nz // p.sz
        if p.nx != self.npx * p.sx or p.ny != self.npy * p.sy or p.nz != self.npz * p.sz:
            raise ValueError("grid is not a multiple of the subdomain size")
        self._template()
        self._solve_groups()

    # -- subdomain numbering
    def num_subdomains(self):
        per_layer = 2 * self.npx * self.npy + self.npx + self.npy
        n = per_layer
        if self.p.nz > 1:
            n += per_layer * self.npz
        return max(n, 1)

    def position(self, sd):
        sx = self.p.sx
        npx, npy = self.npx, self.npy
        per_layer = 2 * npx * npy + npx + npy
        per_row = 2 * npx + 1
        Z = sd // per_layer if per_layer > 0 else 0
        Y = ((sd - Z * per_layer) // per_row) * 2 - 1
        X = ((sd - Z * per_layer) % per_row) * 2
        if X >= npx * 2:
            X -= npx * 2 + 1
            Y += 1
        return (X * sx) // 2, (Y * sx) // 2 + sx // 2, Z * sx

    def skipped(self, sd):
        """GetSubdomainPosition returns 1 (:154-159): the subdomain is the periodic image of another one"""
        p = self.p
        x, y, z = self.position(sd)
        return bool((x == p.nx - p.sx // 2 and p.perio[0]) or (y == p.ny and p.perio[1]) or (z == p.nz and p.perio[2]))

    def subdomain_id(self, x, y, z):
        sx = self.p.sx
        npx, npy = self.npx, self.npy
        dir1, dir2, dir3 = npx + 1, npx, 2 * npx * npy + npx + npy
        xc, yc, zc = x // sx, y // sx, z // sx
        sd = zc * dir3 + yc * (dir2 + dir1) + xc
        x = x - (xc * sx - 1)
        y = y - yc * sx
        z = z - zc * sx
        front = y < sx - x
        right = y < x
        below = z <= y - x
        if right:
            below = z <= sx + y - x
        if not front:
            sd += dir1
        if not right:
            sd += dir2
        if not below:
            sd += dir3
        # across a periodic boundary the subdomain is the one at the other end (:199-206)
        perio = self.p.perio
        if not front and right and perio[0] and xc == npx - 1:
            sd -= dir2
        if not front and not right and perio[1] and yc == npy - 1:
            sd -= dir3 - dir2
        if not below and perio[2] and zc == self.npz - 1:
            sd -= self.npz * dir3
        return sd

    # -- template
    def _template(self):
        p = self.p
        sx, dof = p.sx, p.dof
        nx = sx * 4
        dirX, dirY, dirZ = dof, dof * nx, dof * nx * nx
        first = [dof * sx // 2 + dirY + dirZ * sx, dof * sx // 2 + dirZ * sx,
                 dof * sx // 2 + dirY + dirZ * sx, dof * sx // 2 + dirY + dirZ * sx]
        base_len = [sx // 2, sx // 2 + 1, sx // 2 + 1, sx // 2]
        type_array = [VEL_U, VEL_V, VEL_W, PRESSURE]
        nodes = []
        for typ in range(4):
            lay = [[] for _ in range(2 * sx + 1)]
            nodes.append(lay)
            plane, ptr = _build_plane45(first[typ], base_len[typ], dirX, dirY, typ)
            lay[sx] = list(plane)
            if p.nz <= 1:
                continue
            bottom = []
            top = list(plane)
            row_len = [ptr[i + 1] - ptr[i] - 1 for i in range(len(ptr) - 1)]
            active = list(range(base_len[typ]))
            offset = [row_len[i] for i in active]
            for i in range(sx):
                for j in range(len(active)):
                    val = plane[ptr[active[j]] + offset[j]]
                    bottom.append(val)
                    top = [t for t in top if t != val]
                if type_array[typ] == VEL_W:
                    if i % 2 == 1:
                        lay[sx + i].extend(j + i * dirZ - dirY for j in top)
                        lay[sx + 1 + i].extend(j + (i + 1) * dirZ for j in top)
                    else:
                        lay[i].extend(j - (sx - i) * dirZ for j in bottom)
                        if i > 0:
                            lay[i - 1].extend(j - (sx - i + 1) * dirZ - dirY for j in bottom)
                        else:
                            lay[sx - 1].extend(j - dirZ - dirY for j in plane)
                else:
                    is_p = 1 if type_array[typ] == PRESSURE else 0
                    if i < sx - is_p:
                        lay[i + is_p].extend(j - (sx - i - is_p) * dirZ for j in bottom)
                    lay[sx + 1 + i].extend(j + (i + 1) * dirZ for j in top)
                if i < sx - 1:
                    offset = [o - 1 for o in offset]
                    if type_array[typ] == PRESSURE:
                        if offset[0] < 0:
                            active.append(active[-1] + 1)
                            active.pop(0)
                            offset.append(row_len[active[-1]])
                            offset.pop(0)
                    else:
                        if offset[0] < 0:
                            active.pop(0)
                            offset.pop(0)
                        elif offset[0] == 0:
                            active.append(active[-1] + 1)
                            offset.append(row_len[active[-1]])
        nodes[0] = nodes[0][1:-1]
        nodes[1] = nodes[1][1:-1]
        nodes[2] = nodes[2][:-1]
        nodes[3] = nodes[3][1:-1]
        template = [[]]
        for i in range(dof):
            if p.variable_types[i] == VEL_W:
                template[-1] = [d + i for d in nodes[2][0]]
                nodes[2] = nodes[2][1:]
                break
        for j in range(2 * sx - 1):
            lay = []
            for i in range(dof):
                for typ in range(4):
                    if p.variable_types[i] == type_array[typ]:
                        lay.extend(d + i for d in nodes[typ][j])
            template.append(sorted(lay))
        self.template = template

    def _solve_groups(self):
        p = self.p
        sx, dof = p.sx, p.dof
        nx = sx * 4
        dirX, dirY, dirZ = dof * sx, dof * nx * sx, dof * nx * nx * sx
        first = dirX + dirY + dirZ
        dir1 = (dirY + dirX) // 2
        dir2 = (dirY - dirX) // 2 + dirZ
        dir3 = dirZ
        positions = [0, -dir3, dir3, -dir2, -dir2 - dir3, -dir2 + dir3, dir2, dir2 - dir3, dir2 + dir3,
                     -dir1, -dir1 - dir3, -dir1 + dir3, -dir1 - dir2, -dir1 - dir2 - dir3, -dir1 - dir2 + dir3,
                     -dir1 + dir2, -dir1 + dir2 - dir3, -dir1 + dir2 + dir3, dir1, dir1 - dir3, dir1 + dir3,
                     dir1 - dir2, dir1 - dir2 - dir3, dir1 - dir2 + dir3, dir1 + dir2, dir1 + dir2 - dir3,
                     dir1 + dir2 + dir3]
        temp = [x + first for lay in self.template for x in lay]
        tset = set(temp)
        groups, gdom = [[]], [1]
        for node in temp:
            mask = 0
            for i, pos in enumerate(positions):
                if node - pos in tset:
                    mask += 1 << i
            for gi, m in enumerate(gdom):
                if m == mask:
                    groups[gi].append(node)
                    break
            else:
                groups.append([node])
                gdom.append(mask)
        out = [[groups[0]]]
        for g in groups[1:]:
            cat = [[] for _ in range(dof)]
            for node in g:
                cat[node % dof].append(node)
            out.append(cat)
        self.template_groups = out

    def get_groups(self, sd):
        p = self.p
        sx, dof = p.sx, p.dof
        sdx, sdy, sdz = self.position(sd)
        nx = 4 * sx
        groups = []
        for cat in self.template_groups:
            gc = []
            for group in cat:
                g = []
                for node in group:
                    var = node % dof
                    x = (node // dof) % nx + sdx - 1 - sx
                    y = (node // dof // nx) % nx + sdy - 1 - 3 * sx // 2
                    z = node // dof // nx // nx + sdz - 2 * sx
                    if p.perio[0]:
                        x = (x + p.nx) % p.nx
                    if p.perio[1]:
                        y = (y + p.ny) % p.ny
                    if p.perio[2]:
                        z = (z + p.nz) % p.nz
                    if 0 <= x < p.nx and 0 <= y < p.ny and 0 <= z < p.nz:
                        g.append(x * dof + p.nx * y * dof + p.nx * p.ny * z * dof + var)
                gc.append(g)
            groups.append(gc)
        retained = 0
        for node in list(groups[0][0]):
            if p.variable_types[node % dof] == PRESSURE:
                groups.append([[node]])
                groups[0][0].remove(node)
                retained += 1
                if retained >= p.retain_pressures:
                    break
        interior = list(groups[0][0])
        seps = []
        typ = 1
        for i in range(1, len(groups)):
            typ += 1
            for group in groups[i]:
                new = {}
                for node in group:
                    cell = node // dof
                    gsd = self.subdomain_id(cell % p.nx, (cell // p.nx) % p.ny, cell // (p.nx * p.ny))
                    new.setdefault(gsd, []).append(node)
                for gsd in sorted(new):
                    seps.append([typ if p.link_velocities else -1, new[gsd]])
        for grp in seps:
            for node in list(grp[1]):
                var = node % dof
                cell = node // dof
                x, y, z = cell % p.nx, (cell // p.nx) % p.ny, cell // (p.nx * p.ny)
                vt = p.variable_types[var]
                if dof > 1 and ((x == p.nx - 1 and vt == VEL_U and not p.perio[0]) or
                                (y == p.ny - 1 and vt == VEL_V and not p.perio[1]) or
                                (p.nz > 1 and z == p.nz - 1 and vt == VEL_W and not p.perio[2])):
                    if self.subdomain_id(x, y, z) == sd:
                        interior.append(node)
                    grp[1].remove(node)
        return interior, [(t, g) for (t, g) in seps]
