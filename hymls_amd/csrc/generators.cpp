// generators.cpp -- synthetic input matrices of the benchmark configurations.
// Behaviour of the reference's GaleriExt::Stokes3D on a C grid, without periodicity and (separate row functions
// below) with periodic directions
// (src/GaleriExt_Stokes3D.h:89-285 on top of Darcy3D, src/GaleriExt_Darcy3D.h:45-176) and of
// Galeri "Laplace3D" scaled by -1 (src/HYMLS_MainUtils.cpp:260-348).  Written row by row so a
// 256^3 problem (67 M rows) streams straight into the caller's CSR arrays, and so that a rank of a
// sharded run can generate just the rows it needs.
#include "common.hpp"

namespace hymls {

namespace {
using Row = std::vector<std::pair<int32_t, double>>;

inline void laplace_row(int nx, int ny, int nz, int32_t r, Row& row) {
  row.clear();
  const int i = r % nx, j = (r / nx) % ny, k = r / (nx * ny);
  if (k > 0) row.emplace_back(r - nx * ny, 1.0);
  if (j > 0) row.emplace_back(r - nx, 1.0);
  if (i > 0) row.emplace_back(r - 1, 1.0);
  row.emplace_back(r, -6.0);
  if (i < nx - 1) row.emplace_back(r + 1, 1.0);
  if (j < ny - 1) row.emplace_back(r + nx, 1.0);
  if (k < nz - 1) row.emplace_back(r + nx * ny, 1.0);
}

inline void stokes_row(int nx, int ny, int nz, double a, double b, int32_t r, Row& row) {
  row.clear();
  const int dof = 4;
  const int var = r % dof, c = r / dof;
  const int ijk[3] = {c % nx, (c / nx) % ny, c / (nx * ny)};
  const int n[3] = {nx, ny, nz};
  const int step[3] = {1, nx, nx * ny};
  if (var < 3) {
    const bool on_wall = ijk[var] == n[var] - 1;  // velocity sits on the closing wall
    if (on_wall) {
      row.emplace_back(r, 1.0);  // Dirichlet row
    } else {
      // gradient: +b at this cell's pressure, -b at the next one
      row.emplace_back(c * dof + 3, b);
      row.emplace_back((c + step[var]) * dof + 3, -b);
      double diag = 6.0 * a;
      for (int d = 0; d < 3; d++) {
        if (d == var) continue;
        if (ijk[d] == 0 || ijk[d] == n[d] - 1) diag += a;  // wall-tangential correction
      }
      row.emplace_back(r, -diag);
      for (int d = 0; d < 3; d++) {
        if (ijk[d] > 0) row.emplace_back((c - step[d]) * dof + var, a);
        if (ijk[d] < n[d] - 1) {
          // the coupling to the velocity that lies on the wall is removed
          const bool nb_on_wall = (d == var) && (ijk[d] + 1 == n[d] - 1);
          if (!nb_on_wall) row.emplace_back((c + step[d]) * dof + var, a);
        }
      }
    }
  } else {
    for (int d = 0; d < 3; d++) {
      if (ijk[d] < n[d] - 1) row.emplace_back(c * dof + d, -b);
      if (ijk[d] > 0) row.emplace_back((c - step[d]) * dof + d, b);
    }
  }
  std::sort(row.begin(), row.end());
}

// GaleriExt::Darcy3D (reference src/GaleriExt_Darcy3D.h:45-176; create_matrix passes a = 1, b = -1,
// src/HYMLS_MainUtils.cpp:300-306): diag(a) on the velocities, -b / +b on this and the next cell's pressure,
// divergence rows with c = -b and no diagonal entry.
inline void darcy_row(int nx, int ny, int nz, double a, double b, int32_t r, Row& row) {
  row.clear();
  const int dof = 4;
  const int var = r % dof, c = r / dof;
  const int ijk[3] = {c % nx, (c / nx) % ny, c / (nx * ny)};
  const int n[3] = {nx, ny, nz};
  const int step[3] = {1, nx, nx * ny};
  if (var < 3) {
    row.emplace_back(r, a);
    if (ijk[var] < n[var] - 1) {
      row.emplace_back(c * dof + 3, -b);
      row.emplace_back((c + step[var]) * dof + 3, b);
    }
  } else {
    const double cc = -b;
    for (int d = 0; d < 3; d++) {
      if (ijk[d] < n[d] - 1) row.emplace_back(c * dof + d, -cc);
      if (ijk[d] > 0) row.emplace_back((c - step[d]) * dof + d, cc);
    }
  }
  std::sort(row.begin(), row.end());
}

// ---- periodic directions (reference src/GaleriExt_Periodic.cpp:8-66: the neighbour across a periodic boundary is the
// cell at the other end).  per[d]: direction d is periodic.
struct Nb {
  int n[3], step[3], ijk[3], c;
  const bool* per;
  Nb(int nx, int ny, int nz, int cell, const bool* p) : n{nx, ny, nz}, step{1, nx, nx * ny}, ijk{cell % nx, (cell / nx) % ny, cell / (nx * ny)}, c(cell), per(p) {}
  // neighbour in direction d, sign s (+1 next, -1 previous); -1 outside the box
  int plain(int d, int s) const { const int t = ijk[d] + s; return (t < 0 || t >= n[d]) ? -1 : c + s * step[d]; }
  int wrap(int d, int s) const {
    const int t = ijk[d] + s;
    if (t >= 0 && t < n[d]) return c + s * step[d];
    if (!per[d]) return -1;
    return c + (((t + n[d]) % n[d]) - ijk[d]) * step[d];
  }
};

// GaleriExt::Stokes3D with perio != NO_PERIO, restated as written (quirk included): gradient, divergence and every
// wall decision use the periodic neighbours; the velocity Laplacians are GaleriExt Cross3DN matrices
// (get3DLaplaceMatrixForVar, src/GaleriExt_Stokes3D.h:77-80; src/GaleriExt_Cross3DN.h:55-134), which do not know the
// periodicity: no coupling across a periodic boundary, every missing neighbour's -1 goes to the diagonal.
inline void stokes_row_periodic(int nx, int ny, int nz, double a, double b, const bool* per, int32_t r, Row& row) {
  row.clear();
  const int dof = 4;
  const int var = r % dof, c = r / dof;
  const Nb N(nx, ny, nz, c, per);
  if (var < 3) {
    const int nxt = N.wrap(var, +1);
    if (nxt < 0) { row.emplace_back(r, 1.0); return; }   // velocity on a closing wall: Dirichlet row
    row.emplace_back(c * dof + 3, b);
    row.emplace_back(nxt * dof + 3, -b);
    double lap = 6.0, add = 0.0;
    for (int d = 0; d < 3; d++) for (int s = -1; s <= 1; s += 2) if (N.plain(d, s) < 0) lap -= 1.0;
    for (int d = 0; d < 3; d++) {
      if (d == var) continue;
      if (N.wrap(d, -1) < 0 || N.wrap(d, +1) < 0) add += a;
    }
    row.emplace_back(r, -(lap * a + add));
    const Nb NX(nx, ny, nz, nxt, per);
    const bool nb_on_wall = NX.wrap(var, +1) < 0;          // the next velocity lies on the wall: coupling removed
    for (int d = 0; d < 3; d++)
      for (int s = -1; s <= 1; s += 2) {
        const int q = N.plain(d, s);
        if (q < 0 || (d == var && s == +1 && nb_on_wall)) continue;
        row.emplace_back(q * dof + var, a);
      }
  } else {
    for (int d = 0; d < 3; d++) {
      if (N.wrap(d, +1) >= 0) row.emplace_back(c * dof + d, -b);
      const int prv = N.wrap(d, -1);
      if (prv >= 0) row.emplace_back(prv * dof + d, b);
    }
  }
  std::sort(row.begin(), row.end());
}

inline void darcy_row_periodic(int nx, int ny, int nz, double a, double b, const bool* per, int32_t r, Row& row) {
  row.clear();
  const int dof = 4;
  const int var = r % dof, c = r / dof;
  const Nb N(nx, ny, nz, c, per);
  if (var < 3) {
    row.emplace_back(r, a);
    const int nxt = N.wrap(var, +1);
    if (nxt >= 0) { row.emplace_back(c * dof + 3, -b); row.emplace_back(nxt * dof + 3, b); }
  } else {
    const double cc = -b;
    for (int d = 0; d < 3; d++) {
      if (N.wrap(d, +1) >= 0) row.emplace_back(c * dof + d, -cc);
      const int prv = N.wrap(d, -1);
      if (prv >= 0) row.emplace_back(prv * dof + d, cc);
    }
  }
  std::sort(row.begin(), row.end());
}

// BASELINE configs[3]: a Navier-Stokes-like (Oseen) Jacobian.  The reference ships no 3D Jacobian at Re > 0
// (testSuite/cavity3D.xml reads a file that is a missing blob), so the matrix is synthesised:
// Stokes3D(a, b) plus the central difference of (w . grad) u on every existing velocity-velocity coupling, scaled
// like the diffusion (the Stokes rows are multiplied by Re): +-g w_d towards the next / previous neighbour in
// direction d with g = a Re / (2 nx), i.e. a cell Peclet number of Re |w| h / 2.  w is a fixed swirling field,
// w = (-y + 0.3 z, x - 0.2 z, 0.5 x y) at the cell centre, x = (i + 1/2) / nx - 1/2 etc.  Gradient and divergence
// entries are untouched (the matrix stays an F-matrix), the pattern is that of Stokes3D.
inline void oseen_row(int nx, int ny, int nz, double a, double b, double re, int32_t r, Row& row) {
  stokes_row(nx, ny, nz, a, b, r, row);
  const int dof = 4;
  const int var = r % dof, c = r / dof;
  if (var == 3 || row.size() == 1) return;
  const int i = c % nx, j = (c / nx) % ny, k = c / (nx * ny);
  const double x = (i + 0.5) / nx - 0.5, y = (j + 0.5) / ny - 0.5, z = (k + 0.5) / nz - 0.5;
  const double w[3] = {-y + 0.3 * z, x - 0.2 * z, (0.5 * x) * y};
  const double g = re / (2.0 * nx) * a;
  const int step[3] = {1, nx, nx * ny};
  for (auto& e : row) {
    if (e.first % dof != var || e.first == r) continue;
    const int diff = e.first / dof - c;
    for (int d = 0; d < 3; d++) {
      if (diff == step[d]) e.second += g * w[d];
      else if (diff == -step[d]) e.second += (-g) * w[d];
    }
  }
}
}  // namespace

// rows `gids` (nullptr: all rows 0..nrows-1) of the matrix (equations: 0 Laplace, 1 Stokes3D, 2 Darcy3D,
// 3 Stokes3D + convection at Reynolds number re); returns nnz; if rowptr != nullptr also fills
// the arrays (sorted global columns per row)
int64_t generate_rows(int equations, int nx, int ny, int nz, double a, double b, int64_t nrows, const int32_t* gids,
                      int32_t* rowptr, int32_t* col, double* val, double re, const bool* per) {
  const bool periodic = per && (per[0] || per[1] || per[2]);
  HYMLS_CHECK(!periodic || equations == 1 || equations == 2, -99, "periodic directions: Stokes3D and Darcy3D only");
  HYMLS_CHECK(!periodic || ((!per[0] || nx >= 3) && (!per[1] || ny >= 3) && (!per[2] || nz >= 3)), -2, "a periodic direction needs at least 3 cells");
  Row row;
  int64_t nnz = 0;
  for (int64_t t = 0; t < nrows; t++) {
    const int32_t r = gids ? gids[t] : (int32_t)t;
    if (periodic && equations == 1) stokes_row_periodic(nx, ny, nz, a, b, per, r, row);
    else if (periodic) darcy_row_periodic(nx, ny, nz, a, b, per, r, row);
    else if (equations == 0) laplace_row(nx, ny, nz, r, row);
    else if (equations == 1) stokes_row(nx, ny, nz, a, b, r, row);
    else if (equations == 2) darcy_row(nx, ny, nz, a, b, r, row);
    else oseen_row(nx, ny, nz, a, b, re, r, row);
    if (rowptr) {
      rowptr[t] = (int32_t)nnz;
      for (auto& e : row) { col[nnz] = e.first; val[nnz] = e.second; nnz++; }
    } else {
      nnz += (int64_t)row.size();
    }
  }
  HYMLS_CHECK(nnz < (int64_t)1 << 31, -2, "more than 2^31 matrix entries on one rank need 64-bit row pointers");
  if (rowptr) rowptr[nrows] = (int32_t)nnz;
  return nnz;
}

int64_t generate_laplace3d(int nx, int ny, int nz, int32_t* rowptr, int32_t* col, double* val) {
  return generate_rows(0, nx, ny, nz, 0, 0, (int64_t)nx * ny * nz, nullptr, rowptr, col, val, 0.0, nullptr);
}

int64_t generate_stokes3d(int nx, int ny, int nz, double a, double b, int32_t* rowptr, int32_t* col, double* val) {
  return generate_rows(1, nx, ny, nz, a, b, (int64_t)nx * ny * nz * 4, nullptr, rowptr, col, val, 0.0, nullptr);
}

}  // namespace hymls
