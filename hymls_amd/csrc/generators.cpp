// generators.cpp -- synthetic input matrices of the benchmark configurations.
// Behaviour of the reference's GaleriExt::Stokes3D on a C grid without periodicity
// (src/GaleriExt_Stokes3D.h:89-285 on top of Darcy3D, src/GaleriExt_Darcy3D.h:45-176) and of
// Galeri "Laplace3D" scaled by -1 (src/HYMLS_MainUtils.cpp:260-348).  Written row by row so a
// 256^3 problem (67 M rows) streams straight into the caller's CSR arrays.
#include "common.hpp"

namespace hymls {

namespace {
struct Grid {
  int nx, ny, nz;
  inline int id(int i, int j, int k) const { return (k * ny + j) * nx + i; }
};
}  // namespace

// returns nnz; if rowptr != nullptr also fills the arrays (sorted columns per row)
int64_t generate_laplace3d(int nx, int ny, int nz, int32_t* rowptr, int32_t* col, double* val) {
  Grid g{nx, ny, nz};
  int64_t nnz = 0;
  for (int k = 0; k < nz; k++)
    for (int j = 0; j < ny; j++)
      for (int i = 0; i < nx; i++) {
        const int r = g.id(i, j, k);
        if (rowptr) rowptr[r] = (int32_t)nnz;
        auto put = [&](int c, double v) { if (rowptr) { col[nnz] = c; val[nnz] = v; } nnz++; };
        if (k > 0) put(g.id(i, j, k - 1), 1.0);
        if (j > 0) put(g.id(i, j - 1, k), 1.0);
        if (i > 0) put(g.id(i - 1, j, k), 1.0);
        put(r, -6.0);
        if (i < nx - 1) put(g.id(i + 1, j, k), 1.0);
        if (j < ny - 1) put(g.id(i, j + 1, k), 1.0);
        if (k < nz - 1) put(g.id(i, j, k + 1), 1.0);
      }
  if (rowptr) rowptr[(int64_t)nx * ny * nz] = (int32_t)nnz;
  return nnz;
}

int64_t generate_stokes3d(int nx, int ny, int nz, double a, double b, int32_t* rowptr, int32_t* col, double* val) {
  Grid g{nx, ny, nz};
  const int dof = 4;
  int64_t nnz = 0;
  std::vector<std::pair<int32_t, double>> row;
  for (int k = 0; k < nz; k++)
    for (int j = 0; j < ny; j++)
      for (int i = 0; i < nx; i++) {
        const int c = g.id(i, j, k);
        const int ijk[3] = {i, j, k};
        const int n[3] = {nx, ny, nz};
        const int step[3] = {1, nx, nx * ny};
        for (int var = 0; var < 4; var++) {
          row.clear();
          const int r = c * dof + var;
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
          if (rowptr) {
            rowptr[r] = (int32_t)nnz;
            for (auto& e : row) { col[nnz] = e.first; val[nnz] = e.second; nnz++; }
          } else {
            nnz += (int64_t)row.size();
          }
        }
      }
  if (rowptr) rowptr[(int64_t)nx * ny * nz * dof] = (int32_t)nnz;
  return nnz;
}

}  // namespace hymls
