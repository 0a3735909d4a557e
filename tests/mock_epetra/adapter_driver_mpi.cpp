// The DISTRIBUTED drop-in: include/hymls_mi_epetra.hpp under an Epetra_MpiComm with a row-distributed Epetra_CrsMatrix,
// one MPI rank per shard -- the reference's deployment (src/main.cpp:48-67,330-334; src/HYMLS_Preconditioner.cpp:304-336).
// Run by tests/test_epetra_adapter.py as `mpiexec -n 2|4 adapter_driver_mpi <transport>`.  Everything goes through
// Ifpack_Preconditioner / Epetra_Operator pointers; the checks:
//   1. the matrix lives on Epetra's LINEAR map (contiguous chunks: deliberately not the boxes of the partitioner): the
//      adapter imports the overlapping rows itself; sharded ApplyInverse == the one-rank ApplyInverse (a second,
//      serial preconditioner on the full matrix in the same process) to 1e-12, 3 vectors;
//   2. SetMatrix + Compute with new values; error codes before Compute;
//   3. a CG loop that only sees Epetra_Operator::ApplyInverse / Apply and all-reduced dot products converges in the same
//      number of iterations as on one rank;
//   4. Stokes (Skew Cartesian, two-level) + a border [K V; V' 0] on the distributed operator.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <string>
#include "hymls_mi_epetra.hpp"

static int g_rank = 0;
#define REQUIRE(c) do { if (!(c)) { std::printf("[rank %d] FAILED %s:%d: %s\n", g_rank, __FILE__, __LINE__, #c); std::fflush(stdout); MPI_Abort(MPI_COMM_WORLD, 1); } } while (0)

// rows of the generated matrix that live on `map`, inserted with global column ids
static Teuchos::RCP<Epetra_CrsMatrix> make_matrix(int equations, int n, const Epetra_Map& map, double scale) {
  const int nloc = map.NumMyElements();
  std::vector<int32_t> gids(map.MyGlobalElements(), map.MyGlobalElements() + nloc);
  int64_t nnz = 0;
  REQUIRE(hymls_mi_generate_rows(equations, n, n, n, (double)n * n, 1.0, nloc, gids.data(), &nnz, 0, 0, 0) == 0);
  std::vector<int32_t> rp(nloc + 1), ci(nnz + 1);
  std::vector<double> va(nnz + 1);
  REQUIRE(hymls_mi_generate_rows(equations, n, n, n, (double)n * n, 1.0, nloc, gids.data(), &nnz, rp.data(), ci.data(), va.data()) == 0);
  Teuchos::RCP<Epetra_CrsMatrix> K = Teuchos::rcp(new Epetra_CrsMatrix(Copy, map, 7));
  for (int i = 0; i < nloc; i++) {
    std::vector<double> v(va.begin() + rp[i], va.begin() + rp[i + 1]);
    for (double& x : v) x *= scale;
    K->InsertGlobalValues(gids[i], rp[i + 1] - rp[i], v.data(), ci.data() + rp[i]);
  }
  K->FillComplete();
  return K;
}

// the same pseudo-random value for a global row on every rank
static double val(int gid, int k) { unsigned s = 2654435761u * (unsigned)(gid + 1) + 40503u * (unsigned)(k + 1); s ^= s >> 13; s *= 1274126177u; s ^= s >> 16; return (double)(s & 0xffffff) / (1 << 24) * 2.0 - 1.0; }

static Teuchos::RCP<Teuchos::ParameterList> params(const char* eq, int n, int sx, int levels, const std::string& transport, const char* part = "Cartesian") {
  Teuchos::RCP<Teuchos::ParameterList> p = Teuchos::rcp(new Teuchos::ParameterList());
  p->sublist("Problem").set("Equations", eq).set("Dimension", 3).set("nx", n).set("ny", n).set("nz", n);
  p->sublist("Preconditioner").set("Separator Length", sx).set("Number of Levels", levels).set("Partitioner", part).set("MI Transport", transport.c_str());
  return p;
}

// preconditioned CG on operators only; dot products through Epetra (all-reduced)
static int pcg(const Epetra_Operator& A, const Epetra_Operator& M, const Epetra_MultiVector& b, Epetra_MultiVector& x, double tol, int maxit) {
  const Epetra_BlockMap& map = b.Map();
  const int n = map.NumMyElements();
  Epetra_MultiVector r(map, 1), z(map, 1), p(map, 1), q(map, 1);
  for (int i = 0; i < n; i++) { r[0][i] = b[0][i]; x[0][i] = 0.0; }
  double bn = 0, rz = 0, rz1 = 0, pq = 0, rn = 0;
  b.Norm2(&bn);
  int it = 0;
  for (; it < maxit; it++) {
    REQUIRE(M.ApplyInverse(r, z) == 0);
    r.Dot(z, &rz1);
    for (int i = 0; i < n; i++) p[0][i] = it ? z[0][i] + rz1 / rz * p[0][i] : z[0][i];
    rz = rz1;
    REQUIRE(A.Apply(p, q) == 0);
    p.Dot(q, &pq);
    const double al = rz / pq;
    for (int i = 0; i < n; i++) { x[0][i] += al * p[0][i]; r[0][i] -= al * q[0][i]; }
    r.Norm2(&rn);
    if (rn <= tol * bn) { it++; break; }
  }
  return it;
}

int main(int argc, char** argv) {
  MPI_Init(&argc, &argv);
  const std::string transport = argc > 1 ? argv[1] : "MPI";
  {
    Epetra_MpiComm comm(MPI_COMM_WORLD);
    Epetra_SerialComm self;
    g_rank = comm.MyPID();
    const int P = comm.NumProc();
    // ---- 1.-3. Laplace 16^3, separator length 4, two-level (32^3 subdomain boxes split over 2 / 4 ranks)
    {
      const int n = 16, N = n * n * n;
      Epetra_Map map(N, 0, comm);          // LINEAR distribution
      Epetra_Map full(N, 0, self);         // the whole problem on this rank (the one-rank comparison)
      Teuchos::RCP<Epetra_CrsMatrix> K = make_matrix(0, n, map, -1.0), K1 = make_matrix(0, n, full, -1.0);
      Teuchos::RCP<HYMLS_MI::Preconditioner> P_ = Teuchos::rcp(new HYMLS_MI::Preconditioner(K, params("Laplace", n, 4, 1, transport)));
      Teuchos::RCP<HYMLS_MI::Preconditioner> P1 = Teuchos::rcp(new HYMLS_MI::Preconditioner(K1, params("Laplace", n, 4, 1, transport)));
      Ifpack_Preconditioner* prec = P_.get();
      const Epetra_Operator* op = prec;
      const int nloc = map.NumMyElements();
      Epetra_MultiVector B(map, 3), X(map, 3), B1(full, 3), X1(full, 3);
      for (int v = 0; v < 3; v++) { for (int i = 0; i < nloc; i++) B[v][i] = val(map.GID(i), v); for (int i = 0; i < N; i++) B1[v][i] = val(i, v); }
      REQUIRE(op->ApplyInverse(B, X) == -1);                       // before Compute (reference Preconditioner.cpp:936-939)
      REQUIRE(prec->Compute() == 0);                               // auto-initialises, collective
      REQUIRE(prec->IsInitialized() && prec->IsComputed() && prec->NumInitialize() == 1 && prec->NumCompute() == 1);
      REQUIRE(P1->Compute() == 0);
      REQUIRE(op->ApplyInverse(B, X) == 0 && P1->ApplyInverse(B1, X1) == 0);
      double err = 0, nrm = 0;
      for (int v = 0; v < 3; v++) for (int i = 0; i < nloc; i++) { err = std::max(err, std::abs(X[v][i] - X1[v][map.GID(i)])); nrm = std::max(nrm, std::abs(X1[v][map.GID(i)])); }
      if (g_rank == 0) std::printf("%d ranks (%s): sharded ApplyInverse through Ifpack_Preconditioner* vs one rank: max diff %.2e (max |x| %.2e)\n", P, transport.c_str(), err, nrm);
      REQUIRE(err <= 1e-12 * nrm);
      REQUIRE(op->OperatorDomainMap().SameAs(K->RowMap()) && op->OperatorRangeMap().SameAs(K->RowMap()) && &op->Comm() == &K->Comm());
      // SetMatrix, Compute without Initialize: twice the matrix, half the solution
      Teuchos::RCP<Epetra_CrsMatrix> K2 = make_matrix(0, n, map, -2.0);
      P_->SetMatrix(K2);
      REQUIRE(!prec->IsInitialized() && prec->Compute() == 0 && prec->NumInitialize() == 1 && prec->NumCompute() == 2);
      Epetra_MultiVector Y(map, 3);
      REQUIRE(op->ApplyInverse(B, Y) == 0);
      err = 0;
      for (int v = 0; v < 3; v++) for (int i = 0; i < nloc; i++) err = std::max(err, std::abs(2.0 * Y[v][i] - X[v][i]));
      REQUIRE(err <= 1e-12 * nrm);
      P_->SetMatrix(K);
      REQUIRE(prec->Compute() == 0);
      // CG that only sees the operators
      Epetra_MultiVector xex(map, 1), b(map, 1), x(map, 1), xex1(full, 1), b1(full, 1), x1(full, 1);
      for (int i = 0; i < nloc; i++) xex[0][i] = val(map.GID(i), 7);
      for (int i = 0; i < N; i++) xex1[0][i] = val(i, 7);
      REQUIRE(K->Apply(xex, b) == 0 && K1->Apply(xex1, b1) == 0);
      const int its = pcg(*K, *P_, b, x, 1e-10, 100), its1 = pcg(*K1, *P1, b1, x1, 1e-10, 100);
      err = 0;
      for (int i = 0; i < nloc; i++) err = std::max(err, std::abs(x[0][i] - xex[0][i]));
      if (g_rank == 0) std::printf("%d ranks: preconditioned CG through Epetra_Operator: %d iterations (one rank: %d), error %.2e\n", P, its, its1, err);
      REQUIRE(its == its1 && its < 40 && err < 1e-7);
    }
    // ---- 4. Stokes 16^3, Skew Cartesian, sx = 8, two-level (boxes of 8^3 cells need nx / px >= 8), bordered
    {
      const int n = 16, N = 4 * n * n * n;
      Epetra_Map map(N, 0, comm), full(N, 0, self);
      Teuchos::RCP<Epetra_CrsMatrix> K = make_matrix(1, n, map, 1.0), K1 = make_matrix(1, n, full, 1.0);
      // test vector: 0 on rows that only hold a diagonal entry (create_testvector, reference src/HYMLS_MainUtils.cpp:238-256)
      auto testvec = [](const Epetra_CrsMatrix& A, const Epetra_Map& m) {
        Teuchos::RCP<Epetra_Vector> tv = Teuchos::rcp(new Epetra_Vector(m));
        for (int i = 0; i < A.NumMyRows(); i++) {
          int len = 0; double* v = 0; int* c = 0; A.ExtractMyRowView(i, len, v, c);
          bool diag_only = true;
          for (int k = 0; k < len; k++) if (v[k] != 0.0 && A.GCID(c[k]) != A.GRID(i)) diag_only = false;
          (*tv)[i] = diag_only ? 0.0 : 1.0;
        }
        return tv;
      };
      HYMLS_MI::Preconditioner P_(K, params("Stokes-C", n, 8, 1, transport, "Skew Cartesian"), testvec(*K, map));
      HYMLS_MI::Preconditioner P1(K1, params("Stokes-C", n, 8, 1, transport, "Skew Cartesian"), testvec(*K1, full));
      REQUIRE(P_.Initialize() == 0 && P_.Compute() == 0 && P1.Compute() == 0);
      const int nloc = map.NumMyElements();
      Epetra_MultiVector B(map, 1), X(map, 1), B1(full, 1), X1(full, 1);
      for (int i = 0; i < nloc; i++) B[0][i] = val(map.GID(i), 3);
      for (int i = 0; i < N; i++) B1[0][i] = val(i, 3);
      REQUIRE(P_.ApplyInverse(B, X) == 0 && P1.ApplyInverse(B1, X1) == 0);
      double err = 0, nrm = 0;
      for (int i = 0; i < nloc; i++) { err = std::max(err, std::abs(X[0][i] - X1[0][map.GID(i)])); nrm = std::max(nrm, std::abs(X1[0][map.GID(i)])); }
      if (g_rank == 0) std::printf("%d ranks: Stokes 16^3 Skew two-level sharded vs one rank: max diff %.2e (max |x| %.2e)\n", P, err, nrm);
      REQUIRE(err <= 1e-10 * nrm);
      // border
      Teuchos::RCP<Epetra_MultiVector> V = Teuchos::rcp(new Epetra_MultiVector(map, 2)), V1 = Teuchos::rcp(new Epetra_MultiVector(full, 2));
      for (int v = 0; v < 2; v++) { for (int i = 0; i < nloc; i++) (*V)[v][i] = val(map.GID(i), 11 + v); for (int i = 0; i < N; i++) (*V1)[v][i] = val(i, 11 + v); }
      REQUIRE(P_.SetBorder(V) == 0 && P_.HaveBorder() && P_.Compute() == 0 && P1.SetBorder(V1) == 0 && P1.Compute() == 0);
      Epetra_SerialDenseMatrix T(2, 1), S(2, 1), S1(2, 1);
      T(0, 0) = 0.37; T(1, 0) = -1.2;
      REQUIRE(P_.ApplyInverse(B, T, X, S) == 0 && P1.ApplyInverse(B1, T, X1, S1) == 0);
      err = 0; nrm = 0;
      for (int i = 0; i < nloc; i++) { err = std::max(err, std::abs(X[0][i] - X1[0][map.GID(i)])); nrm = std::max(nrm, std::abs(X1[0][map.GID(i)])); }
      const double serr = std::max(std::abs(S(0, 0) - S1(0, 0)), std::abs(S(1, 0) - S1(1, 0))) / std::max(std::abs(S1(0, 0)), std::abs(S1(1, 0)));
      if (g_rank == 0) std::printf("%d ranks: bordered ApplyInverse sharded vs one rank: x %.2e, s %.2e\n", P, err / nrm, serr);
      REQUIRE(err <= 1e-9 * nrm && serr <= 1e-9);
    }
    // ---- 5. (argv[2] = "periodic") x-periodic Stokes channel 16 x 8 x 8, Skew Cartesian: the halo of a rank's box wraps around
    if (argc > 2 && std::string(argv[2]) == "periodic") {
      const int nx = 16, ny = 8, nz = 8, N = 4 * nx * ny * nz;
      Epetra_Map map(N, 0, comm), full(N, 0, self);
      auto make_periodic = [&](const Epetra_Map& m) {
        const int nloc = m.NumMyElements();
        std::vector<int32_t> gids(m.MyGlobalElements(), m.MyGlobalElements() + nloc);
        int64_t nnz = 0;
        REQUIRE(hymls_mi_generate_problem_periodic(1, nx, ny, nz, (double)nx * nx, 1.0, 0.0, 4 /* X */, nloc, gids.data(), &nnz, 0, 0, 0) == 0);
        std::vector<int32_t> rp(nloc + 1), ci(nnz + 1);
        std::vector<double> va(nnz + 1);
        REQUIRE(hymls_mi_generate_problem_periodic(1, nx, ny, nz, (double)nx * nx, 1.0, 0.0, 4, nloc, gids.data(), &nnz, rp.data(), ci.data(), va.data()) == 0);
        Teuchos::RCP<Epetra_CrsMatrix> K = Teuchos::rcp(new Epetra_CrsMatrix(Copy, m, 7));
        for (int i = 0; i < nloc; i++) K->InsertGlobalValues(gids[i], rp[i + 1] - rp[i], va.data() + rp[i], ci.data() + rp[i]);
        K->FillComplete();
        return K;
      };
      Teuchos::RCP<Epetra_CrsMatrix> K = make_periodic(map), K1 = make_periodic(full);
      auto prm = [&]() {
        Teuchos::RCP<Teuchos::ParameterList> p = Teuchos::rcp(new Teuchos::ParameterList());
        p->sublist("Problem").set("Equations", "Stokes-C").set("Dimension", 3).set("nx", nx).set("ny", ny).set("nz", nz).set("x-periodic", true);
        p->sublist("Preconditioner").set("Separator Length", 4).set("Number of Levels", 1).set("Partitioner", "Skew Cartesian").set("MI Transport", transport.c_str());
        return p;
      };
      HYMLS_MI::Preconditioner P_(K, prm()), P1(K1, prm());
      REQUIRE(P_.Compute() == 0 && P1.Compute() == 0);
      const int nloc = map.NumMyElements();
      Epetra_MultiVector B(map, 1), X(map, 1), B1(full, 1), X1(full, 1);
      for (int i = 0; i < nloc; i++) B[0][i] = val(map.GID(i), 5);
      for (int i = 0; i < N; i++) B1[0][i] = val(i, 5);
      REQUIRE(P_.ApplyInverse(B, X) == 0 && P1.ApplyInverse(B1, X1) == 0);
      double err = 0, nrm = 0;
      for (int i = 0; i < nloc; i++) { err = std::max(err, std::abs(X[0][i] - X1[0][map.GID(i)])); nrm = std::max(nrm, std::abs(X1[0][map.GID(i)])); }
      if (g_rank == 0) std::printf("%d ranks: x-periodic Stokes channel sharded vs one rank: max diff %.2e (max |x| %.2e)\n", P, err, nrm);
      REQUIRE(err <= 1e-10 * nrm);
    }
    comm.Barrier();
    if (g_rank == 0) std::printf("ADAPTER_MPI_OK\n");
  }
  MPI_Finalize();
  return 0;
}
