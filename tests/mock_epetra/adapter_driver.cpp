// Drives include/hymls_mi_epetra.hpp the way a Trilinos application would: builds an Epetra_CrsMatrix, constructs the
// preconditioner from a Teuchos::ParameterList, and uses it only through Ifpack_Preconditioner / Epetra_Operator
// pointers (what Belos::EpetraPrecOp holds, reference src/HYMLS_BaseSolver.cpp:119-139).  Mirrors the reference's
// unit tests of the same interface (testSuite/unit_tests/HYMLS_Preconditioner.cpp:106-136,247-378).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <sstream>
#include "hymls_mi_epetra.hpp"

#define REQUIRE(c) do { if (!(c)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)

static Teuchos::RCP<Epetra_CrsMatrix> make_matrix(int equations, int n, const Epetra_Map& map, double scale) {
  int64_t nrows = 0, nnz = 0;
  hymls_mi_generate_matrix(equations, n, n, n, (double)n * n, 1.0, &nrows, &nnz, 0, 0, 0);
  std::vector<int32_t> rp(nrows + 1), ci(nnz);
  std::vector<double> va(nnz);
  hymls_mi_generate_matrix(equations, n, n, n, (double)n * n, 1.0, &nrows, &nnz, rp.data(), ci.data(), va.data());
  Teuchos::RCP<Epetra_CrsMatrix> K = Teuchos::rcp(new Epetra_CrsMatrix(Copy, map, 7));
  for (int i = 0; i < nrows; i++) {
    std::vector<double> v(va.begin() + rp[i], va.begin() + rp[i + 1]);
    for (double& x : v) x *= scale;
    K->InsertGlobalValues(i, rp[i + 1] - rp[i], v.data(), ci.data() + rp[i]);
  }
  K->FillComplete();
  return K;
}

static double rnd() { static unsigned s = 12345u; s = s * 1664525u + 1013904223u; return (double)(s >> 8) / (1u << 24) * 2.0 - 1.0; }

int main() {
  Epetra_SerialComm comm;
  // ---- 1. Laplace 8^3, Number of Levels = 0: the preconditioner is an exact inverse (reference :247-276), 3 vectors
  {
    const int n = 8, N = n * n * n;
    Epetra_Map map(N, 0, comm);
    Teuchos::RCP<Epetra_CrsMatrix> K = make_matrix(0, n, map, 1.0);
    Teuchos::RCP<Teuchos::ParameterList> params = Teuchos::rcp(new Teuchos::ParameterList());
    params->sublist("Problem").set("Equations", "Laplace").set("Dimension", 3).set("nx", n).set("ny", n).set("nz", n);
    params->sublist("Preconditioner").set("Separator Length", 4).set("Number of Levels", 0);
    Teuchos::RCP<HYMLS_MI::Preconditioner> P = Teuchos::rcp(new HYMLS_MI::Preconditioner(K, params));
    Ifpack_Preconditioner* prec = P.get();           // everything below goes through the Trilinos interfaces
    const Epetra_Operator* op = prec;
    REQUIRE(!prec->IsInitialized() && !prec->IsComputed());
    Epetra_MultiVector X(map, 3), B(map, 3), Y(map, 3);
    for (int v = 0; v < 3; v++) for (int i = 0; i < N; i++) X[v][i] = rnd();
    K->Apply(X, B);
    REQUIRE(op->ApplyInverse(B, Y) == -1);             // before Compute: error (reference Preconditioner.cpp:936-939)
    REQUIRE(prec->Compute() == 0);                     // auto-initialises (:403-409)
    REQUIRE(prec->IsInitialized() && prec->IsComputed() && prec->NumInitialize() == 1 && prec->NumCompute() == 1);
    REQUIRE(op->ApplyInverse(B, Y) == 0);
    double err = 0;
    for (int v = 0; v < 3; v++) for (int i = 0; i < N; i++) err = std::max(err, std::abs(Y[v][i] - X[v][i]));
    std::printf("levels 0: max |P^{-1} K x - x| = %.2e\n", err);
    REQUIRE(err < 1e-10);
    REQUIRE(prec->NumApplyInverse() == 1);
    REQUIRE(op->Apply(X, Y) == -1 && prec->SetUseTranspose(true) == -1 && !op->UseTranspose() && !op->HasNormInf());
    REQUIRE(prec->Condest() == -1.0 && prec->Condest(Ifpack_Cheap, 10, 1e-3, 0) == -1.0);
    REQUIRE(op->OperatorDomainMap().SameAs(K->RowMap()) && op->OperatorRangeMap().SameAs(K->RowMap()));
    REQUIRE(&prec->Matrix() == static_cast<const Epetra_RowMatrix*>(K.get()) && &op->Comm() == &K->Comm());
    REQUIRE(op->ApplyInverse(B, B) == -2);             // aliasing is refused
    std::ostringstream os; os << *prec;
    REQUIRE(os.str().find("SIZE OF A 512") != std::string::npos);
    // SetMatrix: same pattern, values doubled -> the solution halves, the ordering is kept (no new Initialize)
    Teuchos::RCP<Epetra_CrsMatrix> K2 = make_matrix(0, n, map, 2.0);
    P->SetMatrix(K2);
    REQUIRE(prec->Initialize() == 0 && prec->Compute() == 0 && prec->NumInitialize() == 1 && prec->NumCompute() == 2);
    REQUIRE(op->ApplyInverse(B, Y) == 0);
    err = 0;
    for (int v = 0; v < 3; v++) for (int i = 0; i < N; i++) err = std::max(err, std::abs(2.0 * Y[v][i] - X[v][i]));
    REQUIRE(err < 1e-10);
    // SetMatrix, then Compute() WITHOUT Initialize(): the reference's SetMatrix clears initialized_
    // (src/HYMLS_Preconditioner.hpp:244-254) and Compute() initialises by itself (Preconditioner.cpp:403-409); the new
    // values must be the ones that are factored
    Teuchos::RCP<Epetra_CrsMatrix> K4 = make_matrix(0, n, map, 4.0);
    P->SetMatrix(K4);
    REQUIRE(!prec->IsInitialized());
    REQUIRE(prec->Compute() == 0 && prec->IsInitialized() && prec->IsComputed() && prec->NumCompute() == 3);
    REQUIRE(op->ApplyInverse(B, Y) == 0);
    err = 0;
    for (int v = 0; v < 3; v++) for (int i = 0; i < N; i++) err = std::max(err, std::abs(4.0 * Y[v][i] - X[v][i]));
    std::printf("SetMatrix -> Compute without Initialize: max |4 P^{-1} K x - x| = %.2e\n", err);
    REQUIRE(err < 1e-10);
    P->SetMatrix(K2);
    REQUIRE(prec->Compute() == 0);
    // bordered: [K V; V' 0] [x; s] = [y; t] solved exactly on one level (reference :278-378)
    Teuchos::RCP<Epetra_MultiVector> V = Teuchos::rcp(new Epetra_MultiVector(map, 1));
    for (int i = 0; i < N; i++) (*V)[0][i] = rnd();
    REQUIRE(P->SetBorder(V) == 0 && P->HaveBorder() && !prec->IsComputed());
    REQUIRE(prec->Compute() == 0);
    Epetra_MultiVector y1(map, 1), x1(map, 1), r1(map, 1);
    Epetra_SerialDenseMatrix T(1, 1), S(1, 1);
    for (int i = 0; i < N; i++) y1[0][i] = rnd();
    T(0, 0) = 0.37;
    REQUIRE(P->ApplyInverse(y1, T, x1, S) == 0);
    K2->Apply(x1, r1);
    double res = 0, vx = 0;
    for (int i = 0; i < N; i++) { res = std::max(res, std::abs(r1[0][i] + (*V)[0][i] * S(0, 0) - y1[0][i])); vx += (*V)[0][i] * x1[0][i]; }
    std::printf("bordered: residual %.2e, |V'x - t| %.2e\n", res, std::abs(vx - T(0, 0)));
    REQUIRE(res < 1e-9 && std::abs(vx - T(0, 0)) < 1e-9);
    REQUIRE(P->SetBorder(Teuchos::null) == 0 && !P->HaveBorder());
  }
  // ---- 2. Laplace 16^3, two-level method, used as the preconditioner of a CG loop that only sees Epetra_Operator
  {
    const int n = 16, N = n * n * n;
    Epetra_Map map(N, 0, comm);
    Teuchos::RCP<Epetra_CrsMatrix> K = make_matrix(0, n, map, -1.0);   // positive definite
    Teuchos::RCP<Teuchos::ParameterList> params = Teuchos::rcp(new Teuchos::ParameterList());
    params->sublist("Problem").set("Equations", "Laplace").set("Dimension", 3).set("nx", n);
    params->sublist("Preconditioner").set("Separator Length (x)", 4).set("Separator Length (y)", 4).set("Separator Length (z)", 4).set("Number of Levels", 1);   // the "(x|y|z)" spellings, reference src/HYMLS_BasePartitioner.cpp:68-73
    Teuchos::RCP<Epetra_Vector> tv = Teuchos::rcp(new Epetra_Vector(map));
    tv->PutScalar(1.0);
    HYMLS_MI::Preconditioner P(K, params, tv);
    REQUIRE(P.Initialize() == 0 && P.Compute() == 0);
    const Epetra_Operator& A = *K;
    const Epetra_Operator& M = P;
    Epetra_MultiVector xex(map, 1), b(map, 1), x(map, 1), r(map, 1), z(map, 1), p(map, 1), q(map, 1);
    for (int i = 0; i < N; i++) xex[0][i] = rnd();
    A.Apply(xex, b);
    auto dot = [&](const Epetra_MultiVector& u, const Epetra_MultiVector& v) { double s = 0; for (int i = 0; i < N; i++) s += u[0][i] * v[0][i]; return s; };
    for (int i = 0; i < N; i++) r[0][i] = b[0][i];
    const double bn = std::sqrt(dot(b, b));
    double rz = 0;
    int it = 0;
    for (; it < 100; it++) {
      REQUIRE(M.ApplyInverse(r, z) == 0);
      const double rz1 = dot(r, z);
      for (int i = 0; i < N; i++) p[0][i] = it ? z[0][i] + rz1 / rz * p[0][i] : z[0][i];
      rz = rz1;
      A.Apply(p, q);
      const double al = rz / dot(p, q);
      for (int i = 0; i < N; i++) { x[0][i] += al * p[0][i]; r[0][i] -= al * q[0][i]; }
      if (std::sqrt(dot(r, r)) <= 1e-10 * bn) { it++; break; }
    }
    double err = 0;
    for (int i = 0; i < N; i++) err = std::max(err, std::abs(x[0][i] - xex[0][i]));
    std::printf("two-level preconditioned CG through Epetra_Operator: %d iterations, error %.2e\n", it, err);
    REQUIRE(it < 40 && err < 1e-7);
  }
  // ---- 3. errors keep their codes and messages: wrong grid size
  {
    Epetra_Map map(27, 0, comm);
    Teuchos::RCP<Epetra_CrsMatrix> K = make_matrix(0, 3, map, 1.0);
    Teuchos::RCP<Teuchos::ParameterList> params = Teuchos::rcp(new Teuchos::ParameterList());
    params->sublist("Problem").set("Equations", "Laplace").set("nx", 4);
    params->sublist("Preconditioner").set("Separator Length", 2);
    HYMLS_MI::Preconditioner P(K, params);
    REQUIRE(P.Initialize() == -2 && !P.LastError().empty() && !P.IsInitialized());
  }
  std::printf("ADAPTER_OK\n");
  return 0;
}
