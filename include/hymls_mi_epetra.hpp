// hymls_mi_epetra.hpp -- header-only Epetra / Ifpack adapter of the MI355X HYMLS preconditioner.
//
//   class HYMLS_MI::Preconditioner : public Ifpack_Preconditioner
//
// keeps the surface of HYMLS::Preconditioner (reference src/HYMLS_Preconditioner.hpp:56-254): the Ifpack_Preconditioner
// virtuals (:93-140), the Epetra_Operator virtuals (:142-188), the BorderedOperator extension SetBorder /
// ApplyInverse(Y, T, X, S) (:228-241) and SetMatrix (:244-254), and forwards every one of them to the C ABI of
// hymls_mi.h.  It is what BaseSolver::SetPrecond wraps into a Belos::EpetraPrecOp (reference
// src/HYMLS_BaseSolver.cpp:119-139); the only line of an application that changes is the one that constructs the
// preconditioner (reference src/main.cpp:330-334).
//
// Compiled where the Trilinos headers exist.  This repository has none: tests/mock_epetra/ holds minimal stand-ins of
// the Epetra / Teuchos / Ifpack declarations used below, and tests/test_epetra_adapter.py compiles this header
// against them and drives Initialize / Compute / ApplyInverse through it.
//
// One process: the matrix rows are the GIDs 0..N-1 in order (the reference's linear map on one rank).  A distributed
// Epetra application maps onto the sharded entry points of hymls_mi.h (hymls_mi_set_comm + hymls_mi_required_rows +
// hymls_mi_set_matrix_rows); this adapter returns -99 when Comm().NumProc() > 1.
#ifndef HYMLS_MI_EPETRA_HPP
#define HYMLS_MI_EPETRA_HPP

#include <ostream>
#include <stdexcept>
#include <string>
#include <vector>

#include "Epetra_Comm.h"
#include "Epetra_CrsMatrix.h"
#include "Epetra_Map.h"
#include "Epetra_MultiVector.h"
#include "Epetra_RowMatrix.h"
#include "Epetra_SerialDenseMatrix.h"
#include "Epetra_Vector.h"
#include "Ifpack_Preconditioner.h"
#include "Teuchos_ParameterList.hpp"
#include "Teuchos_RCP.hpp"

#include "hymls_mi.h"

namespace HYMLS_MI {

class Preconditioner : public Ifpack_Preconditioner {
 public:
  // HYMLS::Preconditioner(K, params, testVector, myLevel, hid) (reference src/HYMLS_Preconditioner.hpp:80-84); the
  // last two arguments of the reference are for its own recursion and have no counterpart.  device: HIP ordinal.
  Preconditioner(Teuchos::RCP<const Epetra_RowMatrix> K, Teuchos::RCP<Teuchos::ParameterList> params,
                 Teuchos::RCP<Epetra_Vector> testVector = Teuchos::null, int device = 0)
      : params_(params), testVector_(testVector), device_(device), label_("HYMLS_MI::Preconditioner") {
    matrix_ = Teuchos::rcp_dynamic_cast<const Epetra_CrsMatrix>(K);
    if (matrix_ == Teuchos::null)   // reference src/HYMLS_Preconditioner.cpp:419-425
      throw std::runtime_error("HYMLS_MI::Preconditioner needs an Epetra_CrsMatrix");
    if (params_ != Teuchos::null) SetParameters(*params_);
  }
  virtual ~Preconditioner() { if (h_) hymls_mi_destroy(h_); }

  // ---- Ifpack_Preconditioner (reference src/HYMLS_Preconditioner.hpp:93-140)
  // the keys of the "Problem" and "Preconditioner" sublists that BasePartitioner::SetParameters and
  // Preconditioner::setParameterList read (reference src/HYMLS_BasePartitioner.cpp:31-252, Preconditioner.cpp:84-276)
  int SetParameters(Teuchos::ParameterList& List) {
    hymls_mi_default_params(&p_);
    Teuchos::ParameterList& prob = List.sublist("Problem");
    Teuchos::ParameterList& prec = List.sublist("Preconditioner");
    p_.dim = prob.get("Dimension", 3);
    p_.nx = prob.get("nx", -1); p_.ny = prob.get("ny", p_.nx); p_.nz = prob.get("nz", p_.dim > 2 ? p_.nx : 1);
    const std::string eq = prob.get("Equations", std::string("Laplace"));
    p_.equations = eq == "Laplace" ? 0 : (eq == "Stokes-C" ? 1 : -1);
    p_.dof = prob.get("Degrees of Freedom", -1);
    // "(x|y|z)" spellings win over the plain key (reference src/HYMLS_BasePartitioner.cpp:64-102); isParameter first,
    // so that reading a key never writes a default into the caller's list under a name the reference would not write
    p_.sx = prec.isParameter("Separator Length (x)") ? prec.get("Separator Length (x)", -1) : prec.get("Separator Length", 4);
    p_.sy = prec.isParameter("Separator Length (y)") ? prec.get("Separator Length (y)", -1) : -1;
    p_.sz = prec.isParameter("Separator Length (z)") ? prec.get("Separator Length (z)", -1) : -1;
    p_.cx = prec.isParameter("Coarsening Factor (x)") ? prec.get("Coarsening Factor (x)", -1)
                                                      : (prec.isParameter("Coarsening Factor") ? prec.get("Coarsening Factor", -1) : -1);
    p_.cy = prec.isParameter("Coarsening Factor (y)") ? prec.get("Coarsening Factor (y)", -1) : -1;
    p_.cz = prec.isParameter("Coarsening Factor (z)") ? prec.get("Coarsening Factor (z)", -1) : -1;
    p_.levels = prec.get("Number of Levels", 1);
    p_.partitioner = prec.get("Partitioner", std::string("Cartesian")) == "Skew Cartesian" ? 1 : 0;
    // the four spellings of "Retain Nodes" with the reference's precedence (src/HYMLS_BasePartitioner.cpp:108-137):
    // "Retain Nodes at Level k (x|y|z)" > "Retain Nodes (x|y|z)" > "Retain Nodes at Level k" > "Retain Nodes"
    static const char* const axis[3] = {" (x)", " (y)", " (z)"};
    auto opt = [&prec](const std::string& key) { return prec.isParameter(key) ? prec.get(key, -1) : -1; };
    p_.retain_nodes = opt("Retain Nodes");
    for (int d = 0; d < 3; d++) p_.retain_xyz[d] = opt(std::string("Retain Nodes") + axis[d]);
    for (int l = 0; l < 8; l++) {
      const std::string at = "Retain Nodes at Level " + std::to_string(l);
      p_.retain_at_level[l] = opt(at);
      for (int d = 0; d < 3; d++) p_.retain_at_level_xyz[l][d] = opt(at + axis[d]);
    }
    // read from the "Problem" list, as the reference does (src/HYMLS_BasePartitioner.cpp:224,244,263)
    p_.retain_pressures = prob.isParameter("Retained Pressure Nodes") ? prob.get("Retained Pressure Nodes", -1) : -1;
    // "x-periodic" .. or the bit mask "Periodicity" (src/HYMLS_BasePartitioner.cpp:49-62)
    int perio = (prob.get("x-periodic", false) ? 1 : 0) | (p_.dim > 1 && prob.get("y-periodic", false) ? 2 : 0) |
                (p_.dim > 2 && prob.get("z-periodic", false) ? 4 : 0);
    perio = prob.get("Periodicity", perio);
    for (int d = 0; d < 3; d++) p_.periodic[d] = (perio >> d) & 1;
    p_.link_velocities = prec.get("Eliminate Velocities Together", true) ? 1 : 0;
    p_.link_retained = prec.get("Eliminate Retained Nodes Together", true) ? 1 : 0;
    p_.fix_pressure_level = prec.get("Fix Pressure Level", true) ? 1 : 0;
    p_.nfix = 0;
    for (int i = 1; i <= 4; i++) {
      const std::string key = "Fix GID " + std::to_string(i);
      if (!prec.isParameter(key)) break;
      p_.fix_gid[p_.nfix++] = prec.get(key, -1);
    }
    have_params_ = true;
    if (h_) { hymls_mi_destroy(h_); h_ = 0; }   // new parameters: everything is rebuilt by the next Initialize
    return 0;
  }

  int Initialize() {
    if (!have_params_) return fail(-1, "SetParameters has not been called");
    if (matrix_->Comm().NumProc() > 1)
      return fail(-99, "distributed Epetra maps: use the sharded entry points of hymls_mi.h (INTEGRATION.md)");
    int ierr = 0;
    const bool fresh = !h_;
    if (fresh) {
      ierr = hymls_mi_create(&h_, &p_, device_);
      if (ierr) return keep_error(ierr);
    }
    ierr = PassMatrix();
    if (ierr) return keep_error(ierr);
    if (fresh && testVector_ != Teuchos::null) {
      ierr = hymls_mi_set_testvector(h_, testVector_->Values());
      if (ierr) return keep_error(ierr);
    }
    // SetMatrix with an unchanged pattern: the library kept its ordering (reference src/HYMLS_Preconditioner.hpp:244-254)
    if (hymls_mi_is_initialized(h_)) return 0;   // (PassMatrix cleared matrix_dirty_)
    return keep_error(hymls_mi_initialize(h_));
  }
  // false after SetMatrix until the new matrix has been handed over: the reference's SetMatrix sets initialized_ = false
  // (src/HYMLS_Preconditioner.hpp:244-254), so that Compute() initialises by itself (Preconditioner.cpp:403-409)
  bool IsInitialized() const { return h_ && !matrix_dirty_ && hymls_mi_is_initialized(h_); }

  int Compute() {
    if (!IsInitialized()) {   // reference src/HYMLS_Preconditioner.cpp:403-409: "I'll do it for you"
      const int ierr = Initialize();
      if (ierr) return ierr;
    }
    return keep_error(hymls_mi_compute(h_));
  }
  bool IsComputed() const { return h_ && hymls_mi_is_computed(h_); }

  double Condest(const Ifpack_CondestType = Ifpack_Cheap, const int = 1550, const double = 1e-9, Epetra_RowMatrix* = 0) {
    return -1.0;   // reference src/HYMLS_Preconditioner.cpp:607-610
  }
  double Condest() const { return -1.0; }

  // not implemented in the reference either (src/HYMLS_Preconditioner.hpp:120-121)
  int Apply(const Epetra_MultiVector&, Epetra_MultiVector&) const { return -1; }

  // Y = P^{-1} X, any number of vectors, host memory (reference src/HYMLS_Preconditioner.cpp:594-605,930-1070)
  int ApplyInverse(const Epetra_MultiVector& X, Epetra_MultiVector& Y) const {
    if (!IsComputed()) return fail(-1, "The preconditioner has not yet been computed.");
    if (X.NumVectors() != Y.NumVectors() || X.MyLength() != Y.MyLength()) return fail(-2, "ApplyInverse: X and Y differ in shape");
    double *x = 0, *y = 0;
    int ldx = 0, ldy = 0;
    X.ExtractView(&x, &ldx);
    Y.ExtractView(&y, &ldy);
    if (x == y) return fail(-2, "ApplyInverse: X and Y must not alias");
    return keep_error(hymls_mi_apply_inverse(h_, x, ldx, y, ldy, X.NumVectors(), /*on_device=*/0));
  }

  const Epetra_RowMatrix& Matrix() const { return *matrix_; }
  int NumInitialize() const { return h_ ? hymls_mi_num_initialize(h_) : 0; }
  int NumCompute() const { return h_ ? hymls_mi_num_compute(h_) : 0; }
  int NumApplyInverse() const { return h_ ? hymls_mi_num_apply_inverse(h_) : 0; }
  double InitializeTime() const { return h_ ? hymls_mi_initialize_time(h_) : 0.0; }
  double ComputeTime() const { return h_ ? hymls_mi_compute_time(h_) : 0.0; }
  double ApplyInverseTime() const { return h_ ? hymls_mi_apply_inverse_time(h_) : 0.0; }
  double InitializeFlops() const { return 0.0; }
  double ComputeFlops() const { return 0.0; }
  double ApplyInverseFlops() const { return 0.0; }
  std::ostream& Print(std::ostream& os) const {
    os << label_ << ":";
    const int nl = h_ ? hymls_mi_num_levels(h_) : 0;
    for (int l = 0; l < nl; l++)   // the "SIZE OF A / SIZE OF S" banner, reference src/HYMLS_Preconditioner.cpp:362-371
      os << " level " << l << " SIZE OF A " << hymls_mi_level_size(h_, l) << " SIZE OF S " << hymls_mi_level_schur_size(h_, l);
    return os << std::endl;
  }

  // ---- Epetra_Operator (reference src/HYMLS_Preconditioner.hpp:142-188)
  int SetUseTranspose(bool) { return -1; }
  bool HasNormInf() const { return false; }
  double NormInf() const { return -1.0; }
  const char* Label() const { return label_.c_str(); }
  bool UseTranspose() const { return false; }
  const Epetra_Comm& Comm() const { return matrix_->Comm(); }
  const Epetra_Map& OperatorDomainMap() const { return matrix_->RowMatrixRowMap(); }
  const Epetra_Map& OperatorRangeMap() const { return matrix_->RowMatrixRowMap(); }

  // ---- HYMLS::BorderedOperator (reference src/HYMLS_Preconditioner.hpp:228-241, Preconditioner.cpp:844-918)
  int SetBorder(Teuchos::RCP<const Epetra_MultiVector> V, Teuchos::RCP<const Epetra_MultiVector> W = Teuchos::null,
                Teuchos::RCP<const Epetra_SerialDenseMatrix> C = Teuchos::null) {
    if (!IsInitialized()) {
      const int ierr = Initialize();
      if (ierr) return ierr;
    }
    if (V == Teuchos::null) { have_border_ = false; return keep_error(hymls_mi_set_border(h_, 0, 0, 0, 0, 0, 0)); }
    double *v = 0, *w = 0;
    int ldv = 0, ldw = 0;
    V->ExtractView(&v, &ldv);
    if (W != Teuchos::null) W->ExtractView(&w, &ldw);
    const int m = V->NumVectors();
    std::vector<double> c;
    if (C != Teuchos::null) {
      c.resize((size_t)m * m);
      for (int j = 0; j < m; j++) for (int i = 0; i < m; i++) c[i + (size_t)m * j] = (*C)(i, j);
    }
    have_border_ = true;
    return keep_error(hymls_mi_set_border(h_, m, v, ldv, w, ldw, c.empty() ? 0 : c.data()));
  }
  bool HaveBorder() const { return have_border_; }
  // [X S]' = [K V; W' C] \ [Y T]'  (one right-hand side per call of the C ABI)
  int ApplyInverse(const Epetra_MultiVector& Y, const Epetra_SerialDenseMatrix& T, Epetra_MultiVector& X,
                   Epetra_SerialDenseMatrix& S) const {
    if (!IsComputed()) return fail(-1, "The preconditioner has not yet been computed.");
    double *y = 0, *x = 0;
    int ldy = 0, ldx = 0;
    Y.ExtractView(&y, &ldy);
    X.ExtractView(&x, &ldx);
    for (int k = 0; k < Y.NumVectors(); k++) {
      const int ierr = hymls_mi_apply_inverse_bordered(h_, y + (size_t)k * ldy, T.A() + (size_t)k * T.LDA(), x + (size_t)k * ldx,
                                                       S.A() + (size_t)k * S.LDA(), 0);
      if (ierr) return keep_error(ierr);
    }
    return 0;
  }

  // SetMatrix (reference src/HYMLS_Preconditioner.hpp:244-254): same pattern, new values; Initialize / Compute again
  // (IsInitialized() is false from here until the next Initialize; the old factors stay usable until then, as in the
  // reference, which leaves computed_ alone)
  void SetMatrix(Teuchos::RCP<const Epetra_CrsMatrix> matrix) { matrix_ = matrix; matrix_dirty_ = true; }

  // the message behind the last non-zero return code (HYMLS::Exception::what() in the reference)
  const std::string& LastError() const { return error_; }
  hymls_mi_t* Handle() const { return h_; }

 private:
  int PassMatrix() {
    const Epetra_CrsMatrix& K = *matrix_;
    const int n = K.NumMyRows();
    std::vector<int32_t> rp(n + 1, 0), ci;
    std::vector<double> va;
    ci.reserve((size_t)K.NumMyNonzeros());
    va.reserve((size_t)K.NumMyNonzeros());
    for (int i = 0; i < n; i++) {
      if (K.RowMap().GID(i) != i) return fail(-2, "matrix rows have to be the GIDs 0..N-1 in order on one process");
      int len = 0; double* v = 0; int* c = 0;
      if (K.ExtractMyRowView(i, len, v, c)) return fail(-2, "ExtractMyRowView failed (matrix not FillComplete?)");
      for (int k = 0; k < len; k++) { ci.push_back(K.GCID(c[k])); va.push_back(v[k]); }
      rp[i + 1] = (int32_t)ci.size();
    }
    matrix_dirty_ = false;
    return hymls_mi_set_matrix_csr(h_, n, rp.data(), ci.data(), va.data());
  }
  int keep_error(int ierr) const { if (ierr && h_) error_ = hymls_mi_last_error(h_); return ierr; }
  int fail(int code, const char* msg) const { error_ = msg; return code; }

  Teuchos::RCP<const Epetra_CrsMatrix> matrix_;
  Teuchos::RCP<Teuchos::ParameterList> params_;
  Teuchos::RCP<Epetra_Vector> testVector_;
  hymls_mi_params p_;
  hymls_mi_t* h_ = 0;
  int device_;
  bool have_params_ = false, have_border_ = false, matrix_dirty_ = true;
  std::string label_;
  mutable std::string error_;
};

}  // namespace HYMLS_MI
#endif
