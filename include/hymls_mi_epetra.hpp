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
// One process: the matrix rows are the GIDs 0..N-1 in order (the reference's linear map on one rank).
//
// Distributed (Comm().NumProc() > 1, compiled with -DHYMLS_MI_HAVE_MPI, one rank per GPU; the reference's deployment,
// src/main.cpp:48-67): the matrix may live on ANY one-to-one row map.  The adapter does what the reference's Initialize /
// Compute do with their importer (src/HYMLS_Preconditioner.cpp:326-336,428-432):
//   * rank r of the communicator takes box r of hymls_mi_rank_grid (the reference's CreatePIDMap layout);
//   * hymls_mi_required_rows gives the overlapping row map; an Epetra_Import from the matrix' row map brings those rows
//     (and the test vector) to the rank, hymls_mi_set_matrix_rows hands them to the library;
//   * ApplyInverse imports B from the operator's map to the library's ownership (hymls_mi_owned_rows: interiors of the
//     rank's subdomains + the separators it owns) and exports X back, so OperatorDomainMap() == OperatorRangeMap() ==
//     K.RowMatrixRowMap() as in the reference (src/HYMLS_Preconditioner.hpp:182-186) and the Krylov loop is unchanged;
//   * the exchanges inside Compute / ApplyInverse run over the transport named by the "Preconditioner" key
//     "MI Transport": "RCCL" (default: the library's ncclSend/ncclRecv groups on its stream, bootstrapped over MPI) or
//     "MPI" (MPI_Alltoallv with host staging: any MPI library, several ranks per GPU, the test-only host simulator);
//     include/hymls_mi_mpi.h holds both.
#ifndef HYMLS_MI_EPETRA_HPP
#define HYMLS_MI_EPETRA_HPP

#include <cstdlib>
#include <ostream>
#include <stdexcept>
#include <string>
#include <vector>

#include "Epetra_Comm.h"
#include "Epetra_CrsMatrix.h"
#include "Epetra_Import.h"
#include "Epetra_Map.h"
#include "Epetra_MultiVector.h"
#include "Epetra_RowMatrix.h"
#include "Epetra_SerialDenseMatrix.h"
#include "Epetra_Vector.h"
#include "Ifpack_Preconditioner.h"
#include "Teuchos_ParameterList.hpp"
#include "Teuchos_RCP.hpp"

#include "hymls_mi.h"
#ifdef HYMLS_MI_HAVE_MPI   // (the counterpart of Epetra's own EPETRA_MPI switch)
#include "Epetra_MpiComm.h"
#include "hymls_mi_mpi.h"
#endif

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
  virtual ~Preconditioner() { Release(); }

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
    // (GaleriExt::PERIO_Flag: X_PERIO 4, Y_PERIO 2, Z_PERIO 1)
    int perio = (prob.get("x-periodic", false) ? 4 : 0) | (p_.dim > 1 && prob.get("y-periodic", false) ? 2 : 0) |
                (p_.dim > 2 && prob.get("z-periodic", false) ? 1 : 0);
    perio = prob.get("Periodicity", perio);
    for (int d = 0; d < 3; d++) p_.periodic[d] = (perio >> (2 - d)) & 1;
    p_.link_velocities = prec.get("Eliminate Velocities Together", true) ? 1 : 0;
    p_.link_retained = prec.get("Eliminate Retained Nodes Together", true) ? 1 : 0;
    p_.fix_pressure_level = prec.get("Fix Pressure Level", true) ? 1 : 0;
    p_.nfix = 0;
    for (int i = 1; i <= 4; i++) {
      const std::string key = "Fix GID " + std::to_string(i);
      if (!prec.isParameter(key)) break;
      p_.fix_gid[p_.nfix++] = prec.get(key, -1);
    }
    transport_ = prec.get("MI Transport", std::string("RCCL"));
    have_params_ = true;
    Release();   // new parameters: everything is rebuilt by the next Initialize
    return 0;
  }

  int Initialize() {
    if (!have_params_) return fail(-1, "SetParameters has not been called");
    distributed_ = matrix_->Comm().NumProc() > 1;
#ifdef HYMLS_MI_HAVE_MPI
    // (tests: one rank that takes the sharded path and exchanges with itself through the transport)
    if (std::getenv("HYMLS_MI_FORCE_SHARDED") && dynamic_cast<const Epetra_MpiComm*>(&matrix_->Comm())) distributed_ = true;
#endif
    int ierr = 0;
    const bool fresh = !h_;
    if (fresh) {
      ierr = hymls_mi_create(&h_, &p_, device_);
      if (ierr) return keep_error(ierr);
      if (distributed_) {
        ierr = AttachComm();
        if (ierr) return ierr;
      }
    }
    ierr = distributed_ ? PassMatrixRows() : PassMatrix();
    if (ierr) return keep_error(ierr);
    if (fresh && testVector_ != Teuchos::null) {
      if (distributed_) {   // one value per row given: the test vector on the overlapping map
        Epetra_Vector tv(*overlapMap_);
        if (tv.Import(*testVector_, *rowImporter_, Insert)) return fail(-3, "import of the test vector failed");
        ierr = hymls_mi_set_testvector(h_, tv.Values());
      } else {
        ierr = hymls_mi_set_testvector(h_, testVector_->Values());
      }
      if (ierr) return keep_error(ierr);
    }
    // SetMatrix with an unchanged pattern: the library kept its ordering (reference src/HYMLS_Preconditioner.hpp:244-254)
    if (hymls_mi_is_initialized(h_)) return 0;   // (PassMatrix cleared matrix_dirty_)
    ierr = hymls_mi_initialize(h_);
    if (ierr) return keep_error(ierr);
    if (distributed_) {
      // the library's ownership (interiors of this rank's subdomains + the separators it owns) as an Epetra map, and
      // the plan between it and the operator's map
      int64_t n = 0;
      ierr = hymls_mi_owned_rows(h_, &n, 0);
      if (ierr) return keep_error(ierr);
      std::vector<int32_t> own((size_t)n + 1);
      ierr = hymls_mi_owned_rows(h_, &n, own.data());
      if (ierr) return keep_error(ierr);
      std::vector<int> g(own.begin(), own.begin() + n);
      ownedMap_ = Teuchos::rcp(new Epetra_Map(-1, (int)n, g.data(), 0, matrix_->Comm()));
      vecImporter_ = Teuchos::rcp(new Epetra_Import(*ownedMap_, matrix_->RowMap()));
    }
    return 0;
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
    if (!distributed_) return keep_error(hymls_mi_apply_inverse(h_, x, ldx, y, ldy, X.NumVectors(), /*on_device=*/0));
    // operator's map -> the library's ownership and back (the reference imports into its overlapping map and exports
    // the solution the same way, src/HYMLS_Preconditioner.cpp:978-979,1050-1052)
    Epetra_MultiVector Xo(*ownedMap_, X.NumVectors()), Yo(*ownedMap_, X.NumVectors());
    if (Xo.Import(X, *vecImporter_, Insert)) return fail(-3, "ApplyInverse: import of the right-hand side failed");
    Xo.ExtractView(&x, &ldx);
    Yo.ExtractView(&y, &ldy);
    const int ierr = hymls_mi_apply_inverse(h_, x, ldx, y, ldy, X.NumVectors(), /*on_device=*/0);
    if (ierr) return keep_error(ierr);
    if (Y.Export(Yo, *vecImporter_, Insert)) return fail(-3, "ApplyInverse: export of the solution failed");
    return 0;
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
    const int m = V->NumVectors();
    // distributed: every rank passes the rows it owns (hymls_mi_owned_rows)
    Teuchos::RCP<Epetra_MultiVector> Vo, Wo;
    if (distributed_) {
      Vo = Teuchos::rcp(new Epetra_MultiVector(*ownedMap_, m));
      if (Vo->Import(*V, *vecImporter_, Insert)) return fail(-3, "SetBorder: import of V failed");
      Vo->ExtractView(&v, &ldv);
      if (W != Teuchos::null) {
        Wo = Teuchos::rcp(new Epetra_MultiVector(*ownedMap_, m));
        if (Wo->Import(*W, *vecImporter_, Insert)) return fail(-3, "SetBorder: import of W failed");
        Wo->ExtractView(&w, &ldw);
      }
    } else {
      V->ExtractView(&v, &ldv);
      if (W != Teuchos::null) W->ExtractView(&w, &ldw);
    }
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
    Teuchos::RCP<Epetra_MultiVector> Yo, Xo;
    if (distributed_) {
      Yo = Teuchos::rcp(new Epetra_MultiVector(*ownedMap_, Y.NumVectors()));
      Xo = Teuchos::rcp(new Epetra_MultiVector(*ownedMap_, Y.NumVectors()));
      if (Yo->Import(Y, *vecImporter_, Insert)) return fail(-3, "ApplyInverse: import of the right-hand side failed");
      Yo->ExtractView(&y, &ldy);
      Xo->ExtractView(&x, &ldx);
    } else {
      Y.ExtractView(&y, &ldy);
      X.ExtractView(&x, &ldx);
    }
    for (int k = 0; k < Y.NumVectors(); k++) {
      const int ierr = hymls_mi_apply_inverse_bordered(h_, y + (size_t)k * ldy, T.A() + (size_t)k * T.LDA(), x + (size_t)k * ldx,
                                                       S.A() + (size_t)k * S.LDA(), 0);
      if (ierr) return keep_error(ierr);
    }
    if (distributed_ && X.Export(*Xo, *vecImporter_, Insert)) return fail(-3, "ApplyInverse: export of the solution failed");
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
  // the rows this rank has to hold (hymls_mi_required_rows = the reference's overlapping map) brought here by an
  // Epetra_Import from the matrix' row map, as reference src/HYMLS_Preconditioner.cpp:326-336,428-432
  int PassMatrixRows() {
    if (overlapMap_ == Teuchos::null) {
      int64_t n = 0;
      int ierr = hymls_mi_required_rows(h_, &n, 0);
      if (ierr) return ierr;
      std::vector<int32_t> req((size_t)n + 1);
      ierr = hymls_mi_required_rows(h_, &n, req.data());
      if (ierr) return ierr;
      std::vector<int> g(req.begin(), req.begin() + n);
      overlapMap_ = Teuchos::rcp(new Epetra_Map(-1, (int)n, g.data(), 0, matrix_->Comm()));
      rowImporter_ = Teuchos::rcp(new Epetra_Import(*overlapMap_, matrix_->RowMap()));
    }
    Epetra_CrsMatrix Kov(Copy, *overlapMap_, 0);
    if (Kov.Import(*matrix_, *rowImporter_, Insert)) return fail(-3, "import of the overlapping matrix rows failed");
    if (Kov.FillComplete(matrix_->DomainMap(), matrix_->RangeMap())) return fail(-3, "FillComplete of the overlapping matrix failed");
    const int n = Kov.NumMyRows();
    std::vector<int32_t> gid(n), rp(n + 1, 0), ci;
    std::vector<double> va;
    ci.reserve((size_t)Kov.NumMyNonzeros());
    va.reserve((size_t)Kov.NumMyNonzeros());
    for (int i = 0; i < n; i++) {
      gid[i] = overlapMap_->GID(i);
      int len = 0; double* v = 0; int* c = 0;
      if (Kov.ExtractMyRowView(i, len, v, c)) return fail(-2, "ExtractMyRowView failed");
      for (int k = 0; k < len; k++) { ci.push_back(Kov.GCID(c[k])); va.push_back(v[k]); }
      rp[i + 1] = (int32_t)ci.size();
    }
    matrix_dirty_ = false;
    return hymls_mi_set_matrix_rows(h_, n, gid.data(), rp.data(), ci.data(), va.data());
  }
  // one rank = one GPU = one box of the grid; the transport the parameter list names
  int AttachComm() {
#ifdef HYMLS_MI_HAVE_MPI
    const Epetra_MpiComm* mc = dynamic_cast<const Epetra_MpiComm*>(&matrix_->Comm());
    if (!mc) return fail(-2, "distributed matrix whose communicator is not an Epetra_MpiComm");
    int px = 1, py = 1, pz = 1;
    hymls_mi_rank_grid(mc->NumProc(), &px, &py, &pz);
    int ierr;
    if (transport_ == "MPI") ierr = hymls_mi_set_comm_mpi(h_, mc->Comm(), px, py, pz, &mpi_transport_);
    else if (transport_ == "RCCL") ierr = hymls_mi_set_comm_rccl_mpi(h_, mc->Comm(), device_, px, py, pz, &nccl_comm_);
    else return fail(-2, "\"MI Transport\" has to be \"RCCL\" or \"MPI\"");
    return keep_error(ierr);
#else
    return fail(-99, "distributed Epetra maps need the adapter compiled with -DHYMLS_MI_HAVE_MPI");
#endif
  }
  void Release() {
    if (!h_) return;
#ifdef HYMLS_MI_HAVE_MPI
    if (mpi_transport_) { hymls_mi_mpi_transport_free(mpi_transport_); mpi_transport_ = 0; }   // (frees device arenas through h_)
#endif
    hymls_mi_destroy(h_);
    h_ = 0;
#ifdef HYMLS_MI_HAVE_MPI
    if (nccl_comm_) { hymls_mi_rccl_comm_destroy(nccl_comm_); nccl_comm_ = 0; }
#endif
    overlapMap_ = Teuchos::null; ownedMap_ = Teuchos::null; rowImporter_ = Teuchos::null; vecImporter_ = Teuchos::null;
  }
  int keep_error(int ierr) const { if (ierr && h_) error_ = hymls_mi_last_error(h_); return ierr; }
  int fail(int code, const char* msg) const { error_ = msg; return code; }

  Teuchos::RCP<const Epetra_CrsMatrix> matrix_;
  Teuchos::RCP<Teuchos::ParameterList> params_;
  Teuchos::RCP<Epetra_Vector> testVector_;
  hymls_mi_params p_;
  hymls_mi_t* h_ = 0;
  int device_;
  bool have_params_ = false, have_border_ = false, matrix_dirty_ = true, distributed_ = false;
  std::string transport_ = "RCCL";
  Teuchos::RCP<Epetra_Map> overlapMap_, ownedMap_;            // required rows / owned rows of this rank
  Teuchos::RCP<Epetra_Import> rowImporter_, vecImporter_;     // matrix row map -> overlapMap_ / ownedMap_
#ifdef HYMLS_MI_HAVE_MPI
  hymls_mi_mpi_transport* mpi_transport_ = 0;
  void* nccl_comm_ = 0;
#endif
  std::string label_;
  mutable std::string error_;
};

}  // namespace HYMLS_MI
#endif
