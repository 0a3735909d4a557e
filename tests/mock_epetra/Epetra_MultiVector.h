// mock: column-major multivector with a stride; Import / Export through an Epetra_Import plan
#ifndef MOCK_EPETRA_MULTIVECTOR_H
#define MOCK_EPETRA_MULTIVECTOR_H
#include <cmath>
#include <vector>
#include "Epetra_Import.h"
#include "Epetra_Map.h"
class Epetra_MultiVector {
 public:
  Epetra_MultiVector(const Epetra_BlockMap& Map, int NumVectors, bool zeroOut = true)
      : map_(Map), nvec_(NumVectors), lda_(Map.NumMyElements() + 3), data_((size_t)lda_ * NumVectors, 0.0) { (void)zeroOut; }
  virtual ~Epetra_MultiVector() {}
  int NumVectors() const { return nvec_; }
  int MyLength() const { return map_.NumMyElements(); }
  int GlobalLength() const { return map_.NumGlobalElements(); }
  int Stride() const { return lda_; }
  bool ConstantStride() const { return true; }
  const Epetra_BlockMap& Map() const { return map_; }
  int ExtractView(double** A, int* MyLDA) const { *A = const_cast<double*>(data_.data()); *MyLDA = lda_; return 0; }
  double* operator[](int i) { return data_.data() + (size_t)i * lda_; }
  const double* operator[](int i) const { return data_.data() + (size_t)i * lda_; }
  int PutScalar(double v) { for (auto& x : data_) x = v; return 0; }
  // this (on Importer.TargetMap()) <- A (on Importer.SourceMap())
  int Import(const Epetra_MultiVector& A, const Epetra_Import& Importer, Epetra_CombineMode mode) {
    return move(A, Importer.send_to(), Importer.recv_from(), mode);
  }
  // reverse communication with an import plan: this (on Importer.SourceMap()) <- A (on Importer.TargetMap())
  int Export(const Epetra_MultiVector& A, const Epetra_Import& Importer, Epetra_CombineMode mode) {
    return move(A, Importer.recv_from(), Importer.send_to(), mode);
  }
  int Dot(const Epetra_MultiVector& A, double* Result) const {
    std::vector<double> part(nvec_, 0.0);
    for (int v = 0; v < nvec_; v++) for (int i = 0; i < MyLength(); i++) part[v] += (*this)[v][i] * A[v][i];
    return map_.Comm().SumAll(part.data(), Result, nvec_);
  }
  int Norm2(double* Result) const { Dot(*this, Result); for (int v = 0; v < nvec_; v++) Result[v] = std::sqrt(Result[v]); return 0; }
 protected:
  int move(const Epetra_MultiVector& A, const std::vector<std::vector<int> >& out, const std::vector<std::vector<int> >& in, Epetra_CombineMode mode) {
    if (A.NumVectors() != nvec_ || (mode != Insert && mode != Add)) return -1;
    const int P = map_.Comm().NumProc();
    std::vector<std::vector<char> > send(P), recv;
    for (int q = 0; q < P; q++) {
      send[q].resize(out[q].size() * nvec_ * sizeof(double));
      double* s = (double*)send[q].data();
      for (int v = 0; v < nvec_; v++) for (size_t k = 0; k < out[q].size(); k++) *s++ = A[v][out[q][k]];
    }
    map_.Comm().MockAlltoallv(send, recv);
    for (int q = 0; q < P; q++) {
      if (recv[q].size() != in[q].size() * nvec_ * sizeof(double)) return -2;
      const double* r = (const double*)recv[q].data();
      for (int v = 0; v < nvec_; v++) for (size_t k = 0; k < in[q].size(); k++, r++) { if (mode == Add) (*this)[v][in[q][k]] += *r; else (*this)[v][in[q][k]] = *r; }
    }
    return 0;
  }
  Epetra_BlockMap map_;
  int nvec_, lda_;
  std::vector<double> data_;
};
#endif
