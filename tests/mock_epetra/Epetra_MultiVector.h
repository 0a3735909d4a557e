// mock: column-major multivector with a stride
#ifndef MOCK_EPETRA_MULTIVECTOR_H
#define MOCK_EPETRA_MULTIVECTOR_H
#include <vector>
#include "Epetra_Map.h"
class Epetra_MultiVector {
 public:
  Epetra_MultiVector(const Epetra_BlockMap& Map, int NumVectors, bool zeroOut = true)
      : map_(&Map), nvec_(NumVectors), lda_(Map.NumMyElements() + 3), data_((size_t)lda_ * NumVectors, 0.0) { (void)zeroOut; }
  virtual ~Epetra_MultiVector() {}
  int NumVectors() const { return nvec_; }
  int MyLength() const { return map_->NumMyElements(); }
  int GlobalLength() const { return map_->NumGlobalElements(); }
  int Stride() const { return lda_; }
  bool ConstantStride() const { return true; }
  const Epetra_BlockMap& Map() const { return *map_; }
  int ExtractView(double** A, int* MyLDA) const { *A = const_cast<double*>(data_.data()); *MyLDA = lda_; return 0; }
  double* operator[](int i) { return data_.data() + (size_t)i * lda_; }
  const double* operator[](int i) const { return data_.data() + (size_t)i * lda_; }
  int PutScalar(double v) { for (auto& x : data_) x = v; return 0; }
 protected:
  const Epetra_BlockMap* map_;
  int nvec_, lda_;
  std::vector<double> data_;
};
#endif
