#ifndef MOCK_EPETRA_VECTOR_H
#define MOCK_EPETRA_VECTOR_H
#include "Epetra_MultiVector.h"
class Epetra_Vector : public Epetra_MultiVector {
 public:
  explicit Epetra_Vector(const Epetra_BlockMap& Map, bool zeroOut = true) : Epetra_MultiVector(Map, 1, zeroOut) {}
  double* Values() const { return const_cast<double*>(data_.data()); }
  double& operator[](int i) { return data_[i]; }
  const double& operator[](int i) const { return data_[i]; }
};
#endif
