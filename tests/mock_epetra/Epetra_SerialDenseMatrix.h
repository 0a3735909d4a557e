#ifndef MOCK_EPETRA_SERIALDENSEMATRIX_H
#define MOCK_EPETRA_SERIALDENSEMATRIX_H
#include <vector>
class Epetra_SerialDenseMatrix {
 public:
  Epetra_SerialDenseMatrix(int NumRows, int NumCols) : m_(NumRows), n_(NumCols), a_((size_t)NumRows * NumCols, 0.0) {}
  int M() const { return m_; }
  int N() const { return n_; }
  int LDA() const { return m_; }
  double* A() const { return const_cast<double*>(a_.data()); }
  double& operator()(int i, int j) { return a_[i + (size_t)m_ * j]; }
  const double& operator()(int i, int j) const { return a_[i + (size_t)m_ * j]; }
 private:
  int m_, n_;
  std::vector<double> a_;
};
#endif
