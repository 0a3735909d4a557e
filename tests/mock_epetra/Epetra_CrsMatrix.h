// mock: one-process CSR matrix with a column map (local column ids), the view accessors the adapter uses
#ifndef MOCK_EPETRA_CRSMATRIX_H
#define MOCK_EPETRA_CRSMATRIX_H
#include <vector>
#include "Epetra_RowMatrix.h"
enum Epetra_DataAccess { Copy, View };
class Epetra_CrsMatrix : public Epetra_RowMatrix {
 public:
  Epetra_CrsMatrix(Epetra_DataAccess, const Epetra_Map& RowMap, int NumEntriesPerRow)
      : rowmap_(RowMap), rows_(RowMap.NumMyElements()), filled_(false) { (void)NumEntriesPerRow; }
  int InsertGlobalValues(int GlobalRow, int NumEntries, const double* Values, const int* Indices) {
    const int l = rowmap_.LID(GlobalRow);
    if (l < 0 || filled_) return -1;
    for (int k = 0; k < NumEntries; k++) rows_[l].push_back(std::make_pair(Indices[k], Values[k]));
    return 0;
  }
  int FillComplete() {   // column map = row map here (square, one process): local column id = LID of the GID
    rp_.assign(1, 0);
    for (auto& r : rows_) { for (auto& e : r) { ci_.push_back(rowmap_.LID(e.first)); va_.push_back(e.second); } rp_.push_back((int)ci_.size()); }
    rows_.clear(); filled_ = true;
    return 0;
  }
  bool Filled() const { return filled_; }
  int ExtractMyRowView(int MyRow, int& NumEntries, double*& Values, int*& Indices) const {
    if (!filled_ || MyRow < 0 || MyRow + 1 >= (int)rp_.size()) return -1;
    NumEntries = rp_[MyRow + 1] - rp_[MyRow];
    Values = const_cast<double*>(va_.data()) + rp_[MyRow];
    Indices = const_cast<int*>(ci_.data()) + rp_[MyRow];
    return 0;
  }
  int ReplaceMyValue(int MyRow, int k, double v) { va_[rp_[MyRow] + k] = v; return 0; }   // (mock helper for the SetMatrix test)
  int GCID(int LCID_in) const { return rowmap_.GID(LCID_in); }
  int NumMyRows() const { return rowmap_.NumMyElements(); }
  int NumMyNonzeros() const { return (int)ci_.size(); }
  const Epetra_Map& RowMap() const { return rowmap_; }
  const Epetra_Map& ColMap() const { return rowmap_; }
  const Epetra_Map& RowMatrixRowMap() const { return rowmap_; }
  const Epetra_Map& RowMatrixColMap() const { return rowmap_; }
  // Epetra_Operator
  int SetUseTranspose(bool) { return -1; }
  int Apply(const Epetra_MultiVector& X, Epetra_MultiVector& Y) const {
    for (int v = 0; v < X.NumVectors(); v++)
      for (int i = 0; i < NumMyRows(); i++) { double s = 0; for (int e = rp_[i]; e < rp_[i + 1]; e++) s += va_[e] * X[v][ci_[e]]; Y[v][i] = s; }
    return 0;
  }
  int ApplyInverse(const Epetra_MultiVector&, Epetra_MultiVector&) const { return -1; }
  double NormInf() const { return -1.0; }
  const char* Label() const { return "Epetra::CrsMatrix (mock)"; }
  bool UseTranspose() const { return false; }
  bool HasNormInf() const { return false; }
  const Epetra_Comm& Comm() const { return rowmap_.Comm(); }
  const Epetra_Map& OperatorDomainMap() const { return rowmap_; }
  const Epetra_Map& OperatorRangeMap() const { return rowmap_; }
 private:
  Epetra_Map rowmap_;
  std::vector<std::vector<std::pair<int, double> > > rows_;
  std::vector<int> rp_, ci_;
  std::vector<double> va_;
  bool filled_;
};
#endif
