// mock: row-distributed CSR matrix with a column map (local column ids), the view accessors the adapter uses, Import of
// rows through an Epetra_Import (what the reference does to get its overlapping matrix, src/HYMLS_Preconditioner.cpp:
// 428-432) and a distributed Apply
#ifndef MOCK_EPETRA_CRSMATRIX_H
#define MOCK_EPETRA_CRSMATRIX_H
#include <algorithm>
#include <memory>
#include <vector>
#include "Epetra_Import.h"
#include "Epetra_RowMatrix.h"
enum Epetra_DataAccess { Copy, View };
class Epetra_CrsMatrix : public Epetra_RowMatrix {
 public:
  Epetra_CrsMatrix(Epetra_DataAccess, const Epetra_Map& RowMap, int NumEntriesPerRow)
      : rowmap_(RowMap), domainmap_(RowMap), colmap_(0, 0, (const int*)0, 0, RowMap.Comm()), rows_(RowMap.NumMyElements()), filled_(false) { (void)NumEntriesPerRow; }
  int InsertGlobalValues(int GlobalRow, int NumEntries, const double* Values, const int* Indices) {
    const int l = rowmap_.LID(GlobalRow);
    if (l < 0 || filled_) return -1;
    for (int k = 0; k < NumEntries; k++) rows_[l].push_back(std::make_pair(Indices[k], Values[k]));
    return 0;
  }
  // this (rows of Importer.TargetMap()) <- rows of A (Importer.SourceMap())
  int Import(const Epetra_CrsMatrix& A, const Epetra_Import& Importer, Epetra_CombineMode mode) {
    if (filled_ || !A.filled_ || mode != Insert) return -1;
    const int P = rowmap_.Comm().NumProc();
    std::vector<std::vector<char> > send(P), recv;
    for (int q = 0; q < P; q++)
      for (int sl : Importer.send_to()[q]) {
        const int n = A.rp_[sl + 1] - A.rp_[sl];
        put(send[q], &n, sizeof n);
        for (int e = A.rp_[sl]; e < A.rp_[sl + 1]; e++) { const int g = A.GCID(A.ci_[e]); put(send[q], &g, sizeof g); put(send[q], &A.va_[e], sizeof(double)); }
      }
    rowmap_.Comm().MockAlltoallv(send, recv);
    for (int q = 0; q < P; q++) {
      const char* p = recv[q].data();
      for (int tl : Importer.recv_from()[q]) {
        int n; std::memcpy(&n, p, sizeof n); p += sizeof n;
        rows_[tl].clear();
        for (int k = 0; k < n; k++) { int g; double v; std::memcpy(&g, p, sizeof g); p += sizeof g; std::memcpy(&v, p, sizeof v); p += sizeof v; rows_[tl].push_back(std::make_pair(g, v)); }
      }
    }
    return 0;
  }
  int FillComplete() { return FillComplete(rowmap_, rowmap_); }
  int FillComplete(const Epetra_Map& DomainMap, const Epetra_Map& RangeMap) {
    (void)RangeMap;
    domainmap_ = DomainMap;
    // column map: the column GIDs of the local rows, those of the domain map first (in its order), then the others sorted
    std::vector<int> cols;
    for (auto& r : rows_) for (auto& e : r) cols.push_back(e.first);
    std::sort(cols.begin(), cols.end());
    cols.erase(std::unique(cols.begin(), cols.end()), cols.end());
    std::vector<int> mine, other;
    for (int g : cols) (domainmap_.MyGID(g) ? mine : other).push_back(g);
    std::sort(mine.begin(), mine.end(), [&](int a, int b) { return domainmap_.LID(a) < domainmap_.LID(b); });
    mine.insert(mine.end(), other.begin(), other.end());
    colmap_ = Epetra_Map(-1, (int)mine.size(), mine.data(), 0, rowmap_.Comm());
    rp_.assign(1, 0);
    for (auto& r : rows_) { for (auto& e : r) { ci_.push_back(colmap_.LID(e.first)); va_.push_back(e.second); } rp_.push_back((int)ci_.size()); }
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
  int GCID(int LCID_in) const { return colmap_.GID(LCID_in); }
  int GRID(int LRID_in) const { return rowmap_.GID(LRID_in); }
  int NumMyRows() const { return rowmap_.NumMyElements(); }
  int NumMyNonzeros() const { return (int)ci_.size(); }
  const Epetra_Map& RowMap() const { return rowmap_; }
  const Epetra_Map& ColMap() const { return colmap_; }
  const Epetra_Map& DomainMap() const { return domainmap_; }
  const Epetra_Map& RangeMap() const { return rowmap_; }
  const Epetra_Map& RowMatrixRowMap() const { return rowmap_; }
  const Epetra_Map& RowMatrixColMap() const { return colmap_; }
  // Epetra_Operator
  int SetUseTranspose(bool) { return -1; }
  int Apply(const Epetra_MultiVector& X, Epetra_MultiVector& Y) const {   // X on the domain map, Y on the row map
    if (!filled_) return -1;
    if (!colimp_) colimp_.reset(new Epetra_Import(colmap_, domainmap_));
    Epetra_MultiVector Xc(colmap_, X.NumVectors());
    if (Xc.Import(X, *colimp_, Insert)) return -1;
    for (int v = 0; v < X.NumVectors(); v++)
      for (int i = 0; i < NumMyRows(); i++) { double s = 0; for (int e = rp_[i]; e < rp_[i + 1]; e++) s += va_[e] * Xc[v][ci_[e]]; Y[v][i] = s; }
    return 0;
  }
  int ApplyInverse(const Epetra_MultiVector&, Epetra_MultiVector&) const { return -1; }
  double NormInf() const { return -1.0; }
  const char* Label() const { return "Epetra::CrsMatrix (mock)"; }
  bool UseTranspose() const { return false; }
  bool HasNormInf() const { return false; }
  const Epetra_Comm& Comm() const { return rowmap_.Comm(); }
  const Epetra_Map& OperatorDomainMap() const { return domainmap_; }
  const Epetra_Map& OperatorRangeMap() const { return rowmap_; }
 private:
  static void put(std::vector<char>& b, const void* p, size_t n) { b.insert(b.end(), (const char*)p, (const char*)p + n); }
  Epetra_Map rowmap_, domainmap_, colmap_;
  std::vector<std::vector<std::pair<int, double> > > rows_;
  std::vector<int> rp_, ci_;
  std::vector<double> va_;
  bool filled_;
  mutable std::shared_ptr<Epetra_Import> colimp_;
};
#endif
