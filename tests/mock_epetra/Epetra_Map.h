// mock: a linear or arbitrary one-process map
#ifndef MOCK_EPETRA_MAP_H
#define MOCK_EPETRA_MAP_H
#include <vector>
#include "Epetra_Comm.h"
class Epetra_BlockMap {
 public:
  Epetra_BlockMap(int NumGlobalElements, int IndexBase, const Epetra_Comm& Comm) : comm_(&Comm), base_(IndexBase), linear_(true) {
    gids_.resize(NumGlobalElements);
    for (int i = 0; i < NumGlobalElements; i++) gids_[i] = IndexBase + i;
  }
  Epetra_BlockMap(int NumGlobalElements, int NumMyElements, const int* MyGlobalElements, int IndexBase, const Epetra_Comm& Comm)
      : comm_(&Comm), base_(IndexBase), linear_(false), gids_(MyGlobalElements, MyGlobalElements + NumMyElements) { (void)NumGlobalElements; }
  virtual ~Epetra_BlockMap() {}
  int NumMyElements() const { return (int)gids_.size(); }
  int NumGlobalElements() const { return (int)gids_.size(); }
  int GID(int lid) const { return lid >= 0 && lid < (int)gids_.size() ? gids_[lid] : base_ - 1; }
  int LID(int gid) const { if (linear_) return gid >= base_ && gid < base_ + (int)gids_.size() ? gid - base_ : -1; for (int i = 0; i < (int)gids_.size(); i++) if (gids_[i] == gid) return i; return -1; }
  const Epetra_Comm& Comm() const { return *comm_; }
  bool SameAs(const Epetra_BlockMap& o) const { return gids_ == o.gids_; }
 private:
  const Epetra_Comm* comm_;
  int base_;
  bool linear_;
  std::vector<int> gids_;
};
class Epetra_Map : public Epetra_BlockMap {
 public:
  Epetra_Map(int NumGlobalElements, int IndexBase, const Epetra_Comm& Comm) : Epetra_BlockMap(NumGlobalElements, IndexBase, Comm) {}
  Epetra_Map(int NumGlobalElements, int NumMyElements, const int* MyGlobalElements, int IndexBase, const Epetra_Comm& Comm)
      : Epetra_BlockMap(NumGlobalElements, NumMyElements, MyGlobalElements, IndexBase, Comm) {}
};
#endif
