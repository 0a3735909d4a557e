// mock: a linear or arbitrary map, distributed over the ranks of its communicator
#ifndef MOCK_EPETRA_MAP_H
#define MOCK_EPETRA_MAP_H
#include <unordered_map>
#include <vector>
#include "Epetra_Comm.h"
class Epetra_BlockMap {
 public:
  // linear map: contiguous chunks, the first NumGlobalElements % NumProc ranks hold one element more (as Epetra)
  Epetra_BlockMap(int NumGlobalElements, int IndexBase, const Epetra_Comm& Comm) : comm_(&Comm), base_(IndexBase), nglobal_(NumGlobalElements) {
    const int P = Comm.NumProc(), r = Comm.MyPID();
    const int q = NumGlobalElements / P, rem = NumGlobalElements % P;
    const int first = r * q + (r < rem ? r : rem), n = q + (r < rem ? 1 : 0);
    gids_.resize(n);
    for (int i = 0; i < n; i++) gids_[i] = IndexBase + first + i;
    index();
  }
  // arbitrary map; NumGlobalElements = -1: computed (collective)
  Epetra_BlockMap(int NumGlobalElements, int NumMyElements, const int* MyGlobalElements, int IndexBase, const Epetra_Comm& Comm)
      : comm_(&Comm), base_(IndexBase), nglobal_(NumGlobalElements), gids_(MyGlobalElements, MyGlobalElements + NumMyElements) {
    int mine = NumMyElements, all = 0;
    Comm.SumAll(&mine, &all, 1);
    if (nglobal_ < 0) nglobal_ = all;
    index();
  }
  virtual ~Epetra_BlockMap() {}
  int NumMyElements() const { return (int)gids_.size(); }
  int NumGlobalElements() const { return nglobal_; }
  int GID(int lid) const { return lid >= 0 && lid < (int)gids_.size() ? gids_[lid] : base_ - 1; }
  int LID(int gid) const { auto it = lid_.find(gid); return it == lid_.end() ? -1 : it->second; }
  bool MyGID(int gid) const { return lid_.count(gid) > 0; }
  int* MyGlobalElements() const { return const_cast<int*>(gids_.data()); }
  const Epetra_Comm& Comm() const { return *comm_; }
  bool SameAs(const Epetra_BlockMap& o) const {   // collective
    int same = gids_ == o.gids_ ? 1 : 0, all = 0;
    comm_->MinAll(&same, &all, 1);
    return all == 1;
  }
 private:
  void index() { lid_.reserve(gids_.size() * 2); for (int i = 0; i < (int)gids_.size(); i++) lid_.emplace(gids_[i], i); }
  const Epetra_Comm* comm_;
  int base_, nglobal_;
  std::vector<int> gids_;
  std::unordered_map<int, int> lid_;
};
class Epetra_Map : public Epetra_BlockMap {
 public:
  Epetra_Map(int NumGlobalElements, int IndexBase, const Epetra_Comm& Comm) : Epetra_BlockMap(NumGlobalElements, IndexBase, Comm) {}
  Epetra_Map(int NumGlobalElements, int NumMyElements, const int* MyGlobalElements, int IndexBase, const Epetra_Comm& Comm)
      : Epetra_BlockMap(NumGlobalElements, NumMyElements, MyGlobalElements, IndexBase, Comm) {}
};
#endif
