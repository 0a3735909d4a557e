// mock: Epetra_Import(TargetMap, SourceMap) -- the communication plan that brings the entries of a one-to-one source
// map to a (possibly overlapping) target map; used forward by Import() and in reverse by Export(), as in Epetra
// (reference src/HYMLS_Preconditioner.cpp:326-336 builds its importer the same way)
#ifndef MOCK_EPETRA_IMPORT_H
#define MOCK_EPETRA_IMPORT_H
#include <stdexcept>
#include <unordered_map>
#include <vector>
#include "Epetra_Map.h"
enum Epetra_CombineMode { Add, Zero, Insert, InsertAdd, Average, AbsMax };
class Epetra_Import {
 public:
  Epetra_Import(const Epetra_BlockMap& TargetMap, const Epetra_BlockMap& SourceMap) : target_(TargetMap), source_(SourceMap) {
    const Epetra_Comm& comm = SourceMap.Comm();
    const int P = comm.NumProc();
    // directory of the source map (small test problems: every rank learns every source gid)
    std::vector<char> all; std::vector<int> counts;
    comm.MockAllgatherv(SourceMap.MyGlobalElements(), SourceMap.NumMyElements() * (int)sizeof(int), all, counts);
    std::unordered_map<int, std::pair<int, int> > owner;
    { const int* g = (const int*)all.data(); int off = 0;
      for (int q = 0; q < P; q++) { const int n = counts[q] / (int)sizeof(int); for (int i = 0; i < n; i++) owner[g[off + i]] = std::make_pair(q, i); off += n; } }
    std::vector<std::vector<char> > ask(P), asked;
    recv_from_.assign(P, std::vector<int>());
    for (int l = 0; l < TargetMap.NumMyElements(); l++) {
      auto it = owner.find(TargetMap.GID(l));
      if (it == owner.end()) throw std::runtime_error("Epetra_Import (mock): target GID not in the source map");
      const int q = it->second.first, sl = it->second.second;
      recv_from_[q].push_back(l);
      ask[q].insert(ask[q].end(), (const char*)&sl, (const char*)&sl + sizeof(int));
    }
    comm.MockAlltoallv(ask, asked);
    send_to_.assign(P, std::vector<int>());
    for (int q = 0; q < P; q++) send_to_[q].assign((const int*)asked[q].data(), (const int*)(asked[q].data() + asked[q].size()));
  }
  const Epetra_BlockMap& TargetMap() const { return target_; }
  const Epetra_BlockMap& SourceMap() const { return source_; }
  // mock plan: send_to()[q] = source lids whose entries go to rank q, recv_from()[q] = target lids they land in there
  const std::vector<std::vector<int> >& send_to() const { return send_to_; }
  const std::vector<std::vector<int> >& recv_from() const { return recv_from_; }
 private:
  Epetra_BlockMap target_, source_;
  std::vector<std::vector<int> > send_to_, recv_from_;
};
#endif
