// mock: the members of Epetra_Comm the adapter and its test drivers use.  The collectives are the real Epetra ones
// (same names and argument order); Mock* are helpers of these stand-ins only (real Epetra does this with
// Epetra_Distributor / Epetra_Directory objects).
#ifndef MOCK_EPETRA_COMM_H
#define MOCK_EPETRA_COMM_H
#include <cstring>
#include <vector>
class Epetra_Comm {
 public:
  virtual ~Epetra_Comm() {}
  virtual int MyPID() const = 0;
  virtual int NumProc() const = 0;
  virtual void Barrier() const = 0;
  virtual int SumAll(double* PartialSums, double* GlobalSums, int Count) const = 0;
  virtual int SumAll(int* PartialSums, int* GlobalSums, int Count) const = 0;
  virtual int MaxAll(int* PartialMaxs, int* GlobalMaxs, int Count) const = 0;
  virtual int MinAll(int* PartialMins, int* GlobalMins, int Count) const = 0;
  // concatenation of every rank's bytes in rank order; counts[q] = bytes of rank q
  virtual void MockAllgatherv(const void* mine, int nbytes, std::vector<char>& all, std::vector<int>& counts) const = 0;
  // byte segments: send[q] goes to rank q, recv[q] came from rank q
  virtual void MockAlltoallv(const std::vector<std::vector<char> >& send, std::vector<std::vector<char> >& recv) const = 0;
};
class Epetra_SerialComm : public Epetra_Comm {
 public:
  int MyPID() const { return 0; }
  int NumProc() const { return 1; }
  void Barrier() const {}
  int SumAll(double* p, double* g, int n) const { std::memcpy(g, p, n * sizeof(double)); return 0; }
  int SumAll(int* p, int* g, int n) const { std::memcpy(g, p, n * sizeof(int)); return 0; }
  int MaxAll(int* p, int* g, int n) const { std::memcpy(g, p, n * sizeof(int)); return 0; }
  int MinAll(int* p, int* g, int n) const { std::memcpy(g, p, n * sizeof(int)); return 0; }
  void MockAllgatherv(const void* mine, int nbytes, std::vector<char>& all, std::vector<int>& counts) const {
    all.assign((const char*)mine, (const char*)mine + nbytes); counts.assign(1, nbytes);
  }
  void MockAlltoallv(const std::vector<std::vector<char> >& send, std::vector<std::vector<char> >& recv) const { recv = send; }
};
#endif
