// mock: Epetra_MpiComm (Trilinos packages/epetra/src/Epetra_MpiComm.h): the communicator of a distributed Epetra
// application, as the reference constructs it in src/main.cpp:48-67
#ifndef MOCK_EPETRA_MPICOMM_H
#define MOCK_EPETRA_MPICOMM_H
#include <mpi.h>
#include "Epetra_Comm.h"
class Epetra_MpiComm : public Epetra_Comm {
 public:
  explicit Epetra_MpiComm(MPI_Comm comm) : comm_(comm) { MPI_Comm_rank(comm, &rank_); MPI_Comm_size(comm, &size_); }
  MPI_Comm Comm() const { return comm_; }
  MPI_Comm GetMpiComm() const { return comm_; }
  int MyPID() const { return rank_; }
  int NumProc() const { return size_; }
  void Barrier() const { MPI_Barrier(comm_); }
  int SumAll(double* p, double* g, int n) const { return MPI_Allreduce(p, g, n, MPI_DOUBLE, MPI_SUM, comm_); }
  int SumAll(int* p, int* g, int n) const { return MPI_Allreduce(p, g, n, MPI_INT, MPI_SUM, comm_); }
  int MaxAll(int* p, int* g, int n) const { return MPI_Allreduce(p, g, n, MPI_INT, MPI_MAX, comm_); }
  int MinAll(int* p, int* g, int n) const { return MPI_Allreduce(p, g, n, MPI_INT, MPI_MIN, comm_); }
  void MockAllgatherv(const void* mine, int nbytes, std::vector<char>& all, std::vector<int>& counts) const {
    counts.assign(size_, 0);
    MPI_Allgather(&nbytes, 1, MPI_INT, counts.data(), 1, MPI_INT, comm_);
    std::vector<int> displ(size_, 0);
    int total = 0;
    for (int q = 0; q < size_; q++) { displ[q] = total; total += counts[q]; }
    all.resize(total > 0 ? total : 1);
    MPI_Allgatherv(const_cast<void*>(mine), nbytes, MPI_BYTE, all.data(), counts.data(), displ.data(), MPI_BYTE, comm_);
    all.resize(total);
  }
  void MockAlltoallv(const std::vector<std::vector<char> >& send, std::vector<std::vector<char> >& recv) const {
    std::vector<int> sc(size_), rc(size_), sd(size_), rd(size_);
    for (int q = 0; q < size_; q++) sc[q] = (int)send[q].size();
    MPI_Alltoall(sc.data(), 1, MPI_INT, rc.data(), 1, MPI_INT, comm_);
    int ns = 0, nr = 0;
    for (int q = 0; q < size_; q++) { sd[q] = ns; rd[q] = nr; ns += sc[q]; nr += rc[q]; }
    std::vector<char> sb(ns > 0 ? ns : 1), rb(nr > 0 ? nr : 1);
    for (int q = 0; q < size_; q++) if (sc[q]) std::memcpy(sb.data() + sd[q], send[q].data(), sc[q]);
    MPI_Alltoallv(sb.data(), sc.data(), sd.data(), MPI_BYTE, rb.data(), rc.data(), rd.data(), MPI_BYTE, comm_);
    recv.assign(size_, std::vector<char>());
    for (int q = 0; q < size_; q++) recv[q].assign(rb.begin() + rd[q], rb.begin() + rd[q] + rc[q]);
  }
 private:
  MPI_Comm comm_;
  int rank_, size_;
};
#endif
