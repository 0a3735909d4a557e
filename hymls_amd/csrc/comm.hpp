// comm.hpp -- rank-to-rank exchange of the sharded path (one process per GPU).
//
// Replaces the Epetra_Import / Epetra_Export objects of the reference (importer of the
// overlapping matrix and vectors, src/HYMLS_Preconditioner.cpp:304-330; Schur-complement
// export, src/HYMLS_SchurComplement.cpp:195-260; vsumImporter of the next level,
// src/HYMLS_SchurPreconditioner.cpp:520-629).  The library itself only packs and unpacks; the
// transport is supplied by the host application through two callbacks (the Python mirror
// implements them with torch.distributed: RCCL over xGMI on the GPUs, gloo in the CPU tests).
#pragma once
#include <functional>
#include "common.hpp"

namespace hymls {

struct Comm {
  int rank = 0, size = 1;
  int px = 1, py = 1, pz = 1;   // rank grid: rank = (rz * py + ry) * px + rx owns one box of the global grid
  void* ctx = nullptr;
  // all-to-all of contiguous per-peer segments; counts in elements of elem_bytes.
  // on_device != 0: both pointers lie inside arenas obtained from `alloc` (device memory, ordered on the
  // library's stream); otherwise host memory.
  int (*alltoallv)(void* ctx, const void* send, const int64_t* scnt, void* recv, const int64_t* rcnt,
                   int32_t elem_bytes, int32_t on_device) = nullptr;
  // exchange arena the transport can address (a torch tensor in the Python mirror)
  void* (*alloc)(void* ctx, int64_t bytes) = nullptr;

  // built-in RCCL transport (comm_rccl.cpp): ctx is owned by the communicator and released through release_fn
  void (*release_fn)(void* ctx) = nullptr;
  bool native = false;
  void release();               // free what the transport owns (arenas, staging buffers, its ncclComm); idempotent

  bool force = false;           // treat a single rank as sharded (exercises the transport with self-exchanges)
  bool distributed() const { return size > 1 || force; }

  // ---- host helpers (setup time)
  void a2a_host(const void* send, const std::vector<int64_t>& scnt, void* recv, const std::vector<int64_t>& rcnt,
                int elem_bytes) const;
  std::vector<int64_t> exchange_counts(const std::vector<int64_t>& scnt) const;
  // out[q] goes to rank q; returns in[q] = what rank q sent to me
  template <class T>
  std::vector<std::vector<T>> exchange_lists(const std::vector<std::vector<T>>& out) const {
    std::vector<int64_t> sc(size), rc;
    std::vector<T> sbuf;
    for (int q = 0; q < size; q++) { sc[q] = (int64_t)out[q].size(); sbuf.insert(sbuf.end(), out[q].begin(), out[q].end()); }
    rc = exchange_counts(sc);
    int64_t nr = 0;
    for (int64_t c : rc) nr += c;
    std::vector<T> rbuf((size_t)std::max<int64_t>(nr, 1));
    if (sbuf.empty()) sbuf.resize(1);
    a2a_host(sbuf.data(), sc, rbuf.data(), rc, (int)sizeof(T));
    std::vector<std::vector<T>> in(size);
    int64_t off = 0;
    for (int q = 0; q < size; q++) { in[q].assign(rbuf.begin() + off, rbuf.begin() + off + rc[q]); off += rc[q]; }
    return in;
  }
  // concatenation over ranks (rank order) of `mine`; counts[q] = contribution of rank q
  template <class T, class A>
  std::vector<T, A> allgather(const std::vector<T, A>& mine, std::vector<int64_t>* counts = nullptr) const {
    if (!distributed()) { if (counts) counts->assign(1, (int64_t)mine.size()); return mine; }
    // all-to-all with the same segment for every peer: the receive buffer IS the concatenation in rank order
    std::vector<int64_t> sc(size, (int64_t)mine.size());
    std::vector<int64_t> rc = exchange_counts(sc);
    int64_t nr = 0;
    for (int64_t c : rc) nr += c;
    std::vector<T, A> sbuf;
    sbuf.reserve(std::max<size_t>(1, mine.size() * (size_t)size));
    for (int q = 0; q < size; q++) sbuf.insert(sbuf.end(), mine.begin(), mine.end());
    if (sbuf.empty()) sbuf.resize(1);
    std::vector<T, A> all((size_t)std::max<int64_t>(nr, 1));
    a2a_host(sbuf.data(), sc, all.data(), rc, (int)sizeof(T));
    all.resize((size_t)nr);
    if (counts) *counts = rc;
    return all;
  }
  int64_t allsum(int64_t v) const;
  // element-wise sum over the ranks, added in rank order (the same bits on every rank)
  void allsum(std::vector<double>& v) const;

  // ---- device arenas (grown on demand; shared by every exchange of this communicator)
  double* send_arena(int64_t doubles) const;
  double* recv_arena(int64_t doubles) const;

 private:
  mutable double *sarena_ = nullptr, *rarena_ = nullptr;
  mutable int64_t scap_ = 0, rcap_ = 0;
};

// built-in transport on RCCL (product library only; the test-only host simulator has stubs that fail with -99)
void rccl_unique_id(char* id128);                       // ncclGetUniqueId (call on one rank, hand the bytes to the others)
void* rccl_init(const char* id128, int rank, int size, int device); // ncclCommInitRank on `device` -> ncclComm_t (collective)
void rccl_destroy(void* nccl_comm);
void rccl_attach(Comm& c, void* nccl_comm, bool owns);  // point c's callbacks at RCCL send/recv groups on the handle's stream
const char* rccl_last_error(const Comm& c);

// persistent plan of one vector exchange: dst[ridx[k]] on this rank = src[sidx[...]] on the peer
struct Exchange {
  const Comm* comm = nullptr;
  std::vector<int64_t> scnt, rcnt;
  int64_t nsend = 0, nrecv = 0;
  bool any = false;               // some rank sends something: everybody has to enter the collective
  int32_t* d_sidx = nullptr;      // [nsend] positions in the source vector
  int32_t* d_ridx = nullptr;      // [nrecv] positions in the destination vector
  ivec h_sidx, h_ridx;
  ~Exchange();
  Exchange() = default;
  Exchange(const Exchange&) = delete;
  Exchange& operator=(const Exchange&) = delete;
  // want_keys[q]: keys wanted from rank q, want_dst[q]: where each lands in my destination vector;
  // resolve(key): position of that entry in the source vector of the rank that holds it (called there)
  void build(const Comm& c, const std::vector<std::vector<int64_t>>& want_keys, const std::vector<ivec>& want_dst,
             const std::function<int32_t(int64_t)>& resolve);
  void forward(const double* src, double* dst) const;    // dst[ridx] <- peer src[sidx]
  void backward(const double* src, double* dst, bool add = false) const;   // reverse direction: dst[sidx] <- (or +=) peer src[ridx]
};

}  // namespace hymls
