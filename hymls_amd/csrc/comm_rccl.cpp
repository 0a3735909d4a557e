// comm_rccl.cpp -- the built-in transport of the sharded path: RCCL point-to-point groups over xGMI.
//
// Replaces the Epetra_MpiComm + Epetra_Import/Export traffic of the reference (the overlapping-map importer,
// src/HYMLS_Preconditioner.cpp:304-330,978-979,1050-1052; the Schur-complement export, src/HYMLS_SchurComplement.cpp:
// 195-260; the V-sum importer of the next level, src/HYMLS_SchurPreconditioner.cpp:520-629,1076-1078).  Every exchange
// of the library is an all-to-all of contiguous per-peer segments (comm.hpp); here one such exchange is ONE
// ncclGroupStart / ncclSend.. / ncclRecv.. / ncclGroupEnd on the handle's stream: with the 2x2x2 boxes of 8 GPUs every
// rank talks to at most 7 peers = one xGMI link each, messages of a few hundred kB (latency-bound, not bandwidth-bound).
// No host thread, no Python frame and no synchronisation sits inside ApplyInverse; the segment of a rank to itself is
// a device-to-device copy.  Host-side setup exchanges (index lists, counts, the reduced matrix at Compute) are staged
// through device buffers and use the same groups.
//
// librccl is opened at first use (dlopen: the copy the process already holds, e.g. PyTorch's, else the ROCm one), so
// single-GPU users and profilers never load it.
#include <dlfcn.h>
#include <cstring>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include "comm.hpp"
#include "device.hpp"

namespace hymls {

namespace {

struct Api {
  void* lib = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclCommCount) CommCount = nullptr;
  decltype(&ncclCommUserRank) CommUserRank = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

Api& api() {
  static Api a;
  if (a.lib) return a;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char* n : names) { a.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD); if (a.lib) break; }
  if (!a.lib) for (const char* n : names) { a.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (a.lib) break; }
  HYMLS_CHECK(a.lib != nullptr, -3, "librccl.so not found: the built-in multi-GPU transport needs RCCL");
  auto sym = [&](const char* name) { void* p = dlsym(a.lib, name); HYMLS_CHECK(p != nullptr, -3, std::string("librccl lacks ") + name); return p; };
  a.GetUniqueId = (decltype(a.GetUniqueId))sym("ncclGetUniqueId");
  a.CommInitRank = (decltype(a.CommInitRank))sym("ncclCommInitRank");
  a.CommDestroy = (decltype(a.CommDestroy))sym("ncclCommDestroy");
  a.CommCount = (decltype(a.CommCount))sym("ncclCommCount");
  a.CommUserRank = (decltype(a.CommUserRank))sym("ncclCommUserRank");
  a.GroupStart = (decltype(a.GroupStart))sym("ncclGroupStart");
  a.GroupEnd = (decltype(a.GroupEnd))sym("ncclGroupEnd");
  a.Send = (decltype(a.Send))sym("ncclSend");
  a.Recv = (decltype(a.Recv))sym("ncclRecv");
  a.GetErrorString = (decltype(a.GetErrorString))sym("ncclGetErrorString");
  return a;
}

#define NCCL_OK(call) do { ncclResult_t r_ = (call); if (r_ != ncclSuccess) { T->error = std::string("RCCL: ") + api().GetErrorString(r_); return -1; } } while (0)
#define HIP_OK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { T->error = std::string("HIP: ") + hipGetErrorString(e_); return -1; } } while (0)

struct RcclTransport {
  ncclComm_t comm = nullptr;
  bool owns = false;
  int rank = 0, size = 1;
  std::vector<void*> arenas;      // exchange arenas handed to the library (alloc callback)
  char *stage_s = nullptr, *stage_r = nullptr;   // device staging of host-side exchanges
  size_t cap_s = 0, cap_r = 0;
  std::string error;
};

int grow_stage(RcclTransport* T, char*& buf, size_t& cap, size_t need) {
  if (need <= cap) return 0;
  if (buf) (void)hipFree(buf);
  buf = nullptr; cap = 0;
  const size_t want = need + need / 4 + 4096;
  HIP_OK(hipMalloc((void**)&buf, want));
  cap = want;
  return 0;
}

// one exchange = one group; bytes, device pointers, ordered on `stream`
int group_exchange(RcclTransport* T, const char* send, const int64_t* scnt, char* recv, const int64_t* rcnt, int64_t eb, hipStream_t stream) {
  int64_t so = 0, ro = 0, self_so = -1, self_ro = -1;
  bool any = false;
  for (int q = 0; q < T->size; q++) any |= (q != T->rank) && (scnt[q] > 0 || rcnt[q] > 0);
  if (any) NCCL_OK(api().GroupStart());
  // an error inside the open group must not leave the communicator in group mode (every later exchange would hang or
  // fail opaquely): remember the first failure, skip the rest, close the group, then report
  ncclResult_t first = ncclSuccess;
  for (int q = 0; q < T->size; q++) {
    if (q == T->rank) { self_so = so; self_ro = ro; }
    else if (first == ncclSuccess) {
      if (scnt[q] > 0) first = api().Send(send + so * eb, (size_t)(scnt[q] * eb), ncclInt8, q, T->comm, stream);
      if (first == ncclSuccess && rcnt[q] > 0) first = api().Recv(recv + ro * eb, (size_t)(rcnt[q] * eb), ncclInt8, q, T->comm, stream);
    }
    so += scnt[q]; ro += rcnt[q];
  }
  if (any) {
    const ncclResult_t end = api().GroupEnd();
    if (first == ncclSuccess) first = end;
  }
  if (first != ncclSuccess) { T->error = std::string("RCCL: ") + api().GetErrorString(first); return -1; }
  const int64_t ns = scnt[T->rank];
  if (ns != rcnt[T->rank]) { T->error = "all-to-all: a rank's segment to itself differs in its send and receive counts"; return -1; }
  if (ns > 0) HIP_OK(hipMemcpyAsync(recv + self_ro * eb, send + self_so * eb, (size_t)(ns * eb), hipMemcpyDeviceToDevice, stream));
  return 0;
}

int rccl_alltoallv(void* ctx, const void* send, const int64_t* scnt, void* recv, const int64_t* rcnt, int32_t eb, int32_t on_device) {
  RcclTransport* T = (RcclTransport*)ctx;
  try {
    hipStream_t stream = (hipStream_t)dev::stream();
    if (on_device) return group_exchange(T, (const char*)send, scnt, (char*)recv, rcnt, eb, stream);
    int64_t ns = 0, nr = 0;
    for (int q = 0; q < T->size; q++) { ns += scnt[q]; nr += rcnt[q]; }
    if (grow_stage(T, T->stage_s, T->cap_s, (size_t)std::max<int64_t>(ns * eb, 8))) return -1;
    if (grow_stage(T, T->stage_r, T->cap_r, (size_t)std::max<int64_t>(nr * eb, 8))) return -1;
    if (ns > 0) HIP_OK(hipMemcpyAsync(T->stage_s, send, (size_t)(ns * eb), hipMemcpyHostToDevice, stream));
    if (group_exchange(T, T->stage_s, scnt, T->stage_r, rcnt, eb, stream)) return -1;
    if (nr > 0) HIP_OK(hipMemcpyAsync(recv, T->stage_r, (size_t)(nr * eb), hipMemcpyDeviceToHost, stream));
    HIP_OK(hipStreamSynchronize(stream));
    return 0;
  } catch (const std::exception& e) { T->error = e.what(); return -1; }
}

void* rccl_alloc(void* ctx, int64_t bytes) {
  RcclTransport* T = (RcclTransport*)ctx;
  void* p = nullptr;
  if (hipMalloc(&p, (size_t)std::max<int64_t>(bytes, 8)) != hipSuccess) { T->error = "hipMalloc of an exchange arena failed"; return nullptr; }
  T->arenas.push_back(p);
  return p;
}

void rccl_release(void* ctx) {
  RcclTransport* T = (RcclTransport*)ctx;
  if (!T) return;
  for (void* p : T->arenas) (void)hipFree(p);
  if (T->stage_s) (void)hipFree(T->stage_s);
  if (T->stage_r) (void)hipFree(T->stage_r);
  if (T->owns && T->comm) (void)api().CommDestroy(T->comm);
  delete T;
}

}  // namespace

void rccl_unique_id(char* id128) {
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  ncclUniqueId id;
  const ncclResult_t r = api().GetUniqueId(&id);
  HYMLS_CHECK(r == ncclSuccess, -3, std::string("ncclGetUniqueId: ") + api().GetErrorString(r));
  std::memcpy(id128, &id, 128);
}

void* rccl_init(const char* id128, int rank, int size, int device) {
  const hipError_t e = hipSetDevice(device);
  HYMLS_CHECK(e == hipSuccess, -3, std::string("hipSetDevice: ") + hipGetErrorString(e));
  ncclUniqueId id;
  std::memcpy(&id, id128, 128);
  ncclComm_t c = nullptr;
  const ncclResult_t r = api().CommInitRank(&c, size, id, rank);
  HYMLS_CHECK(r == ncclSuccess, -3, std::string("ncclCommInitRank: ") + api().GetErrorString(r));
  return (void*)c;
}

void rccl_destroy(void* nccl_comm) { if (nccl_comm) (void)api().CommDestroy((ncclComm_t)nccl_comm); }

void rccl_attach(Comm& c, void* nccl_comm, bool owns) {
  HYMLS_CHECK(nccl_comm != nullptr, -2, "null ncclComm_t");
  RcclTransport* T = new RcclTransport();
  T->comm = (ncclComm_t)nccl_comm; T->owns = owns;
  ncclResult_t r = api().CommCount(T->comm, &T->size);
  if (r == ncclSuccess) r = api().CommUserRank(T->comm, &T->rank);
  if (r != ncclSuccess) { delete T; throw Error(-3, std::string("ncclCommCount/UserRank: ") + api().GetErrorString(r)); }
  c.release();
  c.rank = T->rank; c.size = T->size; c.ctx = T;
  c.alltoallv = rccl_alltoallv; c.alloc = rccl_alloc; c.release_fn = rccl_release;
  c.native = true;
}

const char* rccl_last_error(const Comm& c) { return (c.native && c.ctx) ? ((RcclTransport*)c.ctx)->error.c_str() : ""; }

}  // namespace hymls
